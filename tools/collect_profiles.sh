#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh'): rocprofv3 kernel statistics of the bench command and separate
# PMC passes over ten full native train steps of the bench scene (tools/fwd_once.py 6000000 10 step).  Every --pmc pass is its
# own rocprofv3 run with --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE takes 3 and WRITE_SIZE 2 of the 4 TCC slots, SQ
# has 8 slots, GRBM 2).  Results land under gpurun_out/profiles (scratch); tools/profile_report.py turns them into profiles/<round>/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/profiles
rm -rf "$OUT"; mkdir -p "$OUT"   # NB: also delete gpurun_out/profiles locally before the call (gpurun merges, it does not mirror)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-sensitivity \
    > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof_stats.err" || exit 1
echo "stats pass done"
timeout -k 10 300 python3 "$R/bench.py" --steps 20 --warmup 5 > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err" || exit 1
echo "plain bench done"
pass() {  # name, counters...
    local name=$1; shift
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 "$R/tools/fwd_once.py" 6000000 10 step \
        > "$OUT/pmc_$name.log" 2>&1 && echo "pmc pass $name done" || echo "pmc pass $name FAILED (see pmc_$name.log)"
}
pass FETCH_SIZE FETCH_SIZE
pass WRITE_SIZE WRITE_SIZE
pass SQ_A SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pass SQ_B SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS_ATOMIC
pass TCC TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum
find "$OUT" -name "*.csv" | head -30
