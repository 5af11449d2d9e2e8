#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh'): rocprofv3 kernel statistics of the bench command and two
# separate PMC passes (FETCH_SIZE, WRITE_SIZE) over a forward+backward render of the bench scene.  Results land under
# gpurun_out/ (scratch); tools/pmc_traffic.py turns them into profiles/<round>/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/profiles
rm -rf "$OUT"; mkdir -p "$OUT"   # NB: also delete gpurun_out/profiles locally before the call (gpurun merges, it does not mirror)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/bench.py" --steps 20 --warmup 5 \
    > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof_stats.err" || exit 1
timeout -k 10 300 python3 "$R/bench.py" --steps 20 --warmup 5 > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err" || exit 1
for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/pmc_$ctr" -- python3 "$R/tools/fwd_once.py" 6000000 2 step \
        > "$OUT/pmc_$ctr.log" 2>&1 || exit 1
done
find "$OUT" -name "*.csv" | head -20
