set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r2_full.log 2>&1; echo rc=$? >> gpurun_out/r2_full.log; tail -3 gpurun_out/r2_full.log
grep -q "rc=0" gpurun_out/r2_full.log || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_smoke.log 2>&1; echo rc=$? >> gpurun_out/r2_smoke.log; tail -2 gpurun_out/r2_smoke.log
bash tools/collect_profiles.sh
