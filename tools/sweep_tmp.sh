set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_native.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r2_t23.log 2>&1; echo rc=$? >> gpurun_out/r2_t23.log; tail -5 gpurun_out/r2_t23.log
grep -q "rc=0" gpurun_out/r2_t23.log || exit 1
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 40 --warmup 12 --no-sensitivity --no-cpu-baseline > gpurun_out/dq_dyn_$i.json 2> gpurun_out/dq_dyn_$i.err || exit 1
  GUT_EARLY_SPLIT=25 timeout -k 10 200 python bench.py --steps 40 --warmup 12 --no-sensitivity --no-cpu-baseline > gpurun_out/dq_s25_$i.json 2> gpurun_out/dq_s25_$i.err || exit 1
done
echo done
