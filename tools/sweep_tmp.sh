set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_native.py tests/test_gpu_dp.py -x -q -m gpu > gpurun_out/r2_t19.log 2>&1; echo rc=$? >> gpurun_out/r2_t19.log; tail -15 gpurun_out/r2_t19.log
grep -q "rc=0" gpurun_out/r2_t19.log || exit 1
for ex in sparse dense; do
  python bench.py --steps 20 --warmup 8 --no-sensitivity --no-cpu-baseline --force-exchange --dp-exchange $ex > gpurun_out/fx_$ex.json 2> gpurun_out/fx_$ex.err || exit 1
done
GUT_BENCH_SHARE_GPU=1 GUT_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 6 --warmup 3 --num-gaussians 1500000 --no-sensitivity --no-cpu-baseline > gpurun_out/reh_sparse.json 2> gpurun_out/reh_sparse.err || exit 1
echo done
