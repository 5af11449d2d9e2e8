set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_native.py tests/test_gpu_dp.py -x -q -m gpu > gpurun_out/r2_t22.log 2>&1; echo rc=$? >> gpurun_out/r2_t22.log; tail -5 gpurun_out/r2_t22.log
grep -q "rc=0" gpurun_out/r2_t22.log || exit 1
python bench.py --steps 20 --warmup 8 --no-sensitivity --no-cpu-baseline --force-exchange > gpurun_out/fx_side.json 2> gpurun_out/fx_side.err || exit 1
python bench.py --steps 20 --warmup 8 --no-sensitivity --no-cpu-baseline --force-exchange --no-dp-side-stream > gpurun_out/fx_noside.json 2> gpurun_out/fx_noside.err || exit 1
GUT_BENCH_SHARE_GPU=1 GUT_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 6 --warmup 3 --num-gaussians 1500000 --no-sensitivity --no-cpu-baseline > gpurun_out/reh_sparse.json 2> gpurun_out/reh_sparse.err || exit 1
echo done
