set -o pipefail
mkdir -p gpurun_out
for i in 1 2 3; do
for v in g0 g1; do
  GUT_HIP_LIB=$PWD/3dgrut_amd/libgut_hip_$v.so python bench.py --steps 40 --warmup 12 --no-sensitivity --no-cpu-baseline > gpurun_out/wg_${v}_$i.json 2> gpurun_out/wg_${v}_$i.err || exit 1
done
done
GUT_HIP_LIB=$PWD/3dgrut_amd/libgut_hip_g1.so python -m pytest tests/test_gpu_native.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r2_t20.log 2>&1; echo rc=$? >> gpurun_out/r2_t20.log; tail -3 gpurun_out/r2_t20.log
echo done
