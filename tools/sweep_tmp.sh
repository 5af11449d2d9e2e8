set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_native.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r2_t18.log 2>&1; echo rc=$? >> gpurun_out/r2_t18.log; tail -3 gpurun_out/r2_t18.log
grep -q "rc=0" gpurun_out/r2_t18.log || exit 1
for i in 1 2 3; do
  GUT_HIP_LIB=$PWD/3dgrut_amd/libgut_hip_old.so python bench.py --steps 40 --warmup 12 --no-sensitivity > gpurun_out/dz_old_$i.json 2> gpurun_out/dz_old_$i.err || exit 1
  python bench.py --steps 40 --warmup 12 --no-sensitivity > gpurun_out/dz_new_$i.json 2> gpurun_out/dz_new_$i.err || exit 1
done
echo done
