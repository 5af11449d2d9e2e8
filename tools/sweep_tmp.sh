set -o pipefail
mkdir -p gpurun_out
for i in 1 2; do
for v in 0_4 1_4 2_4 2_2 2_3; do
  GUT_HIP_LIB=$PWD/3dgrut_amd/libgut_hip_v$v.so python bench.py --steps 40 --warmup 12 --no-sensitivity --no-cpu-baseline > gpurun_out/ev_${v}_$i.json 2> gpurun_out/ev_${v}_$i.err || exit 1
done
done
echo done
