set -o pipefail
mkdir -p gpurun_out
for i in 1 2; do
for v in b1 b2 b4 b6; do
  GUT_HIP_LIB=$PWD/3dgrut_amd/libgut_hip_$v.so python bench.py --steps 40 --warmup 12 --no-sensitivity --no-cpu-baseline > gpurun_out/eb_${v}_$i.json 2> gpurun_out/eb_${v}_$i.err || exit 1
done
done
echo done
