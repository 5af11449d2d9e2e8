# Dev helper (GPU box): headline (twice) and surface leg (once) with two builds of the library, interleaved.  usage: bash tools/ab_lib_short.sh OTHER.so TAG
mkdir -p gpurun_out/r4
for i in 1 2; do
  for v in new prev; do
    if [ $v = prev ]; then export GUT_HIP_LIBRARY=$1; else unset GUT_HIP_LIBRARY; fi
    python bench.py --no-sensitivity --no-drop-in --no-cpu-baseline > gpurun_out/r4/ab_$2_h_${v}_${i}.json 2> gpurun_out/r4/ab_$2.err
    if [ $i = 1 ]; then python bench.py --workload bicycle_like_6M_surface --no-sensitivity --no-drop-in --no-cpu-baseline > gpurun_out/r4/ab_$2_s_${v}_${i}.json 2>> gpurun_out/r4/ab_$2.err; fi
  done
done
