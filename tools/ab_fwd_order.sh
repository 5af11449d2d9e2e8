mkdir -p gpurun_out/r4
for i in 1 2; do
  for v in 0 1; do
    GUT_FWD_TILE_ORDER=$v python bench.py --no-sensitivity --no-drop-in --no-cpu-baseline > gpurun_out/r4/ab_order_h_${v}_${i}.json 2> gpurun_out/r4/ab.err
    GUT_FWD_TILE_ORDER=$v python bench.py --workload bicycle_like_6M_surface --no-drop-in --no-cpu-baseline > gpurun_out/r4/ab_order_s_${v}_${i}.json 2>> gpurun_out/r4/ab.err
  done
done
