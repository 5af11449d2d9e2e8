# Dev helper (GPU box): one bench line per named workload (BASELINE.md §2's table).
mkdir -p gpurun_out/r4
for w in c1_1k_128x128 lego_like_300k_800x800 scannetpp_like_fisheye_300k_1752x1168 garden_like_5M_1297x840; do
  python bench.py --workload $w > gpurun_out/r4/wl_$w.json 2> gpurun_out/r4/wl_$w.err; echo "$w rc $?"
done
python bench.py --force-exchange --no-sensitivity --no-drop-in --no-cpu-baseline > gpurun_out/r4/wl_force_exchange.json 2> gpurun_out/r4/wl_force_exchange.err; echo "force-exchange rc $?"
