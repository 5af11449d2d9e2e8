set -o pipefail
mkdir -p gpurun_out
for w in c1_1k_128x128 lego_like_300k_800x800 scannetpp_like_fisheye_300k_1752x1168 garden_like_5M_1297x840; do
  python bench.py --workload $w --steps 40 --warmup 12 > gpurun_out/wl_$w.json 2> gpurun_out/wl_$w.err || exit 1
done
python bench.py --steps 40 --warmup 12 --no-sensitivity --no-cpu-baseline --no-overlap-optimizer > gpurun_out/wl_bicycle_onepass.json 2> gpurun_out/wl_bicycle_onepass.err || exit 1
python bench.py --steps 40 --warmup 12 --no-sensitivity --no-cpu-baseline > gpurun_out/wl_bicycle_default.json 2> gpurun_out/wl_bicycle_default.err || exit 1
echo done
