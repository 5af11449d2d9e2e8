set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r2_full.log 2>&1; echo rc=$? >> gpurun_out/r2_full.log; tail -3 gpurun_out/r2_full.log
grep -q "rc=0" gpurun_out/r2_full.log || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_smoke.log 2>&1; echo rc=$? >> gpurun_out/r2_smoke.log; tail -2 gpurun_out/r2_smoke.log
for w in c1_1k_128x128 lego_like_300k_800x800 scannetpp_like_fisheye_300k_1752x1168 garden_like_5M_1297x840; do
  python bench.py --workload $w --steps 40 --warmup 12 > gpurun_out/wl_$w.json 2> gpurun_out/wl_$w.err || exit 1
done
bash tools/collect_profiles.sh
