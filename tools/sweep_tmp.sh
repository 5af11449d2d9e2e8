for w in c1_1k_128x128 lego_like_300k_800x800 scannetpp_like_fisheye_300k_1752x1168; do
python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-overlap-optimizer > gpurun_out/wn_$w.json 2> gpurun_out/wn_$w.err
python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/wo_$w.json 2> gpurun_out/wo_$w.err
done
echo done
