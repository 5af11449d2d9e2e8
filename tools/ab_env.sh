# Dev helper (GPU box): bench legs with / without an environment switch, interleaved.  usage: bash tools/ab_env.sh VAR TAG workload...
VAR=$1; TAG=$2; shift 2
mkdir -p gpurun_out/r4
for i in 1 2; do
  for v in off on; do
    for w in "$@"; do
      if [ $v = on ]; then export $VAR=1; else unset $VAR; fi
      python bench.py --workload $w --no-sensitivity --no-drop-in --no-cpu-baseline > gpurun_out/r4/ab_${TAG}_${w}_${v}_${i}.json 2> gpurun_out/r4/ab_${TAG}.err
    done
  done
done
