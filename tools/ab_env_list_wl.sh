# Dev helper (GPU box): one bench workload under a list of environment settings.  usage: bash tools/ab_env_list_wl.sh TAG WORKLOAD "A=1" "X=" ...
tag=$1; wl=$2; shift; shift
mkdir -p gpurun_out/r4
k=0
for envs in "$@"; do
  k=$((k+1))
  env $envs python bench.py --workload $wl --no-sensitivity --no-drop-in --no-cpu-baseline > gpurun_out/r4/env_${tag}_${k}.json 2> gpurun_out/r4/env_${tag}.err
done
