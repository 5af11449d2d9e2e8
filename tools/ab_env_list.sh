# Dev helper (GPU box): the bench's headline leg under a list of environment settings, in the order given, twice around.
# usage: bash tools/ab_env_list.sh TAG "A=1 B=2" "A=0" ...   (an empty string = no setting)
tag=$1; shift
mkdir -p gpurun_out/r4
for i in 1 2; do
  k=0
  for envs in "$@"; do
    k=$((k+1))
    env $envs python bench.py --no-sensitivity --no-drop-in --no-cpu-baseline > gpurun_out/r4/env_${tag}_${k}_${i}.json 2> gpurun_out/r4/env_${tag}.err
  done
done
