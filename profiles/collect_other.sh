#!/bin/bash
# Run ON the GPU box (through gpurun) from the repo root:  bash profiles/collect_other.sh r06 ["configs[2]" "configs[3]" ...]
# Every other_configs entry of bench.py in a process of its own (tests/tools/gpu_one_config.py), three separate rocprofv3 passes each
# (kernel trace + stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE; never combined), then profiles/extract_other.py -> profiles/<tag>_pmc_other.json
TAG=${1:-r06}
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
if [ $# -eq 0 ]; then set -- "configs[0]" "configs[2]" "configs[3]" "configs[1]-heterogeneous" "configs[1]-128-traces" "configs[4]" "movable-lists (SURVEY 8f N4)"; fi
cd /tmp && export TMPDIR=/tmp
export LM_BENCH_ONLY_EXACT=1
i=0
for NAME in "$@"; do
  i=$((i+1))
  for P in stats fetch write; do
    rm -rf $R/gpurun_out/oc_${i}_$P
    case $P in
      stats) ARGS="--kernel-trace --stats";;
      fetch) ARGS="--pmc FETCH_SIZE";;
      write) ARGS="--pmc WRITE_SIZE";;
    esac
    timeout 900 rocprofv3 $ARGS -d $R/gpurun_out/oc_${i}_$P -o $TAG -- python $R/tests/tools/gpu_one_config.py "$NAME" > $R/gpurun_out/${TAG}_oc_${i}_$P.log 2>&1
  done
  (cd $R && python3 profiles/extract_other.py $TAG "$NAME" gpurun_out/oc_${i}_stats/${TAG}_results.db gpurun_out/oc_${i}_fetch/${TAG}_results.db gpurun_out/oc_${i}_write/${TAG}_results.db gpurun_out/${TAG}_oc_${i}_stats.log)
done
mkdir -p $R/gpurun_out/profiles && cp $R/profiles/${TAG}_pmc_other.json $R/gpurun_out/profiles/
cat $R/profiles/${TAG}_pmc_other.json | head -c 3000
