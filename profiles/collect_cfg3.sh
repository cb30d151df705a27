#!/bin/bash
# Run ON the GPU box (through gpurun) from the repo root:  bash profiles/collect_cfg3.sh r04
# configs[2] (LWW Map, 2,048 documents of the config's 10,000 — the counters are per launch): three separate rocprofv3 passes of the
# same command (kernel trace + stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE), then profiles/extract_cfg3.py -> profiles/<tag>_pmc_configs2.json
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/c3_stats $R/gpurun_out/c3_fetch $R/gpurun_out/c3_write
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c3_stats -o $TAG -- python $R/tests/tools/gpu_cfg.py cfg3 2048 8 > $R/gpurun_out/${TAG}_c3_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/c3_fetch -o $TAG -- python $R/tests/tools/gpu_cfg.py cfg3 2048 8 > $R/gpurun_out/${TAG}_c3_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/c3_write -o $TAG -- python $R/tests/tools/gpu_cfg.py cfg3 2048 8 > $R/gpurun_out/${TAG}_c3_write.log 2>&1
cd $R && python3 profiles/extract_cfg3.py gpurun_out/c3_stats/${TAG}_results.db gpurun_out/c3_fetch/${TAG}_results.db gpurun_out/c3_write/${TAG}_results.db $TAG
mkdir -p gpurun_out/profiles && cp profiles/${TAG}_pmc_configs2.json gpurun_out/profiles/
