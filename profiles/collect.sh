#!/bin/bash
# Run ON the GPU box (through gpurun) from the repo root:  bash profiles/collect.sh r01
# Three separate rocprofv3 passes of the same bench command (kernel trace + stats, FETCH_SIZE, WRITE_SIZE), written under
# gpurun_out/; profiles/extract_rocprof.py then turns the three databases into the summaries committed in profiles/.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o $TAG -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-end-to-end > $R/gpurun_out/${TAG}_bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o $TAG -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-end-to-end > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o $TAG -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-end-to-end > $R/gpurun_out/${TAG}_write.log 2>&1
cd $R && python3 profiles/extract_rocprof.py gpurun_out/prof_stats/${TAG}_results.db gpurun_out/prof_fetch/${TAG}_results.db gpurun_out/prof_write/${TAG}_results.db $TAG
mkdir -p gpurun_out/profiles && cp profiles/${TAG}_* gpurun_out/profiles/ && grep '^{"metric"' gpurun_out/${TAG}_bench_under_rocprof.log > gpurun_out/profiles/${TAG}_bench_under_rocprof.log
ls -la gpurun_out/profiles
