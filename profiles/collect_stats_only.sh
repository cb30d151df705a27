#!/bin/bash
# Run ON the GPU box (through gpurun) from the repo root:  bash profiles/collect_stats_only.sh r02_final
# The kernel-trace + stats pass of profiles/collect.sh alone (the two PMC passes are unchanged by host-side work);
# writes profiles/<tag>_kernel_stats.md straight from the rocpd database.
TAG=${1:-r02_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stats
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o $TAG -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-end-to-end > $R/gpurun_out/${TAG}_bench_under_rocprof.log 2>&1
cd $R && python3 - "$TAG" <<'PY'
import sqlite3, sys, glob
tag = sys.argv[1]
db = glob.glob(f"gpurun_out/prof_stats/**/{tag}_results.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
lines = [f"# rocprofv3 --kernel-trace --stats — bench.py --steps 3 --warmup 1 (configs[1], 10k docs, 1 MI355X), tag {tag}", "",
         "| kernel | calls | total ms | avg ms | % |", "|---|---:|---:|---:|---:|"]
for name, calls, tot, avg, pct in rows:
    lines.append(f"| {name} | {calls} | {tot / 1e3:.3f} | {avg / 1e3:.4f} | {pct:.2f} |")
open(f"gpurun_out/{tag}_kernel_stats.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:12]))
PY
grep '^{"metric"' gpurun_out/${TAG}_bench_under_rocprof.log > gpurun_out/${TAG}_bench_line.json
rm -rf gpurun_out/prof_stats
