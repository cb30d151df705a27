#!/bin/bash
# Run ON the GPU box (through gpurun) from the repo root:  bash profiles/collect_sq_mf.sh r06 [docs]
# SQ counters of k_map_fused on configs[2] documents (tests/tools/gpu_cfg.py cfg3): two rocprofv3 --pmc passes, no trace combined.
TAG=${1:-r06}
N=${2:-2048}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_map_fused_sq_counters.log
: > $OUT
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_FLAT" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/sqm_$i
  timeout 600 rocprofv3 --pmc $SET -d /tmp/sqm_$i -o sq -- python $R/tests/tools/gpu_cfg.py cfg3 $N 8 > $R/gpurun_out/${TAG}_sqm_pass$i.log 2>&1
  python3 - $i $N >> $OUT <<'PY'
import sqlite3, glob, sys
i, n = sys.argv[1], int(sys.argv[2])
for f in glob.glob('/tmp/sqm_%s/**/*.db' % i, recursive=True):
    c = sqlite3.connect(f)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    if 'counters_collection' not in tabs:
        print('no counters_collection in', f, tabs[:8]); continue
    for r in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like 'k_map_fused%' group by kernel_name, counter_name"):
        print('%s %s avg per launch %.4e (%d launches) per document %.4e' % (r[0].split('(')[0], r[1], r[2], r[3], r[2] / (n / 2)))
PY
done
cat $OUT
