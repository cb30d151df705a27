#!/bin/bash
# SQ counters of the integrate kernel on base-only configs[1] documents, with and without the linear prefix
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for LIN in 1 0; do
 i=0
 for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_FLAT"; do
  i=$((i+1)); rm -rf /tmp/sq_$i
  LM_LINEAR=$LIN timeout 300 rocprofv3 --pmc $SET -d /tmp/sq_$i -o sq -- python $R/tests/tools/gpu_lin_one.py 5000 ${1:-0} > /tmp/sq_$i.log 2>&1
  echo "LM_LINEAR=$LIN $(tail -1 /tmp/sq_$i.log)"
  python3 - $i <<'PY'
import sqlite3, glob, sys
for f in glob.glob('/tmp/sq_%s/**/*.db' % sys.argv[1], recursive=True):
    c = sqlite3.connect(f)
    for r in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like 'k_integrate_span%' group by kernel_name, counter_name"):
        print('  %s %s %.4e (%d launches)' % (r[0].split('(')[0], r[1], r[2], r[3]))
PY
 done
done
