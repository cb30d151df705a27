#!/bin/bash
# ad-hoc: instruction counters of k_integrate_span per document for two document shapes ($1 docs): a linear 100k-op history vs configs[1]
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for shape in "100000 20" "50000 25000"; do
  set -- $shape
  rm -rf /tmp/sq5
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d /tmp/sq5 -o sq -- python $R/tests/tools/gpu_shape.py 2048 $1 $2 > $R/gpurun_out/sq5_$1.log 2>&1
  tail -2 $R/gpurun_out/sq5_$1.log
  python3 - <<PY
import sqlite3, glob
for f in glob.glob('/tmp/sq5/**/*.db', recursive=True):
    c=sqlite3.connect(f)
    for r in c.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like 'k_integrate_span%' group by counter_name"): print('  shape $1+2x$2', r[0], 'per doc %.4e'%(r[1]/1024))
PY
done
