#!/bin/bash
# ad-hoc: instruction-class counters of k_integrate_span after the scalar diet (run on the GPU box through gpurun); $1 = documents
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
N=${1:-5000}
rm -rf /tmp/sq7
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES -d /tmp/sq7 -o sq -- python $R/tests/tools/gpu_ab.py $N x:LM_DEC_SLOT=1024 > $R/gpurun_out/sq7.log 2>&1
python3 - <<PY
import sqlite3, glob
for f in glob.glob('/tmp/sq7/**/*.db', recursive=True):
    c=sqlite3.connect(f)
    for k in ('k_integrate_span','k_block_decode_wave'):
        for r in c.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by counter_name", (k+'%',)): print(k, r[0], 'avg per launch %.4e (%d launches)'%(r[1], r[2]))
PY
