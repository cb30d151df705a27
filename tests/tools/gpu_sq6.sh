#!/bin/bash
# ad-hoc: counters of the decode / elem_fill / emit kernels on configs[1] ($1 docs)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
N=${1:-5000}
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"; do
  i=$((i+1)); rm -rf /tmp/sq6_$i
  timeout 200 rocprofv3 --pmc $set -d /tmp/sq6_$i -o sq -- python $R/tests/tools/gpu_ab.py $N x:LM_DEC_SLOT=1024 > $R/gpurun_out/sq6_$i.log 2>&1
  python3 - <<PY
import sqlite3, glob
for f in glob.glob('/tmp/sq6_$i/**/*.db', recursive=True):
    c=sqlite3.connect(f)
    for k in ('k_block_decode_wave','k_elem_fill','k_emit_text','k_frame_count'):
        for r in c.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by counter_name", (k+'%',)): print(k, r[0], '%.4e'%r[1])
PY
done
