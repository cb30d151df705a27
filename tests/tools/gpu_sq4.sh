#!/bin/bash
# ad-hoc: instruction-cache and wait counters of k_integrate_span (run on the GPU box through gpurun); $1 = documents
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
N=${1:-5000}
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_WAIT_IFETCH" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1)); rm -rf /tmp/sq4_$i
  timeout 200 rocprofv3 --pmc $set -d /tmp/sq4_$i -o sq -- python $R/tests/tools/gpu_ab.py $N x:LM_DEC_SLOT=1024 > $R/gpurun_out/sq4_$i.log 2>&1
  python3 - <<PY
import sqlite3, glob
for f in glob.glob('/tmp/sq4_$i/**/*.db', recursive=True):
    c=sqlite3.connect(f)
    tabs=[r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    try:
        for r in c.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like 'k_integrate_span%' group by counter_name"): print(r[0], 'avg per launch %.4e (%d launches)'%(r[1], r[2]))
    except Exception as e: print('ERR', e, tabs[:20])
PY
done
