#!/bin/bash
# ad-hoc: instruction-mix counters of k_integrate (run on the GPU box through gpurun)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_sq
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY -d $R/gpurun_out/prof_sq -o sq -- python $R/tests/tools/gpu_probe.py ${1:-2048} 50000 > $R/gpurun_out/sq.log 2>&1
python3 - <<PY
import sqlite3
c=sqlite3.connect('$R/gpurun_out/prof_sq/sq_results.db')
n=${1:-2048}
for r in c.execute("select counter_name, avg(value) from counters_collection where kernel_name='k_integrate' group by counter_name"): print(r[0], 'per doc %.3e'%(r[1]/n))
PY
grep -E "run\(noprof\) 2|k_integrate|parity" $R/gpurun_out/sq.log
