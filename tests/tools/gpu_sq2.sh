#!/bin/bash
# ad-hoc: wave-state counters of k_integrate (run on the GPU box through gpurun)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_sq2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU -d $R/gpurun_out/prof_sq2 -o sq -- python $R/tests/tools/gpu_probe.py ${1:-5000} 50000 > $R/gpurun_out/sq2.log 2>&1
python3 - <<PY
import sqlite3
c=sqlite3.connect('$R/gpurun_out/prof_sq2/sq_results.db')
n=${1:-5000}
for r in c.execute("select counter_name, avg(value) from counters_collection where kernel_name='k_integrate' group by counter_name"): print(r[0], 'per doc %.3e'%(r[1]/n))
PY
grep -E "k_integrate" $R/gpurun_out/sq2.log
