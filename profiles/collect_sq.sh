#!/bin/bash
# Run ON the GPU box (through gpurun) from the repo root:  bash profiles/collect_sq.sh r05 [docs]
# SQ counters of the integrate kernel that runs on configs[1] (tests/tools/gpu_ab.py: one staged batch, streams serialized for
# two passes + four timed runs), three rocprofv3 --pmc passes (the counters do not fit one pass); kernel trace / stats are NOT
# combined with --pmc.  Output: profiles/<tag>_integrate_sq_counters.log (per launch and per document).
TAG=${1:-r05}
N=${2:-10000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_integrate_sq_counters.log
: > $OUT
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_FLAT" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rm -rf /tmp/sq_$i
  timeout 300 rocprofv3 --pmc $SET -d /tmp/sq_$i -o sq -- python $R/tests/tools/gpu_ab.py $N base: > $R/gpurun_out/${TAG}_sq_pass$i.log 2>&1
  tail -1 $R/gpurun_out/${TAG}_sq_pass$i.log >> $OUT
  python3 - $i $N >> $OUT <<'PY'
import sqlite3, glob, sys
i, n = sys.argv[1], int(sys.argv[2])
for f in glob.glob('/tmp/sq_%s/**/*.db' % i, recursive=True):
    c = sqlite3.connect(f)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    t = 'counters_collection' if 'counters_collection' in tabs else None
    if not t:
        print('no counters_collection in', f, tabs[:8]); continue
    for k in ('k_integrate_span', 'k_integrate_linear', 'k_block_decode_wave', 'k_emit_text', 'k_elem_fill'):
        for r in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by kernel_name, counter_name", (k + '%',)):
            print('%s %s avg per launch %.4e (%d launches)' % (r[0].split('(')[0], r[1], r[2], r[3]))
PY
done
mkdir -p $R/gpurun_out/profiles && cp $OUT $R/gpurun_out/profiles/
cat $OUT
