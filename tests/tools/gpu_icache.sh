#!/bin/bash
# Instruction-cache counters of the integrate kernel on configs[1] (tests/tools/gpu_ab.py, one staged batch): requests / hits / misses of
# the shared instruction cache, instruction fetches issued and their average latency.   bash tests/tools/gpu_icache.sh [docs] [lib.so]
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-10000}
SO=${2:-}
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH" "SQ_IFETCH_LEVEL SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQC_ICACHE_BUSY_CYCLES"; do
  i=$((i+1)); rm -rf /tmp/ic_$i
  if [ -n "$SO" ]; then EXTRA="--so $R/$SO"; export LM_BINDING_LENIENT=1; else EXTRA=""; fi
  timeout 300 rocprofv3 --pmc $SET -d /tmp/ic_$i -o ic -- python $R/tests/tools/gpu_ab.py $N base: $EXTRA > /tmp/ic_$i.log 2>&1
  tail -1 /tmp/ic_$i.log | cut -c1-90
  python3 - $i <<'PY'
import sqlite3, glob, sys
for f in glob.glob('/tmp/ic_%s/**/*.db' % sys.argv[1], recursive=True):
    c = sqlite3.connect(f)
    for k in ('k_integrate_span', 'k_block_decode_wave', 'k_emit_text'):
        for r in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by kernel_name, counter_name", (k + '%',)):
            print('  %s %s avg per launch %.4e (%d launches)' % (r[0].split('(')[0], r[1], r[2], r[3]))
PY
done
