#!/bin/bash
# Ad-hoc: the resident entries of bench.py under several builds of the library (LORO_AMD_LIB).  usage: gpu_resident_ab.sh cur res34 ...
cd $GRAFT_REPO_ROOT; out=gpurun_out/r03_ab_res.log; : > $out
for v in "$@"; do
  if [ $v = cur ]; then unset LORO_AMD_LIB; else export LORO_AMD_LIB=$PWD/tests/tools/ab/lib_$v.so; fi
  echo "== $v" >> $out
  python tests/tools/gpu_resident.py 10000 --quick 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read(); d = json.loads(t[t.index('{'):])
for k, v in d.items():
    print(k, {kk: vv for kk, vv in v.items() if kk in ('docs_per_s', 'ms_per_batch', 'lm_import_ms', 'renderings_per_s', 'ms_total', 'from_empty_run_of_base_plus_A_ms', 'error')})
    s = v.get('stage_ms_of_the_import_run_streams_serialized')
    if s: print('   import run:', {a: b for a, b in s.items() if 'integrate' in a})
" >> $out 2>&1
done
cat $out
