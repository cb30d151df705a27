#!/bin/bash
# Ad-hoc: does rocprofv3 change the shader clock?  Samples `rocm-smi --showclocks` while gpu_ab.py runs, with and without the profiler.
cd $GRAFT_REPO_ROOT; out=gpurun_out/r03_clock_probe.log; : > $out
sample() { for i in $(seq 1 ${1:-12}); do rocm-smi --showclocks 2>/dev/null | grep -iE "sclk|fclk|mclk" | tr '\n' ' ' >> $out; echo >> $out; sleep 1; done; }
echo "== idle" >> $out; sample 2
echo "== gpu_ab.py 10000 without a profiler" >> $out
( python tests/tools/gpu_ab.py 10000 cur: 2>&1 | tail -1 > gpurun_out/_ab_plain.txt ) & sleep 14; sample 8; wait; cat gpurun_out/_ab_plain.txt >> $out
echo "== the same under rocprofv3 --kernel-trace --stats" >> $out
cd /tmp && export TMPDIR=/tmp
( rocprofv3 --kernel-trace --stats -d /tmp/clk_prof -o clk -- python $GRAFT_REPO_ROOT/tests/tools/gpu_ab.py 10000 cur: 2>&1 | grep "^\[cur\]" > $GRAFT_REPO_ROOT/gpurun_out/_ab_prof.txt ) & sleep 16; cd $GRAFT_REPO_ROOT; sample 8; wait; cat gpurun_out/_ab_prof.txt >> $out
cat $out
