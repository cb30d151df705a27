#!/bin/bash
# Ad-hoc: PC sampling of the configs[1] pipeline (rocprofv3 beta feature); histogram only comes back.
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for m in stochastic host_trap; do
  rm -rf /tmp/pcs_$m
  if [ $m = stochastic ]; then U="--pc-sampling-unit cycles --pc-sampling-interval 1048576"; else U="--pc-sampling-unit time --pc-sampling-interval 500"; fi
  ( cd $R && timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $m $U --kernel-trace --output-format csv -d /tmp/pcs_$m -o pcs -- python tests/tools/gpu_ab.py ${1:-5000} dbg: --so loro_amd/csrc/libloromerge_dbg.so ) > $R/gpurun_out/pcs_$m.log 2>&1
  echo "rc=$?" >> $R/gpurun_out/pcs_$m.log
  python $R/tests/tools/pcs_agg.py /tmp/pcs_$m $R/gpurun_out/pcs_$m.txt
  ls -laR /tmp/pcs_$m | head -30 >> $R/gpurun_out/pcs_$m.log
  if grep -q "^samples: [1-9]" $R/gpurun_out/pcs_$m.txt; then break; fi
done
tail -5 $R/gpurun_out/pcs_*.log
