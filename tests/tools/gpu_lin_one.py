"""GPU box: one variant of tests/tools/gpu_lin.py for counter passes:  python tests/tools/gpu_lin_one.py <docs> <blobs e.g. 0 or 012> (LM_LINEAR from the environment)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import loro_amd
from loro_amd import workload
n = int(sys.argv[1]); sel = [int(c) for c in sys.argv[2]]
tpl = workload.Cfg2Template(50000, 25000, seed=0, commit_every=10, fuse=True)
docs = [[b[i] for i in sel] for b in (tpl.stamp(d) for d in range(n))]
with loro_amd.MergeEngine(0) as e:
    e.stage(docs); e.run(); e.set_profiling(1); e.run()
    print({k: round(v, 3) for k, v in e.kernel_times() if "integrate" in k})
