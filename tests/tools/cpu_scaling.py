"""Ad-hoc: thread scaling of the CPU oracle on the configs[1] documents (run on the measurement box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _oracle
from loro_amd import workload
tpl = workload.Cfg2Template(50000, 25000, seed=0, commit_every=10, fuse=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
docs = [tpl.stamp(d) for d in range(n)]
packed = _oracle.pack(docs)
for th in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,8,32,64,128,256").split(",")]:
    k = min(n, max(64, th * 8)) if th < 64 else n
    p = _oracle.pack(docs[:k])
    _oracle.merge_batch(None, threads=th, packed=p)
    t = time.time(); _oracle.merge_batch(None, threads=th, packed=p); dt = time.time() - t
    print("%3d threads: %8.1f docs/s (%d docs)" % (th, k / dt, k), flush=True)
