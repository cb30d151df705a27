"""Every document shape bench.py times (configs[0], [2], [3], [4] at 60k ops x 16 checkouts, the heterogeneous configs[1] mix, the
MovableList leg) through the kernel-logic harness against the oracle, with the integrate instantiation that took the batch — run
before a default of the integrate stage changes without a GPU at hand (≈12 min of CPU)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench, _emu, _oracle, _cases
from loro_amd import workload
from loro_amd._cabi import Context
t0=time.time()
def chk(name, docs, fr=None):
    want=_oracle.merge_batch(docs, threads=8, frontiers=fr)
    with Context(_emu.binding()) as c:
        c.set_profiling(1)
        got=c.merge_batch(docs, fr)
        st=[k for k,_ in c.kernel_times() if 'integrate' in k]
    bad=[i for i,(g,w) in enumerate(zip(got,want)) if g!=w]
    print("%6.0fs %-28s %4d docs  stage %s  mismatches %s  statuses %s" % (time.time()-t0, name, len(docs), st, bad[:8], sorted(set(w[0] for w in want))), flush=True)
chk("configs[0]", [workload.cfg1_doc(d) for d in range(100)])
chk("configs[3]", _cases.cfg4_docs(96))
chk("movable", [bench._gen(("movable", d))[0] for d in range(16)])
shapes = [(2000, 1000, 10, True), (10000, 5000, 10, True), (25000, 12500, 10, True), (50000, 25000, 10, True),
          (100000, 50000, 10, True), (5000, 2500, 1, False), (20000, 10000, 1, False)]
tpls=[bench._gen(("tpl", sh))[0] for sh in shapes]
chk("heterogeneous", [tpls[(d * 7919) % len(tpls)].stamp(d) for d in range(14)])
blobs, fr = workload.cfg5_doc(0, n_ops=60000, turn=1000, n_checkouts=16)
chk("configs[4] 60k ops x 16", [blobs]*len(fr), fr)
chk("configs[2] a", [workload.cfg3_doc(0, combined=True)])
chk("configs[2] b", [workload.cfg3_doc(1, combined=False)])
