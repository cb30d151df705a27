"""GPU A/B probe across SEVERAL builds of the library in one process (gpu_ab.py takes one --so): configs[1] batch staged
once per library, environment-selected kernel variants switched between runs (the engine reads them at lm_run time).
    python tests/tools/gpu_ab2.py 10000 [out.log] -- <lib or '-'>:name:ENV=V,ENV=V ...
'-' = the product library.  Every variant's whole output batch is compared with the first variant's, whose first documents
are compared with the oracle.  Lines are appended to out.log as they are produced (a call cut off keeps what ran)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from loro_amd._cabi import Binding, Context
from loro_amd import workload
import loro_amd
import _oracle

t0 = time.time()
sep = sys.argv.index("--")
n_docs = int(sys.argv[1])
log = open(sys.argv[2], "a") if sep > 2 else None


def say(s):
    s = "%6.1fs %s" % (time.time() - t0, s)
    print(s, flush=True)
    if log:
        log.write(s + "\n"); log.flush(); os.fsync(log.fileno())


groups = []                                   # [(lib, [(name, {env})])] in order of first appearance
for a in sys.argv[sep + 1:]:
    lib, name, kv = a.split(":", 2)
    lib = loro_amd.LIB_PATH if lib == "-" else lib
    env = dict(item.split("=", 1) for item in filter(None, kv.split(",")))
    if not groups or groups[-1][0] != lib:
        groups.append((lib, []))
    groups[-1][1].append((name, env))

tpl = workload.Cfg2Template(50000, 25000, seed=0, commit_every=10, fuse=True)
docs = [tpl.stamp(d) for d in range(n_docs)]
want = _oracle.merge_batch(docs[:8], threads=8)
say("batch ready")
ref = None
touched = set()
for lib, variants in groups:
    b = Binding(lib, "lm_")
    with Context(b, 0) as e:
        e.stage(docs)
        for name, env in variants:
            for k in touched:
                os.environ.pop(k, None)
            os.environ.update(env); touched |= set(env)
            e.run()
            e.set_profiling(1)
            agg = {}
            for _ in range(2):
                e.run()
                for kn, ms in e.kernel_times():
                    agg.setdefault(kn, []).append(ms)
            e.set_profiling(0)
            best = 1e9
            for _ in range(3):
                t = time.time(); e.run(); best = min(best, time.time() - t)
            out = e.fetch()
            if ref is None:
                ref = out
                ok = "oracle:%s" % (out[:8] == want)
            else:
                ok = "same-as-first:%s oracle:%s" % (out == ref, out[:8] == want)
            stages = "  ".join("%s=%.2f" % (k.replace("k_", ""), sum(v) / len(v)) for k, v in agg.items())
            say("[%s %s] step %.1f ms (%.0f docs/s) parity %s | per launch alone: %s"
                % (os.path.basename(lib), name, best * 1e3, n_docs / best, ok, stages))
