"""GPU box: the integrate stage of configs[1] documents cut down to base only / base + A / all three blobs, with and without the
linear prefix (LM_LINEAR, lm_k_integrate_linear.h): per-launch kernel time alone (streams serialized).
    python tests/tools/gpu_lin.py [docs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import loro_amd
from loro_amd import workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
tpl = workload.Cfg2Template(50000, 25000, seed=0, commit_every=10, fuse=True)
docs = [tpl.stamp(d) for d in range(n)]
for what, sel in (("base only", (0,)), ("base + A", (0, 1)), ("base + B", (0, 2)), ("base + A + B", (0, 1, 2))):
    dd = [[b[i] for i in sel] for b in docs]
    out = []
    for lin in ("1", "0"):
        os.environ["LM_LINEAR"] = lin
        with loro_amd.MergeEngine(0) as e:
            e.stage(dd); e.run(); e.set_profiling(1)
            acc = {}
            for _ in range(2):
                e.run()
                for name, ms in e.kernel_times():
                    if "integrate" in name:
                        acc.setdefault(name, []).append(ms)
            out.append("LM_LINEAR=%s %s" % (lin, {k: round(sum(v) / len(v), 3) for k, v in acc.items()}))
    print(what, "|", " | ".join(out), flush=True)
