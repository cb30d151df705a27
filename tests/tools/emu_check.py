"""Ad-hoc: run fuzz / trace / checkout cases through the kernel-logic harness built with -DLM_EMU_CHECK (the span kernel's
structural checker: directory vs leaves vs loc[] vs cached prefix after every op) and compare with the oracle.
    g++ -O1 -g -std=c++17 -fPIC -shared -Wno-unknown-pragmas -DLM_EMU_TRACE -DLM_EMU_CHECK -o tests/emu/libloroemu_check.so tests/emu/lm_emu.cpp"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from loro_amd._cabi import Binding, Context
import _oracle, _cases, test_emu_parity

b = Binding(os.path.join(ROOT, "tests", "emu", "libloroemu_check.so"), "lmemu_")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
docs = _cases.fuzz_docs(n) + _cases.fuzz_docs(max(4, n // 4), base=1000, steps=120, peers=4, max_ins=30, sync_prob=0.08) + _cases.trace_docs(3000, n_docs=1)
with Context(b) as c:
    got = c.merge_batch(docs)
want = _oracle.merge_batch(docs, threads=8)
bad = [i for i, (g, w) in enumerate(zip(got, want)) if g != w]
print("plain:", len(docs), "docs, mismatches", bad[:10])
cd, cf = test_emu_parity._checkout_cases()
with Context(b) as c:
    got = c.merge_batch(cd, cf)
want = _oracle.merge_batch(cd, threads=8, frontiers=cf)
bad2 = [i for i, (g, w) in enumerate(zip(got, want)) if ((g != w) if w[0] == 0 else (g[0] != w[0]))]
print("checkout:", len(cd), "docs, mismatches", bad2[:10])
sys.exit(1 if bad or bad2 else 0)
