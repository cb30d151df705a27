"""Direct staging (include/loro_merge.h: lm_host_alloc / lm_staged_direct): blobs that already live in pinned memory of the library
are copied to the device from where they are — same results as the gather, the replay paths (side engine, state-staged snapshots'
history) read the caller's region, anything that does not qualify falls back.  Kernel-logic harness; the GPU suite runs the same."""
import ctypes
import pytest

import _cases, _emu, _fuzz, _oracle
from loro_amd._cabi import Context


def check_direct(ctx_factory):
    docs = [_fuzz.blobs_of(_fuzz.random_session(s, n_peers=3, n_steps=80, kinds=("text", "list", "map"))) for s in range(24)]
    bad, _good = _cases.misnamed_delete_docs(8)          # (replayed by the side engine: read again from the staged bytes)
    docs += bad
    want = _oracle.merge_batch(docs, threads=8)
    with ctx_factory() as c:
        plain = c.merge_batch(docs)
        assert c.b.staged_direct(c.h) == 0 and plain == want
        packed = c.pack_pinned(docs)
        try:
            c.stage_packed(packed); c.run()
            assert c.b.staged_direct(c.h) == 1 and c.fetch() == want
            # a second batch through the same region, then the gather again (the engine's own staging buffer is still there)
            c.stage_packed(packed); c.run()
            assert c.b.staged_direct(c.h) == 1 and c.fetch() == want
            assert c.merge_batch(docs) == want and c.b.staged_direct(c.h) == 0
            # out of order (the documents reversed, their blobs where they were): not the layout of the contract -> gathered
            arr, (n, keep) = packed
            rev = (type(arr[0]) * n)()
            for i in range(n):
                rev[i] = arr[n - 1 - i]
            c.stage_packed((rev, (n, keep))); c.run()
            assert c.b.staged_direct(c.h) == 0 and c.fetch() == want[::-1]
        finally:
            c.free_pinned(packed)
        # resident documents on top of a directly staged batch (lm_import reads nothing of the first batch from the host)
        a, b = docs[:6], [[x] for x in _fuzz.blobs_of(_fuzz.random_session(99, n_peers=2, n_steps=40, kinds=("text",)))[:1]] * 6
        packed = c.pack_pinned(a)
        try:
            c.stage_packed(packed); c.run()
            assert c.b.staged_direct(c.h) == 1
            c.import_more(b); c.run()
            got = c.fetch()
        finally:
            c.free_pinned(packed)
        assert got == _oracle.merge_batch([x + y for x, y in zip(a, b)], threads=4)


def test_direct_staging_on_the_kernel_logic_harness():
    check_direct(lambda: Context(_emu.binding()))
