"""lm_export through the product library on the GPU box (the run that fixes the exported version is the device's)."""
import pytest

import test_emu_export as E
from loro_amd import workload
import _oracle

pytestmark = pytest.mark.gpu


def _engine():
    import loro_amd
    return loro_amd.MergeEngine(0)


def test_export_roundtrips_and_versions():
    E.check_export_roundtrips(_engine)
    E.check_export_from_versions(_engine)


def test_every_bench_blob_comes_back():
    """configs[1] documents (three blobs each): export from the empty version = one blob with the same blocks in (peer, counter)
    order — byte for byte what the template's writer produces for those blocks — and the same document"""
    tpl = workload.Cfg2Template(50000, 25000, seed=0, commit_every=10, fuse=True)
    docs = [tpl.stamp(d) for d in range(8)]
    with _engine() as c:
        c.stage(docs); c.run()
        got = c.fetch()
        out = [c.export(i) for i in range(len(docs))]
    want = _oracle.merge_batch(docs, threads=8)
    assert got == want
    again = _oracle.merge_batch([[o] for o in out], threads=8)
    assert again == want
    for blobs, o in zip(docs, out):   # the frames of the three blobs, in order (A's blocks, then B's: peers ascend), are the export's
        assert o[22:] == b"".join(b[22:] for b in blobs)


def test_summary_exchange_in_the_c_abi():
    """lm_comm_init / lm_summary_allgather on the device: one rank (no RCCL needed), the six words per document of loro_amd/dist.py"""
    import numpy as np
    import _cases
    from loro_amd import dist as lmdist
    names, docs = _cases.edge_case_docs()
    docs = docs * 12
    with _engine() as c:
        c.stage(docs); c.run()
        st, jl, vl, pe = c.result_meta()
        ids = list(range(len(docs)))
        want = lmdist.summarize_device(ids, st, pe, jl, vl, c.result_hashes())
        c.comm_init(0, 1)
        got = c.summary_allgather(ids, len(docs))
        assert (got == want).all()
        uid = bytes(128)
        buf = __import__("ctypes").create_string_buffer(128)
        assert c.b.comm_unique_id(buf) == 0      # librccl loads on the GPU box
