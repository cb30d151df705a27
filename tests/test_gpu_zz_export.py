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
