"""GPU parity, round 6 (through the C ABI, against the oracle):
  * LWW Map documents resolved WITHOUT op rows — k_block_kind / k_doc_kind / k_block_head / k_map_fused (lm_k_map_fused.h; diff_calc.rs:488-616
    over the columns of block_encode.rs:417-428) — configs[2] shapes, random Map sessions, checkouts, damaged documents;
  * the side engine (lm_capi_impl.h redo): a document's verdict depends neither on the kernel its batch's statistics picked nor on
    whether its entries share their blobs;
  * checked-out entries import their whole history by default (loro.rs:1625-1746): the damaged-checkout corpus."""
import json, os
import pytest

import _cases, _oracle
from loro_amd import wire, workload

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import loro_amd
    e = loro_amd.MergeEngine(0)
    yield e
    e.close()


def _force_fused(monkeypatch):
    monkeypatch.setenv("LM_MF_MIN_ROWS", "1"); monkeypatch.setenv("LM_MF_CHG_RATIO", "0")


def test_map_documents_are_resolved_without_op_rows(engine, monkeypatch):
    _force_fused(monkeypatch)
    docs = [workload.cfg3_doc(d, n_peers=6, n_writes=3000, n_keys=300, combined=d % 2 == 0, per_change=100) for d in range(6)]
    docs += _cases.map_sessions(400, scalar_only=False)
    docs = docs * 3   # (enough workgroups for every CU to hold one)
    want = _oracle.merge_batch(docs, threads=8)
    got = engine.merge_batch(docs)
    n_fused, n_redo = engine.b.fused_documents(engine.h), engine.b.redo_documents(engine.h)
    assert got == want
    assert n_fused == len(docs) and 0 < n_redo < len(docs) // 2, (n_fused, n_redo)
    monkeypatch.setenv("LM_MAP_FUSED", "0")
    assert engine.merge_batch(docs) == want and engine.b.fused_documents(engine.h) == 0


def test_the_product_default_takes_large_map_documents_only(engine):
    """default knobs: a document needs 2,048 rows and four rows per change on average for a workgroup of its own"""
    big = workload.cfg3_doc(3, n_peers=8, n_writes=1000, n_keys=200, combined=True, per_change=100)
    small = workload.cfg3_doc(4, n_peers=2, n_writes=100, n_keys=20, combined=True, per_change=10)
    keystrokes = workload.cfg3_doc(5, n_peers=4, n_writes=1000, n_keys=50, combined=True, per_change=2)
    docs = [big, small, keystrokes] * 40
    got = engine.merge_batch(docs)
    assert engine.b.fused_documents(engine.h) == 40
    assert got == _oracle.merge_batch(docs, threads=8)


def test_map_documents_without_op_rows_at_checked_out_versions(engine, monkeypatch):
    _force_fused(monkeypatch)
    monkeypatch.setenv("LM_SHARE_REPLAY", "0")
    docs, fr = [], []
    for d in range(24):
        blobs = workload.cfg3_doc(d, n_peers=3, n_writes=400, n_keys=50, combined=(d % 2 == 0), per_change=20)
        for ctr in (0, 57, 399):
            docs.append(blobs); fr.append(wire.encode_frontiers([(d * 1000 + 1, ctr)] + ([(d * 1000 + 2, 100)] if ctr == 57 else [])))
        docs.append(blobs); fr.append(None)
    got = engine.merge_batch(docs, fr)
    assert engine.b.fused_documents(engine.h) == len(docs)
    assert got == _oracle.merge_batch(docs, threads=8, frontiers=fr)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_damaged_map_documents_get_the_row_decoders_verdicts(engine, monkeypatch, seed):
    docs = _cases.damaged_map_docs(600, seed=seed)
    want = _oracle.merge_batch(docs, threads=8)
    _force_fused(monkeypatch)
    fused = engine.merge_batch(docs)
    assert engine.b.fused_documents(engine.h) > 200 and engine.b.redo_documents(engine.h) > 40
    monkeypatch.setenv("LM_MAP_FUSED", "0")
    rows = engine.merge_batch(docs)
    assert fused == rows
    assert not [i for i in range(len(docs)) if want[i][0] != 0 and fused[i][0] == 0]
    assert not [i for i in range(len(docs)) if want[i][0] == 0 and fused[i][0] == 0 and fused[i] != want[i]]


def test_a_documents_verdict_depends_neither_on_its_neighbours_nor_on_shared_blobs(engine, monkeypatch):
    bad, _good = _cases.misnamed_delete_docs(64)
    empty = b"\x00"
    want_latest = _oracle.merge_batch(bad, threads=8)
    want_empty = _oracle.merge_batch(bad, threads=8, frontiers=[empty] * len(bad))
    docs, fr, want = [], [], []
    for i, d in enumerate(bad):
        docs += [d, d]; fr += [None, empty]; want += [want_latest[i], want_empty[i]]
    for env in ({"LM_SPAN_AUTO": "1", "LM_SHARE_REPLAY": "0"}, {"LM_SPAN_AUTO": "0"}, {"LM_SPAN_AUTO": "1"}, {"LM_SPAN_AUTO": "1", "LM_CHECKOUT_FULL": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = engine.merge_batch(docs, fr)
        assert got == want, env
        for k in env:
            monkeypatch.delenv(k)


def test_checked_out_entries_import_their_whole_history(engine, monkeypatch):
    """the default since round 6 (LM_CHECKOUT_FULL): damage OUTSIDE the rendered version fails the entry like LoroDoc::import in front
    of LoroDoc::checkout does — no document the oracle rejects is rendered, what both accept is rendered alike; the closure replay of
    rounds 1-5 (LM_CHECKOUT_FULL=0 / unshared) still shows the deviation"""
    import test_emu_parity
    docs, fronts = test_emu_parity._damaged_checkout_docs(n=1000, seed=1)
    want = _oracle.merge_batch(docs, threads=8, frontiers=fronts)

    def dev_only(got):
        return sum(1 for g, w in zip(got, want) if g[0] == 0 and w[0] not in (0, 4))
    got = engine.merge_batch(docs, fronts)
    assert dev_only(got) == 0
    n_both = 0
    for i, (g, w) in enumerate(zip(got, want)):
        if g[0] == 0 and w[0] == 0:
            assert g == w, i
            n_both += 1
    assert n_both > 300
    monkeypatch.setenv("LM_CHECKOUT_FULL", "0")
    assert dev_only(engine.merge_batch(docs, fronts)) > 0
    monkeypatch.delenv("LM_CHECKOUT_FULL")
    cd, cf = test_emu_parity._checkout_cases()
    cw = _oracle.merge_batch(cd, frontiers=cf)
    for i, (g, w) in enumerate(zip(engine.merge_batch(cd, cf), cw)):
        assert (g == w) if w[0] == 0 else (g[0] == w[0]), (i, g[:3], w[:3])


def test_richtext_values_of_checked_out_entries_of_a_folded_batch(engine):
    import _richtext
    docs, fronts = _richtext.checkout_cases(n=4)
    res = engine.merge_batch(docs, fronts)
    assert engine.b.shared_documents(engine.h) > 0
    got = engine.richtext()
    assert engine.fetch() == res == _oracle.merge_batch(docs, threads=8, frontiers=fronts)
    _richtext.same(got, _oracle.richtext_batch(docs, fronts), "folded")


def test_direct_staging_from_pinned_host_memory():
    """include/loro_merge.h "Direct staging": blobs inside an lm_host_alloc region reach the device without the host-side gather — same
    results, the replay paths read the caller's region, whatever does not fit the contract is gathered"""
    import loro_amd
    import test_emu_stage_direct
    test_emu_stage_direct.check_direct(lambda: loro_amd.MergeEngine(0))
