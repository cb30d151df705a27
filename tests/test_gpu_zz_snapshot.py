"""GPU parity of the FastSnapshot ingest on written snapshots (tests/test_emu_snapshot.py builds them): many SSTable blocks,
large-value blocks, LZ4 frames, state-only roots, damaged tables — the HIP path through the C ABI against the oracle."""
import pytest

import _oracle, _resident
import test_emu_snapshot as S

pytestmark = pytest.mark.gpu


def _engine():
    import loro_amd
    return loro_amd.MergeEngine(0)


def test_written_snapshots_match_the_oracle():
    names, docs = S.snapshot_docs(48, first=1000)
    docs = docs + S.damaged_snapshots()
    want = _oracle.merge_batch(docs, threads=16)
    with _engine() as e:
        got = e.merge_batch(docs)
    for i, (g, w) in enumerate(zip(got, want)):
        if w[0] == 0:
            assert g == w, (names[i] if i < len(names) else i, g[:2], w[:2])
        else:
            assert g[0] != 0, i


def test_snapshot_of_a_large_document_and_resident_documents():
    import loro_amd
    S_emu = S._emu
    class _B:   # run the harness-based tests of the CPU file on the device
        @staticmethod
        def merge_batch(docs, frontiers=None):
            with _engine() as e:
                return e.merge_batch(docs, frontiers)
        @staticmethod
        def binding():
            return loro_amd._binding()
    S._emu = _B
    try:
        S.test_a_configs1_sized_document_as_one_snapshot()
        S.test_state_only_roots_and_a_movable_list_root_do_not_flag_the_document()
        S.test_snapshots_and_resident_documents()
    finally:
        S._emu = S_emu


def test_documents_staged_from_their_snapshots_state_sections():
    """SURVEY §8f N3 (fast_snapshot.rs:168-258): one snapshot = its state section, no history uploaded / decoded / replayed; shallow
    snapshots at their latest version and at their shallow root (loro_js_interop.rs:129-147)"""
    S.check_state_path(_engine, n=60)


def test_updates_on_top_of_a_snapshots_state():
    """SURVEY §8f N3, second half (lm_snapshot_base.h): snapshot + updates that continue its history — the snapshot's history is
    neither uploaded nor decoded nor replayed; updates concurrent with part of it fall back to the ChangeStore"""
    S.check_state_base(_engine, n=60)


def test_many_base_deletes_on_top_of_a_snapshots_state():
    """a configs[1]-shaped history with its base as a snapshot: thousands of by-position deletes per document (crdt_rope.rs:256-335),
    and the fall-back to the snapshot's history when the by-position list overflows"""
    S.check_state_base_large(_engine, n_base=20000, n_branch=10000, n=4)
    S.check_state_base_many_documents(_engine, n_base=20000, n_branch=10000, n_docs=300, dir_opt_max=16)
