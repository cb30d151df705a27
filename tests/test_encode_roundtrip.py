"""Export / encode side: the product's host-side block encoder (loro_amd/csrc/lm_encode.h through the
C ABI lm_encode_block / lm_encode_updates) against real blobs.  The oracle's decoder supplies each block's tables (it is the
checker's parser; the bytes that come back are the product's).
  * Rust-written `updates.blob` (loro-js/tests/fixtures/rust, rust-interop.test.ts:47-53): every block re-encodes to ITS OWN
    BYTES and the reframed blob is identical — the column segmentation is serde_columnar's, the checksum the envelope's;
  * blobs of the in-repo writer and of random sessions: byte-identical as well (same strategies);
  * TS-written fixtures: the TS writer emits columns as one literal segment, so the bytes differ by design; re-encoding
    must still decode to the same document (JSON + VV through the oracle)."""
import ctypes, json, os
import numpy as np
import pytest
import _oracle, _cases, _fuzz
import loro_amd
from loro_amd import wire, workload

HERE = os.path.dirname(os.path.abspath(__file__))
FX = json.load(open(os.path.join(HERE, "golden", "reference_fixtures.json")))
BLOB = {k: bytes.fromhex(v) for k, v in FX["blobs"].items()}


@pytest.fixture(scope="module")
def lm():
    if not os.path.exists(loro_amd.LIB_PATH):
        loro_amd.build_library()
    L = ctypes.CDLL(loro_amd.LIB_PATH)
    L.lm_encode_block.restype = ctypes.c_int
    L.lm_encode_block.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    L.lm_encode_updates.restype = ctypes.c_int
    L.lm_encode_updates.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    L.lm_free_bytes.argtypes = [ctypes.c_void_p]
    return L


def reencode(lm, blob):
    """(blocks re-encoded by the product from the oracle-decoded tables, the blob's own blocks, the reframed blob)"""
    O = _oracle.lib()
    O.lo_blocks_open.restype = ctypes.c_void_p
    O.lo_blocks_open.argtypes = [ctypes.c_char_p, ctypes.c_uint64]
    O.lo_blocks_count.argtypes = [ctypes.c_void_p]
    O.lo_blocks_tables.restype = ctypes.c_void_p
    O.lo_blocks_tables.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    O.lo_blocks_span.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    O.lo_blocks_close.argtypes = [ctypes.c_void_p]
    h = O.lo_blocks_open(blob, len(blob))
    assert h, "the checker's parser rejected the blob"
    mine, theirs = [], []
    try:
        for i in range(O.lo_blocks_count(h)):
            out, n = ctypes.c_void_p(), ctypes.c_size_t()
            assert lm.lm_encode_block(O.lo_blocks_tables(h, i), ctypes.byref(out), ctypes.byref(n)) == 0
            mine.append(ctypes.string_at(out.value, n.value))
            lm.lm_free_bytes(out)
            off, ln = ctypes.c_uint64(), ctypes.c_uint64()
            O.lo_blocks_span(h, i, ctypes.byref(off), ctypes.byref(ln))
            theirs.append(blob[off.value:off.value + ln.value])
    finally:
        O.lo_blocks_close(h)
    ptrs = (ctypes.c_char_p * max(1, len(mine)))(*mine)
    lens = (ctypes.c_size_t * max(1, len(mine)))(*[len(b) for b in mine])
    out, n = ctypes.c_void_p(), ctypes.c_size_t()
    assert lm.lm_encode_updates(ptrs, lens, len(mine), ctypes.byref(out), ctypes.byref(n)) == 0
    framed = ctypes.string_at(out.value, n.value)
    lm.lm_free_bytes(out)
    return mine, theirs, framed


def test_rust_written_fixture_round_trips_byte_for_byte(lm):
    mine, theirs, framed = reencode(lm, BLOB["updates.blob"])
    assert len(mine) > 1
    for i, (a, b) in enumerate(zip(mine, theirs)):
        assert a == b, f"block {i}: {a.hex()} != {b.hex()}"
    assert framed == BLOB["updates.blob"]


def test_writer_and_random_session_blobs_round_trip_byte_for_byte(lm):
    blobs = []
    for docs in (_cases.fuzz_docs(24), _cases.cfg4_docs(6), [workload.Cfg2Template(3000, 1500, seed=3, commit_every=10, fuse=True).stamp(0)],
                 [workload.cfg3_doc(0, n_peers=4, n_writes=300, n_keys=64, combined=True)], [workload.cfg5_doc(0, n_ops=3000, turn=400, n_checkouts=1)[0]]):
        for d in docs:
            blobs += list(d)
    n_blocks = 0
    for b in blobs:
        mine, theirs, framed = reencode(lm, b)
        assert mine == theirs and framed == b
        n_blocks += len(mine)
    assert n_blocks > 100


@pytest.mark.parametrize("name", ["updates.ts.blob", "runtime-updates.ts.blob", "concurrent-base.ts.blob", "concurrent-left.ts.blob",
                                  "concurrent-right.ts.blob", "fugue-left.ts.blob", "fugue-right.ts.blob"])
def test_ts_written_fixtures_reencode_to_the_same_document(lm, name):
    mine, theirs, framed = reencode(lm, BLOB[name])
    assert len(mine) == len(theirs)
    assert _oracle.merge([framed]) == _oracle.merge([BLOB[name]])
    again, _, framed2 = reencode(lm, framed)      # and the product's own bytes are a fixed point
    assert again == mine and framed2 == framed


def test_empty_updates_blob(lm):
    out, n = ctypes.c_void_p(), ctypes.c_size_t()
    assert lm.lm_encode_updates(None, None, 0, ctypes.byref(out), ctypes.byref(n)) == 0
    assert ctypes.string_at(out.value, n.value) == wire.encode_updates([])
    lm.lm_free_bytes(out)
