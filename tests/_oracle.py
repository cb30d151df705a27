"""ctypes binding of the CPU oracle (oracle/liblorooracle.so).  Test infrastructure only."""
import ctypes, os, subprocess
import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_ROOT, "oracle", "liblorooracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")])
        L = ctypes.CDLL(so)
        L.lo_batch_run.restype = ctypes.c_void_p
        L.lo_batch_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int]
        L.lo_batch_run_at.restype = ctypes.c_void_p
        L.lo_batch_run_at.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.lo_batch_status.restype = ctypes.c_int32
        L.lo_batch_status.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        L.lo_batch_pending.restype = ctypes.c_uint64
        L.lo_batch_pending.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        for f in (L.lo_batch_json, L.lo_batch_vv):
            f.restype = ctypes.c_void_p
            f.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)]
        L.lo_batch_err.restype = ctypes.c_char_p
        L.lo_batch_err.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        L.lo_batch_free.argtypes = [ctypes.c_void_p]
        L.lo_xxh32.restype = ctypes.c_uint32
        L.lo_xxh32.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32]
        L.lo_visible_ids.restype = ctypes.c_int64
        L.lo_visible_ids.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
        L.lo_visible_ids2.restype = ctypes.c_int64
        L.lo_visible_ids2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int32,
                                      ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
        _LIB = L
    return _LIB


def pack(docs):
    """docs: list of list-of-bytes → (data u8 array, blob_off u64, doc_blob u32)."""
    blobs = [b for d in docs for b in d]
    off = np.zeros(len(blobs) + 1, dtype=np.uint64)
    if blobs:
        off[1:] = np.cumsum([len(b) for b in blobs], dtype=np.uint64)
    data = np.frombuffer(b"".join(blobs) or b"\0", dtype=np.uint8).copy()
    doc_blob = np.zeros(len(docs) + 1, dtype=np.uint32)
    doc_blob[1:] = np.cumsum([len(d) for d in docs], dtype=np.uint32)
    return data, off, doc_blob


def merge_batch(docs, threads=1, packed=None, frontiers=None):
    """Returns list of (status, json bytes, vv bytes, pending) — same tuple the HIP path returns.
    frontiers: optional list (one entry per document) of None or encoded Frontiers bytes = checkout target."""
    L = lib()
    data, off, doc_blob = packed if packed is not None else pack(docs)
    n = len(doc_blob) - 1
    if frontiers is not None:
        assert len(frontiers) == n
        fb = [f or b"" for f in frontiers]
        foff = np.zeros(n + 1, dtype=np.uint64)
        foff[1:] = np.cumsum([len(f) for f in fb], dtype=np.uint64)
        fdata = np.frombuffer(b"".join(fb) or b"\0", dtype=np.uint8).copy()
        h = L.lo_batch_run_at(data.ctypes.data, off.ctypes.data, doc_blob.ctypes.data, n, fdata.ctypes.data, foff.ctypes.data, threads)
    else:
        h = L.lo_batch_run(data.ctypes.data, off.ctypes.data, doc_blob.ctypes.data, n, threads)
    out = []
    try:
        ln = ctypes.c_uint64()
        for i in range(n):
            st = L.lo_batch_status(h, i)
            p = L.lo_batch_json(h, i, ctypes.byref(ln))
            js = ctypes.string_at(p, ln.value) if ln.value else b""
            p = L.lo_batch_vv(h, i, ctypes.byref(ln))
            vv = ctypes.string_at(p, ln.value) if ln.value else b""
            out.append((st, js, vv, L.lo_batch_pending(h, i)))
    finally:
        L.lo_batch_free(h)
    return out


def merge(blobs, frontiers=None):
    return merge_batch([list(blobs)], frontiers=None if frontiers is None else [frontiers])[0]


def xxh32(b, seed=0x4F524F4C):
    return lib().lo_xxh32(bytes(b), len(b), seed)


def visible_ids(blobs, name, kind):
    """visible element ids of a sequence container; `name` = root container name or a loro_amd.wire.CID"""
    L = lib()
    data, off, _ = pack([list(blobs)])
    cap = 1 << 20
    peers = np.zeros(cap, dtype=np.uint64)
    ctrs = np.zeros(cap, dtype=np.int32)
    if isinstance(name, str):
        n = L.lo_visible_ids2(data.ctypes.data, off.ctypes.data, len(blobs), name.encode(), 0, 0, kind, peers.ctypes.data, ctrs.ctypes.data, cap)
    elif name.root:
        n = L.lo_visible_ids2(data.ctypes.data, off.ctypes.data, len(blobs), name.name.encode(), 0, 0, kind, peers.ctypes.data, ctrs.ctypes.data, cap)
    else:
        n = L.lo_visible_ids2(data.ctypes.data, off.ctypes.data, len(blobs), None, name.peer, name.counter, kind, peers.ctypes.data, ctrs.ctypes.data, cap)
    assert n >= 0
    return list(zip(peers[:n].tolist(), ctrs[:n].tolist()))
