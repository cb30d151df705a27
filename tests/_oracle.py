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
        L.lo_option_richtext.argtypes = [ctypes.c_int]
        L.lo_batch_richtext.restype = ctypes.c_void_p
        L.lo_batch_richtext.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)]
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


def richtext_batch(docs, frontiers=None, threads=8, with_merge=False):
    """Per document (status, richtext bytes): the richtext value (TextHandler::get_richtext_value) of every Text container in
    which something is visible at the rendered version, as one canonical JSON object {"<container id>": [spans…], …} — the checker of lm_richtext."""
    L = lib()
    data, off, doc_blob = pack(docs)
    n = len(doc_blob) - 1
    L.lo_option_richtext(1)
    try:
        if frontiers is not None:
            fb = [f or b"" for f in frontiers]
            foff = np.zeros(n + 1, dtype=np.uint64)
            foff[1:] = np.cumsum([len(f) for f in fb], dtype=np.uint64)
            fdata = np.frombuffer(b"".join(fb) or b"\0", dtype=np.uint8).copy()
            h = L.lo_batch_run_at(data.ctypes.data, off.ctypes.data, doc_blob.ctypes.data, n, fdata.ctypes.data, foff.ctypes.data, threads)
        else:
            h = L.lo_batch_run(data.ctypes.data, off.ctypes.data, doc_blob.ctypes.data, n, threads)
    finally:
        L.lo_option_richtext(0)
    out, merged = [], []
    try:
        ln = ctypes.c_uint64()
        for i in range(n):
            st = L.lo_batch_status(h, i)
            p = L.lo_batch_richtext(h, i, ctypes.byref(ln))
            out.append((st, ctypes.string_at(p, ln.value) if ln.value else b""))
            if with_merge:   # the same run's merge results (what merge_batch returns)
                p = L.lo_batch_json(h, i, ctypes.byref(ln))
                js = ctypes.string_at(p, ln.value) if ln.value else b""
                p = L.lo_batch_vv(h, i, ctypes.byref(ln))
                vv = ctypes.string_at(p, ln.value) if ln.value else b""
                merged.append((st, js, vv, L.lo_batch_pending(h, i)))
    finally:
        L.lo_batch_free(h)
    return (out, merged) if with_merge else out


def snapshot_state(blob, root_only=False):
    """(status, json bytes): the deep value the STATE section of a FastSnapshot holds, rendered without replaying its history
    (oracle/lo_state.hpp).  root_only: a shallow snapshot's state at its shallow root."""
    L = lib()
    L.lo_snapshot_state_json.restype = ctypes.c_int32
    L.lo_snapshot_state_json.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
    p, n = ctypes.c_void_p(), ctypes.c_uint64()
    st = L.lo_snapshot_state_json(blob, len(blob), 1 if root_only else 0, ctypes.byref(p), ctypes.byref(n))
    return st, (ctypes.string_at(p.value, n.value) if n.value else b"")


def state_entries(blobs):
    """(status, [(key, value)]): the state SSTable's entries of the document `blobs` (updates) import to, with their visible values
    (oracle/lo_state_write.hpp) — what loro_amd.wire.encode_snapshot(state=...) writes as a snapshot's state section."""
    import struct
    L = lib()
    L.lo_state_entries.restype = ctypes.c_int32
    L.lo_state_entries.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
    data = b"".join(blobs)
    off = np.zeros(len(blobs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(b) for b in blobs])
    p, n = ctypes.c_void_p(), ctypes.c_uint64()
    st = L.lo_state_entries(data, off.ctypes.data, len(blobs), ctypes.byref(p), ctypes.byref(n))
    raw = ctypes.string_at(p.value, n.value) if n.value else b""
    out, at = [], 0
    while at < len(raw):
        kl = struct.unpack_from("<I", raw, at)[0]; k = raw[at + 4:at + 4 + kl]; at += 4 + kl
        vl = struct.unpack_from("<I", raw, at)[0]; v = raw[at + 4:at + 4 + vl]; at += 4 + vl
        out.append((k, v))
    return st, out


class Session:
    """One resident document rendered step by step: step(new_blobs, frontiers=None) imports more blobs into the same
    document and renders it — the checker of lm_import + lm_run.  Returns (status, json, vv, pending) like merge()."""

    def __init__(self):
        L = lib()
        L.lo_session_new.restype = ctypes.c_void_p
        L.lo_session_free.argtypes = [ctypes.c_void_p]
        L.lo_session_step.restype = ctypes.c_int32
        L.lo_session_step.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint64]
        L.lo_session_pending.restype = ctypes.c_uint64
        L.lo_session_pending.argtypes = [ctypes.c_void_p]
        L.lo_session_mode.restype = ctypes.c_int32
        L.lo_session_mode.argtypes = [ctypes.c_void_p]
        for f in (L.lo_session_json, L.lo_session_vv, L.lo_session_lca):
            f.restype = ctypes.c_void_p
            f.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        L.lo_session_richtext.restype = ctypes.c_void_p
        L.lo_session_richtext.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        self.L, self.h = L, L.lo_session_new()
        self.want_richtext = False   # True: every step also renders the richtext values (richtext())

    def richtext(self):
        """the richtext values of the document as the last step left it (needs want_richtext = True before that step)"""
        ln = ctypes.c_uint64()
        p = self.L.lo_session_richtext(self.h, ctypes.byref(ln))
        return ctypes.string_at(p, ln.value) if ln.value else b""

    def step(self, new_blobs, frontiers=None):
        L = self.L
        data, off, _ = pack([list(new_blobs)])
        L.lo_option_richtext(1 if self.want_richtext else 0)
        try:
            st = L.lo_session_step(self.h, data.ctypes.data, off.ctypes.data, len(new_blobs), frontiers, len(frontiers) if frontiers else 0)
        finally:
            L.lo_option_richtext(0)
        ln = ctypes.c_uint64()
        p = L.lo_session_json(self.h, ctypes.byref(ln))
        js = ctypes.string_at(p, ln.value) if ln.value else b""
        p = L.lo_session_vv(self.h, ctypes.byref(ln))
        vv = ctypes.string_at(p, ln.value) if ln.value else b""
        return (st, js, vv, L.lo_session_pending(self.h))

    def import_info(self):
        """(DiffMode name, encoded LCA frontiers) of the last successful step's import (oplog.rs:591-615, dag.rs:487-765)"""
        ln = ctypes.c_uint64()
        p = self.L.lo_session_lca(self.h, ctypes.byref(ln))
        return MODES.get(self.L.lo_session_mode(self.h)), (ctypes.string_at(p, ln.value) if ln.value else b"")

    def close(self):
        if self.h:
            self.L.lo_session_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


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


# ---- DAG queries (oracle/lo_dag.hpp).  A test DAG is a list of nodes (peer, counter, len, lamport, [(peer, counter) deps]).
MODES = {0: "Checkout", 1: "Import", 2: "ImportGreaterUpdates", 3: "Linear"}


def _dag_arrays(nodes):
    peers = np.array([n[0] for n in nodes], dtype=np.uint64)
    ctrs = np.array([n[1] for n in nodes], dtype=np.int32)
    lens = np.array([n[2] for n in nodes], dtype=np.int32)
    lams = np.array([n[3] for n in nodes], dtype=np.uint32)
    off = np.zeros(len(nodes) + 1, dtype=np.uint32)
    dp, dc = [], []
    for i, n in enumerate(nodes):
        for (p, c) in n[4]:
            dp.append(p); dc.append(c)
        off[i + 1] = len(dp)
    return peers, ctrs, lens, lams, off, np.array(dp + [0], dtype=np.uint64), np.array(dc + [0], dtype=np.int32)


def _ids(ids):
    return np.array([p for p, _ in ids] + [0], dtype=np.uint64), np.array([c for _, c in ids] + [0], dtype=np.int32)


def dag_lca(nodes, left, right):
    """find_common_ancestor(left, right) → (sorted LCA frontiers, DiffMode name)  (dag.rs:318-332,487-765)"""
    L = lib()
    a = _dag_arrays(nodes)
    lp, lc = _ids(left)
    rp, rc = _ids(right)
    op = np.zeros(64 + len(nodes) * 4, dtype=np.uint64)
    oc = np.zeros(64 + len(nodes) * 4, dtype=np.int32)
    on = ctypes.c_uint32(0)
    L.lo_dag_lca.restype = ctypes.c_int32
    m = L.lo_dag_lca(ctypes.c_uint32(len(nodes)), *[ctypes.c_void_p(x.ctypes.data) for x in a],
                     ctypes.c_uint32(len(left)), ctypes.c_void_p(lp.ctypes.data), ctypes.c_void_p(lc.ctypes.data),
                     ctypes.c_uint32(len(right)), ctypes.c_void_p(rp.ctypes.data), ctypes.c_void_p(rc.ctypes.data),
                     ctypes.c_void_p(op.ctypes.data), ctypes.c_void_p(oc.ctypes.data), ctypes.byref(on))
    assert m >= 0
    return sorted(zip(op[:on.value].tolist(), oc[:on.value].tolist())), MODES[m]


def dag_ancestors(nodes, ids):
    """brute force: the set of ids that are ancestors of (or equal to) one of `ids`  (dag.rs:955-985)"""
    L = lib()
    a = _dag_arrays(nodes)
    ip, ic = _ids(ids)
    cap = sum(n[2] for n in nodes) + 8
    op = np.zeros(cap, dtype=np.uint64)
    oc = np.zeros(cap, dtype=np.int32)
    L.lo_dag_ancestors.restype = ctypes.c_int64
    n = L.lo_dag_ancestors(ctypes.c_uint32(len(nodes)), *[ctypes.c_void_p(x.ctypes.data) for x in a],
                           ctypes.c_uint32(len(ids)), ctypes.c_void_p(ip.ctypes.data), ctypes.c_void_p(ic.ctypes.data),
                           ctypes.c_void_p(op.ctypes.data), ctypes.c_void_p(oc.ctypes.data), ctypes.c_uint64(cap))
    assert 0 <= n <= cap
    return set(zip(op[:n].tolist(), oc[:n].tolist()))


def import_modes(blobs):
    """DiffMode of each LoroDoc::import when the blobs are imported one after another (oplog.rs:591-615)"""
    L = lib()
    data, off, _ = pack([list(blobs)])
    modes = np.full(len(blobs), -1, dtype=np.int32)
    L.lo_import_modes.restype = ctypes.c_int32
    rc = L.lo_import_modes(ctypes.c_void_p(data.ctypes.data), ctypes.c_void_p(off.ctypes.data), ctypes.c_uint32(len(blobs)), ctypes.c_void_p(modes.ctypes.data))
    assert rc == 0
    return [MODES[int(m)] for m in modes]
