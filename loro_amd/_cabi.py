"""ctypes view of the C ABI in include/loro_merge.h, shared by the product binding (loro_amd) and the
kernel-logic test harness (tests/emu).  `prefix` selects the exported symbol family."""
import ctypes
import os


class DocIn(ctypes.Structure):
    _fields_ = [("blobs", ctypes.POINTER(ctypes.c_char_p)), ("blob_lens", ctypes.POINTER(ctypes.c_size_t)), ("n_blobs", ctypes.c_size_t),
                ("checkout_frontiers", ctypes.c_char_p), ("checkout_len", ctypes.c_size_t)]


class DocOut(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int32), ("json", ctypes.c_void_p), ("json_len", ctypes.c_size_t), ("vv", ctypes.c_void_p),
                ("vv_len", ctypes.c_size_t), ("pending_ops", ctypes.c_uint64)]


class RunStats(ctypes.Structure):
    _fields_ = [("n_docs", ctypes.c_uint64), ("n_blobs", ctypes.c_uint64), ("in_bytes", ctypes.c_uint64), ("out_bytes", ctypes.c_uint64),
                ("device_bytes_allocated", ctypes.c_uint64), ("n_kernels", ctypes.c_uint32)]


SYMBOLS = ["create", "destroy", "last_error", "merge_batch", "stage", "run", "fetch", "get_stats", "set_profiling", "kernel_time", "selftest", "result_meta", "result_hashes", "n_streams", "run_async", "wait",
           "encode_block", "encode_updates", "free_bytes", "import", "resident_fresh", "import_modes", "import_lca", "export", "comm_unique_id", "comm_init", "summary_allgather",
           "summary_layout", "summary_rows_device", "summary_allgather_device", "shared_documents", "richtext", "richtext_result", "fused_documents", "redo_documents", "state_documents", "host_alloc", "host_free", "staged_direct"]


class Binding:
    def __init__(self, so_path, prefix):
        self.lib = ctypes.CDLL(so_path)
        lenient = os.environ.get("LM_BINDING_LENIENT") == "1"   # (tests/tools A/B runs against a library of an older commit)

        def g(n):
            try:
                return getattr(self.lib, prefix + n)
            except AttributeError:
                if not lenient:
                    raise

                class _Missing:   # attribute sink that fails at the call, not at load
                    restype = None; argtypes = None
                    def __call__(self, *a):
                        raise RuntimeError("%s does not export %s%s" % (so_path, prefix, n))
                return _Missing()
        self.create = g("create"); self.create.restype = ctypes.c_void_p; self.create.argtypes = [ctypes.c_int]
        self.destroy = g("destroy"); self.destroy.argtypes = [ctypes.c_void_p]
        self.last_error = g("last_error"); self.last_error.restype = ctypes.c_char_p; self.last_error.argtypes = [ctypes.c_void_p]
        self.merge_batch = g("merge_batch"); self.merge_batch.restype = ctypes.c_int
        self.merge_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(DocIn), ctypes.c_size_t, ctypes.POINTER(DocOut)]
        self.stage = g("stage"); self.stage.restype = ctypes.c_int
        self.stage.argtypes = [ctypes.c_void_p, ctypes.POINTER(DocIn), ctypes.c_size_t]
        self.import_ = g("import"); self.import_.restype = ctypes.c_int
        self.import_.argtypes = [ctypes.c_void_p, ctypes.POINTER(DocIn), ctypes.c_size_t]
        self.import_modes = g("import_modes"); self.import_modes.restype = ctypes.c_int; self.import_modes.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.import_lca = g("import_lca"); self.import_lca.restype = ctypes.c_long
        self.import_lca.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
        self.export = g("export"); self.export.restype = ctypes.c_int
        self.export.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
        self.richtext = g("richtext"); self.richtext.restype = ctypes.c_int; self.richtext.argtypes = [ctypes.c_void_p]
        self.richtext_result = g("richtext_result"); self.richtext_result.restype = ctypes.c_int
        self.richtext_result.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
        self.comm_unique_id = g("comm_unique_id"); self.comm_unique_id.restype = ctypes.c_int; self.comm_unique_id.argtypes = [ctypes.c_char_p]
        self.comm_init = g("comm_init"); self.comm_init.restype = ctypes.c_int; self.comm_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
        self.summary_allgather = g("summary_allgather"); self.summary_allgather.restype = ctypes.c_long
        self.summary_allgather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        self.summary_layout = g("summary_layout"); self.summary_layout.restype = ctypes.c_int
        self.summary_layout.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_size_t]
        self.summary_rows_device = g("summary_rows_device"); self.summary_rows_device.restype = ctypes.c_void_p; self.summary_rows_device.argtypes = [ctypes.c_void_p]
        self.summary_allgather_device = g("summary_allgather_device"); self.summary_allgather_device.restype = ctypes.c_long
        self.summary_allgather_device.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        self.resident_fresh = g("resident_fresh"); self.resident_fresh.restype = ctypes.c_int; self.resident_fresh.argtypes = [ctypes.c_void_p]
        self.shared_documents = g("shared_documents"); self.shared_documents.restype = ctypes.c_int; self.shared_documents.argtypes = [ctypes.c_void_p]
        self.host_alloc = g("host_alloc"); self.host_alloc.restype = ctypes.c_void_p; self.host_alloc.argtypes = [ctypes.c_size_t]
        self.host_free = g("host_free"); self.host_free.restype = None; self.host_free.argtypes = [ctypes.c_void_p]
        self.staged_direct = g("staged_direct"); self.staged_direct.restype = ctypes.c_int; self.staged_direct.argtypes = [ctypes.c_void_p]
        self.state_documents = g("state_documents"); self.state_documents.restype = ctypes.c_int; self.state_documents.argtypes = [ctypes.c_void_p]
        self.fused_documents = g("fused_documents"); self.fused_documents.restype = ctypes.c_int; self.fused_documents.argtypes = [ctypes.c_void_p]
        self.redo_documents = g("redo_documents"); self.redo_documents.restype = ctypes.c_int; self.redo_documents.argtypes = [ctypes.c_void_p]
        self.run = g("run"); self.run.restype = ctypes.c_int; self.run.argtypes = [ctypes.c_void_p]
        self.run_async = g("run_async"); self.run_async.restype = ctypes.c_int; self.run_async.argtypes = [ctypes.c_void_p]
        self.wait = g("wait"); self.wait.restype = ctypes.c_int; self.wait.argtypes = [ctypes.c_void_p]
        self.fetch = g("fetch"); self.fetch.restype = ctypes.c_int; self.fetch.argtypes = [ctypes.c_void_p, ctypes.POINTER(DocOut)]
        self.get_stats = g("get_stats"); self.get_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(RunStats)]
        self.set_profiling = g("set_profiling"); self.set_profiling.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self.result_meta = g("result_meta"); self.result_meta.restype = ctypes.c_int
        self.result_meta.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        # export / encode side (host only)
        self.encode_block = g("encode_block"); self.encode_block.restype = ctypes.c_int
        self.encode_block.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
        self.encode_updates = g("encode_updates"); self.encode_updates.restype = ctypes.c_int
        self.encode_updates.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
        self.free_bytes = g("free_bytes"); self.free_bytes.argtypes = [ctypes.c_void_p]
        self.result_hashes = g("result_hashes"); self.result_hashes.restype = ctypes.c_int
        self.result_hashes.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.selftest = g("selftest"); self.selftest.restype = ctypes.c_int; self.selftest.argtypes = [ctypes.c_void_p]
        self.n_streams = g("n_streams"); self.n_streams.restype = ctypes.c_int; self.n_streams.argtypes = [ctypes.c_void_p]
        self.kernel_time = g("kernel_time"); self.kernel_time.restype = ctypes.c_int
        self.kernel_time.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_double)]


def frame_updates(binding, blocks):
    """lm_encode_updates: encoded change blocks → one FastUpdates blob (envelope + checksum)."""
    ptrs = (ctypes.c_char_p * max(1, len(blocks)))(*blocks)
    lens = (ctypes.c_size_t * max(1, len(blocks)))(*[len(b) for b in blocks])
    out, n = ctypes.c_void_p(), ctypes.c_size_t()
    if binding.encode_updates(ptrs, lens, len(blocks), ctypes.byref(out), ctypes.byref(n)) != 0:
        raise RuntimeError("lm_encode_updates failed")
    b = ctypes.string_at(out.value, n.value)
    binding.free_bytes(out)
    return b


class Context:
    """Thin object wrapper: stage/run/fetch over lists of lists of bytes."""

    def __init__(self, binding, device=0):
        self.b = binding
        self.device = device
        self.h = binding.create(device)
        if not self.h:
            raise RuntimeError("lm_create failed: no usable HIP device (there is no CPU fallback)")
        self.n = 0

    def close(self):
        if self.h:
            self.b.destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @staticmethod
    def _pack(docs, frontiers=None):
        n = len(docs)
        assert frontiers is None or len(frontiers) == n
        arr = (DocIn * max(n, 1))()
        keep = []
        for i, blobs in enumerate(docs):
            blobs = [bytes(x) for x in blobs]
            ptrs = (ctypes.c_char_p * max(len(blobs), 1))(*blobs)
            lens = (ctypes.c_size_t * max(len(blobs), 1))(*[len(x) for x in blobs])
            keep.append((blobs, ptrs, lens))
            arr[i].blobs = ctypes.cast(ptrs, ctypes.POINTER(ctypes.c_char_p))
            arr[i].blob_lens = lens
            arr[i].n_blobs = len(blobs)
            f = frontiers[i] if frontiers is not None else None
            if f is not None:
                f = bytes(f)
                keep.append(f)
                arr[i].checkout_frontiers = f
                arr[i].checkout_len = len(f)
        return arr, (n, keep)

    def pack_pinned(self, docs, frontiers=None):
        """_pack with the blobs copied ONCE into pinned memory of lm_host_alloc (16-byte aligned, in staging order, zero padded): what a
        host that receives its blobs into such a region hands to lm_stage — the batch is then staged without the host-side gather
        (include/loro_merge.h "Direct staging").  Returns (arr, keep) like _pack; release with free_pinned(packed)."""
        n = len(docs)
        total = sum(((len(x) + 15) & ~15) for blobs in docs for x in blobs) + 64
        base = self.b.host_alloc(total)
        if not base:
            raise MemoryError("lm_host_alloc failed")
        ctypes.memset(base, 0, total)
        arr = (DocIn * max(n, 1))()
        keep = [("pinned", base)]
        at = 0
        for i, blobs in enumerate(docs):
            ptrs = (ctypes.c_void_p * max(len(blobs), 1))()
            lens = (ctypes.c_size_t * max(len(blobs), 1))(*[len(x) for x in blobs])
            for k, x in enumerate(blobs):
                ctypes.memmove(base + at, bytes(x), len(x))
                ptrs[k] = base + at
                at += (len(x) + 15) & ~15
            keep.append((ptrs, lens))
            arr[i].blobs = ctypes.cast(ptrs, ctypes.POINTER(ctypes.c_char_p))
            arr[i].blob_lens = lens
            arr[i].n_blobs = len(blobs)
            f = frontiers[i] if frontiers is not None else None
            if f is not None:
                f = bytes(f)
                keep.append(f)
                arr[i].checkout_frontiers = f
                arr[i].checkout_len = len(f)
        return arr, (n, keep)

    def free_pinned(self, packed):
        tag, base = packed[1][1][0]
        assert tag == "pinned"
        self.b.host_free(base)

    def stage(self, docs, frontiers=None):
        """docs: list of lists of update blobs.  frontiers: optional list (one per document) of None or encoded
        Frontiers (loro_amd.wire.encode_frontiers) = render the state at that version (LoroDoc::checkout)."""
        arr, keep = self._pack(docs, frontiers)
        self.n = len(docs)
        self._sum_rows = 0   # (lm_stage drops the summary layout: lm_summary_layout is called again for the new batch)
        if self.b.stage(self.h, arr, self.n) != 0:
            raise RuntimeError(self.b.last_error(self.h).decode())
        del keep  # the engine copied the blobs into its staging buffer

    def import_more(self, docs, frontiers=None):
        """lm_import: more blobs (docs[i] may be empty) and the checkout (None = latest) of the next run for every document of
        the resident batch — LoroDoc::import on a document that already holds history, then LoroDoc::checkout."""
        assert len(docs) == self.n
        arr, keep = self._pack(docs, frontiers)
        if self.b.import_(self.h, arr, self.n) != 0:
            raise RuntimeError(self.b.last_error(self.h).decode())
        del keep

    def import_packed(self, packed):
        arr, keep = packed
        if self.b.import_(self.h, arr, self.n) != 0:
            raise RuntimeError(self.b.last_error(self.h).decode())

    def step(self, docs, frontiers=None):
        """import_more + run + fetch"""
        self.import_more(docs, frontiers)
        self.run()
        return self.fetch()

    def export(self, doc, from_vv=None):
        """lm_export: the updates document `doc` holds beyond the version `from_vv` (VersionVector::encode bytes; None = all)"""
        out, n = ctypes.c_void_p(), ctypes.c_size_t()
        if self.b.export(self.h, doc, from_vv, len(from_vv) if from_vv else 0, ctypes.byref(out), ctypes.byref(n)) != 0:
            raise RuntimeError(self.b.last_error(self.h).decode())
        b = ctypes.string_at(out.value, n.value)
        self.b.free_bytes(out)
        return b

    def richtext(self):
        """lm_richtext + lm_richtext_result for every document of the last run: [(status, bytes)] — the richtext values
        (get_richtext_value) of the document's Text containers as one JSON object {"<container id>": [spans]}"""
        if self.b.richtext(self.h) != 0:
            raise RuntimeError(self.b.last_error(self.h).decode())
        out = []
        st, p, n = ctypes.c_int32(), ctypes.c_void_p(), ctypes.c_size_t()
        for i in range(self.n):
            if self.b.richtext_result(self.h, i, ctypes.byref(st), ctypes.byref(p), ctypes.byref(n)) != 0:
                raise RuntimeError(self.b.last_error(self.h).decode())
            out.append((st.value, ctypes.string_at(p.value, n.value) if n.value else b""))
        return out

    def comm_init(self, rank=0, world=1, unique_id=None):
        if self.b.comm_init(self.h, rank, world, unique_id or bytes(128)) != 0:
            raise RuntimeError(self.b.last_error(self.h).decode())

    def summary_allgather(self, doc_ids, total_docs):
        """lm_summary_allgather: the [total_docs, 6] int64 summary table of all ranks' documents (this rank's ids = doc_ids)"""
        import numpy as np
        ids = np.asarray(doc_ids, dtype=np.int64)
        table = np.zeros((total_docs, 6), dtype=np.int64)
        n = self.b.summary_allgather(self.h, ids.ctypes.data, table.ctypes.data, total_docs)
        if n < 0:
            raise RuntimeError(self.b.last_error(self.h).decode())
        return table[:n]

    def summary_layout(self, first_id, stride, rows_padded):
        """lm_summary_layout: from now on every run writes this context's summary rows on the device (ids first_id + i * stride,
        rows_padded rows, the ones beyond the staged documents -1)"""
        if self.b.summary_layout(self.h, first_id, stride, rows_padded) != 0:
            raise RuntimeError(self.b.last_error(self.h).decode())
        self._sum_rows = rows_padded

    def summary_rows_ptr(self):
        """(device address, rows) of the summary rows the last run wrote (lm_summary_rows_device)"""
        ptr = self.b.summary_rows_device(self.h)
        if not ptr:
            raise RuntimeError("no summary rows: lm_summary_layout has not been called for the batch staged last")
        return ptr, getattr(self, "_sum_rows", 0)

    def import_info(self):
        """[(DiffMode name or None, encoded LCA Frontiers or None)] of the last run's import, per resident document"""
        import numpy as np
        modes = np.full(self.n, -1, dtype=np.int32)
        self.b.import_modes(self.h, modes.ctypes.data)
        names = {0: "Checkout", 1: "Import", 2: "ImportGreaterUpdates", 3: "Linear"}
        out = []
        buf = ctypes.create_string_buffer(4096)
        for i in range(self.n):
            n = self.b.import_lca(self.h, i, buf, 4096)
            out.append((names.get(int(modes[i])), buf.raw[:n] if n >= 0 else None))
        return out

    def resident_fresh(self):
        """documents of the last run that were replayed from the empty version (no usable resident tracker)"""
        return self.b.resident_fresh(self.h)

    def stage_packed(self, packed):
        """lm_stage on an lm_doc_in array prepared once with Context._pack (bench.py: the host-side call alone)"""
        arr, keep = packed
        self.n = len(keep) if not isinstance(keep, tuple) else keep[0]
        if self.b.stage(self.h, arr, self.n) != 0:
            raise RuntimeError(self.b.last_error(self.h).decode())

    def run(self):
        if self.b.run(self.h) != 0:
            raise RuntimeError(self.b.last_error(self.h).decode())

    def run_async(self):
        if self.b.run_async(self.h) != 0:
            raise RuntimeError(self.b.last_error(self.h).decode())

    def wait(self):
        if self.b.wait(self.h) != 0:
            raise RuntimeError(self.b.last_error(self.h).decode())

    def fetch_raw(self):
        """lm_fetch alone (JSON + VV copied to host memory owned by the library), without building Python objects"""
        if getattr(self, "_outs", None) is None or len(self._outs) < max(self.n, 1):
            self._outs = (DocOut * max(self.n, 1))()
        if self.b.fetch(self.h, self._outs) != 0:
            raise RuntimeError(self.b.last_error(self.h).decode())

    def fetch(self):
        outs = (DocOut * max(self.n, 1))()
        if self.b.fetch(self.h, outs) != 0:
            raise RuntimeError(self.b.last_error(self.h).decode())
        res = []
        for i in range(self.n):
            o = outs[i]
            js = ctypes.string_at(o.json, o.json_len) if o.json_len else b""
            vv = ctypes.string_at(o.vv, o.vv_len) if o.vv_len else b""
            res.append((o.status, js, vv, o.pending_ops))
        return res

    def result_meta(self):
        """(status i32[n], json_len u64[n], vv_len u64[n], pending u64[n]) of the last run, as numpy arrays."""
        import numpy as np
        st = np.zeros(self.n, dtype=np.int32)
        jl = np.zeros(self.n, dtype=np.uint64)
        vl = np.zeros(self.n, dtype=np.uint64)
        pe = np.zeros(self.n, dtype=np.uint64)
        if self.b.result_meta(self.h, st.ctypes.data, jl.ctypes.data, vl.ctypes.data, pe.ctypes.data) != 0:
            raise RuntimeError(self.b.last_error(self.h).decode())
        return st, jl, vl, pe

    def result_hashes(self):
        """xxh64 (seed 0) of every document's JSON, computed on the device by the last run (u64[n]; 0 = failed document)"""
        import numpy as np
        h = np.zeros(self.n, dtype=np.uint64)
        if self.b.result_hashes(self.h, h.ctypes.data) != 0:
            raise RuntimeError(self.b.last_error(self.h).decode())
        return h

    def sizing(self):
        """(max leaves used, max leaf capacity, max elements, documents re-run with the worst-case directory,
        documents whose JSON overflowed the optimistic output slab and was re-rendered at its exact size)"""
        out = (ctypes.c_uint32 * 5)()
        f = getattr(self.b.lib, [k for k in ("lm_sizing", "lmemu_sizing") if hasattr(self.b.lib, k)][0])
        f.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        f(self.h, out)
        return tuple(out)

    def merge_batch(self, docs, frontiers=None):
        self.stage(docs, frontiers)
        self.run()
        return self.fetch()

    def set_profiling(self, on=True):
        """False/0 off | True/1 stage events with the context's streams run one after the other | 2 events, streams overlapped"""
        self.b.set_profiling(self.h, int(on))

    def stats(self):
        s = RunStats()
        self.b.get_stats(self.h, ctypes.byref(s))
        return s

    def kernel_times(self):
        s = self.stats()
        out = []
        name = ctypes.c_char_p()
        ms = ctypes.c_double()
        for i in range(s.n_kernels):
            if self.b.kernel_time(self.h, i, ctypes.byref(name), ctypes.byref(ms)) == 0:
                out.append((name.value.decode(), ms.value))
        return out
