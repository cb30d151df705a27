#!/usr/bin/env python
"""bench.py — merged docs/s of the batched CRDT merge engine on MI355X (BASELINE.json metric).

One "step" = one pass of the device hot path (envelope/xxh32 → columnar decode → DAG → Event-Graph-Walker
integrate → LWW → JSON + VersionVector) over one batch of synthetic documents already resident in HBM.

Workload at every N: BASELINE.json configs[1] — 10,000 documents × 100k-op automerge-paper-shaped text trace,
2 concurrent peers (base [0,50k) by A; A and B both apply [50k,75k) concurrently), three update blobs per
document — PER GPU (weak scaling: rank r owns its own 10k documents).  Documents shard with no data-path
collective; the single exchange is one all-gather of the per-document merged-state summary per step.

Steps are issued the way a merge server would issue batches: `--inflight` contexts (default 2) each hold the batch in
HBM and alternate, so the decode stages of step i+1 run beside the integrate kernels of step i (double buffering).  Every
step is still one complete pipeline pass over 10,000 documents; all K steps are complete when the timed region closes.
`--inflight 1` runs them strictly one after the other.

    python bench.py                       # N=1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

The JSON line carries, beside the contract fields: `roofline` (dominant kernel: algorithmic bytes per launch over its
HIP-event duration in the timed region and alone; `traffic` from the committed rocprofv3 --pmc passes, named in
`traffic_source`), `cpu_baseline` (the oracle — a CPU restatement of the reference path, kind "port" — as one
single-threaded process per core the container's cgroup grants, with the 1-core and half-cores points), and at N=1:
`end_to_end` (every step stages the blobs from host memory and fetches JSON + VV back: PCIe-inclusive) and
`other_configs` (BASELINE configs[0], [2], [3], [4], a heterogeneous configs[1] batch and — not a BASELINE config, guarded —
a MovableList batch for SURVEY §8f N4, each checked against the
oracle before it is timed).  `--no-cpu-baseline --no-end-to-end --no-other-configs` leaves the timed steps only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

_T0 = time.time()


def note(msg):
    """progress on stderr (the JSON line on stdout stays the only stdout output)"""
    print("[bench %6.1fs] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ≈6300 GB/s achievable


def build_docs(n_docs, first_doc, n_base, n_branch, commit_every, seed):
    from loro_amd import workload
    tpl = workload.Cfg2Template(n_base, n_branch, seed=seed, commit_every=commit_every, fuse=True)
    return tpl, [tpl.stamp(first_doc + d) for d in range(n_docs)]


_CPU_DOCS = None   # set before the workers fork: they read their slice copy-on-write


def _cpu_worker(w, lo, hi, reps, barrier, q):
    """One single-threaded oracle process: merges its slice of the benchmark documents `reps` times."""
    import _oracle
    packed = _oracle.pack(_CPU_DOCS[lo:hi])
    _oracle.merge_batch(None, threads=1, packed=packed)            # warm: library paged in, heap grown
    barrier.wait()
    t0 = time.monotonic()
    ok = True
    for _ in range(reps):
        res = _oracle.merge_batch(None, threads=1, packed=packed)
        ok = ok and all(r[0] == 0 for r in res)
    q.put((w, (hi - lo) * reps, t0, time.monotonic(), ok))


def _cpu_procs(docs, procs, per_proc, reps):
    """docs/s of `procs` independent single-threaded oracle processes (one address space each: no shared heap, no mmap
    lock), all released by one barrier; throughput = merges done / (last finish - first start)."""
    import multiprocessing as mp
    global _CPU_DOCS
    _CPU_DOCS = docs
    ctx = mp.get_context("fork")
    barrier, q = ctx.Barrier(procs), ctx.Queue()
    ps = []
    for w in range(procs):
        lo = (w * per_proc) % max(1, len(docs) - per_proc + 1)
        ps.append(ctx.Process(target=_cpu_worker, args=(w, lo, lo + per_proc, reps, barrier, q)))
    for p_ in ps:
        p_.start()
    out = [q.get() for _ in ps]
    for p_ in ps:
        p_.join()
    assert all(o[4] for o in out)
    n = sum(o[1] for o in out)
    return n / (max(o[3] for o in out) - min(o[2] for o in out)), n


def host_cores():
    """(cores this process may use, hardware threads of the box, why): the container's cgroup CPU quota and affinity mask
    bound what a CPU deployment on this box could use from here."""
    hw = os.cpu_count() or 1
    n, why = hw, "all hardware threads"
    try:
        aff = len(os.sched_getaffinity(0))
        if aff < n:
            n, why = aff, f"affinity mask of {aff} CPUs"
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, per = open(path).read().split()[:2]
            if q != "max":
                lim = max(1, int(int(q) / int(per)))
                if lim < n:
                    n, why = lim, f"cgroup cpu.max = {q} {per} ({lim} CPUs)"
        except Exception:
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and q // per < n:
            n, why = max(1, q // per), f"cgroup cfs quota {q}/{per}"
    except Exception:
        pass
    return n, hw, why


def cpu_baseline(docs, sample, target_s=6.0):
    """The CPU restatement of the reference algorithm (oracle/, kind "port") timed on this box's host cores on a
    bounded sample of the same workload.  Reported next to the GPU number; never the thing measured above.
    Documents are independent, so the port is run the way a CPU deployment would scale it: one single-threaded process
    per core (threads in one process stop scaling early: shared glibc heap and mmap lock).  The box shows 256 hardware
    threads but the container's cgroup grants a fraction of them (cpu.max): more processes than that are throttled, not
    faster — measured at 1 process, at half and at all of the CPUs the container may use; the best is `value`."""
    import _oracle
    cores, hw, why = host_cores()
    sample_docs = docs[:sample]
    one = _oracle.pack(sample_docs[:16])
    res = _oracle.merge_batch(None, threads=1, packed=one)           # warm + the results bench.py cross-checks
    t1 = time.perf_counter()
    _oracle.merge_batch(None, threads=1, packed=one)
    per_doc = (time.perf_counter() - t1) / 16
    assert all(r[0] == 0 for r in res)
    per_proc = 16
    reps = max(1, int(target_s / (per_doc * per_proc)))
    runs = {}
    for procs in sorted({1, max(1, cores // 2), cores}):
        rate, n = _cpu_procs(sample_docs, procs, per_proc, reps if procs > 1 else max(1, reps // 2))
        runs[procs] = (rate, n)
    best = max(runs, key=lambda k: runs[k][0])
    one_rate = runs[1][0]
    return {
        "value": round(runs[best][0], 1), "unit": "docs/s", "cores": best, "kind": "port",
        "sample": f"{runs[best][1]} merges of the benchmark documents (same blobs): {best} single-threaded processes of "
                  f"oracle/liblorooracle.so x {per_proc} documents x {reps} passes, released together; "
                  + "; ".join(f"{k} proc: {v[0]:.0f} docs/s" for k, v in sorted(runs.items()))
                  + f" (the box has {hw} hardware threads; this container may use {cores}: {why})",
        "scaling": {str(k): round(v[0], 1) for k, v in sorted(runs.items())},
        "cores_available": cores, "hardware_threads": hw,
        "linear_extrapolation_to_physical_cores": {"cores": hw // 2, "value": round(one_rate * (hw // 2), 1),
                                                   "note": "1-process rate x physical cores: an upper bound for this port on the whole box, not a measurement"},
    }, res


class StepLoop:
    """The serving loop of one rank: `inflight` contexts hold their batch in HBM and alternate; finishing a step = wait for
    the device pipeline, read the per-document summary (status, pending, lengths + the xxh64 of the JSON computed ON THE
    DEVICE — no JSON crosses PCIe) and, with more than one rank, all-gather it (the single exchange step).
    tests/test_dist.py drives this same class under gloo with the kernel-logic harness standing in for the GPU."""

    def __init__(self, engs, doc_ids, world, dev):
        self.engs, self.doc_ids, self.world, self.dev = engs, doc_ids, world, dev
        self.busy = [False] * len(engs)
        self.timing = {"on": False, "k": {}}
        # The exchange step as ONE collective on device memory (lm_summary_layout): every run writes this rank's summary rows
        # with a kernel, the all-gather sends that buffer as it is (equal counts: rows per rank are computed, not exchanged)
        # and the gathered table stays on the device until host_table() is asked for it.  Document ids must be an arithmetic
        # progression (first + i * stride) — bench.py's and the `doc % world` dealing of loro_amd.dist both are.
        self.dev_rows = False
        self.gather_out = [None] * len(engs)
        n = len(doc_ids)
        stride = (doc_ids[1] - doc_ids[0]) if n > 1 else 1
        if n and all(doc_ids[i] == doc_ids[0] + i * stride for i in range(n)):
            try:
                for e in engs:
                    e.summary_layout(doc_ids[0], stride, n)
                self.dev_rows = True
            except Exception as ex:   # (an older library: the host-table exchange below)
                note(f"summary rows on the device unavailable ({ex}): host-table exchange")

    def finish(self, k):
        from loro_amd import dist as lmdist
        e = self.engs[k]
        e.wait()
        self.busy[k] = False
        if self.timing["on"]:
            for name, ms in e.kernel_times():       # HIP events of this step, recorded without host syncs
                self.timing["k"].setdefault(name, []).append(ms)
        if self.world > 1 and self.dev_rows:
            ptr, rows = e.summary_rows_ptr()
            send = lmdist.rows_tensor(ptr, rows, self.dev)
            self.gather_out[k] = lmdist.all_gather_rows(send, self.gather_out[k])   # the single collective; result left on the device
            return self.gather_out[k]
        st, jl, vl, pe = e.result_meta()
        local = lmdist.summarize_device(self.doc_ids, st, pe, jl, vl, e.result_hashes())
        if self.world > 1:
            return lmdist.all_gather_summaries(local, device=self.dev)
        return local

    @staticmethod
    def host_table(out):
        """the summary table on the host, ordered by document id (a gathered device tensor is read back here, when asked)"""
        from loro_amd import dist as lmdist
        return out if not hasattr(out, "cpu") else lmdist.table_of(out)

    def device_rows_check(self, k=0):
        """the rows the last run of context k wrote on the device (through the tensor view the collective sends) against the
        host-side summary of the same run — run at N=1 too, so the device path is exercised wherever bench.py runs"""
        from loro_amd import dist as lmdist
        e = self.engs[k]
        st, jl, vl, pe = e.result_meta()
        want = lmdist.summarize_device(self.doc_ids, st, pe, jl, vl, e.result_hashes())
        ptr, rows = e.summary_rows_ptr()
        got = lmdist.table_of(lmdist.rows_tensor(ptr, rows, self.dev))
        want = want[want[:, 0].argsort(kind="stable")]
        return got.shape == want.shape and bool((got == want).all())

    def run_steps(self, n):
        """n steps; step i runs on context i % inflight, which is first drained of its previous step"""
        out = None
        for i in range(n):
            k = i % len(self.engs)
            if self.busy[k]:
                out = self.finish(k)
            self.engs[k].run_async()            # device pipeline of one batch; returns at once
            self.busy[k] = True
        for j in range(len(self.engs)):         # drain in launch order
            k = (n + j) % len(self.engs)
            if self.busy[k]:
                out = self.finish(k)
        return out


def _end_to_end_leg(engs, packed, n_docs, steps):
    """`steps` batches through the contexts in rotation: lm_stage (main thread) + lm_run (asynchronous) + lm_fetch (helper thread, behind the
    run of ITS context and beside the staging of the next).  A context is staged again only when the fetch of its previous batch is done."""
    from concurrent.futures import ThreadPoolExecutor
    t_stage = t_fetch = 0.0
    n = len(engs)
    pending = [None] * n     # the wait + fetch of that context's batch in flight
    pool = ThreadPoolExecutor(max_workers=2 if n >= 4 else 1)

    def _fetch(k):
        engs[k].wait()
        t = time.perf_counter(); engs[k].fetch_raw()
        return time.perf_counter() - t
    t0 = time.perf_counter()
    for i in range(steps):
        k = i % n
        if pending[k] is not None:
            t_fetch += pending[k].result(); pending[k] = None
        t = time.perf_counter(); engs[k].stage_packed(packed); t_stage += time.perf_counter() - t
        engs[k].run_async()
        pending[k] = pool.submit(_fetch, k)      # (a fetch waits for the run of ITS context; with four contexts two helpers keep two fetches in flight)
    for k in range(n):
        if pending[k] is not None:
            t_fetch += pending[k].result(); pending[k] = None
    pool.shutdown()
    dt = time.perf_counter() - t0
    return {"value": round(n_docs * steps / dt, 1), "unit": "docs/s", "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps, "contexts": n,
            "lm_stage_ms": round(t_stage / steps * 1e3, 2), "lm_fetch_ms": round(t_fetch / steps * 1e3, 2)}


def end_to_end(engs, docs, steps):
    """PCIe-inclusive rate of the same workload: every step stages the blobs from host memory, runs the pipeline and fetches JSON + VV
    back (lm_fetch).  Two feeds: blobs in PAGEABLE host memory (lm_stage gathers them into its pinned buffer: host -> pinned -> HBM),
    and blobs the host received into memory of lm_host_alloc (include/loro_merge.h "Direct staging": pinned -> HBM, no host copy).
    Four contexts in rotation (one being staged, one or two running, one being fetched) — two are created here and released again.
    The lm_doc_in arrays are built once; the timed region holds only C-ABI calls."""
    import loro_amd
    from loro_amd._cabi import Context
    extras = [loro_amd.MergeEngine(engs[0].device) for _ in range(2)]
    ring = list(engs) + extras
    try:
        packed = Context._pack(docs)
        for e in ring:
            e.stage_packed(packed); e.run(); e.fetch_raw()            # warm: staging buffers, pools
        pageable = _end_to_end_leg(ring, packed, len(docs), steps)
        pinned = ring[0].pack_pinned(docs)
        try:
            ring[0].stage_packed(pinned); ring[0].run()
            direct = bool(ring[0].b.staged_direct(ring[0].h))
            pin = _end_to_end_leg(ring, pinned, len(docs), steps)
        finally:
            for e in ring[:-len(extras)]:
                e.stage_packed(packed); e.run()      # (the region must outlive the batch staged from it: the contexts hold the pageable batch again)
            ring[0].free_pinned(pinned)
        st = engs[0].stats()
    finally:
        for e in extras:
            e.close()
    pin["staged_direct"] = direct
    out = dict(pin)
    out.update({"host_to_device_bytes_per_step": int(st.in_bytes), "device_to_host_bytes_per_step": int(st.out_bytes),
                "from_pageable_host_memory": pageable,
                "what": "lm_stage + lm_run + lm_fetch (JSON + VV -> host) per step, 4 contexts in rotation (staging, running, fetching), the fetches on two helper "
                        "threads (tests/tools/gpu_e2e.py: 3 contexts / 1 helper 337k, 4 / 1 351k, 4 / 2 362k docs/s on one box).  `value`: the blobs live in pinned memory of lm_host_alloc — lm_stage hands the span to the copy engine as it is (direct staging); "
                        "`from_pageable_host_memory`: the blobs are ordinary host allocations — lm_stage gathers them into its pinned buffer first"})
    return out


def _gen(args):
    kind, d = args
    from loro_amd import workload
    if kind == "cfg3a":
        return workload.cfg3_doc(d, combined=True), None
    if kind == "cfg3b":
        return workload.cfg3_doc(d, combined=False), None
    if kind == "cfg5":
        return workload.cfg5_doc(d, n_ops=1000000, turn=1000, n_checkouts=16)
    if kind == "movable":      # concurrent session over MovableLists (root + child, nested children), 3 peers
        import random, _fuzz
        reps = _fuzz.movable_session(7000 + d, n_peers=3, n_steps=500, sync_prob=0.08, nested=True, bulk=300)
        return _fuzz.blobs_of(reps, random.Random(d)), None
    if kind == "snap":         # a configs[1]-shaped history whose BASE (50k ops, one chain) arrives as a snapshot with a real state section
        return workload.cfg2_snapshot_doc(d), None
    if kind == "trace":        # configs[1] at its stated size from ANOTHER synthetic trace (seed d)
        return workload.Cfg2Template(50000, 25000, seed=d, commit_every=10, fuse=True), None
    if kind == "tpl":          # a configs[1]-shaped template of another size / commit granularity
        n_base, n_branch, every, fuse = d
        return workload.Cfg2Template(n_base, n_branch, seed=n_base % 97, commit_every=every, fuse=fuse), None
    raise ValueError(kind)


# stage of lm_run (Engine::times) -> the kernels rocprofv3 sees for it
_STAGE_KERNELS = {"k_block_decode": ("k_block_decode", "k_block_head", "k_block_reclassify"), "k_map_lww": ("k_map_lww", "k_map_fused"), "k_emit": ("k_emit",),
                  "k_frame_fill+k_block_count": ("k_frame_fill", "k_block_desc", "k_block_count", "k_block_kind", "k_doc_kind"), "k_doc_tables+k_remap": ("k_doc_ranges", "k_doc_tables", "k_remap"),
                  "k_frame_count": ("k_frame_count", "k_hash_big_blobs")}


def _pmc_traffic_of(name, dom, n_docs):
    """HBM traffic of the dominant STAGE of an other_configs entry from the newest committed rocprofv3 --pmc record of that entry
    (profiles/<tag>_pmc_other.json, profiles/collect_other.sh: one process per entry, separate FETCH_SIZE / WRITE_SIZE passes):
    sum over the stage's kernels of (2 x FETCH_SIZE + WRITE_SIZE) KiB x dispatches per pipeline run — the gfx950 FETCH correction of
    /opt/skills/guides/MI355X_MICROARCH.md — scaled to this run's document count.  (None, None) when no record names the stage."""
    pdir = os.path.join(ROOT, "profiles")
    for tag in sorted({f.split("_pmc_other.json")[0] for f in os.listdir(pdir) if f.endswith("_pmc_other.json")}, reverse=True):
        try:
            rec = json.load(open(os.path.join(pdir, f"{tag}_pmc_other.json"))).get(name)
        except Exception:
            continue
        if not rec:
            continue
        prefixes = _STAGE_KERNELS.get(dom, (dom,))
        tot, used = 0.0, []
        for kn, kk in rec.get("kernels", {}).items():
            if kn.startswith(prefixes) and "FETCH_SIZE_KiB_per_launch" in kk and "WRITE_SIZE_KiB_per_launch" in kk:
                tot += (2 * kk["FETCH_SIZE_KiB_per_launch"] + kk["WRITE_SIZE_KiB_per_launch"]) * 1024 * kk["calls"] / rec.get("runs_of_the_pipeline", 1)
                used.append(kn)
        if not used:
            continue
        return int(tot / rec["docs"] * n_docs), (f"profiles/{tag}_pmc_other.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `{rec.get('command')}` (not this run), "
                                                 f"(2 x FETCH_SIZE + WRITE_SIZE) KiB summed over the dispatches of {', '.join(sorted(used))} in one pipeline run, per document x {n_docs} documents")
    return None, None


def other_configs(device, cores):
    """The other BASELINE.json configs on one GPU, each checked against the oracle on its distinct documents before it
    is timed (one context, runs strictly one after the other, inputs resident in HBM): docs/s and the algorithmic
    bytes (blobs in + JSON + VV out) per second as a fraction of the HBM peak."""
    import multiprocessing as mp
    import loro_amd, _oracle, _cases
    from loro_amd import workload
    out = {}

    class _Serial:   # LM_BENCH_NO_POOL=1 (profiles/collect_other.sh): rocprofv3 hangs on a process that forks a pool before it touches the GPU
        def __enter__(self): return self
        def __exit__(self, *a): return False
        def map_async(self, f, xs):
            only = os.environ.get("LM_BENCH_ONLY", "")
            want = {"cfg3a": "configs[2]", "cfg3b": "configs[2]", "cfg5": "configs[4]", "tpl": "heterogeneous", "trace": "traces", "movable": "movable", "snap": "snapshot"}
            class R:
                def __init__(s, v): s.v = v
                def get(s): return s.v
            return R([f(x) if (not only or want.get(x[0], "") in only or (x[0] == "cfg5" and "richtext" in only)) else None for x in xs])
    with (_Serial() if os.environ.get("LM_BENCH_NO_POOL") else mp.get_context("fork").Pool(min(24, cores))) as pool:
        g3 = pool.map_async(_gen, [("cfg3a", d) for d in range(8)] + [("cfg3b", d) for d in range(8)])
        g5 = pool.map_async(_gen, [("cfg5", d) for d in range(4)])
        # heterogeneous configs[1]: 2-peer concurrent text documents of seven sizes (4k .. 200k ops), fused and keystroke-per-change
        shapes = [(2000, 1000, 10, True), (10000, 5000, 10, True), (25000, 12500, 10, True), (50000, 25000, 10, True),
                  (100000, 50000, 10, True), (5000, 2500, 1, False), (20000, 10000, 1, False)]
        gh = pool.map_async(_gen, [("tpl", sh) for sh in shapes])
        # configs[1] with 128 DIFFERENT traces (the headline batch stamps ONE trace 10,000 times: identical control flow in every wave)
        gt = pool.map_async(_gen, [("trace", sd) for sd in range(1, 129)])
        gm = pool.map_async(_gen, [("movable", d) for d in range(16)])
        gs = pool.map_async(_gen, [("snap", d) for d in range(8)])
        cfg1 = [workload.cfg1_doc(d) for d in range(100)]
        cfg4_base = _cases.cfg4_docs(96)
        g3, g5, gh, gt = g3.get(), g5.get(), gh.get(), gt.get()
        try:
            gs = gs.get()
        except Exception as ex:   # (not a BASELINE config either)
            gs = ex
        try:
            gm = gm.get()
        except Exception as ex:   # (not a BASELINE config: its generator must not take the bench line down)
            gm = ex
    note("other configs: documents generated")

    only = os.environ.get("LM_BENCH_ONLY")   # profiles/collect_other.sh: ONE entry per process (its kernels are then that entry's in the rocprofv3 record)

    def sel(name):
        if os.environ.get("LM_BENCH_ONLY_EXACT"):   # (profiles/collect_other.sh: the rocprofv3 record must hold ONE entry's kernels)
            return name == only
        return not only or name == only or name.startswith(only + ",")

    def run(name, docs, fronts, distinct, desc, reps=3, extra=None):
        if not sel(name):
            return
        # a leg that fails (its parity assert included) is REPORTED in its own entry — "error" instead of a rate — and leaves the
        # other entries and the headline value (which has its own parity assert) alone
        try:
            _run(name, docs, fronts, distinct, desc, reps, extra)
        except Exception as ex:
            out[name] = {"error": f"{type(ex).__name__}: {ex}"[:300], "workload": desc}
            note(f"other configs: {name} FAILED: {type(ex).__name__}: {ex}"[:200])

    def _run(name, docs, fronts, distinct, desc, reps, extra=None):
        t_cpu = time.perf_counter()
        want = _oracle.merge_batch(docs[:distinct], threads=min(32, cores), frontiers=None if fronts is None else fronts[:distinct])
        t_cpu = time.perf_counter() - t_cpu
        with loro_amd.MergeEngine(device) as e:
            e.stage(docs, fronts)
            e.run()
            got = e.fetch()
            n_checked = distinct
            if name.endswith(("heterogeneous", "traces")):   # every document is its own: all must succeed, and as many as the CPU port gets through in 25 s are compared
                assert got[:distinct] == want and all(g[0] == 0 for g in got), f"{name}: device results differ from the CPU oracle"
                t_chk = time.perf_counter()
                while n_checked < len(docs) and time.perf_counter() - t_chk < 25.0:
                    hi = min(len(docs), n_checked + 512)
                    assert got[n_checked:hi] == _oracle.merge_batch(docs[n_checked:hi], threads=min(32, cores)), f"{name}: device results differ from the CPU oracle (documents {n_checked}..{hi})"
                    n_checked = hi
            else:
                assert all(got[i] == want[i % distinct] for i in range(len(docs))), f"{name}: device results differ from the CPU oracle"
            best = 1e9
            for _ in range(reps):
                t = time.perf_counter(); e.run(); best = min(best, time.perf_counter() - t)
            st = e.stats()
            more = extra(e) if extra else {}
            # every stage's own duration: one more pass with the context's streams run one after the other (HIP events per stage);
            # a stage's figure is the sum over the context's streams = the time the stage needs for the whole batch
            e.set_profiling(1); e.run()
            stage_ms = {}
            for kname, ms in e.kernel_times():
                stage_ms[kname] = round(stage_ms.get(kname, 0.0) + ms, 3)
            e.set_profiling(0)
        alg = float(st.in_bytes + st.out_bytes)
        dom = max(stage_ms, key=stage_ms.get) if stage_ms else None
        # HBM traffic of the dominant stage: only where a committed rocprofv3 --pmc record of this config names the same kernel
        # (profiles/collect_cfg3.sh: 2,048 documents; the counters scale with the documents)
        traffic, traffic_src = None, None
        if dom:
            traffic, traffic_src = _pmc_traffic_of(name, dom, len(docs))
        note(f"other configs: {name} done")
        out[name] = {"docs": len(docs), "distinct_docs": distinct, "docs_per_s": round(len(docs) / best, 1), "ms_per_batch": round(best * 1e3, 2),
                     "pipeline_runs_in_this_process": 2 + reps,
                     "algorithmic_bytes": int(alg), "algorithmic_GBps": round(alg / best / 1e9, 2), "frac_of_hbm_peak": round(alg / best / 1e9 / HBM_PEAK_GBS, 5),
                     "stage_ms": stage_ms,
                     "roofline": None if not dom else {"bound": "hbm", "kernel": dom, "kernel_ms": stage_ms[dom], "achieved": round(alg / (stage_ms[dom] * 1e-3) / 1e9, 2),
                                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / (stage_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                                                       "what": "algorithmic bytes of the batch over the dominant stage's time for the whole batch (streams serialized); per stage: algorithmic bytes / stage_ms"},
                     "cpu_baseline": {"value": round(distinct / t_cpu, 1), "unit": "docs/s", "cores": min(32, cores, distinct), "kind": "port",
                                      "sample": f"the {distinct} distinct documents of this entry through oracle/liblorooracle.so, one pass, {min(32, cores, distinct)} threads of one process "
                                                "(the pass that produced the expected results; threads of one process share a heap and scale worse than the headline's one process per core)"},
                     "gpu_over_cpu": round(len(docs) / best / (distinct / t_cpu), 2),
                     "parity": (f"the first {n_checked} of {len(docs)} results equal to the oracle's (every document differs; as many as the CPU port replays in 25 s), all {len(docs)} succeeded" if name.endswith(("heterogeneous", "traces")) else f"all {len(docs)} results equal to the oracle's"),
                     "workload": desc}
        out[name].update(more)

    run("configs[0]", cfg1, None, 100, "100 docs x 2 peers x 1,000 sequential inserts, 2 blobs/doc")
    d3 = [g[0] for g in g3] if sel("configs[2]") else [[]] * 16
    run("configs[2]", [d3[i % 16] for i in range(10000)], None, 16,
        "LWW Map, 16 peers x 10,000 writes on 1,024 keys per doc (160k ops/doc); the config's 10,000 docs (24 GB of blobs): 8 distinct "
        "histories x {one combined blob, 16 per-peer blobs}", reps=2)
    run("configs[3]", [cfg4_base[i % 96] for i in range(12500)], None, 96,
        "mixed List/Map/Text roots, 4 peers x ~1k ops with pairwise syncs; 12,500 docs = one GPU's share of the config's 100k over 8")
    if sel("configs[1]-heterogeneous"):
        tpls = [g[0] for g in gh]
        mix = [tpls[(d * 7919) % len(tpls)].stamp(d) for d in range(10000)]      # sizes interleaved pseudo-randomly: neighbouring waves differ
        run("configs[1]-heterogeneous", mix, None, 64,
            "10,000 two-peer concurrent text documents of seven shapes interleaved (4k, 20k, 50k, 100k, 200k ops with fused changes; 10k and "
            f"40k ops with one change per keystroke): {sum(t.n_ops for t in tpls) // len(tpls)} ops/doc on average — per-wave load imbalance and divergent control flow")
    ttr = [g[0] for g in gt] if sel("configs[1]-128-traces") else []
    run("configs[1]-128-traces", [ttr[(d * 7919) % len(ttr)].stamp(d) for d in range(10000)] if ttr else [], None, 64,
        "configs[1] at its stated size (10,000 docs x 100k ops, 2 concurrent peers, 3 blobs) from 128 DIFFERENT synthetic traces interleaved "
        "pseudo-randomly, letters stamped per document: neighbouring waves replay different histories (the headline batch stamps one trace)")
    # a configs[1]-shaped history whose base (50k ops, one peer's chain) arrives as a SNAPSHOT and what follows as updates
    # on top of it — staged from the snapshot's state section (lm_snapshot.h / lm_snapshot_base.h): the base's history is neither uploaded
    # nor decoded nor replayed.  Three entries: (1) the updates are one branch that continues the snapshot (the whole document is one
    # chain: replayed by the linear prefix, by position); (2) the same batch with LM_SNAPSHOT_STATE=0 — the snapshot through its
    # ChangeStore, as in rounds 2-5; (3) two concurrent 25k-op branches: the cost rule (lm_snapshot_base.h `pays`) sends such a document
    # through its history — on the state every delete of base content goes through the tracker's by-position path, measured 3.5 x slower
    # (tests/tools/gpu_snapbase.py) — `state_documents` says which path the batch took
    n3 = "snapshot + updates (SURVEY 8f N3)"
    if sel(n3) or (only or "").startswith(n3):
        try:
            if isinstance(gs, Exception):
                raise gs
            ds = [g[0] for g in gs]
            sd_ = lambda eng: {"state_documents": int(eng.b.state_documents(eng.h))}
            chain = [[ds[i % 8][0], ds[i % 8][1]] for i in range(10000)]
            run(n3, chain, None, 8,
                "configs[1]-shaped documents: the 50k-op base as ONE FastSnapshot blob (real state section, stored SSTable blocks), one 25k-op branch that continues it as an "
                "update blob (75k ops per document); 10,000 docs = 8 distinct histories; staged from the state section", extra=sd_)
            os.environ["LM_SNAPSHOT_STATE"] = "0"
            try:
                run(n3 + ", history replayed", chain, None, 8,
                    "the batch above with LM_SNAPSHOT_STATE=0: the snapshot ingested through its ChangeStore (rounds 2-5), the whole history decoded and replayed", reps=2, extra=sd_)
            finally:
                del os.environ["LM_SNAPSHOT_STATE"]
            run(n3 + ", two concurrent branches", [ds[i % 8] for i in range(10000)], None, 8,
                "the 50k-op base as a snapshot + TWO concurrent 25k-op branches as update blobs (100k ops per document): the cost rule declines the state path "
                "(state_documents 0) and the snapshot's history is replayed", reps=2, extra=sd_)
        except Exception as ex:
            out[n3] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    # configs[4]: 256 document INSTANCES (own copies of the blobs of 4 distinct histories) x 16 versions.  The 16 entries of an instance
    # name the same blobs: lm_stage folds them into one document, lm_run imports it once and renders the 16 versions by moving its
    # trackers (include/loro_merge.h "Shared replay"); LM_SHARE_REPLAY=0 replays the history once per entry (rounds 1-4: 1,148 renderings/s)
    n5 = 64
    inst = [[bytes(bytearray(b)) for b in g5[d % 4][0]] for d in range(256)] if sel("configs[4]") or sel("configs[4]-one-replay-per-rendering") else [[]] * 256
    if sel("configs[4]-one-replay-per-rendering"):
        # (the round-4 formulation beside the shared one — every entry a document of its own, replayed up to its version's
        # causal closure, LM_SHARE_REPLAY=0; 64 instances x 16 versions)
        os.environ["LM_SHARE_REPLAY"] = "0"
        try:
            run("configs[4]-one-replay-per-rendering", [inst[i // 16] for i in range(1024)], [g5[(i // 16) % 4][1][i % 16] for i in range(1024)], n5,
                "the configs[4] entry below with LM_SHARE_REPLAY=0: 1,024 renderings = 64 documents x 16 versions, every rendering a replay of its own (rounds 1-4)", reps=2)
        finally:
            del os.environ["LM_SHARE_REPLAY"]
    run("configs[4]", [inst[i // 16] for i in range(4096)], [g5[(i // 16) % 4][1][i % 16] for i in range(4096)] if sel("configs[4]") else None, n5,
        "1M-op rich-text documents (2 peers alternating every 1k trace actions, ~1% marks), 16 checkouts each: 4,096 renderings = "
        "256 documents x 16 versions (of the config's 1,000 documents; 4 distinct histories); the 16 entries of a document share their "
        "blobs: imported once per lm_run, every version rendered by a move of the document's trackers", reps=2)
    # not a BASELINE config: the MovableList row of SURVEY §8f (N4), timed like the others so the row has a number on hardware;
    # guarded — a failure here is reported in its own entry and leaves the BASELINE entries and the headline value alone
    try:
        if isinstance(gm, Exception):
            raise gm
        dm = [g[0] for g in gm] if sel("movable-lists (SURVEY 8f N4)") else [[]] * 16
        run("movable-lists (SURVEY 8f N4)", [dm[i % 16] for i in range(4096)], None, 16,
            "MovableList documents: a root and a child list of ~300-500 elements, 3 peers x ~500 concurrent insert / move / set / delete ops "
            "with pairwise syncs, nested child containers; 4,096 docs = 16 distinct histories")
    except Exception as ex:
        out["movable-lists (SURVEY 8f N4)"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    # the richtext row of SURVEY §8f N4 (lm_richtext, lm_k_richtext.h): the configs[4] histories (1M ops, ~1 % marks: ≈10k StyleOps per
    # document) at the latest version, 256 document instances; timed = lm_richtext (one launch of k_richtext + the copy back), every
    # result compared with the oracle's Doc::to_richtext of its history.  Guarded like the entry above.
    name = "richtext (SURVEY 8f N4)"
    if not sel(name):
        return out
    try:
        import json as _json
        rt_docs = [[bytes(bytearray(b)) for b in g5[d % 4][0]] for d in range(256)]
        t_cpu = time.perf_counter()
        want = _oracle.richtext_batch([g5[d][0] for d in range(4)])
        t_cpu = time.perf_counter() - t_cpu
        with loro_amd.MergeEngine(device) as e:
            e.stage(rt_docs, None); e.run()
            got = e.richtext()
            assert all(g[0] == 0 for g in got) and all(w[0] == 0 for w in want)
            assert all(_json.loads(got[i][1]) == _json.loads(want[i % 4][1]) and len(got[i][1]) == len(want[i % 4][1]) for i in range(len(rt_docs))), "richtext values differ from the CPU oracle's"
            best = 1e9
            for _ in range(3):
                t = time.perf_counter(); e.richtext(); best = min(best, time.perf_counter() - t)
        rt_bytes = sum(len(g[1]) for g in got)
        out[name] = {"docs": len(rt_docs), "distinct_docs": 4, "docs_per_s": round(len(rt_docs) / best, 1), "ms_per_batch": round(best * 1e3, 2),
                     "richtext_bytes": rt_bytes, "spans_with_attributes_per_doc": got[0][1].count(b'"attributes"'),
                     "timed": "lm_richtext after lm_run: ONE launch of k_richtext into optimistic slabs (a second, at exact sizes, only when a slab overflows) + the copy back; the import itself is the configs[4] entry's",
                     "cpu_baseline": {"value": round(4 / t_cpu, 2), "unit": "docs/s", "cores": 1, "kind": "port", "sample": "import + Doc::to_richtext of the 4 distinct histories, one thread (the replay dominates)"},
                     "parity": f"all {len(rt_docs)} results equal to the oracle's",
                     "workload": "1M-op rich-text documents (2 peers alternating every 1k trace actions, ~1% of the actions are bold marks), richtext value of the root Text at the latest version; 256 instances of 4 distinct histories"}
        note(f"other configs: {name} done")
    except Exception as ex:
        out[name] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        note(f"other configs: {name} FAILED: {type(ex).__name__}: {ex}"[:200])
    return out


def resident_configs(device, cores, n_docs=10000, full=True):
    """Resident documents (lm_import), each checked against the oracle:
    configs[1]-incremental — base + A's branch resident in HBM (trackers included), B's concurrent 25k-op branch imported: the
    M REMOTE ops of the metric are all that is integrated; timed = lm_run after lm_import (the import's host-to-device copy of
    B's blobs is reported beside it);
    configs[4]-resident — 1M-op documents rendered at 16 versions: one replay, then 16 moves of the resident tracker."""
    import multiprocessing as mp
    import loro_amd, _oracle
    from loro_amd import workload
    from loro_amd._cabi import Context
    out = {}
    try:
        tpl = workload.Cfg2Template(50000, 25000, seed=0, commit_every=10, fuse=True)
        docs = [tpl.stamp(d) for d in range(n_docs)]
        base = Context._pack([b[:2] for b in docs]); inc = Context._pack([b[2:] for b in docs]); none = Context._pack([[] for _ in docs])
        want = _oracle.merge_batch(docs[:64], threads=min(32, cores))
        t_run, t_imp, t_base = [], [], []
        with loro_amd.MergeEngine(device) as e:
            kt_base, kt_inc = {}, {}
            for rep in range(4):
                prof = rep == 3   # the last repetition with the streams serialized: every stage's own duration
                e.set_profiling(1 if prof else 0)
                e.stage_packed(base); e.import_packed(none)
                t = time.perf_counter(); e.run(); dt = time.perf_counter() - t
                if prof:
                    for name, ms in e.kernel_times():
                        kt_base[name] = round(kt_base.get(name, 0.0) + ms, 3)
                else:
                    t_base.append(dt)
                assert e.resident_fresh() == n_docs
                t = time.perf_counter(); e.import_packed(inc); dt = time.perf_counter() - t
                if not prof:
                    t_imp.append(dt)
                t = time.perf_counter(); e.run(); dt = time.perf_counter() - t
                if prof:
                    for name, ms in e.kernel_times():
                        kt_inc[name] = round(kt_inc.get(name, 0.0) + ms, 3)
                else:
                    t_run.append(dt)
                assert e.resident_fresh() == 0, "the import did not continue from the resident trackers"
            e.set_profiling(0)
            got = e.fetch()
            assert got[:64] == want and all(g[0] == 0 for g in got), "configs[1]-incremental: device results differ from the CPU oracle"
            e.set_profiling(1); e.import_packed(none); e.run()   # (same version again: the per-stage times of a run that reuses its tables)
            kt_reuse = {}
            for name, ms in e.kernel_times():
                kt_reuse[name] = round(kt_reuse.get(name, 0.0) + ms, 3)
            e.set_profiling(0)
            st = e.stats()
        best = min(t_run)
        out["configs[1]-incremental"] = {
            "docs": n_docs, "docs_per_s": round(n_docs / best, 1), "ms_per_batch": round(best * 1e3, 2),
            "lm_import_ms": round(min(t_imp) * 1e3, 2), "from_empty_run_of_base_plus_A_ms": round(min(t_base) * 1e3, 2),
            "algorithmic_bytes": int(st.in_bytes + st.out_bytes),
            "parity": f"first 64 results equal to the oracle's batch of all three blobs, all {n_docs} succeeded, every document continued from its resident tracker",
            "stage_ms_of_the_import_run_streams_serialized": kt_inc, "stage_ms_of_the_from_empty_run_streams_serialized": kt_base,
            "stage_ms_of_a_run_that_reuses_its_tables": kt_reuse,
            "workload": "configs[1] documents: base + A's branch (75k ops, 2 blobs) resident with their trackers; B's concurrent branch (25k ops, 1 blob) imported by lm_import; "
                        "timed: the lm_run that follows (decode of all blobs, DAG, integrate of B's rows only after retreating A's branch, render)"}
        note("resident: configs[1]-incremental done")
    except Exception as ex:
        out["configs[1]-incremental"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        note(f"resident: configs[1]-incremental FAILED: {type(ex).__name__}: {ex}"[:300])
    if not full:
        return out
    try:
        with mp.get_context("fork").Pool(min(8, cores)) as pool:
            g5 = pool.map(_gen, [("cfg5", d) for d in range(4)])
        n5 = 1000 if full else 64   # the config's stated size: 1,000 documents x 16 versions
        docs5 = [g5[i % 4][0] for i in range(n5)]
        flat_docs, flat_fr = [], []
        for d in range(4):
            flat_docs += [g5[d][0]] * 16; flat_fr += g5[d][1]
        want = _oracle.merge_batch(flat_docs, threads=min(32, cores), frontiers=flat_fr)
        none = Context._pack([[] for _ in docs5])
        with loro_amd.MergeEngine(device) as e:
            best = 1e9
            for rep in range(2):
                t0 = time.perf_counter()
                e.stage(docs5); e.import_packed(none); e.run()
                t_replay = time.perf_counter() - t0
                for k in range(16):
                    e.import_more([[] for _ in docs5], [g5[i % 4][1][k] for i in range(n5)])
                    e.run()
                    if rep == 0:
                        got = e.fetch()
                        assert all(got[i] == want[(i % 4) * 16 + k] for i in range(n5)), "configs[4]-resident: device results differ from the CPU oracle"
                if rep:
                    best = min(best, time.perf_counter() - t0)
            # one more move (from the last version back to the eighth) with the stages timed on their own
            e.set_profiling(1)
            e.import_more([[] for _ in docs5], [g5[i % 4][1][7] for i in range(n5)]); e.run()
            kt_move = {}
            for name, ms in e.kernel_times():
                kt_move[name] = round(kt_move.get(name, 0.0) + ms, 3)
            e.set_profiling(0)
        out["configs[4]-resident"] = {
            "renderings": n5 * 16, "renderings_per_s": round(n5 * 16 / best, 1), "ms_total": round(best * 1e3, 2), "ms_replay_incl_staging": round(t_replay * 1e3, 2),
            "parity": f"all {n5 * 16} renderings equal to the oracle's",
            "stage_ms_of_one_move_and_rendering_streams_serialized": kt_move,
            "workload": f"{n5} 1M-op rich-text documents (4 distinct) — the config's stated size — each staged and replayed ONCE, then rendered at 16 versions by moving "
                        "the resident trackers (lm_import with frontiers only + lm_run, 16 times); the time includes staging (host to device), the replay and all 16 runs. "
                        "One wave per document: 1,000 documents keep 1,000 waves busy, where the batch entry above replays every rendering from the empty version"}
        note("resident: configs[4]-resident done")
    except Exception as ex:
        out["configs[4]-resident"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        note(f"resident: configs[4]-resident FAILED: {type(ex).__name__}: {ex}"[:300])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--docs", type=int, default=10000, help="documents per GPU (configs[1]: 10,000)")
    ap.add_argument("--base-ops", type=int, default=50000)
    ap.add_argument("--branch-ops", type=int, default=25000)
    ap.add_argument("--commit-every", type=int, default=10)
    ap.add_argument("--cpu-sample", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other BASELINE configs (N=1 only; ~2 min)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the PCIe-inclusive measurement (N=1 only)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="contexts in flight per GPU (double buffering: one batch's decode stages run beside the other's integrate "
                         "kernels); every step is still one full pipeline pass over one context's batch")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import loro_amd
    from loro_amd import dist as lmdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the merge engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    tpl, docs = build_docs(args.docs, rank * args.docs, args.base_ops, args.branch_ops, args.commit_every, seed=0)
    doc_ids = list(range(rank * args.docs, (rank + 1) * args.docs))
    note("documents built")
    # CPU baseline first (rank 0 at N=1 only): its worker processes are forked before this process touches the GPU
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, _ = cpu_baseline(docs, min(args.cpu_sample, len(docs)))
    note("cpu baseline done" if cpu else "no cpu baseline")
    engs = [loro_amd.MergeEngine(local_rank) for _ in range(max(1, args.inflight))]
    for e in engs:
        e.stage(docs)                     # blobs → HBM (outside the timed region); every context holds the batch
        e.run()                           # first run of a context allocates its work pools (≈47 GB): part of set-up
    eng = engs[0]
    loop = StepLoop(engs, doc_ids, world, dev)
    run_steps, timing = loop.run_steps, loop.timing

    note("contexts staged")
    run_steps(args.warmup)
    for e in engs:
        e.set_profiling(2)                 # stage events on the engine streams, streams overlapped as in production
    timing["on"] = True
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    timing["on"] = False
    note("timed steps done")
    for e in engs:
        e.set_profiling(0)
    # ---- everything below is outside the timed region
    st, jl, vl, pe = eng.result_meta()
    assert int((st != 0).sum()) == 0, "documents failed on the device path"
    stats = eng.stats()
    # per-kernel durations of the timed steps came from HIP events on the engine streams (timing["k"]); two extra passes
    # with the streams run one after the other give every stage's own duration for the breakdown
    eng.set_profiling(True)
    # (a context splits the batch over lm_n_streams HIP streams: every stage is launched once per stream, on that
    # stream's share of the documents.  In the timed steps the streams overlap; in these two profiled passes they run
    # one after the other, so a kernel's duration is its own — durations are per launch, as rocprofv3 reports them)
    ktimes = {}
    for _ in range(2):
        eng.run()
        for name, ms in eng.kernel_times():
            ktimes.setdefault(name, []).append(ms)
    eng.set_profiling(False)
    n_streams = eng.b.n_streams(eng.h)
    kalone = {k: sum(v) / len(v) for k, v in ktimes.items()}           # one launch, nothing beside it
    kavg = {k: sum(v) / len(v) for k, v in timing["k"].items()}        # one launch, averaged over the TIMED steps
    # The dominant kernel is the one with the largest SERIALIZED duration x launches per step (every stage is launched once per
    # stream, so the launch counts are equal and the serialized per-launch duration decides).  The overlapped HIP-event averages of
    # the timed steps are begin-to-end times of kernels sharing the GPU: a latency-bound stage that starts beside the integrate
    # kernel shows a long begin-to-end time there without being the stage that costs the step (the driver's
    # line once named k_block_decode this way).  They stay in the line under their own keys.
    dom = max(kalone, key=kalone.get) if kalone else max(kavg, key=kavg.get)
    dom_overlapped = max(kavg, key=kavg.get) if kavg else dom
    alg_bytes = float(stats.in_bytes + stats.out_bytes)  # Σ blob bytes in + JSON + VV bytes out
    alg_per_launch = alg_bytes / n_streams                # one launch of the dominant kernel covers 1/n_streams of the batch
    # The roofline is priced on the dominant kernel's OWN duration: one launch with nothing beside it (the two serialized passes
    # above; rocprofv3 reports the same figure for those dispatches — profiles/rNN_kernel_stats.md, "serialized passes").  In the
    # timed steps up to inflight x streams launches of the same kernel share the GPU, so a launch's begin-to-end time there is
    # inflated by its neighbours and (x launches per step) exceeds the step: it is reported separately, not used for `frac`.
    k_ms = kalone.get(dom) or kavg[dom]
    achieved = alg_per_launch / (k_ms * 1e-3) / 1e9
    # HBM traffic of the dominant kernel cannot be read inside this process: it comes from separate rocprofv3 --pmc passes
    # of this same command (profiles/collect.sh), committed per round.  Only a record counted on the kernel that ran here is
    # quoted; otherwise traffic is null
    traffic, traffic_src, prof_alone, prof_kernel = None, None, None, None
    for tag in sorted({f.split("_pmc_integrate.json")[0] for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_integrate.json")}, reverse=True):
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_integrate.json")))
        except Exception:
            continue
        if prof_kernel is None:
            prof_kernel = (f"profiles/{tag}_pmc_integrate.json", pj.get("dominant_kernel"))   # the newest committed record
        if pj.get("dominant_kernel") != dom:
            continue
        traffic = pj.get("hbm_bytes_per_launch")
        prof_alone = pj.get("kernels", {}).get(dom, {}).get("alone_avg_ms")
        traffic_src = (f"profiles/{tag}_pmc_integrate.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of this command (not this run), "
                       f"counted on {dom}; (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, the gfx950 FETCH correction of the microarchitecture guide")
        break

    if prof_kernel is not None and prof_kernel[1] != dom:
        # a flip of the dominant kernel against the committed profile is loud: stderr always, fatal under LM_BENCH_STRICT=1
        # (profiles/collect.sh sets it, so a stale record cannot be committed beside a line that names another kernel)
        note(f"WARNING: the dominant kernel here is {dom}, the newest committed PMC record ({prof_kernel[0]}) is of {prof_kernel[1]}: roofline.traffic is null")
        assert os.environ.get("LM_BENCH_STRICT", "0") in ("", "0"), f"dominant kernel {dom} != profiled kernel {prof_kernel[1]}"
    line = None
    if rank == 0:
        # parity spot check of what was just timed (oracle = checker only)
        got = eng.fetch()
        for e in engs[1:]:
            assert e.fetch() == got, "contexts disagree"
        parity_note = None
        if world == 1:
            import _oracle
            # EVERY document of the timed batch against the oracle (every document differs: B's letters are stamped per document),
            # 2,048 at a time so that the checker's copies stay small (the line used to check the first 2,048)
            n_chk = len(docs) if os.environ.get("LM_BENCH_PARITY", "all") == "all" else min(2048, len(docs))
            for c0 in range(0, n_chk, 2048):
                want = _oracle.merge_batch(docs[c0:min(n_chk, c0 + 2048)], threads=min(32, os.cpu_count() or 1))
                assert got[c0:c0 + len(want)] == want, f"device results differ from the CPU oracle (documents {c0}..{c0 + len(want)})"
            parity_note = f"all {n_chk} documents of the timed batch equal to the oracle's (JSON, version vector, status, pending)"
        n_total = args.docs * world
        line = {
            "metric": "merged docs/sec (batch of N docs x M remote ops)",
            "value": round(n_total * args.steps / dt, 1),
            "unit": "docs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": f"configs[1]: {args.docs} docs/GPU x {args.base_ops + 2 * args.branch_ops}-op automerge-paper-shaped text "
                            f"trace, 2 concurrent peers, 3 FastUpdates blobs/doc ({tpl.n_runs} op runs, {tpl.n_changes} changes, "
                            f"{sum(len(b) for b in docs[0])} blob bytes/doc)",
                "docs_per_gpu": args.docs, "ops_per_doc": args.base_ops + 2 * args.branch_ops,
                "sharding": f"doc-sharded x{world}, one all-gather of per-doc summaries per step" if world > 1 else "single GPU",
                "contexts_in_flight": len(engs), "streams_per_context": n_streams,
            },
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": int(alg_per_launch), "launches_per_step": n_streams,
                "kernel_ms": round(k_ms, 3),
                "kernel_ms_source": "HIP events on the engine stream, streams serialized (one launch, nothing beside it), after the timed steps",
                "kernel_ms_rocprofv3": prof_alone,
                "dominant_by": "largest serialized per-launch duration (kernels_ms_per_launch_alone); every stage is launched launches_per_step times per step",
                "dominant_kernel_of_the_committed_profile": None if prof_kernel is None else {"file": prof_kernel[0], "kernel": prof_kernel[1], "same_kernel": prof_kernel[1] == dom},
                "largest_overlapped_event_average": {"kernel": dom_overlapped, "ms": round(kavg.get(dom_overlapped, 0.0), 3),
                                                     "note": "begin-to-end time while sharing the GPU with the other context's kernels; not used for frac"},
                "kernel_ms_in_timed_region_overlapped": round(kavg.get(dom, k_ms), 3),
                "frac_in_timed_region_overlapped": round(alg_per_launch / (kavg.get(dom, k_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "pipeline_achieved": round(alg_bytes / (dt / args.steps) / 1e9, 2),
                "pipeline_frac": round(alg_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
            },
            "parity": parity_note if world == 1 else "checked at N=1 (the oracle runs on rank 0's host cores)",
            "kernels_ms_per_launch": {k: round(v, 3) for k, v in kavg.items()},
            "kernels_ms_per_launch_alone": {k: round(v, 3) for k, v in kalone.items()},
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
            line["gpu_over_cpu"] = {"measured_on_available_cores": round(line["value"] / cpu["value"], 2),
                                    "vs_linear_extrapolation_to_physical_cores": round(line["value"] / cpu["linear_extrapolation_to_physical_cores"]["value"], 2)}
        table = StepLoop.host_table(out)
        assert table.shape[0] == n_total and (table[:, 1] == 0).all() and (table[:, 5] != 0).all()
        if loop.dev_rows:
            assert loop.device_rows_check(0), "summary rows written on the device differ from the host-side summary"
            line["config"]["summary_exchange"] = "rows written by k_summary_rows into the send buffer; one all_gather_into_tensor of equal counts per step, result left on the device"
        import xxhash
        hashes = np.ascontiguousarray(table[:, 5]).view(np.uint64)
        for i in (0, len(got) // 2, len(got) - 1):   # the summary's content word is the hash of what lm_fetch returns
            assert int(hashes[i]) == xxhash.xxh64(got[i][1]).intdigest(), "device xxh64 differs from the fetched JSON's"
        note("parity + summary checked")
        if world == 1 and not args.no_end_to_end:
            try:   # (a leg outside the metric: its failure is reported in its own entry and leaves the line alone)
                line["end_to_end"] = end_to_end(engs, docs, max(16, args.steps))   # (fill and drain of the four-deep pipeline are inside the timed region: enough steps to see the steady state)
            except Exception as ex:
                line["end_to_end"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
            note("end-to-end done")
    for e in engs:
        e.close()
    if rank == 0 and world == 1 and not args.no_other_configs:
        line["other_configs"] = other_configs(local_rank, host_cores()[0])
        note("other configs done")
        line["other_configs"].update(resident_configs(local_rank, host_cores()[0], n_docs=args.docs))
        note("resident configs done")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
