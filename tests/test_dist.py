"""N>1 path on CPU: document sharding + the single all-gather of per-document summaries, world_size 2, gloo.
The merge itself is stood in by the oracle here (no GPU in this container); what is under test is
loro_amd.dist — ownership, padding of uneven shards, ordering of the gathered table."""
import os, socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _oracle, _cases
from loro_amd import dist as lmdist


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_docs, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    docs = _cases.fuzz_docs(n_docs)
    mine = lmdist.owned_docs(n_docs, rank, world)
    res = _oracle.merge_batch([docs[d] for d in mine])
    table = lmdist.all_gather_summaries(lmdist.summarize(mine, res))
    np.save(os.path.join(out_dir, f"table{rank}.npy"), table)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_docs", [7, 10])
def test_sharded_summary_all_gather(tmp_path, n_docs):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_docs, str(tmp_path)), nprocs=world, join=True)
    docs = _cases.fuzz_docs(n_docs)
    ref = lmdist.summarize(list(range(n_docs)), _oracle.merge_batch(docs))
    for r in range(world):
        t = np.load(os.path.join(str(tmp_path), f"table{r}.npy"))
        assert t.shape == ref.shape and (t == ref).all(), f"rank {r} holds a different merged-state table"


def test_ownership_partitions_documents():
    for n, w in ((10, 2), (7, 4), (3, 8)):
        owned = [lmdist.owned_docs(n, r, w) for r in range(w)]
        assert sorted(d for o in owned for d in o) == list(range(n))


def _bench_worker(rank, world, port, per_rank, out_dir):
    """bench.py's serving loop (StepLoop: contexts in flight, device-side summary incl. the xxh64 of the JSON, the single
    all-gather) with the kernel-logic harness standing in for the GPU and gloo for RCCL."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench, _emu
    from loro_amd._cabi import Context
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    docs = _cases.fuzz_docs(per_rank * world)[rank * per_rank:(rank + 1) * per_rank]
    doc_ids = list(range(rank * per_rank, (rank + 1) * per_rank))
    engs = [Context(_emu.binding()) for _ in range(2)]
    for e in engs:
        e.stage(docs)
        e.run()
    loop = bench.StepLoop(engs, doc_ids, world, None)
    loop.run_steps(1)
    table = bench.StepLoop.host_table(loop.run_steps(3))
    assert loop.dev_rows and loop.device_rows_check(0) and loop.device_rows_check(1)   # the one-collective path ran, on rows a kernel wrote
    np.save(os.path.join(out_dir, f"bench_table{rank}.npy"), table)
    for e in engs:
        e.close()
    dist.barrier()
    dist.destroy_process_group()


def test_bench_step_loop_two_ranks(tmp_path):
    world, per_rank = 2, 5
    _emu_binding_is_built()
    mp.spawn(_bench_worker, args=(world, _free_port(), per_rank, str(tmp_path)), nprocs=world, join=True)
    docs = _cases.fuzz_docs(per_rank * world)
    ref = lmdist.summarize(list(range(per_rank * world)), _oracle.merge_batch(docs))   # host-side xxh64 of the oracle's JSON
    for r in range(world):
        t = np.load(os.path.join(str(tmp_path), f"bench_table{r}.npy"))
        assert t.shape == ref.shape and (t == ref).all(), f"rank {r}: gathered summary differs (device-side hash included)"


def _emu_binding_is_built():
    import _emu
    _emu.binding()   # compile once in the parent, not concurrently in both ranks


def test_c_abi_summary_matches_the_python_exchange():
    """lm_summary_allgather (the exchange step in the C ABI, for hosts without torch): with one rank it returns this context's
    table — the same six words per document loro_amd/dist.py builds and all-gathers"""
    import numpy as np
    import _emu, _cases, _oracle
    from loro_amd._cabi import Context
    from loro_amd import dist as lmdist
    names, docs = _cases.edge_case_docs()
    with Context(_emu.binding()) as c:
        c.stage(docs); c.run()
        st, jl, vl, pe = c.result_meta()
        ids = [1000 + 3 * i for i in range(len(docs))][::-1]          # any global numbering: the table comes back sorted by id
        want = lmdist.summarize_device(ids, st, pe, jl, vl, c.result_hashes())
        want = want[np.argsort(want[:, 0], kind="stable")]
        c.comm_init(0, 1)
        got = c.summary_allgather(ids, len(docs))
        assert got.shape == want.shape and (got == want).all()
        assert (got == lmdist.summarize(sorted(ids), [c.fetch()[len(docs) - 1 - k] for k in range(len(docs))])).all()


def test_summary_rows_written_on_the_device_match_the_host_summary():
    """lm_summary_layout: every run writes the context's summary rows with a kernel (k_summary_rows) — ids first + i * stride,
    padding rows -1, statuses as lm_result_meta reports them (failed documents, out-of-scope containers) — and
    lm_summary_allgather_device with one rank hands that table back without a copy"""
    import ctypes
    import _emu
    from loro_amd._cabi import Context
    names, docs = _cases.edge_case_docs()
    docs = docs + _cases.corrupted_docs(12, seed=3)
    with Context(_emu.binding()) as c:
        c.stage(docs)
        c.summary_layout(500, 3, len(docs) + 5)
        c.run()
        st, jl, vl, pe = c.result_meta()
        want = lmdist.summarize_device([500 + 3 * i for i in range(len(docs))], st, pe, jl, vl, c.result_hashes())
        ptr, rows = c.summary_rows_ptr()
        t = lmdist.rows_tensor(ptr, rows).numpy()
        assert rows == len(docs) + 5 and (t[len(docs):] == -1).all() and (t[:len(docs)] == want).all()
        assert (lmdist.table_of(lmdist.rows_tensor(ptr, rows)) == want).all()
        c.comm_init(0, 1)
        out = ctypes.c_void_p()
        assert c.b.summary_allgather_device(c.h, ctypes.byref(out)) == rows and out.value == ptr
        c.run()   # the rows are rewritten by every run
        assert (lmdist.rows_tensor(ptr, rows).numpy()[:len(docs)] == want).all()


def test_summary_rows_of_folded_batches_and_of_replayed_documents():
    """round 6: lm_summary_layout works on a batch whose entries were folded (shared blobs / checked-out entries: the
    rows are written from the host after the run) and the rows of documents the side engine replayed (DF_REDO) are the replay's"""
    import _emu, _oracle
    from loro_amd._cabi import Context
    bad, _good = _cases.misnamed_delete_docs(6)
    empty = b"\x00"
    docs, fr = [], []
    for d in bad:
        docs += [d, d]; fr += [None, empty]
    want = _oracle.merge_batch(docs, frontiers=fr)
    for env in ({"LM_SPAN_AUTO": "1", "LM_SHARE_REPLAY": "0"}, {"LM_SPAN_AUTO": "0"}):
        os.environ.update(env)
        try:
            with Context(_emu.binding()) as c:
                c.stage(docs, fr)
                c.summary_layout(7, 2, len(docs) + 3)
                c.run()
                assert c.fetch() == want
                st, jl, vl, pe = c.result_meta()
                exp = lmdist.summarize_device([7 + 2 * i for i in range(len(docs))], st, pe, jl, vl, c.result_hashes())
                ptr, rows = c.summary_rows_ptr()
                t = lmdist.rows_tensor(ptr, rows).numpy()
                assert (t[:len(docs)] == exp).all() and (t[len(docs):] == -1).all(), env
                assert list(st) == [w[0] for w in want]
        finally:
            for k in env:
                del os.environ[k]


def _stub_rccl():
    """tests/emu/rccl_stub.c → a shared library with the four RCCL entry points the product resolves by dlsym, on host pointers"""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    src, so = os.path.join(here, "emu", "rccl_stub.c"), os.path.join(here, "emu", "librccl_stub.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        tmp = f"{so}.{os.getpid()}.tmp"
        subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", "-o", tmp, src])
        os.replace(tmp, so)
    return so


def _cabi_worker(rank, world, id_hex, per_rank, stub, out_dir):
    """the C ABI's own exchange (lm_comm_init + lm_summary_layout + lm_summary_allgather_device, and the host-table form
    lm_summary_allgather) between two processes: the kernel-logic harness for the GPU, the stub for librccl.so (LM_RCCL_LIB)"""
    import ctypes, sys
    os.environ["LM_RCCL_LIB"] = stub
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import _emu
    from loro_amd._cabi import Context
    total = per_rank * world + 1                      # uneven shards: rank 0 holds one document more
    docs = _cases.fuzz_docs(total)
    mine = lmdist.owned_docs(total, rank, world)      # documents dealt doc % world
    rows_padded = (total + world - 1) // world
    with Context(_emu.binding()) as c:
        c.comm_init(rank, world, bytes.fromhex(id_hex))
        c.stage([docs[d] for d in mine])
        c.summary_layout(rank, world, rows_padded)
        c.run()
        tp = ctypes.c_void_p()
        n = c.b.summary_allgather_device(c.h, ctypes.byref(tp))
        assert n == world * rows_padded, c.b.last_error(c.h)
        dev = np.ctypeslib.as_array(ctypes.cast(tp.value, ctypes.POINTER(ctypes.c_int64)), shape=(n, 6)).copy()
        host = c.summary_allgather(mine, total)
    np.save(os.path.join(out_dir, f"dev{rank}.npy"), dev)
    np.save(os.path.join(out_dir, f"host{rank}.npy"), host)


def test_c_abi_exchange_between_two_processes_through_a_stub_rccl(tmp_path):
    """lm_comm_unique_id / lm_comm_init (dlopen of the collective library) / lm_summary_layout /
    lm_summary_allgather_device — the ONE collective on device memory — and lm_summary_allgather, world 2, end to end before the
    first run on an 8-GPU node.  Unmeasured on hardware: what is under test is the ABI's plumbing (ranks, padding of uneven shards,
    the gathered table's layout and content: document id, status, pending, lengths, xxh64 of the JSON computed by k_hash_json)."""
    import xxhash
    stub = _stub_rccl()
    os.environ["LM_RCCL_LIB"] = stub
    os.environ["LM_RCCL_STUB_DIR"] = str(tmp_path)
    try:
        import _emu
        from loro_amd._cabi import Context
        import ctypes
        buf = ctypes.create_string_buffer(128)
        assert _emu.binding().comm_unique_id(buf) == 0
        world, per_rank = 2, 5
        mp.spawn(_cabi_worker, args=(world, buf.raw.hex(), per_rank, stub, str(tmp_path)), nprocs=world, join=True)
    finally:
        del os.environ["LM_RCCL_LIB"]; del os.environ["LM_RCCL_STUB_DIR"]
    total = per_rank * world + 1
    docs = _cases.fuzz_docs(total)
    want = _oracle.merge_batch(docs)
    ref = np.array([[d, w[0], w[3], len(w[1]), len(w[2]), np.uint64(xxhash.xxh64(w[1]).intdigest() if w[0] in (0, 4) and w[1] else 0).astype(np.int64)]
                    for d, w in enumerate(want)], dtype=np.int64)
    rows_padded = (total + world - 1) // world
    for r in range(world):
        host = np.load(os.path.join(str(tmp_path), f"host{r}.npy"))
        assert host.shape == ref.shape and (host == ref).all(), f"rank {r}: lm_summary_allgather"
        dev = np.load(os.path.join(str(tmp_path), f"dev{r}.npy"))
        assert dev.shape == (world * rows_padded, 6)
        got = dev[dev[:, 0] >= 0]
        got = got[np.argsort(got[:, 0])]
        assert (got == ref).all(), f"rank {r}: lm_summary_allgather_device"
        assert (dev[rows_padded + (total // world):, 0] == -1).all()   # rank 1's padding row
