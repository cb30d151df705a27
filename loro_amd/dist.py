"""Multi-GPU sharding of a document batch: documents are independent (no cross-document state in
LoroDocInner, crates/loro-internal/src/lib.rs:142-172), so rank r owns documents d with d % world == r
and runs the whole per-document pipeline locally.  The only exchange is ONE all-gather (RCCL over xGMI on
ROCm; gloo in the CPU tests) of a fixed-size per-document summary so that every rank holds the merged-state
table; the JSON itself stays with the owning rank."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

SUMMARY_WORDS = 6  # doc index, status, pending ops, json length, vv length, xxh64(json)


def owned_docs(n_docs: int, rank: int, world: int) -> List[int]:
    return list(range(rank, n_docs, world))


def summarize_device(doc_ids: Sequence[int], status, pending, json_len, vv_len, json_xxh64) -> np.ndarray:
    """The per-document summary table from what the engine returns WITHOUT fetching the JSON: lm_result_meta +
    lm_result_hashes (the hash is computed on the device, k_hash_json)."""
    out = np.zeros((len(doc_ids), SUMMARY_WORDS), dtype=np.int64)
    out[:, 0] = np.asarray(doc_ids, dtype=np.int64)
    out[:, 1] = np.asarray(status, dtype=np.int64)
    out[:, 2] = np.asarray(pending, dtype=np.uint64).astype(np.int64)
    out[:, 3] = np.asarray(json_len, dtype=np.uint64).astype(np.int64)
    out[:, 4] = np.asarray(vv_len, dtype=np.uint64).astype(np.int64)
    out[:, 5] = np.asarray(json_xxh64, dtype=np.uint64).view(np.int64)
    return out


def summarize(doc_ids: Sequence[int], results: Sequence[Tuple[int, bytes, bytes, int]]) -> np.ndarray:
    """Same table from fetched results (host-side xxh64): what summarize_device must agree with."""
    import xxhash
    hashes = np.array([xxhash.xxh64(js).intdigest() if st in (0, 4) and js else 0 for st, js, _, _ in results], dtype=np.uint64)
    return summarize_device(doc_ids, [r[0] for r in results], [r[3] for r in results], [len(r[1]) for r in results],
                            [len(r[2]) for r in results], hashes)


def all_gather_summaries(local: np.ndarray, device=None):
    """One collective: every rank contributes its [n_local, SUMMARY_WORDS] table (padded to the largest
    shard) and receives the table of all documents, ordered by document index."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)            # tiny: shard sizes (part of the same exchange step)
    n_max = int(max(int(s.item()) for s in sizes))
    buf = torch.full((n_max, SUMMARY_WORDS), -1, dtype=torch.int64, device=device)
    if local.shape[0]:
        buf[: local.shape[0]] = torch.from_numpy(local).to(buf.device)
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf)             # THE all-gather of the merged-state summary
    table = torch.cat([g[: int(s.item())] for g, s in zip(gathered, sizes)], dim=0).cpu().numpy()
    return table[np.argsort(table[:, 0], kind="stable")]


# ---- the exchange as ONE collective on device memory (lm_summary_layout: the rows are written by a kernel of every run)

def rows_padded(n_total: int, world: int) -> int:
    """rows every rank contributes when documents are dealt `doc % world`: the largest shard — computed, never exchanged"""
    return (n_total + world - 1) // world


class _DevRows:
    """a device buffer of the library seen through __cuda_array_interface__ (no copy): rows x SUMMARY_WORDS int64"""

    def __init__(self, ptr, rows):
        self.__cuda_array_interface__ = {"shape": (rows, SUMMARY_WORDS), "typestr": "<i8", "data": (int(ptr), False), "version": 2}   # (torch refuses read-only views; nobody writes through this one)


def rows_tensor(ptr: int, rows: int, device=None):
    """the library's summary rows as a tensor that aliases them: a device tensor on a GPU, a CPU tensor under the kernel-logic
    harness (whose "device" memory is host memory)"""
    import ctypes
    import torch
    if device is not None and getattr(device, "type", str(device)).startswith("cuda"):
        return torch.as_tensor(_DevRows(ptr, rows), device=device)
    buf = (ctypes.c_int64 * (rows * SUMMARY_WORDS)).from_address(ptr)
    return torch.from_numpy(np.frombuffer(buf, dtype=np.int64).reshape(rows, SUMMARY_WORDS))


def all_gather_rows(send, out=None):
    """THE all-gather of the merged-state summary: one collective, equal counts, device memory in and out.  `send` = this rank's
    [rows_padded, SUMMARY_WORDS] rows; returns the gathered [world * rows_padded, SUMMARY_WORDS] tensor (rank-major; rows of -1
    are padding), which stays where it is until table_of() is asked for the host table."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    if out is None:
        out = torch.empty((world * send.shape[0], SUMMARY_WORDS), dtype=torch.int64, device=send.device)
    try:
        dist.all_gather_into_tensor(out, send)
    except (RuntimeError, NotImplementedError, AttributeError):   # a backend without the flat form
        dist.all_gather(list(out.view(world, send.shape[0], SUMMARY_WORDS).unbind(0)), send)
    return out


def table_of(gathered) -> np.ndarray:
    """host table of a gathered tensor: padding dropped, ordered by document id"""
    t = gathered.cpu().numpy()
    t = t[t[:, 0] >= 0]
    return t[np.argsort(t[:, 0], kind="stable")]
