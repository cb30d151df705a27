"""Resident documents (lm_import): a document's history delivered in several steps, each step rendered — shared by the
kernel-logic (CPU) and the GPU parity tests.  The checker is the oracle's Session (oracle/lo_capi.cpp)."""
import random
from loro_amd import wire
import _oracle


def chunked_blobs(reps, rng, max_chunk=4):
    """every replica's OWN changes as several update blobs of 1..max_chunk consecutive changes each"""
    out = []
    for r in reps:
        own = r.changes.get(r.peer, [])
        i = 0
        while i < len(own):
            k = rng.randint(1, max_chunk)
            out.append(wire.encode_updates(wire.split_blocks(own[i:i + k])))
            i += k
    return out


def plan_steps(blobs, rng, n_steps, versions=(), shuffle=0.3, checkout_prob=0.35):
    """[(new blobs, frontiers or None)] — the blobs in a mostly causal order (a fraction is shuffled: those arrive before
    their dependencies and wait as pending changes), some steps rendered at one of `versions` (which may not be imported yet)"""
    order = list(blobs)
    for _ in range(int(len(order) * shuffle)):
        a, b = rng.randrange(len(order)), rng.randrange(len(order))
        order[a], order[b] = order[b], order[a]
    cuts = sorted(rng.randint(0, len(order)) for _ in range(n_steps - 1))
    steps, lo = [], 0
    for c in cuts + [len(order)]:
        f = None
        if versions and rng.random() < checkout_prob:
            f = wire.encode_frontiers(rng.choice(versions))
        steps.append((order[lo:c], f))
        lo = c
    return steps


def run_sessions(ctx, sessions):
    """sessions: list (one per document) of step lists [(blobs, frontiers)], all of the same length.  Drives `ctx` (a
    loro_amd._cabi.Context) with lm_stage for the first step and lm_import for the others; returns per step the list of results."""
    n_steps = len(sessions[0])
    assert all(len(s) == n_steps for s in sessions)
    got = []
    for k in range(n_steps):
        docs = [s[k][0] for s in sessions]
        fr = [s[k][1] for s in sessions]
        if k == 0:
            ctx.stage(docs, fr)
            ctx.import_more([[] for _ in docs], fr)   # resident from the first run on
        else:
            ctx.import_more(docs, fr)
        ctx.run()
        got.append(ctx.fetch())
    return got


def oracle_sessions(sessions):
    out = [[] for _ in sessions[0]]
    for s in sessions:
        o = _oracle.Session()
        for k, (blobs, f) in enumerate(s):
            out[k].append(o.step(blobs, f))
        o.close()
    return out
