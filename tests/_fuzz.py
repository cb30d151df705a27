"""Random multi-peer editing sessions → update blobs (used by CPU and GPU parity tests)."""
import random
import _oracle
from loro_amd import wire

ALPHA = "abcdefghijklmnopqrstuvwxyz ABCDEFGH\n\"\\\t" + "äßλ中😀"


def random_session(seed, n_peers=3, n_steps=60, kinds=("text",), sync_prob=0.15, max_ins=6, commit_prob=0.4,
                   peer_base=None, styles=False, snapshots=None, solo_steps=0, max_del=4, solo_peer=0):
    """Returns (list of blobs in a random delivery order, replicas).  Replicas edit concurrently and sync
    pairwise; after a sync the receiver's visible sequences are refreshed from the oracle."""
    rng = random.Random(seed)
    base = peer_base if peer_base is not None else rng.randrange(1, 1 << 40)
    ids = []
    for i in range(n_peers):
        p = base + rng.randrange(0, 1000)
        while p in ids:
            p += 1
        ids.append(p)
    reps = [wire.Replica(p) for p in ids]

    def refresh(r):
        blob = r.export()
        for cid in list(r.seq.keys()) + [wire.root_cid("text", wire.KIND_TEXT), wire.root_cid("list", wire.KIND_LIST)]:
            if (cid.kind == wire.KIND_TEXT and "text" in kinds) or (cid.kind == wire.KIND_LIST and "list" in kinds):
                r.set_visible(cid.name, cid.kind, _oracle.visible_ids([blob], cid.name, cid.kind))

    # solo_steps: the session begins with that many steps of ONE replica (reps[solo_peer]) — a history that is a single chain — which
    # every other replica then imports: the version at the hand-over is a critical version (the linear prefix of the batch replay,
    # lm_k_integrate_linear.h); max_del: longest delete of a step (whole items, leaves emptied)
    for step in range(solo_steps + n_steps):
        if step == solo_steps and solo_steps and n_peers > 1:
            reps[solo_peer].commit()
            for o in reps:
                if o is not reps[solo_peer] and o.merge_from(reps[solo_peer]):
                    refresh(o)
        r = reps[solo_peer] if step < solo_steps else rng.choice(reps)
        kind = rng.choice(kinds)
        if kind == "text":
            ids = r.seq.setdefault(wire.root_cid("text", wire.KIND_TEXT), [])
            if ids and rng.random() < 0.35:
                pos = rng.randrange(len(ids))
                n = min(len(ids) - pos, rng.randint(1, max_del))
                r.text_delete("text", pos, n)
            elif styles and len(ids) >= 2 and rng.random() < 0.1:
                s = rng.randrange(len(ids) - 1)
                e = rng.randrange(s + 1, len(ids))
                if styles == "rich":   # several keys, values of several types, unmarks (null) — what lm_richtext resolves (lm_k_richtext.h)
                    r.text_mark("text", s, e, rng.choice(["bold", "link", "color", "a\"b"]),
                                rng.choice([True, True, None, None, "https://x.y/?q=\"1\"", 7, -2.5, "red", [1, "z"], {"k": 1}]))
                else:
                    r.text_mark("text", s, e, "bold", True)
            else:
                pos = rng.randint(0, len(ids))
                s = "".join(rng.choice(ALPHA) for _ in range(rng.randint(1, max_ins)))
                r.text_insert("text", pos, s)
        elif kind == "list":
            ids = r.seq.setdefault(wire.root_cid("list", wire.KIND_LIST), [])
            if ids and rng.random() < 0.3:
                pos = rng.randrange(len(ids))
                r.list_delete("list", pos, min(len(ids) - pos, rng.randint(1, max(3, max_del // 2))))
            else:
                vals = [rng.choice([None, True, False, rng.randint(-10**12, 10**12), "s%d" % rng.randint(0, 99), b"\x00\x01\xff",
                                    [1, "x", [None]], rng.uniform(-1e3, 1e3), rng.choice([0.5, 1e-9, 3.0e22, -0.0]),
                                    {"b": rng.randint(0, 9), "a": [1.25, {"d": None, "c": "x"}]}]) for _ in range(rng.randint(1, 3))]
                r.list_insert("list", rng.randint(0, len(ids)), vals)
        else:
            key = "k%d" % rng.randint(0, 7)
            if rng.random() < 0.2:
                r.map_delete("map", key)
            else:
                r.map_set("map", key, rng.choice([None, True, rng.randint(-5, 5), "v\"%d" % rng.randint(0, 9), [1, 2, "z"], rng.random() * 10 ** rng.randint(-8, 20),
                                                  {"k%d" % rng.randint(0, 3): rng.randint(0, 5), "z": {"y": 1, "x": [2.5]}, "": "e"}]))
        if rng.random() < commit_prob:
            r.commit()
            if snapshots is not None and r.frontiers and rng.random() < 0.5:
                # (version, updates holding exactly that version's causal history)
                snapshots.append((list(r.frontiers), r.export()))
        if rng.random() < sync_prob and n_peers > 1 and step >= solo_steps:
            a, b = rng.sample(reps, 2)
            a.commit(); b.commit()
            if a.merge_from(b):
                refresh(a)
    for r in reps:
        r.commit()
    return reps


def blobs_of(reps, rng=None, split=False):
    """Each replica exports only its OWN changes (as a relay would hold them); order optionally shuffled."""
    out = []
    for r in reps:
        own = wire.Replica(r.peer)
        own.changes = {r.peer: r.changes.get(r.peer, [])}
        if own.changes[r.peer]:
            out.append(own.export())
    if rng:
        rng.shuffle(out)
    return out


def nested_session(seed, n_peers=3, n_steps=120, sync_prob=0.1, max_depth=4):
    """Random concurrent session over NESTED containers: root Map "nm" and root List "nl" hold child Map / List / Text
    containers (created with insert_container), which hold further children; peers edit any container whose creating
    op they have seen, children get overwritten / deleted (unreachable afterwards), some children never receive an op."""
    rng = random.Random(seed)
    base = rng.randrange(1, 1 << 40)
    reps = [wire.Replica(base + 7 * i) for i in range(n_peers)]
    K = wire
    conts = [(wire.root_cid("nm", K.KIND_MAP), None, 0), (wire.root_cid("nl", K.KIND_LIST), None, 0)]   # (cid, creator id, depth)

    def usable(r):
        return [c for c in conts if c[1] is None or r.vv.get(c[1][0], 0) > c[1][1] or (c[1][0] == r.peer and c[1][1] < r.next_counter)]

    def refresh(r):
        blob = r.export()
        for cid, _, _ in usable(r):
            if cid.kind in (K.KIND_TEXT, K.KIND_LIST):
                r.set_visible(cid, cid.kind, _oracle.visible_ids([blob], cid, cid.kind))

    for _ in range(n_steps):
        r = rng.choice(reps)
        cid, _, depth = rng.choice(usable(r))
        roll = rng.random()
        if cid.kind == K.KIND_MAP:
            key = "k%d" % rng.randint(0, 5)
            if roll < 0.25 and depth < max_depth:
                ch = r.map_set_container(cid, key, rng.choice([K.KIND_MAP, K.KIND_LIST, K.KIND_TEXT]))
                conts.append((ch, (r.peer, ch.counter), depth + 1))
            elif roll < 0.4:
                r.map_delete(cid, key)
            else:
                r.map_set(cid, key, rng.choice([None, False, rng.randint(-99, 99), "s%d" % rng.randint(0, 9), [1, ["a"]]]))
        elif cid.kind == K.KIND_LIST:
            ids = r.seq.setdefault(cid, [])
            if roll < 0.25 and depth < max_depth:
                ch = r.list_insert_container(cid, rng.randint(0, len(ids)), rng.choice([K.KIND_MAP, K.KIND_LIST, K.KIND_TEXT]))
                conts.append((ch, (r.peer, ch.counter), depth + 1))
            elif roll < 0.45 and ids:
                pos = rng.randrange(len(ids))
                r.list_delete(cid, pos, min(len(ids) - pos, rng.randint(1, 2)))
            else:
                r.list_insert(cid, rng.randint(0, len(ids)), [rng.choice([True, rng.randint(0, 9), "v"]) for _ in range(rng.randint(1, 3))])
        else:
            ids = r.seq.setdefault(cid, [])
            if roll < 0.3 and ids:
                pos = rng.randrange(len(ids))
                r.text_delete(cid, pos, min(len(ids) - pos, rng.randint(1, 3)))
            else:
                r.text_insert(cid, rng.randint(0, len(ids)), "".join(rng.choice(ALPHA) for _ in range(rng.randint(1, 5))))
        if rng.random() < 0.4:
            r.commit()
        if rng.random() < sync_prob and n_peers > 1:
            a, b = rng.sample(reps, 2)
            a.commit(); b.commit()
            if a.merge_from(b):
                refresh(a)
    for r in reps:
        r.commit()
    return reps


def movable_session(seed, n_peers=3, n_steps=80, sync_prob=0.15, nested=False, snapshots=None, bulk=0):
    """Random concurrent session over a root MovableList "ml" (+ a Map "map", and with `nested` child containers created
    by insert_container / set_container and a child MovableList under the Map).  Peers see the list through the writer's
    local element view (wire.Replica.mlist_*), refreshed from the oracle's item order after a sync.  `bulk` > 0 starts
    with that many elements inserted by the first peer and synced to everyone (multi-leaf lists)."""
    rng = random.Random(seed)
    base = rng.randrange(1, 1 << 40)
    reps = [wire.Replica(base + 3 * i) for i in range(n_peers)]
    K = wire
    ML = K.KIND_MOVABLE
    lists = [(K.root_cid("ml", ML), None)]          # (cid, creating op id)
    kids = []                                        # child Text / Map / List containers: (cid, creating op id)

    def known(r, made):
        return made is None or r.vv.get(made[0], 0) > made[1] or (made[0] == r.peer and made[1] < r.next_counter)

    def refresh(r):
        blob = r.export()
        for cid, made in lists + kids:
            if known(r, made) and cid.kind in (K.KIND_TEXT, K.KIND_LIST, ML):
                r.set_visible(cid, cid.kind, _oracle.visible_ids([blob], cid, cid.kind))

    if bulk:
        for i in range(0, bulk, 7):
            reps[0].mlist_insert("ml", reps[0].mlist_len("ml"), ["b%d" % k for k in range(i, min(bulk, i + 7))])
        reps[0].commit()
        for r in reps[1:]:
            r.merge_from(reps[0])
            refresh(r)
    if nested:
        ch = reps[0].map_set_container("map", "child_ml", ML)
        lists.append((ch, (reps[0].peer, ch.counter)))
    for _ in range(n_steps):
        r = rng.choice(reps)
        roll = rng.random()
        if kids and roll < 0.12:
            cid, made = rng.choice(kids)
            if known(r, made):
                if cid.kind == K.KIND_TEXT:
                    ids = r.seq.setdefault(cid, [])
                    r.text_insert(cid, rng.randint(0, len(ids)), "".join(rng.choice(ALPHA) for _ in range(rng.randint(1, 4))))
                elif cid.kind == K.KIND_MAP:
                    r.map_set(cid, "k%d" % rng.randint(0, 3), rng.randint(0, 99))
                else:
                    ids = r.seq.setdefault(cid, [])
                    r.list_insert(cid, rng.randint(0, len(ids)), [rng.randint(0, 9)])
        else:
            cid, made = rng.choice(lists)
            if not known(r, made):
                continue
            n = r.mlist_len(cid)
            roll = rng.random()
            if n == 0 or roll < 0.3:
                r.mlist_insert(cid, rng.randint(0, n), [rng.choice([None, True, rng.randint(-99, 99), "s%d" % rng.randint(0, 9), [1, {"k": 2.5}], 0.25])
                                                        for _ in range(rng.randint(1, 3))])
            elif roll < 0.55 and n >= 2:
                r.mlist_move(cid, rng.randrange(n), rng.randrange(n))
            elif roll < 0.72:
                r.mlist_set(cid, rng.randrange(n), rng.choice([False, rng.randint(0, 9), "t%d" % rng.randint(0, 9), {"m": [1, 2]}]))
            elif roll < 0.86:
                p = rng.randrange(n)
                r.mlist_delete(cid, p, min(n - p, rng.randint(1, 2)))
            elif nested and roll < 0.93:
                kind = rng.choice([K.KIND_TEXT, K.KIND_MAP, K.KIND_LIST])
                ch = r.mlist_insert_container(cid, rng.randint(0, n), kind) if rng.random() < 0.5 else r.mlist_set_container(cid, rng.randrange(n), kind)
                kids.append((ch, (r.peer, ch.counter)))
            else:
                r.map_set("map", "k%d" % rng.randint(0, 3), rng.randint(0, 9))
        if rng.random() < 0.4:
            r.commit()
            if snapshots is not None and r.frontiers and rng.random() < 0.5:
                snapshots.append((list(r.frontiers), r.export()))
        if rng.random() < sync_prob and n_peers > 1:
            a, b = rng.sample(reps, 2)
            a.commit(); b.commit()
            if a.merge_from(b):
                refresh(a)
    for r in reps:
        r.commit()
    return reps
