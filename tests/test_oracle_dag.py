"""The oracle's LCA / DiffMode restatement (oracle/lo_dag.hpp) against the reference's own known answers
(crates/loro-internal/src/dag.rs:1108-1290) and its brute-force property (dag.rs:955-1009, 1292-1340) on random DAGs."""
import random
import _oracle
from loro_amd import wire


def node(peer, counter, length, lamport, deps):
    return (peer, counter, length, lamport, list(deps))


def last(n):
    return (n[0], n[1] + n[2] - 1)


def first(n):
    return (n[0], n[1])


def lca(nodes, left, right):
    return _oracle.dag_lca(nodes, left, right)


def test_empty_linear_same_span_and_parent_child_cases():   # dag.rs:1108-1134
    a = node(1, 0, 2, 0, [])
    b = node(1, 2, 2, 2, [(1, 1)])
    d = [a, b]
    assert lca(d, [], [(1, 3)]) == ([], "Linear")
    assert lca(d, [(1, 3)], []) == ([], "Checkout")
    assert lca(d, [(1, 0)], [(1, 1)]) == ([(1, 0)], "Linear")
    assert lca(d, [(1, 1)], [(1, 0)]) == ([(1, 0)], "Checkout")
    assert lca(d, [(1, 1)], [(1, 3)]) == ([(1, 1)], "Linear")


def test_left_empty_stops_at_missing_shallow_dependency():   # dag.rs:1136-1145
    assert lca([node(1, 1, 1, 1, [(1, 0)])], [], [(1, 1)]) == ([], "ImportGreaterUpdates")


def test_parallel_branches_share_the_dependency():   # dag.rs:1147-1164
    root = node(1, 0, 1, 0, [])
    left = node(2, 0, 1, 1, [first(root)])
    right = node(3, 0, 1, 2, [first(root)])
    merge = node(4, 0, 1, 3, [first(left), first(right)])
    d = [root, left, right, merge]
    assert lca(d, [first(left)], [first(right)]) == ([first(root)], "Checkout")
    assert lca(d, [first(root)], [first(merge)])[0] == [first(root)]


def test_falls_back_before_independent_branch():   # dag.rs:1166-1180
    left = node(1, 0, 1, 0, [])
    ind = node(2, 0, 1, 1, [])
    merge = node(3, 0, 1, 2, [first(left), first(ind)])
    assert lca([left, ind, merge], [first(left)], [first(merge)]) == ([], "Checkout")


def test_falls_back_before_unmatched_branch_with_multiple_left_frontiers():   # dag.rs:1182-1204
    a = node(1, 0, 1, 0, [])
    b = node(2, 0, 1, 1, [])
    ind = node(3, 0, 1, 2, [])
    merge = node(4, 0, 1, 3, [first(a), first(b), first(ind)])
    assert lca([a, b, ind, merge], [first(a), first(b)], [first(merge)]) == ([], "Checkout")


def test_cross_peer_direct_dependency_is_a_greater_update():   # dag.rs:1206-1215
    left = node(1, 0, 1, 0, [])
    right = node(2, 0, 1, 1, [first(left)])
    assert lca([left, right], [first(left)], [first(right)]) == ([first(left)], "ImportGreaterUpdates")


def test_falls_back_when_right_adds_concurrent_branch_from_shared_root():   # dag.rs:1217-1232
    root = node(1, 0, 1, 0, [])
    left = node(2, 0, 1, 1, [first(root)])
    conc = node(3, 0, 1, 2, [first(root)])
    merge = node(4, 0, 1, 3, [first(left), first(conc)])
    assert lca([root, left, conc, merge], [first(left)], [first(merge)]) == ([], "Checkout")


def test_keeps_target_when_checking_out_to_ancestor_with_extra_branch():   # dag.rs:1234-1256
    root = node(1, 0, 1, 0, [])
    left = node(2, 0, 2, 1, [first(root)])
    right = node(3, 0, 2, 3, [first(root)])
    extra = node(4, 0, 1, 5, [last(left), last(right)])
    target = [(2, 0), (3, 0)]
    current = [first(extra), last(left), last(right)]
    assert lca([root, left, right, extra], current, target) == (sorted(target), "Checkout")


def test_does_not_keep_ancestor_of_shared_descendant():   # dag.rs:1258-1276
    root = node(1, 0, 1, 0, [])
    shared = node(2, 0, 1, 1, [first(root)])
    lo = node(3, 0, 1, 2, [first(root)])
    ro = node(4, 0, 1, 3, [first(root)])
    assert lca([root, shared, lo, ro], [first(shared), first(lo)], [first(shared), first(ro)]) == ([first(shared)], "Checkout")


# ---- the brute-force property (dag.rs:987-1009, assert_common_ancestor_valid_against_oracle)
def _lamport_of(nodes, i):
    for n in nodes:
        if n[0] == i[0] and n[1] <= i[1] < n[1] + n[2]:
            return n[3] + (i[1] - n[1])
    raise KeyError(i)


def _maximal(nodes, ids):
    ids = sorted(set(ids), key=lambda i: (_lamport_of(nodes, i), i))
    fr = []
    for i in reversed(ids):
        if any(i in _oracle.dag_ancestors(nodes, [f]) for f in fr):
            continue
        fr = [f for f in fr if f not in _oracle.dag_ancestors(nodes, [i])]
        fr.append(i)
    return sorted(fr)


def _random_dag(seed, count):   # the shape of dag.rs:1052-1080 (another RNG: the property is what is pinned)
    rng = random.Random(seed)
    nxt = [0] * 8
    nodes, ids = [], []
    for i in range(count):
        peer = rng.randrange(1, 8)
        ln = rng.randint(1, 3)
        k = rng.randint(0, min(3, len(ids))) if ids else 0
        deps = _maximal(nodes, [rng.choice(ids) for _ in range(k)]) if k else []
        n = node(peer, nxt[peer], ln, i * 4, deps)
        if nxt[peer] > 0 and (peer, nxt[peer] - 1) not in _oracle.dag_ancestors(nodes + [n], [first(n)]) - {first(n)}:
            n = node(peer, nxt[peer], ln, i * 4, _maximal(nodes, list(deps) + [(peer, nxt[peer] - 1)]))   # a peer's ops are causally ordered
        nxt[peer] += ln
        ids += [(peer, n[1] + o) for o in range(ln)]
        nodes.append(n)
    return nodes, ids


def _check(nodes, left, right):
    actual, mode = lca(nodes, left, right)
    la = _oracle.dag_ancestors(nodes, left)
    ra = _oracle.dag_ancestors(nodes, right)
    common = la & ra
    for i in actual:
        assert i in common, (left, right, actual, mode)
    for a in actual:
        for b in actual:
            if a != b:
                assert a not in _oracle.dag_ancestors(nodes, [b]), (left, right, actual, mode)
    if mode != "Checkout":
        expected = _maximal(nodes, common)
        assert actual == expected == sorted(left), (left, right, actual, expected, mode)
        assert all(i in ra for i in left)


def test_valid_against_brute_force_on_random_dags():
    for seed in range(128):
        rng = random.Random(1000 + seed)
        nodes, ids = _random_dag(seed, rng.randint(1, 18))
        for _ in range(48):
            fr = []
            for _side in range(2):
                if rng.random() < 0.1:
                    fr.append([])
                else:
                    fr.append(_maximal(nodes, [rng.choice(ids) for _ in range(rng.randint(1, min(4, len(ids))))]))
            _check(nodes, fr[0], fr[1])


def test_valid_against_brute_force_on_layered_merge_dag():   # dag.rs:1082-1105, 1300-1318
    root = node(1, 0, 3, 0, [])
    left = node(1, 3, 2, 4, [(1, 2)])
    right = node(2, 0, 3, 5, [(1, 1)])
    late_right = node(2, 3, 2, 9, [(2, 2)])
    third = node(3, 0, 2, 6, [(1, 2)])
    mlr = node(4, 0, 1, 12, [last(left), last(right)])
    mall = node(5, 0, 2, 16, [last(mlr), last(late_right), last(third)])
    ind = node(6, 0, 2, 20, [])
    fin = node(7, 0, 1, 25, [last(mall), last(ind)])
    nodes = [root, left, right, late_right, third, mlr, mall, ind, fin]
    ids = [(n[0], n[1] + o) for n in nodes for o in range(n[2])]
    frs = [[], [last(fin)]] + [[i] for i in ids] + [[(1, 4), (2, 2)], [(1, 4), (3, 1)], [(4, 0), (2, 4), (3, 1)], [(5, 1), (6, 1)]]
    for l in frs:
        for r in frs:
            _check(nodes, sorted(l), sorted(r))


def test_import_modes_of_real_blobs():
    """DiffMode per LoroDoc::import on blobs from the workload writer: a single linear history is Linear, a second peer's
    concurrent history is an Import (Checkout promoted because the version only grows, oplog.rs:610-615), a peer that
    continues after seeing everything is a greater update."""
    a, b = wire.Replica(1), wire.Replica(2)
    a.text_insert("text", 0, "hello"); a.commit()
    b.text_insert("text", 0, "world"); b.commit()
    assert _oracle.import_modes([a.export()]) == ["Linear"]
    assert _oracle.import_modes([a.export(), b.export()]) == ["Linear", "Import"]
    assert _oracle.import_modes([a.export(), a.export()]) == ["Linear", "Linear"]   # nothing new: before == after (diff_calc.rs:150-152)
