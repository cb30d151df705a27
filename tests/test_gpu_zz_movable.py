"""GPU parity for MovableList containers — the HIP path through the C ABI against the oracle and the
reference's known answers.  Same cases as the kernel-logic harness (test_emu_movable.py) at larger sizes; this file sorts
after test_gpu_parity.py so the suites of the earlier rows run first."""
import json, random
import pytest

import _oracle, _fuzz
import test_emu_movable as cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import loro_amd
    e = loro_amd.MergeEngine(0)
    yield e
    e.close()


def test_known_answers_and_damaged_rows(engine):
    docs, check = cases.known_answer_docs()
    check(cases._check(docs, run=engine.merge_batch))
    got = cases._check(cases.damaged_docs(), run=engine.merge_batch)
    assert [g[0] for g in got] == [3] * 5
    docs, fronts, want = cases.existence_docs()
    assert [g[1] for g in cases._check(docs, fronts, run=engine.merge_batch)] == want
    got = cases._check(cases.sliced_docs(), run=engine.merge_batch)
    assert all(g[:2] == (0, b'{"ml":["a","D","c"]}') for g in got)
    docs, want = cases.nesting_docs()
    got = cases._check(docs, run=engine.merge_batch)
    assert [g[:2] for g in got] == want and got[1][3] == 6


def test_random_sessions_nested_children_and_overlapping_histories(engine):
    docs = cases.session_docs(range(100, 140)) + cases.session_docs(range(200, 224), nested=True, n_steps=160, n_peers=4)
    got = cases._check(docs, run=engine.merge_batch)
    assert all(g[0] == 0 for g in got)


def test_multi_leaf_lists(engine):
    docs = cases.session_docs(range(300, 306), bulk=1500, n_steps=500, sync_prob=0.08)
    got = cases._check(docs, run=engine.merge_batch)
    assert all(len(json.loads(g[1])["ml"]) > 1000 for g in got)


def test_checkouts(engine):
    docs, fronts = cases.checkout_docs(range(400, 424), nested=True, n_steps=120)
    cases._check(docs, fronts, run=engine.merge_batch)


def test_mixed_batch_with_text_documents(engine):
    # MovableList documents next to plain Text documents in one batch: k_mlist_post only touches the flagged ones
    from loro_amd import workload
    tpl = workload.Cfg2Template(3000, 1500, seed=9, commit_every=10, fuse=True)
    docs = []
    for i, d in enumerate(cases.session_docs(range(500, 508), nested=True)):
        docs.append(d)
        docs.append(tpl.stamp(i))
    cases._check(docs, run=engine.merge_batch)
