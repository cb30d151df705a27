"""GPU parity for richtext values (lm_richtext; SURVEY.md §8f N4, second half): k_richtext on the device through the C ABI against
the oracle (Doc::to_richtext) and the reference's known answers.  Same cases as the kernel-logic harness (test_emu_richtext.py) at
larger sizes, plus configs[4]-shaped documents (1 % of the ops are bold marks) at the latest version and at checkouts."""
import json, os
import pytest

import _oracle, _richtext

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import loro_amd
    e = loro_amd.MergeEngine(0)
    yield e
    e.close()


def _run(engine, docs, fronts=None):
    res = engine.merge_batch(docs, fronts)
    return res, engine.richtext()


def test_known_answers(engine):
    ka = _richtext.known_answers()
    res, got = _run(engine, [b for _, b, _ in ka])
    for (name, _, want), (st, js) in zip(ka, got):
        assert st == 0 and json.loads(js) == {"cid:root-text:Text": want}, (name, js)


def test_the_answer_the_reference_asserts_for_its_runtime_fixtures(engine):
    """crates/loro/tests/loro_js_interop.rs:86-94: get_richtext_value of runtime-snapshot.ts.blob / runtime-updates.ts.blob
    == [{"insert":"b","attributes":{"bold":true}}] — reference-held answer, reference-shipped blobs, through the HIP path"""
    ra = _richtext.reference_held()
    res, got = _run(engine, [b for _, b, _ in ra])
    for (name, _, want), (st, js) in zip(ra, got):
        assert json.loads(js) == {"cid:root-text:Text": want}, (name, st, js)
    # (the document's status is LM_UNSUPPORTED for its Tree / Counter containers; the richtext result has a status of its own)
    assert [j for _, j in got] == [j for _, j in _oracle.richtext_batch([b for _, b, _ in ra])]


@pytest.mark.parametrize("span", ["1", "0"])
def test_batch_documents_under_both_integrate_kernels(engine, monkeypatch, span):
    monkeypatch.setenv("LM_SPAN", span)
    docs = [b for _, b in _richtext.hand_cases()] + _richtext.fuzz_docs(120, n_steps=140) + _richtext.nested_docs(16)
    docs.append([docs[0][0][:-2] + b"\x00\x01"])
    res, got = _run(engine, docs)
    assert res == _oracle.merge_batch(docs, threads=8)
    _richtext.same(got, _oracle.richtext_batch(docs), "span=" + span)


def test_product_default_kernel_choice(engine, monkeypatch):
    monkeypatch.setenv("LM_SPAN_AUTO", "1")
    docs = _richtext.fuzz_docs(64, base=5600)
    res, got = _run(engine, docs)
    _richtext.same(got, _oracle.richtext_batch(docs), "auto")


def test_checkouts(engine, monkeypatch):
    monkeypatch.setenv("LM_SHARE_REPLAY", "0")
    docs, fronts = _richtext.checkout_cases(n=16)
    res, got = _run(engine, docs, fronts)
    assert res == _oracle.merge_batch(docs, frontiers=fronts)
    _richtext.same(got, _oracle.richtext_batch(docs, frontiers=fronts), "checkout")


def test_resident_documents_step_by_step(engine):
    sessions = _richtext.resident_sessions(range(6200, 6232), n_steps=6)
    want = _richtext.oracle_resident(sessions)
    got = _richtext.run_resident(engine, sessions)
    for k, (g, w) in enumerate(zip(got, want)):
        keep = [x[0] == 0 for x in g]
        _richtext.same([(x[1], x[2]) for x, kp in zip(g, keep) if kp], [y for y, kp in zip(w, keep) if kp], "step %d" % k)


def test_config5_shaped_documents_with_bold_marks(engine, monkeypatch):
    """two peers alternating every 1,000 trace actions, 1 % of the actions are bold marks (workload.cfg5_doc): several hundred
    StyleOps per document, many leaves — at the latest version and at recorded checkouts"""
    from loro_amd import workload
    monkeypatch.setenv("LM_SHARE_REPLAY", "0")
    docs, fronts = [], []
    for d in range(6):
        blobs, fr = workload.cfg5_doc(d, n_ops=20000, n_checkouts=3)
        docs.append(blobs); fronts.append(None)
        for f in fr:
            docs.append(list(blobs)); fronts.append(f)
    res, got = _run(engine, docs, fronts)
    want = _oracle.richtext_batch(docs, frontiers=fronts)
    _richtext.same(got, want, "cfg5")
    assert all(st == 0 for st, _ in got) and max(js.count(b'"attributes"') for _, js in got) > 50


@pytest.mark.parametrize("auto", ["0", "1"])
def test_damaged_rich_text_documents_are_rendered_like_the_reference_or_rejected(engine, monkeypatch, auto):
    """400 damaged rich-text documents, suites' kernel choice and product default: what both sides accept is rendered alike — JSON,
    version vector, richtext — and the device never renders a document the oracle rejects (more seeds: the kernel-logic suite)"""
    monkeypatch.setenv("LM_SPAN_AUTO", auto)
    nb, _ = _richtext.check_damaged(lambda docs: _run(engine, docs), _richtext.damaged_docs(400, seed=5))
    assert nb >= 40


@pytest.mark.parametrize("seed,auto", [(3, "1"), (5, "1"), (6, "1"), (16, "0"), (901, "1")])
def test_damaged_mixed_documents(engine, monkeypatch, seed, auto):
    """damaged documents over rich-text / list / map sessions, nested containers and MovableLists (600 per seed; 3, 5, 6 are the seeds
    that turned up the last-lamport rule, the surplus message-length run and the insert beyond the end): rendered like the reference
    or rejected, never a document the oracle rejects"""
    monkeypatch.setenv("LM_SPAN_AUTO", auto)
    nb, _ = _richtext.check_damaged(lambda docs: _run(engine, docs), _richtext.damaged_mixed_docs(600, seed=seed))
    assert nb >= 30


def test_a_damaged_option_tag_in_front_of_an_empty_delta_of_delta_column(engine, monkeypatch):
    import _cases
    bad = bytes.fromhex(_cases.DOD_TAG_BLOB_HEX)
    for dec in ("1", "0"):
        monkeypatch.setenv("LM_DECODE", dec)
        assert engine.merge_batch([[bad]])[0][0] == 1 == _oracle.merge_batch([[bad]])[0][0]


def test_damaged_change_meta_columns_are_data_corruption_like_the_reference(engine, monkeypatch):
    """block_encode.rs:563-571: both change_meta decoders' failures are DecodeDataCorruptionError"""
    import _cases
    names, docs = _cases.damaged_change_meta_docs()
    assert [w[0] for w in _oracle.merge_batch(docs)] == [3] * len(docs)
    for dec in ("1", "0"):
        monkeypatch.setenv("LM_DECODE", dec)
        got = engine.merge_batch(docs)
        assert [g[0] for g in got] == [3] * len(docs), (dec, list(zip(names, [g[0] for g in got])))
