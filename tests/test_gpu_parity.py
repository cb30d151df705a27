"""GPU parity: the HIP path (through the C ABI) must be bit-exact with the CPU oracle."""
import json, os, random
import pytest

import _oracle, _fuzz
from loro_amd import workload, wire

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_fixtures.json")


@pytest.fixture(scope="module")
def engine():
    import loro_amd
    e = loro_amd.MergeEngine(0)
    yield e
    e.close()


def _same(engine, docs):
    got = engine.merge_batch(docs)
    want = _oracle.merge_batch(docs, threads=8)
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, f"doc {i}: gpu={g[:2]!r} oracle={w[:2]!r}"
    return got


def test_reference_fixture_blobs(engine):
    fx = json.load(open(GOLD))
    b = {k: bytes.fromhex(v) for k, v in fx["blobs"].items()}
    docs = [[b["fugue-left.ts.blob"], b["fugue-right.ts.blob"]], [b["fugue-right.ts.blob"], b["fugue-left.ts.blob"]]]
    got = _same(engine, docs)
    assert got[0][1] == b'{"text":"Hello World!"}' and got[1][1] == got[0][1]
    # fixtures with containers outside the device scope must be flagged, not guessed
    st = engine.merge_batch([[b["updates.blob"]]])[0][0]
    assert st == 4


def test_fuzz_sessions(engine):
    docs = []
    for seed in range(200):
        kinds = [("text",), ("text", "list"), ("text", "list", "map"), ("map",)][seed % 4]
        reps = _fuzz.random_session(seed, n_peers=2 + seed % 3, n_steps=40 + seed % 60, kinds=kinds)
        docs.append(_fuzz.blobs_of(reps, random.Random(seed)))
    _same(engine, docs)


def test_trace_shape_both_orders(engine):
    for ce, fuse in ((10, True), (10, False), (0, True)):
        tpl = workload.Cfg2Template(6000, 3000, seed=3, commit_every=ce, fuse=fuse)
        docs = []
        for d in range(8):
            s = tpl.stamp(d)
            docs.append(s)
            docs.append([s[0], s[2], s[1]])
        got = _same(engine, docs)
        for d in range(8):
            assert got[2 * d][1] == got[2 * d + 1][1] and got[2 * d][2] == got[2 * d + 1][2]


def test_wave_primitives_selftest(engine):
    # DPP prefix scan / ballot ranks against the bpermute formulation, on the device
    assert engine.b.selftest(engine.h) == 0
