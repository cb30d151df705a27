"""bench.py's synthetic-input generators run without a GPU: the guarded MovableList leg of `other_configs`
must hand the engine valid documents — checked here against the oracle and the kernel-logic harness."""
import os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import _oracle, _emu


def test_movable_leg_documents_are_valid_and_render_the_same_everywhere():
    docs = [bench._gen(("movable", d))[0] for d in range(2)]
    want = _oracle.merge_batch(docs)
    assert all(w[0] == 0 and b'"ml":[' in w[1] for w in want)
    assert _emu.merge_batch(docs) == want
