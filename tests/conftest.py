import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The suites pin the batch's integrate kernel: LM_SPAN selects it where a test wants the element-granular one, and the product's
# rule "a batch of small common-kernel documents takes the element-granular kernel" (lm_pipeline.h, LM_SPAN_AUTO) is switched off so
# that the span-granular kernels keep the coverage these small test documents give them; the rule itself has its own tests
# (test_small_document_batches_pick_the_element_granular_kernel, CPU and GPU), which switch it back on.
os.environ.setdefault("LM_SPAN_AUTO", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
