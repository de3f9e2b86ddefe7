import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    """-> (spec, image uint8, tokens int32[T], logits f32[T,V]) from tests/golden/<name>.npz
    (written by tests/golden/make_golden.py from the reference's own Python code)."""
    from kuiperllama_amd import binfmt
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    kw = {k[5:]: z[k].item() for k in z.files if k.startswith("spec_")}
    spec = binfmt.ModelSpec(**kw)
    return spec, np.ascontiguousarray(z["image"]), z["tokens"], z["logits"]


GOLDEN_MODELS = ["ref_llama_gqa_tied", "ref_llama_mha_untied", "ref_qwen_bias_interleaved",
                 "ref_llama_int8_untied", "hf_llama_half", "hf_qwen2_half"]


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    # nproc can be a lie inside a cgroup (256 logical CPUs seen, 16 granted): an OpenMP team sized
    # by nproc thrashes, so size it by what the process may really use
    O.set_threads(max(1, min(O.effective_cpus(), 32)))
    return O


@pytest.fixture(scope="session")
def gpu():
    """torch device for -m gpu tests; the HIP library must load and see the device."""
    import torch
    from kuiperllama_amd import _ffi
    assert torch.cuda.is_available(), "GPU test selected but torch sees no GPU"
    assert _ffi.lib().kh_device_count() >= 1, "libkuiper_hip.so sees no HIP device"
    return torch.device("cuda:0")
