"""C++ host adapter (include/kuiper_hip_adapter.hpp): kernels_interface.h-shaped functions over
the C-ABI, exercised by a C++ program that restates the reference's own op tests."""
import subprocess

import pytest

from kuiperllama_amd import build


def _run():
    exe = build.build_adapter_test()
    return subprocess.run([exe], capture_output=True, text=True, timeout=120)


def test_adapter_compiles_and_links_without_gpu():
    import torch
    r = _run()
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        assert r.returncode == 77 and "no HIP device" in r.stdout  # loads the .so, no compute


@pytest.mark.gpu
def test_adapter_reference_op_tests_on_gpu(gpu):
    r = _run()
    assert r.returncode == 0 and "OK adapter tests passed" in r.stdout, r.stdout + r.stderr
