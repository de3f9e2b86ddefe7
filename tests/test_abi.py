"""The C-ABI library loads without a GPU and exports every symbol include/kuiper_hip.h declares;
argument validation happens before any device call.  CPU only (no compute)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, load_golden
from kuiperllama_amd import _ffi, build


@pytest.fixture(scope="module")
def lib():
    build.build_lib()
    return _ffi.lib()


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "kuiper_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kh_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared_functions()
    assert len(names) >= 28
    for n in names:
        assert hasattr(lib, n), f"{n} declared in kuiper_hip.h but not exported"
    assert sorted(_ffi.EXPORTS) == names


def test_version_and_error_strings(lib):
    assert lib.kh_version() == 100
    assert _ffi.error_string(0) == "success"
    assert "invalid" in _ffi.error_string(-1)
    assert lib.kh_kclass_name(3).decode() == "ffn13"


def test_invalid_arguments_are_rejected_without_touching_the_device(lib):
    assert lib.kh_add_f32(None, None, None, 4, None) == -1
    assert lib.kh_matmul_f32(None, None, None, 4, 4, 1.0, None) == -1
    assert lib.kh_matmul_q8(None, None, None, 64, None, 64, 4, None) == -1
    assert lib.kh_rmsnorm_f32(None, None, None, 0, 1e-5, None) == -1
    assert lib.kh_argmax_f32(None, 0, None, None) == -1
    assert lib.kh_model_get_config(None, None) == -1
    h = C.c_void_p()
    assert lib.kh_model_create_from_file(b"/nonexistent/model.bin",
                                         C.byref(_ffi.ModelOpts(0, 0, 0, 1e4, 1e-5, 0, 0, 0)),
                                         C.byref(h)) == -3  # KH_ERR_IO (reference: PathNotValid)


def test_model_creation_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from kuiperllama_amd.model import KuiperModel
    spec, img, _, _ = load_golden("ref_llama_gqa_tied")
    with pytest.raises(_ffi.KhError) as ei:
        KuiperModel.from_host_image(img, spec)
    assert ei.value.code == -5  # KH_ERR_NO_DEVICE: no silent CPU fallback


def test_ops_refuse_cpu_tensors():
    import torch
    from kuiperllama_amd import ops
    a = torch.zeros(4)
    with pytest.raises(ValueError):
        ops.add(a, a, a)
