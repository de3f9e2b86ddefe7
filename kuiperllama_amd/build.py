"""Build the HIP shared library in-tree (hipcc cross-compiles gfx950 without a GPU).

    python -m kuiperllama_amd.build [--force]

Output: kuiperllama_amd/lib/libkuiper_hip.so (git-ignored, shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
LIB_DIR = os.path.join(_PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libkuiper_hip.so")
SOURCES = ["kh_ops.hip", "kh_model_load.hip", "kh_model_step.hip", "kh_model_prefill.hip", "kh_model_profile.hip", "kh_model_selftest.hip",
           "kh_tokenizer.cpp", "kh_bpe.cpp", "kh_debug.cpp"]
HEADERS = ["kh_common.h", "kh_gemv.h", "kh_attn.h", "kh_fused.h", "kh_q8ring.h", "kh_fused_ring.h", "kh_prefill.h", "kh_gemm.h", "kh_pattn.h", "kh_unicode_tables.h",
           "kh_model_internal.h",
           "../../include/kuiper_hip.h"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=...)")


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    for f in SOURCES + HEADERS:
        p = os.path.normpath(os.path.join(CSRC, f))
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, os.path.splitext(src)[0] + ".o")
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall",
               # no implicit FMA contraction: results must not depend on which kernel a stage is
               # inlined into (decode vs B-token prefill kernels are compared bit for bit) and the
               # CPU oracle is built the same way; explicit __builtin_fmaf calls stay FMAs
               "-ffp-contract=off",
               "-Wno-unused-function", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))  # the translation units compile side by side
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


def build_variant(name: str, defines: list[str], commit: str | None = None) -> str:
    """Experiment build: kuiperllama_amd/lib/<name>.so = the library with extra -D flags (KH_LIB selects it at run
    time, kuiperllama_amd/_ffi.py).  Used by tools/gpu_job_*.sh for same-box A/Bs of compile-time constants.
    `commit`: build the sources as they were at that commit (`--variant-at <commit> <name> ...`: an A/B against an
    earlier form of a kernel without keeping the earlier form alive behind a macro)."""
    out = os.path.join(LIB_DIR, name + ".so")
    if commit:
        import tarfile
        import tempfile
        root = os.path.normpath(os.path.join(_PKG, ".."))
        with tempfile.TemporaryDirectory() as td:
            tar = os.path.join(td, "src.tar")
            subprocess.check_call(["git", "-C", root, "archive", "-o", tar, commit, "kuiperllama_amd/csrc", "include"])
            with tarfile.open(tar) as tf:
                tf.extractall(td)
            global CSRC
            saved, CSRC = CSRC, os.path.join(td, "kuiperllama_amd", "csrc")
            try:
                return build_variant(name, defines)
            finally:
                CSRC = saved
    objs, procs = [], []
    od = os.path.join(LIB_DIR, "_" + name)
    os.makedirs(od, exist_ok=True)
    for src in SOURCES:
        if not os.path.exists(os.path.join(CSRC, src)):  # --variant-at an older commit: that source came later
            continue
        obj = os.path.join(od, os.path.splitext(src)[0] + ".o")
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-w"] + \
              [f"-D{d}" for d in defines] + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    subprocess.check_call([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + objs)
    shutil.rmtree(od, ignore_errors=True)
    return out


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":  # python -m kuiperllama_amd.build --variant exp_ts128 KH_ATTN_MIN_TS=128
        print(build_variant(sys.argv[2], sys.argv[3:]))
    elif len(sys.argv) > 3 and sys.argv[1] == "--variant-at":  # ... --variant-at <commit> exp_old [defines]
        print(build_variant(sys.argv[3], sys.argv[4:], commit=sys.argv[2]))
    else:
        print(build_lib(force="--force" in sys.argv, verbose=True))


ADAPTER_TEST_SRC = os.path.normpath(os.path.join(_PKG, "..", "tests", "cpp", "test_adapter.cpp"))
ADAPTER_TEST_BIN = os.path.join(LIB_DIR, "test_adapter")


def build_adapter_test(force: bool = False) -> str:
    """C++ host-side test of include/kuiper_hip_adapter.hpp, linked against the C-ABI .so."""
    build_lib()
    inc = os.path.normpath(os.path.join(_PKG, "..", "include"))
    deps = [ADAPTER_TEST_SRC, os.path.join(os.path.dirname(ADAPTER_TEST_SRC), "adapter_cases.hpp"),
            os.path.join(inc, "kuiper_hip_adapter.hpp"), os.path.join(inc, "kuiper_hip_alloc.hpp"), os.path.join(inc, "kuiper_hip.h"), LIB_PATH]
    if (not force and os.path.exists(ADAPTER_TEST_BIN)
            and all(os.path.getmtime(d) <= os.path.getmtime(ADAPTER_TEST_BIN) for d in deps)):
        return ADAPTER_TEST_BIN
    subprocess.check_call([_hipcc(), "-std=c++17", "-O2", f"-I{inc}", ADAPTER_TEST_SRC, "-o",
                           ADAPTER_TEST_BIN, f"-L{LIB_DIR}", "-lkuiper_hip",
                           "-Wl,-rpath,$ORIGIN"])
    return ADAPTER_TEST_BIN


DEMO_SRC = os.path.normpath(os.path.join(_PKG, "..", "tools", "kuiper_demo.cpp"))
DEMO_BIN = os.path.join(LIB_DIR, "kuiper_demo")


def build_demo(force: bool = False) -> str:
    """Command-line twin of the reference's demo/main.cpp over the C-ABI."""
    build_lib()
    inc = os.path.normpath(os.path.join(_PKG, "..", "include"))
    deps = [DEMO_SRC, os.path.join(inc, "kuiper_hip.h"), LIB_PATH]
    if (not force and os.path.exists(DEMO_BIN)
            and all(os.path.getmtime(d) <= os.path.getmtime(DEMO_BIN) for d in deps)):
        return DEMO_BIN
    subprocess.check_call([_hipcc(), "-std=c++17", "-O2", f"-I{inc}", DEMO_SRC, "-o", DEMO_BIN,
                           f"-L{LIB_DIR}", "-lkuiper_hip", "-Wl,-rpath,$ORIGIN"])
    return DEMO_BIN


REF_ROOT = os.environ.get("KUIPER_REF", "/root/reference")
REF_BINDING_BIN = os.path.normpath(os.path.join(_PKG, "..", "oracle", "_ref", "test_ref_binding"))


def build_ref_binding() -> str | None:
    """tests/cpp/test_ref_binding.cpp against the reference's REAL headers and tensor sources
    (oracle/Makefile `ref`): only where the reference checkout exists; the binary lands in
    oracle/_ref/ (git-ignored) and travels to the GPU box.  Returns None without a checkout."""
    if not os.path.isdir(os.path.join(REF_ROOT, "kuiper", "include")):
        return REF_BINDING_BIN if os.path.exists(REF_BINDING_BIN) else None
    build_lib()
    subprocess.check_call(["make", "-s", "-C", os.path.normpath(os.path.join(_PKG, "..", "oracle")),
                           "ref", f"KUIPER_REF={REF_ROOT}", f"HIPCC={_hipcc()}"])
    return REF_BINDING_BIN


REF_LAYERS_BIN = os.path.normpath(os.path.join(_PKG, "..", "oracle", "_ref", "test_ref_layers"))


def build_ref_layers() -> str | None:
    """tests/cpp/test_ref_layers.cpp: the reference's OWN op::*Layer classes (op/{layer,matmul,
    rmsnorm,rope,mha,swiglu,add,embedding}.cpp compiled where they lie) over
    tests/cpp/kernels_interfaces_hip.cpp + libkuiper_hip.so (oracle/Makefile `ref_layers`).  Only
    where the reference checkout exists; the binary lands in oracle/_ref/ and travels to the GPU box."""
    if not os.path.isdir(os.path.join(REF_ROOT, "kuiper", "include")):
        return REF_LAYERS_BIN if os.path.exists(REF_LAYERS_BIN) else None
    build_lib()
    subprocess.check_call(["make", "-s", "-C", os.path.normpath(os.path.join(_PKG, "..", "oracle")),
                           "ref_layers", f"KUIPER_REF={REF_ROOT}", f"HIPCC={_hipcc()}"])
    return REF_LAYERS_BIN


REF_MODEL_BIN = os.path.normpath(os.path.join(_PKG, "..", "oracle", "_ref", "test_ref_model"))


def build_ref_model(qwen2: bool = False) -> str | None:
    """tests/cpp/test_ref_model.cpp: the reference's OWN model::LLama2Model (model/{model,llama3,raw_model_data}.cpp,
    sampler/argmax_sampler.cpp, op/encode.cpp compiled where they lie) over the HIP getters + libkuiper_hip.so
    (oracle/Makefile `ref_model`); qwen2=True: the model::Qwen2Model twin (model/qwen2.cpp, `ref_model_qwen2`).
    Only where the reference checkout exists; the binary travels to the GPU box."""
    exe = REF_MODEL_BIN + ("_qwen2" if qwen2 else "")
    if not os.path.isdir(os.path.join(REF_ROOT, "kuiper", "include")):
        return exe if os.path.exists(exe) else None
    build_lib()
    subprocess.check_call(["make", "-s", "-C", os.path.normpath(os.path.join(_PKG, "..", "oracle")),
                           "ref_model_qwen2" if qwen2 else "ref_model", f"KUIPER_REF={REF_ROOT}", f"HIPCC={_hipcc()}"])
    return exe


REF_CPU_BINS = {flavor: os.path.normpath(os.path.join(_PKG, "..", "oracle", "_ref", "ref_cpu_model" + suffix))
                for flavor, suffix in (("default", ""), ("llama3", "_llama3"), ("qwen2", "_qwen2"))}


def build_ref_cpu() -> dict | None:
    """oracle/_ref/ref_cpu_model{,_llama3,_qwen2}: the reference's OWN CPU backend - its ten CPU kernels
    (kuiper/source/op/kernels/cpu/*.cpp compiled where they lie over tests/cpp/ref_stubs/armadillo + numpy's OpenBLAS), CPU
    getters, operator and model classes - in its three compile-time flavours (oracle/Makefile `ref_cpu`).  The checker
    of tests/test_ref_cpu_backend.py and bench.py's timed CPU baseline (kind "reference").  Only where the reference
    checkout exists; the binaries travel to the GPU box.  Returns {flavour: path} or None."""
    if os.path.isdir(os.path.join(REF_ROOT, "kuiper", "include")):
        build_lib()
        subprocess.check_call(["make", "-s", "-C", os.path.normpath(os.path.join(_PKG, "..", "oracle")),
                               "ref_cpu", f"KUIPER_REF={REF_ROOT}", f"HIPCC={_hipcc()}"])
    return dict(REF_CPU_BINS) if all(os.path.exists(p) for p in REF_CPU_BINS.values()) else None


def ref_cpu_flavor(spec) -> str | None:
    """Which build of the reference CPU backend runs a model of this spec (its RoPE flavour / theta / eps are
    compile-time switches: cpu/rope_kernel.cpp:3-16,43,83, cpu/rmsnorm_kernel.cpp:24-28); None: no such build."""
    from . import binfmt
    if spec.quant:
        return None  # the reference has no CPU int8 path
    key = (spec.family, spec.rope_mode, float(spec.rope_theta), float(spec.rms_eps))
    table = {(binfmt.FAMILY_LLAMA, binfmt.ROPE_INTERLEAVED, 10000.0, 1e-5): "default",
             (binfmt.FAMILY_LLAMA, binfmt.ROPE_HALF, 500000.0, 1e-5): "llama3",
             (binfmt.FAMILY_QWEN2, binfmt.ROPE_HALF, 1000000.0, 1e-6): "qwen2"}
    return table.get(key)


def kernel_sources_sha1() -> str:
    """sha1 over the device-code headers of the decode / prefill kernels (csrc/kh_*.h except the host-only
    kh_model_internal.h), in name order.  profiles/pmc_traffic.json is stamped with it when the PMC passes are
    collected; bench.py recomputes it, so a kernel change after the collection shows in the record
    (roofline.traffic_source.kernel_sources_unchanged) instead of silently keeping a stale traffic ratio."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    for f in sorted(os.listdir(d)):
        if f.startswith("kh_") and f.endswith(".h") and f != "kh_model_internal.h":
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()

