"""The gfx950 code objects inside the built library: no kernel may use scratch memory (a register spill in a
latency-bound decode kernel is a memory round trip per use), and the kernels the decode step launches exist.
Reads the .hip_fatbin section of kuiperllama_amd/lib/libkuiper_hip.so with the LLVM tools of the ROCm install (no GPU).
Round 6 found 20-24 bytes of scratch in six k_wo_comb instantiations after a branch had been removed from their
merge loop (the compiler hoisted every factor read and crossed the 128-register cap): this test is the tripwire."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from kuiperllama_amd import build

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_object_notes():
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("LLVM tools of the ROCm install not found")
    lib = build.build_lib()
    td = tempfile.mkdtemp(prefix="kh_co_")
    try:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([tools[0], "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        assert starts, "no offload bundle in .hip_fatbin"
        notes = []
        for i, s in enumerate(starts):
            chunk = os.path.join(td, f"bundle{i}.bin")
            with open(chunk, "wb") as f:
                f.write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(td, f"co{i}.elf")
            subprocess.check_call([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                   f"--input={chunk}", f"--output={co}"], stderr=subprocess.DEVNULL)
            if os.path.getsize(co) == 0:
                continue  # a translation unit without device code
            notes.append(subprocess.check_output([tools[2], "--notes", co], text=True))
        return "\n".join(notes)
    finally:
        shutil.rmtree(td, ignore_errors=True)


def test_no_kernel_uses_scratch_and_step_kernels_exist():
    text = _code_object_notes()
    kernels = {}
    name = None
    for line in text.splitlines():
        m = re.search(r"\.name:\s+(\S+)", line)
        if m and m.group(1).startswith("_Z"):
            name = m.group(1)
        m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", line)
        if m:
            kernels.setdefault(name or f"?{len(kernels)}", int(m.group(1)))
            name = None
    assert len(kernels) > 100, f"only {len(kernels)} kernel descriptors parsed"
    spilled = {k: v for k, v in kernels.items() if v}
    assert not spilled, f"kernels with scratch (bytes): {spilled}"
    for stem in ("k_qkv", "k_attn_decode", "k_gemv_res", "k_wo_comb", "k_ffn13", "k_ffn13_ring", "k_cls", "k_cls_ring",
                 "k_sample", "k_pg_gemm", "k_pg_attn"):
        assert any(stem in k for k in kernels), f"no {stem} instantiation in the library"
