"""Host-side logic of bench.py that a GPU-less box can check: the CPU-baseline leg (oracle timing on a
bounded sample + the token-for-token comparison with the GPU's words) and the algorithmic-bytes
helpers.  Round 1 shipped a comparison that was False whenever --steps < --cpu-tokens; this pins the
common-prefix semantics."""
import importlib.util
import os
import sys

import numpy as np

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bench_mod"] = mod
    spec.loader.exec_module(mod)
    return mod


def test_cpu_baseline_compares_common_prefix(oracle):
    bench = _bench()
    import torch
    from kuiperllama_amd import binfmt
    spec = binfmt.ModelSpec(64, 160, 2, 4, 2, 320, 64, True, binfmt.FAMILY_LLAMA, False, 64,
                            binfmt.ROPE_HALF, 500000.0, 1e-5, "bench-logic")  # vocab covers PROMPT
    img = binfmt.synth_image(spec, seed=5, device=torch.device("cpu")).numpy()
    prompt = bench.PROMPT
    want = oracle.OracleModel.from_spec(img, spec).generate(prompt, 24)
    # GPU list longer than the CPU pass, equal, shorter (the driver's --steps 20 case), diverging
    for gpu_words, n_cmp, ok, div in ((want, 16, True, None), (want[:16], 16, True, None),
                                      (want[:9], 9, True, None),
                                      (want[:5] + [(want[5] + 1) % spec.vocab_size] + want[6:], 16, False, 5)):
        r = bench.cpu_baseline(spec, img, gpu_words, max_tokens=16, budget_s=2.0, min_sample_s=0.2)
        assert r["tokens_compared"] == n_cmp and r["tokens_match_gpu"] is ok and r["first_divergence"] == div
        assert r["value"] > 0 and r["kind"] == "port" and r["cores"] >= 1 and "decode steps" in r["sample"]
    # nothing to compare is not a match
    r = bench.cpu_baseline(spec, img, [], max_tokens=4, budget_s=1.0, min_sample_s=0.1)
    assert r["tokens_compared"] == 0 and r["tokens_match_gpu"] is False


def test_algorithmic_bytes_match_survey_table():
    """bytes per token at mean pos 63.5 and the ffn13 launch bytes of DESIGN §3.2."""
    from kuiperllama_amd import binfmt
    bench = _bench()
    p = binfmt.PRESETS
    assert abs(p["llama3.2-1b"].algorithmic_bytes_per_token(63.5) / 1e9 - 4.948) < 0.002
    assert abs(p["llama2-7b-int8"].algorithmic_bytes_per_token(63.5) / 1e9 - 7.089) < 0.003
    assert abs(p["llama2-7b"].algorithmic_bytes_per_token(63.5) / 1e9 - 26.50) < 0.01
    assert abs(p["qwen2.5-0.5b"].algorithmic_bytes_per_token(63.5) / 1e9 - 1.978) < 0.002
    assert bench.ffn13_bytes(p["llama3.2-1b"]) == 2 * 8192 * 2048 * 4 + 2 * 2048 * 4 + 8192 * 4
    n = 2 * 11008 * 4096
    assert bench.ffn13_bytes(p["llama2-7b-int8"]) == n + n // 64 * 4 + 2 * 4096 * 4 + 11008 * 4


def test_traffic_provenance_names_the_kernel_sources():
    """roofline.traffic is read from profiles/pmc_traffic.json (separate PMC passes).  The file carries the sha1 of
    the device-code headers it was collected on and bench.py compares it with the tree it runs from, so a kernel change
    after the collection shows in the record (kernel_sources_unchanged: false) instead of keeping a stale ratio."""
    import bench
    from kuiperllama_amd.build import kernel_sources_sha1
    tr, src = bench.load_traffic("llama3.2-1b:ffn13")
    assert tr and tr > 1e8
    assert src["collected_on_commit"] and src["kernel_sources_unchanged"] in (True, False)
    assert len(kernel_sources_sha1()) == 40

