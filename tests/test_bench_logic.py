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
    # per-launch bytes of every weight-streaming kernel (DESIGN 3.2's table, profiles/README.md's generated tables)
    kb = p["llama3.2-1b"].kernel_bytes()
    assert kb["ffn13"] == bench.ffn13_bytes(p["llama3.2-1b"]) == 134266880
    assert [round(kb[k] / 1e6, 1) for k in ("qkv", "wo", "w2", "cls")] == [25.2, 16.8, 67.2, 1051.2]
    kq = p["llama2-7b-int8"].kernel_bytes()
    assert kq["ffn13"] == bench.ffn13_bytes(p["llama2-7b-int8"])
    assert [round(kq[k] / 1e6, 1) for k in ("qkv", "wo", "w2", "cls")] == [53.6, 17.9, 48.0, 139.4]
    # the five kernels x L layers + classifier carry the weight bytes of a token
    for name in ("llama3.2-1b", "llama2-7b-int8"):
        sp, k = p[name], p[name].kernel_bytes()
        per_tok = sp.n_layers * (k["qkv"] + k["wo"] + k["ffn13"] + k["w2"]) + k["cls"]
        assert abs(per_tok / sp.algorithmic_bytes_per_token(0) - 1) < 0.005


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



def test_line_layout_carries_both_halves_of_the_metric_as_scalars():
    """the driver's record keeps the contract's objects (`config`, `roofline`, `cpu_baseline`) with
    their scalar members and only the NAMES of other top-level keys, so the Llama-2-7B int8 half of BASELINE's metric
    and both whole-step fractions must sit inside those objects as scalars; contract scalars first, bulky sections
    last, `summary` at the very end."""
    import argparse
    import json
    from kuiperllama_amd import binfmt
    bench = _bench()
    spec = binfmt.PRESETS["llama3.2-1b"]
    args = argparse.Namespace(steps=20, warmup=5)
    roof = {"bound": "hbm", "kernel": "k_ffn13", "achieved": 6200.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.775,
            "traffic": 1.345e8, "traffic_source": {"file": "x"}, "bytes_per_launch": 1.34e8, "avg_launch_us": 21.6,
            "step": {"achieved": 5300.0, "frac": 0.6625, "bytes_per_token": 4.948e9}, "kernels_avg_us": {"ffn13": 21.6},
            "kernels_sum_us_per_token": 916.0}
    res = {"value": 1070.0, "ms_per_step": 0.9346, "roofline": roof, "replicas": {"n": 1}, "tok_s_128_steps": 1060.0,
           "cpu_baseline": {"value": 33.0, "unit": "tokens/s", "cores": 16, "kind": "reference", "sample": "s",
                            "host": {"nproc": 256}, "port": {"value": 20.0}}}
    sroof = dict(roof, achieved=5700.0, frac=0.7125, step={"achieved": 4430.0, "frac": 0.55375, "bytes_per_token": 7.03e9})
    secondary = {"config": {"workload": "llama2-7b-int8 greedy decode"}, "value": 630.0, "unit": "tokens/s",
                 "ms_per_step": 1.587, "roofline": sroof, "tok_s_128_steps": 621.0}
    line = bench.assemble_line(spec, args, 1, res, secondary, {"tinyllama-1.1b": {"value": 1090.0, "tokens_match": True}},
                               [], "none")
    keys = list(line)
    # the contract's keys, in the contract's order, ahead of everything else
    contract = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype"]
    assert keys[:len(contract)] == contract
    # small scalars before any object; the bulky sections behind the contract's objects; summary last
    first_obj = next(i for i, k in enumerate(keys) if isinstance(line[k], (dict, list)))
    for k in ("tok_s_128_steps", "secondary_value", "secondary_ms_per_step", "secondary_step_frac"):
        assert keys.index(k) < first_obj and isinstance(line[k], float)
    for k in ("secondary", "other_configs", "prefill", "runs"):
        assert keys.index(k) > keys.index("cpu_baseline") > keys.index("roofline") > keys.index("config")
    assert keys[-1] == "summary"
    # what a parser that keeps only scalar members of the contract's objects still sees
    r = {k: v for k, v in line["roofline"].items() if not isinstance(v, (dict, list))}
    assert r["frac"] == 0.775 and r["step_frac"] == 0.6625 and r["step_achieved"] == 5300.0
    assert r["secondary_frac"] == 0.7125 and r["secondary_step_frac"] == 0.55375
    c = {k: v for k, v in line["config"].items() if not isinstance(v, (dict, list))}
    assert c["secondary_workload"].startswith("llama2-7b-int8") and c["secondary_value"] == 630.0
    assert c["secondary_tok_s_128_steps"] == 621.0 and c["tok_s_128_steps"] == 1060.0
    # scalars of an object come before its nested members
    rk = list(line["roofline"])
    assert max(rk.index(k) for k in r) < min(rk.index(k) for k in ("traffic_source", "step", "kernels_avg_us"))
    assert list(line["cpu_baseline"])[-2:] == ["host", "port"]
    assert line["summary"]["secondary_value"] == 630.0 and line["north_star_floor"]["met"] is True
    json.dumps(line)
    # without a secondary workload (N > 1) the keys stay, as nulls
    line = bench.assemble_line(spec, args, 2, res, None, None, ["load"], "nccl")
    assert line["secondary_value"] is None and line["roofline"]["secondary_step_frac"] is None
    assert "secondary" not in line and line["skipped_sections"] == ["load"] and list(line)[-1] == "summary"


def test_profiles_readme_tables_are_generated_from_the_profile_files():
    """profiles/README.md quoted microseconds by hand and they drifted from the CSV they came
    from.  The per-round kernel tables are now written by tools/profiles_readme.py from <round>_kernel_stats.csv,
    <round>_pmc_<workload>.csv and the bench JSONs; this test regenerates them and requires README.md to hold
    exactly that text, so a re-collected profile without a regenerated README (or a hand edit) fails here."""
    spec = importlib.util.spec_from_file_location("profiles_readme", os.path.join(ROOT, "tools", "profiles_readme.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    readme = open(os.path.join(ROOT, "profiles", "README.md")).read()
    for rnd in ("r5", "r6"):
        b = gen.block(rnd)
        assert b in readme, f"profiles/README.md is stale for {rnd}: run `python tools/profiles_readme.py --round {rnd} --write`"
        rows = [ln for ln in b.split("\n") if ln.startswith("| llama")]
        assert len(rows) == 10  # five weight-streaming kernels of each metric workload
        for ln in rows:
            cells = [c.strip() for c in ln.strip("|").split("|")]
            frac, ratio = float(cells[6]), cells[7]
            assert 0.3 < frac < 0.9
            assert ratio == "—" or 0.99 < float(ratio) < 1.05  # nothing is read twice from HBM
    # the classifier keeps kernels of the metric workloads apart by instantiation, wo / w2 by workgroup width
    assert gen.classify("k_gemv_res<true, 2, 6, 2>", 512) == ("llama2-7b-int8", "w2")
    assert gen.classify("k_gemv_res<false, 4, 2, 2>", 256) == ("llama3.2-1b", "wo")
    assert gen.classify("k_ffn13_ring<2, 4, false, StagerAsm<true, 4, 0> >", 256) == ("llama2-7b-int8", "ffn13")
    assert gen.classify("k_pf_ffn13<true, 4>", 256) is None
