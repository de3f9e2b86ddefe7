"""bench.py's own N > 1 branch, executed end to end on CPU: two gloo ranks run bench.main() with a
stub model (the CPU oracle on a tiny golden image stands in for the HIP model - what is under test is
the control flow of main()/measure() under world > 1, not the kernels): default workload = BASELINE
config 5 (Llama-2-7B fp32 replicas), no secondary, no other configs, per-replica gather, the
max-over-ranks value, `timing_backend`, and the schema of the ONE JSON line rank 0 prints.
(the reference pins device 0, kuiper/source/model/llama3.cpp:118.)"""
import json
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _StubModel:
    """KuiperModel's surface as bench.measure() uses it, backed by the oracle on a tiny image."""

    def __init__(self, rank):
        import time
        from conftest import load_golden
        from oracle import oracle as O
        O.set_threads(1)
        self._time = time
        self._O = O
        self.rank = rank
        self.spec, self.img, self.toks, _ = load_golden("ref_llama_gqa_tied")
        self.calls = []

    def generate(self, prompt, steps, exec="graph", stop=None):
        t0 = self._time.perf_counter()
        # the golden model has a 32-row cache and a small vocabulary: wrap what bench.py asks for
        n = min(steps, self.spec.seq_len)
        words = self._O.OracleModel.from_spec(self.img, self.spec).generate(
            [int(t) % self.spec.vocab_size for t in prompt], n)
        self._time.sleep(0.01 * (self.rank + 1))  # uneven replicas
        self.calls.append(("generate", steps))
        return (words * (steps // n + 1))[:steps], (self._time.perf_counter() - t0) * 1e3

    def time_step(self, pos, reps=9):
        return [100.0 + self.rank] * reps

    def time_prefill(self, tokens, pos0=0, mode="gemm"):
        return 1.0

    def profile_kernels(self, pos, reps=8):
        return {"qkv": 6.0, "attn": 4.0, "wo": 5.0, "ffn13": 20.0, "w2": 12.0, "cls": 100.0, "sample": 3.0}

    def profile_step(self, start_pos, n_steps):
        return {k: {"avg_us": v, "launches_per_step": 1} for k, v in self.profile_kernels(0).items()}

    def close(self):
        pass


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KH_BENCH_DIST_BACKEND="gloo")
    import contextlib
    import io
    import torch
    import bench
    seen = {}

    def fake_build_model(spec, device_index, seed=1234, max_seq_len=0):
        seen["workload"] = spec.name
        return _StubModel(rank), torch.zeros(8, dtype=torch.uint8)

    bench.build_model = fake_build_model
    torch.cuda.is_available = lambda: True          # main() refuses to run without a GPU
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.empty_cache = lambda *a, **k: None
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--steps", "12", "--warmup", "2"])
    q.put((rank, buf.getvalue(), seen.get("workload")))


@pytest.mark.timeout(180)
def test_bench_main_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    (_, out0, wl0), (_, out1, wl1) = res
    assert wl0 == wl1 == "llama2-7b"                 # default workload for N > 1 = config 5
    assert out1.strip() == ""                        # only rank 0 prints
    lines = [ln for ln in out0.splitlines() if ln.strip()]
    assert len(lines) == 1, out0                     # exactly ONE JSON line
    j = json.loads(lines[0])
    # the driver's contract keys
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "replicas"):
        assert k in j, k
    assert j["n_gpus"] == 2 and j["steps"] == 12 and j["warmup"] == 2 and j["scaling"] == "weak"
    assert j["higher_is_better"] is True and j["vs_baseline"] is None and j["unit"] == "tokens/s"
    assert j["config"]["workload"].startswith("llama2-7b greedy decode")
    assert j["config"]["timing_backend"] == "gloo" and "replicas2" in j["config"]["parallelism"]
    # whole-job aggregate = world * steps / max wall; ms_per_step is that max wall per step
    assert abs(j["value"] - 2 * 12 / (j["ms_per_step"] * 12e-3)) < 1e-6 * j["value"]
    assert j["ms_per_step"] * 12e-3 >= 0.02          # the slower replica (0.02 s sleep) bounds the wall
    rep = j["replicas"]
    assert len(rep["per_replica"]) == 2 and rep["min"] <= rep["max"] and 0 <= rep["spread_frac"] < 1
    # N > 1: no secondary workload, no other configs, no CPU baseline, no long-context probe
    for k in ("secondary", "other_configs", "cpu_baseline", "long_context", "north_star_floor"):
        assert k not in j, k
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["traffic_source"]["measured_in_this_run"] is False and rf["traffic_source"]["file"].endswith(".json")
    from kuiperllama_amd import binfmt
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bm)
    assert rf["bytes_per_launch"] == bm.ffn13_bytes(binfmt.PRESETS["llama2-7b"])
    assert abs(rf["achieved"] - rf["bytes_per_launch"] / 20e-6 / 1e9) < 1e-6 * rf["achieved"]


def test_bench_flags_can_be_switched_off():
    """--prefill-secondary was store_true with default True (could never be off)."""
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod3", os.path.join(ROOT, "bench.py"))
    bm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bm)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "BooleanOptionalAction" in src and "--no-others" in src
    assert bm.DEFAULT_OTHERS.split(",") == ["stories15M", "qwen2.5-0.5b", "tinyllama-1.1b", "llama2-7b"]
    assert bm.NORTH_STAR_FLOOR["config"] == "tinyllama-1.1b" and bm.NORTH_STAR_FLOOR["target_tok_s"] == 60.0
