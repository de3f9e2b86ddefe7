"""N>1 path = independent replicas + timing protocol (kuiperllama_amd/replicas.py), covered with
world_size-2 gloo processes on CPU.  The per-replica workload here is the CPU oracle decoding a
tiny golden model (the HIP path needs a GPU); what is under test is the protocol: barrier on
both sides, max over ranks, aggregate = world*steps / max wall, identical tokens on every
replica (same seed/prompt)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import time
    from conftest import load_golden
    from kuiperllama_amd import replicas
    from oracle import oracle as O
    O.set_threads(1)
    r, w, lr = replicas.init_from_env("gloo")
    assert (r, w, lr) == (rank, world, rank)
    spec, img, toks, _ = load_golden("ref_llama_gqa_tied")
    om = O.OracleModel.from_spec(img, spec)
    out = {}

    def run():
        out["words"] = om.generate([int(t) for t in toks[:2]], 16)
        time.sleep(0.05 * (rank + 1))  # uneven replicas: the max must win

    wall, agg = replicas.timed_replica_run(run, 16, w, torch.device("cpu"), lambda: None)
    own = 16 / (0.05 * (rank + 1))  # a per-replica figure that differs by rank
    per = replicas.gather_per_replica(own, w, torch.device("cpu"))
    q.put((rank, wall, agg, out["words"], per, replicas.backend_in_use(w),
           replicas.default_workload(w)))
    replicas.shutdown(w)


@pytest.mark.timeout(120)
def test_two_replicas_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    (r0, w0, a0, words0, per0, be0, wl0), (r1, w1, a1, words1, per1, be1, wl1) = res
    # N > 1 measures BASELINE config 5 (Llama-2-7B fp32 replicas) and reports every replica
    assert wl0 == wl1 == "llama2-7b" and be0 == be1 == "gloo"
    assert per0 == per1 and len(per0) == world and per0[0] == 2 * per0[1]
    from kuiperllama_amd import replicas
    sp = replicas.spread(per0)
    assert sp["min"] == per0[1] and sp["max"] == per0[0] and abs(sp["spread_frac"] - 0.5) < 1e-12
    assert words0 == words1 and len(words0) == 16   # replicas decode the same tokens
    assert w0 == w1                                  # max over ranks is shared
    assert w0 >= 0.1                                 # the slower replica (0.1 s sleep) dominates
    assert abs(a0 - world * 16 / w0) < 1e-9 and a0 == a1


def test_single_process_path_needs_no_process_group():
    sys.path.insert(0, ROOT)
    from kuiperllama_amd import replicas
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    assert replicas.init_from_env("gloo") == (0, 1, 0)
    assert replicas.default_workload(1) == "llama3.2-1b" and replicas.backend_in_use(1) == "none"
    assert replicas.gather_per_replica(3.5, 1, torch.device("cpu")) == [3.5]
    wall, agg = replicas.timed_replica_run(lambda: None, 10, 1, torch.device("cpu"), lambda: None)
    assert wall > 0 and abs(agg - 10 / wall) < 1e-6
    replicas.shutdown(1)


@pytest.mark.gpu
def test_timing_protocol_on_rccl_single_rank(gpu):
    """The N > 1 bench path on the backend it really uses: a one-rank RCCL ("nccl") process group on
    the GPU box carries the barrier, the max-reduce of the wall time and the per-replica gather on
    DEVICE tensors, and the bench line would report that backend.  (N > 1 itself is covered by the
    world_size-2 gloo test above; the driver runs the 8-GPU launch.)"""
    import socket
    import torch.distributed as dist
    from kuiperllama_amd import replicas
    assert not dist.is_initialized()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    old = {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu)
        assert replicas.backend_in_use(1) == "nccl"
        x = torch.zeros(1 << 20, device=gpu)

        def run():
            for _ in range(8):
                x.add_(1.0)

        wall, agg = replicas.timed_replica_run(run, 8, 1, gpu, torch.cuda.synchronize)
        assert wall > 0 and abs(agg - 8 / wall) < 1e-6 * agg
        per = replicas.gather_per_replica(123.5, 1, gpu)
        assert per == [123.5]
        assert replicas.spread(per)["spread_frac"] == 0.0
        assert float(x[0]) == 8.0
        replicas.shutdown(1)
        assert not dist.is_initialized()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
