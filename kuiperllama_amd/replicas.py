"""Multi-GPU = independent replicas: batch-1 greedy decode is single-stream and
autoregressive, the reference has no sharding and pins device 0 (llama3.cpp:118), so N GPUs run
N copies of the path with NO data-path collective.  The only communication is the timing
protocol of the benchmark: barrier, then max-over-ranks of the per-rank wall time.

Kept separate from bench.py so the N>1 logic is covered by world_size-2 gloo tests on CPU.
"""
from __future__ import annotations

import os
from typing import Callable, Tuple

import torch


# BASELINE.json config 5: the 8-GPU number is Llama-2-7B fp32, one independent replica per GPU
# (the reference pins device 0, llama3.cpp:118); N = 1 is configs[1].
REPLICA_WORKLOAD = "llama2-7b"
SINGLE_WORKLOAD = "llama3.2-1b"


def default_workload(world: int) -> str:
    """Workload bench.py measures when --workload is not given."""
    return REPLICA_WORKLOAD if world > 1 else SINGLE_WORKLOAD


def _dist():
    """torch.distributed when a process group is up (N > 1, or a test that brought one up), else None."""
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def backend_in_use(world: int) -> str:
    """Collective backend the timing protocol actually runs on ("none" for one process)."""
    dist = _dist()
    if dist is not None:
        return str(dist.get_backend())
    return "none" if world <= 1 else "uninitialised"


def init_from_env(backend: str, device: torch.device | None = None) -> Tuple[int, int, int]:
    """-> (rank, world, local_rank); initialises torch.distributed when WORLD_SIZE > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if not dist.is_initialized():
            kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
            try:
                dist.init_process_group(backend, rank=rank, world_size=world, **kw)
            except Exception as e:  # noqa: BLE001
                # The replicas exchange nothing but the timing protocol (a barrier and one
                # 8-byte max-reduce): if RCCL cannot come up on this node, gloo over the host
                # carries it just as well and the measurement is unaffected.
                if backend != "nccl":
                    raise
                print(f"[replicas] RCCL init failed ({e!r}); timing protocol falls back to gloo",
                      flush=True)
                dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, world, local_rank


def timed_replica_run(run: Callable[[], float | None], steps: int, world: int,
                      device: torch.device, sync: Callable[[], None]) -> Tuple[float, float]:
    """Time `run()` (one replica's K steps) bracketed by barrier + device sync on both sides;
    returns (max wall seconds over ranks, aggregate steps/s = world*steps / max wall)."""
    import time
    dist = _dist()  # None for a single process without a group
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    run()
    sync()
    t1 = time.perf_counter()
    if dist is not None:
        dist.barrier()
    wall = t1 - t0
    if dist is not None:
        red_dev = device if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([wall], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    return wall, world * steps / wall


def gather_per_replica(value: float, world: int, device: torch.device) -> list:
    """Every rank's `value` (its own wall seconds, tok/s ...), in rank order, on every rank.
    Measurement plumbing only - the data path exchanges nothing."""
    dist = _dist()
    if dist is None:
        return [float(value)]
    red_dev = device if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor([value], dtype=torch.float64, device=red_dev)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def spread(per_replica: list) -> dict:
    """min / max / relative spread of per-replica throughputs."""
    lo, hi = min(per_replica), max(per_replica)
    return {"per_replica": [round(v, 3) for v in per_replica], "min": lo, "max": hi,
            "spread_frac": (hi - lo) / hi if hi > 0 else 0.0}


def shutdown(world: int) -> None:
    dist = _dist()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
