"""Multi-GPU plumbing for batched throughput (BASELINE.json configs[4]): one process per GPU, one
independent frame-pair stream per rank, no collective on the data path.  The only exchange is the
gather of the per-step pose records (a few hundred bytes per rank and step) -- RCCL over xGMI when the
backend is "nccl", gloo on CPU for the tests.  The payload is latency-bound, so the records of a whole
run are gathered with ONE collective at the end instead of one per step.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

RECORD_WIDTH = 16      # stream, step, valid, n_inliers, rvec[3], tvec[3], t_prev_curr[3], pad[3]


def stream_seed(base_seed: int, rank: int) -> int:
    """Seed of the stream owned by `rank` (20250910 + stream)."""
    return base_seed + rank


def init(backend: str, device_index: int | None = None) -> tuple[int, int]:
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {}
        if backend == "nccl" and device_index is not None:
            kw["device_id"] = torch.device("cuda", device_index)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def make_record(stream: int, step: int, res) -> torch.Tensor:
    r = torch.zeros(RECORD_WIDTH, dtype=torch.float64)
    r[0], r[1], r[2], r[3] = stream, step, res.valid, res.n_inliers
    r[4:7] = torch.tensor(list(res.rvec), dtype=torch.float64)
    r[7:10] = torch.tensor(list(res.tvec), dtype=torch.float64)
    r[10:13] = torch.tensor(list(res.t_prev_curr), dtype=torch.float64)
    return r


def fill_record(records, i: int, stream: int, step: int, res) -> None:
    """The same record written into row i of a preallocated [steps, RECORD_WIDTH] float64 numpy array (a few microseconds:
    the timed loop of bench.py runs at ~3000 steps/s on one host thread)."""
    row = records[i]
    row[0] = stream; row[1] = step; row[2] = res.valid; row[3] = res.n_inliers
    row[4:7] = res.rvec
    row[7:10] = res.tvec
    row[10:13] = res.t_prev_curr


def gather_records(records: torch.Tensor, device: torch.device | None = None) -> torch.Tensor:
    """[steps, RECORD_WIDTH] per rank -> [world, steps, RECORD_WIDTH] on every rank (one all-gather)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return records.unsqueeze(0)
    mine = records.to(device) if device is not None else records
    world = dist.get_world_size()
    mine = mine.contiguous()
    out = torch.empty((world * mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(out, mine)          # concatenated along dim 0 in rank order
    return out.view((world,) + tuple(mine.shape))


def max_over_ranks(seconds: float, device: torch.device | None = None) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
