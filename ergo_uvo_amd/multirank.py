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


def pin_rank_to_cores(local_rank: int, local_world: int) -> list[int]:
    """Give this rank its own slice of the host's cores (call before the first GPU call, so the HIP runtime's helper threads and
    the pipeline's lane workers inherit it): rank r of n gets cores [r*k, (r+1)*k) of the sorted set this process may run on,
    k = cores // n.  Returns the cores now in force (unchanged when there are fewer cores than ranks, or on platforms without
    sched_setaffinity).  A rank runs 1 submitting thread (polls) + `depth` lane workers, of which at most max_b (3) poll at a
    time; unpinned, the ranks' pollers migrate over each other's cores."""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return []
    k = len(cores) // max(local_world, 1)
    if local_world > 1 and k >= 1:
        mine = cores[local_rank * k:(local_rank + 1) * k]
        os.sched_setaffinity(0, set(mine))
        return mine
    return cores


def init(backend: str, device_index: int | None = None, force: bool = False) -> tuple[int, int]:
    """One process per GPU.  The process group exists only when there is more than one rank -- or when `force` asks for it, so
    that a single GPU can run the very collectives of the N-rank path (RCCL communicator of one rank: `bench.py --force-dist`)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world > 1 or force) and not dist.is_initialized():
        kw = {}
        if backend == "nccl" and device_index is not None:
            kw["device_id"] = torch.device("cuda", device_index)
        rdzv = os.environ.get("UVO_RDZV_FILE")                 # bench.py's own launcher: a file store, no port to guess
        if rdzv:
            kw["init_method"] = "file://" + rdzv
        else:                                                  # torchrun / the driver: MASTER_ADDR / MASTER_PORT from the environment
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
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


def records_from_results(results, stream: int, first_step: int = 0):
    """A ctypes array of StereoResult (filled in place by Context.stereo_collect(out=...)) -> the [steps, RECORD_WIDTH] float64
    record matrix of fill_record, in one vectorised pass (the timed loop of bench.py then does no per-step conversion)."""
    import ctypes
    import numpy as np
    n = len(results)
    dt = np.dtype([("valid", "i4"), ("initialized", "i4"), ("n_left", "i4"), ("n_right", "i4"), ("n_stereo_matches", "i4"),
                   ("n_tri_matches", "i4"), ("n_good3d", "i4"), ("n_inliers", "i4"), ("rvec", "f8", 3), ("tvec", "f8", 3),
                   ("t_prev_curr", "f8", 3), ("velocity", "f8", 3)])
    assert dt.itemsize == ctypes.sizeof(results) // max(n, 1)
    a = np.frombuffer(results, dtype=dt, count=n)
    rec = np.zeros((n, RECORD_WIDTH), dtype=np.float64)
    rec[:, 0] = stream
    rec[:, 1] = np.arange(first_step, first_step + n)
    rec[:, 2] = a["valid"]; rec[:, 3] = a["n_inliers"]
    rec[:, 4:7] = a["rvec"]; rec[:, 7:10] = a["tvec"]; rec[:, 10:13] = a["t_prev_curr"]
    return rec


def gather_records(records: torch.Tensor, device: torch.device | None = None) -> torch.Tensor:
    """[steps, RECORD_WIDTH] per rank -> [world, steps, RECORD_WIDTH] on every rank (one all-gather)."""
    if not dist.is_initialized():
        return records.unsqueeze(0)
    mine = records.to(device) if device is not None else records
    world = dist.get_world_size()
    mine = mine.contiguous()
    out = torch.empty((world * mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(out, mine)          # concatenated along dim 0 in rank order
    return out.view((world,) + tuple(mine.shape))


def max_over_ranks(seconds: float, device: torch.device | None = None) -> float:
    if not dist.is_initialized():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_ints(values: list[int], device: torch.device | None = None) -> torch.Tensor:
    """[k] ints per rank -> [world, k] on every rank (rank 0 checks that it sees N distinct ranks on N distinct devices)."""
    mine = torch.tensor(values, dtype=torch.int64, device=device)
    if not dist.is_initialized():
        return mine.unsqueeze(0).cpu()
    out = torch.empty((dist.get_world_size() * mine.numel(),), dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(out, mine)
    return out.view(dist.get_world_size(), -1).cpu()


def barrier():
    """A process group of one rank (force) still runs the collective."""
    if dist.is_initialized():
        dist.barrier()
