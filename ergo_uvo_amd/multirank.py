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


def _parse_cpulist(text: str) -> list[int]:
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def _by_physical_core(cpus: list[int], sys_root: str = "/sys") -> list[int]:
    """Hardware threads ordered so that the siblings of one physical core are adjacent (cpu0, cpu128, cpu1, cpu129, ... on a
    64-core SMT-2 socket): a slice of the list is then a set of WHOLE cores, and two ranks never share one."""
    def first_sibling(c):
        try:
            return min(_parse_cpulist(open(os.path.join(sys_root, "devices/system/cpu", f"cpu{c}", "topology/thread_siblings_list")).read()))
        except (OSError, ValueError):
            return c
    return sorted(cpus, key=lambda c: (first_sibling(c), c))


def visible_gpus(sys_root: str = "/sys", dev_root: str = "/dev", env=None) -> list[dict]:
    """The GPUs this process's HIP runtime will enumerate, in its order, WITHOUT touching the GPU: KFD topology nodes with SIMDs
    (`/sys/class/kfd/kfd/topology/nodes/*/properties`) whose render node can be opened, filtered by ROCR_VISIBLE_DEVICES and then
    HIP_VISIBLE_DEVICES when those are plain index lists.  Each entry: {"pci": "dddd:bb:dd.f", "numa_node": int | None,
    "cpus": [...]} from the PCI device's `numa_node` / `local_cpulist`.  Empty when the topology cannot be read (no amdgpu)."""
    env = os.environ if env is None else env
    base = os.path.join(sys_root, "class/kfd/kfd/topology/nodes")
    try:
        nodes = sorted((int(n) for n in os.listdir(base) if n.isdigit()))
    except OSError:
        return []
    gpus = []
    for n in nodes:
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(base, str(n), "properties")) if len(line.split()) >= 2)
        except OSError:
            continue                                     # a node of another container's GPU: not readable, not ours
        if int(props.get("simd_count", "0")) == 0:
            continue                                     # a CPU node
        minor = int(props.get("drm_render_minor", "-1"))
        if minor >= 0 and not os.access(os.path.join(dev_root, "dri", f"renderD{minor}"), os.R_OK | os.W_OK):
            continue                                     # not handed to this container: ROCr skips it too
        loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
        pci = f"{dom:04x}:{(loc >> 8) & 0xFF:02x}:{(loc >> 3) & 0x1F:02x}.{loc & 7:x}"
        g = {"pci": pci, "numa_node": None, "cpus": [], "kfd_node": n}
        try:
            g["numa_node"] = int(open(os.path.join(sys_root, "bus/pci/devices", pci, "numa_node")).read())
            g["cpus"] = _by_physical_core(_parse_cpulist(open(os.path.join(sys_root, "bus/pci/devices", pci, "local_cpulist")).read()), sys_root)
        except (OSError, ValueError):
            pass
        gpus.append(g)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES" if env.get("HIP_VISIBLE_DEVICES") not in (None, "") else "CUDA_VISIBLE_DEVICES"):     # HIP honours one of the two, HIP_VISIBLE_DEVICES first
        v = env.get(var)
        if v is None or v == "":
            continue
        try:
            idx = [int(x) for x in v.split(",") if x.strip() != ""]
        except ValueError:
            return []                                    # UUID forms: do not guess
        gpus = [gpus[i] for i in idx if 0 <= i < len(gpus)]
    return gpus


def cpu_quota(root: str = "/sys/fs/cgroup") -> float | None:
    """CPUs' worth of time this container may use per scheduling period (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`);
    None when unlimited or unreadable.  A process group that runs more busy threads than that is THROTTLED -- every thread stopped
    until the period ends: measured as two 8-9 ms holes in a 4200-pair bench run on a 256-thread host with a 16-CPU quota."""
    try:
        with open(os.path.join(root, "cpu.max")) as f:
            q, p = f.read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open(os.path.join(root, "cpu", "cpu.cfs_quota_us")) as f:
            q = float(f.read())
        with open(os.path.join(root, "cpu", "cpu.cfs_period_us")) as f:
            p = float(f.read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def _within_quota(cores: list[int], local_world: int, quota: float | None) -> list[int]:
    """The rank's cores cut to its share of the container's CPU quota (never below two): threads confined to that many CPUs cannot
    run the container into the throttle."""
    if quota is None:
        return cores
    share = max(2, int(quota // max(local_world, 1)))
    return cores[:share] if len(cores) > share else cores


def pin_rank(local_rank: int, local_world: int, share_devices: bool = False, gpus: list[dict] | None = None, quota: float | None | str = "auto",
             cut_to_quota: bool = False) -> dict:
    """Pin this rank -- before its first GPU call, so that the HIP runtime's helper threads and the pipeline's lane workers inherit
    it -- to the host cores NEXT TO ITS GPU: the `local_cpulist` of the device LOCAL_RANK will open (the NUMA node the card hangs
    off, `/sys/bus/pci/devices/<addr>/numa_node`); ranks whose GPUs share a node split that node's cores evenly, in rank order.
    Falls back to an index slice of the allowed cores (rank r of n: cores [r*k, (r+1)*k)) when the topology cannot be read.
    Returns {"cores", "source", "numa_node", "pci", "cpu_quota", "quota_share"}.  A rank runs 1 submitting thread + `depth` lane workers,
    all polling while pairs are in flight.  `cpu_quota` (cgroup) / local_world = `quota_share` is reported, and
    the cores are cut to it only on request: on a loaded host a rank confined to as many CPUs as it has busy threads was preempted for
    whole time slices (gaps of 2-5 ms in one 600-pair run of three, none in three with the whole NUMA node to move about in); staying
    under the quota is a matter of how many threads are busy, not of where they may run."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return {"cores": [], "source": "unsupported", "numa_node": None, "pci": None}
    if quota == "auto":
        quota = cpu_quota()
    local_world = max(local_world, 1)
    share = None if quota is None else quota / local_world
    out = {"cores": allowed, "source": "unpinned", "numa_node": None, "pci": None, "cpu_quota": quota, "quota_share": share}
    if not cut_to_quota:
        quota_cut = None
    else:
        quota_cut = quota
    gpus = visible_gpus() if gpus is None else gpus
    if gpus and (share_devices or local_rank < len(gpus)):
        mine = gpus[local_rank % len(gpus)]
        near = [c for c in mine["cpus"] if c in set(allowed)]
        # the ranks that will sit on the same NUMA node (their devices are LOCAL_RANK modulo the device count), in rank order
        same = [r for r in range(local_world) if (share_devices or r < len(gpus)) and gpus[r % len(gpus)]["numa_node"] == mine["numa_node"]]
        k = len(near) // max(len(same), 1)
        if near and k >= 1 and local_rank in same:
            j = same.index(local_rank)
            cores = _within_quota(near[j * k:(j + 1) * k], local_world, quota_cut)
            os.sched_setaffinity(0, set(cores))
            return {"cores": cores, "source": "numa node of the GPU (kfd topology + pci local_cpulist)" + ("" if quota_cut is None else ", cut to the rank's share of the cgroup CPU quota"),
                    "numa_node": mine["numa_node"], "pci": mine["pci"], "cpu_quota": quota, "quota_share": share}
    k = len(allowed) // local_world
    if local_world > 1 and k >= 1:
        cores = _within_quota(allowed[(local_rank % local_world) * k:(local_rank % local_world + 1) * k], local_world, quota_cut)
        os.sched_setaffinity(0, set(cores))
        out.update(cores=cores, source="index slice of the allowed cores (GPU topology not readable)")
    elif quota_cut is not None and len(allowed) > max(2, int(quota_cut)):
        cores = _within_quota(allowed, 1, quota_cut)
        os.sched_setaffinity(0, set(cores))
        out.update(cores=cores, source="the first allowed cores, as many as the cgroup CPU quota")
    return out


def pin_matches_device(pin: dict, props) -> bool | None:
    """After the GPU is initialised: is the device this rank opened the one pin_rank assumed (PCI address)?  None = nothing to compare."""
    if not pin.get("pci"):
        return None
    try:
        got = f"{int(getattr(props, 'pci_domain_id')):04x}:{int(getattr(props, 'pci_bus_id')):02x}:{int(getattr(props, 'pci_device_id')):02x}"
    except (AttributeError, TypeError, ValueError):
        return None
    return pin["pci"].startswith(got)


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
