#!/usr/bin/env python3
"""bench.py -- stereo frame-pairs/sec (detect + match + pose) of the HIP hot path.

  python bench.py --gpus N --steps K --warmup W
  N > 1 without a launcher (WORLD_SIZE unset): this process starts N child ranks itself, before it touches the GPU, and
  relays rank 0's JSON line; under torchrun (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are taken from the environment.

One step = one iteration body of stereo_VO (visual_odometry.h:531-739, minus get_image / decode /
publish): 2x SURF detect+describe, L<->R match, prev<->curr match, gathers, triangulation,
extract_3Dpoints, EPnP PnP-RANSAC + refit, pose inversion, on one synthetic 1920x1080 pair with
~3000 keypoints per image (BASELINE.json configs[2], "C3"); images are resident in HBM before the
timed region.  With N ranks every rank runs its own independent stream (configs[4]; weak scaling)
and the per-step pose records are all-gathered over RCCL at the end of the timed region.

The timed region of the contract -- fence, exactly K steps, gather, fence -- is run `--blocks` (7) times in a row in the same
process; `value` / `ms_per_step` are the MEDIAN block's (MAX over ranks per block), the first / slowest / fastest block and every
block's rate are listed beside it, with the gaps between consecutive collects (`collect_gap_ms`) and, for K <= 64, the device and
host timestamps of every pair's phases (`pipeline_trace`), so that a stall inside one 10 ms window shows where it was.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant
kernel (k_hessian_nms_all: algorithmic bytes of SURVEY.md 8(d) / HIP-event time measured here)
and `cpu_baseline` (the CPU oracle, 1 thread, on a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import json
import os

# one hardware queue per HIP stream of the pipeline lanes (the ROCm default of 4 makes lanes share queues and
# serialises them); must be set before the HIP runtime starts
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# numpy / torch size their thread pools by the host's CPU count (256 on the GPU boxes) and their idle threads spin for a while after
# every parallel region: nothing here needs them, and on a host with a cgroup CPU quota they run the container into the throttle
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "4")
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT = 1920, 1080
MIN_HESSIAN_C3 = 6387            # frozen: frame 0 of seed 20250906 gives 3001 / 3008 keypoints
MFMA_F32_PEAK_TFLOPS = 157.3    # dense f32-input MFMA, MI355X_MICROARCH.md
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes_hessian_launch(w: int, h: int, nimg: int, octaves=(0, 1, 2, 3)) -> int:
    """SURVEY.md 8(d) B_det restricted to what the dominant detection launch produces.  Since round 2 that launch
    (k_hessian_nms_all) evaluates all four octaves; per octave o of an image the model charges one read of the integral
    image (4(W+1)(H+1): "read once per octave") + det and trace written (2*S_o) + det read by the NMS (S_o), with
    S_o = 3 layers * 4 B * (H >> o)(W >> o) -- the launch evaluates the three middle layers of an octave for every sample; the
    two outer layers are only evaluated around the few thousand NMS survivors by k_hessian_finish, so their 2/5 of the
    contract's five-layer figure is NOT credited to this kernel."""
    per_image = sum(4 * (w + 1) * (h + 1) + 3 * (3 * 4 * (h >> o) * (w >> o)) for o in octaves)
    return nimg * per_image


def algorithmic_bytes_pair(w: int, h: int, n_kp: int, b_desc_pair=None) -> float:
    """SURVEY.md 8(d) B_pair = 2*B_det + 2*B_desc + 2*B_match.  b_desc_pair: the descriptor term of both images from the run's
    actual windows (the synthetic scene's windows are large: ~185 MB per pair); without it SURVEY's ~7.5 MB per image midpoint."""
    s = sum(5 * 4 * (h >> o) * (w >> o) for o in range(4))
    b_det = w * h + 4 * (w + 1) * (h + 1) + 4 * 4 * (w + 1) * (h + 1) + 3 * s
    b_match = (n_kp + n_kp) * 256 + n_kp * 16
    return 2 * b_det + (2 * 7.5e6 if b_desc_pair is None else b_desc_pair) + 2 * b_match


def ping_pong(n_frames: int):
    k, d = 0, 1
    while True:
        yield k
        if n_frames == 1:
            continue
        if k + d < 0 or k + d >= n_frames:
            d = -d
        k += d


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without torchrun: start N fresh child ranks (one process per GPU) with RANK / LOCAL_RANK /
    WORLD_SIZE set.  The parent has made no GPU call and never execs; rank 0's child prints the JSON line on the inherited
    stdout.  The ranks meet through a file store in a private temporary directory (UVO_RDZV_FILE -> init_method file://...), so
    no TCP port has to be guessed.  Returns the worst child exit code."""
    import subprocess
    import tempfile
    import time as _time
    rc = 0
    with tempfile.TemporaryDirectory(prefix="uvo_rdzv_") as tmp:
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       UVO_RDZV_FILE=os.path.join(tmp, "store"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=(_REAL_STDOUT or None) if r == 0 else subprocess.DEVNULL))      # rank 0 prints the line
        # a rank that dies before the rendezvous would leave the others waiting for it: when one fails, the rest are ended (by PID)
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                rc = max(rc, abs(code))
                if code != 0:
                    for q in live:
                        q.terminate()
            _time.sleep(0.05)
    return rc


def measure_hbm_traffic(pattern: str = "k_hessian_nms_all", steps: int = 6, timeout_s: float = 300.0):      # (a fresh box pages torch in for 1-2 minutes: the first child pays that instead of this process)
    """HBM bytes per launch of the roofline kernel FROM THE COUNTERS, in this run: two `rocprofv3 --kernel-trace --pmc <counter>` child
    passes (FETCH_SIZE, then WRITE_SIZE -- separate runs, and the gfx950 reading of MI355X_MICROARCH.md: both count KiB, FETCH_SIZE is
    doubled) of tools/prof_stereo.py, the same C3 stereo step the roofline leg times.  Must be called BEFORE this process touches the
    GPU (the children are ordinary child processes; nothing here execs).  -> (bytes per launch or None, details)."""
    import csv, glob, shutil, signal, subprocess, tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, {"skipped": "rocprofv3 not on PATH"}
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_CTOR")):
        return None, {"skipped": "this process runs under a profiler already"}
    per_counter, details = {}, {"command": "rocprofv3 --kernel-trace --pmc <counter> --output-format csv -- python3 tools/prof_stereo.py %d" % steps}
    tmp = tempfile.mkdtemp(prefix="uvo_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.join(ROOT, "tools", "prof_stereo.py"), str(steps)]
            p = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = p.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)                   # the group this call started, nothing else
                p.wait()
                return None, dict(details, skipped=f"the {counter} pass did not finish in {timeout_s:.0f} s")
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if rc != 0 or not files:
                return None, dict(details, skipped=f"the {counter} pass failed (rc {rc})")
            per, order = {}, []
            for r in csv.DictReader(open(files[0])):
                if pattern not in r["Kernel_Name"] or r["Counter_Name"] != counter:
                    continue
                k = r["Dispatch_Id"]
                if k not in per:
                    per[k] = 0.0; order.append(k)
                per[k] += float(r["Counter_Value"])                # one row per XCD: the launch's total is their sum
            vals = [per[k] for k in order[2:]]                      # the first launches of a context warm caches and clocks
            if not vals:
                return None, dict(details, skipped=f"no {pattern} launch in the {counter} pass")
            per_counter[counter] = sum(vals) / len(vals)
            details[counter + "_KiB_per_launch"] = round(per_counter[counter], 1)
            details["launches_averaged"] = len(vals)
    except Exception as e:                                          # a counter pass is evidence, never a reason for the bench to fail
        return None, dict(details, skipped=f"{type(e).__name__}: {e}")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    nbytes = int((2.0 * per_counter["FETCH_SIZE"] + per_counter["WRITE_SIZE"]) * 1024.0)
    details["how"] = "(2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, counters summed over the XCDs, separate passes, measured in this run before the timed region"
    return nbytes, details


def summarise_trace(tr) -> dict:
    """uvo_trace_read's rows (every pipelined pair of the timed blocks) -> the figures that locate a stall: device durations of a
    pair's stage A (detect .. extract_3Dpoints) and PnP stage, the device-side gap between them, the cadence of stage-A ends, and
    on the host the submitting thread's pacing wait and call time and the lane worker's wait for a PnP slot and stage-B time."""
    def stats(v, idx=None):
        v = np.asarray(v, np.float64)
        if v.size == 0:
            return None
        out = {"p50": round(float(np.median(v)), 4), "max": round(float(v.max()), 4)}
        if idx is not None:
            out["argmax_pair"] = int(idx[int(v.argmax())])
        return out
    dev, host, pair = tr["dev_ms"].astype(np.float64), tr["host_ms"], tr["pair"]
    b = tr["b_used"] != 0
    out = {"pairs": int(len(tr)),
           "dev_stage_a_ms": stats(dev[:, 2] - dev[:, 0], pair), "dev_detector_ms": stats(dev[:, 1] - dev[:, 0], pair),
           "dev_pnp_stage_ms": stats(dev[b, 5] - dev[b, 3], pair[b]), "dev_a_end_to_pnp_begin_ms": stats(dev[b, 3] - dev[b, 2], pair[b]),
           "dev_pair_latency_ms": stats(dev[b, 5] - dev[b, 0], pair[b]),
           "dev_a_end_cadence_ms": stats(np.diff(dev[:, 2]), pair[1:]),
           "dev_detection_launch_ms": (stats((dev[:, 7] - dev[:, 6])[dev[:, 6] >= 0], pair[dev[:, 6] >= 0]) if dev.shape[1] > 7 else None),
           "host_pacing_wait_ms": stats(host[:, 1] - host[:, 0], pair), "host_submit_call_ms": stats(host[:, 2] - host[:, 0], pair),
           "host_pnp_slot_wait_ms": stats(host[b, 4] - host[b, 3], pair[b]), "host_stage_b_ms": stats(host[b, 5] - host[b, 4], pair[b]),
           "per_lane_p50_ms": {int(l): {"stage_a": round(float(np.median((dev[:, 2] - dev[:, 0])[tr["lane"] == l])), 4),
                                        "pnp_stage": (round(float(np.median((dev[:, 5] - dev[:, 3])[(tr["lane"] == l) & b])), 4) if ((tr["lane"] == l) & b).any() else None)}
                               for l in sorted(set(tr["lane"].tolist()))},
           "what": "HIP events on the lanes' streams / steady clock on the host, every pair of the timed blocks (include/uvo_hip.h: uvo_trace_row)"}
    return out


def hbm_copy_peak_gbs(torch, seconds: float = 0.05, nbytes: int = 1 << 30) -> float:
    """Device-to-device copy rate of this GPU, read + write bytes per second ("peak from the box at run time"):
    a 1 GiB buffer copied back and forth for ~`seconds`, timed with events on the stream the copies run on."""
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    b.copy_(a); a.copy_(b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps, ms = 4, 0.0
    while True:
        e0.record()
        for _ in range(reps):
            b.copy_(a)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        if ms >= seconds * 1e3 or reps >= 4096:
            break
        reps *= 2
    del a, b
    torch.cuda.empty_cache()
    return 2.0 * nbytes * reps / (ms * 1e-3) / 1e9


_REAL_STDOUT = None


def emit(line: str) -> None:
    """The contract's one JSON line, on the process's real standard output."""
    out = _REAL_STDOUT or sys.stdout
    out.write(line + "\n")
    out.flush()


def claim_stdout() -> None:
    """Standard output carries the JSON line and nothing else: everything a library prints there (RCCL's version block at communicator
    creation, runtime notices) goes to standard error for the rest of the run, the line itself to a duplicate of the original descriptor.
    (Run as a script only: tests import this module.)"""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=4, help="distinct synthetic stereo pairs (ping-pong order)")
    ap.add_argument("--depth", type=int, default=int(os.environ.get("UVO_PIPELINE_DEPTH", "6")),
                    help="consecutive pairs in flight per image stream (uvo_stereo_set_depth)")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("UVO_BENCH_BATCH", "1")), choices=[1, 2],
                    help="pairs per launch set (uvo_stereo_set_batch): 2 = consecutive pairs of the stream are queued two at a time")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU-oracle baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the two rocprofv3 counter passes behind roofline.traffic (rank 0 at N = 1 runs them "
                    "before its first GPU call, ~40 s; without them the figure of the newest matching committed counter summary is reported)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="collective backend of the pose-record gather "
                    "(nccl = RCCL; gloo only for rehearsing N ranks on fewer devices)")
    ap.add_argument("--share-devices", action="store_true", help="rehearsal: ranks may share a device (LOCAL_RANK modulo the device count)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group even with one rank, so that --gpus 1 "
                    "runs the N-rank path's collectives (RCCL communicator of one rank)")
    ap.add_argument("--no-pin", action="store_true", help="do not pin the rank to the cores next to its GPU")
    ap.add_argument("--cpus", type=int, default=0, help="measurement: confine this rank to N logical CPUs (sched_setaffinity before the first GPU "
                    "call, whole physical cores next to the GPU first) and tell the library that N is its CPU budget")
    ap.add_argument("--blocks", type=int, default=7, help="the timed region (K steps between two fences) is run this many times in a row; "
                    "`value` is the median block, the first / slowest / fastest are reported beside it")
    ap.add_argument("--no-trace", action="store_true", help="do not record the per-pair pipeline trace (for --steps <= 64 one more, untimed-for-value block runs with it)")
    ap.add_argument("--trace-all", action="store_true", help="trace every timed block instead (costs the 20-step form ~2.4 %%)")
    ap.add_argument("--timed-only", action="store_true", help="diagnostics: stop after the timed region (no latency / roofline / CPU legs), "
                    "so that a UVO_TRACE file holds the timed pairs")
    ap.add_argument("--dump-records", default=None, help="rank 0 writes the gathered [world, steps, 16] pose records to this .npy")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    # roofline.traffic from the counters, measured in this run: child passes, before this process touches the GPU (rank 0 at N = 1, the
    # run that also carries the CPU baseline)
    pmc_traffic, pmc_details = None, {"skipped": "not requested for this form (--no-pmc / --no-cpu-baseline / --timed-only / N > 1)"}
    if args.gpus == 1 and "WORLD_SIZE" not in os.environ and not (args.no_pmc or args.no_cpu_baseline or args.timed_only or args.force_dist):
        pmc_traffic, pmc_details = measure_hbm_traffic()

    import torch
    import torch.distributed as dist
    import ergo_uvo_amd as uvo
    from ergo_uvo_amd import synth, multirank

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # host budget of a rank: its own slice of the cores, taken before the first GPU call so that the HIP runtime's threads and the
    # lane workers inherit it (one polling submitter + <= max_b polling PnP workers per rank; the other workers sleep on events)
    host_cores_all = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    # UVO_BENCH_PIN=quota: the rank's cores cut to its share of the cgroup CPU quota (measurement: gaps of 2-5 ms on a loaded host)
    pin = {"cores": [], "source": "off"} if args.no_pin else multirank.pin_rank(local_rank, local_world, share_devices=args.share_devices,
                                                                                cut_to_quota=os.environ.get("UVO_BENCH_PIN") == "quota")
    # The rank's CPU budget = its share of the container's CPU quota (or --cpus): handed to the library (UVO_CPU_BUDGET), which picks
    # how its host side waits and who drives the PnP stage of pipelined pairs (include/uvo_hip.h: uvo_ctx_host_policy; measured per
    # budget in profiles/r05_host_budget.json).  With a CPU per lane the lane workers poll; below depth + 3 CPUs the first RANSAC round
    # runs device-driven and the rank keeps ONE host thread busy -- eight ranks on a 16-CPU quota get 2 CPUs each.
    if args.cpus > 0:
        cores = (pin["cores"] or sorted(os.sched_getaffinity(0)))[:args.cpus]
        os.sched_setaffinity(0, set(cores))
        pin = dict(pin, cores=cores, source=(pin.get("source") or "") + f", cut to --cpus {args.cpus}")
        os.environ["UVO_CPU_BUDGET"] = str(args.cpus)
    elif pin.get("quota_share") is not None:
        os.environ.setdefault("UVO_CPU_BUDGET", f"{pin['quota_share']:.3f}")
    my_cores = pin["cores"]
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if args.share_devices:
        local_rank %= n_dev
    elif local_rank >= n_dev:
        raise SystemExit(f"rank with LOCAL_RANK={local_rank} but only {n_dev} device(s) visible (--share-devices rehearses N ranks on fewer)")
    torch.cuda.set_device(local_rank)
    rank, world = multirank.init(args.backend, local_rank, force=args.force_dist)    # "nccl" is RCCL on ROCm
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dist_on = dist.is_initialized()
    dev = torch.device("cuda", local_rank) if (args.backend == "nccl" and dist_on) else None     # where the gathered records travel from
    # every rank must be a different rank on (unless rehearsing) a different device: gathered once, checked by rank 0
    props = torch.cuda.get_device_properties(local_rank)
    import zlib
    dev_id = f"{getattr(props, 'uuid', '')}|{getattr(props, 'pci_domain_id', '')}:{getattr(props, 'pci_bus_id', '')}:{getattr(props, 'pci_device_id', '')}"
    dev_key = zlib.crc32(dev_id.encode())                       # stable across processes (Python's hash() is salted per process)
    pin["pci_match"] = multirank.pin_matches_device(pin, props)
    who = multirank.gather_ints([rank, local_rank, dev_key, os.getpid(), pin.get("numa_node", -1) if pin.get("numa_node") is not None else -1, len(my_cores)], dev)
    if rank == 0:
        assert sorted(who[:, 0].tolist()) == list(range(world)), f"ranks seen: {who[:, 0].tolist()}"
        assert len(set(who[:, 3].tolist())) == world, "ranks share a process"
        if not args.share_devices:
            assert len(set(who[:, 1].tolist())) == world, f"ranks share a device: local ranks {who[:, 1].tolist()}"
            assert len(set(who[:, 2].tolist())) == world, f"ranks share a device: device keys {who[:, 2].tolist()} (uuid / PCI address of each rank's cuda device)"

    # ---- synthetic workload (seeded; one independent stream per rank) ----
    seed = synth.SEEDS["C3"] if world == 1 else multirank.stream_seed(synth.SEEDS["C5"], rank)
    min_hessian = MIN_HESSIAN_C3
    scene = synth.Scene(seed, WIDTH)
    host_frames = [synth.stereo_pair(scene, k, WIDTH, HEIGHT) for k in range(args.frames)]
    dev_frames = [(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()) for L, R in host_frames]
    rig = synth.stereo_rig(WIDTH)
    params = uvo.Params.stereo(SURF_MIN_HESSIAN=min_hessian)
    ctx = uvo.Context(params, local_rank, WIDTH, HEIGHT, 8192)
    ctx.stereo_set_depth(args.depth)
    ctx.stereo_set_batch(args.batch)
    host_policy = ctx.host_policy()
    ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)

    order = ping_pong(args.frames)

    def step():
        k = next(order)
        ks_seen.append(k)
        L, R = dev_frames[k]
        return ctx.stereo_step(L, R, 0.05)

    ks_seen = []                                   # frame index of every pair handed to the VO loop, in order

    def submit():
        k = next(order)
        ks_seen.append(k)
        L, R = dev_frames[k]
        ctx.stereo_submit(L, R)

    # warm-up: the first step is consumed by the VO init phase (synchronous by definition); the remaining ones go through the
    # same submit/collect pipeline as the timed region, so every lane's buffers, streams and worker thread have been used
    # device-to-device copy rate of this GPU (roofline.peak_measured), before the warm-up (placing it here rather than at process
    # start was measured not to matter for a 20-step run: 3500 vs 3515 pairs/s -- the clocks are not what a short run waits for)
    hbm_measured = hbm_copy_peak_gbs(torch)
    import gc

    def fence():
        torch.cuda.synchronize()
        multirank.barrier()
        torch.cuda.synchronize()

    def timed_block(n_steps, results, stamps):
        """The timed region of the contract: fence, exactly n_steps pairs through the submit/collect pipeline (one image stream,
        `depth` consecutive pairs in flight on separate lanes -- own buffers, HIP streams and PnP worker thread each; a pair only
        waits for the previous pair's "after stereo match" set; every result is identical to the synchronous uvo_stereo_step's,
        tests/test_gpu_parity.py), the records of all streams gathered (one RCCL all-gather when N > 1), fence.  stamps[i] = host
        clock right after the i-th collect returned (stamps[-1] = region start).  -> (seconds on this rank, records, gathered)."""
        fence()
        t0 = time.perf_counter()
        submitted = 0
        for i in range(n_steps):
            while submitted < n_steps and submitted - i < args.depth:
                submit(); submitted += 1
            ctx.stereo_collect(0.05, out=results[i])
            stamps[i] = time.perf_counter()
        records = multirank.records_from_results(results, rank)            # [steps, 16] float64, one vectorised pass
        allrec = multirank.gather_records(torch.from_numpy(records), dev)  # pose records of all streams
        fence()
        dt_local = time.perf_counter() - t0
        stamps[n_steps] = t0
        return dt_local, records, allrec

    # warm-up: the first pairs are consumed by the VO init phase (synchronous by definition); the remaining ones go through the very
    # function the timed blocks run -- pipeline, record conversion, gather, fences -- so that every lane (buffers, streams, worker
    # thread) and every host code path of the timed region has run before the clock starts.  At least 2 x depth pipelined pairs,
    # whatever --warmup says (with --warmup 5 and six lanes, three lanes used to meet their first pair inside the timed region).
    n_warm = max(args.warmup, 2)
    for _ in range(2):
        r = step()
    n_pipe_warm = max(n_warm - 2, 2 * args.depth)
    warm_results = (uvo.StereoResult * n_pipe_warm)()
    warm_stamps = [0.0] * (n_pipe_warm + 1)
    timed_block(n_pipe_warm, warm_results, warm_stamps)
    multirank.max_over_ranks(0.0, dev)
    del warm_results

    # The per-pair pipeline trace (HIP timing events on the lanes' streams + host stamps) costs the submitting thread ~6 us per pair --
    # 2.4 % of the driver's 20-step form (3830 with it, 3920 without, same box) -- so the timed blocks run WITHOUT it (their stalls, if
    # any, are located by the per-collect host stamps, collect_gap_ms) and one more block of the same shape, not part of `value`, runs
    # with it for runs of <= 64 steps (--trace-all: every block traced, as round 4's first bench lines were)
    trace_on = (not args.no_trace) and args.steps <= 64 and not args.timed_only
    if trace_on and args.trace_all:
        ctx.trace_enable(True)
    n_blocks = max(args.blocks, 1)
    block_dt, block_local, block_cpu, block_stamps = [], [], [], []
    all_results, all_records, all_allrec, block_first_k = [], [], [], []
    gc.collect(); gc.freeze(); gc.disable()                                # no collector pause inside a 10 ms window
    try:
        for b in range(n_blocks):
            res_b = (uvo.StereoResult * max(args.steps, 1))()              # the block's results land here in place
            stamps = [0.0] * (args.steps + 1)
            block_first_k.append(len(ks_seen))
            cpu0 = time.process_time()
            dt_local, records, allrec = timed_block(args.steps, res_b, stamps)
            block_cpu.append((time.process_time() - cpu0) / max(dt_local, 1e-9))
            block_local.append(dt_local)
            block_dt.append(multirank.max_over_ranks(dt_local, dev))      # the contract's MAX over ranks, per block
            block_stamps.append(stamps); all_results.append(res_b); all_records.append(records); all_allrec.append(allrec)
    finally:
        gc.enable()
    order_b = sorted(range(n_blocks), key=lambda b: block_dt[b])
    med = order_b[(n_blocks - 1) // 2]                                     # the median block (lower median when the count is even)
    dt, dt_local = block_dt[med], block_local[med]
    results, records, allrec = all_results[med], all_records[med], all_allrec[med]
    n_before_timed = block_first_k[med]
    busy_threads = block_cpu[med]                                          # host threads this rank kept busy on average (CPU seconds per second)
    n_valid = sum(r.valid for r in results[:args.steps])                   # statistics of the run, outside the timed regions
    n_valid_all = sum(r.valid for rb in all_results for r in rb[:args.steps])
    kp_sum = sum(r.n_left for r in results[:args.steps])
    assert [int(v) for v in allrec[:, 0, 0].tolist()] == list(range(world))
    ks_timed = ks_seen[n_before_timed - 1:n_before_timed + args.steps]     # [previous frame, then the K timed frames] of the median block
    if dist_on and world == 1:                                             # --force-dist: the collective must hand back exactly what went in
        assert allrec.shape[0] == 1 and np.array_equal(allrec[0].cpu().numpy().view(np.uint64), records.view(np.uint64)), "1-rank gather changed the records"
    total_pairs = args.steps * world
    value = total_pairs / dt
    block_values = [total_pairs / d for d in block_dt]
    # every rank's own rate in the median block (an imbalance between the GPUs of a node shows here; `value` uses the slowest)
    per_rank = multirank.gather_ints([int(round(args.steps / block_local[med] * 1000))], dev)[:, 0].tolist()
    # gaps between consecutive collects of every block (host clock): a stall inside a timed window shows as one large gap
    gaps = []
    for b, stamps in enumerate(block_stamps):
        prev_t = stamps[args.steps]
        for i in range(args.steps):
            gaps.append(((stamps[i] - prev_t) * 1e3, b, i)); prev_t = stamps[i]
    gs = sorted(g[0] for g in gaps)
    worst = max(gaps)
    collect_gap_ms = {"p50": round(gs[len(gs) // 2], 4), "p99": round(gs[min(len(gs) - 1, int(len(gs) * 0.99))], 4), "max": round(worst[0], 4),
                      "argmax": {"block": worst[1], "step": worst[2]}, "first_of_block_p50": round(sorted(g[0] for g in gaps if g[2] == 0)[n_blocks // 2], 4),
                      "what": "host clock between consecutive uvo_stereo_collect returns over all blocks; the first gap of a block is the pipeline fill"}
    trace_summary = None
    det_pipelined_ms = None                                                # the detection launch's duration with other pairs' kernels beside it
    if trace_on:
        if not args.trace_all:                                             # the diagnostic block: same shape, traced, not part of `value`
            ctx.trace_enable(True)
            res_t = (uvo.StereoResult * max(args.steps, 1))()
            dt_t, _, _ = timed_block(args.steps, res_t, [0.0] * (args.steps + 1))
            dt_t = multirank.max_over_ranks(dt_t, dev)
        tr = ctx.trace_read()
        ctx.trace_enable(False)
        trace_summary = summarise_trace(tr)
        trace_summary["traced"] = "every timed block" if args.trace_all else "one more block after the timed ones (not part of `value`)"
        if not args.trace_all:
            trace_summary["traced_block_value"] = round(total_pairs / dt_t, 3)
        if trace_summary.get("dev_detection_launch_ms"):
            det_pipelined_ms = trace_summary["dev_detection_launch_ms"]["p50"]
    elif not args.no_trace and not args.timed_only:                        # long blocks: one traced block of 64 pairs for the pipelined launch duration alone
        ctx.trace_enable(True)
        res_t = (uvo.StereoResult * 64)()
        timed_block(64, res_t, [0.0] * 65)
        tr = ctx.trace_read()
        ctx.trace_enable(False)
        d = summarise_trace(tr).get("dev_detection_launch_ms")
        det_pipelined_ms = d["p50"] if d else None

    if rank == 0 and args.dump_records:
        np.save(args.dump_records, allrec.cpu().numpy())

    if args.timed_only:
        if rank == 0:
            emit(json.dumps({"value": round(value, 3), "unit": "pairs/s", "steps": args.steps, "warmup": args.warmup, "timed_only": True,
                             "blocks": n_blocks, "value_first_block": round(block_values[0], 3), "block_values": [round(v, 1) for v in block_values],
                             "collect_gap_ms": collect_gap_ms, "busy_host_threads_rank0": round(busy_threads, 2),
                             "busy_host_threads_all_blocks": [round(v, 2) for v in block_cpu], "host_policy": host_policy,
                             "cores_of_this_rank": len(my_cores), "valid_steps_all_blocks": [n_valid_all, args.steps * n_blocks]}))
        if dist_on:
            multirank.barrier(); dist.destroy_process_group()
        ctx.close()
        return

    # ---- the same K steps with the images in (pageable) host memory: upload of both images inside the timed region ----
    # (SURVEY 8(d) figure (ii); reported beside `value`, never as `value`)
    h2d_value = None
    if world == 1:
        ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)       # fresh VO state
        order_h = ping_pong(args.frames)
        ctx.stereo_step(*host_frames[next(order_h)], 0.05)                          # init pair
        n_h = args.steps
        torch.cuda.synchronize()
        th = time.perf_counter()
        sub_h = 0
        for i in range(n_h):
            while sub_h < n_h and sub_h - i < args.depth:
                ctx.stereo_submit(*host_frames[next(order_h)]); sub_h += 1
            ctx.stereo_collect(0.05)
        torch.cuda.synchronize()
        h2d_value = n_h / (time.perf_counter() - th)

    out = None
    if rank == 0:
        # ---- latency leg: synchronous uvo_stereo_step, one pair in flight, wall clock per call ----
        lat = []
        for _ in range(200):
            a = time.perf_counter()
            rl = step()
            lat.append((time.perf_counter() - a) * 1e3)
        lat.sort()
        latency = {"median": round(lat[len(lat) // 2], 4), "p95": round(lat[int(len(lat) * 0.95)], 4), "samples": len(lat),
                   "what": "uvo_stereo_step, synchronous (one pair in flight), images resident in HBM"}
        # ---- roofline leg: HIP events on the context's stream around each stage ----
        ctx.timing_enable(True)
        ctx.timing_reset()
        for _ in range(10):
            rl = step()
        tm = ctx.timing()
        ctx.timing_enable(False)
        ms, n = tm["hessian_nms_o0"]
        avg_ms = ms / max(n, 1)
        alg_bytes = algorithmic_bytes_hessian_launch(WIDTH, HEIGHT, 2)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        # HBM bytes per launch from the counters: the two child passes run at the top of main() (measure_hbm_traffic) ...
        traffic, traffic_source = None, None
        if pmc_traffic is not None:
            traffic, traffic_source = pmc_traffic, "measured in this run: " + pmc_details["how"]
        else:
            # ... or, when the counter passes were not run here, the figure of the newest committed counter summary whose
            # `kernel_source_sha` matches the detector source being run (tools/pmc_summary.py) -- null when none does, never a stale number
            try:
                import hashlib
                sha = hashlib.sha256(open(os.path.join(ROOT, "ergo_uvo_amd", "csrc", "surf.hip"), "rb").read()).hexdigest()[:16]
                for name in ("r05_pmc_stage_kernels.json", "r04_pmc_stage_kernels.json", "r03_pmc_stage_kernels.json"):
                    d = json.load(open(os.path.join(ROOT, "profiles", name)))
                    if d.get("kernel_source_sha", {}).get("surf.hip") == sha:
                        traffic = int(d["kernels"]["hessian_all_octaves"]["hbm_bytes_fetch_x2"])
                        traffic_source = f"profiles/{name} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; surf.hip sha256 {sha}); not measured in this run: {pmc_details.get('skipped')}"
                        break
            except Exception:
                pass
        # the all-pairs contraction of the matcher against the f32 MFMA peak (F = 2 Nq Nt 64 per call)
        mm_ms, mm_n = tm["match_top2"]
        f_pair = 2.0 * 64 * (rl.n_left * rl.n_right + rl.n_stereo_matches * rl.n_left)
        mm_tflops = f_pair / max(mm_ms / 10.0 * 1e-3, 1e-12) / 1e12            # both matches of a pair (one launch since round 2) over the 10 timed pairs
        stage_ms = {k: round(v[0] / max(v[1], 1), 4) for k, v in tm.items() if v[1]}
        stage_calls = {k: v[1] // 10 for k, v in tm.items() if v[1]}
        # descriptor stage against HBM: B_desc of SURVEY 8(d) from the windows of the last pair's actual keypoints
        b_desc = 0
        for which in ("kps_left", "kps_right"):
            kp = ctx.stereo_get(which)
            win = np.floor(21.0 * (kp["size"].astype(np.float32) * np.float32(1.2) / np.float32(9.0))).astype(np.int64)
            b_desc += int((win * win).sum()) + len(kp) * (256 + 28)
        d_ms = tm["descriptor64"][0] / 10.0                               # per pair (all launches of the stage, both images)
        desc_gbs = b_desc / (d_ms * 1e-3) / 1e9

        cpu = None
        parity = {"parity_checked": False}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import pyoracle as po
            try:                                                           # one core, pinned
                aff = os.sched_getaffinity(0)
                os.sched_setaffinity(0, {sorted(aff)[-1]})
                pinned = True
            except Exception:
                aff, pinned = None, False
            ovo = po.StereoVO(po.stereo_params(min_hessian), rig.K_left, rig.K_right, rig.R_right, rig.t_right)
            o_order = ping_pong(args.frames)
            ks = [next(o_order)]
            ores = [ovo.step(*host_frames[ks[0]], 0.05)]                   # init pair, untimed
            oinl = [ovo.get("inliers")]
            t_each = []
            while len(t_each) < 3 or (len(t_each) < 20 and sum(t_each) < 2 * args.cpu_seconds) or (len(t_each) < 24 and sum(t_each) < args.cpu_seconds):   # >= 20 samples unless the host is very slow
                k = next(o_order); ks.append(k)
                a = time.perf_counter()
                ores.append(ovo.step(*host_frames[k], 0.05))
                t_each.append(time.perf_counter() - a)
                oinl.append(ovo.get("inliers"))
            if pinned:
                os.sched_setaffinity(0, aff)
            t_each.sort()
            med = t_each[len(t_each) // 2]
            cpu = {"value": round(1.0 / med, 4), "unit": "pairs/s", "cores": 1, "kind": "port", "pinned": pinned,
                   "sample": f"median of {len(t_each)} consecutive pairs of the same 1920x1080 sequence through oracle/ (C restatement, "
                             f"gcc -O2, 1 thread; not OpenCV)"}
            # ---- parity of the timed configuration: the same pairs through the HIP path, compared with the oracle's results ----
            ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
            bad, max_rel = [], 0.0
            for i, k in enumerate(ks):
                r = ctx.stereo_step(*dev_frames[k], 0.05)
                o = ores[i]
                for f in ("valid", "initialized", "n_left", "n_right", "n_stereo_matches", "n_tri_matches", "n_good3d", "n_inliers"):
                    if getattr(r, f) != getattr(o, f):
                        bad.append(f"pair {i}: {f} {getattr(r, f)} != {getattr(o, f)}")
                if not np.array_equal(ctx.stereo_get("inliers"), oinl[i]):
                    bad.append(f"pair {i}: inlier sets differ")
                for name in ("rvec", "tvec", "t_prev_curr"):
                    x, y = np.array(list(getattr(r, name))), np.array(list(getattr(o, name)))
                    rel = float(np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-300)) if np.linalg.norm(y) > 0 else float(np.linalg.norm(x))
                    max_rel = max(max_rel, rel)
                    if rel > 1e-4:
                        bad.append(f"pair {i}: {name} differs by {rel:.3g} relative")
            # ---- the records the TIMED submit/collect loop produced, against the oracle's result for the same frame transition
            # (a pair's pose depends on the previous pair only through its "after stereo match" set, so every (previous frame ->
            # frame) transition of the ping-pong order has one answer; the oracle's leg above visits all of them)
            want = {}
            for i in range(1, len(ks)):
                want.setdefault((ks[i - 1], ks[i]), ores[i])
            timed_checked, timed_rel = 0, 0.0
            for i in range(args.steps):
                o = want.get((ks_timed[i], ks_timed[i + 1]))
                if o is None:
                    continue
                timed_checked += 1
                row = records[i]
                if int(row[2]) != o.valid or int(row[3]) != o.n_inliers:
                    bad.append(f"timed step {i}: valid/n_inliers {int(row[2])}/{int(row[3])} != {o.valid}/{o.n_inliers}")
                for name, sl in (("rvec", slice(4, 7)), ("tvec", slice(7, 10)), ("t_prev_curr", slice(10, 13))):
                    x, y = row[sl], np.array(list(getattr(o, name)))
                    rel = float(np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-300))
                    timed_rel = max(timed_rel, rel)
                    if rel > 1e-4:
                        bad.append(f"timed step {i}: {name} differs by {rel:.3g} relative")
            parity = {"parity_checked": not bad, "parity": {"against": "oracle/ (CPU restatement; parity vs OpenCV unpinned)", "pairs": len(ks),
                      "compared": "gate counts, keypoint/match/3-D point counts, PnP inlier sets (bitwise), rvec/tvec/t_prev_curr (<= 1e-4 rel.)",
                      "max_pose_rel_diff": max_rel, "timed_loop_records_checked": timed_checked, "timed_loop_max_pose_rel_diff": timed_rel,
                      "mismatches": bad[:8]}}

        out = {
            "metric": "stereo frame-pairs/sec (detect+match+pose) @1920x1080, 3k kpts",
            "value": round(value, 3), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32+f64", "data": "synthetic",
            "config": {"workload": "C3: stereo UVO synthetic 1920x1080 pair, ~3000 SURF kpts/image, EPnP PnP-RANSAC"
                                   if world == 1 else "C5: one independent 1920x1080 stereo stream per GPU",
                       "min_hessian": min_hessian, "kpts_per_image": round(kp_sum / max(args.steps, 1), 1),
                       "valid_steps": n_valid, "valid_steps_all_blocks": [n_valid_all, args.steps * n_blocks], "frames": args.frames, "parallelism": f"streams{world}", "pipeline": f"submit/collect, {args.depth} pairs in flight per image stream"},
            "value_device_resident": round(value, 3),
            # the timed region (fence, K steps, gather, fence) run `blocks` times in a row in this process: `value` / `ms_per_step` are
            # the MEDIAN block's, every block's rate is listed in run order
            "blocks": n_blocks, "value_is": "median block", "value_first_block": round(block_values[0], 3),
            "value_min": round(min(block_values), 3), "value_max": round(max(block_values), 3), "block_values": [round(v, 1) for v in block_values],
            "warmup_pipelined_pairs": n_pipe_warm, "median_block": med, "pairs_before_median_block": n_before_timed,
            "collect_gap_ms": collect_gap_ms, "pipeline_trace": trace_summary,
            "per_rank_value": [v / 1000.0 for v in per_rank], "ranks_seen": sorted(who[:, 0].tolist()),
            "rccl_ranks_seen": (len(set(who[:, 0].tolist())) if (dist_on and args.backend == "nccl") else 0),
            "value_h2d_inclusive": None if h2d_value is None else round(h2d_value, 3),
            "roofline": {"bound": "hbm", "kernel": "k_hessian_nms_all (the four octaves, 3 middle layers each, 2 images per launch)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source, "traffic_counters": pmc_details,
                         "peak_measured": round(hbm_measured, 1), "frac_of_measured": round(achieved / hbm_measured, 5),
                         "peak_measured_what": "device-to-device copy of 1 GiB on this GPU at bench start, read + write bytes per second",
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(avg_ms, 5),
                         "frac_pipelined": (None if not det_pipelined_ms else round(alg_bytes / (det_pipelined_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)),
                         "pipelined_launch_ms": det_pipelined_ms,
                         "frac_is": "the launch alone on the chip (synchronous steps, HIP events around the stage); frac_pipelined: the same launch "
                                    "inside the pipeline `value` is timed in, other pairs' kernels beside it (median over a traced block, uvo_trace_row::dev_ms[6..7])"},
            "roofline_desc": {"bound": "hbm", "kernel": "descriptor stage of a pair (k_big_sort, k_descriptor64: small- and large-window blocks in one launch, k_descriptor64_big_finish)",
                              "achieved": round(desc_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(desc_gbs / HBM_PEAK_GBS, 6),
                              "algorithmic_bytes_per_pair": b_desc, "stage_ms_per_pair": round(d_ms, 5)},
            # the f32 contraction (F = 2 Nq Nt 64) is priced against the f32 MFMA peak; it is executed on the bf16 pipe
            # as three bf16 products per f32 product (hi.hi + hi.lo + lo.hi), so the executed rate is 3x, against the bf16 peak
            "roofline_match": {"bound": "mfma", "kernel": "k_match_mfma (v_mfma_f32_32x32x16_bf16 on bf16 hi/lo splits), both matches of a pair in one launch",
                               "achieved": round(mm_tflops, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(mm_tflops / MFMA_F32_PEAK_TFLOPS, 5), "flops_per_pair": f_pair,
                               "executed": {"achieved": round(3 * mm_tflops, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s (bf16)",
                                            "frac": round(3 * mm_tflops / MFMA_BF16_PEAK_TFLOPS, 5)}},
            "step_latency_ms": latency,
            "pair_hbm_frac": round(algorithmic_bytes_pair(WIDTH, HEIGHT, 3000, b_desc) * value / world / 1e9 / HBM_PEAK_GBS, 5),
            "algorithmic_bytes_per_pair": int(algorithmic_bytes_pair(WIDTH, HEIGHT, 3000, b_desc)),
            "stage_ms": stage_ms, "stage_launches_per_step": stage_calls,
            "cpu_baseline": cpu,
            "host": {"cores_visible": host_cores_all, "cores_of_this_rank": len(my_cores), "pinned": len(my_cores) < host_cores_all and bool(my_cores),
                     "pin_source": pin.get("source"), "cgroup_cpu_quota": pin.get("cpu_quota"), "cpu_quota_share_of_this_rank": pin.get("quota_share"),
                     "worker_wait": os.environ.get("UVO_WORKER_WAIT", "auto"), "stage_b": os.environ.get("UVO_STAGE_B", "auto"), "policy": host_policy,
                     "cpu_budget_env": os.environ.get("UVO_CPU_BUDGET"), "numa_node": pin.get("numa_node"), "gpu_pci": pin.get("pci"), "pin_matches_opened_device": pin.get("pci_match"),
                     "numa_nodes_of_ranks": who[:, 4].tolist(), "cores_of_ranks": who[:, 5].tolist(), "gc": "frozen and disabled during the timed blocks",
                     "threads_per_rank": ("1 submit / collect thread (polls); the PnP round of a pipelined pair is device-driven, the lane workers sleep"
                                          if host_policy.get("stage_b") == "device" else
                                          f"1 submitter (polls) + {args.depth} lane workers ({host_policy.get('wait')} for stage A's end; <= 3 at a time poll inside the PnP stage)"),
                     "busy_host_threads_rank0": round(busy_threads, 2),
                     "collectives": ("none (single process)" if not dist_on else f"{args.backend}: all_gather_into_tensor of the pose records, all_reduce(MAX) of the time, barriers; {world} rank(s)")},
        }
        out.update(parity)
    if dist_on:
        multirank.barrier()
        dist.destroy_process_group()
    ctx.close()
    if out is not None:
        emit(json.dumps(out))


if __name__ == "__main__":
    claim_stdout()
    main()
