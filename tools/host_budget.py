#!/usr/bin/env python3
"""tools/host_budget.py -- what one rank needs from the host: the table behind bench.py's choice of how a rank waits and who drives
the PnP stage of pipelined pairs.

  python tools/host_budget.py [--out profiles/r05_host_budget.json] [--cpus 2 4 16] [--modes spin sleep block-all device] [--reps 1]

For every {mode} x {CPU budget} it runs bench.py twice as a child process -- the 7 x 600-pair form and the driver's 7 x 20-pair form
(`--steps 20 --warmup 5`) -- with `--cpus N`: the child confines itself to N logical CPUs with os.sched_setaffinity BEFORE its first
GPU call (never a re-exec), whole physical cores next to the GPU first, and hands N to the library as its CPU budget.  Modes:
  spin       UVO_STAGE_B=worker UVO_WORKER_WAIT=spin       lane workers poll for stage A's end and inside the PnP stage
  sleep      UVO_STAGE_B=worker UVO_WORKER_WAIT=sleep      workers sleep on a timer through most of stage A, then poll
  block-all  UVO_STAGE_B=worker UVO_WORKER_WAIT=block-all  every wait sleeps on the GPU's interrupt
  device     UVO_STAGE_B=device                            first RANSAC round device-driven, uvo_stereo_collect confirms it
  auto       nothing set: what the library picks for the budget
Per cell: value (median block), first block, every block, busy host threads (process CPU seconds per second of the timed region),
p99 / max gap between consecutive collects, the largest gap that is not a block's pipeline fill.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MODES = {
    "spin": {"UVO_STAGE_B": "worker", "UVO_WORKER_WAIT": "spin"},
    "sleep": {"UVO_STAGE_B": "worker", "UVO_WORKER_WAIT": "sleep"},
    "block-all": {"UVO_STAGE_B": "worker", "UVO_WORKER_WAIT": "block-all"},
    "device": {"UVO_STAGE_B": "device"},
    "device-block": {"UVO_STAGE_B": "device", "UVO_WORKER_WAIT": "block-all"},
    "auto": {},
}


def run_cell(mode: str, cpus: int, steps: int, warmup: int, extra: list[str]) -> dict:
    env = {k: v for k, v in os.environ.items() if k not in ("UVO_STAGE_B", "UVO_WORKER_WAIT", "UVO_CPU_BUDGET")}
    env.update(MODES[mode])
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", str(warmup), "--timed-only", "--cpus", str(cpus)] + extra
    t0 = time.time()
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    if p.returncode != 0:
        return {"error": p.stderr[-600:], "wall_s": round(time.time() - t0, 1)}
    d = json.loads(p.stdout.strip().splitlines()[-1])
    g = d["collect_gap_ms"]
    return {"value": d["value"], "value_first_block": d["value_first_block"], "block_values": d["block_values"],
            "busy_host_threads": d["busy_host_threads_rank0"], "busy_host_threads_max": max(d["busy_host_threads_all_blocks"]),
            "collect_gap_ms": {"p50": g["p50"], "p99": g["p99"], "max": g["max"], "argmax": g["argmax"], "first_of_block_p50": g["first_of_block_p50"]},
            "valid_steps": d["valid_steps_all_blocks"], "host_policy": d["host_policy"], "cores": d["cores_of_this_rank"], "wall_s": round(time.time() - t0, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_host_budget.json"))
    ap.add_argument("--cpus", type=int, nargs="+", default=[2, 4, 16])
    ap.add_argument("--modes", nargs="+", default=["spin", "sleep", "block-all", "device"], choices=sorted(MODES))
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--forms", nargs="+", default=["600", "20"], choices=["600", "20"])
    ap.add_argument("bench_args", nargs="*", help="further bench.py arguments (after --)")
    args = ap.parse_args()
    table = {"what": __doc__.split("\n\n")[0], "modes": {m: MODES[m] for m in args.modes}, "cells": []}
    for cpus in args.cpus:
        for mode in args.modes:
            for rep in range(args.reps):
                cell = {"mode": mode, "cpus": cpus, "rep": rep}
                if "600" in args.forms:
                    cell["form_7x600"] = run_cell(mode, cpus, 600, 20, args.bench_args)
                if "20" in args.forms:
                    cell["form_7x20_driver"] = run_cell(mode, cpus, 20, 5, args.bench_args)
                table["cells"].append(cell)
                a, b = cell.get("form_7x600", {}), cell.get("form_7x20_driver", {})
                print(f"cpus {cpus:3d} {mode:12s} | 600: {a.get('value')} busy {a.get('busy_host_threads')} gap p99/max {a.get('collect_gap_ms', {}).get('p99')}/{a.get('collect_gap_ms', {}).get('max')}"
                      f" | 20: {b.get('value')} first {b.get('value_first_block')} busy {b.get('busy_host_threads')} gap max {b.get('collect_gap_ms', {}).get('max')}"
                      f" | {a.get('host_policy') or b.get('host_policy') or a.get('error') or b.get('error')}", flush=True)
                with open(args.out, "w") as f:
                    json.dump(table, f, indent=1)
    print("written", args.out)


if __name__ == "__main__":
    main()
