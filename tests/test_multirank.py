"""N > 1 path on CPU: two and eight gloo ranks, one independent stereo stream each (driven here by the CPU oracle,
since the HIP path needs a GPU); the gathered record of rank r must equal the single-process result of
stream r bit for bit, and the max-over-ranks timing reduction must work.  Also: a forced process group of ONE
rank runs the same collectives (what `bench.py --gpus 1 --force-dist` does over RCCL), ranks rendezvous through a
file store, and the per-rank core slices are disjoint."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, STEPS, BASE_SEED = 320, 180, 3, 20250910


def _run_stream(stream: int):
    sys.path.insert(0, ROOT)
    from ergo_uvo_amd import synth, multirank
    from oracle import pyoracle as po
    scene = synth.Scene(multirank.stream_seed(BASE_SEED, stream), W)
    rig = synth.stereo_rig(W)
    vo = po.StereoVO(po.stereo_params(800), rig.K_left, rig.K_right, rig.R_right, rig.t_right, 4096)
    recs = torch.zeros((STEPS, multirank.RECORD_WIDTH), dtype=torch.float64)
    for k in range(STEPS):
        L, R = synth.stereo_pair(scene, k, W, H)
        recs[k] = multirank.make_record(stream, k, vo.step(L, R, 0.05))
    return recs


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except Exception as e:                      # surface failures instead of leaving the parent waiting
        q.put((rank, repr(e), None))
        raise


def _worker_body(rank, world, rdzv, q):
    # rendezvous the way bench.py's own launcher does it: a file store, no port to guess
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), UVO_RDZV_FILE=rdzv)
    for k in ("MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)
    sys.path.insert(0, ROOT)
    from ergo_uvo_amd import multirank
    cores = multirank.pin_rank(rank, world, gpus=[], quota=None)["cores"]        # no GPU topology on the CPU test host: index slices
    r, w = multirank.init("gloo", force=True)
    assert (r, w) == (rank, world)
    recs = _run_stream(rank)
    multirank.barrier()
    allrec = multirank.gather_records(recs)
    tmax = multirank.max_over_ranks(1.0 + rank)
    who = multirank.gather_ints([rank, os.getpid()] + [len(cores), cores[0] if cores else -1])
    q.put((rank, allrec.numpy().copy(), tmax, who.numpy().copy()))
    multirank.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 8])
def test_gloo_ranks_gather_matches_single_process(world, tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    rdzv = str(tmp_path / "store")
    procs = [ctx.Process(target=_worker, args=(r, world, rdzv, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, allrec, tmax, who = q.get(timeout=600)
        assert not isinstance(allrec, str), allrec
        got[rank] = (allrec, tmax, who)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([_run_stream(s).numpy() for s in range(world)])
    ncores = len(os.sched_getaffinity(0))
    for rank in range(world):
        allrec, tmax, who = got[rank]
        assert allrec.shape == (world, STEPS, 16)
        assert np.array_equal(allrec.view(np.uint64), want.view(np.uint64))       # stream i == single-process stream i, bitwise
        assert tmax == float(world)                                               # MAX over ranks
        assert who[:, 0].tolist() == list(range(world)) and len(set(who[:, 1].tolist())) == world     # N distinct ranks, N processes
        if world > 1 and ncores >= world:                                         # disjoint, equal core slices
            k = ncores // world
            assert who[:, 2].tolist() == [k] * world and len(set(who[:, 3].tolist())) == world
    assert want[0, 1, 2] == 1                                                     # the streams track after the init pair
    if world > 1:
        assert want[1, 1, 2] == 1
        assert not np.array_equal(want[0, 1, 7:10], want[1, 1, 7:10])             # and they are different streams


def _fake_topology(root, gpus_per_node=4, nodes=2, cores_per_node=16, hidden=()):
    """A sysfs / dev tree shaped like an 8-GPU host: KFD nodes 0..nodes-1 are CPUs, the rest GPUs (render minors 128..),
    each GPU's PCI device carrying numa_node / local_cpulist.  `hidden`: GPU ordinals whose render node is absent (another
    container's cards)."""
    base = root / "sys/class/kfd/kfd/topology/nodes"
    for n in range(nodes):
        (base / str(n)).mkdir(parents=True)
        (base / str(n) / "properties").write_text("cpu_cores_count %d\nsimd_count 0\n" % cores_per_node)
    (root / "dev/dri").mkdir(parents=True)
    g = 0
    for n in range(nodes):
        for _ in range(gpus_per_node):
            bus = 0x05 + 0x10 * g
            d = base / str(nodes + g)
            d.mkdir(parents=True)
            d.joinpath("properties").write_text("cpu_cores_count 0\nsimd_count 1024\ndomain 0\nlocation_id %d\ndrm_render_minor %d\n" % (bus << 8, 128 + g))
            pci = root / "sys/bus/pci/devices" / ("0000:%02x:00.0" % bus)
            pci.mkdir(parents=True)
            pci.joinpath("numa_node").write_text("%d\n" % n)
            pci.joinpath("local_cpulist").write_text("%d-%d\n" % (n * cores_per_node, (n + 1) * cores_per_node - 1))
            for c in range(n * cores_per_node, (n + 1) * cores_per_node):      # SMT-2: cpu c and cpu c + half a node are one core
                half = cores_per_node // 2
                lo = n * cores_per_node + (c - n * cores_per_node) % half
                t = root / "sys/devices/system/cpu" / ("cpu%d" % c) / "topology"
                t.mkdir(parents=True, exist_ok=True)
                t.joinpath("thread_siblings_list").write_text("%d,%d\n" % (lo, lo + half))
            if g not in hidden:
                (root / "dev/dri" / ("renderD%d" % (128 + g))).write_text("")
            g += 1


def test_ranks_are_pinned_to_the_numa_node_of_their_gpu(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    from ergo_uvo_amd import multirank
    _fake_topology(tmp_path)
    gpus = multirank.visible_gpus(str(tmp_path / "sys"), str(tmp_path / "dev"), env={})
    assert [g["pci"] for g in gpus] == ["0000:%02x:00.0" % (0x05 + 0x10 * i) for i in range(8)]
    assert [g["numa_node"] for g in gpus] == [0, 0, 0, 0, 1, 1, 1, 1]
    # HIP_VISIBLE_DEVICES reorders / filters; a container that was handed one card sees only that one
    assert [g["numa_node"] for g in multirank.visible_gpus(str(tmp_path / "sys"), str(tmp_path / "dev"), env={"HIP_VISIBLE_DEVICES": "5,1"})] == [1, 0]
    # both spellings set (launchers export the pair): HIP honours HIP_VISIBLE_DEVICES alone -- the list is not filtered twice
    assert [g["numa_node"] for g in multirank.visible_gpus(str(tmp_path / "sys"), str(tmp_path / "dev"), env={"HIP_VISIBLE_DEVICES": "2,3", "CUDA_VISIBLE_DEVICES": "2,3"})] == \
           [g["numa_node"] for g in multirank.visible_gpus(str(tmp_path / "sys"), str(tmp_path / "dev"), env={"HIP_VISIBLE_DEVICES": "2,3"})]
    assert len(multirank.visible_gpus(str(tmp_path / "sys"), str(tmp_path / "dev"), env={"CUDA_VISIBLE_DEVICES": "2,3"})) == 2
    assert multirank.visible_gpus(str(tmp_path / "sys"), str(tmp_path / "dev"), env={"ROCR_VISIBLE_DEVICES": "GPU-abc"}) == []
    one = tmp_path / "one"
    _fake_topology(one, hidden=(0, 1, 2, 3, 4, 6, 7))
    g1 = multirank.visible_gpus(str(one / "sys"), str(one / "dev"), env={})
    smt = lambda base: [base + c // 2 + 8 * (c % 2) for c in range(16)]          # a node's threads, siblings adjacent: 0, 8, 1, 9, ...
    assert len(g1) == 1 and g1[0]["numa_node"] == 1 and g1[0]["cpus"] == smt(16)

    # pin_rank on the fake 2 x 16-core host: no real affinity call (the test machine has other cores)
    calls = []
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(32)))
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cores: calls.append(sorted(cores)))
    pins = [multirank.pin_rank(r, 8, gpus=gpus, quota=None) for r in range(8)]
    assert [p["numa_node"] for p in pins] == [0, 0, 0, 0, 1, 1, 1, 1]
    assert [p["cores"] for p in pins] == [smt(16 * (r // 4))[4 * (r % 4):4 * (r % 4) + 4] for r in range(8)]   # 4 ranks split each node's 16 threads ...
    assert pins[0]["cores"] == [0, 8, 1, 9] and [sorted(p["cores"]) for p in pins] == calls                    # ... into whole cores (both SMT siblings)
    # one rank on a one-card box: the whole node next to the card, not the whole machine
    calls.clear()
    p = multirank.pin_rank(0, 1, gpus=g1, quota=None)
    assert sorted(p["cores"]) == list(range(16, 32)) and calls == [list(range(16, 32))] and p["pci"] == "0000:55:00.0"
    # five rehearsal ranks sharing the one card split its node
    calls.clear()
    pins = [multirank.pin_rank(r, 5, share_devices=True, gpus=g1, quota=None) for r in range(5)]
    assert [len(p["cores"]) for p in pins] == [3] * 5 and len({c for p in pins for c in p["cores"]}) == 15
    # topology unreadable: index slices as before, and a single rank stays unpinned
    calls.clear()
    pins = [multirank.pin_rank(r, 4, gpus=[], quota=None) for r in range(4)]
    assert [p["cores"] for p in pins] == [list(range(8 * r, 8 * r + 8)) for r in range(4)]
    assert multirank.pin_rank(0, 1, gpus=[], quota=None)["source"] == "unpinned"
    # a cgroup CPU quota (container limited to so many CPUs' worth of time): a rank keeps only its share of it -- whole cores first --
    # (on request: cut_to_quota) so that its busy threads cannot run the container into the throttle; read from cpu.max (v2) or cfs_quota / cfs_period (v1)
    calls.clear()
    q = multirank.pin_rank(0, 1, gpus=g1, quota=6.0, cut_to_quota=True)
    assert q["cores"] == smt(16)[:6] and calls == [sorted(smt(16)[:6])] and q["cpu_quota"] == 6.0
    assert [len(multirank.pin_rank(r, 4, gpus=gpus, quota=10.0, cut_to_quota=True)["cores"]) for r in range(4)] == [2, 2, 2, 2]      # 10 // 4, never below two
    assert len(multirank.pin_rank(0, 1, gpus=[], quota=8.0, cut_to_quota=True)["cores"]) == 8
    # by default the quota is only reported (with the rank's share of it): the rank keeps the cores next to its GPU
    q = multirank.pin_rank(1, 4, gpus=gpus, quota=10.0)
    assert len(q["cores"]) == 4 and q["cpu_quota"] == 10.0 and q["quota_share"] == 2.5
    cg = tmp_path / "cg"; cg.mkdir()
    (cg / "cpu.max").write_text("1600000 100000\n")
    assert multirank.cpu_quota(str(cg)) == 16.0
    (cg / "cpu.max").write_text("max 100000\n")
    assert multirank.cpu_quota(str(cg)) is None
    (cg / "cpu.max").unlink(); (cg / "cpu").mkdir()
    (cg / "cpu" / "cpu.cfs_quota_us").write_text("250000\n"); (cg / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert multirank.cpu_quota(str(cg)) == 2.5
    assert multirank.cpu_quota(str(tmp_path / "nowhere")) is None

    class Props: pci_domain_id, pci_bus_id, pci_device_id = 0, 0x55, 0
    assert multirank.pin_matches_device(p, Props) is True
    Props.pci_bus_id = 0x15
    assert multirank.pin_matches_device(p, Props) is False
    assert multirank.pin_matches_device({"pci": None}, Props) is None
