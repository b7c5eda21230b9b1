"""N > 1 path on CPU: two gloo ranks, one independent stereo stream each (driven here by the CPU oracle,
since the HIP path needs a GPU); the gathered record of rank r must equal the single-process result of
stream r bit for bit, and the max-over-ranks timing reduction must work."""
import os
import sys

import numpy as np
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


def _worker_body(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from ergo_uvo_amd import multirank
    r, w = multirank.init("gloo")
    assert (r, w) == (rank, world)
    recs = _run_stream(rank)
    multirank.barrier()
    allrec = multirank.gather_records(recs)
    tmax = multirank.max_over_ranks(1.0 + rank)
    q.put((rank, allrec.numpy().copy(), tmax))
    multirank.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_gather_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, allrec, tmax = q.get(timeout=240)
        assert not isinstance(allrec, str), allrec
        got[rank] = (allrec, tmax)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([_run_stream(s).numpy() for s in range(2)])
    for rank in range(2):
        allrec, tmax = got[rank]
        assert allrec.shape == (2, STEPS, 16)
        assert np.array_equal(allrec.view(np.uint64), want.view(np.uint64))       # stream i == single-process stream i, bitwise
        assert tmax == 2.0                                                        # MAX over ranks
    assert want[0, 1, 2] == 1 and want[1, 1, 2] == 1                              # both streams track after the init pair
    assert not np.array_equal(want[0, 1, 7:10], want[1, 1, 7:10])                 # and they are different streams
