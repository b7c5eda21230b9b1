"""The eight streams of BASELINE configs[4] (seeds SEEDS["C5"] + rank) on ONE device, one after the other: keypoint counts against the
context's capacity and pose validity for each, so that an 8-GPU run does not meet a stream for the first time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ergo_uvo_amd as uvo
from ergo_uvo_amd import synth, multirank
W, H = 1920, 1080
for rank in range(8):
    seed = multirank.stream_seed(synth.SEEDS["C5"], rank)
    scene = synth.Scene(seed, W)
    rig = synth.stereo_rig(W)
    ctx = uvo.Context(uvo.Params.stereo(SURF_MIN_HESSIAN=6387), 0, W, H, 8192)
    ctx.stereo_set_rig(rig.K_left, rig.K_right, rig.R_right, rig.t_right)
    nl, nv, inl = [], 0, []
    for k in (0, 1, 2, 3, 2, 1, 0):
        L, R = synth.stereo_pair(scene, k, W, H)
        r = ctx.stereo_step(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda(), 0.05)
        nl.append(max(r.n_left, r.n_right)); nv += r.valid; inl.append(r.n_inliers)
    print(f"stream {rank} seed {seed}: keypoints per image <= {max(nl)} of 8192, valid {nv} of 6, inliers {min(inl[1:])}..{max(inl)}", flush=True)
    ctx.close()
