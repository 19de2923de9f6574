#!/usr/bin/env python3
"""Operator-level workloads for the rocprofv3 passes of tools/prof_operator.sh (profiles/r02_operator.md).  Dispatch order is fixed so the
summary can tell the cases apart:  network_kernel<0>: REPS x random samples, then REPS x ray-ordered samples;  network_kernel<1> (density):
REPS x random;  grid_refresh_kernel: 1 + REPS x aabb 1 (2^21 samples), then 1 + REPS x aabb 16 (5 * 2^21 samples), one cage operator each."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerfshop_amd import runtime, synth  # noqa: E402

REPS, N = 4, 1 << 22


def timed(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps


def main():
    ctx = runtime.Context(0)
    d = synth.model_desc(1)
    params = synth.make_params(d, sigma_raw=synth.default_sigma_raw(1), shaped=True)
    net = runtime.NerfNetwork(ctx, d)
    net.set_params(params)
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = torch.rand((N, 7), generator=g, device="cuda", dtype=torch.float32)
    # ray-ordered: a 2048 x 512 pinhole image, 4 consecutive steps of 1/1024 * sqrt(3); sample index = step * n_rays + ray (the order
    # generate_next_nerf_network_inputs writes, testbed_nerf.cu:1023: all rays' step j are contiguous)
    W, H, S = 2048, 512, 4
    xs = (torch.arange(W, device="cuda", dtype=torch.float32) + 0.5) / W - 0.5
    ys = ((torch.arange(H, device="cuda", dtype=torch.float32) + 0.5) / H - 0.5) * (H / W)
    dirs = torch.stack([xs[None, :].expand(H, W), ys[:, None].expand(H, W), torch.full((H, W), 0.9, device="cuda")], -1).reshape(-1, 3)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    o = torch.tensor([0.5, 0.5, -0.6], device="cuda")
    t = 0.9 + torch.arange(S, device="cuda", dtype=torch.float32) * (3 ** 0.5 / 1024)
    pos = o[None, None, :] + t[:, None, None] * dirs[None, :, :]
    ordered = torch.zeros((N, 7), device="cuda", dtype=torch.float32)
    ordered[:, :3] = pos.reshape(-1, 3).clamp(0.0, 1.0)
    ordered[:, 3] = 3 ** 0.5 / 1024
    ordered[:, 4:] = ((dirs + 1) * 0.5).repeat(S, 1)
    out = torch.zeros((N, 16), device="cuda", dtype=torch.float16)
    res = {}
    net.inference_mixed_precision(None, rnd, out)
    res["inference_random_ms"] = timed(lambda: net.inference_mixed_precision(None, rnd, out), REPS - 1)
    res["inference_ordered_ms"] = timed(lambda: net.inference_mixed_precision(None, ordered, out), REPS)
    res["density_random_ms"] = timed(lambda: net.density(None, rnd, out), REPS)
    net.close()
    for aabb_scale, max_cascade in ((1, 0), (16, 4)):
        d = synth.model_desc(aabb_scale)
        params = synth.make_params(d, sigma_raw=synth.default_sigma_raw(aabb_scale), shaped=True, aabb_scale=aabb_scale)
        tb = runtime.Testbed(ctx, d, aabb_scale)
        tb.nerf_network.set_params(params)
        e = synth.make_cage_edit(lattice_n=10, scene_scale=1.0 if aabb_scale == 1 else 6.0)
        tb.add_edit_operator(runtime.CageDeformation(ctx, d, e))
        u = tb.new_grid_update(max_cascade=max_cascade)
        u.reset_grid = 1
        tb.update_density_grid_nerf_operator(u)
        u.reset_grid = 0
        res[f"refresh_aabb{aabb_scale}_ms"] = timed(lambda: tb.update_density_grid_nerf_operator(u), REPS)
    res.update(reps=REPS, n=N)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
