"""1/8 share (and 1/4) with forced lane-team sizes and more frames in flight: is one lane per ray the better choice once several frames overlap?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nerfshop_amd import runtime as rt, synth, tiles
ctx = rt.Context(0)
scene = bench.build_scene("lego_cage", rt, synth, ctx, torch)
tb = scene["tb"]
W, H, T = 1920, 1080, bench.TILE
warm = synth.render_params(W, H, bench.camera_for(0, synth, 1), aabb_scale=1)
wf = torch.zeros((H, W, 4), device="cuda:0"); wd = torch.zeros((H, W), device="cuda:0")
for _ in range(150):
    tb.render_with_params(tb.nerf_network, warm, wf, wd, None, None)
torch.cuda.synchronize()
base = None
for N, team, F in [(1, 0, 1)] + [(n, t, f) for n in (8, 4) for t in (0, 1, 2, -1) for f in (1, 2, 4, 8)]:
    ctx.set_lane_teams(team)
    shs = [tiles.TileSharder(W, H, T, 0, N, "cuda:0") for _ in range(F)]
    streams = [torch.cuda.Stream() for _ in range(F)]
    samples = 0
    for step in range(8):
        p = synth.render_params(W, H, bench.camera_for(step, synth, 1), aabb_scale=1)
        shs[0].fill(p)
        samples += tb.render_with_params(tb.nerf_network, p, shs[0].local_frame, shs[0].local_depth, None, None, want_stats=True).n_samples
    for rep in range(2):
        torch.cuda.synchronize()
        K = 64
        t0 = time.perf_counter()
        for step in range(K):
            b = step % F
            p = synth.render_params(W, H, bench.camera_for(step % 8, synth, 1), aabb_scale=1)
            shs[b].fill(p)
            with torch.cuda.stream(streams[b]):
                shs[b].clear()
                tb.render_with_params(tb.nerf_network, p, shs[b].local_frame, shs[b].local_depth, None, streams[b])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / K
    rate = samples * (K / 8) / (ms * K) / 1e3
    base = base or rate
    print(f"N={N} team={team} in_flight={F}: {ms:.3f} ms, {rate:.0f} Msamples/s, retained {rate / base:.2f}", flush=True)
