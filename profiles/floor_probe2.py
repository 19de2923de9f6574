"""Where does the time of a tiny frame go?  64x40 pixels (900 rays) under 1 / 2 / 4 lanes per ray, with and without the
cell-record cache, with and without the edit, and the per-pixel step counts (longest ray)."""
import sys, json; sys.path.insert(0, '.')
import numpy as np
import torch, bench
from nerfshop_amd import runtime as rt, synth
ctx = rt.Context(0)
scene = bench.build_scene("lego_cage", rt, synth, ctx, torch)
tb = scene["tb"]
def run(W, H, team, ops=True, reps=20):
    ctx.set_lane_teams(team)
    frame = torch.zeros((H, W, 4), device="cuda"); depth = torch.zeros((H, W), device="cuda")
    steps = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    p = synth.render_params(W, H, bench.camera_for(0, synth, 1).copy(), aabb_scale=1, apply_operators=ops)
    ts = []
    for i in range(reps + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); tb.render_with_params(tb.nerf_network, p, frame, depth, steps if i == 0 else None, None); e1.record()
        torch.cuda.synchronize()
        if i >= 3: ts.append(e0.elapsed_time(e1))
    s = steps.cpu().numpy()
    return round(sum(ts) / len(ts), 4), int(s.max()), float(s[s > 0].mean()) if (s > 0).any() else 0.0
for cache in (10 << 30,):
    tb.nerf_network.set_cell_cache(cache)
    for ops in (True,):
        for team in (1, 2, 4, -2):
            for (W, H) in ((64, 40), (8, 8), (16, 16), (32, 32)):
                ms, longest, mean = run(W, H, team, ops)
                print(json.dumps({"cell_cache": cache >> 30, "edit": ops, "lanes_per_ray": team, "frame": [W, H], "ms": ms, "longest_ray_samples": longest, "mean_samples": round(mean, 1)}))
