"""Latency floor of a render call: fixed overhead, an all-miss frame, and small frames under both lane schedules."""
import sys, json; sys.path.insert(0, '.')
import torch, bench
from nerfshop_amd import runtime as rt, synth
ctx = rt.Context(0)
scene = bench.build_scene("lego_cage", rt, synth, ctx, torch)
tb = scene["tb"]
def run(W, H, stats, flip=False, reps=20, schedule=0):
    frame = torch.zeros((H, W, 4), device="cuda"); depth = torch.zeros((H, W), device="cuda")
    cam = bench.camera_for(0, synth, 1).copy()
    if flip: cam[6:9] *= -1
    p = synth.render_params(W, H, cam, aabb_scale=1)
    ts = []
    for i in range(reps + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); st = tb.render_with_params(tb.nerf_network, p, frame, depth, None, None, want_stats=stats); e1.record()
        torch.cuda.synchronize()
        if i >= 3: ts.append(e0.elapsed_time(e1))
    return round(sum(ts) / len(ts), 4)
print("8x8 away, stats      ", run(8, 8, True, True))
print("8x8 away, no stats   ", run(8, 8, False, True))
print("1080p away, no stats ", run(1920, 1080, False, True))
for (W, H) in ((64, 40), (256, 144), (640, 360), (960, 540)):
    print(f"{W}x{H} hit:", run(W, H, False))
