"""ms per 1080p frame of the bench's lego + cage scene in a render mode other than Shade (the EXTRA instantiations: tn:905-937) -- usage: python tools/render_mode_driver.py [mode ...]
(NRS_DEV_KNOBS=1 NRS_RENDER_CFG=1 sends the frame to the catch-all instantiation: the A/B of round 6's lean one)."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from nerfshop_amd import runtime as rt, synth, _abi
ctx = rt.Context(0)
sc = bench.build_scene("lego_cage", rt, synth, ctx, torch)
tb = sc["tb"]
W, H = 1920, 1080
frame = torch.zeros((H, W, 4), device="cuda:0"); depth = torch.zeros((H, W), device="cuda:0")
modes = [int(v) for v in sys.argv[1:]] or [_abi.RENDER_DEPTH, _abi.RENDER_AO, _abi.RENDER_POSITIONS]
out = {}
for mode in modes:
    def step(k, stats=False):
        p = synth.render_params(W, H, bench.camera_for(k, synth, 1), aabb_scale=1, apply_operators=True)
        p.render_mode = mode
        frame.zero_()
        return tb.render_with_params(tb.nerf_network, p, frame, depth, None, None, want_stats=stats)
    ns = sum(int(step(k, True).n_samples) for k in range(8))
    for k in range(3): step(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(16): step(k)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / 16
    out[str(mode)] = {"ms": round(ms, 3), "msamples_per_s": round(ns / 8 / ms / 1e3, 1)}
print(json.dumps(out))
