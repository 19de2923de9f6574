"""Soak of the lattice jump (nrs_device.cuh): the first samples of every pixel of 300 random cameras (a tenth of them within a hundredth of a degree of a coordinate
axis), 11 M rays, bit for bit against the oracle's cell-by-cell walk.  Too long for the test tier (tests/test_gpu_lattice_jump.py is its short form); run through gpurun:
    python tools/lattice_soak.py
Round 5's run: 11 059 200 rays, 5 028 799 with samples, cameras with any differing bit: 0; then 120 whole frames (the render kernel's per-round walk): none with a
pixel more than one sample apart or a different ray count, smallest share of pixels with equal sample counts 0.999946."""
import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
from conftest import Scene, GpuRig
scene = Scene(aabb_scale=1, with_edit=True, lattice_n=6)
rig = GpuRig(scene)
rng = np.random.default_rng(2026)
W, H = 256, 144
bad = 0; rays = 0; hit = 0
for k in range(300):
    use_edit = (k % 3 == 0)
    if k % 50 == 0: rig.use_edit(use_edit)
    az, el = float(rng.uniform(0, 360)), float(rng.uniform(-85, 85))
    if k % 10 == 0: az, el = float(rng.choice([0, 90, 180, 270])) + float(rng.normal(0, 0.01)), float(rng.normal(0, 0.01))   # near the axes
    p = scene.synth.render_params(W, H, scene.synth.orbit_camera(az, el, scale=0.33), snap=bool(k & 1), spp_index=k)
    idx = np.arange(W * H, dtype=np.uint32)
    t_ref, dt_ref, c_ref = scene.oracle_model.trace_samples(p, idx, 4)
    t, dt, c = rig.testbed.trace_samples(p, torch.from_numpy(idx.astype(np.int32)).cuda(), 4)
    ok = np.array_equal(c.cpu().numpy().astype(np.uint32), c_ref) and np.array_equal(t.cpu().numpy().view(np.uint32), t_ref.view(np.uint32))
    bad += 0 if ok else 1
    rays += W * H; hit += int((c_ref > 0).sum())
print(f"soak: {300} cameras, {rays} rays, {hit} with samples, cameras with ANY differing bit: {bad}")

# ---- whole frames: the per-round walk of the render kernel (its own copy of the jump), per-pixel sample counts against the oracle
worst, off = 1.0, 0
for k in range(120):
    if k % 40 == 0:
        rig.use_edit(k % 80 == 0)
    az, el = float(rng.uniform(0, 360)), float(rng.uniform(-85, 85))
    if k % 8 == 0:
        az, el = float(rng.choice([0, 90, 180, 270])) + float(rng.normal(0, 0.01)), float(rng.normal(0, 0.01))
    p = scene.synth.render_params(W, H, scene.synth.orbit_camera(az, el, scale=0.33), snap=True)
    frame, depth, steps, stats = rig.render(p)
    edits = [scene.oracle_edit] if rig.testbed.edit_operators else []
    ref_frame, ref_depth, ref_steps, ref_stats = scene.oracle_model.render(p, edits)
    ds = np.abs(steps.astype(np.int64) - ref_steps.astype(np.int64))
    off += int(ds.max() > 1) + int(stats.n_rays_alive != ref_stats.n_alive0)
    worst = min(worst, float((ds == 0).mean()))
print(f"soak: 120 frames, frames with a pixel more than one sample apart or a different ray count: {off}; smallest share of pixels with equal sample counts: {worst:.6f}")
