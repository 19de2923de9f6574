"""Stateful soak of the objects behind the C-ABI: ONE model + ONE cage operator per scene scale driven through a seeded random sequence of the calls an editing viewer
makes -- cage moves (nrs_edit_update_cage: LUT / rotations / plane records rebuilt on the device, the fine look-up table dropped and rebuilt by the second frame at rest),
record-cache changes (nrs_model_set_cell_cache 0 / 1 / 10 GiB, nrs_model_set_sparse_cell_cache none / 4 GiB / 80 GiB: the launch takes the default kernel or the GATE
instantiation with two, three or four phases), new parameters (nrs_model_set_params: records rebuilt), occupancy changes -- and, after every step, one to three frames at
random cameras against the oracle in the state the sequence has reached (bars of tests/test_gpu_parity.py::_compare_frames).  What it is after: stale derived state
(a table that survives the change that should have dropped it).  Too long for the test tier; run through gpurun:
    python tools/state_soak.py [steps]
Prints one line per step and a summary; exit code 1 when a frame misses a bar.
Round 6's run (profiles/r06/state_soak.txt): 240 steps, 455 frames, none beyond the bars."""
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from nerfshop_amd import runtime, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(66)
W, H = 192, 108


class Rig:
    def __init__(self, aabb_scale):
        self.aabb_scale = aabb_scale
        self.scale = 1.0 if aabb_scale == 1 else 6.0
        self.ctx = runtime.Context(0)
        self.desc = synth.model_desc(aabb_scale)
        self.tb = runtime.Testbed(self.ctx, self.desc, aabb_scale)
        self.net = self.tb.nerf_network
        self.edit = synth.make_cage_edit(lattice_n=5, scene_scale=self.scale)
        self.op = runtime.CageDeformation(self.ctx, self.desc, self.edit, device_authoring=True)
        self.op.set_mvc(self.edit.mvc_weights)
        self.tb.add_edit_operator(self.op)
        self.verts = self.edit.vertices
        self.grid = synth.density_grid(aabb_scale)
        self.seed = 0
        self.new_params()
        self.new_pose((0.10, 0.05, 0.0), 20.0)

    def new_params(self):
        self.seed += 1
        self.params = synth.make_params(self.desc, seed=1000 + self.seed, sigma_raw=synth.default_sigma_raw(self.aabb_scale))  # (another table and other weights, the same opacity)
        self.net.set_params(self.params)

    def new_pose(self, translate, twist):
        cage = synth.deform_cage(self.edit.cage_vertices, tuple(t * self.scale for t in translate), twist)
        self.verts = orc.mvc_apply(self.edit.mvc_weights, cage)
        self.op.update_cage(None, cage)
        off, idx, _, mx = orc.tet_lut_build(self.verts, self.edit.tets)
        e = self.edit
        host = synth.CageEdit(vertices=np.ascontiguousarray(self.verts, np.float32), original_vertices=e.original_vertices, tets=e.tets, lut_offsets=off,
                              lut_idx=idx if idx.size else np.zeros(1, np.uint32), original_bitfield=e.original_bitfield,
                              local_rotations=synth.local_rotations(self.verts, e.original_vertices, e.tets), copy=False, cage_vertices=e.cage_vertices,
                              cage_triangles=e.cage_triangles, cage_deformed=cage, mvc_weights=e.mvc_weights, mvc_labels=e.mvc_labels, max_per_cell=mx)
        self.oracle_edit = orc.Edit(self.desc, host.tet_mesh_struct(), keepalive=host)
        grid2 = synth.deformed_density_grid(self.grid, self.desc, self.oracle_edit.map_positions, self.aabb_scale)
        self.bitfield = synth.grid_to_bitfield(grid2)
        self.mask = self.bitfield | synth.grid_to_bitfield(self.grid)
        self.net.set_density_bitfield(self.bitfield)

    def frame_ok(self):
        from test_gpu_parity import _compare_frames
        model = orc.Model(self.desc, self.params, self.bitfield)
        az, el = float(rng.uniform(0, 360)), float(rng.uniform(-60, 60))
        p = synth.render_params(W, H, synth.orbit_camera(az, el, scale=0.33 * self.scale * float(rng.uniform(0.8, 1.3))), aabb_scale=self.aabb_scale)
        frame = torch.zeros((H, W, 4), device="cuda:0"); depth = torch.zeros((H, W), device="cuda:0"); steps = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
        self.tb.render_with_params(self.net, p, frame, depth, steps, None)
        torch.cuda.synchronize()
        ref_frame, ref_depth, ref_steps, _ = model.render(p, [self.oracle_edit])
        try:
            _compare_frames(frame.cpu().numpy(), depth.cpu().numpy(), steps.cpu().numpy(), ref_frame, ref_depth * 1.0, ref_steps) if self.aabb_scale == 1 else \
                _compare_scaled(frame.cpu().numpy(), depth.cpu().numpy(), steps.cpu().numpy(), ref_frame, ref_depth, ref_steps)
            return True
        except AssertionError as e:
            print("   MISSED:", str(e)[:200])
            return False


def _compare_scaled(frame, depth, steps, ref_frame, ref_depth, ref_steps):
    d = np.abs(frame - ref_frame)
    assert d.max() < 6e-3 and d.mean() < 2e-4, (d.max(), d.mean())
    ds = np.abs(steps.astype(np.int64) - ref_steps.astype(np.int64))
    assert ds.max() <= 1 and (ds == 0).mean() >= 0.998, (ds.max(), (ds == 0).mean())
    hit = (ref_frame[..., 3] > 0.2) & (frame[..., 3] > 0.2) & (ds == 0)
    assert np.allclose(depth[hit], ref_depth[hit], rtol=0, atol=2e-3 * 16)


rigs = {1: Rig(1), 16: Rig(16)}
bad = frames = 0
for step in range(n_steps):
    r = rigs[16 if step % 3 == 2 else 1]
    op = int(rng.integers(0, 5))
    if op == 0:
        what = "cage move"
        r.new_pose(tuple(float(v) for v in rng.uniform(-0.08, 0.12, 3)), float(rng.choice([0.0, 15.0, 40.0, 70.0])))
    elif op == 1:
        b = int(rng.choice([0, 1 << 30, 10 << 30]))
        what = f"cell cache {b >> 30} GiB"
        r.net.set_cell_cache(b)
    elif op == 2 and r.aabb_scale == 16:
        b = int(rng.choice([0, 4 << 30, 80 << 30]))
        what = f"sparse cache {b >> 30} GiB"
        r.net.set_sparse_cell_cache(r.mask if b else None, b)
    elif op == 3:
        what = "new parameters"
        r.new_params()
    else:
        what = "frames only"
    n = int(rng.integers(1, 4))
    ok = all(r.frame_ok() for _ in range(n))
    frames += n
    bad += 0 if ok else 1
    print(f"step {step:3d} aabb {r.aabb_scale:2d}: {what:22s} -> {n} frame(s) {'ok' if ok else 'MISSED'}; records: dense {r.net.cell_cache()[1]} levels, sparse {r.net.sparse_cell_cache()[2]} levels", flush=True)
print(f"state soak: {n_steps} steps, {frames} frames against the oracle, steps with a frame beyond the bars: {bad}")
sys.exit(1 if bad else 0)
