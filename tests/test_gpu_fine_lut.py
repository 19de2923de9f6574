"""The fine look-up table under the cage's 128^3 cell -> tet LUT (round 6; nrs_cage.hip fine_lists_kernel, nrs_device.cuh find_tet) is a shortcut of the reference's
scan (interpolate_tet, cage_deformation.cu:197-269: the first tet of the cell's list that contains the sample): every result must be the plain scan's, bit for bit.
The plain scan is the same library with NRS_NO_FINE_LUT (a measurement knob: its own process); both are also held against the oracle elsewhere
(tests/test_gpu_parity.py::test_map_rays_bit_exact, tests/test_gpu_bench_parity.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
from nerfshop_amd import runtime, synth
aabb_scale, lattice, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
ctx = runtime.Context(0)
desc = synth.model_desc(aabb_scale)
scale = 1.0 if aabb_scale == 1 else 6.0
edit = synth.make_cage_edit(lattice_n=lattice, scene_scale=scale)
op = runtime.CageDeformation(ctx, desc, edit, device_authoring=True)
op.set_mvc(edit.mvc_weights)
res = {}
rng = np.random.default_rng(11)
mn, mx = synth.scene_aabb(aabb_scale)
mn, mx = np.array(mn, np.float32), np.array(mx, np.float32)


def probe(tag, verts):
    n = 1 << 20
    lo, hi = verts.min(0), verts.max(0)
    ext = hi - lo
    world = rng.uniform(lo - 0.03 * ext, hi + 0.03 * ext, size=(n, 3)).astype(np.float32)
    # a fifth of the points ON vertices / edge midpoints / face centres of tets (where two tets both pass the test: the list ORDER decides), and just beside them
    t = edit.tets[rng.integers(0, edit.tets.shape[0], n // 5)]
    w = rng.dirichlet([0.4, 0.4, 0.4, 0.4], size=n // 5).astype(np.float32)
    w[: n // 15] = np.eye(4, dtype=np.float32)[rng.integers(0, 4, n // 15)]
    w[n // 15: 2 * n // 15] = 0.5 * (np.eye(4, dtype=np.float32)[rng.integers(0, 4, n // 15)] + np.eye(4, dtype=np.float32)[rng.integers(0, 4, n // 15)])
    on = np.einsum("nk,nkd->nd", w, verts[t]).astype(np.float32)
    world[: n // 5] = on
    world[n // 10: n // 5] += rng.normal(0, 2e-7, size=(n // 5 - n // 10, 3)).astype(np.float32) * np.float32(scale)
    c = np.zeros((n, 7), np.float32)
    c[:, :3] = (world - mn) / (mx - mn)   # warped [0, 1] coordinates
    c[:, 3] = 1e-3
    c[:, 4:] = rng.uniform(0, 1, size=(n, 3)).astype(np.float32)
    dc = torch.from_numpy(c).cuda()
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    op.map_rays(None, dc, mask)
    torch.cuda.synchronize()
    res[tag + "_coords"] = dc.cpu().numpy()
    res[tag + "_mask"] = mask.cpu().numpy()
    res[tag + "_moved"] = np.array([(res[tag + "_coords"][:, :3] != c[:, :3]).any(axis=1).sum()])


probe("created", edit.vertices)
pose = synth.deform_cage(edit.cage_vertices, tuple(t * scale for t in (0.07, 0.03, -0.02)), 33.0)
op.update_cage(None, pose)
torch.cuda.synchronize()
# a move drops the fine table; the second frame rendered after it builds the new one (nrs_api.cpp: nrs_edit::fine_stale)
tb = runtime.Testbed(ctx, desc, aabb_scale)
tb.nerf_network.set_cell_cache(0)
tb.nerf_network.set_params(synth.make_params(desc, sigma_raw=synth.default_sigma_raw(aabb_scale)))
tb.nerf_network.set_density_bitfield(synth.grid_to_bitfield(synth.density_grid(aabb_scale)))
tb.add_edit_operator(op)
p = synth.render_params(64, 36, synth.orbit_camera(30.0, 30.0, scale=0.33 * scale), aabb_scale=aabb_scale)
frame = torch.zeros((36, 64, 4), device="cuda:0"); depth = torch.zeros((36, 64), device="cuda:0")
for _ in range(2):
    tb.render_with_params(tb.nerf_network, p, frame, depth, None, None)
torch.cuda.synchronize()
verts2 = synth.mvc_apply(edit.mvc_weights, pose).astype(np.float32)
probe("moved", verts2)
np.savez(out, **res)
"""


@pytest.mark.parametrize("aabb_scale,lattice", [(1, 10), (16, 6), (1, 20)])
def test_fine_lut_is_the_plain_scan(built, tmp_path, aabb_scale, lattice):
    outs = {}
    for tag, env_extra in (("fine", {"NRS_DEV_KNOBS": "1", "NRS_FINE_LOG": "1"}), ("plain", {"NRS_DEV_KNOBS": "1", "NRS_FINE_LOG": "1", "NRS_NO_FINE_LUT": "1"})):
        out = str(tmp_path / f"{tag}.npz")
        env = dict(os.environ, **env_extra)
        r = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(aabb_scale), str(lattice), out], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = np.load(out)
        # the table was built at creation and again by the second frame after the move -- and never in the plain run
        assert r.stderr.count("[nrs fine lut]") == (2 if tag == "fine" else 0), r.stderr[-2000:]
    for key in outs["fine"].files:
        a, b = outs["fine"][key], outs["plain"][key]
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b), key
    # the probe is not vacuous: a good share of the points was carried back by a tet
    assert int(outs["fine"]["created_moved"][0]) > 100000 and int(outs["fine"]["moved_moved"][0]) > 100000
