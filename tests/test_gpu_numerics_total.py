"""tiny-cuda-nn's two ambiguous roundings (nrs_model_set_numerics) through EVERY entry point that evaluates the network:
round 2 built them for the operator kernels and for cage-edit renders with one lane per ray and refused the rest.  Now each has a run-time twin
(render_kernel<..., NUM = kNumRuntime>, slice / selection / grid-eval / refresh kernels alike), checked here against the oracle in the same mode:
membrane-correction renders, AffineDuplication renders, lane-team / hybrid / small-launch schedules and tiles (bit-identical to one lane per ray),
the occupancy refresh, density / RGBA on a grid, selection rays, the Slice mode, the membrane boundary fit."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_parity import _compare_frames

pytestmark = pytest.mark.gpu

MODES = [(1, 0), (0, 1), (1, 1)]
VOL = 128 ** 3


def _set(rig, g, m):
    rig.net.set_numerics(g, m)
    rig.scene.oracle_model.set_numerics(g, m)


@pytest.fixture
def anyrig(request):
    rigs = []

    def get(name):
        r = request.getfixturevalue(name)
        rigs.append(r)
        return r
    yield get
    for r in rigs:
        _set(r, 0, 0)
        r.use_edit(False)
        r.ctx.set_lane_teams(0)


@pytest.mark.parametrize("grid_acc,mlp_acc", MODES)
def test_membrane_render(anyrig, grid_acc, mlp_acc):
    from nerfshop_amd import runtime
    from oracle import oracle as orc
    rig = anyrig("rig")
    scene = rig.scene
    edit = scene.edit.with_membrane(residual_amplitude=0.8)
    op = runtime.CageDeformation(rig.ctx, scene.desc, edit)
    o_edit = orc.Edit(scene.desc, edit.tet_mesh_struct(), keepalive=edit)
    rig.use_edit(True)
    saved = rig.testbed.edit_operators
    try:
        rig.testbed.edit_operators = [op]
        _set(rig, grid_acc, mlp_acc)
        for target in (0, 1):
            p = scene.params_for(192, 108, 60.0)
            p.poisson_target = target
            got = rig.render(p)
            ref = scene.oracle_model.render(p, [o_edit])
            assert ref[3].n_hit > 500
            _compare_frames(got[0], got[1], got[2], ref[0], ref[1], ref[2])
    finally:
        rig.testbed.edit_operators = saved
        op.close()


@pytest.mark.parametrize("grid_acc,mlp_acc", MODES)
def test_affine_render(anyrig, grid_acc, mlp_acc):
    from test_gpu_affine import _edited_bitfield
    rig = anyrig("rig")
    scene = rig.scene
    op = scene.synth.make_affine_edit(hide_original=False)
    dev = rig.rt.AffineDuplication(rig.ctx, scene.desc, op)
    ref_op = scene.orc.AffineEdit(scene.desc, op)
    bits = _edited_bitfield(scene, [scene.oracle_edit, ref_op])
    saved = rig.testbed.edit_operators
    try:
        rig.testbed.edit_operators = [rig.op, dev]
        rig.net.set_density_bitfield(bits)
        scene.oracle_model.set_bitfield(bits)
        _set(rig, grid_acc, mlp_acc)
        p = scene.params_for(192, 108, 60.0)
        got = rig.render(p)
        ref = scene.oracle_model.render(p, [scene.oracle_edit, ref_op])
        assert got[3].n_rays_hit > 500
        _compare_frames(got[0], got[1], got[2], ref[0], ref[1], ref[2])
    finally:
        rig.testbed.edit_operators = saved
        dev.close()


@pytest.mark.parametrize("grid_acc,mlp_acc", [(1, 1), (0, 1)])
def test_every_schedule_and_tiles(anyrig, grid_acc, mlp_acc):
    """lane teams / hybrid / small-launch schedules and tiled launches with the other roundings: the bits of the one-lane-per-ray kernel, which the
    oracle check above covers (a tiled launch with numerics set used team geometry with the one-lane kernel)."""
    from nerfshop_amd import _abi
    rig = anyrig("rig")
    torch = rig.torch
    rig.use_edit(True)
    _set(rig, grid_acc, mlp_acc)
    p = rig.scene.params_for(320, 180, 45.0)
    out = {}
    for team in (1, 2, 4, -1, -2, 0):
        rig.ctx.set_lane_teams(team)
        out[team] = rig.render(p)
    ref = rig.scene.oracle_model.render(p, [rig.scene.oracle_edit])
    _compare_frames(out[1][0], out[1][1], out[1][2], ref[0], ref[1], ref[2])
    for team, (frame, depth, steps, stats) in out.items():
        assert np.array_equal(frame.view(np.uint32), out[1][0].view(np.uint32)) and np.array_equal(depth.view(np.uint32), out[1][1].view(np.uint32)), team
        assert np.array_equal(steps, out[1][2]) and stats.n_samples == out[1][3].n_samples, team
    # tiles (automatic choice = lane teams for a small share): reassembled == the whole image
    W, H, T = 320, 180, 64
    tiles_x = ((W + T - 1) // T) | 1   # the odd row pitch of the tile index (nrs.h)
    for team in (0, 4):
        image = np.zeros((H, W, 4), np.float32)
        rig.ctx.set_lane_teams(team)
        for rank in range(4):
            q = rig.scene.params_for(W, H, 45.0)
            q.tile_size, q.tile_first, q.tile_stride = T, rank, 4
            owned = _abi.load().nrs_render_owned_tiles(C.byref(q))
            frame = torch.zeros((owned, T, T, 4), dtype=torch.float32, device="cuda:0")
            depth = torch.zeros((owned, T, T), dtype=torch.float32, device="cuda:0")
            rig.testbed.render_with_params(rig.net, q, frame, depth, None, None, want_stats=True)
            torch.cuda.synchronize()
            f = frame.cpu().numpy()
            for k in range(owned):
                t = rank + 4 * k
                tx, ty = t % tiles_x, t // tiles_x
                h, w = min(T, H - ty * T), max(0, min(T, W - tx * T))
                image[ty * T:ty * T + h, tx * T:tx * T + w] = f[k, :h, :w]
        assert np.array_equal(image.view(np.uint32), out[1][0].view(np.uint32)), team


@pytest.mark.parametrize("grid_acc,mlp_acc", MODES)
def test_refresh_and_grid_evaluators(anyrig, grid_acc, mlp_acc):
    from test_gpu_grid_refresh import _compare, _oracle_update
    rig = anyrig("rig_shaped")
    scene, tb = rig.scene, rig.testbed
    tb.edit_operators = [rig.op]
    try:
        _set(rig, grid_acc, mlp_acc)
        u = tb.new_grid_update(max_cascade=0, seed=1337)
        u.reset_grid = 1
        ref_grid = np.zeros(5 * VOL, np.float32)
        u_ref, ref_bits = _oracle_update(scene, u, [scene.oracle_edit], ref_grid)
        tb.update_density_grid_nerf_operator(u)
        _compare(rig, scene, u, u_ref, ref_grid, ref_bits, 1)
        # and the modes are not a no-op for the refresh
        scene.oracle_model.set_numerics(0, 0)
        base = np.zeros(5 * VOL, np.float32)
        u0 = tb.new_grid_update(max_cascade=0, seed=1337)
        u0.reset_grid = 1
        _oracle_update(scene, u0, [scene.oracle_edit], base)
        assert (base[:VOL] != ref_grid[:VOL]).mean() > 0.01
        _set(rig, grid_acc, mlp_acc)
        # density / rgba on a grid
        rig.net.set_density_grid(scene.grid)
        res, mn, mx = (40, 36, 33), (0.1, 0.15, 0.2), (0.9, 0.8, 0.85)
        got = tb.get_density_on_grid(res, mn, mx).cpu().numpy().reshape(-1)
        ref = scene.oracle_model.density_on_grid(res, mn, mx, scene.grid)
        assert np.array_equal(got == -10000.0, ref == -10000.0)
        live = ref != -10000.0
        ulp = np.maximum(np.abs(ref[live]), 2.0 ** -14) * 2.0 ** -10
        assert (np.abs(got[live] - ref[live]) <= 4 * ulp).all() and (got[live] == ref[live]).mean() > 0.85
        rgba = tb.get_rgba_on_grid((32, 32, 32), (0.3, -0.5, 0.8)).cpu().numpy().reshape(-1, 4)
        ref_rgba = scene.oracle_model.rgba_on_grid((32, 32, 32), tb.render_aabb[0], tb.render_aabb[1], (0.3, -0.5, 0.8))
        assert np.abs(rgba - ref_rgba).max() < 4e-3
    finally:
        tb.edit_operators = []


@pytest.mark.parametrize("grid_acc,mlp_acc", MODES)
def test_selection_slice_and_boundary(anyrig, grid_acc, mlp_acc):
    from test_gpu_selection import _scribble
    rig = anyrig("rig")
    scene = rig.scene
    rig.use_edit(False)
    _set(rig, grid_acc, mlp_acc)
    w, h = 640, 360
    p = scene.params_for(w, h, 50.0)
    px = _scribble(w, h, 3000, 3)
    (pos, cells, found), _ = rig.testbed.project_selection_pixels(p, px, 0.1)
    ref_pos, ref_cells, ref_found = scene.oracle_model.project_selection_pixels(p, px, 0.1)
    same = (found == ref_found) & (cells == ref_cells) & (pos.view(np.uint32) == ref_pos.view(np.uint32)).all(1)
    assert same.mean() > 0.99, f"{(~same).sum()} of {same.size} rays differ"
    # Slice
    q = scene.params_for(200, 120, 40.0)
    q.render_mode, q.slice_plane_z = 9, 1.3
    got = rig.render(q)
    ref = scene.oracle_model.render(q, [])
    assert np.abs(got[0] - ref[0]).max() < 6e-3 and got[3].n_rays_hit == 200 * 120
    # membrane boundary fit (network_kernel already carried the modes; the fit kernel consumes its outputs)
    r = np.random.default_rng(5)
    verts = r.uniform(0.3, 0.7, (40, 3)).astype(np.float32)
    jitter = r.uniform(0, 1, (40 * 64, 2)).astype(np.float32)
    dens, sh = rig.testbed.compute_poisson_boundary(verts, True, jitter, 8, 8)
    ref_d, ref_sh, _ = scene.oracle_model.poisson_boundary(verts, 8, 8, jitter, True)
    assert np.allclose(dens, ref_d, rtol=3e-2, atol=1e-6) and np.abs(sh - ref_sh).max() < 8e-3 * max(1.0, np.abs(ref_sh).max())
