"""The rest of Testbed::render_nerf's surface (VERDICT r2 missing #1, #2) through the C-ABI on an MI355X, each case against the
oracle, which tests/test_ref_pin.py pins bit for bit to the reference's own composite / shade / init kernels for the same cases:
  * composite_kernel_nerf's per-sample render modes AO / Positions / Depth / Distance / Stepsize (testbed_nerf.cu:905-937),
  * show_accel: opaque samples (:788-790) and the occupancy-cell colouring of Positions (:911-920, tcnn::default_rng_t),
  * shade_kernel_nerf's mode handling (:2466-2482),
  * pixel_to_ray's thin-lens branch (m_dof, common_device.cuh:285-293),
  * render mode Slice (:3067-3070, 3109-3162).
  * the camera model and the background: OpenCV (iterative) and f-theta lens distortion, the distortion map, the environment map, render mode
    Distortion (init_rays_with_payload_kernel_nerf :2523-2613, common_device.cuh:146-243, 262-280, envmap.cuh:30-63).
Normals / EncodingVis (tiny-cuda-nn's input gradient / visualize_activation, restated): tests/test_gpu_introspection.py.

Tolerances.  The modes replace the network's colour by a function of the (bit-exact) sample position, so the frame bar is the Shade bar
(6e-3 max, 2e-4 mean) scaled by the magnitude of the colours a mode produces (depths and distances in scene units).  Depth of field is the one
place where ray origins are not bit-identical to the host-compiled reference: square2disk_shirley calls sincosf (device library here, glibc in
the oracle, CUDA's in the reference) -- a last-bit difference in the lens sample moves the ray by an ulp, and a ray that grazes a cell border
may then gain or lose a sample: stated as >= 99 % of the pixels with equal sample counts and <= 1 % of the pixels above the Shade bar."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

AO, SHADE, NORMALS, POSITIONS, DEPTH, DISTANCE, STEPSIZE, DISTORTION, COST, SLICE = range(10)


def _params(rig, w, h, az, **fields):
    p = rig.scene.params_for(w, h, az)
    for k, v in fields.items():
        setattr(p, k, v)
    return p


def _compare(got, ref, depth_atol=2e-3):
    frame, depth, steps, _ = got
    ref_frame, ref_depth, ref_steps, _ = ref
    scale = max(1.0, float(np.abs(ref_frame[..., :3]).max()))
    d = np.abs(frame - ref_frame)
    assert d.max() < 6e-3 * scale, (d.max(), scale)
    assert d.mean() < 2e-4 * scale, d.mean()
    ds = np.abs(steps.astype(np.int64) - ref_steps.astype(np.int64))
    assert ds.max() <= 1 and (ds == 0).mean() >= 0.998, (ds.max(), (ds == 0).mean())
    hit = (ref_frame[..., 3] > 0.2) & (frame[..., 3] > 0.2) & (ds == 0)
    assert np.allclose(depth[hit], ref_depth[hit], rtol=0, atol=depth_atol)
    miss = (ref_frame[..., 3] == 0) & (frame[..., 3] == 0)
    assert (depth[miss] == 1e10).all() and (ref_depth[miss] == 1e10).all()


@pytest.mark.parametrize("mode,fields", [
    (COST, {}), (AO, {}), (POSITIONS, {}), (DEPTH, {"depth_scale": 0.7}), (DISTANCE, {"depth_scale": 1.3}), (STEPSIZE, {}),
    (POSITIONS, {"show_accel": 1, "min_mip": 0}), (POSITIONS, {"show_accel": 1, "min_mip": 2}), (SHADE, {"show_accel": 1, "min_mip": 0}),
    (COST, {"show_accel": 1, "min_mip": 1}), (AO, {"linear_colors": 1, "snap_to_pixel_centers": 0, "spp_index": 4}),
])
@pytest.mark.parametrize("edit", [False, True])
def test_render_modes_lego(rig, mode, fields, edit):
    rig.use_edit(edit)
    try:
        p = _params(rig, 256, 144, 60.0, render_mode=mode, **fields)
        got = rig.render(p)
        ref = rig.scene.oracle_model.render(p, [rig.scene.oracle_edit] if edit else [])
        assert ref[3].n_hit > 1000 and got[3].n_rays_alive == ref[3].n_alive0
        _compare(got, ref)
        assert abs(int(got[3].n_samples) - int(ref[3].composited)) <= 0.002 * ref[3].composited + 2
        if mode != SHADE:  # and the mode is not a no-op
            q = _params(rig, 256, 144, 60.0, **{k: v for k, v in fields.items() if k != "show_accel"})
            plain = rig.render(q)
            assert np.abs(plain[0] - got[0]).max() > 0.05
    finally:
        rig.use_edit(False)


@pytest.mark.parametrize("mode,fields", [(AO, {}), (POSITIONS, {"show_accel": 1, "min_mip": 1}), (DEPTH, {"depth_scale": 0.5}), (DISTANCE, {"depth_scale": 1.0}), (STEPSIZE, {})])
def test_render_modes_aabb16(rig16, mode, fields):
    """cone stepping, 5 cascades, cage edit: Stepsize shows the growing dt, Positions the cascade of the cell (rgb.x = 1 - mip / 4)"""
    rig = rig16
    rig.use_edit(True)
    try:
        p = _params(rig, 256, 144, 30.0, render_mode=mode, **fields)
        got = rig.render(p)
        ref = rig.scene.oracle_model.render(p, [rig.scene.oracle_edit])
        assert ref[3].n_hit > 10000
        _compare(got, ref, depth_atol=2e-3 * 16)
    finally:
        rig.use_edit(False)


def test_render_modes_with_membrane_correction(rig):
    """the residual blend uses the mode's colour (tn:939-943): AO and Depth with apply_poisson on"""
    from nerfshop_amd import runtime
    from oracle import oracle as orc
    scene = rig.scene
    edit = scene.edit.with_membrane(residual_amplitude=0.8)
    op = runtime.CageDeformation(rig.ctx, scene.desc, edit)
    o_edit = orc.Edit(scene.desc, edit.tet_mesh_struct(), keepalive=edit)
    rig.use_edit(True)
    saved = rig.testbed.edit_operators
    try:
        rig.testbed.edit_operators = [op]
        for mode in (AO, DEPTH, SHADE):
            p = _params(rig, 192, 108, 60.0, render_mode=mode, depth_scale=1.0, dof=0.0 if mode != SHADE else 0.0, show_accel=0)
            got = rig.render(p)
            ref = scene.oracle_model.render(p, [o_edit])
            _compare(got, ref)
    finally:
        rig.testbed.edit_operators = saved
        rig.use_edit(False)
        op.close()


@pytest.mark.parametrize("which,dof,focus", [("rig", 0.02, 1.2), ("rig", 0.1, 0.9), ("rig16", 0.05, 2.5)])
def test_depth_of_field(request, which, dof, focus):
    rig = request.getfixturevalue(which)
    rig.use_edit(True)
    try:
        p = _params(rig, 256, 144, 60.0, dof=dof, slice_plane_z=focus, snap_to_pixel_centers=0, spp_index=3)
        frame, depth, steps, stats = rig.render(p)
        ref_frame, ref_depth, ref_steps, ref_stats = rig.scene.oracle_model.render(p, [rig.scene.oracle_edit])
        assert ref_stats.n_hit > 1000
        ds = np.abs(steps.astype(np.int64) - ref_steps.astype(np.int64))
        d = np.abs(frame - ref_frame).max(-1)
        print(f"dof {dof}: steps equal {(ds == 0).mean():.5f}, max step diff {ds.max()}, pixels above 6e-3: {(d > 6e-3).mean():.5f}, max {d.max():.3e}, mean {np.abs(frame - ref_frame).mean():.2e}")
        assert (ds == 0).mean() >= 0.99
        assert (d > 6e-3).mean() <= 0.01 and np.abs(frame - ref_frame).mean() < 5e-4
        # the aperture changes the picture (it is not silently ignored), and dof = 0 is the pinhole frame bit for bit whatever the focus distance
        q = _params(rig, 256, 144, 60.0, snap_to_pixel_centers=0, spp_index=3)
        pin = rig.render(q)
        assert np.abs(pin[0] - frame).max() > 0.02
        q.slice_plane_z = focus
        pin2 = rig.render(q)
        assert np.array_equal(pin[0].view(np.uint32), pin2[0].view(np.uint32))
    finally:
        rig.use_edit(False)


@pytest.mark.parametrize("which,plane,fields", [("rig", 1.3, {}), ("rig", 1.1, {"linear_colors": 1, "snap_to_pixel_centers": 0, "spp_index": 6}), ("rig16", 6.0, {})])
def test_slice(request, which, plane, fields):
    rig = request.getfixturevalue(which)
    rig.use_edit(False)
    W, H = 250, 130  # ragged: not a multiple of the 8x8 packets
    p = _params(rig, W, H, 40.0, render_mode=SLICE, slice_plane_z=plane, **fields)
    torch = rig.torch
    frame = torch.full((H, W, 4), 0.25, dtype=torch.float32, device="cuda:0")  # over a background, as shade_kernel_nerf composites
    depth = torch.zeros((H, W), dtype=torch.float32, device="cuda:0")
    steps = torch.full((H, W), 7, dtype=torch.int32, device="cuda:0")
    stats = rig.testbed.render_with_params(rig.net, p, frame, depth, steps, None, want_stats=True)
    torch.cuda.synchronize()
    lib = rig.scene.orc.load()
    ref_frame = np.full((H, W, 4), 0.25, np.float32)
    ref_depth = np.zeros((H, W), np.float32)
    ref_steps = np.zeros((H, W), np.uint32)
    st = rig.scene.orc.OrcRenderStats()
    lib.orc_render(rig.scene.oracle_model.h, C.byref(p), (C.c_void_p * 1)(), 0, ref_frame.ctypes.data, ref_depth.ctypes.data, ref_steps.ctypes.data, C.byref(st), 0, 0)
    got = frame.cpu().numpy()
    assert stats.n_rays_hit == W * H == st.n_hit and stats.n_samples == W * H
    assert np.abs(got - ref_frame).max() < 6e-3 and np.abs(got - ref_frame).mean() < 2e-4
    assert (got[..., 3] > 0.26).sum() > 500 and np.unique(got[..., :3]).size > 1000    # something was drawn over the background, and it varies (the hash grid's colours)
    assert np.array_equal(depth.cpu().numpy(), np.full((H, W), plane, np.float32)) and np.array_equal(ref_depth, depth.cpu().numpy())  # tn:2583
    assert (steps.cpu().numpy() == 0).all()


def test_slice_on_tiles(rig):
    """Slice through the multi-GPU tile interface: the owned tiles of three 'ranks' reassemble the whole-image frame bit for bit"""
    from nerfshop_amd import _abi
    rig.use_edit(False)
    torch = rig.torch
    W, H, T = 320, 200, 64
    whole = rig.render(_params(rig, W, H, 40.0, render_mode=SLICE, slice_plane_z=1.3))
    tiles_x, tiles_y = ((W + T - 1) // T) | 1, (H + T - 1) // T   # the odd row pitch of the tile index (nrs.h)
    image = np.zeros((H, W, 4), np.float32)
    for rank in range(3):
        p = _params(rig, W, H, 40.0, render_mode=SLICE, slice_plane_z=1.3, tile_size=T, tile_first=rank, tile_stride=3)
        owned = _abi.load().nrs_render_owned_tiles(C.byref(p))
        frame = torch.zeros((owned, T, T, 4), dtype=torch.float32, device="cuda:0")
        depth = torch.zeros((owned, T, T), dtype=torch.float32, device="cuda:0")
        rig.testbed.render_with_params(rig.net, p, frame, depth, None, None, want_stats=True)
        torch.cuda.synchronize()
        f = frame.cpu().numpy()
        for k in range(owned):
            t = rank + 3 * k
            tx, ty = t % tiles_x, t // tiles_x
            h, w = min(T, H - ty * T), max(0, min(T, W - tx * T))
            image[ty * T:ty * T + h, tx * T:tx * T + w] = f[k, :h, :w]
    assert tiles_x * tiles_y > 9
    assert np.array_equal(image.view(np.uint32), whole[0].view(np.uint32))


def test_modes_through_every_boundary_flavour(rig):
    """a mode with forced lane-team settings (the EXTRA instantiation is one lane per ray whatever the context asks for); unknown modes, an unknown lens
    model, a lens without a focus distance and a parameter struct of another size are refused"""
    from nerfshop_amd._abi import NrsError
    rig.use_edit(True)
    try:
        p = _params(rig, 256, 144, 60.0, render_mode=DEPTH, depth_scale=1.0)
        ref = rig.scene.oracle_model.render(p, [rig.scene.oracle_edit])
        for team in (4, -1, -2, 0):
            rig.ctx.set_lane_teams(team)
            _compare(rig.render(p), ref)
        rig.ctx.set_lane_teams(0)
        for bad in (10, 12):   # NumRenderModes, out of range (Normals = 2 and EncodingVis = 11 are modes since round 4: tests/test_gpu_introspection.py)
            with pytest.raises(NrsError):
                rig.render(_params(rig, 64, 36, 60.0, render_mode=bad))
        with pytest.raises(NrsError):
            rig.render(_params(rig, 64, 36, 60.0, dof=0.1, slice_plane_z=0.0))
        with pytest.raises(NrsError):
            rig.render(_params(rig, 64, 36, 60.0, depth_scale=float("nan")))
        with pytest.raises(NrsError):
            rig.render(_params(rig, 64, 36, 60.0, distortion_mode=3))
        old_client = _params(rig, 64, 36, 60.0)
        old_client.struct_size -= 8   # built against another nrs.h (ABI 3: the size is the contract, nothing is read past a foreign struct)
        with pytest.raises(NrsError):
            rig.render(old_client)
    finally:
        rig.ctx.set_lane_teams(0)
        rig.use_edit(False)


def test_modes_with_affine_stack_and_on_tiles(rig):
    """the EXTRA instantiation is the catch-all: AffineDuplication + cage operators in a mode against the oracle, and a mode rendered through the
    multi-GPU tile interface (three 'ranks') reassembles the whole-image frame bit for bit"""
    from nerfshop_amd import _abi
    from test_gpu_affine import _edited_bitfield
    scene, torch = rig.scene, rig.torch
    op = scene.synth.make_affine_edit(hide_original=False)
    dev = rig.rt.AffineDuplication(rig.ctx, scene.desc, op)
    ref_op = scene.orc.AffineEdit(scene.desc, op)
    bits = _edited_bitfield(scene, [scene.oracle_edit, ref_op])
    saved = rig.testbed.edit_operators
    try:
        rig.testbed.edit_operators = [rig.op, dev]
        rig.net.set_density_bitfield(bits)
        scene.oracle_model.set_bitfield(bits)
        W, H, T = 256, 144, 32
        p = _params(rig, W, H, 60.0, render_mode=POSITIONS)
        whole = rig.render(p)
        ref = scene.oracle_model.render(p, [scene.oracle_edit, ref_op])
        assert ref[3].n_hit > 1000
        _compare(whole, ref)
        tiles_x = ((W + T - 1) // T) | 1   # the odd row pitch of the tile index (nrs.h)
        image = np.zeros((H, W, 4), np.float32)
        for rank in range(3):
            q = _params(rig, W, H, 60.0, render_mode=POSITIONS, tile_size=T, tile_first=rank, tile_stride=3)
            owned = _abi.load().nrs_render_owned_tiles(C.byref(q))
            frame = torch.zeros((owned, T, T, 4), dtype=torch.float32, device="cuda:0")
            depth = torch.zeros((owned, T, T), dtype=torch.float32, device="cuda:0")
            rig.testbed.render_with_params(rig.net, q, frame, depth, None, None, want_stats=True)
            torch.cuda.synchronize()
            f = frame.cpu().numpy()
            for k in range(owned):
                t = rank + 3 * k
                tx, ty = t % tiles_x, t // tiles_x
                h, w = min(T, H - ty * T), max(0, min(T, W - tx * T))
                image[ty * T:ty * T + h, tx * T:tx * T + w] = f[k, :h, :w]
        assert np.array_equal(image.view(np.uint32), whole[0].view(np.uint32))
    finally:
        rig.testbed.edit_operators = saved
        rig.use_edit(False)
        dev.close()


# ---- camera model and background: lens distortion, distortion map, environment map, render mode Distortion (init_rays :2523-2613) ----------------------
def _camera_pair(rig, w, h, az, fields, envmap=None, distmap=None):
    """the same nrs_render_params twice: DEVICE pointers for libnrs, HOST pointers for the oracle"""
    from nerfshop_amd import runtime
    from ref_pin_cases import camera_extras
    torch = rig.torch
    over = {}
    if envmap:
        over["_envmap"] = envmap
    if distmap:
        over["_distmap"] = distmap
    p_cpu, p_gpu = _params(rig, w, h, az), _params(rig, w, h, az)
    for p in (p_cpu, p_gpu):
        for k, v in fields.items():
            if k == "distortion_params":
                p.distortion_params[:] = v
            else:
                setattr(p, k, v)
    host = camera_extras(p_cpu, over)
    keep, i = [host], 0
    env_t = dm_t = None
    if envmap:
        env_t = torch.from_numpy(host[i]).cuda(); i += 1
    if distmap:
        dm_t = torch.from_numpy(host[i]).cuda()
    runtime.set_camera_extras(p_gpu, None, dm_t, env_t)
    keep += [env_t, dm_t]
    return p_gpu, p_cpu, keep


@pytest.mark.parametrize("name,fields,envmap,distmap,exact_rays", [
    ("envmap", {}, (32, 16, 5), None, True),
    ("opencv", {"distortion_mode": 1, "distortion_params": (0.12, -0.05, 0.004, -0.003, 0, 0, 0), "snap_to_pixel_centers": 0, "spp_index": 2}, None, None, True),
    ("ftheta", {"distortion_mode": 2, "distortion_params": (0.0, 0.7, 0.02, -0.01, 0.002, 1.0, 0.5625)}, None, None, False),
    ("ftheta_wide", {"distortion_mode": 2, "distortion_params": (0.0, 2.9, 0.0, 0.0, 0.0, 1.0, 0.5625)}, (32, 16, 5), None, False),
    ("distmap_envmap", {}, (40, 20, 6), (24, 12, 9, 0.02), True),
    ("everything_in_depth_mode", {"render_mode": DEPTH, "depth_scale": 1.0, "distortion_mode": 1, "distortion_params": (-0.2, 0.08, 0.0, 0.0, 0, 0, 0), "dof": 0.02, "slice_plane_z": 1.2,
                                  "snap_to_pixel_centers": 0, "spp_index": 7}, (32, 16, 5), (24, 12, 9, 0.01), False),
])
def test_camera_model_and_background(rig, name, fields, envmap, distmap, exact_rays):
    rig.use_edit(True)
    try:
        p_gpu, p_cpu, keep = _camera_pair(rig, 256, 144, 60.0, fields, envmap, distmap)
        got = rig.render(p_gpu)
        ref = rig.scene.oracle_model.render(p_cpu, [rig.scene.oracle_edit])
        assert ref[3].n_hit > 500
        frame, depth, steps, _ = got
        ds = np.abs(steps.astype(np.int64) - ref[2].astype(np.int64))
        d = np.abs(frame - ref[0]).max(-1)
        if exact_rays:   # plain fp32 arithmetic in the reference's order: the rays are the oracle's bits, the usual frame bar applies (acosf / atan2f of the envmap lookup: 1e-6 in the angle)
            assert ds.max() <= 1 and (ds == 0).mean() >= 0.998
            assert d.max() < 6e-3 and np.abs(frame - ref[0]).mean() < 2e-4, (d.max(), np.abs(frame - ref[0]).mean())
        else:            # sincosf (f-theta, thin lens): the depth-of-field bar
            assert (ds == 0).mean() >= 0.99 and (d > 6e-3 * max(1.0, float(ref[0][..., :3].max()))).mean() <= 0.01, ((ds == 0).mean(), (d > 6e-3).mean())
        plain = rig.render(_params(rig, 256, 144, 60.0))
        assert np.abs(plain[0] - frame).max() > 0.02   # not a no-op
        if envmap:
            assert (frame[..., 3] > 0).all()             # every pixel got its background
    finally:
        rig.use_edit(False)


@pytest.mark.parametrize("distmap", [None, (24, 12, 9, 0.004)])
def test_render_mode_distortion(rig, distmap):
    rig.use_edit(False)
    p_gpu, p_cpu, keep = _camera_pair(rig, 250, 130, 40.0, {"render_mode": DISTORTION}, None, distmap)
    torch = rig.torch
    frame = torch.full((130, 250, 4), 0.25, dtype=torch.float32, device="cuda:0")
    depth = torch.zeros((130, 250), dtype=torch.float32, device="cuda:0")
    stats = rig.testbed.render_with_params(rig.net, p_gpu, frame, depth, None, None, want_stats=True)
    torch.cuda.synchronize()
    lib = rig.scene.orc.load()
    ref_frame = np.full((130, 250, 4), 0.25, np.float32)
    ref_depth = np.zeros((130, 250), np.float32)
    ref_steps = np.zeros((130, 250), np.uint32)
    st = rig.scene.orc.OrcRenderStats()
    lib.orc_render(rig.scene.oracle_model.h, C.byref(p_cpu), (C.c_void_p * 1)(), 0, ref_frame.ctypes.data, ref_depth.ctypes.data, ref_steps.ctypes.data, C.byref(st), 0, 0)
    got = frame.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref_frame.view(np.uint32))     # plain arithmetic: the same bits
    assert np.array_equal(depth.cpu().numpy(), ref_depth) and stats.n_samples == 0 and stats.n_rays_hit == 0
    assert (got[..., 3] == 1.0).sum() > 1000 and ((ref_depth == 1.0) | (ref_depth == 1e10)).all()


def test_slice_through_a_distorted_lens(rig):
    """Slice takes its rays through pixel_to_ray like every mode: lens distortion applies (the aperture does not, tn:2543-2545)"""
    rig.use_edit(False)
    fields = {"render_mode": SLICE, "slice_plane_z": 1.3, "distortion_mode": 1, "distortion_params": (0.15, -0.05, 0.003, 0.002, 0, 0, 0), "dof": 0.05}
    p_gpu, p_cpu, keep = _camera_pair(rig, 200, 120, 40.0, fields, None, (24, 12, 9, 0.01))
    got = rig.render(p_gpu)
    ref = rig.scene.oracle_model.render(p_cpu, [])
    assert np.abs(got[0] - ref[0]).max() < 6e-3 and np.abs(got[0] - ref[0]).mean() < 2e-4
    straight = rig.render(_params(rig, 200, 120, 40.0, render_mode=SLICE, slice_plane_z=1.3))
    assert np.abs(straight[0] - got[0]).max() > 0.01


@pytest.mark.parametrize("which,glow_mode,cutoff", [("rig", 3, 0.55), ("rig", 24, 0.5), ("rig", 7, 0.5), ("rig16", 5, 0.6)])
def test_glow_overlay(request, which, glow_mode, cutoff):
    """composite_kernel_nerf's grid / cut-line overlay (tn:806-903): green grid + cut line, radial grid-only, all three with the mask scaling the weights
    (which changes how far rays march), on the aabb-16 scene too.  cosf of the device library against glibc's: the Shade bar scaled by the overlay's brightness."""
    rig = request.getfixturevalue(which)
    rig.use_edit(True)
    try:
        p = _params(rig, 256, 144, 60.0, glow_mode=glow_mode, glow_y_cutoff=cutoff)
        got = rig.render(p)
        ref = rig.scene.oracle_model.render(p, [rig.scene.oracle_edit])
        assert ref[3].n_hit > 1000
        _compare(got, ref, depth_atol=2e-3 * (16 if which == "rig16" else 1))
        plain = rig.render(_params(rig, 256, 144, 60.0))
        assert np.abs(plain[0] - got[0]).max() > 0.05
    finally:
        rig.use_edit(False)
