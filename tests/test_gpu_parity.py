"""Parity of the HIP path (through the C-ABI of libnrs.so) against the CPU oracle, on an MI355X.

Bars: bit-exact for integer / index work and for every fp32 quantity whose operation order we control (hash-grid
features, ray/sample positions, cage warp); stated tolerances where the MFMA's fp32 accumulation order or a fast
exp() enters (network outputs, composited colour).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rand_coords(n, seed):
    rng = np.random.default_rng(seed)
    c = rng.uniform(0.0, 1.0, size=(n, 7)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    c[:, 4:7] = ((d + 1.0) * 0.5).astype(np.float32)
    return c


def _half_ulp_distance(a_u16, b_u16):
    """distance in fp16 ulps between two fp16 bit patterns (monotone integer mapping)."""
    def key(u):
        u = u.astype(np.int32)
        return np.where(u & 0x8000, -(u & 0x7FFF), u & 0x7FFF)
    return np.abs(key(a_u16) - key(b_u16))


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 100003])
def test_hashgrid_bit_exact(rig, n):
    torch = rig.torch
    c = _rand_coords(n, 11 + n)
    # include the corners of the unit cube and exact cell boundaries
    c[: min(n, 8), :3] = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], np.float32)[: min(n, 8)]
    ref = rig.scene.oracle_model.hashgrid_encode(c)
    out = torch.zeros((n, 32), dtype=torch.float16, device="cuda:0")
    rig.net.hashgrid_encode(None, torch.from_numpy(c).cuda(), out)
    got = out.cpu().numpy().view(np.uint16)
    assert np.array_equal(got, ref), f"{(got != ref).sum()} of {got.size} features differ"


@pytest.mark.parametrize("layout", [0, 1])
def test_network_inference_tolerance(rig, layout):
    """fp16 outputs within 4 fp16 ulps (or 2e-3 abs near zero): the MFMA sums fp16 products in fp32 in its own order, the
    oracle sums them exactly; a hidden activation that lands on the other side of an fp16 rounding boundary moves the
    outputs by a few ulps."""
    torch = rig.torch
    n = 20000
    c = _rand_coords(n, 5)
    ref = rig.scene.oracle_model.inference(c, layout)
    out = torch.zeros((16, n) if layout == 0 else (n, 16), dtype=torch.float16, device="cuda:0")
    rig.net.inference_mixed_precision(None, torch.from_numpy(c).cuda(), out)
    got = out.cpu().numpy()
    ulps = _half_ulp_distance(got.view(np.uint16), ref)
    absd = np.abs(got.astype(np.float32) - ref.view(np.float16).astype(np.float32))
    ok = (ulps <= 4) | (absd <= 2e-3)
    assert ok.all(), f"max ulps {ulps.max()}, max abs {absd.max()}, bad {np.count_nonzero(~ok)}"
    assert (ulps == 0).mean() > 0.9  # the vast majority is bit-identical


def test_network_density_tolerance(rig):
    torch = rig.torch
    n = 10001
    c = _rand_coords(n, 6)[:, :3].copy()
    ref = rig.scene.oracle_model.density(c, 0)
    out = torch.zeros((16, n), dtype=torch.float16, device="cuda:0")
    rig.net.density(None, torch.from_numpy(c).cuda(), out)
    got = out.cpu().numpy()
    ulps = _half_ulp_distance(got.view(np.uint16), ref)
    absd = np.abs(got.astype(np.float32) - ref.view(np.float16).astype(np.float32))
    assert ((ulps <= 4) | (absd <= 2e-3)).all(), f"max ulps {ulps.max()} max abs {absd.max()}"


def test_network_padded_planes(rig):
    """planes layout with n_el > n (the renderer pads to 128): only the first n columns are written."""
    torch = rig.torch
    n, n_el = 100, 128
    c = _rand_coords(n, 7)
    out = torch.full((16, n_el), 7.0, dtype=torch.float16, device="cuda:0")
    rig.net.inference_mixed_precision(None, torch.from_numpy(c).cuda(), out)
    got = out.cpu().numpy()
    assert (got[:, n:] == 7.0).all()
    ref = rig.scene.oracle_model.inference(c, 0).view(np.float16).astype(np.float32)
    assert np.abs(got[:, :n].astype(np.float32) - ref).max() < 5e-2


@pytest.mark.parametrize("az", [30.0, 120.0, 255.0])
def test_trace_samples_bit_exact(rig, az):
    """ray / sample indexing: every (t, dt) the marcher emits is bit-identical to the oracle's."""
    torch = rig.torch
    rig.use_edit(False)
    W, H = 160, 90
    p = rig.scene.params_for(W, H, az, snap=False, spp_index=3)
    idx = np.arange(W * H, dtype=np.uint32)
    t_ref, dt_ref, c_ref = rig.scene.oracle_model.trace_samples(p, idx, 48)
    t, dt, c = rig.testbed.trace_samples(p, torch.from_numpy(idx.astype(np.int32)).cuda(), 48)
    assert np.array_equal(c.cpu().numpy().astype(np.uint32), c_ref)
    assert c_ref.max() > 10
    assert np.array_equal(t.cpu().numpy().view(np.uint32), t_ref.view(np.uint32))
    assert np.array_equal(dt.cpu().numpy().view(np.uint32), dt_ref.view(np.uint32))


def _trace_equal(rig, scene, p, n_pixels, max_samples):
    torch = rig.torch
    idx = np.arange(n_pixels, dtype=np.uint32)
    t_ref, dt_ref, c_ref = scene.oracle_model.trace_samples(p, idx, max_samples)
    t, dt, c = rig.testbed.trace_samples(p, torch.from_numpy(idx.astype(np.int32)).cuda(), max_samples)
    assert np.array_equal(c.cpu().numpy().astype(np.uint32), c_ref)
    assert np.array_equal(t.cpu().numpy().view(np.uint32), t_ref.view(np.uint32))
    assert np.array_equal(dt.cpu().numpy().view(np.uint32), dt_ref.view(np.uint32))
    return c_ref


def test_trace_samples_bit_exact_at_1080p(rig):
    """BASELINE's full size: all 2 073 600 rays of a 1920x1080 view, first 12 samples each.  This is the regression net of the
    marching accelerator (occupied box, coarse mask look-ahead, lean walk, exact-mip bounds): any ray retired early, any
    lattice point skipped or any hand-over at the wrong step changes a count or a bit."""
    rig.use_edit(True)   # the edited occupancy: the arm is in a different place than in the plain scene
    try:
        p = rig.scene.params_for(1920, 1080, 75.0, snap=False, spp_index=2)
        c = _trace_equal(rig, rig.scene, p, 1920 * 1080, 12)
        assert (c > 0).sum() > 500000 and (c == 0).sum() > 500000
    finally:
        rig.use_edit(False)


@pytest.mark.parametrize("case", ["cropped_render_box", "min_mip_1", "camera_inside", "single_cell", "empty"])
def test_trace_samples_accelerator_corner_cases(rig, case):
    """The shortcuts must hand over correctly when the render box is smaller than the occupied box, fall back to the general
    bounds when min_mip != 0, cope with an origin inside the occupied region, a one-cell scene and an empty one."""
    scene = rig.scene
    rig.use_edit(False)
    W, H = 480, 270
    p = scene.params_for(W, H, 140.0, snap=False, spp_index=7)
    bits = scene.bitfield
    try:
        if case == "cropped_render_box":
            p.render_aabb_min[:] = [0.30, 0.25, 0.35]
            p.render_aabb_max[:] = [0.70, 0.55, 0.62]
        elif case == "min_mip_1":
            p.min_mip = 1
        elif case == "camera_inside":
            cam = np.array(p.camera_matrix1[:], np.float32)
            cam[9:12] = [0.5, 0.5, 0.5]
            p.camera_matrix0[:] = [float(v) for v in cam]
            p.camera_matrix1[:] = [float(v) for v in cam]
        elif case == "single_cell":
            bits = np.zeros_like(scene.bitfield)
            from nerfshop_amd import synth
            m = int(synth.morton3d(np.array([70], np.uint32), np.array([60], np.uint32), np.array([66], np.uint32))[0])
            bits[m // 8] = 1 << (m % 8)
        elif case == "empty":
            bits = np.zeros_like(scene.bitfield)
        rig.net.set_density_bitfield(bits)
        scene.oracle_model.set_bitfield(bits)
        c = _trace_equal(rig, scene, p, W * H, 24)
        if case == "empty":
            assert c.max() == 0
        elif case == "single_cell":
            assert 0 < (c > 0).sum() < 2000
        else:
            assert (c > 0).sum() > 5000
    finally:
        rig.use_edit(False)


def test_map_rays_bit_exact(rig):
    torch = rig.torch
    e = rig.scene.edit
    rng = np.random.default_rng(3)
    lo, hi = e.vertices.min(0) - 0.05, e.vertices.max(0) + 0.05
    n = 50000
    c = _rand_coords(n, 9)
    c[:, :3] = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)  # aabb_scale 1: warped == world
    c[: n // 4, :3] = rng.uniform(e.original_vertices.min(0) - 0.02, e.original_vertices.max(0) + 0.02, size=(n // 4, 3)).astype(np.float32)
    ref_c, ref_empty = rig.scene.oracle_edit.map_rays(c)
    dc = torch.from_numpy(c).cuda()
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    rig.op.map_rays(None, dc, mask)
    got = dc.cpu().numpy()
    assert (ref_c != c).any() and ref_empty.any()
    assert np.array_equal(got.view(np.uint32), ref_c.view(np.uint32))
    assert np.array_equal(mask.cpu().numpy(), ref_empty)
    # map_positions (no direction, ignores the copy flag)
    pos = c[:, :3].copy()
    ref_p, ref_e2 = rig.scene.oracle_edit.map_positions(pos)
    dp = torch.from_numpy(pos).cuda()
    mask2 = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    rig.op.map_positions(None, dp, mask2)
    assert np.array_equal(dp.cpu().numpy().view(np.uint32), ref_p.view(np.uint32))
    assert np.array_equal(mask2.cpu().numpy(), ref_e2)


def _compare_frames(frame, depth, steps, ref_frame, ref_depth, ref_steps):
    """RGBA tolerance: 6e-3 max abs after srgb_to_linear (slope <= 2.3), 2e-4 mean abs.  It covers (i) network outputs
    differing by a few fp16 ulps (test above), (ii) __expf vs expf in alpha, (iii) the rare ray whose accumulated alpha
    crosses 1 - min_transmittance one sample earlier/later (|d rgba| <= alpha_sample * min_transmittance ~ 1.4e-3
    before shading).  Sample counts must agree for >= 99.8 % of the pixels and never differ by more than 1."""
    d = np.abs(frame - ref_frame)
    assert d.max() < 6e-3, d.max()
    assert d.mean() < 2e-4, d.mean()
    ds = np.abs(steps.astype(np.int64) - ref_steps.astype(np.int64))
    assert ds.max() <= 1, ds.max()
    assert (ds == 0).mean() >= 0.998, (ds == 0).mean()
    same = ds == 0
    hit = (ref_frame[..., 3] > 0.2) & (frame[..., 3] > 0.2) & same
    assert np.allclose(depth[hit], ref_depth[hit], rtol=0, atol=2e-3)
    miss = (ref_frame[..., 3] == 0) & (frame[..., 3] == 0)
    assert (depth[miss] == 1e10).all() and (ref_depth[miss] == 1e10).all()


@pytest.mark.parametrize("az,snap", [(30.0, True), (200.0, False)])
def test_render_no_edit(rig, az, snap):
    rig.use_edit(False)
    p = rig.scene.params_for(256, 144, az, snap=snap, spp_index=0 if snap else 5)
    frame, depth, steps, stats = rig.render(p)
    ref_frame, ref_depth, ref_steps, ref_stats = rig.scene.oracle_model.render(p)
    assert ref_stats.n_hit > 1000
    assert stats.n_rays_alive == ref_stats.n_alive0
    _compare_frames(frame, depth, steps, ref_frame, ref_depth, ref_steps)
    assert abs(int(stats.n_samples) - int(ref_stats.composited)) <= 0.002 * ref_stats.composited
    assert abs(int(stats.n_rays_hit) - int(ref_stats.n_hit)) <= 2


def test_render_with_cage_edit(rig):
    rig.use_edit(True)
    try:
        p = rig.scene.params_for(256, 144, 60.0)
        frame, depth, steps, stats = rig.render(p)
        ref_frame, ref_depth, ref_steps, ref_stats = rig.scene.oracle_model.render(p, [rig.scene.oracle_edit])
        _compare_frames(frame, depth, steps, ref_frame, ref_depth, ref_steps)
        # the edit must actually change the picture
        p0 = rig.scene.params_for(256, 144, 60.0, apply_operators=False)
        frame0, _, _, _ = rig.render(p0)
        assert np.abs(frame0 - frame).max() > 0.01
    finally:
        rig.use_edit(False)


@pytest.mark.parametrize("poisson_target", [0, 1])
def test_render_with_membrane_correction(rig, poisson_target):
    """compute_poisson_full_residuals + the Poisson branches of composite_kernel_nerf (second, un-deformed network
    pass only where density_out_boundary > 1e-9)."""
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
        p = scene.params_for(256, 144, 60.0)
        p.poisson_target = poisson_target
        frame, depth, steps, stats = rig.render(p)
        ref_frame, ref_depth, ref_steps, ref_stats = scene.oracle_model.render(p, [o_edit])
        _compare_frames(frame, depth, steps, ref_frame, ref_depth, ref_steps)
        # the correction must be visible: compare with the same edit without membrane terms
        rig.testbed.edit_operators = saved
        plain, _, _, _ = rig.render(p)
        assert np.abs(plain - frame).max() > 0.02
    finally:
        rig.testbed.edit_operators = saved
        rig.use_edit(False)


def test_render_empty_and_degenerate(rig):
    """A camera looking away from the box: nothing alive, frame untouched, depth 1e10 everywhere."""
    rig.use_edit(False)
    cam = rig.scene.camera(30.0).copy()
    cam[6:9] *= -1.0  # flip the forward axis
    p = rig.scene.synth.render_params(64, 40, cam)
    frame, depth, steps, stats = rig.render(p)
    assert stats.n_samples == 0 and stats.n_rays_hit == 0
    assert (frame == 0).all() and (depth == 1e10).all() and (steps == 0).all()
    # ragged resolution (not a multiple of the 8x8 packet)
    p = rig.scene.params_for(61, 37, 30.0)
    frame, depth, steps, stats = rig.render(p)
    ref_frame, ref_depth, ref_steps, _ = rig.scene.oracle_model.render(p)
    _compare_frames(frame, depth, steps, ref_frame, ref_depth, ref_steps)


def test_render_accumulates_into_frame(rig):
    """shade_kernel_nerf composites over what is already in the frame buffer: frame = tmp + frame * (1 - tmp.a)."""
    rig.use_edit(False)
    torch = rig.torch
    p = rig.scene.params_for(96, 54, 30.0)
    base, _, _, _ = rig.render(p)
    frame = torch.full((54, 96, 4), 0.25, dtype=torch.float32, device="cuda:0")
    depth = torch.zeros((54, 96), dtype=torch.float32, device="cuda:0")
    rig.testbed.render_with_params(rig.net, p, frame, depth, None, None, want_stats=True)
    got = frame.cpu().numpy()
    expect = base + 0.25 * (1.0 - base[..., 3:4])
    assert np.allclose(got, expect, atol=1e-6)


def test_tiled_render_matches_whole(rig):
    """image-tile sharding: rendering tiles rank by rank into compact buffers and de-tiling reproduces the un-tiled frame
    bit for bit (per-pixel results do not depend on which rays share a launch)."""
    import ctypes as C
    rig.use_edit(False)
    torch = rig.torch
    W, H, tile, n_ranks = 200, 120, 32, 3
    p = rig.scene.params_for(W, H, 30.0)
    whole, whole_depth, _, whole_stats = rig.render(p, want_steps=False)
    lib = rig.ctx.lib
    p.tile_size, p.tile_stride = tile, n_ranks
    counts = []
    for r in range(n_ranks):
        p.tile_first = r
        counts.append(lib.nrs_render_owned_tiles(C.byref(p)))
    pad = max(counts)
    tiles = torch.zeros((n_ranks, pad, tile, tile, 4), dtype=torch.float32, device="cuda:0")
    dtiles = torch.zeros((n_ranks, pad, tile, tile), dtype=torch.float32, device="cuda:0")
    total = 0
    for r in range(n_ranks):
        p.tile_first = r
        st = rig.testbed.render_with_params(rig.net, p, tiles[r], dtiles[r], None, None, want_stats=True)
        total += st.n_samples
    image = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    dimage = torch.zeros((H, W), dtype=torch.float32, device="cuda:0")
    from nerfshop_amd._abi import check
    check(lib.nrs_detile(rig.ctx.h, None, C.byref(p), n_ranks, pad, tiles.data_ptr(), 4, 0, image.data_ptr()))
    check(lib.nrs_detile(rig.ctx.h, None, C.byref(p), n_ranks, pad, dtiles.data_ptr(), 1, 0, dimage.data_ptr()))
    torch.cuda.synchronize()
    assert total == whole_stats.n_samples
    assert np.array_equal(image.cpu().numpy(), whole)
    assert np.array_equal(dimage.cpu().numpy(), whole_depth)


def test_tile_sharder_fused_buffer_on_device(rig):
    """bench.py's N > 1 data path minus the collective: every rank renders into its [frame block | depth block] buffer
    (tiles.TileSharder), the blocks are laid out rank-major as one gather delivers them, and the strided nrs_detile
    reproduces the un-tiled frame bit for bit."""
    import ctypes as C
    from nerfshop_amd import tiles
    from nerfshop_amd._abi import check
    rig.use_edit(False)
    torch = rig.torch
    W, H, tile, world = 200, 120, 32, 3
    p = rig.scene.params_for(W, H, 30.0)
    whole, whole_depth, _, _ = rig.render(p, want_steps=False)
    shards = [tiles.TileSharder(W, H, tile, r, world, "cuda:0") for r in range(world)]
    for sh in shards:
        sh.fill(p)
        sh.clear()
        rig.testbed.render_with_params(rig.net, p, sh.local_frame, sh.local_depth, None, None)
    root = shards[0]
    for r, sh in enumerate(shards):
        root.all[r].copy_(sh.local)          # what dist.gather does
    n_px = root.padded * tile * tile
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    depth = torch.zeros((H, W), dtype=torch.float32, device="cuda:0")
    lib = rig.ctx.lib
    root.fill(p)
    check(lib.nrs_detile(rig.ctx.h, None, C.byref(p), world, root.padded, root.all.data_ptr(), 4, n_px * 5, frame.data_ptr()))
    check(lib.nrs_detile(rig.ctx.h, None, C.byref(p), world, root.padded, root.all.data_ptr() + n_px * 16, 1, n_px * 5, depth.data_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(frame.cpu().numpy(), whole)
    assert np.array_equal(depth.cpu().numpy(), whole_depth)
    # world == 1 goes through the same code (bench.py --gpus 1 does not shard, this keeps the path honest)
    one = tiles.TileSharder(W, H, tile, 0, 1, "cuda:0")
    one.fill(p)
    one.clear()
    rig.testbed.render_with_params(rig.net, p, one.local_frame, one.local_depth, None, None)
    frame.zero_(); depth.zero_()
    one.gather(rig.ctx, p, frame, depth)
    torch.cuda.synchronize()
    assert np.array_equal(frame.cpu().numpy(), whole)
    assert np.array_equal(depth.cpu().numpy(), whole_depth)


def test_concurrent_launches_on_two_streams(rig):
    """Double-buffered frames: render calls issued back to back on different streams run concurrently (each has its own packet
    counter and operator table, nrs_ctx::kInFlight) and produce exactly the frames of one-at-a-time rendering."""
    rig.use_edit(True)
    torch = rig.torch
    try:
        views = [rig.scene.params_for(320, 200, az) for az in (10.0, 100.0, 190.0, 280.0)]
        ref = [rig.render(p, want_steps=False)[0] for p in views]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        frames = [torch.zeros((200, 320, 4), dtype=torch.float32, device="cuda:0") for _ in views]
        depths = [torch.zeros((200, 320), dtype=torch.float32, device="cuda:0") for _ in views]
        torch.cuda.synchronize()
        for rep in range(3):
            for f in frames:
                f.zero_()
            torch.cuda.synchronize()
            for i, p in enumerate(views):
                st = streams[i % 2]
                with torch.cuda.stream(st):
                    rig.testbed.render_with_params(rig.net, p, frames[i], depths[i], None, st)
            torch.cuda.synchronize()
            for i in range(len(views)):
                assert np.array_equal(frames[i].cpu().numpy(), ref[i]), (rep, i)
    finally:
        rig.use_edit(False)


def test_density_grid_to_bitfield_device(rig):
    """nrs_model_set_density_grid == update_density_grid_mean_and_bitfield: bit-exact vs the oracle."""
    from oracle import oracle as orc
    try:
        rig.net.set_density_grid(rig.scene.grid)
        got = rig.net.get_density_bitfield()
        assert np.array_equal(got, orc.density_grid_to_bitfield(rig.scene.grid))
    finally:
        rig.use_edit(False)


def test_render_aabb16_with_edit(rig16):
    """garden-style configuration: aabb_scale 16, 5 cascades, cone stepping (dt grows with t), one cage edit."""
    rig16.use_edit(True)
    try:
        p = rig16.scene.params_for(192, 108, 40.0)
        assert p.cone_angle_constant > 0
        frame, depth, steps, stats = rig16.render(p)
        ref_frame, ref_depth, ref_steps, ref_stats = rig16.scene.oracle_model.render(p, [rig16.scene.oracle_edit])
        assert ref_stats.n_hit > 500
        _compare_frames(frame, depth, steps, ref_frame, ref_depth, ref_steps)
        idx = np.arange(192 * 108, dtype=np.uint32)
        t_ref, dt_ref, c_ref = rig16.scene.oracle_model.trace_samples(p, idx, 64)
        t, dt, c = rig16.testbed.trace_samples(p, rig16.torch.from_numpy(idx.astype(np.int32)).cuda(), 64)
        assert np.array_equal(c.cpu().numpy().astype(np.uint32), c_ref)
        assert np.array_equal(t.cpu().numpy().view(np.uint32), t_ref.view(np.uint32))
        assert len(np.unique(dt_ref[dt_ref > 0])) > 10  # cone stepping really varies dt
    finally:
        rig16.use_edit(False)


def test_trace_samples_aabb16_at_scale(rig16):
    """518 400 rays through the 5-cascade scene with cone stepping (general accelerator bounds, generic step loop)."""
    rig16.use_edit(True)
    try:
        p = rig16.scene.params_for(960, 540, 200.0, snap=False, spp_index=4)
        c = _trace_equal(rig16, rig16.scene, p, 960 * 540, 12)
        assert (c > 0).sum() > 100000
    finally:
        rig16.use_edit(False)


def test_render_1080p_properties(rig):
    """Full-size frame, properties that need no oracle render: every pixel's sample count is consistent with the totals, the
    frame equals the tile-sharded rendering of the same view bit for bit, and background pixels are untouched."""
    import ctypes as C
    from nerfshop_amd import tiles
    rig.use_edit(True)
    torch = rig.torch
    try:
        W, H = 1920, 1080
        p = rig.scene.params_for(W, H, 75.0)
        frame, depth, steps, stats = rig.render(p)
        assert int(steps.astype(np.int64).sum()) == int(stats.n_samples) > 20_000_000
        assert int((frame[..., 3] > 0).sum()) == int(stats.n_rays_hit) or abs(int((frame[..., 3] > 0).sum()) - int(stats.n_rays_hit)) <= 8
        bg = steps == 0
        assert (frame[bg] == 0).all() and (depth[bg] == 1e10).all()
        sh = [tiles.TileSharder(W, H, 64, r, 4, "cuda:0") for r in range(4)]
        for s_ in sh:
            s_.fill(p)
            s_.clear()
            rig.testbed.render_with_params(rig.net, p, s_.local_frame, s_.local_depth, None, None)
        for r, s_ in enumerate(sh):
            sh[0].all[r].copy_(s_.local)
        n_px = sh[0].padded * 64 * 64
        out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
        sh[0].fill(p)
        from nerfshop_amd._abi import check
        check(rig.ctx.lib.nrs_detile(rig.ctx.h, None, C.byref(p), 4, sh[0].padded, sh[0].all.data_ptr(), 4, n_px * 5, out.data_ptr()))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), frame)
    finally:
        rig.use_edit(False)


def test_render_is_deterministic(rig):
    """The persistent kernel hands packets out through an atomic queue and refills lanes in place, so WHICH wave renders a ray
    changes from run to run; every pixel must not.  30 runs of a 720p view (with the cage edit), all bit-identical."""
    rig.use_edit(True)
    try:
        p = rig.scene.params_for(1280, 720, 110.0, snap=False, spp_index=9)
        first = None
        for i in range(30):
            frame, depth, steps, stats = rig.render(p)
            if first is None:
                first = (frame, depth, steps, int(stats.n_samples))
                continue
            assert int(stats.n_samples) == first[3]
            assert np.array_equal(steps, first[2]) and np.array_equal(frame, first[0]) and np.array_equal(depth, first[1]), i
    finally:
        rig.use_edit(False)


def test_errors_are_loud(rig):
    import ctypes as C
    from nerfshop_amd import runtime, synth
    from nerfshop_amd._abi import NrsError
    torch = rig.torch
    net = runtime.NerfNetwork(rig.ctx, rig.scene.desc)
    with pytest.raises(NrsError):  # parameters not set
        net.inference_mixed_precision(None, torch.zeros((4, 7), device="cuda:0"), torch.zeros((16, 4), dtype=torch.float16, device="cuda:0"))
    with pytest.raises(NrsError):  # wrong parameter count
        net.set_params(rig.scene.params[:-2])
    with pytest.raises(NrsError):  # CPU tensor
        rig.net.inference_mixed_precision(None, torch.zeros((4, 7)), torch.zeros((16, 4), dtype=torch.float16, device="cuda:0"))
    bad = synth.model_desc(1)
    bad.n_levels = 8
    with pytest.raises(NrsError):
        runtime.NerfNetwork(rig.ctx, bad)
    p = rig.scene.params_for(32, 32, 30.0)
    p.render_mode = 10  # NumRenderModes: not a mode (Normals / EncodingVis are modes since round 4)
    with pytest.raises(NrsError):
        rig.render(p)
