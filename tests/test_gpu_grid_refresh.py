"""the deformed-space occupancy refresh (Testbed::update_density_grid_nerf_operator,
testbed_nerf.cu:3533) through nrs_model_update_density_grid, against the oracle's restatement.

Bars: sample generation (pcg32, cell choice, in-cell position), the cage warp and the max-splat cell are integer /
order-controlled fp32 work -> the SET of cells written and the rng / ema state are bit-exact.  The value written is
fp16(act(fp16 density-MLP output)) * dt_min: the MLP output carries the network tolerance (MFMA fp32 accumulation order
vs the oracle's exact sum: at most 1 fp16 ulp, i.e. 2^-8 absolute on raw values in [4, 8) => 0.4 % on exp()), so grid
values are compared with rtol 1e-2 and >= 99 % of the cells must be bit-identical; a bitfield bit may differ only where the
oracle's value lies within that tolerance of the threshold.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

VOL = 128 ** 3


def _oracle_update(scene, u_gpu_template, edits, grid):
    from nerfshop_amd import _abi
    u = _abi.GridUpdate()
    for f, _ in _abi.GridUpdate._fields_:
        setattr(u, f, getattr(u_gpu_template, f))
    bits = scene.oracle_model.update_density_grid(grid, u, edits)
    return u, bits


def _compare(rig, scene, u, u_ref, ref_grid, ref_bits, n_cascades):
    got_grid = rig.net.get_density_grid()
    got_bits = rig.net.get_density_bitfield()
    assert (u.ema_step, u.rng_state, u.rng_inc) == (u_ref.ema_step, u_ref.rng_state, u_ref.rng_inc)
    n = VOL * n_cascades
    # cells beyond the sampled cascades stay untouched
    assert np.array_equal(got_grid[n:], ref_grid[n:])
    g, r = got_grid[:n], ref_grid[:n]
    assert np.array_equal(g == 0, r == 0), "different set of cells written"
    same = (g.view(np.uint32) == r.view(np.uint32)).mean()
    assert same >= 0.99, f"only {same:.4f} of the cells bit-identical"
    np.testing.assert_allclose(g, r, rtol=1e-2, atol=1e-7)
    diff = np.unpackbits(got_bits, bitorder="little") != np.unpackbits(ref_bits, bitorder="little")
    if diff.any():
        thresh = scene.orc.load().orc_density_grid_threshold(ref_grid.ctypes.data)
        bad0 = np.nonzero(diff[:VOL])[0]  # cascade 0 is a pure threshold test (coarser cascades also receive pooled bits)
        assert np.all(np.abs(ref_grid[bad0] - thresh) <= 1.5e-2 * thresh), "bitfield differs away from the threshold"
        assert diff.sum() <= 64
    return got_grid, got_bits


@pytest.mark.parametrize("with_edit", [False, True])
def test_grid_refresh_matches_oracle(rig_shaped, with_edit):
    rig, scene = rig_shaped, rig_shaped.scene
    tb = rig.testbed
    tb.edit_operators = [rig.op] if with_edit else []
    try:
        u = tb.new_grid_update(max_cascade=0, seed=1337)
        ref_grid = np.zeros(5 * VOL, np.float32)
        # two iterations, the first with reset (update_density_grid_nerf_render(2, true))
        for it in range(2):
            u.reset_grid = 1 if it == 0 else 0
            u_ref, ref_bits = _oracle_update(scene, u, [scene.oracle_edit] if with_edit else [], ref_grid)
            tb.update_density_grid_nerf_operator(u)
            got_grid, got_bits = _compare(rig, scene, u, u_ref, ref_grid, ref_bits, 1)
        occ = np.unpackbits(got_bits[: VOL // 8]).sum()
        assert 0.02 * VOL < occ < 0.08 * VOL  # the solid, not "everything" and not "nothing"
        if with_edit:
            # the edit must have moved occupancy: differs from the un-edited refresh
            tb.edit_operators = []
            u2 = tb.new_grid_update(max_cascade=0, seed=1337)
            u2.reset_grid = 1
            tb.update_density_grid_nerf_operator(u2)
            plain = rig.net.get_density_bitfield()
            assert (np.unpackbits(plain[: VOL // 8]) != np.unpackbits(got_bits[: VOL // 8])).sum() > 1000
    finally:
        tb.edit_operators = []
        rig.use_edit(False)


def test_grid_refresh_nonuniform_and_decay(rig_shaped):
    """Second draw (threshold NERF_MIN_OPTICAL_THICKNESS, rng advanced by 2^32 between the draws), ragged sample counts
    (not multiples of 64) and the decay branch: cells not re-sampled keep prev * decay."""
    rig, scene = rig_shaped, rig_shaped.scene
    tb = rig.testbed
    tb.edit_operators = []
    u = tb.new_grid_update(max_cascade=0, seed=99)
    u.reset_grid = 1
    ref_grid = np.zeros(5 * VOL, np.float32)
    u_ref, ref_bits = _oracle_update(scene, u, [], ref_grid)
    tb.update_density_grid_nerf_operator(u)
    _compare(rig, scene, u, u_ref, ref_grid, ref_bits, 1)
    u.reset_grid = 0
    u.n_uniform_samples = 100003
    u.n_nonuniform_samples = 200001
    u.decay = 0.5
    before = ref_grid.copy()
    u_ref, ref_bits = _oracle_update(scene, u, [], ref_grid)
    tb.update_density_grid_nerf_operator(u)
    got_grid, _ = _compare(rig, scene, u, u_ref, ref_grid, ref_bits, 1)
    decayed = got_grid[:VOL] == before[:VOL] * np.float32(0.5)
    assert 0.5 < decayed.mean() < 1.0
    rig.use_edit(False)


def test_grid_refresh_membrane_residual(rig_shaped):
    """compute_poisson_residual_density (cage_deformation.cu:645): the fp16 residual add, including negative sums whose
    bit pattern wins the unsigned atomicMax (the reference's behaviour, restated)."""
    from nerfshop_amd import runtime
    from oracle import oracle as orc
    rig, scene = rig_shaped, rig_shaped.scene
    edit = scene.edit.with_membrane(residual_amplitude=0.8)
    op = runtime.CageDeformation(rig.ctx, scene.desc, edit)
    o_edit = orc.Edit(scene.desc, edit.tet_mesh_struct(), keepalive=edit)
    tb = rig.testbed
    tb.edit_operators = [op]
    try:
        u = tb.new_grid_update(max_cascade=0, seed=5)
        u.reset_grid = 1
        ref_grid = np.zeros(5 * VOL, np.float32)
        u_ref, ref_bits = _oracle_update(scene, u, [o_edit], ref_grid)
        tb.update_density_grid_nerf_operator(u)
        _compare(rig, scene, u, u_ref, ref_grid, ref_bits, 1)
    finally:
        tb.edit_operators = []
        rig.use_edit(False)


def test_grid_refresh_all_cascades(rig16_shaped):
    """aabb_scale 16: max_cascade = 4, 5 * 128^3 samples in one launch, with the cage edit."""
    rig, scene = rig16_shaped, rig16_shaped.scene
    tb = rig.testbed
    tb.edit_operators = [rig.op]
    try:
        u = tb.new_grid_update(max_cascade=4, seed=1337)
        u.reset_grid = 1
        ref_grid = np.zeros(5 * VOL, np.float32)
        u_ref, ref_bits = _oracle_update(scene, u, [scene.oracle_edit], ref_grid)
        tb.update_density_grid_nerf_operator(u)
        _compare(rig, scene, u, u_ref, ref_grid, ref_bits, 5)
    finally:
        tb.edit_operators = []
        rig.use_edit(False)


def test_render_after_refresh(rig_shaped):
    """End to end: edit -> refresh on the device -> render; the oracle does the same on its side."""
    from test_gpu_parity import _compare_frames
    rig, scene = rig_shaped, rig_shaped.scene
    tb = rig.testbed
    tb.edit_operators = [rig.op]
    try:
        u = tb.new_grid_update(max_cascade=0, seed=1337)
        ref_grid = np.zeros(5 * VOL, np.float32)
        for it in range(2):
            u.reset_grid = 1 if it == 0 else 0
            u_ref, ref_bits = _oracle_update(scene, u, [scene.oracle_edit], ref_grid)
            tb.update_density_grid_nerf_operator(u)
        got_bits = rig.net.get_density_bitfield()
        if not np.array_equal(got_bits, ref_bits):
            # threshold-straddling cells (see module docstring): render both sides on the device's occupancy
            scene.oracle_model.set_bitfield(got_bits)
        p = scene.params_for(256, 144, 60.0)
        frame, depth, steps, stats = rig.render(p)
        ref_frame, ref_depth, ref_steps, _ = scene.oracle_model.render(p, [scene.oracle_edit])
        assert stats.n_rays_hit > 1000
        _compare_frames(frame, depth, steps, ref_frame, ref_depth, ref_steps)
    finally:
        tb.edit_operators = []
        rig.use_edit(False)


def test_grid_refresh_errors(rig_shaped):
    from nerfshop_amd._abi import NrsError
    tb = rig_shaped.testbed
    u = tb.new_grid_update(max_cascade=0)
    u.max_cascade = 5
    with pytest.raises(NrsError):
        tb.update_density_grid_nerf_operator(u)


def test_density_and_rgba_on_grid(rig_shaped):
    """Testbed::get_density_on_grid / get_rgba_on_grid (testbed_nerf.cu:4538 / :4588) against the oracle: the grid point
    positions are integer / order-controlled fp32 (so the density-grid mask is bit-exact: the same cells become -10000); the
    network values carry the network tolerance (<= 2 fp16 ulps of the raw outputs)."""
    rig, scene = rig_shaped, rig_shaped.scene
    tb = rig.testbed
    grid = scene.grid
    rig.net.set_density_grid(grid)
    try:
        res = (48, 40, 33)   # ragged on purpose (not multiples of 64)
        mn, mx = (0.1, 0.15, 0.2), (0.9, 0.8, 0.85)
        got = tb.get_density_on_grid(res, mn, mx).cpu().numpy().reshape(-1)
        ref = scene.oracle_model.density_on_grid(res, mn, mx, grid)
        assert np.array_equal(got == -10000.0, ref == -10000.0)
        live = ref != -10000.0
        assert 1000 < live.sum() < live.size
        ulp = np.maximum(np.abs(ref[live]), 2.0 ** -14) * 2.0 ** -10
        assert (np.abs(got[live] - ref[live]) <= 2 * ulp).all()
        assert (got[live] == ref[live]).mean() > 0.97
        unmasked = tb.get_density_on_grid(res, mn, mx, mask_with_density_grid=False).cpu().numpy().reshape(-1)
        assert (unmasked != -10000.0).all() and np.array_equal(unmasked[live], got[live])
        rgba = tb.get_rgba_on_grid((32, 32, 32), (0.3, -0.5, 0.8)).cpu().numpy().reshape(-1, 4)
        ref_rgba = scene.oracle_model.rgba_on_grid((32, 32, 32), tb.render_aabb[0], tb.render_aabb[1], (0.3, -0.5, 0.8))
        assert np.abs(rgba - ref_rgba).max() < 4e-3
        assert rgba[:, 3].max() > 0.5 and rgba[:, 3].min() < 1e-3      # solid inside, (almost) nothing outside
    finally:
        rig.use_edit(False)
