"""the AffineDuplication edit operator (editing/affine_duplication.h, affine_duplication.cu:69-150) behind
the same operator interface as the cage deformation.  map_rays / map_positions are order-controlled fp32 => bit-exact
against the oracle; rendering follows the renderer's stated tolerances."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _coords_around(op, n, seed):
    """positions concentrated around the selection and destination boxes (warped = world for aabb_scale 1)."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(0.0, 1.0, size=(n, 7)).astype(np.float32)
    sel = np.array(op.selection_center[:], np.float32)
    dst = sel + np.array(op.translation[:], np.float32)
    ext = np.array(op.selection_scale[:], np.float32)
    k = n // 3
    c[:k, :3] = sel + rng.uniform(-0.75, 0.75, size=(k, 3)).astype(np.float32) * ext
    c[k:2 * k, :3] = dst + rng.uniform(-0.9, 0.9, size=(k, 3)).astype(np.float32) * ext
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    c[:, 4:7] = ((d + 1.0) * 0.5).astype(np.float32)
    return c


@pytest.mark.parametrize("hide,correct_dir", [(False, True), (True, True), (True, False)])
def test_affine_map_rays_and_positions_bit_exact(rig, hide, correct_dir):
    torch = rig.torch
    scene = rig.scene
    op = scene.synth.make_affine_edit(hide_original=hide, correct_dir=correct_dir)
    dev = rig.rt.AffineDuplication(rig.ctx, scene.desc, op)
    ref = scene.orc.AffineEdit(scene.desc, op)
    c = _coords_around(op, 60001, 3)
    ref_c, ref_e = ref.map_rays(c.copy())
    t = torch.from_numpy(c).cuda()
    mask = torch.zeros(c.shape[0], dtype=torch.uint8, device="cuda:0")
    dev.map_rays(None, t, mask)
    assert np.array_equal(t.cpu().numpy().view(np.uint32), ref_c.view(np.uint32))
    assert np.array_equal(mask.cpu().numpy(), ref_e)
    moved = np.any(ref_c[:, :3] != c[:, :3], axis=1)
    assert moved.sum() > 2000 and (ref_e.sum() > 2000) == hide
    turned = np.any(ref_c[:, 4:] != c[:, 4:], axis=1)
    assert turned.any() == correct_dir and not (turned & ~moved).any()
    p = np.ascontiguousarray(c[:, :4])
    ref_p, ref_pe = ref.map_positions(p.copy())
    tp = torch.from_numpy(p).cuda()
    mask.zero_()
    dev.map_positions(None, tp, mask)
    assert np.array_equal(tp.cpu().numpy().view(np.uint32), ref_p.view(np.uint32))
    assert np.array_equal(mask.cpu().numpy(), ref_pe)
    dev.close()


def _edited_bitfield(scene, oracle_ops):
    """occupancy of the scene seen through the operators (applied last-to-first, as the tracer does)."""
    def chain(warped):
        pos = warped.copy()
        empty = np.zeros(pos.shape[0], np.uint8)
        for o in reversed(oracle_ops):
            pos, e = o.map_positions(pos)
            empty |= e
        return pos, empty
    grid = scene.synth.deformed_density_grid(scene.grid, scene.desc, chain, scene.aabb_scale)
    return scene.synth.grid_to_bitfield(grid)


@pytest.mark.parametrize("stack", ["affine", "cage+affine"])
def test_render_with_affine_duplication(rig, stack):
    from test_gpu_parity import _compare_frames
    scene = rig.scene
    op = scene.synth.make_affine_edit(hide_original=(stack == "affine"))
    dev = rig.rt.AffineDuplication(rig.ctx, scene.desc, op)
    ref = scene.orc.AffineEdit(scene.desc, op)
    dev_ops = [dev] if stack == "affine" else [rig.op, dev]
    ref_ops = [ref] if stack == "affine" else [scene.oracle_edit, ref]
    bits = _edited_bitfield(scene, ref_ops)
    saved = rig.testbed.edit_operators
    try:
        rig.testbed.edit_operators = dev_ops
        rig.net.set_density_bitfield(bits)
        scene.oracle_model.set_bitfield(bits)
        p = scene.params_for(256, 144, 60.0)
        frame, depth, steps, stats = rig.render(p)
        ref_frame, ref_depth, ref_steps, _ = scene.oracle_model.render(p, ref_ops)
        assert stats.n_rays_hit > 1000
        _compare_frames(frame, depth, steps, ref_frame, ref_depth, ref_steps)
        rig.testbed.edit_operators = [] if stack == "affine" else [rig.op]
        plain, _, _, _ = rig.render(p)
        assert np.abs(plain - frame).max() > 0.01     # the duplicate is visible
    finally:
        rig.testbed.edit_operators = saved
        rig.use_edit(False)
        dev.close()


def test_affine_operator_guards(rig):
    from nerfshop_amd._abi import NrsError
    scene = rig.scene
    op = scene.synth.make_affine_edit()
    dev = rig.rt.AffineDuplication(rig.ctx, scene.desc, op)
    with pytest.raises(NrsError):
        dev.set_mvc(np.zeros((0, 3), np.float32))
    with pytest.raises(NrsError):
        dev.download()
    bad = scene.synth.make_affine_edit(scale=(1.0, 0.0, 1.0))
    with pytest.raises(NrsError):
        rig.rt.AffineDuplication(rig.ctx, scene.desc, bad)
    dev.close()
