"""Files -> device: a scene saved in the reference's snapshot (.ingp) and edits (.json) formats, loaded through
nrs_snapshot_open / nrs_edits_open, with the operator's tables built on the device, renders bit-identically to the same
scene handed over from memory with host-built tables."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_render_from_files_is_bit_identical(rig, tmp_path):
    from nerfshop_amd import formats, runtime
    scene = rig.scene
    formats.save_snapshot(tmp_path / "scene.ingp", scene.desc, 1, scene.params, scene.edited_grid, camera=scene.camera(60.0))
    formats.save_edits(tmp_path / "edits.json", [scene.edit])
    snap = formats.load_snapshot(tmp_path / "scene.ingp")
    tb = runtime.Testbed(rig.ctx, snap.desc, snap.aabb_scale)
    tb.nerf_network.set_params(snap.params)
    tb.nerf_network.set_density_grid(snap.density_grid)
    ops = formats.load_edits(tmp_path / "edits.json")
    op = runtime.CageDeformation(rig.ctx, snap.desc, ops[0], device_authoring=True)
    tb.add_edit_operator(op)
    p = scene.synth.render_params(256, 144, snap.camera)
    torch = rig.torch
    frame = torch.zeros((144, 256, 4), dtype=torch.float32, device="cuda:0")
    depth = torch.zeros((144, 256), dtype=torch.float32, device="cuda:0")
    tb.render_with_params(tb.nerf_network, p, frame, depth, None, None, want_stats=True)
    torch.cuda.synchronize()
    assert tb.last_stats.n_rays_hit > 1000
    rig.use_edit(True)
    try:
        ref_frame, ref_depth, _, _ = rig.render(scene.params_for(256, 144, 60.0))
    finally:
        rig.use_edit(False)
    assert np.array_equal(frame.cpu().numpy(), ref_frame)
    assert np.array_equal(depth.cpu().numpy(), ref_depth)
    # a cage move through the loaded MVC weights, then back: same picture again
    op.set_mvc(ops[0].mvc_weights)
    op.update_cage(None, ops[0].cage_vertices)       # rest pose
    op.update_cage(None, ops[0].cage_deformed)       # the saved pose
    frame2 = torch.zeros_like(frame)
    tb.render_with_params(tb.nerf_network, p, frame2, depth, None, None)
    torch.cuda.synchronize()
    assert np.array_equal(frame2.cpu().numpy(), ref_frame)
