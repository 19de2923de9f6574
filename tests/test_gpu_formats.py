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


def test_render_from_the_reference_written_edits_file(rig, tmp_path):
    """The edits file the REFERENCE's own writer produced (tests/golden/ref_edits_golden.json.gz: Testbed::save_edits compiled from /root/reference, operator 0 = this scene's
    cage deformation) drives the renderer: loaded through nrs_edits_open, tables built on the device, the frame equals the one rendered from the in-memory edit bit for bit."""
    import gzip
    import os
    from nerfshop_amd import _abi, formats, runtime
    scene = rig.scene
    path = tmp_path / "ref_edits.json"
    path.write_bytes(gzip.open(os.path.join(os.path.dirname(__file__), "golden", "ref_edits_golden.json.gz"), "rb").read())
    ops = formats.load_edits(path)
    assert np.array_equal(ops[0].vertices, scene.edit.vertices) and np.array_equal(ops[0].tets, scene.edit.tets)   # (the golden was made from this scene's edit)
    torch = rig.torch
    p = scene.params_for(256, 144, 60.0)

    def render_with(operators):
        tb = runtime.Testbed(rig.ctx, scene.desc, 1)
        tb.nerf_network.set_params(scene.params)
        tb.nerf_network.set_density_bitfield(scene.edited_bitfield)
        for o in operators:
            tb.add_edit_operator(o)
        frame = torch.zeros((144, 256, 4), dtype=torch.float32, device="cuda:0")
        depth = torch.zeros((144, 256), dtype=torch.float32, device="cuda:0")
        tb.render_with_params(tb.nerf_network, p, frame, depth, None, None, want_stats=True)
        torch.cuda.synchronize()
        assert tb.last_stats.n_rays_hit > 1000
        return frame.cpu().numpy(), depth.cpu().numpy()

    from_file = render_with([runtime.CageDeformation(rig.ctx, scene.desc, ops[0], device_authoring=True)])
    from_memory = render_with([runtime.CageDeformation(rig.ctx, scene.desc, scene.edit)])
    assert np.array_equal(from_file[0], from_memory[0]) and np.array_equal(from_file[1], from_memory[1])
    # the file's affine duplication is accepted by the device operator as it comes out of the reader (its effect on pictures: tests/test_gpu_affine.py)
    assert isinstance(ops[1], _abi.AffineDuplicationOp)
    both = render_with([runtime.CageDeformation(rig.ctx, scene.desc, ops[0], device_authoring=True), runtime.AffineDuplication(rig.ctx, scene.desc, ops[1])])
    assert np.isfinite(both[0]).all()
