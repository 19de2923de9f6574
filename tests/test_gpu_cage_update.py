"""the per-gizmo-move chain on the device -- Cage::interpolate_with_mvc (cage.cu:38),
TetMesh::post_update_vertices (tet_mesh.cu:12), build_tet_grid (:368), update_local_rotations (:37) -- through
nrs_edit_create(device authoring) / nrs_edit_update_cage / nrs_edit_update_vertices, against the oracle's builders.

Bars: vertices after the MVC apply, the bounding box, the CSR offsets, the per-cell tet lists (ascending) and the touched-cell
bitfield are BIT-EXACT (integer work + order-controlled fp32 tests).  Rotations: bit-exact too -- device, host and oracle restate the reference's
approximate fp32 SVD (editing/tools/svd3.h) step by step, pinned by tests/golden/ref_rotations_golden.npz.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check_tables(op, scene, verts, orig_bits_expected=None):
    from nerfshop_amd import synth
    orc = scene.orc
    e = scene.edit
    got = op.download(rotations=True)
    assert np.array_equal(got["vertices"], verts)
    off, idx, _, mx = orc.tet_lut_build(verts, e.tets)
    assert np.array_equal(got["lut_offsets"], off)
    assert np.array_equal(got["lut_idx"], idx)
    assert op.lut_size() == (idx.size, mx)
    assert np.array_equal(got["bbox"], np.concatenate([verts.min(0), verts.max(0)]))
    if orig_bits_expected is not None:
        assert np.array_equal(got["original_bitfield"], orig_bits_expected)
    r_host = synth.local_rotations(verts, e.original_vertices, e.tets)
    assert np.array_equal(got["rotations"].reshape(-1), r_host.reshape(-1)), np.abs(got["rotations"].reshape(-1) - r_host.reshape(-1)).max()
    r_orc = orc.local_rotations(verts, e.original_vertices, e.tets)
    assert np.array_equal(got["rotations"].reshape(-1), r_orc.reshape(-1))   # device == oracle == the reference's svd3.h (test_oracle_kat)
    return got


def test_device_authoring_at_create(rig):
    """nrs_edit_create with only vertices + tets: LUT, canonical bitfield and rotations built on the device."""
    scene = rig.scene
    op = rig.rt.CageDeformation(rig.ctx, scene.desc, scene.edit, device_authoring=True)
    _check_tables(op, scene, scene.edit.vertices, scene.edit.original_bitfield)
    # and the host-built tables handed over the classic way download unchanged
    got = rig.op.download(rotations=True)
    assert np.array_equal(got["lut_offsets"], scene.edit.lut_offsets)
    assert np.array_equal(got["lut_idx"], scene.edit.lut_idx[: got["lut_idx"].size])
    op.close()


@pytest.mark.parametrize("via", ["cage", "vertices"])
def test_cage_moves_rebuild_on_device(rig, via):
    """A sequence of gizmo moves: each rebuild must equal a from-scratch oracle build, including moves that grow the
    LUT beyond its first allocation and a move back to the rest pose."""
    scene = rig.scene
    e, synth, orc = scene.edit, scene.synth, scene.orc
    op = rig.rt.CageDeformation(rig.ctx, scene.desc, e, device_authoring=True)
    op.set_mvc(e.mvc_weights)
    moves = [((0.02, 0.0, 0.01), 5.0), ((0.16, 0.10, -0.05), 55.0), ((0.0, 0.0, 0.0), 0.0), ((0.10, 0.05, 0.0), 20.0)]
    sizes = []
    for translate, twist in moves:
        cage_def = synth.deform_cage(e.cage_vertices, translate, twist)
        verts = orc.mvc_apply(e.mvc_weights, cage_def)
        if via == "cage":
            op.update_cage(None, cage_def)
        else:
            op.update_vertices(None, verts)
        _check_tables(op, scene, verts)
        sizes.append(op.lut_size()[0])
    assert len(set(sizes)) > 2     # the moves really changed the table
    # the last move is scene.edit's own pose: rendering through the device-built operator matches the oracle
    from test_gpu_parity import _compare_frames
    rig.use_edit(True)
    saved = rig.testbed.edit_operators
    try:
        rig.testbed.edit_operators = [op]
        p = scene.params_for(256, 144, 60.0)
        frame, depth, steps, _ = rig.render(p)
        ref_frame, ref_depth, ref_steps, _ = scene.oracle_model.render(p, [scene.oracle_edit])
        _compare_frames(frame, depth, steps, ref_frame, ref_depth, ref_steps)
    finally:
        rig.testbed.edit_operators = saved
        rig.use_edit(False)
        op.close()


@pytest.mark.parametrize("lattice", [10, 18])
def test_large_meshes_take_the_other_launch_shapes(rig, lattice):
    """The LUT passes choose their lane teams by mesh size and tet size (nrs_cage.hip launch_tet_mark: a wave per item at cascade 0 for the bench's 6 000-tet cage,
    eight lanes / one lane per item for a 35 000-tet one), cells with more than 24 tets are sorted by a wave in LDS and cells with more than 128 (the coarse
    cascades of the large mesh) by a workgroup's bitmap pass: the tables are the oracle's builder's whatever the shape, after creation and after a move."""
    scene = rig.scene
    synth, orc = scene.synth, scene.orc
    e = synth.make_cage_edit(lattice_n=lattice)
    op = rig.rt.CageDeformation(rig.ctx, scene.desc, e, device_authoring=True)
    op.set_mvc(e.mvc_weights)
    try:
        for verts in (e.vertices, orc.mvc_apply(e.mvc_weights, synth.deform_cage(e.cage_vertices, (0.05, -0.04, 0.03), 47.0))):
            if verts is not e.vertices:
                op.update_vertices(None, verts)
            got = op.download(rotations=False)
            off, idx, _, mx = orc.tet_lut_build(verts, e.tets)
            assert np.array_equal(got["lut_offsets"], off) and np.array_equal(got["lut_idx"], idx) and op.lut_size() == (idx.size, mx)
            if lattice == 18:
                assert mx > 1024   # the bitmap pass ran (lists longer than a wave's share of LDS)
            if verts is e.vertices:
                assert np.array_equal(got["original_bitfield"], e.original_bitfield)   # (built at creation from the canonical mesh by the same passes)
    finally:
        op.close()


def test_cage_update_on_aabb16(rig16):
    """Tets that span several cascades (scene box [-7.5, 8.5]^3): cells of cascades 1..4 are exercised."""
    scene = rig16.scene
    e = scene.edit
    op = rig16.rt.CageDeformation(rig16.ctx, scene.desc, e, device_authoring=True)
    got = _check_tables(op, scene, e.vertices, e.original_bitfield)
    per_level = np.diff(got["lut_offsets"][:: 128 ** 3].astype(np.int64))
    assert (per_level[1:] > 0).all()   # (the x6 cage lies outside cascade 0's unit cube except for clamped border cells)
    op.close()


def test_cage_update_errors(rig):
    from nerfshop_amd._abi import NrsError
    scene = rig.scene
    op = rig.rt.CageDeformation(rig.ctx, scene.desc, scene.edit, device_authoring=True)
    with pytest.raises(NrsError):
        op.update_cage(None, scene.edit.cage_vertices)          # no MVC weights yet
    op.set_mvc(scene.edit.mvc_weights)
    with pytest.raises(NrsError):
        op.update_cage(None, scene.edit.cage_vertices[:-1])     # wrong cage size
    with pytest.raises(NrsError):
        op.update_vertices(None, scene.edit.vertices[:-1])
    op.close()
