"""boundary values of the membrane ("Poisson") correction, GrowingSelection::compute_poisson_boundary
(growing_selection.cu:2220-2348), through nrs_poisson_boundary against the oracle's restatement.  Directions are formed on the
host with the host libm on both sides (bit-identical inputs); the colours come out of the MFMA MLPs, so the fit carries the
network tolerance of test_network_inference_tolerance (fp16 outputs within a few ulps) averaged over 100 samples."""
import ctypes as C

import numpy as np
import pytest

from nerfshop_amd import _abi

pytestmark = pytest.mark.gpu


def _vertices(scene, n, seed):
    rng = np.random.default_rng(seed)
    v = rng.uniform(0.3, 0.7, size=(n, 3)).astype(np.float32)       # in and around the solid
    v[: n // 4] = rng.uniform(0.02, 0.98, size=(n // 4, 3))           # some far from it (empty occupancy)
    return v


@pytest.mark.parametrize("is_inside", [False, True])
def test_boundary_matches_oracle(rig, is_inside):
    rig.use_edit(False)
    n, w = 300, 10
    v = _vertices(rig.scene, n, 4)
    jitter = np.random.default_rng(9).uniform(0, 1, size=(n * w * w, 2)).astype(np.float32)
    density, sh = rig.testbed.compute_poisson_boundary(v, is_inside, jitter, w, w)
    ref_density, ref_sh, ref_coords = rig.scene.oracle_model.poisson_boundary(v, w, w, jitter, is_inside)
    # host half: identical network inputs
    coords = np.zeros_like(ref_coords)
    mn, mx = (C.c_float * 3)(*rig.scene.desc.aabb_min), (C.c_float * 3)(*rig.scene.desc.aabb_max)
    _abi.load().nrs_poisson_sample_coords(v.ctypes.data, n, w, w, jitter.ctypes.data, mn, mx, coords.ctypes.data)
    assert np.array_equal(coords.view(np.uint32), ref_coords.view(np.uint32))
    d = np.linalg.norm(coords[:, 4:7] * 2 - 1, axis=1)
    assert np.abs(d - 1).max() < 1e-5                                # unit directions covering the sphere
    assert np.abs((coords[:, 4:7] * 2 - 1).mean(0)).max() < 0.02
    # densities: exp of an fp16 network output -> relative tolerance of a few fp16 ulps; exact zeros where filter_empty strikes
    if is_inside:
        assert np.array_equal(density == 0, ref_density == 0) and (ref_density == 0).sum() > 10
    nz = ref_density > 0
    assert np.allclose(density[nz], ref_density[nz], rtol=2e-2, atol=1e-6)
    # SH fit: colours in [0, 1], 100-sample means times 4 pi
    assert np.abs(sh - ref_sh).max() < 5e-3 * max(1.0, np.abs(ref_sh).max())
    assert np.abs(ref_sh).max() > 0.1


def test_sh_fit_reproduces_a_constant_colour(rig):
    """domain property: the L00 coefficient of the fit, evaluated, gives back the mean colour around the vertex"""
    rig.use_edit(False)
    n, w = 64, 10
    v = _vertices(rig.scene, n, 6)
    jitter = np.random.default_rng(10).uniform(0, 1, size=(n * w * w, 2)).astype(np.float32)
    density, sh = rig.testbed.compute_poisson_boundary(v, False, jitter, w, w)
    # mean colour from the network directly
    coords = np.zeros((n * w * w, 7), np.float32)
    mn, mx = (C.c_float * 3)(*rig.scene.desc.aabb_min), (C.c_float * 3)(*rig.scene.desc.aabb_max)
    _abi.load().nrs_poisson_sample_coords(v.ctypes.data, n, w, w, jitter.ctypes.data, mn, mx, coords.ctypes.data)
    torch = rig.torch
    out = torch.zeros((coords.shape[0], 16), dtype=torch.float16, device="cuda:0")
    rig.net.inference_mixed_precision(None, torch.from_numpy(coords).cuda(), out)   # [n, 16]: interleaved
    raw = out.cpu().numpy().astype(np.float32)[:, :3].reshape(n, w * w, 3)
    rgb = 1.0 / (1.0 + np.exp(-raw))                                  # logistic rgb activation (nerf base config)
    mean = rgb.mean(1)
    l00 = sh.reshape(n, 3, 9)[:, :, 0] * 0.282095                     # Y00 * coefficient
    assert np.abs(l00 - mean).max() < 2e-2


def test_errors(rig):
    with pytest.raises(_abi.NrsError):
        rig.testbed.compute_poisson_boundary(np.zeros((2, 3), np.float32), False, np.zeros((5, 2), np.float32))
    v = np.zeros((1, 3), np.float32)
    with pytest.raises(_abi.NrsError):
        rig.testbed.compute_poisson_boundary(v, False, np.zeros((100 * 100, 2), np.float32), 100, 100)


def _cage_membrane_terms(rig, edit, w=8):
    """compute_poisson_boundary at the CAGE vertices, inside and outside (growing_selection.cu:2362-2368), on the device and by the oracle"""
    cv = np.ascontiguousarray(edit.cage_vertices, np.float32)
    n = cv.shape[0]
    jit = np.random.default_rng(21).uniform(0, 1, size=(n * w * w, 2)).astype(np.float32)
    dev = [rig.testbed.compute_poisson_boundary(cv, inside, jit, w, w) for inside in (True, False)]
    orc = [rig.scene.oracle_model.poisson_boundary(cv, w, w, jit, inside)[:2] for inside in (True, False)]
    return dev, orc


def test_poisson_interpolate_matches_oracle_bit_for_bit(rig):
    """nrs_edit_poisson_interpolate = GrowingSelection::interpolate_poisson_boundary (growing_selection.cu:2350-2395): the V_tet x V_cage weighted
    sums on the device in the reference's order; per-cage-vertex factors from the host libm as in the reference.  Same inputs -> same bits as the
    oracle's restatement (itself pinned to the reference's compiled loop, tests/test_ref_pin.py)."""
    from oracle import oracle as orc
    scene = rig.scene
    edit = scene.edit
    op = rig.rt.CageDeformation(rig.ctx, scene.desc, edit)
    try:
        n_cv = edit.cage_vertices.shape[0]
        rng = np.random.default_rng(5)
        inside_d = rng.uniform(0, 60, n_cv).astype(np.float32); inside_d[::4] = 0
        outside_d = rng.uniform(0.5, 80, n_cv).astype(np.float32)
        inside_s, outside_s = rng.normal(scale=0.5, size=(n_cv, 27)).astype(np.float32), rng.normal(scale=0.5, size=(n_cv, 27)).astype(np.float32)
        ref = orc.poisson_interpolate(edit.mvc_weights, inside_d, outside_d, inside_s, outside_s)
        # with explicit gamma coordinates ...
        op.poisson_interpolate(None, inside_d, outside_d, inside_s, outside_s, 0.8, gamma=edit.mvc_weights)
        got = op.download_poisson(edit.vertices.shape[0])
        for a, b in zip(got, ref):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        # ... and with the operator's own MVC weights
        op.set_mvc(edit.mvc_weights)
        op.poisson_interpolate(None, inside_d * 0.5, outside_d, inside_s, outside_s, 0.8)
        got2 = op.download_poisson(edit.vertices.shape[0])
        ref2 = orc.poisson_interpolate(edit.mvc_weights, inside_d * 0.5, outside_d, inside_s, outside_s)
        for a, b in zip(got2, ref2):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert (ref[2] > 0).sum() > 50 and np.abs(ref[0]).max() > 0.05
    finally:
        op.close()


def test_membrane_chain_end_to_end(rig):
    """cage-vertex boundary fit (nrs_poisson_boundary, inside and outside) -> nrs_edit_poisson_interpolate -> render with the membrane correction,
    against the same chain through the oracle (poisson_boundary -> poisson_interpolate -> Edit with membrane arrays -> render)."""
    import copy
    from oracle import oracle as orc
    from test_gpu_parity import _compare_frames
    scene = rig.scene
    edit = scene.edit
    rig.use_edit(True)
    op = rig.rt.CageDeformation(rig.ctx, scene.desc, edit)
    saved = rig.testbed.edit_operators
    try:
        (dev_in, dev_out), (orc_in, orc_out) = _cage_membrane_terms(rig, edit)
        assert (orc_out[0] > 0).any()
        op.poisson_interpolate(None, dev_in[0], dev_out[0], dev_in[1], dev_out[1], 0.8, gamma=edit.mvc_weights)
        sh, od, rd = orc.poisson_interpolate(edit.mvc_weights, orc_in[0], orc_out[0], orc_in[1], orc_out[1])
        got = op.download_poisson(edit.vertices.shape[0])
        assert np.abs(got[0] - sh).max() < 2e-2 * max(1.0, np.abs(sh).max()) and np.allclose(got[1], od, rtol=3e-2, atol=1e-4)   # boundary-fit tolerance carried through
        e2 = copy.copy(edit)
        e2.boundary_shs, e2.boundary_outside_density, e2.boundary_residual_density, e2.residual_amplitude = sh, od, rd, 0.8
        o_edit = orc.Edit(scene.desc, e2.tet_mesh_struct(), keepalive=e2)
        # render with the ORACLE's per-vertex terms on both sides (the fit's tolerance is checked above; here the chain's last link)
        op.poisson_interpolate(None, orc_in[0], orc_out[0], orc_in[1], orc_out[1], 0.8, gamma=edit.mvc_weights)
        rig.testbed.edit_operators = [op]
        p = scene.params_for(256, 144, 60.0)
        frame, depth, steps, stats = rig.render(p)
        ref_frame, ref_depth, ref_steps, _ = scene.oracle_model.render(p, [o_edit])
        _compare_frames(frame, depth, steps, ref_frame, ref_depth, ref_steps)
        rig.testbed.edit_operators = saved
        plain, _, _, _ = rig.render(p)
        assert np.abs(plain - frame).max() > 1e-3   # the correction is visible
    finally:
        rig.testbed.edit_operators = saved
        op.close()
        rig.use_edit(False)
