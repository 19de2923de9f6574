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
