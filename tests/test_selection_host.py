"""Host-only bookkeeping of the selection tool (get_upper_cell_idx, the cell list of project_selection_pixels)."""
import numpy as np

from nerfshop_amd import _abi, synth


def test_upper_cell_idx(built):
    """get_upper_cell_idx (selection_utils.cu:36): cell (x, y, z) of cascade l covers cell (x/2+32, ...) of cascade l+1"""
    lib = _abi.load()
    rng = np.random.default_rng(2)
    vol = 128 ** 3
    for _ in range(200):
        x, y, z = (int(v) for v in rng.integers(0, 128, 3))
        l0 = int(rng.integers(0, 4))
        l1 = int(rng.integers(l0, 5))
        cell = l0 * vol + int(synth.morton3d(np.array([x]), np.array([y]), np.array([z]))[0])
        ux, uy, uz = x, y, z
        for _ in range(l1 - l0):
            ux, uy, uz = ux // 2 + 32, uy // 2 + 32, uz // 2 + 32
        want = l1 * vol + int(synth.morton3d(np.array([ux]), np.array([uy]), np.array([uz]))[0])
        assert lib.nrs_upper_cell_idx(cell, l1) == want


def test_selection_cells_lifts_to_the_highest_cascade(built):
    import ctypes as C
    lib = _abi.load()
    vol = 128 ** 3
    m = lambda x, y, z: int(synth.morton3d(np.array([x]), np.array([y]), np.array([z]))[0])
    cells = np.array([m(10, 20, 30), vol + m(37, 42, 47), m(11, 21, 31), 0, m(100, 100, 100)], np.uint32)
    found = np.array([1, 1, 1, 0, 1], np.uint8)
    pos = np.arange(15, dtype=np.float32).reshape(5, 3)
    out_c, out_p, n_out, level = np.zeros(5, np.uint32), np.zeros((5, 3), np.float32), C.c_uint32(), C.c_uint32(0)
    _abi.check(lib.nrs_selection_cells(pos.ctypes.data, cells.ctypes.data, found.ctypes.data, 5, 1, C.byref(level), out_c.ctypes.data,
                                       out_p.ctypes.data, C.byref(n_out)))
    # cascade 1 is the highest found; (10,20,30) and (11,21,31) of cascade 0 both lift to (37,42,47) of cascade 1 = ray 1's cell
    assert level.value == 1 and n_out.value == 2
    assert out_c[:2].tolist() == [vol + m(37, 42, 47), vol + m(82, 82, 82)]
    assert np.array_equal(out_p[:2], pos[[0, 4]])
    # fixed level 0: the cascade-1 ray is dropped (growing_selection.cu:1995-1998)
    level = C.c_uint32(0)
    _abi.check(lib.nrs_selection_cells(pos.ctypes.data, cells.ctypes.data, found.ctypes.data, 5, 0, C.byref(level), out_c.ctypes.data,
                                       out_p.ctypes.data, C.byref(n_out)))
    assert n_out.value == 3 and out_c[:3].tolist() == [m(10, 20, 30), m(11, 21, 31), m(100, 100, 100)]


def test_poisson_sample_coords_match_oracle(scene):
    """host half of nrs_poisson_boundary (growing_selection.cu:2241-2261): the network inputs, bit for bit the oracle's"""
    import ctypes as C
    n, w = 50, 10
    rng = np.random.default_rng(12)
    v = rng.uniform(0.1, 0.9, size=(n, 3)).astype(np.float32)
    jitter = rng.uniform(0, 1, size=(n * w * w, 2)).astype(np.float32)
    coords = np.zeros((n * w * w, 7), np.float32)
    mn, mx = (C.c_float * 3)(*scene.desc.aabb_min), (C.c_float * 3)(*scene.desc.aabb_max)
    _abi.load().nrs_poisson_sample_coords(v.ctypes.data, n, w, w, jitter.ctypes.data, mn, mx, coords.ctypes.data)
    _, _, ref = scene.oracle_model.poisson_boundary(v, w, w, jitter, False)
    assert np.array_equal(coords.view(np.uint32), ref.view(np.uint32))
    d = coords[:, 4:7] * 2 - 1
    assert np.abs(np.linalg.norm(d, axis=1) - 1).max() < 1e-5
    # stratification: cell (i, j) of the 10 x 10 grid in (u, v) holds exactly one sample per vertex
    u = (d[:, 2] + 1) / 2                     # z = cos(phi) = 2u - 1
    th = np.mod(np.arctan2(d[:, 1], d[:, 0]), 2 * np.pi) / (2 * np.pi)
    cells = (np.minimum((u * w).astype(int), w - 1) * w + np.minimum((th * w).astype(int), w - 1)).reshape(n, w * w)
    assert (np.sort(cells, axis=1) == np.arange(w * w)).mean() > 0.97   # (rounding at cell borders aside)
