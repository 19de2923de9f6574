"""Known-answer tests that pin the CPU oracle: the reference's Sobol table (golden JSON generated from the reference's own
data), and hand-derived values for the in-tree formulas of SURVEY.md App. A.  No GPU."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from oracle import oracle as orc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def lib(built):
    return orc.load()


def f3(*v):
    a = np.array(v, np.float32)
    return a, a.ctypes.data


def test_sobol_matches_reference_table(lib):
    g = json.load(open(os.path.join(GOLDEN, "sobol_golden.json")))
    # the oracle GENERATES direction numbers for dims 0/1; the golden file holds the reference's tabulated ones
    for bit in range(32):
        assert lib.orc_sobol(1 << bit, 0) == g["directions_dim0"][bit]
        assert lib.orc_sobol(1 << bit, 1) == g["directions_dim1"][bit]
    for i, d, v in g["sobol"]:
        assert lib.orc_sobol(i, d) == v
    for i, s, v in g["ld_random_val"]:
        assert lib.orc_ld_random_val(i, s, 0) == np.float32(v)
    out = np.zeros(2, np.float32)
    for spp, v in g["pixel_offset"]:
        lib.orc_ld_random_pixel_offset(spp, out.ctypes.data)
        assert out[0] == np.float32(v[0]) and out[1] == np.float32(v[1])
    # snap_to_pixel_centers uses spp 0: offset is exactly the pixel centre
    lib.orc_ld_random_pixel_offset(0, out.ctypes.data)
    assert out[0] == 0.5 and out[1] == 0.5


def test_fp16_conversion_all_values(lib):
    """software fp16 <-> fp32 of the oracle against numpy for every fp16 bit pattern and for rounding boundaries."""
    bits = np.arange(65536, dtype=np.uint16)
    ref = bits.view(np.float16).astype(np.float32)
    for b in list(range(0, 65536, 97)) + [0, 1, 0x3ff, 0x400, 0x7bff, 0x7c00, 0x8000, 0x8001, 0xfbff, 0xfc00]:
        v = lib.orc_h2f(b)
        if np.isnan(ref[b]):
            assert math.isnan(v)
        else:
            assert np.float32(v) == ref[b]
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.normal(size=2000).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1.0, 100.0, 7e4)])
    # exact ties between neighbouring halfs
    h = rng.integers(0, 0x7bff, size=500).astype(np.uint16)
    lo, hi = h.view(np.float16).astype(np.float64), (h + 1).astype(np.uint16).view(np.float16).astype(np.float64)
    xs = np.concatenate([xs, ((lo + hi) / 2).astype(np.float32), np.array([65504.0, 65519.9, 65520.0, 1e9, 5.96e-8, 2.98e-8, 2.99e-8], np.float32)])
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    for x, w in zip(xs, want):
        assert lib.orc_f2h(float(x)) == w, (x, w)


def test_morton(lib):
    assert lib.orc_morton3D(1, 0, 0) == 1 and lib.orc_morton3D(0, 1, 0) == 2 and lib.orc_morton3D(0, 0, 1) == 4
    assert lib.orc_morton3D(127, 127, 127) == 128 ** 3 - 1
    for x, y, z in [(5, 3, 6), (100, 27, 64), (0, 127, 1)]:
        m = lib.orc_morton3D(x, y, z)
        assert (lib.orc_morton3D_invert(m), lib.orc_morton3D_invert(m >> 1), lib.orc_morton3D_invert(m >> 2)) == (x, y, z)
    # bit interleave by hand: x=5 (101), y=3 (011), z=6 (110): bits z2y2x2 z1y1x1 z0y0x0 = 101 110 011
    assert lib.orc_morton3D(5, 3, 6) == 0b101110011


def test_step_constants_and_dt(lib):
    mn = np.float32(1.73205080757) / np.float32(1024)
    assert lib.orc_min_step() == mn
    assert lib.orc_max_step() == mn * np.float32(128)            # STEPSIZE * 16 * 1024 / 128
    assert lib.orc_calc_dt(1.0, 0.0) == mn                        # cone 0: constant step (lego)
    assert lib.orc_calc_dt(2.0, 1.0 / 256.0) == np.float32(2.0 / 256.0)
    assert lib.orc_calc_dt(100.0, 1.0 / 256.0) == mn * np.float32(128)
    # warp_dt / unwarp_dt use MIN * 16 as the upper end (common_nerf.cu:28-36)
    assert lib.orc_warp_dt(float(mn)) == 0.0
    assert abs(lib.orc_warp_dt(float(mn * 16)) - 1.0) < 1e-6
    assert abs(lib.orc_unwarp_dt(0.5) - float(mn) * 8.5) < 1e-9


def test_mip_selection(lib):
    """mip_from_pos: frexpf(max|pos - 0.5|) -> min(4, max(0, e + 1)) (common_nerf.cu:163-168)."""
    cases = [((0.5, 0.5, 0.5), 1),      # frexpf(0) yields exponent 0 -> mip 1 (reference quirk, kept)
             ((0.6, 0.5, 0.5), 0), ((0.99, 0.5, 0.5), 0),   # 0.49 = 0.98 * 2^-1 -> e = -1
             ((1.0, 0.5, 0.5), 1),      # 0.5 = 0.5 * 2^0 -> e = 0
             ((1.49, 0.5, 0.5), 1), ((1.5, 0.5, 0.5), 2), ((0.5, -1.6, 0.5), 3), ((4.6, 0.5, 0.5), 4), ((100.0, 0.5, 0.5), 4)]
    for pos, want in cases:
        _, ptr = a = f3(*pos)
        assert lib.orc_mip_from_pos(a[1]) == want, pos
    a = f3(0.6, 0.5, 0.5)
    assert lib.orc_mip_from_dt(1.0 / 512.0, a[1]) == 0          # dt * 256 < 1 -> position mip
    assert lib.orc_mip_from_dt(1.0 / 256.0, a[1]) == 1          # dt * 256 = 1 = 0.5 * 2^1
    assert lib.orc_mip_from_dt(3.0 / 256.0, a[1]) == 2
    assert lib.orc_mip_from_dt(1.0, a[1]) == 4
    for x, want in [(0.0, 0), (1.0, 1), (0.5, 0), (0.75, 0), (3.0, 2), (1e-40, -132), (2.0 ** -126, -125)]:
        assert lib.orc_frexp_exponent(x) == want, x
        if x:
            assert math.frexp(np.float32(x))[1] == want


def test_cascaded_grid_index(lib):
    a = f3(0.5 + 0.25, 0.5, 0.5)      # level 0: cell (96, 64, 64)
    assert lib.orc_cascaded_grid_idx_at(a[1], 0) == lib.orc_morton3D(96, 64, 64)
    assert lib.orc_cascaded_grid_idx_at(a[1], 1) == lib.orc_morton3D(80, 64, 64)   # (0.25/2 + 0.5) * 128
    a = f3(-3.0, 0.5, 9.0)           # clamped to the grid
    assert lib.orc_cascaded_grid_idx_at(a[1], 0) == lib.orc_morton3D(0, 64, 127)


def test_dda(lib):
    """distance_to_next_voxel / advance_to_next_voxel (common_nerf.cu:93-115) on an axis-aligned ray."""
    pos, d = f3(0.503, 0.5, 0.5), f3(1.0, 1e-9, 1e-9)
    dist = lib.orc_distance_to_next_voxel(pos[1], d[1], 128)
    # p = 64.384 -> next boundary 65 -> (65 - 64.384) / 128
    assert abs(dist - (65 - 0.503 * 128) / 128) < 1e-6
    mn = float(lib.orc_min_step())
    t0 = 0.2
    t1 = lib.orc_advance_to_next_voxel(t0, 0.0, pos[1], d[1], 128)
    n = round((t1 - t0) / mn)
    assert n == math.ceil(dist / mn) or n == math.ceil(dist / mn) + 1   # first t >= t_target, stepping by dt
    assert t1 >= t0 + dist - 1e-7 and t1 - mn < t0 + dist + 1e-7


def test_tet_primitives(lib):
    tet = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    out = np.zeros(4, np.float32)
    p = f3(0.1, 0.2, 0.3)
    lib.orc_bary_tet(tet.ctypes.data, p[1], out.ctypes.data)
    assert np.allclose(out, [0.4, 0.1, 0.2, 0.3], atol=1e-6)
    assert lib.orc_point_in_tet(tet.ctypes.data, p[1]) == 1
    assert lib.orc_point_in_tet(tet.ctypes.data, f3(0.5, 0.5, 0.5)[1]) == 0
    assert lib.orc_point_in_tet(tet.ctypes.data, f3(-0.01, 0.2, 0.2)[1]) == 0
    # a flipped (negatively oriented) tet gives the same answers: the test compares signs, not orientation
    flipped = tet[[1, 0, 2, 3]].copy()
    assert lib.orc_point_in_tet(flipped.ctypes.data, p[1]) == 1
    box = np.array([0.2, 0.2, 0.2, 0.4, 0.4, 0.4], np.float32)
    tri_hit = np.array([0.0, 0.3, 0.3, 1.0, 0.3, 0.3, 0.3, 1.0, 0.3], np.float32)
    tri_miss = np.array([0.0, 0.0, 0.9, 1.0, 0.0, 0.9, 0.0, 1.0, 0.9], np.float32)
    tri_diag_miss = np.array([0.41, 0.2, 0.2, 0.6, 0.2, 0.39, 0.6, 0.41, 0.2], np.float32)  # bbox overlaps, plane separates
    assert lib.orc_box_intersects_triangle(box.ctypes.data, tri_hit.ctypes.data) == 1
    assert lib.orc_box_intersects_triangle(box.ctypes.data, tri_miss.ctypes.data) == 0
    assert lib.orc_box_intersects_triangle(box.ctypes.data, tri_diag_miss.ctypes.data) == 0


def test_srgb_and_sh9(lib):
    assert lib.orc_srgb_to_linear(0.04) == np.float32(0.04) / np.float32(12.92)
    assert abs(lib.orc_srgb_to_linear(1.0) - 1.0) < 1e-6
    assert abs(lib.orc_srgb_to_linear(0.5) - ((0.5 + 0.055) / 1.055) ** 2.4) < 1e-6
    sh = np.zeros(27, np.float32)
    sh[0], sh[9 + 2], sh[18 + 6] = 1.0, 1.0, 1.0   # R: Y00, G: Y10 (z), B: Y20
    rgb = np.zeros(3, np.float32)
    d = f3(0.0, 0.0, 1.0)
    lib.orc_evaluate_sh9(sh.ctypes.data, d[1], rgb.ctypes.data)
    assert np.allclose(rgb, [0.2820947917738781, 0.4886025119029199, 0.9461746957575601 - 0.3153915652525201], atol=1e-7)


def test_hashgrid_level_table_and_lookup(scene):
    """tcnn grid geometry: instant-ngp's well-known 12 196 240 encoding parameters for lego, dense levels 0-4."""
    lt = scene.synth.level_table(scene.desc)
    assert list(lt["resolution"][:6]) == [16, 23, 31, 43, 59, 81]
    assert list(lt["count"][:5]) == [4096, 12168, 29792, 79512, 205384]
    assert all(lt["count"][5:] == 2 ** 19) and list(lt["hashed"]) == [0] * 5 + [1] * 11
    assert int(lt["count"].sum()) * 2 == 12196240
    assert len(scene.params) == 3072 + 7168 + 12196240
    lib = orc.load()
    scale = np.zeros(16, np.float32)
    res, off, cnt, hashed = (np.zeros(16, np.uint32) for _ in range(4))
    assert lib.orc_model_level_table(C.byref(scene.desc), scale.ctypes.data, res.ctypes.data, off.ctypes.data, cnt.ctypes.data, hashed.ctypes.data) == 0
    assert np.array_equal(scale, lt["scale"]) and np.array_equal(off, lt["offset"]) and np.array_equal(cnt, lt["count"])
    # level 0, feature 0 is the constant 1.0 everywhere (synthetic opacity channel); a lattice point returns its entry
    feats = scene.oracle_model.hashgrid_encode(np.array([[0.3, 0.7, 0.2], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0]], np.float32)).view(np.float16)
    assert (feats[:, 0] == 1.0).all()
    grid = scene.params[3072 + 7168:].view(np.float16)
    # position whose level-0 coordinates are exact integers: x*15 + 0.5 = g + 0.5 -> weights 0.5 ... use level-0 cell centre shift:
    # pos = (g - 0.5 + 0.5) / 15 -> fractional part 0 -> only corner 0 contributes
    g = np.array([3, 7, 11])
    pos = ((g - 0.5) / 15.0).astype(np.float32)
    f = scene.oracle_model.hashgrid_encode(pos[None, :]).view(np.float16)[0]
    frac = np.float32(15.0) * pos + np.float32(0.5)
    if np.all(frac == np.floor(frac)):
        idx = int(g[0] + g[1] * 16 + g[2] * 256)
        assert f[1] == grid[2 * idx + 1]


def test_level_scale_is_evaluated_in_float(built):
    """tiny-cuda-nn evaluates the level scale in FLOAT -- exp2f(level * log2f(per_level_scale)) * base_resolution - 1.0f --
    and ceil(scale) + 1 decides every later level offset, so one ulp next to an integer silently mis-addresses a checkpoint.  Oracle and
    product follow the float formula; checked here against the host libm's exp2f / log2f called through ctypes (numpy's float32 exp2 is its own
    SIMD routine and differs from libm in the last bit) over a sweep of per_level_scale values
    (aabb scales 1..128, base resolutions 16 / 32 and values chosen so that scales land on or next to integers)."""
    from nerfshop_amd import synth
    lib = orc.load()
    libm = C.CDLL("libm.so.6")
    libm.exp2f.restype = libm.log2f.restype = C.c_float
    libm.exp2f.argtypes = libm.log2f.argtypes = [C.c_float]
    n_doubles_differ = 0
    sweep = [synth.per_level_scale(a, b) for a in (1, 2, 4, 8, 16, 32, 64, 128) for b in (16, 32)] + [1.25, 1.5, 2.0, 1.3819128, 1.4472692, 1.38191288] + \
            [float(np.float32(v)) for v in np.linspace(1.26, 2.0, 75)]
    for base in (16, 32):
        for pls in sweep:
            d = synth.model_desc(1)
            d.per_level_scale, d.base_resolution = pls, base
            lt = synth.level_table(d)
            scale = np.zeros(16, np.float32)
            res, off, cnt, hashed = (np.zeros(16, np.uint32) for _ in range(4))
            assert lib.orc_model_level_table(C.byref(d), scale.ctypes.data, res.ctypes.data, off.ctypes.data, cnt.ctypes.data, hashed.ctypes.data) == 0
            lvl = np.arange(16, dtype=np.float32)
            l2 = np.float32(libm.log2f(C.c_float(d.per_level_scale)))
            want = np.array([np.float32(libm.exp2f(C.c_float(np.float32(l) * l2))) * np.float32(base) - np.float32(1.0) for l in lvl], np.float32)
            want_res = np.ceil(want).astype(np.uint32) + 1
            assert np.array_equal(scale.view(np.uint32), want.view(np.uint32)) and np.array_equal(lt["scale"].view(np.uint32), want.view(np.uint32)), (pls, base)
            assert np.array_equal(res, want_res) and np.array_equal(lt["resolution"], want_res)
            assert np.array_equal(off, lt["offset"]) and np.array_equal(cnt, lt["count"]) and np.array_equal(hashed, lt["hashed"])
            in_double = (np.exp2(lvl.astype(np.float64) * np.log2(np.float64(np.float32(d.per_level_scale)))) * base - 1.0).astype(np.float32)
            n_doubles_differ += int((in_double.view(np.uint32) != want.view(np.uint32)).sum())
    assert n_doubles_differ > 50  # the sweep does contain levels where double evaluation lands on another float


def test_network_oracle_against_numpy(scene):
    """The C++ oracle's MLP wiring against an independent numpy evaluation (float64 accumulate, fp16 rounding points)."""
    rng = np.random.default_rng(1)
    n = 64
    c = rng.uniform(0, 1, size=(n, 7)).astype(np.float32)
    out = scene.oracle_model.inference(c, 1).view(np.float16)          # [n, 16]
    feat = scene.oracle_model.hashgrid_encode(c).view(np.float16).astype(np.float64)
    w = scene.params[:3072 + 7168].view(np.float16).astype(np.float64)
    dw1, dw2 = w[:2048].reshape(64, 32), w[2048:3072].reshape(16, 64)
    rw1, rw2, rw3 = w[3072:5120].reshape(64, 32), w[5120:9216].reshape(64, 64), w[9216:].reshape(16, 64)
    r16 = lambda x: x.astype(np.float32).astype(np.float16).astype(np.float64)
    h = r16(np.maximum(feat @ dw1.T, 0))
    dout = r16(h @ dw2.T)
    x, y, z = (c[:, 4:7].astype(np.float32) * np.float32(2) - np.float32(1)).T
    sh = np.stack([np.full(n, 0.28209479177387814, np.float32), -0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x,
                   1.0925484305920792 * x * y, -1.0925484305920792 * y * z, 0.94617469575755997 * z * z - 0.31539156525251999,
                   -1.0925484305920792 * x * z, 0.54627421529603959 * x * x - 0.54627421529603959 * y * y,
                   0.59004358992664352 * y * (-3.0 * x * x + y * y), 2.8906114426405538 * x * y * z,
                   0.45704579946446572 * y * (1.0 - 5.0 * z * z), 0.3731763325901154 * z * (5.0 * z * z - 3.0),
                   0.45704579946446572 * x * (1.0 - 5.0 * z * z), 1.4453057213202769 * z * (x * x - y * y),
                   0.59004358992664352 * x * (-x * x + 3.0 * y * y)], axis=1)
    rin = np.concatenate([dout, r16(sh)], axis=1)
    h1 = r16(np.maximum(rin @ rw1.T, 0))
    h2 = r16(np.maximum(h1 @ rw2.T, 0))
    rout = r16(h2 @ rw3.T)
    rout[:, 3] = dout[:, 0]
    got = out.astype(np.float64)
    # SH is evaluated in float32 by the oracle and float64-ish here: allow one fp16 ulp of slack on the colour rows
    assert np.abs(got - rout).max() <= 2e-2 * max(1.0, np.abs(rout).max())
    assert (got[:, 3] == rout[:, 3]).all()                      # density path has no transcendental / SH: exact
    assert (got == rout).mean() > 0.95


@pytest.mark.parametrize("kw", [dict(rgb_hidden_layers=1), dict(rgb_hidden_layers=3), dict(rgb_hidden_layers=0), dict(no_dir=True), dict(rgb_hidden_layers=0, density_hidden_layers=0)],
                         ids=["rgb1", "rgb3", "rgb0", "nodir", "linear"])
def test_network_family_oracle_against_numpy(built, kw):
    """configs/nerf/base_{0,1,3}layer.json and base_nodir.json in the oracle against an independent numpy evaluation: the rgb network's L hidden layers (or its single
    matrix, or no rgb network: colour = density-network outputs 1..3, NerfNetworkNoDir::inference_mixed_precision_impl, nerf_network_nodir.h:47-91) and the
    parameter layout [density | rgb | grid] -- also the layer numbering of visualize_activation."""
    from nerfshop_amd import synth
    d = synth.model_desc(1, **kw)
    params = synth.make_params(d)
    m = orc.Model(d, params, None)
    L, nodir = d.rgb_hidden_layers, d.sh_degree == 0
    n_rgb = 0 if nodir else (256 if L == 0 else 2048 + (L - 1) * 4096 + 1024)
    linear = d.density_hidden_layers == 0
    n_den = 512 if linear else 3072
    lt = synth.level_table(d)
    assert params.size == n_den + n_rgb + 2 * int(lt["count"].sum())
    rng = np.random.default_rng(2)
    n = 96
    c = rng.uniform(0, 1, size=(n, 7)).astype(np.float32)
    out = m.inference(c, 1).view(np.float16).astype(np.float64)
    feat = m.hashgrid_encode(c).view(np.float16).astype(np.float64)
    w = params[:n_den + n_rgb].view(np.float16).astype(np.float64)
    r16 = lambda x: x.astype(np.float32).astype(np.float16).astype(np.float64)
    if linear:  # configs/nerf/linear.json: density outputs = one [16 x 32] matrix on the features
        dout = r16(feat @ w[:512].reshape(16, 32).T)
        # its input gradient: dL/dfeatures = fp16(128 W[0]), then the grid's own derivative (checked against numpy below for the default network)
        g = m.density_input_gradient(c)
        assert np.isfinite(g).all() and np.abs(g).max() > 0
    else:
        dw1, dw2 = w[:2048].reshape(64, 32), w[2048:3072].reshape(16, 64)
        h = r16(np.maximum(feat @ dw1.T, 0))
        dout = r16(h @ dw2.T)
    assert (out[:, 3] == dout[:, 0]).all()
    if nodir:
        assert (out[:, :3] == dout[:, 1:4]).all() and (out[:, 4:] == 0).all()
        with pytest.raises(ValueError):
            m.network_activation(c, 2, 0)
        assert m.network_activation(c, 1, 63) is not None
        return
    li = 1 if linear else 2  # the rgb network's input in forward_activations' numbering
    sh = np.stack([m.network_activation(c, li, 16 + k) for k in range(16)], axis=1).astype(np.float64)  # the oracle's own SH (checked against numpy above)
    rin = np.concatenate([dout, sh], axis=1)
    r = w[n_den:]
    hidden = []
    if L == 0:
        rout = np.zeros((n, 16))
        rout[:, :8] = r16(rin @ r.reshape(8, 32).T)
    else:
        x = r16(np.maximum(rin @ r[:2048].reshape(64, 32).T, 0))
        hidden.append(x)
        off = 2048
        for _ in range(L - 1):
            x = r16(np.maximum(x @ r[off:off + 4096].reshape(64, 64).T, 0))
            hidden.append(x)
            off += 4096
        rout = r16(x @ r[off:off + 1024].reshape(16, 64).T)
    rout[:, 3] = dout[:, 0]
    assert np.abs(out - rout).max() <= 2e-2 * max(1.0, np.abs(rout).max()) and (out == rout).mean() > 0.95
    for l, x in enumerate(hidden):  # forward_activations(3 + l)
        got = m.network_activation(c, li + 1 + l, 17).astype(np.float64)
        assert (got == x[:, 17]).mean() > 0.95 and np.abs(got - x[:, 17]).max() < 2e-2 * max(1.0, np.abs(x).max())
    with pytest.raises(ValueError):  # one past the last layer
        m.network_activation(c, li + 1 + L, 0)


def test_pcg32_published_vector(built):
    """PCG32 XSH-RR known-answer vector: the output of the reference implementation's demo (pcg32-demo, seed 42, stream 54),
    as printed in the PCG C library's expected output.  tcnn::pcg32 is this generator (absent submodule; see oracle header)."""
    import ctypes as C
    lib = orc.load()
    st, inc = C.c_uint64(0), (54 << 1) | 1
    lib.orc_pcg32_next_uint(C.byref(st), inc)
    st.value = (st.value + 42) & (2 ** 64 - 1)
    lib.orc_pcg32_next_uint(C.byref(st), inc)
    got = [lib.orc_pcg32_next_uint(C.byref(st), inc) for _ in range(6)]
    assert got == [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]
    # advance(n) == n x next_uint; next_float in [0, 1) from the top 23 bits
    a, b = orc.Pcg32(1337), orc.Pcg32(1337)
    for _ in range(1000):
        a.next_uint()
    b.advance(1000)
    assert a.state.value == b.state.value
    b.advance(1 << 32)
    c = orc.Pcg32(1337)
    c.advance((1 << 32) + 1000)
    assert b.state.value == c.state.value
    r = orc.Pcg32(7)
    u = orc.Pcg32(7).next_uint()
    assert r.next_float() == np.float32(np.uint32((u >> 9) | 0x3F800000).view(np.float32) - np.float32(1.0))


def test_occupancy_refresh_oracle_properties(built):
    """update_density_grid_nerf_operator restated (tn:3533): the uniform draw of 128^3 samples visits every cell of
    cascade 0 exactly once ((i * 56924617 + c) mod 2^21 is a bijection), the refreshed occupancy is the network's solid,
    a cage edit moves it, and the Testbed-side state advances."""
    from nerfshop_amd import _abi, synth
    from conftest import Scene
    sc = Scene(1, True, 6, shaped=True)
    VOL = 128 ** 3

    def new_update(seed=1337):
        u = _abi.GridUpdate()
        u.n_uniform_samples, u.n_nonuniform_samples, u.reset_grid, u.max_cascade, u.decay, u.ema_step = VOL, 0, 1, 0, 0.95, 0
        r = orc.Pcg32(seed)
        u.rng_state, u.rng_inc = r.state.value, r.inc.value
        return u

    grid = np.full(5 * VOL, 3.0, np.float32)
    u = new_update()
    bits = sc.oracle_model.update_density_grid(grid, u, [])
    assert u.ema_step == 1
    r = orc.Pcg32(1337)
    r.advance(2 << 32)
    assert u.rng_state == r.state.value
    assert (grid[:VOL] > 0).all() and (grid[VOL:] == 0).all()        # reset + every cell of cascade 0 written once
    occ = np.unpackbits(bits[: VOL // 8], bitorder="little").astype(bool)
    solid = np.unpackbits(sc.bitfield[: VOL // 8], bitorder="little").astype(bool)
    assert 0.02 * VOL < occ.sum() < solid.sum()                        # trilinear shoulder erodes the border cells
    assert (occ & ~solid).sum() < 0.02 * occ.sum()                     # essentially inside the solid
    inside = grid[:VOL][occ]
    assert abs(np.median(inside) - 0.15) < 0.03                        # exp(sigma_raw) * dt_min, synth.default_sigma_raw
    # untrained cells (negative) stay negative and are never chosen by either draw
    grid2 = grid.copy()
    grid2[:1000] = -1.0
    u2 = new_update()
    u2.reset_grid = 0
    sc.oracle_model.update_density_grid(grid2, u2, [])
    assert (grid2[:1000] == -1.0).all()
    # with the cage edit the occupancy moves: cells appear where the deformed cage carries content
    grid3 = np.zeros(5 * VOL, np.float32)
    u3 = new_update()
    bits3 = sc.oracle_model.update_density_grid(grid3, u3, [sc.oracle_edit])
    occ3 = np.unpackbits(bits3[: VOL // 8], bitorder="little").astype(bool)
    moved_ref = np.unpackbits(sc.edited_bitfield[: VOL // 8], bitorder="little").astype(bool)
    gained, lost = occ3 & ~occ, occ & ~occ3
    assert gained.sum() > 500 and lost.sum() > 500
    assert (gained & ~moved_ref).sum() < 0.1 * gained.sum()            # new cells lie where the analytic deformed solid is
    sc.oracle_model.set_bitfield(sc.bitfield)


def test_mvc_pinned_to_the_reference_code(built):
    """MVC3D::computeCoordinatesCustomCode (include/neural-graphics-primitives/editing/tools/mvc.h) compiled from the reference into
    oracle/_ref/libref_render.so (point_t = Eigen::Vector3f against the Eigen stand-in) and run on the test cage by
    tests/golden/make_ref_mvc_golden.py.  Oracle and product reproduce its weights (550 points incl. cage
    vertices, points on cage faces, points outside the cage) and its success labels."""
    import os
    from nerfshop_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = np.load(os.path.join(root, "tests", "golden", "ref_mvc_golden.npz"))
    for name, fn in (("oracle", orc.mvc_compute), ("product", synth.mvc_weights)):
        w, labels = fn(g["cage_vertices"], g["cage_triangles"], g["points"])
        assert np.array_equal(labels, g["labels"]), name
        assert np.abs(w - g["weights"]).max() <= 1e-6, (name, np.abs(w - g["weights"]).max())
    assert sorted(np.bincount(g["labels"]).tolist()) == [3, 547]
    ref_lib = os.path.join(root, "oracle", "_ref", "libref_render.so")
    if os.path.exists(ref_lib):   # this container: the fixture is what the reference code produces today
        import importlib.util
        spec = importlib.util.spec_from_file_location("make_ref_mvc_golden", os.path.join(root, "tests", "golden", "make_ref_mvc_golden.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        w, labels = mod.ref_mvc(g["cage_vertices"], g["cage_triangles"], g["points"])
        assert np.array_equal(w, g["weights"]) and np.array_equal(labels, g["labels"])


def test_local_rotations_pinned_to_the_reference_code(built):
    """TetMesh::update_local_rotations uses the approximate McAdams SVD of editing/tools/svd3.h; oracle/ref_render.cpp compiles that
    header and the reference's loop in place and tests/golden/make_ref_rotations_golden.py records its R = U V^T for 2592 tets (a mild and a harsh
    deformation).  The oracle's restatement and the product's (host here, device in tests/test_gpu_cage_update.py) match bit for bit."""
    import os
    from nerfshop_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = np.load(os.path.join(root, "tests", "golden", "ref_rotations_golden.npz"))
    want = g["rotations"].view(np.uint32)
    r_orc = orc.local_rotations(g["vertices"], g["original_vertices"], g["tets"]).reshape(-1, 9)
    r_lib = synth.local_rotations(g["vertices"], g["original_vertices"], g["tets"]).reshape(-1, 9)
    assert np.array_equal(r_orc.view(np.uint32), want)
    assert np.array_equal(r_lib.view(np.uint32), want)
    # the procedure is approximate by design: close to, but not, the exact polar rotation
    v = g["vertices"][g["tets"]].astype(np.float64)
    o = g["original_vertices"][g["tets"]].astype(np.float64)
    A = np.einsum("nji,njk->nik", o - o.mean(1, keepdims=True), v - v.mean(1, keepdims=True))
    U, _, Vt = np.linalg.svd(A)
    exact = (U @ Vt).transpose(0, 2, 1).reshape(-1, 9)      # column-major like the fixture
    err = np.abs(exact - g["rotations"]).max(1)
    assert np.median(err) < 1e-5 and 1e-3 < err.max() < 5e-2
    ref_lib = os.path.join(root, "oracle", "_ref", "libref_render.so")
    if os.path.exists(ref_lib):
        import importlib.util
        spec = importlib.util.spec_from_file_location("make_ref_rotations_golden", os.path.join(root, "tests", "golden", "make_ref_rotations_golden.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        assert np.array_equal(mod.ref_rotations(g["vertices"], g["original_vertices"], g["tets"]).view(np.uint32), want)


def test_fast_flavour_stays_within_the_network_tolerance(built):
    """Model.set_fast (bench.py's cpu_baseline leg only): hardware half conversions and fp32-accumulated MLP sums instead of the checker's exact ones.
    Same algorithm and rounding points: hash-grid features identical, network outputs within the tolerance the GPU kernels are held to (<= 4 fp16
    ulps or 2e-3 abs), nearly all identical; and a frame rendered with it is the checker's frame within the renderer's colour bar."""
    from nerfshop_amd import synth
    from oracle import oracle as orc
    desc = synth.model_desc(1)
    params = synth.make_params(desc, sigma_raw=synth.default_sigma_raw(1))
    model = orc.Model(desc, params, synth.grid_to_bitfield(synth.density_grid(1)))
    c = np.random.default_rng(3).uniform(0, 1, (20000, 7)).astype(np.float32)
    a = model.inference(c, 0)
    p = synth.render_params(96, 54, synth.orbit_camera(30.0), aabb_scale=1, apply_operators=False)
    fa, da, sa, _ = model.render(p, [])
    model.set_fast(True)
    b = model.inference(c, 0)
    fb, db, sb, _ = model.render(p, [])
    af, bf = a.view(np.float16).astype(np.float32), b.view(np.float16).astype(np.float32)
    ulp = np.maximum(np.abs(af), 2.0 ** -14) * 2.0 ** -10
    assert ((np.abs(af - bf) <= 4 * ulp) | (np.abs(af - bf) <= 2e-3)).all() and (a == b).mean() > 0.98
    assert np.abs(fa - fb).max() < 6e-3 and np.abs(sa.astype(np.int64) - sb.astype(np.int64)).max() <= 1


def _np_grid_features_and_jacobian(scene, pos):
    """float64 trilinear hash-grid features [n, 32] and their Jacobian w.r.t. the position [n, 32, 3] from the parameter blob, with tcnn's index
    function restated in numpy (an evaluation independent of oracle/nrs_oracle.cpp)."""
    lt = scene.synth.level_table(scene.desc)
    grid = scene.params[3072 + 7168:].view(np.float16).astype(np.float64).reshape(-1, 2)
    n = pos.shape[0]
    feat, jac = np.zeros((n, 32)), np.zeros((n, 32, 3))
    for l in range(16):
        scale, res, off, cnt, hashed = np.float32(lt["scale"][l]), int(lt["resolution"][l]), int(lt["offset"][l]), int(lt["count"][l]), int(lt["hashed"][l])
        p = (scale * pos.astype(np.float32) + np.float32(0.5)).astype(np.float64)
        g = np.floor(p).astype(np.int64)
        w = p - g
        for c in range(8):
            bit = np.array([(c >> d) & 1 for d in range(3)])
            gl = (g + bit).astype(np.uint64)
            if hashed:
                idx = ((gl[:, 0] * np.uint64(1)) ^ (gl[:, 1] * np.uint64(2654435761)) ^ (gl[:, 2] * np.uint64(805459861))) & np.uint64(0xffffffff)
            else:
                idx = gl[:, 0] + gl[:, 1] * np.uint64(res) + gl[:, 2] * np.uint64(res * res)
            val = grid[off + (idx % np.uint64(cnt)).astype(np.int64)]            # [n, 2]
            wd = np.where(bit[None, :] == 1, w, 1.0 - w)                         # [n, 3]
            feat[:, 2 * l:2 * l + 2] += wd.prod(axis=1)[:, None] * val
            for d in range(3):
                others = [k for k in range(3) if k != d]
                dw = (1.0 if bit[d] else -1.0) * float(scale) * wd[:, others[0]] * wd[:, others[1]]
                jac[:, 2 * l:2 * l + 2, d] += dw[:, None] * val
    return feat, jac


def test_density_input_gradient_against_numpy(scene_shaped):
    """Render mode Normals rests on tcnn's input_gradient, restated in the oracle (density_input_gradient_one): checked here against the chain rule
    evaluated independently in float64 numpy -- d density_raw / d x = W2[0] . (relu'(W1 f(x)) * (W1 J_f(x))) -- which pins the wiring (one-hot on output
    row 3 = density row 0, transposed matrices, the level scale in the feature Jacobian, the 1 / 128 un-scaling); the oracle's own fp16 rounding points
    (dL/dhidden, dL/dfeatures) account for the tolerance."""
    scene = scene_shaped
    rng = np.random.default_rng(7)
    n = 600
    c = np.zeros((n, 7), np.float32)
    c[:, :3] = rng.uniform(0.15, 0.85, size=(n, 3))
    c[:, 4:] = 0.5
    got = scene.oracle_model.density_input_gradient(c).astype(np.float64)
    feat, jac = _np_grid_features_and_jacobian(scene, c[:, :3])
    assert np.abs(feat - scene.oracle_model.hashgrid_encode(c).view(np.float16).astype(np.float64)).max() < 2e-3   # the numpy grid is the oracle's grid
    w = scene.params[:3072].view(np.float16).astype(np.float64)
    w1, w2 = w[:2048].reshape(64, 32), w[2048:3072].reshape(16, 64)
    pre = feat @ w1.T                                                    # [n, 64]
    # (the ReLU mask of the oracle's own fp16 hidden layer: a unit within rounding of its kink must not flip between the two evaluations)
    mask = np.stack([scene.oracle_model.network_activation(c, 1, k) > 0 for k in range(64)], axis=1)
    assert (mask == (pre > 0)).mean() > 0.995
    want = np.einsum("k,nk,nkd->nd", w2[0], mask.astype(np.float64), np.einsum("ki,nid->nkd", w1, jac))
    safe = np.ones(n, bool)
    scale = np.abs(want[safe]).max()
    assert scale > 1.0
    err = np.abs(got[safe] - want[safe]).max(axis=1)
    assert np.median(err) < 2e-3 * scale and err.max() < 2e-2 * scale, (np.median(err) / scale, err.max() / scale)
    # both accumulation models of the backward GEMM stay within that tolerance of each other
    scene.oracle_model.set_numerics(1, 1)
    try:
        got16 = scene.oracle_model.density_input_gradient(c).astype(np.float64)
    finally:
        scene.oracle_model.set_numerics(0, 0)
    assert np.abs(got16[safe] - want[safe]).max() < 5e-2 * scale and not np.array_equal(got16, got)


def test_network_activation_layers(scene):
    """Render mode EncodingVis rests on tcnn's visualize_activation, restated in the oracle (network_activation_one): layer 0 is the hash-grid output, layer 2
    the rgb network's input [density outputs | SH], hidden layers are non-negative (after ReLU); unknown units are refused."""
    rng = np.random.default_rng(3)
    c = rng.uniform(0.05, 0.95, size=(200, 7)).astype(np.float32)
    m = scene.oracle_model
    feat = m.hashgrid_encode(c).view(np.float16).astype(np.float32)
    dens = m.density(c[:, :3].copy(), 1).view(np.float16).astype(np.float32)   # [n, 16]
    sh = np.zeros((200, 16), np.uint16)
    dirs = np.ascontiguousarray(c[:, 4:7])
    orc.load().orc_sh4_encode(200, dirs.ctypes.data, 3, sh.ctypes.data)
    for dim in (0, 5, 31):
        assert np.array_equal(m.network_activation(c, 0, dim), feat[:, dim])
    for dim in (0, 3, 15):
        assert np.array_equal(m.network_activation(c, 2, dim), dens[:, dim])
        assert np.array_equal(m.network_activation(c, 2, 16 + dim), sh.view(np.float16).astype(np.float32)[:, dim])
    for layer in (1, 3, 4):
        a = m.network_activation(c, layer, 17)
        assert (a >= 0).all() and (a > 0).any()
    for layer, dim in ((0, 32), (2, 32), (1, 64), (5, 0)):
        with pytest.raises(ValueError):
            m.network_activation(c, layer, dim)
