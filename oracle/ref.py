"""ctypes wrapper of oracle/_ref/libref_render.so: the REFERENCE's own render-path sources compiled as host code.

TEST INFRASTRUCTURE ONLY.  The library exists only where /root/reference is mounted (this container, never the GPU box); it pins
oracle/nrs_oracle.cpp (tests/test_ref_pin.py) and generates the fixtures tests/golden/ref_*.npz (tests/golden/make_ref_*.py).
See oracle/ref_render.cpp for what is compiled from the reference and what is a stand-in.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libref_render.so")
_lib = None

NETWORK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int)


class RefRenderStats(C.Structure):
    _fields_ = [("generated", C.c_uint64), ("composited", C.c_uint64), ("n_alive0", C.c_uint32), ("n_hit", C.c_uint32),
                ("iterations", C.c_uint32), ("pad", C.c_uint32)]


def available():
    return os.path.exists(LIB_PATH)


def use_fma_build(on):
    """Switch to oracle/_ref/libref_render_fma.so: the same reference sources compiled with -ffp-contract=fast (a model of nvcc's default FMA
    contraction; measurement only, tools/fma_report.py)."""
    global _lib, LIB_PATH
    _lib = None
    LIB_PATH = os.path.join(_HERE, "_ref", "libref_render_fma.so" if on else "libref_render.so")


def load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB_PATH)
        for fn in ("ref_render_frame", "ref_accumulate", "ref_set_introspection", "ref_trace_coords", "ref_edit_map_rays", "ref_edit_map_positions", "ref_edit_poisson_residuals", "ref_build_tet_grid",
                   "ref_grid_to_bitfield", "ref_bary_tet", "ref_point_in_tet", "ref_ld_random_val", "ref_ld_random_pixel_offset", "ref_sobol", "ref_ray_intersect",
                   "ref_box_intersects_triangle", "ref_grid_math", "ref_warp", "ref_evaluate_sh9", "ref_activations", "ref_pixel_to_ray", "ref_cell_functions", "ref_local_rotations", "ref_mvc_compute", "ref_mvc_apply", "ref_poisson_interpolate", "ref_affine_map_rays", "ref_affine_map_positions"):
            getattr(_lib, fn).restype = None
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def _u32(a):
    return np.ascontiguousarray(a, np.uint32)


def render_frame(desc, params, bitfield, meshes, oracle_model, frame=None, want_steps=True):
    """Testbed::render_nerf with the reference's kernels; the network is the oracle's (tiny-cuda-nn is not in the reference checkout)."""
    lib = load()
    from . import oracle as orc
    olib = orc.load()
    W, H = params.resolution[0], params.resolution[1]
    frame = np.zeros((H, W, 4), np.float32) if frame is None else np.ascontiguousarray(frame, np.float32)
    depth = np.zeros((H, W), np.float32)
    steps = np.zeros((H, W), np.uint32) if want_steps else None
    stats = RefRenderStats()
    bf = np.ascontiguousarray(bitfield, np.uint8)
    arr = (C.c_void_p * max(len(meshes), 1))(*[C.cast(C.pointer(m), C.c_void_p) for m in meshes])
    net = C.cast(olib.orc_network_inference, C.c_void_p)
    lib.ref_set_introspection(C.cast(olib.orc_density_input_gradient7, C.c_void_p), C.cast(olib.orc_visualize_activation7, C.c_void_p))  # tcnn's input_gradient / visualize_activation: the oracle's restatements
    lib.ref_render_frame(C.byref(desc), C.byref(params), _p(bf), arr, C.c_int(len(meshes)), net, C.c_void_p(oracle_model.h), _p(frame), _p(depth), _p(steps), C.byref(stats))
    return frame, depth, steps, stats


def accumulate(frame, accum, sample_count, color_space=0):
    """CudaRenderBuffer::accumulate with the reference's accumulate_kernel; `accum` [H, W, 4] f32 is updated in place."""
    f = _f32(frame)
    assert accum.dtype == np.float32 and accum.flags.c_contiguous and accum.shape == f.shape
    load().ref_accumulate(C.c_int(f.shape[1]), C.c_int(f.shape[0]), _p(f), _p(accum), C.c_uint32(int(sample_count)), C.c_int(int(color_space)))
    return accum


def update_density_grid(desc, meshes, grid, update, oracle_model):
    """Testbed::update_density_grid_nerf_operator with the reference's kernels (density network = the oracle's); grid [5*128^3] is updated in
    place, `update` (nrs_grid_update) advances as the reference advances m_rng / density_grid_ema_step."""
    lib = load()
    from . import oracle as orc
    olib = orc.load()
    g = grid
    assert g.dtype == np.float32 and g.flags.c_contiguous and g.size == 5 * 128 ** 3
    arr = (C.c_void_p * max(len(meshes), 1))(*[C.cast(C.pointer(m), C.c_void_p) for m in meshes])
    net = C.cast(olib.orc_network_density, C.c_void_p)
    lib.ref_update_density_grid(C.byref(desc), arr, C.c_int(len(meshes)), C.byref(update), _p(g), net, C.c_void_p(oracle_model.h))
    return g


def density_on_grid(desc, res3d, box_min, box_max, oracle_model, density_grid=None):
    """Testbed::get_density_on_grid with the reference's generator and conversion kernels (density network = the oracle's)."""
    lib = load()
    from . import oracle as orc
    olib = orc.load()
    res = (C.c_uint32 * 3)(*res3d)
    mn, mx = (C.c_float * 3)(*box_min), (C.c_float * 3)(*box_max)
    out = np.zeros(int(res3d[0]) * int(res3d[1]) * int(res3d[2]), np.float32)
    g = np.ascontiguousarray(density_grid, np.float32) if density_grid is not None else None
    lib.ref_density_on_grid(C.byref(desc), res, mn, mx, _p(g), C.cast(olib.orc_network_density, C.c_void_p), C.c_void_p(oracle_model.h), _p(out))
    return out


def rgba_on_grid(desc, res3d, box_min, box_max, ray_dir, oracle_model):
    """Testbed::get_rgba_on_grid with the reference's generator and compute_nerf_density (network = the oracle's)."""
    lib = load()
    from . import oracle as orc
    olib = orc.load()
    res = (C.c_uint32 * 3)(*res3d)
    mn, mx, rd = (C.c_float * 3)(*box_min), (C.c_float * 3)(*box_max), (C.c_float * 3)(*ray_dir)
    out = np.zeros((int(res3d[0]) * int(res3d[1]) * int(res3d[2]), 4), np.float32)
    lib.ref_rgba_on_grid(C.byref(desc), res, mn, mx, rd, C.cast(olib.orc_network_inference, C.c_void_p), C.c_void_p(oracle_model.h), _p(out))
    return out


def poisson_boundary(desc, vertices, sh_width, hemisphere_width, seed, is_inside, bitfield, oracle_model):
    """GrowingSelection::compute_poisson_boundary with the reference's loops and kernels (network = the oracle's) after srand(seed)
    -> (density [n], sh [n, 27], jitter [n * sh_width^2, 2] = the std::rand draws the reference consumed)."""
    lib = load()
    from . import oracle as orc
    olib = orc.load()
    v = _f32(vertices)
    n = v.shape[0]
    dens, sh = np.zeros(n, np.float32), np.zeros((n, 27), np.float32)
    jitter = np.zeros((n * sh_width * sh_width, 2), np.float32)
    bf = np.ascontiguousarray(bitfield, np.uint8)
    lib.ref_poisson_boundary(C.byref(desc), _p(v), C.c_uint32(n), C.c_uint32(sh_width), C.c_uint32(hemisphere_width), C.c_uint(seed), C.c_int(1 if is_inside else 0), _p(bf),
                             C.cast(olib.orc_network_inference, C.c_void_p), C.c_void_p(oracle_model.h), _p(dens), _p(sh), _p(jitter))
    return dens, sh, jitter


def upper_cell_idx(cell_idx, target_level):
    lib = load()
    c, t = _u32(cell_idx), _u32(target_level)
    out = np.zeros(c.size, np.uint32)
    lib.ref_upper_cell_idx(C.c_uint32(c.size), _p(c), _p(t), _p(out))
    return out


def project_selection_pixels(desc, params, bitfield, pixels_xy, oracle_model, threshold=0.1):
    """GrowingSelection::project_selection_pixels up to the host bookkeeping, with the reference's kernels (density network = the oracle's):
    per pixel the projected position, its grid cell and whether the transmittance threshold was reached."""
    lib = load()
    from . import oracle as orc
    olib = orc.load()
    px = np.ascontiguousarray(pixels_xy, np.int32).reshape(-1, 2)
    n = px.shape[0]
    pos, cells, found = np.zeros((n, 3), np.float32), np.zeros(n, np.uint32), np.zeros(n, np.uint8)
    bf = np.ascontiguousarray(bitfield, np.uint8)
    net = C.cast(olib.orc_network_density, C.c_void_p)
    lib.ref_project_selection_pixels(C.byref(desc), C.byref(params), _p(bf), _p(px), C.c_uint32(n), C.c_float(threshold), net, C.c_void_p(oracle_model.h), _p(pos), _p(cells), _p(found))
    return pos, cells, found


def trace_coords(desc, params, bitfield, pixel_idx, max_samples, which="ref"):
    fn = _fn(which, "trace_coords")
    px = _u32(pixel_idx)
    n = px.size
    coords = np.zeros((n, max_samples, 7), np.float32)
    t_after = np.zeros((n, max_samples), np.float32)
    cnt = np.zeros(n, np.uint32)
    odt = np.zeros((n, 7), np.float32)
    bf = np.ascontiguousarray(bitfield, np.uint8)
    fn(C.byref(desc), C.byref(params), _p(bf), C.c_uint32(n), _p(px), C.c_uint32(max_samples), _p(coords), _p(t_after), _p(cnt), _p(odt))
    return coords, t_after, cnt, odt


def edit_map_rays(desc, mesh, coords7):
    lib = load()
    c = np.array(coords7, np.float32, copy=True)
    empty = np.zeros(c.shape[0], np.uint8)
    lib.ref_edit_map_rays(C.byref(desc), C.byref(mesh), C.c_uint32(c.shape[0]), _p(c), _p(empty))
    return c, empty


def edit_map_positions(desc, mesh, pos3):
    lib = load()
    c = np.array(pos3, np.float32, copy=True)
    assert c.shape[1] == 3
    empty = np.zeros(c.shape[0], np.uint8)
    lib.ref_edit_map_positions(C.byref(desc), C.byref(mesh), C.c_uint32(c.shape[0]), _p(c), _p(empty))
    return c, empty


def edit_poisson_residuals(desc, mesh, coords7):
    lib = load()
    c = _f32(coords7)
    n = c.shape[0]
    sh, od, rd = np.zeros((n, 27), np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    lib.ref_edit_poisson_residuals(C.byref(desc), C.byref(mesh), C.c_uint32(n), _p(c), _p(sh), _p(od), _p(rd))
    return sh, od, rd


def build_tet_grid(vertices, original_vertices, tets):
    lib = load()
    v, o, t = _f32(vertices), _f32(original_vertices), _u32(tets)
    n_cells = 5 * 128 ** 3
    offsets = np.zeros(n_cells + 1, np.uint32)
    bitfield = np.zeros(n_cells // 8, np.uint8)
    n_idx, mx = C.c_uint32(0), C.c_uint32(0)
    cap = 1 << 22
    while True:
        idx = np.zeros(cap, np.uint32)
        lib.ref_build_tet_grid(_p(v), _p(o), C.c_uint32(v.shape[0]), _p(t), C.c_uint32(t.shape[0]), _p(offsets), _p(idx), C.c_uint32(cap), C.byref(n_idx), _p(bitfield),
                               C.byref(mx))
        if n_idx.value <= cap:
            return offsets, idx[:n_idx.value].copy(), bitfield, mx.value
        cap = int(n_idx.value)


def grid_to_bitfield(grid, mean):
    lib = load()
    g = _f32(grid)
    out = np.zeros(5 * 128 ** 3 // 8, np.uint8)
    lib.ref_grid_to_bitfield(_p(g), C.c_float(mean), _p(out))
    return out


# ---- element-wise probes: which = "ref" (the reference's code, libref_render.so) or "orc" (the oracle's restatement, orc_p_*) -------
def _fn(which, name):
    if which == "ref":
        return getattr(load(), "ref_" + name)
    from . import oracle as orc
    fn = getattr(orc.load(), "orc_p_" + name)
    fn.restype = None
    return fn


def bary_tet(abcd, p, which="ref"):
    a, q = _f32(abcd).reshape(-1, 12), _f32(p).reshape(-1, 3)
    out = np.zeros((a.shape[0], 4), np.float32)
    _fn(which, "bary_tet")(C.c_uint32(a.shape[0]), _p(a), _p(q), _p(out))
    return out


def point_in_tet(abcd, p, which="ref"):
    a, q = _f32(abcd).reshape(-1, 12), _f32(p).reshape(-1, 3)
    out = np.zeros(a.shape[0], np.uint8)
    _fn(which, "point_in_tet")(C.c_uint32(a.shape[0]), _p(a), _p(q), _p(out))
    return out


def ld_random_val(index, seed, which="ref"):
    i, s = _u32(index), _u32(seed)
    out = np.zeros(i.size, np.float32)
    _fn(which, "ld_random_val")(C.c_uint32(i.size), _p(i), _p(s), _p(out))
    return out


def ld_random_pixel_offset(spp, which="ref"):
    s = _u32(spp)
    out = np.zeros((s.size, 2), np.float32)
    _fn(which, "ld_random_pixel_offset")(C.c_uint32(s.size), _p(s), _p(out))
    return out


def sobol(index, dim, which="ref"):
    i = _u32(index)
    out = np.zeros(i.size, np.uint32)
    _fn(which, "sobol")(C.c_uint32(i.size), _p(i), C.c_uint32(dim), _p(out))
    return out


def ray_intersect(box6, o, d, which="ref"):
    b, o, d = _f32(box6).reshape(-1, 6), _f32(o).reshape(-1, 3), _f32(d).reshape(-1, 3)
    out, inside = np.zeros((o.shape[0], 2), np.float32), np.zeros(o.shape[0], np.uint8)
    _fn(which, "ray_intersect")(C.c_uint32(o.shape[0]), _p(b), _p(o), _p(d), _p(out), _p(inside))
    return out, inside


def box_intersects_triangle(box6, tri9, which="ref"):
    b, t = _f32(box6).reshape(-1, 6), _f32(tri9).reshape(-1, 9)
    out = np.zeros(b.shape[0], np.uint8)
    _fn(which, "box_intersects_triangle")(C.c_uint32(b.shape[0]), _p(b), _p(t), _p(out))
    return out


def grid_math(pos, direction, t, cone, mip, which="ref"):
    pos, direction, t, cone, mip = _f32(pos), _f32(direction), _f32(t), _f32(cone), _u32(mip)
    n = t.size
    dt, mfp, mfd, cell, dist, adv = (np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.uint32), np.zeros(n, np.float32),
                                      np.zeros(n, np.float32))
    _fn(which, "grid_math")(C.c_uint32(n), _p(pos), _p(direction), _p(t), _p(cone), _p(mip), _p(dt), _p(mfp), _p(mfd), _p(cell), _p(dist), _p(adv))
    return {"calc_dt": dt, "mip_from_pos": mfp, "mip_from_dt": mfd, "cell_idx": cell, "distance_to_next_voxel": dist, "advance_to_next_voxel": adv}


def warp(box6, pos, dt, which="ref"):
    b, pos, dt = _f32(box6), _f32(pos), _f32(dt)
    n = dt.size
    outs = [np.zeros((n, 3), np.float32) for _ in range(4)] + [np.zeros(n, np.float32) for _ in range(2)]
    _fn(which, "warp")(C.c_uint32(n), _p(b), _p(pos), _p(dt), *[_p(o) for o in outs])
    return dict(zip(["warp_position", "unwarp_position", "warp_direction", "unwarp_direction", "warp_dt", "unwarp_dt"], outs))


def evaluate_sh9(sh27, direction, which="ref"):
    sh, d = _f32(sh27).reshape(-1, 27), _f32(direction).reshape(-1, 3)
    out = np.zeros((sh.shape[0], 3), np.float32)
    _fn(which, "evaluate_sh9")(C.c_uint32(sh.shape[0]), _p(sh), _p(d), _p(out))
    return out


def activations(x, which="ref"):
    x = _f32(x)
    outs = [np.zeros(x.size, np.float32) for _ in range(4)]
    _fn(which, "activations")(C.c_uint32(x.size), _p(x), *[_p(o) for o in outs])
    return dict(zip(["srgb_to_linear", "rgb_logistic", "rgb_exponential", "density_exponential"], outs))


def pixel_to_ray(pixels, params, which="ref"):
    px = np.ascontiguousarray(pixels, np.int32).reshape(-1, 2)
    o, d = np.zeros((px.shape[0], 3), np.float32), np.zeros((px.shape[0], 3), np.float32)
    _fn(which, "pixel_to_ray")(C.c_uint32(px.shape[0]), _p(px), C.byref(params), _p(o), _p(d))
    return o, d


def cell_functions(xyz_level, pos, which="ref"):
    q, pos = _u32(xyz_level).reshape(-1, 4), _f32(pos).reshape(-1, 3)
    cp, ca = np.zeros((q.shape[0], 3), np.float32), np.zeros((q.shape[0], 3), np.int32)
    _fn(which, "cell_functions")(C.c_uint32(q.shape[0]), _p(q), _p(pos), _p(cp), _p(ca))
    return cp, ca


def local_rotations(vertices, original, tets):
    lib = load(); v, o, t = _f32(vertices), _f32(original), _u32(tets)
    out = np.zeros((t.shape[0], 9), np.float32)
    lib.ref_local_rotations(_p(v), _p(o), C.c_uint32(v.shape[0]), _p(t), C.c_uint32(t.shape[0]), _p(out))
    return out


def mvc_compute(cage_v, cage_t, points):
    lib = load(); cv, tr, pts = _f32(cage_v), _u32(cage_t), _f32(points)
    w, labels = np.zeros((pts.shape[0], cv.shape[0]), np.float32), np.zeros(pts.shape[0], np.uint8)
    lib.ref_mvc_compute(_p(cv), C.c_uint32(cv.shape[0]), _p(tr), C.c_uint32(tr.shape[0]), _p(pts), C.c_uint32(pts.shape[0]), _p(w), _p(labels))
    return w, labels


def mvc_apply(weights, cage_v):
    lib = load(); w, cv = _f32(weights), _f32(cage_v)
    out = np.zeros((w.shape[0], 3), np.float32)
    lib.ref_mvc_apply(_p(w), _p(cv), C.c_uint32(cv.shape[0]), C.c_uint32(w.shape[0]), _p(out))
    return out


def affine_map_rays(desc, op, coords7):
    lib = load()
    c = np.array(coords7, np.float32, copy=True)
    empty = np.zeros(c.shape[0], np.uint8)
    lib.ref_affine_map_rays(C.byref(desc), C.byref(op), C.c_uint32(c.shape[0]), _p(c), _p(empty))
    return c, empty


def affine_map_positions(desc, op, pos3):
    lib = load()
    c = np.array(pos3, np.float32, copy=True)
    assert c.shape[1] == 3
    empty = np.zeros(c.shape[0], np.uint8)
    lib.ref_affine_map_positions(C.byref(desc), C.byref(op), C.c_uint32(c.shape[0]), _p(c), _p(empty))
    return c, empty


def poisson_interpolate(gamma, inside_density, outside_density, inside_shs, outside_shs):
    lib = load()
    g = _f32(gamma)
    i_d, o_d, i_s, o_s = _f32(inside_density), _f32(outside_density), _f32(inside_shs).reshape(-1, 27), _f32(outside_shs).reshape(-1, 27)
    n = g.shape[0]
    sh, od, rd = np.zeros((n, 27), np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    lib.ref_poisson_interpolate(_p(g), C.c_uint32(n), C.c_uint32(g.shape[1]), _p(i_d), _p(o_d), _p(i_s), _p(o_s), _p(sh), _p(od), _p(rd))
    return sh, od, rd
