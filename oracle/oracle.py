"""ctypes wrapper of the CPU oracle (oracle/nrs_oracle.cpp).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from nerfshop_amd/.
Pinned to the reference's compiled code (oracle/ref.py, tests/test_ref_pin.py) except at the tiny-cuda-nn boundary, where parity is
UNPINNED and the two ambiguous roundings are switchable (Model.set_numerics; see the .cpp header).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnrs_oracle.so")
_lib = None


class OrcRenderStats(C.Structure):
    _fields_ = [("generated", C.c_uint64), ("composited", C.c_uint64), ("n_alive0", C.c_uint32), ("n_hit", C.c_uint32),
                ("iterations", C.c_uint32), ("pad", C.c_uint32)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    P, U32, F, I = C.c_void_p, C.c_uint32, C.c_float, C.c_int
    sig = {
        "orc_f2h": (C.c_uint16, [F]), "orc_h2f": (F, [C.c_uint16]),
        "orc_morton3D": (U32, [U32, U32, U32]), "orc_morton3D_invert": (U32, [U32]),
        "orc_sobol": (U32, [U32, U32]), "orc_ld_random_val": (F, [U32, U32, U32]), "orc_ld_random_pixel_offset": (None, [U32, P]),
        "orc_mip_from_pos": (I, [P]), "orc_mip_from_dt": (I, [F, P]), "orc_cascaded_grid_idx_at": (U32, [P, U32]),
        "orc_calc_dt": (F, [F, F]), "orc_min_step": (F, []), "orc_max_step": (F, []), "orc_warp_dt": (F, [F]), "orc_unwarp_dt": (F, [F]),
        "orc_distance_to_next_voxel": (F, [P, P, U32]), "orc_advance_to_next_voxel": (F, [F, F, P, P, U32]),
        "orc_frexp_exponent": (I, [F]), "orc_srgb_to_linear": (F, [F]), "orc_evaluate_sh9": (None, [P, P, P]),
        "orc_bary_tet": (None, [P, P, P]), "orc_point_in_tet": (I, [P, P]), "orc_box_intersects_triangle": (I, [P, P]),
        "orc_model_n_params": (C.c_size_t, [P]), "orc_model_level_table": (I, [P, P, P, P, P, P]),
        "orc_model_create": (P, [P, P, C.c_size_t, P]), "orc_model_set_bitfield": (None, [P, P]), "orc_model_destroy": (None, [P]),
        "orc_hashgrid_encode": (None, [P, U32, P, U32, P]), "orc_sh4_encode": (None, [U32, P, U32, P]),
        "orc_network_inference": (None, [P, U32, P, P, U32, I]), "orc_network_density": (None, [P, U32, P, U32, P, U32, I]),
        "orc_edit_create": (P, [P, P]), "orc_edit_create_affine": (P, [P, P]), "orc_edit_destroy": (None, [P]),
        "orc_edit_map_rays": (None, [P, U32, P, P]), "orc_edit_map_positions": (None, [P, U32, P, U32, P]),
        "orc_render": (None, [P, P, P, I, P, P, P, P, I, I]), "orc_max_threads": (I, []),
        "orc_trace_samples": (None, [P, P, U32, P, U32, P, P, P]),
        "orc_tet_lut_build": (P, [P, U32, P, U32]), "orc_tet_lut_n_idx": (U32, [P]), "orc_tet_lut_max_per_cell": (U32, [P]),
        "orc_tet_lut_copy": (None, [P, P, P, P]), "orc_tet_lut_destroy": (None, [P]),
        "orc_mvc_compute": (None, [P, U32, P, U32, P, U32, P, P]), "orc_mvc_apply": (None, [P, P, U32, U32, P]),
        "orc_tet_local_rotations": (None, [P, P, P, U32, P]),
        "orc_density_grid_to_bitfield": (None, [P, P]), "orc_density_grid_threshold": (F, [P]),
        "orc_pcg32_seed": (None, [C.c_uint64, P, P]), "orc_pcg32_next_uint": (U32, [P, C.c_uint64]),
        "orc_pcg32_next_float": (F, [P, C.c_uint64]), "orc_pcg32_advance": (None, [P, C.c_uint64, C.c_uint64]),
        "orc_update_density_grid": (None, [P, P, I, P, P, P]),
        "orc_density_on_grid": (None, [P, P, P, P, P, P]), "orc_rgba_on_grid": (None, [P, P, P, P, P, P]),
        "orc_project_selection_pixels": (None, [P, P, P, U32, F, P, P, P]),
        "orc_poisson_boundary": (None, [P, P, U32, U32, U32, P, I, P, P, P]),
        "orc_accumulate": (None, [I, I, P, P, U32, I]),
        "orc_density_input_gradient": (None, [P, U32, P, P]), "orc_network_activation": (I, [P, U32, P, U32, U32, P]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


class Model:
    """NerfNetwork + occupancy on the CPU."""

    def __init__(self, desc, params_u16, bitfield=None):
        self.lib = load()
        self.desc = desc
        params_u16 = np.ascontiguousarray(params_u16, np.uint16)
        bf = np.ascontiguousarray(bitfield, np.uint8) if bitfield is not None else None
        self.h = self.lib.orc_model_create(C.byref(desc), params_u16.ctypes.data, params_u16.size, bf.ctypes.data if bf is not None else None)
        if not self.h:
            raise ValueError("oracle: unsupported model description or wrong parameter count")

    def set_fast(self, on=True):
        """bench.py's cpu_baseline flavour (F16C conversions, fp32-accumulated MLP sums): for TIMING only, never as the checker."""
        self.lib.orc_model_set_fast.restype = None
        self.lib.orc_model_set_fast(C.c_void_p(self.h), C.c_int(1 if on else 0))

    def set_numerics(self, grid_acc=0, mlp_acc=0):
        self.lib.orc_model_set_numerics.restype = None
        self.lib.orc_model_set_numerics(C.c_void_p(self.h), C.c_uint32(grid_acc), C.c_uint32(mlp_acc))

    def set_bitfield(self, bitfield):
        bf = np.ascontiguousarray(bitfield, np.uint8)
        self.lib.orc_model_set_bitfield(self.h, bf.ctypes.data)

    def hashgrid_encode(self, pos):
        pos = _f32(pos)
        out = np.zeros((pos.shape[0], 32), np.uint16)
        self.lib.orc_hashgrid_encode(self.h, pos.shape[0], pos.ctypes.data, pos.shape[1], out.ctypes.data)
        return out

    def inference(self, coords7, layout=0):
        coords7 = _f32(coords7)
        n = coords7.shape[0]
        out = np.zeros((16, n) if layout == 0 else (n, 16), np.uint16)
        self.lib.orc_network_inference(self.h, n, coords7.ctypes.data, out.ctypes.data, n, layout)
        return out

    def density(self, pos, layout=0):
        pos = _f32(pos)
        n = pos.shape[0]
        out = np.zeros((16, n) if layout == 0 else (n, 16), np.uint16)
        self.lib.orc_network_density(self.h, n, pos.ctypes.data, pos.shape[1], out.ctypes.data, n, layout)
        return out

    def density_input_gradient(self, coords7):
        """tcnn input_gradient(stream, 3, ...) as restated: d density_raw / d warped position, [n, 3] f32 (render mode Normals)."""
        coords7 = _f32(coords7)
        out = np.zeros((coords7.shape[0], 3), np.float32)
        self.lib.orc_density_input_gradient(self.h, coords7.shape[0], coords7.ctypes.data, out.ctypes.data)
        return out

    def network_activation(self, coords7, layer, dim):
        """tcnn visualize_activation as restated: unit `dim` of forward_activations(layer), [n] f32 (render mode EncodingVis)."""
        coords7 = _f32(coords7)
        out = np.zeros(coords7.shape[0], np.float32)
        if not self.lib.orc_network_activation(self.h, coords7.shape[0], coords7.ctypes.data, int(layer), int(dim), out.ctypes.data):
            raise ValueError(f"no unit {dim} in layer {layer}")
        return out

    def update_density_grid(self, grid, update, edits=()):
        """One update_density_grid_nerf_operator call.  grid [5*128^3] f32 is updated in place; `update` (a
        nerfshop_amd._abi.GridUpdate) has rng / ema_step advanced.  Returns the new bitfield (also installed)."""
        assert grid.dtype == np.float32 and grid.flags.c_contiguous and grid.size == 5 * 128 ** 3
        bits = np.zeros(5 * 128 ** 3 // 8, np.uint8)
        arr = (C.c_void_p * max(len(edits), 1))(*[e.h for e in edits])
        self.lib.orc_update_density_grid(self.h, arr, len(edits), grid.ctypes.data, bits.ctypes.data, C.byref(update))
        self.set_bitfield(bits)
        return bits

    def density_on_grid(self, res3d, box_min, box_max, density_grid=None):
        res = (C.c_uint32 * 3)(*res3d)
        mn, mx = (C.c_float * 3)(*box_min), (C.c_float * 3)(*box_max)
        out = np.zeros(int(res3d[0]) * int(res3d[1]) * int(res3d[2]), np.float32)
        g = np.ascontiguousarray(density_grid, np.float32) if density_grid is not None else None
        self.lib.orc_density_on_grid(self.h, res, mn, mx, g.ctypes.data if g is not None else None, out.ctypes.data)
        return out

    def poisson_boundary(self, vertices, sh_width, hemisphere_width, jitter, is_inside):
        v = _f32(vertices).reshape(-1, 3)
        jt = _f32(jitter)
        n = v.shape[0]
        density, sh, coords = np.zeros(n, np.float32), np.zeros((n, 27), np.float32), np.zeros((n * sh_width * sh_width, 7), np.float32)
        self.lib.orc_poisson_boundary(self.h, v.ctypes.data, n, sh_width, hemisphere_width, jt.ctypes.data, 1 if is_inside else 0,
                                      density.ctypes.data, sh.ctypes.data, coords.ctypes.data)
        return density, sh, coords

    def project_selection_pixels(self, params, pixels_xy, threshold=0.1):
        px = np.ascontiguousarray(pixels_xy, np.int32).reshape(-1, 2)
        n = px.shape[0]
        pos, cells, found = np.zeros((n, 3), np.float32), np.zeros(n, np.uint32), np.zeros(n, np.uint8)
        self.lib.orc_project_selection_pixels(self.h, C.byref(params), px.ctypes.data, n, C.c_float(threshold), pos.ctypes.data,
                                              cells.ctypes.data, found.ctypes.data)
        return pos, cells, found

    def rgba_on_grid(self, res3d, box_min, box_max, ray_dir):
        res = (C.c_uint32 * 3)(*res3d)
        mn, mx, rd = (C.c_float * 3)(*box_min), (C.c_float * 3)(*box_max), (C.c_float * 3)(*ray_dir)
        out = np.zeros((int(res3d[0]) * int(res3d[1]) * int(res3d[2]), 4), np.float32)
        self.lib.orc_rgba_on_grid(self.h, res, mn, mx, rd, out.ctypes.data)
        return out

    def trace_samples(self, params, pixel_idx, max_samples):
        pixel_idx = np.ascontiguousarray(pixel_idx, np.uint32)
        n = pixel_idx.size
        t = np.zeros((n, max_samples), np.float32)
        dt = np.zeros((n, max_samples), np.float32)
        cnt = np.zeros(n, np.uint32)
        self.lib.orc_trace_samples(self.h, C.byref(params), n, pixel_idx.ctypes.data, max_samples, t.ctypes.data, dt.ctypes.data, cnt.ctypes.data)
        return t, dt, cnt

    def render(self, params, edits=(), fixed_S=0, n_threads=0):
        W, H = params.resolution[0], params.resolution[1]
        frame = np.zeros((H, W, 4), np.float32)
        depth = np.zeros((H, W), np.float32)
        steps = np.zeros((H, W), np.uint32)
        stats = OrcRenderStats()
        arr = (C.c_void_p * max(len(edits), 1))(*[e.h for e in edits])
        self.lib.orc_render(self.h, C.byref(params), arr, len(edits), frame.ctypes.data, depth.ctypes.data, steps.ctypes.data, C.byref(stats),
                            fixed_S, n_threads)
        return frame, depth, steps, stats

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_model_destroy(self.h)
            self.h = None


class Edit:
    def __init__(self, desc, tet_mesh_struct, keepalive=None):
        self.lib = load()
        self.keepalive = keepalive
        self.h = self.lib.orc_edit_create(C.byref(desc), C.byref(tet_mesh_struct))

    def map_rays(self, coords7):
        c = np.array(coords7, np.float32, copy=True)
        empty = np.zeros(c.shape[0], np.uint8)
        self.lib.orc_edit_map_rays(self.h, c.shape[0], c.ctypes.data, empty.ctypes.data)
        return c, empty

    def map_positions(self, pos):
        c = np.array(pos, np.float32, copy=True)
        empty = np.zeros(c.shape[0], np.uint8)
        self.lib.orc_edit_map_positions(self.h, c.shape[0], c.ctypes.data, c.shape[1], empty.ctypes.data)
        return c, empty

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_edit_destroy(self.h)
            self.h = None


class AffineEdit(Edit):
    """AffineDuplication restated; same map_rays / map_positions interface as Edit."""

    def __init__(self, desc, affine_op):
        self.lib = load()
        self.keepalive = affine_op
        self.h = self.lib.orc_edit_create_affine(C.byref(desc), C.byref(affine_op))


def accumulate(frame, accum, sample_count, color_space=0):
    """CudaRenderBuffer::accumulate: `accum` [H, W, 4] f32 is updated in place with frame number sample_count (0-based) of the view."""
    f = np.ascontiguousarray(frame, np.float32)
    assert accum.dtype == np.float32 and accum.flags.c_contiguous and accum.shape == f.shape
    load().orc_accumulate(f.shape[1], f.shape[0], f.ctypes.data, accum.ctypes.data, int(sample_count), int(color_space))
    return accum


def tet_lut_build(vertices, tets):
    lib = load()
    v = _f32(vertices)
    t = np.ascontiguousarray(tets, np.uint32)
    h = lib.orc_tet_lut_build(v.ctypes.data, v.shape[0], t.ctypes.data, t.shape[0])
    n_idx = lib.orc_tet_lut_n_idx(h)
    offsets = np.zeros(5 * 128 ** 3 + 1, np.uint32)
    idx = np.zeros(max(n_idx, 1), np.uint32)
    bitfield = np.zeros(5 * 128 ** 3 // 8, np.uint8)
    lib.orc_tet_lut_copy(h, offsets.ctypes.data, idx.ctypes.data, bitfield.ctypes.data)
    mx = lib.orc_tet_lut_max_per_cell(h)
    lib.orc_tet_lut_destroy(h)
    return offsets, idx[:n_idx], bitfield, mx


def mvc_compute(cage_v, cage_t, points):
    lib = load()
    cv, tr, pts = _f32(cage_v), np.ascontiguousarray(cage_t, np.uint32), _f32(points)
    w = np.zeros((pts.shape[0], cv.shape[0]), np.float32)
    labels = np.zeros(pts.shape[0], np.uint8)
    lib.orc_mvc_compute(cv.ctypes.data, cv.shape[0], tr.ctypes.data, tr.shape[0], pts.ctypes.data, pts.shape[0], w.ctypes.data, labels.ctypes.data)
    return w, labels


def mvc_apply(weights, cage_v):
    lib = load()
    w, cv = _f32(weights), _f32(cage_v)
    out = np.zeros((w.shape[0], 3), np.float32)
    lib.orc_mvc_apply(w.ctypes.data, cv.ctypes.data, cv.shape[0], w.shape[0], out.ctypes.data)
    return out


def poisson_interpolate(gamma, inside_density, outside_density, inside_shs, outside_shs):
    """GrowingSelection::interpolate_poisson_boundary restated (per-cage-vertex membrane terms -> per-tet-vertex)."""
    lib = load()
    g = _f32(gamma)
    i_d, o_d, i_s, o_s = _f32(inside_density), _f32(outside_density), _f32(inside_shs).reshape(-1, 27), _f32(outside_shs).reshape(-1, 27)
    n = g.shape[0]
    sh, od, rd = np.zeros((n, 27), np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    lib.orc_poisson_interpolate.restype = None
    lib.orc_poisson_interpolate(C.c_void_p(g.ctypes.data), C.c_uint32(n), C.c_uint32(g.shape[1]), C.c_void_p(i_d.ctypes.data), C.c_void_p(o_d.ctypes.data),
                                C.c_void_p(i_s.ctypes.data), C.c_void_p(o_s.ctypes.data), C.c_void_p(sh.ctypes.data), C.c_void_p(od.ctypes.data), C.c_void_p(rd.ctypes.data))
    return sh, od, rd


def local_rotations(vertices, original, tets):
    lib = load()
    v, o, t = _f32(vertices), _f32(original), np.ascontiguousarray(tets, np.uint32)
    out = np.zeros((t.shape[0], 9), np.float32)
    lib.orc_tet_local_rotations(v.ctypes.data, o.ctypes.data, t.ctypes.data, t.shape[0], out.ctypes.data)
    return out


class Pcg32:
    """tcnn::pcg32 restated (see nrs_oracle.cpp)."""

    def __init__(self, seed=1337):
        self.lib = load()
        self.state, self.inc = C.c_uint64(), C.c_uint64()
        self.lib.orc_pcg32_seed(seed, C.byref(self.state), C.byref(self.inc))

    def next_uint(self):
        return self.lib.orc_pcg32_next_uint(C.byref(self.state), self.inc)

    def next_float(self):
        return self.lib.orc_pcg32_next_float(C.byref(self.state), self.inc)

    def advance(self, delta=1 << 32):
        self.lib.orc_pcg32_advance(C.byref(self.state), self.inc, delta)


def density_grid_to_bitfield(grid):
    lib = load()
    g = _f32(grid)
    out = np.zeros(5 * 128 ** 3 // 8, np.uint8)
    lib.orc_density_grid_to_bitfield(g.ctypes.data, out.ctypes.data)
    return out
