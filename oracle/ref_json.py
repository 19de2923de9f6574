"""ctypes wrapper of oracle/_ref/libref_json.so: the REFERENCE's own edit serialisation (Testbed::save_edits and the to_json / from_json it calls) compiled as
host code -- see oracle/ref_json.cpp for what is compiled from the reference and what is a stand-in.

TEST INFRASTRUCTURE ONLY.  The library exists only where /root/reference is mounted (this container, never the GPU box); it writes the fixtures
tests/golden/ref_edits_*.json.gz (tests/golden/make_ref_edits_golden.py) and, live, the files of tests/test_ref_pin.py::test_edits_reader_*_live.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libref_json.so")
_lib = None


class RefJsonOp(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("n_cage_vertices", C.c_uint32), ("n_cage_indices", C.c_uint32),
        ("cage_vertices", C.c_void_p), ("cage_original_vertices", C.c_void_p), ("cage_indices", C.c_void_p),
        ("cage_inside_density", C.c_void_p), ("cage_outside_density", C.c_void_p), ("cage_inside_shs", C.c_void_p), ("cage_outside_shs", C.c_void_p),
        ("n_vertices", C.c_uint32), ("n_tets", C.c_uint32), ("n_surface_indices", C.c_uint32),
        ("vertices", C.c_void_p), ("original_vertices", C.c_void_p), ("tets", C.c_void_p), ("surface_indices", C.c_void_p),
        ("mvc_coordinates", C.c_void_p), ("gamma_coordinates", C.c_void_p),
        ("selection_center", C.c_float * 3), ("selection_scale", C.c_float * 3), ("selection_rot", C.c_float * 9),
        ("selection_min", C.c_float * 3), ("selection_max", C.c_float * 3),
        ("translation", C.c_float * 3), ("scale", C.c_float * 3), ("rotation", C.c_float * 9),
        ("hide_original", C.c_int32), ("correct_dir", C.c_int32),
    ]


def available():
    return os.path.exists(LIB_PATH)


def load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB_PATH)
        _lib.refjson_last_error.restype = C.c_char_p
        _lib.refjson_save_edits.restype = C.c_int
        _lib.refjson_save_edits.argtypes = [C.c_char_p, C.POINTER(RefJsonOp), C.c_uint32]
        _lib.refjson_reload_edits.restype = C.c_int
        _lib.refjson_reload_edits.argtypes = [C.c_char_p, C.c_char_p]
        _lib.refjson_save_snapshot.restype = C.c_int
        _lib.refjson_save_snapshot.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.c_float]
    return _lib


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, np.float32)


def _u32(a):
    return None if a is None else np.ascontiguousarray(a, np.uint32)


def cage_op(cage_vertices, cage_original_vertices, cage_triangles, vertices=None, original_vertices=None, tets=None, mvc=None, gamma=None, surface_indices=None,
            inside_density=None, outside_density=None, inside_shs=None, outside_shs=None):
    """A cage_deformation operator for save_edits: the proxy cage and (optionally) the interpolation mesh.  Returns (RefJsonOp, keep-alive list)."""
    op = RefJsonOp()
    keep = []

    def ptr(a):
        if a is None:
            return None
        keep.append(a)
        return a.ctypes.data

    cv, cov, ct = _f32(cage_vertices), _f32(cage_original_vertices), _u32(np.asarray(cage_triangles).reshape(-1))
    op.kind = 0
    op.n_cage_vertices, op.n_cage_indices = cv.shape[0], ct.size
    op.cage_vertices, op.cage_original_vertices, op.cage_indices = ptr(cv), ptr(cov), ptr(ct)
    op.cage_inside_density, op.cage_outside_density = ptr(_f32(inside_density)), ptr(_f32(outside_density))
    op.cage_inside_shs, op.cage_outside_shs = ptr(_f32(inside_shs)), ptr(_f32(outside_shs))
    if vertices is not None:
        v, ov, t = _f32(vertices), _f32(original_vertices), _u32(tets)
        op.n_vertices, op.n_tets = v.shape[0], t.shape[0]
        op.vertices, op.original_vertices, op.tets = ptr(v), ptr(ov), ptr(t)
        si = _u32(None if surface_indices is None else np.asarray(surface_indices).reshape(-1))
        op.n_surface_indices = 0 if si is None else si.size
        op.surface_indices = ptr(si)
        op.mvc_coordinates, op.gamma_coordinates = ptr(_f32(mvc)), ptr(_f32(gamma))
    return op, keep


def affine_op(a, selection_min=(0, 0, 0), selection_max=(0, 0, 0)):
    """An affine_duplication operator from an nrs_affine_duplication (nerfshop_amd._abi.AffineDuplicationOp)."""
    op = RefJsonOp()
    op.kind = 1
    for name in ("selection_center", "selection_scale", "selection_rot", "translation", "scale", "rotation"):
        setattr(op, name, getattr(a, name))
    op.selection_min = (C.c_float * 3)(*selection_min)
    op.selection_max = (C.c_float * 3)(*selection_max)
    op.hide_original, op.correct_dir = int(a.hide_original), int(a.correct_dir)
    return op, []


def save_edits(path, ops):
    """Testbed::save_edits (src/testbed.cu:3190-3204) over `ops` = [(RefJsonOp, keep-alive), ...]."""
    lib = load()
    arr = (RefJsonOp * len(ops))(*[o for o, _ in ops])
    if lib.refjson_save_edits(str(path).encode(), arr, len(ops)) != 0:
        raise RuntimeError("refjson_save_edits: " + lib.refjson_last_error().decode())


def reload_edits(path, out_path):
    """The reference's readers (Testbed::load_edits' dispatch + from_json of Cage / TetMesh / AffineBoundingBox) on `path`; what they loaded is written back by
    the reference's writers to `out_path`.  Returns the operator count; raises if the reference's readers reject the file."""
    lib = load()
    n = lib.refjson_reload_edits(str(path).encode(), str(out_path).encode())
    if n < 0:
        raise RuntimeError("refjson_reload_edits: " + lib.refjson_last_error().decode())
    return n


def save_snapshot(path, config_path, params_u16, density_grid, aabb_scale, log2_hashmap_size=0, training_step=35000, loss=0.001):
    """Testbed::save_snapshot (src/testbed.cu:3090-3113) with `config_path` -- one of the reference's configs/nerf/*.json, parsed where it lies -- as the network
    config.  The trainer's three keys (params_binary / params_type / n_params) and the binary conversion of the density grid are tiny-cuda-nn's, restated in
    oracle/ref_json.cpp; every other byte of the file comes out of reference code."""
    lib = load()
    p = np.ascontiguousarray(params_u16, np.uint16)
    g = np.ascontiguousarray(density_grid, np.float32)
    if lib.refjson_save_snapshot(str(path).encode(), str(config_path).encode(), int(log2_hashmap_size), p.ctypes.data, p.size, g.ctypes.data, g.size, int(aabb_scale), int(training_step), float(loss)) != 0:
        raise RuntimeError("refjson_save_snapshot: " + lib.refjson_last_error().decode())
