"""On-disk formats either side of the render path, over the host-only entry points of libnrs:

  load_snapshot / save_snapshot   Testbed::load_snapshot / save_snapshot / export_snapshot   src/testbed.cu:3054-3190
  load_edits / save_edits         Testbed::load_edits / save_edits                           src/testbed.cu:3190-3236

The readers are the product (C++, nrs_formats.cpp).  The writers here are harness code: they emit the reference's
schemas with Python's msgpack / json / zlib so that tests and the synthetic scenes can produce the files the reference
itself would write (there is no real snapshot in the build container and no network to fetch one).
"""
import ctypes as C
import json
import zlib

import numpy as np

from . import _abi
from ._abi import ModelDesc, TetMesh, check


class Snapshot:
    """desc, aabb_scale, params (uint16 fp16 bits), density_grid (float32 [5*128^3]), camera (12 floats or None)"""


def load_snapshot(path):
    lib = _abi.load()
    h = C.c_void_p()
    check(lib.nrs_snapshot_open(str(path).encode(), C.byref(h)))
    try:
        s = Snapshot()
        s.desc = ModelDesc()
        scale = C.c_uint32()
        check(lib.nrs_snapshot_model_desc(h, C.byref(s.desc), C.byref(scale)))
        s.aabb_scale = scale.value
        n = C.c_size_t()
        p = lib.nrs_snapshot_params_fp16(h, C.byref(n))
        s.params = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint16)), (n.value,)).copy()
        p = lib.nrs_snapshot_density_grid(h, C.byref(n))
        s.density_grid = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), (n.value,)).copy()
        cam = np.zeros(12, np.float32)
        s.camera = cam if lib.nrs_snapshot_camera(h, cam.ctypes.data) == _abi.NRS_OK else None
        return s
    finally:
        lib.nrs_snapshot_close(h)


def network_config(desc, explicit_per_level_scale=False):
    """configs/nerf/base.json's network part for `desc` (per_level_scale is derived on load unless stored)."""
    enc = {"otype": "HashGrid", "n_levels": int(desc.n_levels), "n_features_per_level": int(desc.n_features_per_level),
           "log2_hashmap_size": int(desc.log2_hashmap_size), "base_resolution": int(desc.base_resolution)}
    if explicit_per_level_scale:
        enc["per_level_scale"] = float(desc.per_level_scale)
    cfg = {
        "loss": {"otype": "Huber"},
        "encoding": enc,
        "network": {"otype": "FullyFusedMLP" if desc.density_hidden_layers else "CutlassMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": int(desc.n_neurons),
                    "n_hidden_layers": int(desc.density_hidden_layers)},
    }
    if desc.sh_degree:  # (configs/nerf/base_nodir.json has neither block: NerfNetworkNoDir, testbed.cu:2314)
        cfg["dir_encoding"] = {"otype": "Composite", "nested": [{"n_dims_to_encode": 3, "otype": "SphericalHarmonics", "degree": int(desc.sh_degree)},
                                                                 {"otype": "Identity", "n_bins": 4, "degree": 4}]}
        cfg["rgb_network"] = {"otype": "FullyFusedMLP" if desc.rgb_hidden_layers else "CutlassMLP", "activation": "ReLU", "output_activation": "None",
                              "n_neurons": int(desc.n_neurons), "n_hidden_layers": int(desc.rgb_hidden_layers)}
    return cfg


def save_snapshot(path, desc, aabb_scale, params_u16, density_grid, camera=None, exported=None, training_step=35000):
    """Write the reference's snapshot schema.  exported=False: Testbed::save_snapshot (float grid, aabb_scale inside the
    dataset object); exported=True: Testbed::export_snapshot (fp16 grid of the used cascades, snapshot.nerf.aabb_scale);
    default: exported iff the path ends in .ingp.  `.ingp` files are zlib-compressed (zstr)."""
    import msgpack
    path = str(path)
    if exported is None:
        exported = path.lower().endswith(".ingp")
    params_u16 = np.ascontiguousarray(params_u16, np.uint16)
    grid = np.ascontiguousarray(density_grid, np.float32)
    cfg = network_config(desc)
    snap = {"density_grid_size": 128, "training_step": int(training_step), "loss": 0.001,
            "n_params": int(params_u16.size), "params_type": "__half", "params_binary": params_u16.tobytes()}
    rgb = {"rays_per_batch": 4096, "measured_batch_size": 262144, "measured_batch_size_before_compaction": 1048576}
    if exported:
        max_cascade = int(np.log2(aabb_scale))
        snap["version"] = 1
        snap["density_grid_binary"] = grid[: (max_cascade + 1) * 128 ** 3].astype(np.float16).tobytes()
        snap["nerf"] = {"aabb_scale": int(aabb_scale), "rgb": rgb}
        half = 0.5 * min(16, aabb_scale)
        snap["aabb"] = {"min": [0.5 - half] * 3, "max": [0.5 + half] * 3}
    else:
        snap["density_grid_binary"] = grid.tobytes()
        snap["nerf"] = {"rgb": rgb, "dataset": {"aabb_scale": int(aabb_scale), "scale": 0.33, "offset": [0.5, 0.5, 0.5], "n_images": 0}}
    if camera is not None:
        cam = np.asarray(camera, np.float32).reshape(4, 3).T  # column-major 3x4 -> rows (Eigen's to_json, json_binding.h:30-43)
        snap["camera"] = {"matrix": [[float(v) for v in row] for row in cam], "fov_axis": 1, "zoom": 1.0, "scale": 1.0}
    cfg["snapshot"] = snap
    blob = msgpack.packb(cfg, use_bin_type=True)
    if path.lower().endswith(".ingp"):
        blob = zlib.compress(blob, 6)
    with open(path, "wb") as f:
        f.write(blob)


def _vec3_rows(a):
    return [[float(x) for x in row] for row in np.asarray(a, np.float32).reshape(-1, 3)]


def _bbox(v):
    return {"min": [float(x) for x in v.min(0)], "max": [float(x) for x in v.max(0)]}


def save_edits(path, cage_edits):
    """Testbed::save_edits for synth.CageEdit objects: the fields the reference's JSON constructor needs to rebuild the
    operator (proxy_cage, interpolation_mesh; growing_selection.cu:96-115); selection bookkeeping is written empty."""
    ops = []
    for e in cage_edits:
        if isinstance(e, _abi.AffineDuplicationOp):   # AffineDuplication::to_json, affine_duplication.cu:356-369
            def m3(a):   # column-major 9 -> rows (Eigen to_json)
                return [[float(a[3 * c + r]) for c in range(3)] for r in range(3)]
            c, sc = [float(v) for v in e.selection_center], [float(v) for v in e.selection_scale]
            box = {"min": c, "max": c, "rot_matrix": m3(e.selection_rot), "u": [1.0, 0.0, 0.0], "v": [0.0, 1.0, 0.0], "w": [0.0, 0.0, 1.0],
                   "center": c, "scale": sc}   # min / max / u / v / w are rebuilt on load (warp_box), written as placeholders
            ops.append({"type": "affine_duplication", "selection_box": box, "translation": [float(v) for v in e.translation],
                        "scale": [float(v) for v in e.scale], "rotation_matrix": m3(e.rotation), "hide_original": bool(e.hide_original),
                        "correct_dir": bool(e.correct_dir)})
            continue
        V = e.vertices.shape[0]
        cage = {"vertices": _vec3_rows(e.cage_deformed), "indices": [int(i) for i in e.cage_triangles.reshape(-1)], "normals": [],
                "initial_normals": [], "labels": [0] * e.cage_vertices.shape[0], "original_vertices": _vec3_rows(e.cage_vertices), "colors": [],
                "outside_colors": [], "initial_colors": [], "new_shs": [], "initial_shs": [], "inside_shs": [], "outside_shs": [],
                "inside_density": [], "outside_density": []}
        mesh = {"bbox": _bbox(e.vertices), "original_bbox": _bbox(e.original_vertices), "warped_bbox": _bbox(e.vertices),
                "original_warped_bbox": _bbox(e.original_vertices), "vertices": _vec3_rows(e.vertices), "indices": [],
                "original_vertices": _vec3_rows(e.original_vertices), "mvc_coordinates": [[float(w) for w in row] for row in e.mvc_weights],
                "gamma_coordinates": [], "tets": [int(i) for i in e.tets.reshape(-1)], "labels": [0] * V, "colors": [], "all_indices": []}
        ops.append({"type": "cage_deformation", "projected_pixels": [], "projected_labels": [], "projected_cell_idx": [], "selection_points": [],
                    "selection_labels": [], "selection_cell_idx": [], "m_selection_grid_bitfield": [], "growing_level": 0, "region_growing": {},
                    "selection_mesh": {"vertices": [], "indices": [], "normals": []}, "proxy_cage": cage, "interpolation_mesh": mesh})
    with open(str(path), "w") as f:
        json.dump({"edit_operators": ops}, f)
        f.write("\n")


class LoadedCage:
    """One cage_deformation operator of an edits file: arrays + the nrs_tet_mesh for device authoring."""

    def tet_mesh_struct(self, device_authoring=True):
        m = TetMesh()
        m.n_vertices, m.n_tets = self.vertices.shape[0], self.tets.shape[0]
        m.h_vertices = self.vertices.ctypes.data
        m.h_original_vertices = self.original_vertices.ctypes.data
        m.h_tets = self.tets.ctypes.data
        m.residual_amplitude = 1.0
        m.correct_direction = 1
        return m


def load_edits(path):
    lib = _abi.load()
    h = C.c_void_p()
    check(lib.nrs_edits_open(str(path).encode(), C.byref(h)))
    out = []
    try:
        for i in range(lib.nrs_edits_count(h)):
            kind = lib.nrs_edits_type(h, i).decode()
            if kind == "affine_duplication":
                op = _abi.AffineDuplicationOp()
                check(lib.nrs_edits_affine(h, i, C.byref(op)))
                out.append(op)
                continue
            if kind != "cage_deformation":
                out.append(kind)
                continue
            mesh = TetMesh()
            mvc, cv, cov, ct = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
            ncv, nct = C.c_uint32(), C.c_uint32()
            check(lib.nrs_edits_cage(h, i, C.byref(mesh), C.byref(mvc), C.byref(cv), C.byref(cov), C.byref(ct), C.byref(ncv), C.byref(nct)))

            def f32(p, n):
                return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), (n,)).copy()

            def u32(p, n):
                return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), (n,)).copy()

            c = LoadedCage()
            V, T = mesh.n_vertices, mesh.n_tets
            c.vertices = f32(mesh.h_vertices, 3 * V).reshape(V, 3) if V else np.zeros((0, 3), np.float32)
            c.original_vertices = f32(mesh.h_original_vertices, 3 * V).reshape(V, 3) if V else np.zeros((0, 3), np.float32)
            c.tets = u32(mesh.h_tets, 4 * T).reshape(T, 4) if T else np.zeros((0, 4), np.uint32)
            c.mvc_weights = f32(mvc.value, V * ncv.value).reshape(V, ncv.value) if (mvc.value and V) else None
            c.cage_deformed = f32(cv.value, 3 * ncv.value).reshape(-1, 3)
            c.cage_vertices = f32(cov.value, 3 * ncv.value).reshape(-1, 3)
            c.cage_triangles = u32(ct.value, 3 * nct.value).reshape(-1, 3)
            c.local_rotations = True  # correct_direction
            c.copy = False
            out.append(c)
        return out
    finally:
        lib.nrs_edits_close(h)
