"""ctypes mirror of include/nrs.h (the C-ABI of the render path) and the loader of libnrs.so.

The library is the product: hand-written HIP kernels for gfx950 + the C++ host code around them.  There is
no CPU fallback anywhere in this package -- if the shared library is missing, `load()` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NRS_LIB_PATH") or os.path.join(_HERE, "csrc", "libnrs.so")  # NRS_LIB_PATH: A/B builds while profiling

NRS_OK = 0
GRID_SIZE = 128
GRID_CASCADES = 5
GRID_VOLUME = GRID_SIZE ** 3
BITFIELD_BYTES = GRID_VOLUME * GRID_CASCADES // 8
N_LUT_CELLS = GRID_VOLUME * GRID_CASCADES

ACT_NONE, ACT_RELU, ACT_LOGISTIC, ACT_EXPONENTIAL = 0, 1, 2, 3
RENDER_AO, RENDER_SHADE, RENDER_NORMALS, RENDER_POSITIONS, RENDER_DEPTH, RENDER_DISTANCE, RENDER_STEPSIZE, RENDER_DISTORTION, RENDER_COST, RENDER_SLICE = range(10)
RENDER_ENCODING_VIS = 11
LAYOUT_PLANES, LAYOUT_INTERLEAVED = 0, 1


class ModelDesc(C.Structure):
    _fields_ = [
        ("n_levels", C.c_uint32),
        ("n_features_per_level", C.c_uint32),
        ("log2_hashmap_size", C.c_uint32),
        ("base_resolution", C.c_uint32),
        ("per_level_scale", C.c_float),
        ("n_neurons", C.c_uint32),
        ("density_hidden_layers", C.c_uint32),
        ("density_output_dims", C.c_uint32),
        ("rgb_hidden_layers", C.c_uint32),
        ("sh_degree", C.c_uint32),
        ("rgb_activation", C.c_uint32),
        ("density_activation", C.c_uint32),
        ("aabb_min", C.c_float * 3),
        ("aabb_max", C.c_float * 3),
    ]


class TetMesh(C.Structure):
    _fields_ = [
        ("n_vertices", C.c_uint32),
        ("n_tets", C.c_uint32),
        ("h_vertices", C.c_void_p),
        ("h_original_vertices", C.c_void_p),
        ("h_tets", C.c_void_p),
        ("h_lut_offsets", C.c_void_p),
        ("h_lut_idx", C.c_void_p),
        ("h_original_bitfield", C.c_void_p),
        ("h_local_rotations", C.c_void_p),
        ("copy", C.c_uint32),
        ("apply_poisson", C.c_uint32),
        ("residual_amplitude", C.c_float),
        ("h_boundary_shs", C.c_void_p),
        ("h_boundary_outside_density", C.c_void_p),
        ("h_boundary_residual_density", C.c_void_p),
        ("correct_direction", C.c_uint32),
    ]


class AffineDuplicationOp(C.Structure):
    _fields_ = [
        ("selection_center", C.c_float * 3),
        ("selection_scale", C.c_float * 3),
        ("selection_rot", C.c_float * 9),
        ("translation", C.c_float * 3),
        ("scale", C.c_float * 3),
        ("rotation", C.c_float * 9),
        ("hide_original", C.c_uint32),
        ("correct_dir", C.c_uint32),
    ]


class RenderParams(C.Structure):
    """nrs_render_params; struct_size is filled in on construction (NRS_RENDER_PARAMS_INIT)."""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("resolution", C.c_int32 * 2),
        ("focal_length", C.c_float * 2),
        ("camera_matrix0", C.c_float * 12),
        ("camera_matrix1", C.c_float * 12),
        ("rolling_shutter", C.c_float * 4),
        ("screen_center", C.c_float * 2),
        ("render_aabb_min", C.c_float * 3),
        ("render_aabb_max", C.c_float * 3),
        ("spp_index", C.c_uint32),
        ("snap_to_pixel_centers", C.c_uint32),
        ("min_transmittance", C.c_float),
        ("cone_angle_constant", C.c_float),
        ("render_mode", C.c_uint32),
        ("linear_colors", C.c_uint32),
        ("apply_operators", C.c_uint32),
        ("poisson_target", C.c_uint32),
        ("min_mip", C.c_uint32),
        ("max_march_steps", C.c_uint32),
        ("tile_size", C.c_uint32),
        ("tile_first", C.c_uint32),
        ("tile_stride", C.c_uint32),
        ("dof", C.c_float),
        ("slice_plane_z", C.c_float),
        ("depth_scale", C.c_float),
        ("show_accel", C.c_uint32),
        ("distortion_mode", C.c_uint32),
        ("distortion_params", C.c_float * 7),
        ("d_distortion_map", C.c_void_p),
        ("distortion_resolution", C.c_int32 * 2),
        ("envmap_resolution", C.c_int32 * 2),
        ("d_envmap", C.c_void_p),
        ("glow_mode", C.c_uint32),
        ("glow_y_cutoff", C.c_float),
        ("visualized_layer", C.c_uint32),
        ("visualized_dimension", C.c_uint32),
    ]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(RenderParams)


class RenderStats(C.Structure):
    _fields_ = [("n_samples", C.c_uint64), ("n_rays_alive", C.c_uint32), ("n_rays_hit", C.c_uint32)]


class GridUpdate(C.Structure):
    _fields_ = [
        ("n_uniform_samples", C.c_uint32),
        ("n_nonuniform_samples", C.c_uint32),
        ("reset_grid", C.c_uint32),
        ("max_cascade", C.c_uint32),
        ("decay", C.c_float),
        ("ema_step", C.c_uint32),
        ("rng_state", C.c_uint64),
        ("rng_inc", C.c_uint64),
    ]


# every symbol include/nrs.h declares; tests check the library exports exactly these
EXPORTS = [
    "nrs_last_error", "nrs_abi_version", "nrs_edit_poisson_interpolate", "nrs_edit_download_poisson", "nrs_comm_unique_id", "nrs_comm_create", "nrs_comm_info", "nrs_comm_destroy", "nrs_gather_tiles", "nrs_comm_probe_self_p2p",
    "nrs_ctx_create", "nrs_ctx_destroy", "nrs_ctx_device_info", "nrs_ctx_set_lane_teams", "nrs_ctx_set_ray_handover", "nrs_ctx_ray_handovers",
    "nrs_model_create", "nrs_model_destroy", "nrs_model_n_params", "nrs_model_level_table",
    "nrs_model_set_params", "nrs_model_set_params_device", "nrs_model_set_numerics", "nrs_model_set_cell_cache", "nrs_model_cell_cache_bytes", "nrs_model_set_sparse_cell_cache", "nrs_model_sparse_cell_cache_bytes", "nrs_model_set_density_bitfield", "nrs_model_set_density_grid",
    "nrs_model_get_density_bitfield", "nrs_model_get_march_accelerator", "nrs_model_get_density_grid", "nrs_model_update_density_grid", "nrs_rng_seed",
    "nrs_network_inference", "nrs_network_density", "nrs_hashgrid_encode", "nrs_density_on_grid", "nrs_rgba_on_grid", "nrs_network_input_gradient", "nrs_network_visualize_activation",
    "nrs_poisson_boundary", "nrs_poisson_sample_coords", "nrs_project_selection_pixels", "nrs_upper_cell_idx", "nrs_selection_cells",
    "nrs_edit_create", "nrs_edit_create_affine", "nrs_edit_destroy", "nrs_edit_map_rays", "nrs_edit_map_positions",
    "nrs_edit_set_mvc", "nrs_edit_update_cage", "nrs_edit_update_vertices", "nrs_edit_lut_size", "nrs_edit_download",
    "nrs_render_nerf", "nrs_render_owned_tiles", "nrs_render_tile_pitch", "nrs_detile", "nrs_trace_samples", "nrs_accumulate",
    "nrs_snapshot_open", "nrs_snapshot_close", "nrs_snapshot_model_desc", "nrs_snapshot_params_fp16", "nrs_snapshot_density_grid",
    "nrs_snapshot_camera", "nrs_edits_open", "nrs_edits_close", "nrs_edits_count", "nrs_edits_type", "nrs_edits_cage", "nrs_edits_affine",
    "nrs_tet_lut_build", "nrs_tet_lut_n_idx", "nrs_tet_lut_max_per_cell", "nrs_tet_lut_offsets",
    "nrs_tet_lut_idx", "nrs_tet_lut_bitfield", "nrs_tet_lut_destroy",
    "nrs_mvc_compute", "nrs_mvc_apply", "nrs_tet_local_rotations",
]

_lib = None


class NrsError(RuntimeError):
    pass


def load():
    """Load libnrs.so (built in-tree by __graft_entry__.build()).  Fails loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NrsError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    # One HIP runtime per process: PyTorch bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  Importing torch
    # first makes the dynamic linker bind libnrs.so's NEEDED libamdhip64.so.7 to that already-loaded copy, so torch
    # tensors and our kernels share a device context (two runtimes in one process cannot both open the GPU).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    P, U32, I = C.c_void_p, C.c_uint32, C.c_int
    lib.nrs_last_error.restype = C.c_char_p
    lib.nrs_abi_version.restype = I
    lib.nrs_ctx_create.argtypes = [I, C.POINTER(P)]
    lib.nrs_ctx_destroy.argtypes = [P]
    lib.nrs_ctx_destroy.restype = None
    lib.nrs_ctx_device_info.argtypes = [P, C.c_char_p, C.c_size_t, C.POINTER(I), C.POINTER(C.c_size_t)]
    lib.nrs_model_create.argtypes = [P, C.POINTER(ModelDesc), C.POINTER(P)]
    lib.nrs_ctx_set_lane_teams.argtypes = [P, C.c_int]
    lib.nrs_ctx_set_ray_handover.argtypes = [P, C.c_int]
    lib.nrs_ctx_ray_handovers.argtypes = [P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.nrs_model_destroy.argtypes = [P]
    lib.nrs_model_destroy.restype = None
    lib.nrs_model_n_params.argtypes = [C.POINTER(ModelDesc)]
    lib.nrs_model_n_params.restype = C.c_size_t
    lib.nrs_model_level_table.argtypes = [C.POINTER(ModelDesc), P, P, P, P, P]
    lib.nrs_model_set_params.argtypes = [P, P, C.c_size_t]
    lib.nrs_project_selection_pixels.argtypes = [P, P, C.POINTER(RenderParams), P, C.c_uint32, C.c_float, P, P, P]
    lib.nrs_poisson_boundary.argtypes = [P, P, C.c_uint32, C.c_uint32, C.c_uint32, P, C.c_int, P, P]
    lib.nrs_poisson_sample_coords.argtypes = [P, C.c_uint32, C.c_uint32, C.c_uint32, P, P, P, P]
    lib.nrs_poisson_sample_coords.restype = None
    lib.nrs_upper_cell_idx.argtypes = [C.c_uint32, C.c_uint32]
    lib.nrs_upper_cell_idx.restype = C.c_uint32
    lib.nrs_selection_cells.argtypes = [P, P, P, C.c_uint32, C.c_int, P, P, P, P]
    lib.nrs_model_set_cell_cache.argtypes = [P, C.c_size_t]
    lib.nrs_model_cell_cache_bytes.argtypes = [P, P]
    lib.nrs_model_cell_cache_bytes.restype = C.c_size_t
    lib.nrs_model_set_numerics.argtypes = [P, C.c_uint32, C.c_uint32]
    lib.nrs_model_set_params_device.argtypes = [P, P, C.c_size_t, P]
    lib.nrs_edit_poisson_interpolate.argtypes = [P, P, P, C.c_uint32, P, P, P, P, C.c_float]
    lib.nrs_edit_download_poisson.argtypes = [P, P, P, P]
    lib.nrs_comm_unique_id.argtypes = [P]
    lib.nrs_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, P, P]
    lib.nrs_comm_info.argtypes = [P, P, P, P, P, C.c_size_t]
    lib.nrs_comm_destroy.argtypes = [P]
    lib.nrs_comm_destroy.restype = None
    lib.nrs_gather_tiles.argtypes = [P, P, C.c_int, P, C.c_uint32, P, P, P, P, P]
    lib.nrs_comm_probe_self_p2p.argtypes = [P, P, P, C.c_size_t, C.c_int, P]
    lib.nrs_model_set_sparse_cell_cache.argtypes = [P, P, C.c_size_t]
    lib.nrs_model_sparse_cell_cache_bytes.argtypes = [P, P, P]
    lib.nrs_model_sparse_cell_cache_bytes.restype = C.c_size_t
    lib.nrs_model_set_density_bitfield.argtypes = [P, P, C.c_size_t]
    lib.nrs_model_set_density_grid.argtypes = [P, P, C.c_size_t]
    lib.nrs_model_get_density_bitfield.argtypes = [P, P, C.c_size_t]
    lib.nrs_model_get_density_grid.argtypes = [P, P, C.c_size_t]
    lib.nrs_model_get_march_accelerator.argtypes = [P, I, P, P]
    lib.nrs_model_update_density_grid.argtypes = [P, C.POINTER(P), I, C.POINTER(GridUpdate), P]
    lib.nrs_rng_seed.argtypes = [C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.nrs_rng_seed.restype = None
    lib.nrs_network_inference.argtypes = [P, P, U32, P, P, U32, I]
    lib.nrs_network_density.argtypes = [P, P, U32, P, U32, P, U32, I]
    lib.nrs_hashgrid_encode.argtypes = [P, P, U32, P, U32, P]
    lib.nrs_network_input_gradient.argtypes = [P, P, U32, P, U32, P]
    lib.nrs_network_visualize_activation.argtypes = [P, P, U32, U32, U32, P, P]
    lib.nrs_density_on_grid.argtypes = [P, P, C.POINTER(U32 * 3), C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3), I, P]
    lib.nrs_rgba_on_grid.argtypes = [P, P, C.POINTER(U32 * 3), C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3), P]
    lib.nrs_edit_create.argtypes = [P, C.POINTER(ModelDesc), C.POINTER(TetMesh), C.POINTER(P)]
    lib.nrs_edit_create_affine.argtypes = [P, C.POINTER(ModelDesc), C.POINTER(AffineDuplicationOp), C.POINTER(P)]
    lib.nrs_edits_affine.argtypes = [P, U32, C.POINTER(AffineDuplicationOp)]
    lib.nrs_edit_destroy.argtypes = [P]
    lib.nrs_edit_destroy.restype = None
    lib.nrs_edit_map_rays.argtypes = [P, P, U32, P, P]
    lib.nrs_edit_map_positions.argtypes = [P, P, U32, P, U32, P]
    lib.nrs_edit_set_mvc.argtypes = [P, P, U32]
    lib.nrs_edit_update_cage.argtypes = [P, P, P, U32]
    lib.nrs_edit_update_vertices.argtypes = [P, P, P, U32]
    lib.nrs_edit_lut_size.argtypes = [P, C.POINTER(U32), C.POINTER(U32)]
    lib.nrs_edit_download.argtypes = [P, P, P, P, P, P, P]
    lib.nrs_render_nerf.argtypes = [P, C.POINTER(RenderParams), C.POINTER(P), I, P, P, P, P, C.POINTER(RenderStats)]
    lib.nrs_render_owned_tiles.argtypes = [C.POINTER(RenderParams)]
    lib.nrs_render_owned_tiles.restype = U32
    lib.nrs_render_tile_pitch.argtypes = [C.POINTER(RenderParams)]
    lib.nrs_render_tile_pitch.restype = U32
    lib.nrs_detile.argtypes = [P, P, C.POINTER(RenderParams), U32, U32, P, U32, C.c_size_t, P]
    lib.nrs_trace_samples.argtypes = [P, C.POINTER(RenderParams), P, U32, P, U32, P, P, P]
    lib.nrs_accumulate.argtypes = [P, P, U32, U32, P, P, U32, U32]
    lib.nrs_snapshot_open.argtypes = [C.c_char_p, C.POINTER(P)]
    lib.nrs_snapshot_close.argtypes = [P]
    lib.nrs_snapshot_close.restype = None
    lib.nrs_snapshot_model_desc.argtypes = [P, C.POINTER(ModelDesc), C.POINTER(U32)]
    lib.nrs_snapshot_params_fp16.argtypes = [P, C.POINTER(C.c_size_t)]
    lib.nrs_snapshot_params_fp16.restype = P
    lib.nrs_snapshot_density_grid.argtypes = [P, C.POINTER(C.c_size_t)]
    lib.nrs_snapshot_density_grid.restype = P
    lib.nrs_snapshot_camera.argtypes = [P, P]
    lib.nrs_edits_open.argtypes = [C.c_char_p, C.POINTER(P)]
    lib.nrs_edits_close.argtypes = [P]
    lib.nrs_edits_close.restype = None
    lib.nrs_edits_count.argtypes = [P]
    lib.nrs_edits_count.restype = U32
    lib.nrs_edits_type.argtypes = [P, U32]
    lib.nrs_edits_type.restype = C.c_char_p
    lib.nrs_edits_cage.argtypes = [P, U32, C.POINTER(TetMesh), C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(U32), C.POINTER(U32)]
    lib.nrs_tet_lut_build.argtypes = [P, U32, P, U32, I, C.POINTER(P)]
    for name in ("nrs_tet_lut_n_idx", "nrs_tet_lut_max_per_cell"):
        getattr(lib, name).argtypes = [P]
        getattr(lib, name).restype = U32
    for name in ("nrs_tet_lut_offsets", "nrs_tet_lut_idx", "nrs_tet_lut_bitfield"):
        getattr(lib, name).argtypes = [P]
        getattr(lib, name).restype = P
    lib.nrs_tet_lut_destroy.argtypes = [P]
    lib.nrs_tet_lut_destroy.restype = None
    lib.nrs_mvc_compute.argtypes = [P, U32, P, U32, P, U32, P, P]
    lib.nrs_mvc_apply.argtypes = [P, P, U32, U32, P]
    lib.nrs_tet_local_rotations.argtypes = [P, P, P, U32, P]
    _lib = lib
    return lib


def check(status):
    if status != NRS_OK:
        raise NrsError(f"nrs error {status}: {load().nrs_last_error().decode()}")
