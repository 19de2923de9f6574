/*
 * nrs.h -- C-ABI of the MI355X-native NeRFshop render path ("nrs" = NeRFshop Render path, Standalone).
 *
 * This is the drop-in boundary for ONE hot path of graphdeco-inria/nerfshop: the volumetric render path
 * (occupancy-grid marching -> cage/tet warp of samples -> hash-grid encoding -> fused MLPs -> compositing).
 * The reference has no FFI for this path; it sits behind C++ members called from one host thread on one
 * stream.  Every entry point below names the reference interface it replaces (paths relative to the
 * reference checkout):
 *
 *   nrs_render_nerf          <- Testbed::render_nerf                   src/testbed_nerf.cu:3066 (decl testbed.h:305)
 *                               = NerfTracer::init_rays_from_camera    src/testbed_nerf.cu:2683
 *                               + NerfTracer::trace                    src/testbed_nerf.cu:2772
 *                               + shade_kernel_nerf                    src/testbed_nerf.cu:2448
 *   nrs_network_inference    <- NerfNetwork<T>::inference_mixed_precision   include/.../nerf_network_full.h:62
 *   nrs_network_density      <- NerfNetwork<T>::density                     include/.../nerf_network_full.h:223
 *   nrs_density_on_grid / nrs_rgba_on_grid <- Testbed::get_density_on_grid / get_rgba_on_grid   src/testbed_nerf.cu:4538 / :4588  ("next" row f4)
 *   nrs_model_set_params / _device <- NerfNetwork<T>::set_params            include/.../nerf_network_full.h:316 (host blob / device pointers)
 *   nrs_model_set_density_grid <- Testbed::update_density_grid_mean_and_bitfield   src/testbed_nerf.cu:3642
 *   nrs_edit_create          <- TetMesh GPU members + upload                       tet_mesh.h:80-94, tet_mesh.cu:651-667
 *   nrs_edit_update_cage / _vertices <- interpolate_with_mvc + build_tet_grid + update_local_rotations, on the device   ("next" row f1)
 *   nrs_edit_create_affine   <- AffineDuplication ctor + update_destination         editing/affine_duplication.h:26, :77   ("next" row f4)
 *   nrs_edit_map_rays        <- EditOperator::map_rays      edit_operator.h:43, CageDeformation::map_rays  cage_deformation.cu:547
 *   nrs_edit_map_positions   <- EditOperator::map_positions edit_operator.h:51, cage_deformation.cu:624
 *   nrs_edit_poisson_interpolate <- GrowingSelection::interpolate_poisson_boundary  src/editing/tools/growing_selection.cu:2350
 *   nrs_model_update_density_grid <- Testbed::update_density_grid_nerf_operator       src/testbed_nerf.cu:3533   ("next" row f2)
 *   nrs_snapshot_open        <- Testbed::load_snapshot / load_network_config        src/testbed.cu:3054 / :152       ("next" row f3)
 *   nrs_edits_open           <- Testbed::load_edits                                 src/testbed.cu:3205              ("next" row f3)
 *   nrs_tet_lut_build        <- TetMesh::build_tet_grid / build_original_tet_grid  tet_mesh.cu:368 / :76   (host, "next" row f1)
 *   nrs_mvc_compute / nrs_mvc_apply <- Cage::compute_mvc / interpolate_with_mvc    cage.cu:6 / :38          (host, "next" row f1)
 *   nrs_tet_local_rotations  <- TetMesh::update_local_rotations                    tet_mesh.cu:37           (host, "next" row f1)
 *   nrs_trace_samples        <- (test hook) the (t, dt) stream generate_next_nerf_network_inputs emits, testbed_nerf.cu:637
 *   nrs_accumulate           <- CudaRenderBuffer::accumulate / accumulate_kernel                                   src/render_buffer.cu:540 / :217
 *   nrs_detile               <- (new) inverse of the multi-GPU tile packing, no reference counterpart
 *
 * Conventions: every function returns NRS_OK (0) or a negative nrs_status; nrs_last_error() returns a
 * thread-local message.  All buffers named d_* are DEVICE pointers owned by the caller; h_* are HOST
 * pointers.  `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous
 * on `stream` unless stated; the caller synchronises (the reference does the same, src/testbed.cu:2843).
 * A context is not thread-safe; use one per host thread / GPU.  Render calls issued back to back on DIFFERENT streams may
 * overlap on the device (double-buffered frames; up to 8 in flight per context).  No torch / C++ types cross this boundary.
 */
#ifndef NRS_H
#define NRS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRS_ABI_VERSION 3   /* 3: nrs_render_params starts with struct_size (every entry point that takes the struct refuses a size other than its own: a client
                             * built against another header is turned away instead of being read past its struct); visualized_layer / _dimension appended.
                             * 2: dof / slice_plane_z / depth_scale / show_accel, camera model, background, glow */

typedef enum nrs_status {
	NRS_OK = 0,
	NRS_ERR_INVALID_ARG = -1,
	NRS_ERR_UNSUPPORTED = -2,   /* configuration outside configs/nerf/base.json's architecture */
	NRS_ERR_HIP = -3,           /* a HIP runtime call failed; message carries hipGetErrorString */
	NRS_ERR_NO_DEVICE = -4,     /* no gfx950 device visible: the product path never falls back to CPU */
	NRS_ERR_STATE = -5          /* e.g. rendering before params / density grid were set */
} nrs_status;

/* ENerfActivation, include/neural-graphics-primitives/common.h:107 */
typedef enum nrs_activation {
	NRS_ACT_NONE = 0, NRS_ACT_RELU = 1, NRS_ACT_LOGISTIC = 2, NRS_ACT_EXPONENTIAL = 3
} nrs_activation;

/* ERenderMode, common.h:71.  Implemented: AO, Shade, Positions, Depth, Distance, Stepsize, Distortion, Cost, Slice (composite_kernel_nerf's per-sample
 * branches testbed_nerf.cu:905-937, shade_kernel_nerf :2466-2482, init_rays' Distortion branch :2602-2613, the Slice path :3111-3175), and since ABI 3
 * Normals (the network's input gradient, :2924, :905-910, :2466-2468) and EncodingVis (visualized_layer / visualized_dimension, :2926, :925) -- the two
 * modes whose per-sample input lives in tiny-cuda-nn: restated like the forward pass (nrs_network_input_gradient / _visualize_activation below). */
typedef enum nrs_render_mode {
	NRS_RENDER_AO = 0, NRS_RENDER_SHADE = 1, NRS_RENDER_NORMALS = 2, NRS_RENDER_POSITIONS = 3,
	NRS_RENDER_DEPTH = 4, NRS_RENDER_DISTANCE = 5, NRS_RENDER_STEPSIZE = 6, NRS_RENDER_DISTORTION = 7,
	NRS_RENDER_COST = 8, NRS_RENDER_SLICE = 9, NRS_RENDER_ENCODING_VIS = 11 /* ERenderMode::EncodingVis sits behind NumRenderModes (10) */
} nrs_render_mode;

/* GPUMatrixDynamic<T>::layout() of the network output: planes = row-major [16 x n_el]
 * (renderer, density-grid update), interleaved = column-major, 16 halfs per sample (selection, Poisson). */
typedef enum nrs_layout { NRS_PLANES = 0, NRS_INTERLEAVED = 1 } nrs_layout;

/* The two roundings of tiny-cuda-nn that cannot be read off the reference checkout (its tiny-cuda-nn submodule is empty and un-pinned), switchable
 * per model (nrs_model_set_numerics).  Defaults are the wider ones.
 *   grid: FP32    = the trilinear sum of a level runs in fp32 (one fmaf per corner, corners 0..7) and is rounded to fp16 once;
 *         NETWORK = kernel_grid as recalled from NVlabs/tiny-cuda-nn of 2022: per corner `result[f] += (T)(weight * (float)value[f])`, T = __half
 *                   -- the fp32 product is rounded to fp16 and accumulated in fp16.
 *   mlp:  FP32    = fp16 x fp16 products accumulated in fp32 (MFMA accumulators), one rounding per layer output;
 *         FP16    = FullyFusedMLP's wmma fragments carry fp16 accumulators: modelled as one fp16 rounding of the running sum after every 16-wide
 *                   k block (how a tensor core sums inside a block is unspecified: a model of it, within tolerance of any implementation). */
typedef enum nrs_grid_acc { NRS_GRID_ACC_FP32 = 0, NRS_GRID_ACC_NETWORK = 1 } nrs_grid_acc;
typedef enum nrs_mlp_acc { NRS_MLP_ACC_FP32 = 0, NRS_MLP_ACC_FP16 = 1 } nrs_mlp_acc;

/* Network hyper-parameters: configs/nerf/base.json:23-58 + src/testbed.cu:2257-2333.
 * Accepted: base.json's family -- the 16 x 2 hash grid with any table size (base_14 / small / base / big.json: log2_hashmap_size 14 / 15 / 19 / 21), the 64-wide
 * density network with one hidden layer (or none: one [16 x 32] matrix, linear.json), and an rgb network of 0 (CutlassMLP, base_0layer.json), 1, 2 (base.json) or 3 hidden layers (base_{1,2,3}layer.json) on the
 * degree-4 spherical harmonics -- or NO direction encoding and rgb network (base_nodir.json -> NerfNetworkNoDir, testbed.cu:2314-2353): sh_degree = 0,
 * rgb_hidden_layers = 0; the colour is then the density network's outputs 1..3 (nerf_network_nodir.h:47-91).  Parameter blob: [density | rgb | grid] with the rgb part
 * [64 x 32] + (L - 1) [64 x 64] + [16 x 64] for L >= 1 hidden layers, one [8 x 32] matrix for L = 0, nothing for NoDir (tiny-cuda-nn's layouts as recalled;
 * nrs_model_n_params says what a description implies).  Other encodings (configs/nerf/{frequency,densegrid,tensor,...}.json) are refused: NRS_ERR_UNSUPPORTED. */
typedef struct nrs_model_desc {
	uint32_t n_levels;             /* 16 */
	uint32_t n_features_per_level; /* 2  */
	uint32_t log2_hashmap_size;    /* 19 */
	uint32_t base_resolution;      /* 16 */
	float    per_level_scale;      /* exp(ln(2048*aabb_scale/16)/15), testbed.cu:2288-2292 */
	uint32_t n_neurons;            /* 64 (both MLPs) */
	uint32_t density_hidden_layers;/* 1  (0..1) */
	uint32_t density_output_dims;  /* 16, nerf_network_full.h:47-49 */
	uint32_t rgb_hidden_layers;    /* 2  (0..3; 0 with sh_degree 0) */
	uint32_t sh_degree;            /* 4  (0 = NerfNetworkNoDir) */
	uint32_t rgb_activation;       /* nrs_activation; lego: Logistic   (testbed.h:636) */
	uint32_t density_activation;   /* nrs_activation; Exponential      (testbed.h:637) */
	float    aabb_min[3];          /* Testbed::m_aabb (training box): [0,1]^3 for aabb_scale 1 */
	float    aabb_max[3];
} nrs_model_desc;

#define NRS_GRID_SIZE       128u                     /* NERF_GRIDSIZE,  common_nerf.h:16 */
#define NRS_GRID_CASCADES   5u                       /* NERF_CASCADES,  common_nerf.h:26 */
#define NRS_GRID_VOLUME     (128u * 128u * 128u)
#define NRS_BITFIELD_BYTES  (NRS_GRID_VOLUME * NRS_GRID_CASCADES / 8u)   /* 1 310 720 */
#define NRS_NETWORK_INPUT_FLOATS 7u                  /* NerfCoordinate: pos3, dt, dir3; nerf.h:73 */
#define NRS_NETWORK_OUTPUT_WIDTH 16u                 /* padded_output_width() */

/* One cage-deformation operator as the renderer consumes it: the GPU members of TetMesh (tet_mesh.h:80-94)
 * plus the flags CageDeformation::map_rays passes (cage_deformation.cu:547-572).  All arrays are HOST
 * pointers; nrs_edit_create copies them to the device. */
typedef struct nrs_tet_mesh {
	uint32_t        n_vertices;
	uint32_t        n_tets;
	const float*    h_vertices;          /* [V*3] deformed, un-warped world units */
	const float*    h_original_vertices; /* [V*3] canonical */
	const uint32_t* h_tets;              /* [T*4] */
	const uint32_t* h_lut_offsets;       /* [5*128^3 + 1] CSR over level*128^3 + morton, deformed mesh */
	const uint32_t* h_lut_idx;           /* [h_lut_offsets[last]] */
	const uint8_t*  h_original_bitfield; /* [NRS_BITFIELD_BYTES] cells touched by the canonical mesh */
	const float*    h_local_rotations;   /* [T*9] Eigen column-major, or NULL (m_correct_direction off) */
	uint32_t        copy;                /* GrowingSelection::m_copy: keep the source visible */
	/* membrane ("Poisson") correction, cage_deformation.h:163; arrays may be NULL when apply_poisson == 0 */
	uint32_t        apply_poisson;
	float           residual_amplitude;
	const float*    h_boundary_shs;              /* [V*27] SH9RGB per vertex, Eigen col-major 9x3 */
	const float*    h_boundary_outside_density;  /* [V] */
	const float*    h_boundary_residual_density; /* [V] */
	/* Device-side authoring ("next" row f1).  h_lut_offsets == NULL: nrs_edit_create builds the cell->tet LUT of
	 * h_vertices on the device (TetMesh::build_tet_grid, tet_mesh.cu:368); h_original_bitfield == NULL: likewise the
	 * touched cells of h_original_vertices (build_original_tet_grid, :76); h_local_rotations == NULL with
	 * correct_direction != 0: the rotations are computed on the device (update_local_rotations, :37) and kept current
	 * by nrs_edit_update_*.  (m_correct_direction, cage_deformation.h) */
	uint32_t        correct_direction;
} nrs_tet_mesh;

/* AffineDuplication operator (include/.../editing/affine_duplication.h:23-106): the content of an oriented selection box
 * is shown again translated / scaled / rotated.  Fields = what the operator's JSON stores (affine_duplication.cu:356-369);
 * of the selection AffineBoundingBox only center / scale / rot_matrix matter (warp_box and update_destination rebuild
 * u, v, w, min, max from them, affine_bounding_box.cuh:40-101).  Matrices are Eigen column-major 3x3, world units. */
typedef struct nrs_affine_duplication {
	float    selection_center[3];
	float    selection_scale[3];   /* edge lengths of the box */
	float    selection_rot[9];
	float    translation[3];
	float    scale[3];
	float    rotation[9];
	uint32_t hide_original;        /* m_hide_original (false) */
	uint32_t correct_dir;          /* m_correct_dir (true): rotate the view direction too */
} nrs_affine_duplication;

/* Arguments + implicit Testbed members of render_nerf (SURVEY 8b "Renderer"). */
typedef struct nrs_render_params {
	uint32_t struct_size;         /* = sizeof(nrs_render_params) of the header the caller was built with (NRS_RENDER_PARAMS_INIT); anything else is
	                               * NRS_ERR_INVALID_ARG -- the library never reads a struct of another layout */
	int32_t  resolution[2];       /* render_buffer.in_resolution() */
	float    focal_length[2];
	float    camera_matrix0[12];  /* 3x4, Eigen column-major: col0,col1,col2 = axes, col3 = origin */
	float    camera_matrix1[12];
	float    rolling_shutter[4];
	float    screen_center[2];
	float    render_aabb_min[3];  /* m_render_aabb */
	float    render_aabb_max[3];
	uint32_t spp_index;           /* render_buffer.spp(): Sobol sample index */
	uint32_t snap_to_pixel_centers;
	float    min_transmittance;   /* m_nerf.rendering_min_transmittance (0.01) */
	float    cone_angle_constant; /* m_nerf.cone_angle_constant: 0 for aabb_scale 1, 1/256 otherwise */
	uint32_t render_mode;         /* nrs_render_mode (m_render_mode); see the enum for what is implemented */
	uint32_t linear_colors;       /* m_nerf.training.linear_colors: skip srgb_to_linear in shade */
	uint32_t apply_operators;     /* m_enable_edits && !m_distill */
	uint32_t poisson_target;      /* NerfTracer::m_poisson_target */
	uint32_t min_mip;             /* show_accel >= 0 ? show_accel : 0 (marching only) */
	uint32_t max_march_steps;     /* 0 = reference bound (MARCH_ITER, testbed_nerf.cu:56) */
	/* Multi-GPU image-tile sharding (no reference counterpart).  The image is cut into
	 * tile_size x tile_size pixel tiles; tile (Tx, Ty) has the index t = Ty * pitch + Tx with the ODD row pitch
	 * pitch = ceil(W / tile_size) | 1 (nrs_render_tile_pitch) -- with an even pitch and a power-of-two number of ranks, t mod ranks would put
	 * every rank on two columns of tiles; an odd pitch turns the deal into diagonals (max / mean samples per rank at 8 ranks, 32-pixel tiles:
	 * 1.032 -> 1.016).  Indices whose Tx lies beyond the image are virtual (no pixels).  This call renders tiles
	 * t = tile_first, tile_first + tile_stride, ...  tile_size == 0 means "whole image" and frame/depth
	 * are indexed x + W*y.  With tiling, pixel (tx,ty) of the k-th owned tile is written at
	 * ((k * tile_size + ty) * tile_size + tx): a compact buffer ready for one RCCL gather. */
	uint32_t tile_size;           /* multiple of 8, or 0 */
	uint32_t tile_first;
	uint32_t tile_stride;
	/* ---- ABI 2: the remaining implicit Testbed members of render_nerf.  All zero = the behaviour of ABI 1. ---- */
	float    dof;                 /* m_dof: aperture radius of pixel_to_ray's thin-lens branch (common_device.cuh:285-293); 0 = pinhole */
	float    slice_plane_z;       /* m_slice_plane_z + m_scale (testbed_nerf.cu:3067): focus distance of the aperture branch, and the distance
	                               * of the slice plane in render mode Slice (the call negates it itself, :3068-3070) */
	float    depth_scale;         /* 1 / m_nerf.training.dataset.scale (:3113): factor of render modes Depth and Distance */
	uint32_t show_accel;          /* m_nerf.show_accel >= 0 (the level is min_mip): every sample becomes opaque (:788-790) and render mode
	                               * Positions colours the occupancy cell (:911-920) */
	/* camera model and background (init_rays_with_payload_kernel_nerf :2523-2533, 2585-2613; pixel_to_ray common_device.cuh:262-280).  The pointers are
	 * DEVICE pointers for nrs_render_nerf (HOST pointers for the CPU oracle, which takes the same struct); NULL / 0 = absent. */
	uint32_t distortion_mode;     /* ECameraDistortionMode, common.h:166: 0 None, 1 Iterative (OpenCV k1 k2 p1 p2, Newton undistortion :162-199), 2 FTheta (:231-243)
	                               * = m_nerf.render_distortion when m_nerf.render_with_camera_distortion (:3078-3080) */
	float    distortion_params[7];
	const float* d_distortion_map;        /* m_distortion.map->params_inference(): float2 [res.y][res.x], added to the ray's xy (read_image<2>, :278-280) */
	int32_t  distortion_resolution[2];
	int32_t  envmap_resolution[2];
	const float* d_envmap;                /* m_envmap.envmap->params_inference(): float RGBA [res.y][res.x]; every pixel's frame value is REPLACED by the
	                                       * environment seen along its ray before the NeRF composites over it (read_envmap, envmap.cuh:30-63; :2590-2592) */
	uint32_t glow_mode;           /* m_nerf.m_glow_mode: composite_kernel_nerf's grid / cut-line overlay (:806-903); bits 1 green grid, 2 cut line, 4 mask to
	                               * alpha, 8 radial, 16 grid only.  0 = off (the reference's default) */
	float    glow_y_cutoff;       /* m_nerf.m_glow_y_cutoff */
	/* ---- ABI 3: render mode EncodingVis (render_mode = m_visualized_dimension > -1 ? EncodingVis : m_render_mode, testbed_nerf.cu:3072) ---- */
	uint32_t visualized_layer;     /* m_visualized_layer: 0 = hash-grid output (32 wide), 1 = density MLP hidden layer (64), 2 = rgb network input
	                                * (16 density outputs | 16 SH coefficients), 3 / 4 = rgb MLP hidden layers (64); NerfNetworkFull::forward_activations,
	                                * nerf_network_full.h:523-534 */
	uint32_t visualized_dimension; /* m_visualized_dimension (>= 0): the unit of that layer whose activation is shown (< NerfNetworkFull::width(layer), :507-517) */
} nrs_render_params;
/* envmap / distortion map contract: d_envmap holds envmap_resolution[0] x envmap_resolution[1] float4 texels, d_distortion_map
 * distortion_resolution[0] x distortion_resolution[1] float2 texels (both >= 1 x 1, row-major); every lookup clamps (y; the envmap wraps in x) to that extent. */
#define NRS_RENDER_PARAMS_INIT { (uint32_t)sizeof(nrs_render_params) }

typedef struct nrs_render_stats {
	uint64_t n_samples;      /* network-evaluated live samples (sum of per-ray n_steps)            */
	uint32_t n_rays_alive;   /* rays that found an occupied cell (entered the sample loop)          */
	uint32_t n_rays_hit;     /* rays shaded into the frame buffer (alpha > 0.001), = trace()'s n_hit */
} nrs_render_stats;

/* State update_density_grid_nerf_operator reads and advances on Testbed (src/testbed_nerf.cu:3533-3640), handed over
 * explicitly.  One call = one iteration of update_density_grid_nerf_render's loop (:3514-3520), which passes
 * n_uniform = 128^3 * (max_cascade + 1), n_nonuniform = 0 and reset_grid on its first iteration. */
typedef struct nrs_grid_update {
	uint32_t n_uniform_samples;    /* cells drawn with threshold -0.01 (every trained cell)            :3565 */
	uint32_t n_nonuniform_samples; /* cells drawn with threshold NERF_MIN_OPTICAL_THICKNESS (occupied)  :3578 */
	uint32_t reset_grid;           /* zero the float grid first                                         :3558 */
	uint32_t max_cascade;          /* m_nerf.max_cascade: cascades 0..max_cascade are sampled                 */
	float    decay;                /* m_nerf.training.density_grid_decay, 0.95 (testbed.h:604)                */
	uint32_t ema_step;             /* IN/OUT m_nerf.density_grid_ema_step (incremented by the call)     :3636 */
	uint64_t rng_state;            /* IN/OUT m_rng (tcnn::pcg32) state; advanced by 2 * 2^32            :3577,3590 */
	uint64_t rng_inc;              /* m_rng stream constant                                                   */
} nrs_grid_update;

typedef struct nrs_ctx   nrs_ctx;
typedef struct nrs_model nrs_model;
typedef struct nrs_edit  nrs_edit;

const char* nrs_last_error(void);
int         nrs_abi_version(void);

/* ---- context / model ------------------------------------------------------------------------------- */
int  nrs_ctx_create(int device, nrs_ctx** out);
void nrs_ctx_destroy(nrs_ctx* ctx);
int  nrs_ctx_device_info(const nrs_ctx* ctx, char* name_out, size_t name_len, int* n_cus, size_t* hbm_bytes);
/* Lane teams: how many lanes of a wavefront share one ray in nrs_render_nerf.  0 (default) = automatic: 2 or 4 when a
 * launch owns so few pixels (one GPU's tiles of a frame sharded over 4-8 GPUs, small viewports) that its duration would
 * otherwise be one ray's latency chain; one lane per ray for launches that fill the GPU, with teams only for the last
 * third of the frame's work queue ("hybrid", whole-image mode).  1 / 2 / 4 force a size for every ray, -1 forces the
 * hybrid schedule, -2 / -3 / -4 the small-launch schedule (team size chosen per generation from the rays a wave has pending) with packets of 16 / 32 / 64 pixels.
 * Since round 3 the automatic choice is the small-launch schedule with packets sized by the launch; the hybrid schedule runs only when forced.  Pixel values, depth, step counts and statistics do not depend on it (tests/test_gpu_lane_teams.py). */
int  nrs_ctx_set_lane_teams(nrs_ctx* ctx, int lanes_per_ray);
/* Ray hand-over (on by default): in whole-image and small-launch ("hybrid") schedules a wave that has run out of work takes rays from a sibling wave of its
 * workgroup -- rays that wait in the sibling's ring for its next generation, or half of the rays it holds in lanes (both halves then run with more lanes per
 * ray) -- instead of leaving its lanes idle for the frame's tail.  Results do not depend on it (tests/test_gpu_lane_teams.py).  nrs_ctx_ray_handovers reports
 * what the last render launch that returned statistics (h_stats != NULL) handed over: rays moved, hand-overs. */
int  nrs_ctx_set_ray_handover(nrs_ctx* ctx, int enabled);
int  nrs_ctx_ray_handovers(const nrs_ctx* ctx, uint64_t* n_rays, uint64_t* n_handovers);

/* MEMORY NOTE: a model holds its parameters only (24-27 MB).  The cell-record cache -- a memory-for-instructions trade, about 15 % of a lego frame (bench.py key
 * lego_cage_norecords) -- is OPT-IN: nrs_model_set_cell_cache(model, budget) below (10 GiB holds levels 0..11 of base.json's table: 9.2 GB), allocated and filled inside
 * that call and rebuilt inside every later nrs_model_set_params (1.9 ms for 9.2 GB); a caller whose parameters change every frame -- a training viewer -- leaves it off.
 * Results are bit-identical either way.  NRS_CELL_CACHE_GB in the environment switches it on at nrs_model_create for a host that cannot be changed. */
int    nrs_model_create(nrs_ctx* ctx, const nrs_model_desc* desc, nrs_model** out);
void   nrs_model_destroy(nrs_model* model);
/* number of fp16 parameters the description implies (density MLP | rgb MLP | hash grid), host-only */
size_t nrs_model_n_params(const nrs_model_desc* desc);
/* per-level hash-grid table (host-only): scale, resolution, first entry, entry count, hashed flag */
int    nrs_model_level_table(const nrs_model_desc* desc, float* scale, uint32_t* resolution,
                             uint32_t* entry_offset, uint32_t* entry_count, uint32_t* hashed);
/* fp16 parameter blob in tiny-cuda-nn order: density MLP | rgb MLP | hash grid (nerf_network_full.h:316-349).
 * h_params is a HOST pointer (what Trainer::deserialize hands over); synchronous. */
int    nrs_model_set_params(nrs_model* model, const void* h_params_fp16, size_t n_params);
/* The same from a DEVICE pointer, as NerfNetworkFull::set_params receives it (nerf_network_full.h:316-349: pointers into the trainer's blob).
 * Asynchronous: the hash grid is copied device-to-device, the MLP weights are re-arranged into MFMA fragments by a small kernel and the cell
 * records (if any are kept) rebuilt, all enqueued on `stream`; launches enqueued on the same stream afterwards see the new parameters, and the
 * blob may be overwritten once the stream has passed this call.  Copy semantics: call again after every optimiser step. */
int    nrs_model_set_params_device(nrs_model* model, const void* d_params_fp16, size_t n_params, void* stream);
/* nrs_grid_acc / nrs_mlp_acc above.  Applies to every entry point that evaluates the network -- nrs_render_nerf in every schedule, mode and operator
 * combination, the network operators, the occupancy refresh, the grid evaluators, selection rays, the membrane boundary (round 3: no combination is
 * refused any more; the non-default modes run the run-time twins of the kernels, a little slower than the default instantiations). */
int    nrs_model_set_numerics(nrs_model* model, uint32_t grid_acc, uint32_t mlp_acc);
/* Cell-record cache (no counterpart in the reference: a memory-for-bandwidth trade the 288 GB of HBM allow).  For the
 * coarsest levels that fit `max_bytes` (an even number of them), every grid cell gets a 32-byte record holding its 8
 * corner entries, fetched with the level's own index function (tiny-cuda-nn grid.h:76-95), so that a sample reads two
 * 16-byte loads from one cache line instead of hashing eight corners and gathering them from four lines.  Results are
 * bit-identical with or without the cache.  The records are rebuilt inside every nrs_model_set_params (a few ms per
 * 10 GB); callers that change parameters every frame (training) pass 0.  Levels 0..11 of base.json's table take
 * 9.3 GB, 0..13 take 64 GB.  Synchronises the device.  max_bytes = 0 drops the cache. */
int    nrs_model_set_cell_cache(nrs_model* model, size_t max_bytes);
size_t nrs_model_cell_cache_bytes(const nrs_model* model, uint32_t* n_levels_out);
/* Sparse cell records for levels the dense cache cannot hold (aabb_scale-16 scenes: the fine levels of a 16^3-unit box).  The cells of
 * a level are grouped in 8 x 8 x 8 bricks (16 KiB of records); a brick is allocated if it touches a cell marked in h_mask_bitfield (the
 * density-bitfield layout, NRS_BITFIELD_BYTES): pass the occupancy of every place hash-grid lookups can happen -- the current occupancy,
 * OR-ed with the un-edited one when edit operators carry samples back to canonical space.  A sample whose brick has no records gathers the
 * hashed way, so results never depend on the mask, only speed does.  Levels are taken in pairs after the dense ones while brick tables +
 * records fit max_bytes.  Kept current by nrs_model_set_params; dropped by nrs_model_set_cell_cache (set the dense budget first).
 * h_mask_bitfield == NULL or max_bytes == 0 drops them.  Synchronises the device. */
int    nrs_model_set_sparse_cell_cache(nrs_model* model, const uint8_t* h_mask_bitfield, size_t max_bytes);
size_t nrs_model_sparse_cell_cache_bytes(const nrs_model* model, uint32_t* first_level_out, uint32_t* n_levels_out);
/* occupancy: either the ready-made bitfield (NRS_BITFIELD_BYTES, Morton order, mips pooled) ... */
int    nrs_model_set_density_bitfield(nrs_model* model, const uint8_t* h_bitfield, size_t n_bytes);
/* ... or the float density grid [5*128^3]; thresholded with min(0.01, mean) and OR-pooled on the device
 * exactly as update_density_grid_mean_and_bitfield does (testbed_nerf.cu:514-555, 3642-3657). */
int    nrs_model_set_density_grid(nrs_model* model, const float* h_grid, size_t n_floats);
int    nrs_model_get_density_bitfield(nrs_model* model, uint8_t* h_bitfield_out, size_t n_bytes);
/* Test hook: the marching accelerator the library derived from the bitfield (no reference counterpart; the reference marches cell by
 * cell, testbed_nerf.cu:1100-1131).  which 0: the flavour used for general step parameters, 1: for cone_angle == 0 && min_mip == 0.
 * h_box12 = {min[3], max[3], cell[3], 1/cell[3]} of the occupied box and its 32^3 look-ahead blocks, h_mask = 32^3 / 32 words. */
int    nrs_model_get_march_accelerator(nrs_model* model, int which, float* h_box12, uint32_t* h_mask1024);
/* The float grid the bitfield was last derived from ([5*128^3]; zeros if only a bitfield was ever set). */
int    nrs_model_get_density_grid(nrs_model* model, float* h_grid_out, size_t n_floats);
/* Deformed-space occupancy refresh ("next" row f2): Testbed::update_density_grid_nerf_operator, testbed_nerf.cu:3533.
 * Draws grid-cell samples (generate_grid_samples_nerf_nonuniform, common_nerf.cu:179), maps them through the edit
 * operators last-to-first (map_positions, cage_deformation.cu:624), evaluates density(), empties masked samples
 * (clear_empty_space :2759), activates, adds the membrane residual (compute_poisson_residual_density,
 * cage_deformation.cu:645), max-splats per cell (:447), decays (ema_grid_samples_nerf :483) and rebuilds mean,
 * bitfield and mips (:3642) -- all on the device, one fused kernel for everything up to the splat.
 * Synchronous (the reference syncs m_inference_stream at :3519); updates *u. */
int    nrs_model_update_density_grid(nrs_model* model, nrs_edit* const* edits, int n_edits, nrs_grid_update* u, void* stream);
/* tcnn::pcg32(seed) -- Testbed seeds m_rng = default_rng_t{m_seed = 1337} (src/testbed.cu:2220). host-only */
void   nrs_rng_seed(uint64_t seed, uint64_t* state_out, uint64_t* inc_out);

/* ---- multi-GPU: one exchange step per frame (the reference is single-GPU, README.md:423-425) --------------------------- */
/* The frame is cut into tile_size x tile_size tiles dealt round-robin to the ranks (nrs_render_params.tile_*); every rank renders into a compact
 * buffer [tiles_padded * tile^2 * 4 floats of frame | tiles_padded * tile^2 floats of depth]; nrs_gather_tiles moves them to the root with
 * ncclGroupStart / ncclSend / ncclRecv x (N - 1) / ncclGroupEnd (RCCL point-to-point over xGMI) on `stream` and de-tiles there.
 * RCCL is dlopen'ed on first use (no link-time dependency).  Bootstrap as with NCCL: rank 0 makes a 128-byte id, the application distributes
 * it (its own channel), every rank creates the communicator with it (collective call). */
typedef struct nrs_comm nrs_comm;
int  nrs_comm_unique_id(uint8_t* out128);
int  nrs_comm_create(int device, int rank, int n_ranks, const uint8_t* unique_id128, nrs_comm** out);
void nrs_comm_destroy(nrs_comm* comm);
/* What the communicator is: this process's rank, the number of ranks RCCL connected, ncclGetVersion() (0 if the library has none) and the library
 * that was loaded (NRS_RCCL_LIB overrides the search: tests/fake_rccl runs several ranks on one GPU).  Any pointer may be NULL. */
int  nrs_comm_info(const nrs_comm* comm, int* rank_out, int* n_ranks_out, int* rccl_version_out, char* lib_path_out, size_t lib_path_len);
/* Diagnostic (no reference counterpart): `pairs` ncclSend / ncclRecv pairs of n_floats floats from this rank to itself in one group on `stream` -- RCCL's
 * enqueue path of nrs_gather_tiles with one rank, for timing its host-side cost on a one-GPU box (tools/gather_probe.py). d_src / d_dst: pairs * n_floats. */
int  nrs_comm_probe_self_p2p(nrs_comm* comm, const float* d_src, float* d_dst, size_t n_floats, int pairs, void* stream);
/* d_local: this rank's buffer.  Root only: d_recv = n_ranks such buffers (rank-major); d_image [H*W*4] / d_depth [H*W] may be NULL. */
int  nrs_gather_tiles(nrs_ctx* ctx, nrs_comm* comm, int root, const nrs_render_params* p, uint32_t tiles_per_rank_padded, const float* d_local,
                      float* d_recv, float* d_image, float* d_depth, void* stream);

/* ---- NerfNetwork operator -------------------------------------------------------------------------- */
/* d_in: [n x 7] f32 (column-major 7 x n in tcnn terms).  d_out: fp16, n_el = n_padded samples wide:
 * planes -> d_out[c * ld_out + s], interleaved -> d_out[s * 16 + c]; c 0..2 rgb raw, c 3 density raw,
 * c 4..15 the rgb network's padding outputs.  ld_out >= n (planes only). */
int nrs_network_inference(nrs_model* model, void* stream, uint32_t n, const float* d_in,
                          void* d_out_fp16, uint32_t ld_out, int layout);
/* density(): d_in is [n x ld_in] f32 with ld_in 3..7, only floats 0..2 of each record are read
 * (nerf_network_full.h:231-236).  Output = the density MLP's 16 outputs (c 0 = density raw). */
int nrs_network_density(nrs_model* model, void* stream, uint32_t n, const float* d_in, uint32_t ld_in,
                        void* d_out_fp16, uint32_t ld_out, int layout);
/* The network's introspection entry points (tiny-cuda-nn's, restated: the submodule is absent from the reference checkout -- oracle/nrs_oracle.cpp
 * density_input_gradient_one / network_activation_one carry the algorithm and its provenance):
 * nrs_network_input_gradient <- NerfNetwork::input_gradient(stream, 3, positions, gradients) (render mode Normals, src/testbed_nerf.cu:2924; mesh vertex
 *   normals :4491): d density_raw / d position of every sample through the density MLP and the hash grid, backprop scale 128 as in tiny-cuda-nn,
 *   rows 0..2 of the result ([n x 3] f32; rows 3..6 -- dt and the direction -- are zero in the reference's 7-row matrix: the density does not depend on them).
 * nrs_network_visualize_activation <- Network::visualize_activation(stream, layer, dimension, input, output) (render mode EncodingVis, :2926, :3159): unit
 *   `dimension` of NerfNetworkFull::forward_activations(layer) (nerf_network_full.h:523-534: 0 hash grid [32], 1 density hidden [64], 2 rgb network input
 *   = 16 density outputs | 16 SH coefficients, 3 / 4 rgb hidden [64]) as f32 [n]; the reference's output kernel writes max(-v, 0), max(v, 0), 0, 1, 1, 1, 1
 *   into a 7-row matrix -- the caller that wants that picture forms it from v.  d_in is [n x 7] f32. */
int nrs_network_input_gradient(nrs_model* model, void* stream, uint32_t n, const float* d_in, uint32_t ld_in, float* d_grad_out_nx3);
int nrs_network_visualize_activation(nrs_model* model, void* stream, uint32_t layer, uint32_t dimension, uint32_t n, const float* d_in, float* d_out_n);
/* The network on a regular grid ("next" row f4: the marching-cubes / volume-export callers of the operator).
 * nrs_density_on_grid <- Testbed::get_density_on_grid (src/testbed_nerf.cu:4538): point (x,y,z) of the res3d grid sits at
 *   aabb_min + (x/rx, y/ry, z/rz) * (aabb_max - aabb_min); d_out[x + y*rx + z*rx*ry] = raw density (fp16 network output as float),
 *   or -10000 where the resident density grid is below NERF_MIN_OPTICAL_THICKNESS at that position (mask_with_density_grid != 0,
 *   grid_samples_half_to_float :464).
 * nrs_rgba_on_grid <- Testbed::get_rgba_on_grid (:4588): the grid spans the render box, every point is seen from ray_dir;
 *   d_out_rgba[i] = (rgb * a, a), a = clamp(1 - exp(-density / 100), 0, 1)  (compute_nerf_density :624). */
int nrs_density_on_grid(nrs_model* model, void* stream, const uint32_t res3d[3], const float aabb_min[3], const float aabb_max[3],
                        int mask_with_density_grid, float* d_out);
int nrs_rgba_on_grid(nrs_model* model, void* stream, const uint32_t res3d[3], const float render_aabb_min[3],
                     const float render_aabb_max[3], const float ray_dir[3], float* d_out_rgba);
/* Selection tool, first step ("next" row f4) <- GrowingSelection::project_selection_pixels (growing_selection.cu:1832-2035:
 * shoot_selection_rays_kernel :1673 + NerfNetwork::density + composite_shot_rays :1768), one launch.  For every scribbled
 * pixel (x, y of `params`' resolution; camera = params->camera_matrix1, focal_length, screen_center, cone_angle_constant)
 * a ray is stepped through the occupied cells of the model's train box (pixel_to_ray with spp 0, direction not
 * normalised, up to NERF_STEPS = 1024 samples) while the density composites; the first sample reached with
 * transmittance <= threshold (the reference's 0.1) gives d_positions[i] (world space) and d_cells[i] =
 * mip * 128^3 + morton(cell); d_found[i] = 0 and position = aabb_min - 1 when the ray never gets there.  The camera is a PINHOLE whatever the lens fields
 * of `params` say: the reference's shoot_selection_rays_kernel calls pixel_to_ray with its default arguments (growing_selection.cu:1696-1703: no distortion,
 * no aperture, spp 0), so dof / distortion_* / d_distortion_map are not read here -- unlike nrs_render_nerf and nrs_trace_samples, which march the lens
 * the renderer uses. */
int nrs_project_selection_pixels(nrs_model* model, void* stream, const nrs_render_params* params, const int32_t* d_pixels_xy,
                                 uint32_t n_pixels, float transmittance_threshold, float* d_positions, uint32_t* d_cells, uint8_t* d_found);
/* host-only bookkeeping that follows (:1964-2021): automatic growing level = highest cascade found, cells below it lifted
 * to it (get_upper_cell_idx, selection_utils.cu:36), duplicates dropped; outputs in pixel order, sized n. */
uint32_t nrs_upper_cell_idx(uint32_t cell_idx, uint32_t target_level);
int nrs_selection_cells(const float* h_positions, const uint32_t* h_cells, const uint8_t* h_found, uint32_t n, int automatic_max_level,
                        uint32_t* growing_level_inout, uint32_t* out_cells, float* out_positions, uint32_t* n_out);
/* Membrane ("Poisson") boundary values of a proxy cage ("next" row f4) <- GrowingSelection::compute_poisson_boundary
 * (growing_selection.cu:2220-2348): for every cage vertex, sh_width^2 directions on the sphere (stratified in (u, v) with one
 * jitter pair per sample -- the reference draws them with std::rand(); here the caller supplies them, [n_verts * sh_width^2][2]
 * in [0, 1]), the full network at the vertex seen from those directions, then the vertex's density (its first sample; zeroed
 * where the occupancy is empty when is_inside, filter_empty :2200) and the SH9 fit of the colours (project_sh9, times
 * 4 pi / n).  h_sh_out: [n_verts][27], SH9RGB column-major (coefficient k of colour c at 9 c + k) -- what the render path's
 * membrane correction consumes.  Host pointers, synchronous, like the reference.  nrs_poisson_sample_coords is the host half
 * on its own (the [n][7] network inputs), exported for tests. */
int  nrs_poisson_boundary(nrs_model* model, const float* h_vertices, uint32_t n_verts, uint32_t sh_width, uint32_t hemisphere_width,
                          const float* h_jitter, int is_inside, float* h_density_out, float* h_sh_out);
void nrs_poisson_sample_coords(const float* vertices, uint32_t n_verts, uint32_t sh_width, uint32_t hemisphere_width, const float* jitter,
                               const float aabb_min[3], const float aabb_max[3], float* coords7_out);
/* hash-grid encoding alone (test hook; tcnn Encoding::inference_mixed_precision): d_out [n x 32] fp16 */
int nrs_hashgrid_encode(nrs_model* model, void* stream, uint32_t n, const float* d_in, uint32_t ld_in,
                        void* d_out_fp16);

/* ---- edit operators -------------------------------------------------------------------------------- */
int  nrs_edit_create(nrs_ctx* ctx, const nrs_model_desc* desc, const nrs_tet_mesh* mesh, nrs_edit** out);
/* AffineDuplication(selection_box, translation, aabb) + update_destination (affine_duplication.h:26, 77-90); the
 * resulting nrs_edit is used exactly like a cage edit (map_rays / map_positions / render / occupancy refresh). */
int  nrs_edit_create_affine(nrs_ctx* ctx, const nrs_model_desc* desc, const nrs_affine_duplication* op, nrs_edit** out);
void nrs_edit_destroy(nrs_edit* edit);
/* map_rays: in-place on d_coords [n x 7] f32, OR-accumulates into d_empty_mask [n] u8 */
int  nrs_edit_map_rays(nrs_edit* edit, void* stream, uint32_t n, float* d_coords, uint8_t* d_empty_mask);
/* map_positions: in-place on d_pos [n x ld] f32 (ld >= 3), OR-accumulates into d_empty_mask */
int  nrs_edit_map_positions(nrs_edit* edit, void* stream, uint32_t n, float* d_pos, uint32_t ld,
                            uint8_t* d_empty_mask);

/* Per-gizmo-move chain on the device ("next" row f1).  The reference redoes all of this on the CPU for every move and
 * re-uploads ~45 MB (tet_mesh.cu:651-667); here only the cage vertices (or the tet vertices) cross PCIe.
 *   nrs_edit_set_mvc          <- Cage::compute_mvc's result (cage.cu:6), [V x n_cage_vertices] row-major, uploaded once
 *   nrs_edit_update_cage      <- Cage::interpolate_with_mvc (cage.cu:38) + TetMesh::post_update_vertices (tet_mesh.cu:12)
 *                                + build_tet_grid (:368) + update_local_rotations (:37)
 *   nrs_edit_update_vertices  <- the same chain from explicit deformed tet vertices (TetMesh::update_vertices)
 * Both synchronise `stream` (the LUT size and the bounding box come back to the host), like the reference's host code. */
int  nrs_edit_set_mvc(nrs_edit* edit, const float* h_weights, uint32_t n_cage_vertices);
int  nrs_edit_update_cage(nrs_edit* edit, void* stream, const float* h_cage_vertices, uint32_t n_cage_vertices);
int  nrs_edit_update_vertices(nrs_edit* edit, void* stream, const float* h_vertices, uint32_t n_vertices);
int  nrs_edit_lut_size(const nrs_edit* edit, uint32_t* n_idx_out, uint32_t* max_per_cell_out);
/* GrowingSelection::interpolate_poisson_boundary (src/editing/tools/growing_selection.cu:2350-2395): the link between nrs_poisson_boundary
 * (membrane terms per CAGE vertex, inside and outside) and the renderer's membrane path (terms per TET vertex): boundary_shs, outside density and
 * residual density of every tet vertex as gamma-weighted sums over the cage vertices.  h_gamma = TetMesh::gamma_coordinates [V x n_cage_vertices]
 * (= compute_mvc of the canonical vertices with mvc_gamma); NULL: the weights given to nrs_edit_set_mvc (mvc_gamma = 1, the default).
 * The operator then applies the membrane correction (apply_poisson, residual_amplitude) in every later render.  Synchronises `stream`. */
int  nrs_edit_poisson_interpolate(nrs_edit* edit, void* stream, const float* h_gamma, uint32_t n_cage_vertices, const float* h_inside_density,
                                  const float* h_outside_density, const float* h_inside_shs, const float* h_outside_shs, float residual_amplitude);
int  nrs_edit_download_poisson(nrs_edit* edit, float* h_boundary_shs, float* h_outside_density, float* h_residual_density);
/* read the device-side tables back (tests / inspection); any pointer may be NULL.  h_lut_idx holds n_idx entries,
 * h_bbox6 = min xyz, max xyz of the deformed mesh. */
int  nrs_edit_download(nrs_edit* edit, float* h_vertices, uint32_t* h_lut_offsets, uint32_t* h_lut_idx, float* h_rotations,
                       uint8_t* h_original_bitfield, float* h_bbox6);

/* ---- renderer -------------------------------------------------------------------------------------- */
/* d_frame: f32x4 premultiplied linear RGBA, pre-cleared by the caller (clear_frame, testbed.cu:2635);
 * d_depth: f32 (written 1e10 for every pixel of the owned tiles first, as init_rays does);
 * d_steps: optional u32 per pixel = samples composited (the reference's payload.n_steps, tn:957-960, minus the one it counts for a ray that ran out of samples), may be NULL.
 * edits are applied last-to-first (testbed_nerf.cu:2899).  h_stats may be NULL; when non-NULL the call
 * synchronises the stream before returning (the reference's trace() syncs to read n_hit). */
int nrs_render_nerf(nrs_model* model, const nrs_render_params* params, nrs_edit* const* edits, int n_edits,
                    float* d_frame, float* d_depth, uint32_t* d_steps, void* stream, nrs_render_stats* h_stats);
/* spp accumulation <- CudaRenderBuffer::accumulate (src/render_buffer.cu:540-560; accumulate_kernel :217-254): d_accumulate [H*W] f32x4 becomes the running mean
 * of the frames rendered so far for this view; sample_count = frames already in it (0: the buffer is overwritten, as the reference clears it first).  The caller
 * renders frame k with spp_index = k (the Sobol pixel offsets, snap_to_pixel_centers off) into a cleared d_frame and calls this: scripts/run.py's 8-spp test
 * images are 8 such rounds.  color_space = EColorSpace (common.h:122): Linear | SRGB (linear_to_srgb applied to the frame before the mean) | VisPosNeg (EncodingVis). */
typedef enum nrs_color_space { NRS_COLOR_LINEAR = 0, NRS_COLOR_SRGB = 1, NRS_COLOR_VISPOSNEG = 2 } nrs_color_space;
int nrs_accumulate(nrs_ctx* ctx, void* stream, uint32_t width, uint32_t height, const float* d_frame, float* d_accumulate, uint32_t sample_count, uint32_t color_space);
/* number of tiles this rank owns / pixels of the compact buffer for given params (host-only; virtual tiles of the odd pitch included) */
uint32_t nrs_render_owned_tiles(const nrs_render_params* params);
/* the row pitch of the tile index: ceil(W / tile_size) | 1 (0 when tile_size == 0) */
uint32_t nrs_render_tile_pitch(const nrs_render_params* params);
/* scatter compact tile buffers of all ranks (rank-major, as a gather delivers them) back into a full W x H image.
 * Rank r's tiles start at d_tiles + r * rank_stride_floats (0 = densely packed: tiles_per_rank_padded * tile^2 * channels);
 * a stride lets frame and depth share one gathered buffer [rank][frame block | depth block] -> one collective per frame. */
int nrs_detile(nrs_ctx* ctx, void* stream, const nrs_render_params* params, uint32_t n_ranks,
               uint32_t tiles_per_rank_padded, const float* d_tiles, uint32_t channels, size_t rank_stride_floats, float* d_image);
/* Test hook for bit-exact ray/sample indexing: for each listed pixel, march exactly as the renderer does
 * (init -> jitter -> first hit -> successive samples) ignoring compositing, and emit up to max_samples
 * (t, dt) pairs.  d_t, d_dt: [n_pixels x max_samples] f32; d_count: [n_pixels] u32. */
int nrs_trace_samples(nrs_model* model, const nrs_render_params* params, void* stream, uint32_t n_pixels,
                      const uint32_t* d_pixel_idx, uint32_t max_samples, float* d_t, float* d_dt, uint32_t* d_count);

/* ---- on-disk formats either side of the path ("next" row f3; host-only, no device needed) ------------------------ */
/* Snapshot: Testbed::load_network_config + load_snapshot (src/testbed.cu:152-184, 3054-3087).  `.msgpack` is
 * nlohmann::json::to_msgpack of the network config with a "snapshot" object; `.ingp` is the same behind zlib (zstr).
 * Read here: encoding / network / rgb_network / dir_encoding hyper-parameters, snapshot.nerf.aabb_scale (or
 * .dataset.aabb_scale), snapshot.params_binary (tcnn Trainer::serialize, fp16 or float), snapshot.density_grid_binary
 * (float [5*128^3] from save_snapshot, fp16 [(max_cascade+1)*128^3] from export_snapshot), snapshot.camera.matrix. */
typedef struct nrs_snapshot nrs_snapshot;
int          nrs_snapshot_open(const char* path, nrs_snapshot** out);
void         nrs_snapshot_close(nrs_snapshot* snapshot);
int          nrs_snapshot_model_desc(const nrs_snapshot* snapshot, nrs_model_desc* desc_out, uint32_t* aabb_scale_out);
const void*  nrs_snapshot_params_fp16(const nrs_snapshot* snapshot, size_t* n_params_out);   /* -> nrs_model_set_params */
const float* nrs_snapshot_density_grid(const nrs_snapshot* snapshot, size_t* n_floats_out);  /* -> nrs_model_set_density_grid */
int          nrs_snapshot_camera(const nrs_snapshot* snapshot, float* camera_matrix12_out);   /* column-major 3x4 */
/* Edits: Testbed::load_edits (src/testbed.cu:3205-3236): {"edit_operators": [{"type": "cage_deformation",
 * "proxy_cage": Cage (cage.h:100-145), "interpolation_mesh": TetMesh (tet_mesh.h:136-174), ...}]}.
 * nrs_edits_cage fills an nrs_tet_mesh for DEVICE authoring (vertices, original vertices, tets; LUT / bitfield /
 * rotations left NULL so nrs_edit_create builds them, as the reference's JSON constructor rebuilds the tet grid,
 * growing_selection.cu:112-115) and hands out the MVC weights and the proxy cage.  Pointers live as long as `edits`.
 * An operator saved before its cage was tetrahedralised has no interpolation mesh (growing_selection.cu:2477): mesh_out->n_tets == 0, the cage is still returned.
 * Empty lists may appear as `null` in the file (the reference's vector writers leave an empty vector's value untouched, json_binding.h:231-321): read as empty. */
typedef struct nrs_edits nrs_edits;
int          nrs_edits_open(const char* path, nrs_edits** out);
void         nrs_edits_close(nrs_edits* edits);
uint32_t     nrs_edits_count(const nrs_edits* edits);
const char*  nrs_edits_type(const nrs_edits* edits, uint32_t i);   /* "cage_deformation", "affine_duplication", "twist" */
int          nrs_edits_affine(const nrs_edits* edits, uint32_t i, nrs_affine_duplication* op_out);
int          nrs_edits_cage(const nrs_edits* edits, uint32_t i, nrs_tet_mesh* mesh_out, const float** h_mvc_weights_out,
                            const float** h_cage_vertices_out, const float** h_cage_original_vertices_out,
                            const uint32_t** h_cage_triangles_out, uint32_t* n_cage_vertices_out, uint32_t* n_cage_triangles_out);

/* ---- host-side edit authoring ("next" row f1; CPU like the reference, no device needed) -------------- */
typedef struct nrs_tet_lut nrs_tet_lut;
/* builds the cell->tet CSR of `h_vertices` and the touched-cell bitfield; n_threads 0 = hardware */
int  nrs_tet_lut_build(const float* h_vertices, uint32_t n_vertices, const uint32_t* h_tets, uint32_t n_tets,
                       int n_threads, nrs_tet_lut** out);
uint32_t        nrs_tet_lut_n_idx(const nrs_tet_lut* lut);
uint32_t        nrs_tet_lut_max_per_cell(const nrs_tet_lut* lut);
const uint32_t* nrs_tet_lut_offsets(const nrs_tet_lut* lut);   /* [5*128^3+1] */
const uint32_t* nrs_tet_lut_idx(const nrs_tet_lut* lut);
const uint8_t*  nrs_tet_lut_bitfield(const nrs_tet_lut* lut);  /* [NRS_BITFIELD_BYTES] */
void            nrs_tet_lut_destroy(nrs_tet_lut* lut);
/* MVC weights of n_points points w.r.t. a closed triangulated cage (mvc.h:125-188), float arithmetic.
 * h_weights_out [n_points x n_cage_vertices]; h_labels_out [n_points] (1 = degenerate case hit), may be NULL */
int nrs_mvc_compute(const float* h_cage_vertices, uint32_t n_cage_vertices, const uint32_t* h_cage_triangles,
                    uint32_t n_cage_triangles, const float* h_points, uint32_t n_points,
                    float* h_weights_out, uint8_t* h_labels_out);
/* points[i] = sum_j w[i][j] * cage[j]   (cage.cu:38-49) */
int nrs_mvc_apply(const float* h_weights, const float* h_cage_vertices, uint32_t n_cage_vertices,
                  uint32_t n_points, float* h_points_out);
/* per-tet deformed->canonical rotation R = U V^T of sum (orig-c0)(def-c1)^T (tet_mesh.cu:37-74); [T*9] col-major */
int nrs_tet_local_rotations(const float* h_vertices, const float* h_original_vertices, const uint32_t* h_tets,
                            uint32_t n_tets, float* h_rotations_out);

#ifdef __cplusplus
}
#endif
#endif /* NRS_H */
