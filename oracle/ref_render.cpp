// ref_render.cpp -- driver around the REFERENCE's own render-path code, compiled from where it lies under /root/reference.
// TEST INFRASTRUCTURE ONLY: built by oracle/Makefile into oracle/_ref/libref_render.so where the reference is mounted; used to pin
// oracle/nrs_oracle.cpp (tests/test_ref_pin.py) and to generate tests/golden/ref_*.npz.  Nothing of the product links or loads it.
//
// What is the reference's code here (compiled, not restated):
//   * whole headers: common.h, common_device.cuh (srgb_to_linear, pixel_to_ray), bounding_box.cuh (ray_intersect, contains,
//     intersects(Triangle)), triangle.cuh, random_val.cuh (Sobol, scramble, ld_random_*), nerf.h (NerfPayload, NerfCoordinate),
//     common_nerf.h, envmap.cuh, editing/tools/selection_utils.h (bary_tet, point_in_tet);
//   * whole file: src/common_nerf.cu (step / mip / Morton-cell / warp math, activations, evaluate_sh9, grid sample generator);
//   * kernels cut out of the .cu files by oracle/ref_extract.py (each fragment carries a #line into the reference):
//     testbed_nerf.cu: advance_pos_nerf, generate_next_nerf_network_inputs, composite_kernel_nerf, shade_kernel_nerf,
//     compact_kernel_nerf, init_rays_with_payload_kernel_nerf, grid_to_bitfield, bitfield_max_pool, ema_grid_samples_nerf, ...;
//     cage_deformation.cu: interpolate_tet(_pos), compute_residual_poisson_kernel, compute_poisson_residual_density_kernel;
//     selection_utils.cu: get_cell_pos, get_cell_at_pos; tet_mesh.cu: the marking loops of TetMesh::build_tet_grid.
// What is NOT the reference's code: <Eigen/Dense>, tiny-cuda-nn and the CUDA keywords come from oracle/ref_stubs (the submodules are
// empty; see the headers there for the evaluation-order model), the network is a caller-supplied callback (tcnn: unpinned), and the
// host loops below that call the kernels restate Testbed::render_nerf / NerfTracer::init_rays_from_camera / NerfTracer::trace
// (testbed_nerf.cu:3066, :2683, :2772) and CageDeformation::map_rays / compute_poisson_full_residuals (cage_deformation.cu:547, :682)
// -- kernel order, arguments and buffer roles, cited line by line.
#include <tiny-cuda-nn/common.h>
thread_local uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

#include <neural-graphics-primitives/common.h>
#include <neural-graphics-primitives/common_device.cuh>
#include <neural-graphics-primitives/common_nerf.h>
#include <neural-graphics-primitives/envmap.cuh>
#include <neural-graphics-primitives/editing/tools/selection_utils.h>

#include <src/common_nerf.cu> // -I/root/reference
#include <src/editing/tools/sh_utils.cu> // project_sh9

#include <memory>
#include <tuple>

// svd3.h calls rsqrt(), which nvcc's host math headers provide (1 / sqrt in host code); g++ has no such function
inline float rsqrt(float x) { return 1.0f / std::sqrt(x); }
#include <neural-graphics-primitives/editing/tools/svd3.h>
#include <neural-graphics-primitives/editing/tools/mvc.h>

#include "../include/nrs.h"

using namespace Eigen;
using namespace tcnn;

NGP_NAMESPACE_BEGIN
#include "accumulate_kernel.inc"
#include "march_constants.inc"
#include "network_to_rgb_derivative.inc"
#include "network_to_density_derivative.inc"
#include "splat_already_activated.inc"
#include "ema_grid_samples_nerf.inc"
#include "grid_to_bitfield.inc"
#include "bitfield_max_pool.inc"
#include "advance_pos_nerf.inc"
#include "generate_next_nerf_network_inputs.inc"
#include "composite_kernel_nerf.inc"
#include "shade_kernel_nerf.inc"
#include "compact_kernel_nerf.inc"
#include "init_rays_with_payload_kernel_nerf.inc"
#include "activate_network_density.inc"
#include "generate_grid_samples_nerf_uniform.inc"
#include "generate_grid_samples_nerf_uniform_dir.inc"
#include "grid_samples_half_to_float.inc"
#include "compute_nerf_density.inc"
#include "generate_nerf_network_inputs_at_current_position.inc"
template <typename T>
#include "clear_empty_space.inc"
#include "interpolate_tet_pos.inc"
#include "interpolate_tet.inc"
#include "compute_poisson_residual_density_kernel.inc"
#include "compute_residual_poisson_kernel.inc"
#include "get_cell_pos.inc"
#include "get_cell_at_pos.inc"
#include "get_upper_cell_idx.inc"
#include "shoot_selection_rays_kernel.inc"
#include "composite_shot_rays.inc"
#include "activate_network_output.inc"
#include "filter_empty.inc"
#include "corner_offsets.inc"
#include "affine_bounding_box_struct.inc"
}; // closes struct AffineBoundingBox: the fragment stops before its nlohmann::json members (affine_bounding_box.cuh:105-143), which are not compiled
#include "warp_direction_ad.inc"
#include "unwarp_direction_ad.inc"
#include "translate_in_box_pos.inc"
#include "translate_in_box.inc"
NGP_NAMESPACE_END

using namespace ngp;

namespace {

// tcnn::linear_kernel: one CUDA thread per element.  Here: one call per element with blockDim = 1, blockIdx = element.
template <typename K, typename... Args> void launch_linear(bool parallel, uint32_t n, K kernel, const Args&... args) {
#pragma omp parallel for schedule(dynamic, 256) if (parallel)
	for (int64_t i = 0; i < (int64_t)n; ++i) {
		blockDim.x = 1; blockDim.y = 1; blockDim.z = 1;
		threadIdx.x = threadIdx.y = threadIdx.z = 0;
		blockIdx.x = (uint32_t)i; blockIdx.y = blockIdx.z = 0;
		kernel(n, args...);
	}
}

template <typename K, typename... Args> void launch_3d(const Vector3i& res, K kernel, const Args&... args) {
	blockDim.x = blockDim.y = blockDim.z = 1;
	threadIdx.x = threadIdx.y = threadIdx.z = 0;
	for (int z = 0; z < res.z(); ++z)
		for (int y = 0; y < res.y(); ++y)
			for (int x = 0; x < res.x(); ++x) {
				blockIdx.x = (uint32_t)x; blockIdx.y = (uint32_t)y; blockIdx.z = (uint32_t)z;
				kernel(args...);
			}
	blockIdx.y = blockIdx.z = 0;
}
inline Vector3f v3(const float* p) { return Vector3f(p[0], p[1], p[2]); }
inline Matrix<float, 3, 4> m34(const float* p) { Matrix<float, 3, 4> m; memcpy(m.data(), p, 48); return m; }
inline BoundingBox box(const float* mn, const float* mx) { return BoundingBox(v3(mn), v3(mx)); }

// One CageDeformation as the tracer sees it: the GPU members of TetMesh (tet_mesh.h:80-94) + the bounding boxes its
// post_update_vertices derives (tet_mesh.cu:12-20: bbox = enlarge over the vertices, warped_bbox = bbox.warp_box(scene aabb)).
struct RefEdit {
	BoundingBox scene_aabb, bbox, warped_bbox, original_bbox, original_warped_bbox;
	const nrs_tet_mesh* mesh;
	std::vector<Matrix3f> rotations;
	std::vector<SH9RGB> boundary_shs;
};

RefEdit make_edit(const nrs_model_desc* d, const nrs_tet_mesh* mesh) {
	RefEdit e;
	e.mesh = mesh;
	e.scene_aabb = box(d->aabb_min, d->aabb_max);
	e.bbox = BoundingBox();
	e.original_bbox = BoundingBox();
	for (uint32_t i = 0; i < mesh->n_vertices; ++i) {
		e.bbox.enlarge(v3(mesh->h_vertices + 3 * i));
		e.original_bbox.enlarge(v3(mesh->h_original_vertices + 3 * i));
	}
	e.warped_bbox = e.bbox; e.warped_bbox.warp_box(e.scene_aabb);
	e.original_warped_bbox = e.original_bbox; e.original_warped_bbox.warp_box(e.scene_aabb);
	if (mesh->h_local_rotations) {
		e.rotations.resize(mesh->n_tets);
		memcpy((void*)e.rotations.data(), mesh->h_local_rotations, sizeof(float) * 9 * mesh->n_tets);
	}
	if (mesh->apply_poisson && mesh->h_boundary_shs) {
		e.boundary_shs.resize(mesh->n_vertices);
		memcpy((void*)e.boundary_shs.data(), mesh->h_boundary_shs, sizeof(float) * 27 * mesh->n_vertices);
	}
	return e;
}

// CageDeformation::map_rays, cage_deformation.cu:547-572
void edit_map_rays(const RefEdit& e, PitchedPtr<NerfCoordinate> coords, bool* empty_mask, uint32_t n) {
	const nrs_tet_mesh& m = *e.mesh;
	if (m.n_tets == 0) return;
	launch_linear(true, n, interpolate_tet, coords, empty_mask, (bool)m.copy, e.scene_aabb, e.warped_bbox, e.original_warped_bbox, m.h_lut_idx, m.h_lut_offsets,
	              m.h_tets, (const Vector3f*)m.h_vertices, (const Vector3f*)m.h_original_vertices, e.rotations.empty() ? (const Matrix3f*)nullptr : e.rotations.data(),
	              m.h_original_bitfield);
}

// CageDeformation::map_positions, cage_deformation.cu:624-645
void edit_map_positions(const RefEdit& e, PitchedPtr<NerfPosition> pos, bool* empty_mask, uint32_t n) {
	const nrs_tet_mesh& m = *e.mesh;
	if (m.n_tets == 0) return;
	launch_linear(true, n, interpolate_tet_pos, pos, empty_mask, e.scene_aabb, e.warped_bbox, e.original_warped_bbox, m.h_lut_idx, m.h_lut_offsets, m.h_tets,
	              (const Vector3f*)m.h_vertices, (const Vector3f*)m.h_original_vertices, m.h_original_bitfield);
}

// CageDeformation::compute_poisson_full_residuals, cage_deformation.cu:682-722
void edit_poisson_full_residuals(const RefEdit& e, uint32_t n_alive, NerfPayload* payloads, PitchedPtr<NerfCoordinate> input, SH9RGB* sh, float* out_density, float* res_density) {
	const nrs_tet_mesh& m = *e.mesh;
	if (!m.apply_poisson || m.n_tets == 0) return;
	launch_linear(true, n_alive, compute_residual_poisson_kernel, payloads, input, sh, out_density, res_density, e.scene_aabb, e.original_bbox, e.bbox,
	              (const uint32_t*)nullptr, m.h_lut_idx, (const uint32_t*)nullptr, m.h_lut_offsets, m.h_tets, (const Vector3f*)m.h_original_vertices, (const Vector3f*)m.h_vertices,
	              (const SH9RGB*)e.boundary_shs.data(), m.h_boundary_outside_density, m.h_boundary_residual_density, m.residual_amplitude);
}

// AffineDuplication's box bookkeeping: the members update_destination() touches + update_destination itself (affine_duplication.h:77-90), compiled
// from the reference.  Of the selection box only center / scale / rot_matrix matter (scale_with_vector, rotate and warp_box rebuild u, v, w, min, max).
struct RefAffine {
	BoundingBox m_scene_aabb;
	AffineBoundingBox m_selection_box, m_destination_box, m_warped_destination_box, m_warped_selection_box;
	Vector3f m_translation, m_warped_translation, m_scale;
	Matrix3f m_rotation_matrix;
#include "update_destination.inc"
	RefAffine(const nrs_model_desc* d, const nrs_affine_duplication* op) {
		m_scene_aabb = box(d->aabb_min, d->aabb_max);
		m_selection_box.center = v3(op->selection_center);
		m_selection_box.scale = v3(op->selection_scale);
		memcpy(m_selection_box.rot_matrix.data(), op->selection_rot, 36);
		m_translation = v3(op->translation);
		m_scale = v3(op->scale);
		memcpy(m_rotation_matrix.data(), op->rotation, 36);
		update_destination();
	}
};

} // namespace

extern "C" {

// AffineDuplication::map_rays / map_positions, affine_duplication.cu:121-152
void ref_affine_map_rays(const nrs_model_desc* desc, const nrs_affine_duplication* op, uint32_t n, float* coords7, uint8_t* empty) {
	RefAffine a(desc, op);
	launch_linear(true, n, translate_in_box, PitchedPtr<NerfCoordinate>((NerfCoordinate*)coords7, 1, 0, 0), op->hide_original ? (bool*)empty : (bool*)nullptr, a.m_warped_selection_box,
	              a.m_warped_destination_box, a.m_warped_translation, a.m_scale, a.m_rotation_matrix, (bool)op->correct_dir);
}
void ref_affine_map_positions(const nrs_model_desc* desc, const nrs_affine_duplication* op, uint32_t n, float* pos3, uint8_t* empty) {
	RefAffine a(desc, op);
	launch_linear(true, n, translate_in_box_pos, PitchedPtr<NerfPosition>((NerfPosition*)pos3, 1, 0, 0), op->hide_original ? (bool*)empty : (bool*)nullptr, a.m_warped_selection_box,
	              a.m_warped_destination_box, a.m_warped_translation, a.m_scale, a.m_rotation_matrix);
}

// NerfNetwork<T>::inference_mixed_precision as a callback (the network is tiny-cuda-nn: not part of the reference checkout).
// in7: [n x 7] f32; out: fp16 planes, out[c * ld + s].  Signature = oracle/nrs_oracle.cpp's orc_network_inference.
typedef void (*ref_network_fn)(void* user, uint32_t n, const float* in7, uint16_t* out, uint32_t ld_out, int layout);

// The two tiny-cuda-nn entry points render modes Normals / EncodingVis call (testbed_nerf.cu:2923-2927), as callbacks like the network itself:
// gradient: network.input_gradient(stream, 3, positions, gradients) -> [n x 7] (rows 0..2 the position's, the rest zero);
// visualize: network.visualize_activation(stream, layer, dim, positions, positions) -- it OVERWRITES the input records, as the reference's call does.
typedef void (*ref_gradient_fn)(void* user, uint32_t n, const float* in7, float* grad7);
typedef void (*ref_visualize_fn)(void* user, uint32_t n, float* in7_inout, uint32_t layer, uint32_t dim);
static ref_gradient_fn g_gradient = nullptr;
static ref_visualize_fn g_visualize = nullptr;
void ref_set_introspection(ref_gradient_fn grad, ref_visualize_fn vis) { g_gradient = grad; g_visualize = vis; }

struct ref_render_stats { uint64_t generated, composited; uint32_t n_alive0, n_hit, iterations, pad; };

// One frame: Testbed::render_nerf (testbed_nerf.cu:3066-3201) = init_rays_from_camera (:2683) + trace (:2772) + shade (:3180), Shade / Cost
// modes, whole image.  frame is read and written (alpha-over, :2479); depth is written.  steps_out (optional) = samples composited per pixel;
// probe (optional): for the n_probe listed pixels the network-input records of every generated sample, [n_probe][max_probe][7] + counts.
void ref_render_frame(const nrs_model_desc* desc, const nrs_render_params* p, const uint8_t* bitfield, const nrs_tet_mesh* const* meshes, int n_edits,
                      ref_network_fn net, void* net_user, float* frame, float* depth_buffer, uint32_t* steps_out, ref_render_stats* stats) {
	const Vector2i resolution(p->resolution[0], p->resolution[1]);
	const uint32_t N = (uint32_t)resolution.x() * (uint32_t)resolution.y();
	const Vector2f focal_length(p->focal_length[0], p->focal_length[1]), screen_center(p->screen_center[0], p->screen_center[1]);
	const Matrix<float, 3, 4> camera_matrix0 = m34(p->camera_matrix0), camera_matrix1 = m34(p->camera_matrix1);
	const Vector4f rolling_shutter(p->rolling_shutter[0], p->rolling_shutter[1], p->rolling_shutter[2], p->rolling_shutter[3]);
	const BoundingBox render_aabb = box(p->render_aabb_min, p->render_aabb_max), train_aabb = box(desc->aabb_min, desc->aabb_max);
	const ERenderMode render_mode = (ERenderMode)p->render_mode;
	const ENerfActivation rgb_activation = (ENerfActivation)desc->rgb_activation, density_activation = (ENerfActivation)desc->density_activation;
	const int show_accel = p->show_accel ? (int)p->min_mip : -1; // m_nerf.show_accel; min_mip = (show_accel >= 0) ? show_accel : 0, :2751, :2849
	const bool apply_operators = p->apply_operators && n_edits > 0;

	CameraDistortion camera_distortion{};                       // m_nerf.render_distortion (render_with_camera_distortion), :3078-3100
	camera_distortion.mode = (ECameraDistortionMode)p->distortion_mode;
	for (int k = 0; k < 7; ++k) camera_distortion.params[k] = p->distortion_params[k];

	std::vector<RefEdit> edits;
	bool any_poisson = false;
	for (int i = 0; i < n_edits; ++i) { edits.push_back(make_edit(desc, meshes[i])); any_poisson |= meshes[i]->apply_poisson != 0; }

	// NerfTracer::enlarge, :3004-3064: three ray sets + per-sample arrays
	struct Rays { std::vector<Array4f> rgba; std::vector<float> depth; std::vector<Array3f> normal; std::vector<NerfPayload> payload; };
	Rays rays[2], rays_hit;
	for (Rays* r : {&rays[0], &rays[1], &rays_hit}) { r->rgba.resize(N); r->depth.resize(N); r->normal.resize(N); r->payload.resize(N); memset((void*)r->payload.data(), 0, sizeof(NerfPayload) * N); }
	const uint32_t n_max = next_multiple(N, (uint32_t)batch_size_granularity) + batch_size_granularity * 8;
	std::vector<NerfCoordinate> network_input(n_max, NerfCoordinate(Vector3f::Zero(), Vector3f::Zero(), 0.f)), network_gradient(render_mode == ERenderMode::Normals ? n_max : 1, NerfCoordinate(Vector3f::Zero(), Vector3f::Zero(), 0.f));
	std::vector<network_precision_t> network_output((size_t)n_max * 16), network_output_old((size_t)n_max * 16);
	std::vector<SH9RGB> sh_boundary(n_max);
	std::vector<float> density_out_boundary(n_max), density_residual_boundary(n_max);
	std::vector<uint8_t> empty_mask_store(n_max);

	// ---- init_rays_from_camera, :2709-2755.  plane_z = m_slice_plane_z + m_scale, negated in Slice mode (:3067-3070); no envmap / distortion
	{
		const float plane_z = render_mode == ERenderMode::Slice ? -p->slice_plane_z : p->slice_plane_z, dof = p->dof;
#pragma omp parallel for schedule(dynamic, 16)
		for (int64_t y = 0; y < resolution.y(); ++y)
			for (int64_t x = 0; x < resolution.x(); ++x) {
				blockDim.x = blockDim.y = blockDim.z = 1;
				threadIdx.x = threadIdx.y = threadIdx.z = 0;
				blockIdx.x = (uint32_t)x; blockIdx.y = (uint32_t)y; blockIdx.z = 0;
				init_rays_with_payload_kernel_nerf(p->spp_index, rays[0].payload.data(), resolution, focal_length, camera_matrix0, camera_matrix1, rolling_shutter, screen_center,
				                                   (bool)p->snap_to_pixel_centers, render_aabb, plane_z, dof, camera_distortion, p->d_envmap, Vector2i(p->envmap_resolution[0], p->envmap_resolution[1]),
				                                   (Array4f*)frame, depth_buffer, p->d_distortion_map, Vector2i(p->distortion_resolution[0], p->distortion_resolution[1]), render_mode);
			}
		const uint32_t n_rays_initialized = N; // :2739
		for (uint32_t i = 0; i < N; ++i) { rays[0].rgba[i] = Array4f::Zero(); rays[0].depth[i] = 0.f; rays[0].normal[i] = Array3f::Zero(); } // :2741-2743
		launch_linear(true, n_rays_initialized, advance_pos_nerf, render_aabb, (Vector3f)camera_matrix1.col(2), focal_length, p->spp_index, rays[0].payload.data(), bitfield,
		              (uint32_t)((show_accel >= 0) ? show_accel : 0), p->cone_angle_constant);
	}

	if (render_mode == ERenderMode::Slice) { // :3109, :3126-3162: n_hit = n_rays_initialized, rays_hit = rays_init
		const uint32_t n_hit = N, n_elements = next_multiple(n_hit, (uint32_t)batch_size_granularity);
		std::vector<NerfCoordinate> vis_input(n_elements, NerfCoordinate(Vector3f::Zero(), Vector3f::Zero(), 0.f));
		std::vector<Array4f> vis_rgba(n_elements);
		launch_linear(true, n_hit, generate_nerf_network_inputs_at_current_position, train_aabb, (const NerfPayload*)rays[0].payload.data(), PitchedPtr<NerfCoordinate>(vis_input.data(), 1, 0, 0),
		              Vector3f(Vector3f::Zero()));
		// network.inference(stream, positions, rgbsigma): tcnn runs inference_mixed_precision and hands the fp16 outputs back as floats (4 x n, column-major)
		std::vector<uint16_t> out16((size_t)n_elements * 16);
		net(net_user, n_elements, (const float*)vis_input.data(), out16.data(), n_elements, 0);
		for (uint32_t k = 0; k < n_elements; ++k) {
			network_precision_t h[4];
			for (int c = 0; c < 4; ++c) memcpy((void*)&h[c], &out16[(size_t)c * n_elements + k], 2);
			vis_rgba[k] = Array4f((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
		}
		launch_linear(true, n_hit, compute_nerf_density, vis_rgba.data(), rgb_activation, density_activation);
		launch_linear(false, n_hit, shade_kernel_nerf, vis_rgba.data(), (float*)nullptr, rays[0].normal.data(), rays[0].payload.data(), render_mode, (bool)p->linear_colors, (Array4f*)frame,
		              depth_buffer);
		if (steps_out) memset(steps_out, 0, sizeof(uint32_t) * N);
		if (stats) { stats->generated = N; stats->composited = N; stats->n_alive0 = N; stats->n_hit = N; stats->iterations = 0; stats->pad = 0; }
		return;
	}

	// ---- trace, :2796-3001
	const uint32_t n_rays_initialized = N;
	uint32_t hit_counter = 0, alive_counter = 0;
	uint32_t n_alive = n_rays_initialized;
	uint32_t i = 1, double_buffer_index = 0;
	uint64_t generated = 0, composited = 0;
	uint32_t n_alive0 = 0, iterations = 0;
	const uint32_t march_iter = p->max_march_steps ? p->max_march_steps : MARCH_ITER;
	if (steps_out) memset(steps_out, 0, sizeof(uint32_t) * N);
	while (i < march_iter) {
		Rays& rays_current = rays[(double_buffer_index + 1) % 2];
		Rays& rays_tmp = rays[double_buffer_index % 2];
		++double_buffer_index;
		alive_counter = 0; // :2821
		launch_linear(false, n_alive, compact_kernel_nerf, rays_tmp.rgba.data(), rays_tmp.depth.data(), rays_tmp.normal.data(), rays_tmp.payload.data(), rays_current.rgba.data(),
		              rays_current.depth.data(), rays_current.normal.data(), rays_current.payload.data(), rays_hit.rgba.data(), rays_hit.depth.data(), rays_hit.normal.data(),
		              rays_hit.payload.data(), &alive_counter, &hit_counter);
		n_alive = alive_counter;
		if (iterations == 0) n_alive0 = n_alive;
		if (n_alive == 0) break;
		++iterations;
		const uint32_t n_steps_between_compaction = tcnn::clamp(n_rays_initialized / n_alive, (uint32_t)MIN_STEPS_INBETWEEN_COMPACTION, (uint32_t)MAX_STEPS_INBETWEEN_COMPACTION); // :2835
		PitchedPtr<NerfCoordinate> input_data(network_input.data(), 1, 0, 0), gradient_data(network_gradient.data(), 1, 0, 0); // n_extra_dims() = 0
		launch_linear(true, n_alive, generate_next_nerf_network_inputs, render_aabb, train_aabb, focal_length, (Vector3f)camera_matrix1.col(2), rays_current.payload.data(), input_data,
		              n_steps_between_compaction, bitfield, (uint32_t)((show_accel >= 0) ? show_accel : 0), p->cone_angle_constant, Vector3f(Vector3f::Zero()));
		const uint32_t n_elements = next_multiple(n_alive * n_steps_between_compaction, (uint32_t)batch_size_granularity); // :2858
		for (uint32_t k = 0; k < n_alive; ++k) generated += rays_current.payload[k].n_steps;

		// :2863-2883
		memset((void*)sh_boundary.data(), 0, sizeof(SH9RGB) * n_elements);
		memset(density_out_boundary.data(), 0, sizeof(float) * n_elements);
		memset(density_residual_boundary.data(), 0, sizeof(float) * n_elements);
		if (apply_operators)
			for (int k = (int)edits.size() - 1; k >= 0; --k)
				edit_poisson_full_residuals(edits[k], n_alive, rays_current.payload.data(), input_data, sh_boundary.data(), density_out_boundary.data(), density_residual_boundary.data());

		// first network pass on the un-mapped coordinates, :2890-2892.  Its output is read only where density_out_boundary > 1e-9
		// (composite_kernel_nerf :770-773), which stays 0 unless an operator applies the membrane correction: skipped otherwise.
		if (apply_operators && any_poisson) net(net_user, n_elements, (const float*)network_input.data(), (uint16_t*)network_output_old.data(), n_elements, 0);

		bool* empty_mask_ptr = nullptr; // :2885-2904
		if (apply_operators) {
			memset(empty_mask_store.data(), 0, n_elements);
			for (int k = (int)edits.size() - 1; k >= 0; --k) edit_map_rays(edits[k], input_data, (bool*)empty_mask_store.data(), n_elements);
			empty_mask_ptr = (bool*)empty_mask_store.data();
		}

		net(net_user, n_elements, (const float*)network_input.data(), (uint16_t*)network_output.data(), n_elements, 0); // :2913
		// clear_empty_space (:2919) has an empty body (:2759-2770)
		if (render_mode == ERenderMode::Normals && g_gradient) g_gradient(net_user, n_elements, (const float*)network_input.data(), (float*)network_gradient.data()); // :2924
		else if (render_mode == ERenderMode::EncodingVis && g_visualize) g_visualize(net_user, n_elements, (float*)network_input.data(), p->visualized_layer, p->visualized_dimension); // :2926

		std::vector<uint16_t> n_steps_before(n_alive); // payload.n_steps as generate_next_nerf_network_inputs left it = samples this ray holds
		for (uint32_t k = 0; k < n_alive; ++k) n_steps_before[k] = rays_current.payload[k].n_steps;
		launch_linear(true, n_alive, composite_kernel_nerf, n_elements, i, train_aabb, p->glow_y_cutoff, (int)p->glow_mode, 0u, (const TrainingXForm*)nullptr, camera_matrix1, focal_length,
		              p->depth_scale, rays_current.rgba.data(), rays_current.depth.data(), rays_current.normal.data(), rays_current.payload.data(), input_data, gradient_data,
		              (const network_precision_t*)network_output_old.data(), (const network_precision_t*)network_output.data(), (const SH9RGB*)sh_boundary.data(),
		              (const float*)density_out_boundary.data(), (const float*)density_residual_boundary.data(), 16u, n_steps_between_compaction, render_mode, bitfield, rgb_activation,
		              density_activation, show_accel, p->min_transmittance, (const bool*)empty_mask_ptr, (bool)p->poisson_target);
		for (uint32_t k = 0; k < n_alive; ++k) {
			const NerfPayload& pl = rays_current.payload[k];
			uint32_t c;
			if (pl.alive) c = pl.n_steps; // all n_steps_between_compaction samples composited
			else { // payload.n_steps = j + current_step (:959); j < the samples held: the loop broke at j after compositing it (:951-954)
				const uint32_t j = (uint32_t)pl.n_steps - i;
				c = j < n_steps_before[k] ? j + 1 : j;
			}
			composited += c;
			if (steps_out) steps_out[pl.idx] += c;
		}
		i += n_steps_between_compaction; // :2989
	}
	const uint32_t n_hit = hit_counter;

	// ---- shade, :3180-3191
	launch_linear(false, n_hit, shade_kernel_nerf, rays_hit.rgba.data(), rays_hit.depth.data(), rays_hit.normal.data(), rays_hit.payload.data(), render_mode,
	              (bool)p->linear_colors, (Array4f*)frame, depth_buffer);
	if (stats) { stats->generated = generated; stats->composited = composited; stats->n_alive0 = n_alive0; stats->n_hit = n_hit; stats->iterations = iterations; stats->pad = 0; }
}

// Per listed pixel: init ray -> jitter + first hit -> sample after sample (generate_next_nerf_network_inputs with n_steps = 1), ignoring compositing.
// Outputs per sample the network-input record the reference writes (warped position, warped dt, warped direction) and payload.t after it (= t + dt).
void ref_trace_coords(const nrs_model_desc* desc, const nrs_render_params* p, const uint8_t* bitfield, uint32_t n_pixels, const uint32_t* pixel_idx, uint32_t max_samples,
                      float* coords_out /*[n][max][7]*/, float* t_after_out /*[n][max]*/, uint32_t* count_out, float* origin_dir_t0_out /*[n][7] or NULL*/) {
	const Vector2i resolution(p->resolution[0], p->resolution[1]);
	const Vector2f focal_length(p->focal_length[0], p->focal_length[1]), screen_center(p->screen_center[0], p->screen_center[1]);
	const Matrix<float, 3, 4> camera_matrix0 = m34(p->camera_matrix0), camera_matrix1 = m34(p->camera_matrix1);
	const Vector4f rolling_shutter(p->rolling_shutter[0], p->rolling_shutter[1], p->rolling_shutter[2], p->rolling_shutter[3]);
	const BoundingBox render_aabb = box(p->render_aabb_min, p->render_aabb_max), train_aabb = box(desc->aabb_min, desc->aabb_max);
	const uint32_t W = (uint32_t)resolution.x();
	const uint32_t N = W * (uint32_t)resolution.y();
#pragma omp parallel for schedule(dynamic, 64)
	for (int64_t k = 0; k < (int64_t)n_pixels; ++k) {
		const uint32_t idx = pixel_idx[k];
		// the kernels index payloads[pixel]; give them a window whose element `idx` is ours
		NerfPayload mine; memset((void*)&mine, 0, sizeof(mine));
		NerfPayload* base = &mine - idx;
		Array4f fb_dummy = Array4f::Zero(); float depth_dummy = 0.f;
		blockDim.x = blockDim.y = blockDim.z = 1; threadIdx.x = threadIdx.y = threadIdx.z = 0;
		blockIdx.x = idx % W; blockIdx.y = idx / W; blockIdx.z = 0;
		init_rays_with_payload_kernel_nerf(p->spp_index, base, resolution, focal_length, camera_matrix0, camera_matrix1, rolling_shutter, screen_center, (bool)p->snap_to_pixel_centers, render_aabb,
		                                   1.0f, 0.0f, CameraDistortion{}, (const float*)nullptr, Vector2i(0, 0), &fb_dummy - idx, &depth_dummy - idx, (const float*)nullptr, Vector2i(0, 0),
		                                   ERenderMode::Shade);
		blockIdx.x = idx; blockIdx.y = 0;
		advance_pos_nerf(N, render_aabb, (Vector3f)camera_matrix1.col(2), focal_length, p->spp_index, base, bitfield, p->min_mip, p->cone_angle_constant);
		if (origin_dir_t0_out) {
			float* o = origin_dir_t0_out + 7 * (size_t)k;
			for (int c = 0; c < 3; ++c) { o[c] = mine.origin[c]; o[3 + c] = mine.dir[c]; }
			o[6] = mine.t;
		}
		uint32_t cnt = 0;
		NerfCoordinate rec(Vector3f::Zero(), Vector3f::Zero(), 0.f);
		while (mine.alive && cnt < max_samples) {
			blockIdx.x = 0;
			PitchedPtr<NerfCoordinate> out(&rec, 1, 0, 0);
			generate_next_nerf_network_inputs(1, render_aabb, train_aabb, focal_length, (Vector3f)camera_matrix1.col(2), &mine, out, 1, bitfield, p->min_mip, p->cone_angle_constant,
			                                  Vector3f(Vector3f::Zero()));
			if (mine.n_steps == 0) break; // left the render box (:675-677)
			memcpy(coords_out + ((size_t)k * max_samples + cnt) * 7, &rec, 28);
			t_after_out[(size_t)k * max_samples + cnt] = mine.t;
			++cnt;
		}
		count_out[k] = cnt;
	}
}

// ---- EditOperator::map_rays / map_positions / compute_poisson_full_residuals on caller batches ----------------------------------------
void ref_edit_map_rays(const nrs_model_desc* desc, const nrs_tet_mesh* mesh, uint32_t n, float* coords7, uint8_t* empty) {
	RefEdit e = make_edit(desc, mesh);
	edit_map_rays(e, PitchedPtr<NerfCoordinate>((NerfCoordinate*)coords7, 1, 0, 0), (bool*)empty, n);
}
void ref_edit_map_positions(const nrs_model_desc* desc, const nrs_tet_mesh* mesh, uint32_t n, float* pos3, uint8_t* empty) {
	RefEdit e = make_edit(desc, mesh);
	edit_map_positions(e, PitchedPtr<NerfPosition>((NerfPosition*)pos3, 1, 0, 0), (bool*)empty, n);
}
// one sample per "ray" (n_elements = n, n_steps = 1)
void ref_edit_poisson_residuals(const nrs_model_desc* desc, const nrs_tet_mesh* mesh, uint32_t n, const float* coords7, float* sh27, float* out_density, float* res_density) {
	RefEdit e = make_edit(desc, mesh);
	std::vector<NerfPayload> payloads(n);
	memset((void*)payloads.data(), 0, sizeof(NerfPayload) * n);
	for (uint32_t i = 0; i < n; ++i) payloads[i].n_steps = 1;
	edit_poisson_full_residuals(e, n, payloads.data(), PitchedPtr<NerfCoordinate>((NerfCoordinate*)coords7, 1, 0, 0), (SH9RGB*)sh27, out_density, res_density);
}

// ---- TetMesh::build_tet_grid, tet_mesh.cu:368-673: marking loops compiled from the reference, the merge around them restated --------------
// (float_t = float, point_t = Eigen::Vector3f, growing_selection.h:89-90).  The reference runs the loop on min(n_tets, 32) std::async threads over
// contiguous tet ranges and merges the per-thread lists in thread order, i.e. in ascending tet order: one range [0, n_tets) gives the same lists.
// Deviation kept out of the pin: the reference counts tets per cell in uint8_t (tet_counts, tet_allocated; :383, :501), which wraps beyond 255
// tets per cell; counts here are 32-bit (the tests' meshes stay below 256 per cell where that matters, and report max_per_cell).
void ref_build_tet_grid(const float* vertices_f, const float* original_vertices_f, uint32_t n_vertices, const uint32_t* tets_in, uint32_t n_tets, uint32_t* lut_offsets /*[5*128^3+1]*/,
                        uint32_t* lut_idx /* capacity */, uint32_t lut_capacity, uint32_t* n_idx_out, uint8_t* original_bitfield_out /*[5*128^3/8]*/, uint32_t* max_per_cell_out) {
	typedef float float_t;
	typedef Vector3f point_t;
	std::vector<point_t> vertices(n_vertices), original_vertices(n_vertices);
	for (uint32_t i = 0; i < n_vertices; ++i) { vertices[i] = v3(vertices_f + 3 * i); original_vertices[i] = v3(original_vertices_f + 3 * i); }
	std::vector<uint32_t> tets(tets_in, tets_in + 4 * (size_t)n_tets);
	const uint32_t n_elements = NERF_GRIDSIZE() * NERF_GRIDSIZE() * NERF_GRIDSIZE() * NERF_CASCADES();
	std::vector<std::vector<std::tuple<int, int, int, int>>> up_ids(1);
	std::vector<int> tet_sums(1, 0);
	uint32_t intersection_marked = 0;
	std::vector<uint8_t> original_bitfield(n_elements / 8, 0);
	{
		const int tt = 0, beginn = 0, endingg = (int)n_tets;
#include "build_tet_grid_mark_deformed.inc"
	}
	std::vector<uint32_t> tet_counts(n_elements, 0);
	for (auto& q : up_ids[0]) tet_counts[std::get<0>(q)]++; // :476-481
	uint32_t counter = 0, max_tet_lookup = 0;
	for (uint32_t i = 0; i < n_elements; i++) { // :488-495
		lut_offsets[i] = counter;
		counter += tet_counts[i];
		if (tet_counts[i] > max_tet_lookup) max_tet_lookup = tet_counts[i];
	}
	lut_offsets[n_elements] = counter;
	*n_idx_out = counter;
	*max_per_cell_out = max_tet_lookup;
	if (counter <= lut_capacity) {
		std::vector<uint32_t> tet_allocated(n_elements, 0);
		for (auto& q : up_ids[0]) { // :502-513
			const int cell_idx = std::get<0>(q), tet = std::get<1>(q);
			lut_idx[lut_offsets[cell_idx] + tet_allocated[cell_idx]] = tet;
			tet_allocated[cell_idx]++;
		}
	}
	{
		const int beginn = 0, endingg = (int)n_tets;
#include "build_tet_grid_mark_canonical.inc"
	}
	memcpy(original_bitfield_out, original_bitfield.data(), n_elements / 8);
	(void)intersection_marked;
}

// ---- TetMesh::update_local_rotations, tet_mesh.cu:37-74 (loop compiled from the reference; svd_eigen from editing/tools/svd3.h) ---------------
void ref_local_rotations(const float* def, const float* org, uint32_t n_vertices, const uint32_t* tets_in, uint32_t n_tets_in, float* out9_colmajor) {
	std::vector<Vector3f> vertices(n_vertices), original_vertices(n_vertices);
	for (uint32_t i = 0; i < n_vertices; ++i) { vertices[i] = v3(def + 3 * i); original_vertices[i] = v3(org + 3 * i); }
	std::vector<uint32_t> tets(tets_in, tets_in + 4 * (size_t)n_tets_in);
	const uint32_t n_tets = n_tets_in;
	std::vector<Eigen::Matrix3f> local_rotations_host;
	local_rotations_host.reserve(n_tets);
#include "update_local_rotations_loop.inc"
	memcpy(out9_colmajor, (const void*)local_rotations_host.data(), sizeof(float) * 9 * (size_t)n_tets);
}

// ---- Cage::compute_mvc (cage.cu:6-22, gamma = 1) around MVC3D::computeCoordinatesCustomCode (editing/tools/mvc.h:125-188), point_t = Eigen::Vector3f ---
void ref_mvc_compute(const float* cage_vertices, uint32_t n_cv, const uint32_t* cage_triangles, uint32_t n_tris, const float* points, uint32_t n_points, float* weights_out,
                     uint8_t* labels_out) {
	std::vector<Vector3f> cv(n_cv), normals;
	for (uint32_t i = 0; i < n_cv; ++i) cv[i] = v3(cage_vertices + 3 * i);
	std::vector<uint32_t> tris(cage_triangles, cage_triangles + 3 * (size_t)n_tris);
	std::vector<float> w, ww;
	for (uint32_t q = 0; q < n_points; ++q) {
		const bool success = MVC3D::computeCoordinatesCustomCode<uint32_t, float, Vector3f>(v3(points + 3 * (size_t)q), tris, cv, normals, w, ww);
		for (uint32_t v = 0; v < n_cv; ++v) weights_out[(size_t)q * n_cv + v] = w[v];
		labels_out[q] = success ? 0 : 1;
	}
}
// ---- Cage::interpolate_with_mvc, cage.cu:38-49 (loop compiled from the reference) ------------------------------------------------------------
void ref_mvc_apply(const float* weights_in, const float* cage_vertices, uint32_t n_cv, uint32_t n_pts, float* out3) {
	typedef Vector3f point_t;
	std::vector<std::vector<float>> weights(n_pts, std::vector<float>(n_cv));
	for (uint32_t i = 0; i < n_pts; ++i) for (uint32_t v = 0; v < n_cv; ++v) weights[i][v] = weights_in[(size_t)i * n_cv + v];
	std::vector<point_t> vertices(n_cv), points(n_pts, point_t::Zero());
	for (uint32_t v = 0; v < n_cv; ++v) vertices[v] = v3(cage_vertices + 3 * v);
	const uint32_t n_points = n_pts, n_vertices = n_cv;
#include "interpolate_with_mvc_loop.inc"
	for (uint32_t i = 0; i < n_pts; ++i) for (int c = 0; c < 3; ++c) out3[3 * (size_t)i + c] = points[i][c];
}

// ---- GrowingSelection::interpolate_poisson_boundary, growing_selection.cu:2350-2395: loop compiled from the reference --------------------------------
// per-cage-vertex inside / outside densities and SH9RGB colours (compute_poisson_boundary) -> per-tet-vertex boundary_shs / outside / residual density
void ref_poisson_interpolate(const float* gamma /*[V_tet x V_cage]*/, uint32_t n_tet_vertices_in, uint32_t n_cage_vertices_in, const float* inside_density, const float* outside_density,
                             const float* inside_shs27, const float* outside_shs27, float* boundary_shs27_out, float* outside_density_out, float* residual_density_out) {
	struct { std::vector<float> outside_density, inside_density; std::vector<SH9RGB> outside_shs, inside_shs; } proxy_cage;
	struct TetMeshGamma { std::vector<std::vector<float>> gamma_coordinates; } tet_mesh_storage;
	TetMeshGamma* tet_interpolation_mesh = &tet_mesh_storage;
	const uint32_t n_tet_vertices = n_tet_vertices_in, n_cage_vertices = n_cage_vertices_in;
	proxy_cage.outside_density.assign(outside_density, outside_density + n_cage_vertices);
	proxy_cage.inside_density.assign(inside_density, inside_density + n_cage_vertices);
	proxy_cage.outside_shs.resize(n_cage_vertices); proxy_cage.inside_shs.resize(n_cage_vertices);
	memcpy((void*)proxy_cage.outside_shs.data(), outside_shs27, 108 * (size_t)n_cage_vertices);
	memcpy((void*)proxy_cage.inside_shs.data(), inside_shs27, 108 * (size_t)n_cage_vertices);
	tet_mesh_storage.gamma_coordinates.resize(n_tet_vertices);
	for (uint32_t i = 0; i < n_tet_vertices; ++i) tet_mesh_storage.gamma_coordinates[i].assign(gamma + (size_t)i * n_cage_vertices, gamma + (size_t)(i + 1) * n_cage_vertices);
	std::vector<SH9RGB> boundary_shs_host(n_tet_vertices, SH9RGB::Zero());
	std::vector<float> boundary_residual_density_host(n_tet_vertices, 0.f), boundary_outside_density_host(n_tet_vertices, 0.f), boundary_inside_density_host(n_tet_vertices, 0.f);
#include "interpolate_poisson_boundary_loop.inc"
	memcpy(boundary_shs27_out, (const void*)boundary_shs_host.data(), 108 * (size_t)n_tet_vertices);
	memcpy(outside_density_out, boundary_outside_density_host.data(), 4 * (size_t)n_tet_vertices);
	memcpy(residual_density_out, boundary_residual_density_host.data(), 4 * (size_t)n_tet_vertices);
}

// ---- Testbed::update_density_grid_nerf_operator, testbed_nerf.cu:3533-3640: the host sequence restated around the reference's kernels ----------
// (generate_grid_samples_nerf_nonuniform x 2, map_positions per operator, density() = the callback, clear_empty_space -- whose body is commented
// out in the reference --, activate_network_density, compute_poisson_residual_density per operator, the max-splat and the decayed maximum).
// Deviation kept out of the pin, as in the oracle: clear_empty_space / compute_poisson_residual_density are launched over the n_samples samples
// that exist, not over n_elements = 5 * 128^3 threads (:3606, :3622 read past the arrays when max_cascade < 4).
typedef void (*ref_density_fn)(void* user, uint32_t n, const float* in, uint32_t ld_in, uint16_t* out, uint32_t ld_out, int layout);
void ref_update_density_grid(const nrs_model_desc* desc, const nrs_tet_mesh* const* meshes, int n_edits, nrs_grid_update* u, float* density_grid /*[5*128^3] in/out*/,
                             ref_density_fn density, void* user) {
	const uint32_t n_elements = NERF_GRIDSIZE() * NERF_GRIDSIZE() * NERF_GRIDSIZE() * NERF_CASCADES();
	const uint32_t n_uniform = u->n_uniform_samples, n_nonuniform = u->n_nonuniform_samples, n_samples = n_uniform + n_nonuniform;
	const uint32_t padded_output_width = 16; // NerfNetwork::padded_density_output_width()
	std::vector<RefEdit> edits;
	for (int i = 0; i < n_edits; ++i) edits.push_back(make_edit(desc, meshes[i]));
	const BoundingBox aabb = box(desc->aabb_min, desc->aabb_max);
	std::vector<NerfPosition> positions(n_samples, NerfPosition(Vector3f::Zero(), 0.f));
	std::vector<uint32_t> indices(n_samples);
	std::vector<float> density_grid_tmp(n_elements, 0.f);
	std::vector<network_precision_t> mlp_out((size_t)n_samples * padded_output_width);
	if (u->reset_grid) memset(density_grid, 0, sizeof(float) * n_elements);
	default_rng_t rng;
	rng.state = u->rng_state; rng.inc = u->rng_inc;
	launch_linear(true, n_uniform, generate_grid_samples_nerf_nonuniform, rng, (uint32_t)u->ema_step, aabb, (const float*)density_grid, positions.data(), indices.data(),
	              u->max_cascade + 1u, -0.01f);
	rng.advance();
	launch_linear(true, n_nonuniform, generate_grid_samples_nerf_nonuniform, rng, (uint32_t)u->ema_step, aabb, (const float*)density_grid, positions.data() + n_uniform,
	              indices.data() + n_uniform, u->max_cascade + 1u, NERF_MIN_OPTICAL_THICKNESS());
	rng.advance();
	std::unique_ptr<bool[]> empty_mask;
	if (n_edits > 0) {
		empty_mask.reset(new bool[n_elements]());
		for (int i = n_edits - 1; i >= 0; --i) edit_map_positions(edits[i], PitchedPtr<NerfPosition>(positions.data(), 1, 0, 0), empty_mask.get(), n_samples);
	}
	// m_nerf_network->density(stream, positions 3 x n, density_matrix (row-major padded_output_width x n), false)
	density(user, n_samples, (const float*)positions.data(), (uint32_t)(sizeof(NerfPosition) / sizeof(float)), (uint16_t*)mlp_out.data(), n_samples, NRS_PLANES);
	if (n_edits > 0) launch_linear(true, n_samples, clear_empty_space<network_precision_t>, (const bool*)empty_mask.get(), mlp_out.data());
	launch_linear(true, n_samples, activate_network_density, mlp_out.data(), (ENerfActivation)desc->density_activation);
	for (int i = n_edits - 1; i >= 0; --i) { // CageDeformation::compute_poisson_residual_density, cage_deformation.cu:647-672
		const nrs_tet_mesh& m = *edits[i].mesh;
		if (!m.apply_poisson || m.n_tets == 0) continue;
		launch_linear(true, n_samples, compute_poisson_residual_density_kernel, PitchedPtr<NerfPosition>(positions.data(), 1, 0, 0), mlp_out.data(), edits[i].scene_aabb,
		              Vector3f(Vector3f::Zero()), Vector3f(Vector3f::Zero()), edits[i].bbox, m.h_lut_idx, m.h_lut_offsets, m.h_tets, (const Vector3f*)m.h_vertices,
		              m.h_boundary_residual_density);
	}
	launch_linear(false, n_samples, splat_grid_samples_nerf_max_nearest_neighbor_already_activated, (const uint32_t*)indices.data(), (const network_precision_t*)mlp_out.data(),
	              density_grid_tmp.data());
	launch_linear(true, n_elements, ema_grid_samples_nerf, u->decay, (uint32_t)u->ema_step, density_grid, (const float*)density_grid_tmp.data());
	u->ema_step += 1;
	u->rng_state = rng.state;
}

// ---- GrowingSelection::project_selection_pixels, growing_selection.cu:1832-1960: shoot_selection_rays_kernel -> density() -> composite_shot_rays ----
// Per-pixel outputs in the caller's order (the reference compacts the rays with atomics; ray_indices maps back).  The kernels run one thread after
// the other here, so the atomics hand out the slots in pixel order.  A ray that never reaches the transmittance threshold gets the reference's
// sentinel (aabb.min - 1, :1826); the slots of rays without samples and of rays that reach the threshold only behind their last sample are never
// written by the reference (uninitialised workspace): they start from the same sentinel here.  The bookkeeping that follows (:1962-2020) is host code
// on these arrays and is restated, not compiled (std::set order).
void ref_project_selection_pixels(const nrs_model_desc* desc, const nrs_render_params* p, const uint8_t* bitfield, const int32_t* pixels_xy, uint32_t n_rays,
                                  float transmittance_threshold, ref_density_fn density, void* user, float* positions_out /*[n][3]*/, uint32_t* cells_out, uint8_t* found_out) {
	const BoundingBox aabb = box(desc->aabb_min, desc->aabb_max);
	const uint32_t padded_density_output_width = 16, floats_per_coord = sizeof(NerfCoordinate) / sizeof(float), max_samples = n_rays * NERF_STEPS();
	std::vector<uint32_t> ray_indices(n_rays), numsteps(2 * (size_t)n_rays), grid_indices(n_rays, 0u);
	std::vector<Ray> rays(n_rays);
	std::vector<float> coords((size_t)max_samples * floats_per_coord, 0.f);
	std::vector<Vector2i> pixels(n_rays);
	for (uint32_t i = 0; i < n_rays; ++i) pixels[i] = Vector2i(pixels_xy[2 * i], pixels_xy[2 * i + 1]);
	const Vector3f sentinel = aabb.min + Vector3f(-1.f, -1.f, -1.f);
	std::vector<Vector3f> coords_projected(n_rays, sentinel);
	uint32_t ray_counter = 0, numsteps_counter = 0;
	launch_linear(false, n_rays, shoot_selection_rays_kernel, pixels.data(), Vector2i(p->resolution[0], p->resolution[1]), Vector2f(p->focal_length[0], p->focal_length[1]),
	              m34(p->camera_matrix1), Vector2f(p->screen_center[0], p->screen_center[1]), aabb, max_samples, &ray_counter, &numsteps_counter, ray_indices.data(), rays.data(),
	              numsteps.data(), PitchedPtr<NerfCoordinate>((NerfCoordinate*)coords.data(), 1, 0, 0), bitfield, p->cone_angle_constant, Vector3f(Vector3f(1.f, 0.f, 0.f)));
	for (uint32_t i = 0; i < n_rays; ++i) {
		found_out[i] = 0; cells_out[i] = 0;
		for (int k = 0; k < 3; ++k) positions_out[3 * i + k] = sentinel[k];
	}
	if (numsteps_counter == 0) return;
	std::vector<network_precision_t> mlp_out((size_t)numsteps_counter * padded_density_output_width);
	density(user, numsteps_counter, coords.data(), floats_per_coord, (uint16_t*)mlp_out.data(), padded_density_output_width, NRS_INTERLEAVED); // column-major 16 x n
	launch_linear(true, n_rays, composite_shot_rays, aabb, (const uint32_t*)&ray_counter, (int)padded_density_output_width, (const network_precision_t*)mlp_out.data(), &numsteps_counter,
	              (const Ray*)rays.data(), numsteps.data(), PitchedPtr<const NerfCoordinate>((const NerfCoordinate*)coords.data(), 1, 0, 0), (ENerfActivation)desc->rgb_activation,
	              (ENerfActivation)desc->density_activation, coords_projected.data(), grid_indices.data(), transmittance_threshold);
	for (uint32_t r = 0; r < ray_counter; ++r) {
		const uint32_t i = ray_indices[r];
		for (int k = 0; k < 3; ++k) positions_out[3 * i + k] = coords_projected[r][k];
		if (aabb.contains(coords_projected[r])) { found_out[i] = 1; cells_out[i] = grid_indices[r]; }
	}
}

// ---- Testbed::get_density_on_grid (testbed_nerf.cu:4538-4586) and get_rgba_on_grid (:4588-4613): generators, network callback, conversion kernels ----
// (the reference launches the generators on a 3-D grid of 16 x 8 x 1 blocks: one call per grid point here; the 2^20-point batching only bounds
// its scratch memory)
void ref_density_on_grid(const nrs_model_desc* desc, const uint32_t* res3d_in, const float* box_min, const float* box_max, const float* density_grid /*nullable*/,
                         ref_density_fn density, void* user, float* out) {
	const Vector3i res3d((int)res3d_in[0], (int)res3d_in[1], (int)res3d_in[2]);
	const uint32_t n_elements = (uint32_t)(res3d.x() * res3d.y() * res3d.z());
	const BoundingBox m_aabb = box(desc->aabb_min, desc->aabb_max), aabb = box(box_min, box_max);
	std::vector<NerfPosition> positions(n_elements, NerfPosition(Vector3f::Zero(), 0.f));
	launch_3d(res3d, generate_grid_samples_nerf_uniform, res3d, 0u, aabb, m_aabb, positions.data());
	std::vector<network_precision_t> mlp_out((size_t)n_elements * 16);
	density(user, n_elements, (const float*)positions.data(), (uint32_t)(sizeof(NerfPosition) / sizeof(float)), (uint16_t*)mlp_out.data(), n_elements, NRS_PLANES); // RM 16 x n
	launch_linear(true, n_elements, grid_samples_half_to_float, m_aabb, out, (const network_precision_t*)mlp_out.data(), (ENerfActivation)desc->density_activation,
	              (const NerfPosition*)positions.data(), density_grid);
}
// m_network->inference (full-precision 4 x n output) = the fp16 operator's rows 0..3 widened to float (exact)
void ref_rgba_on_grid(const nrs_model_desc* desc, const uint32_t* res3d_in, const float* render_min, const float* render_max, const float* ray_dir, ref_network_fn inference,
                      void* user, float* out_rgba) {
	const Vector3i res3d((int)res3d_in[0], (int)res3d_in[1], (int)res3d_in[2]);
	const uint32_t n_elements = (uint32_t)(res3d.x() * res3d.y() * res3d.z());
	const BoundingBox m_aabb = box(desc->aabb_min, desc->aabb_max), render_aabb = box(render_min, render_max);
	std::vector<NerfCoordinate> positions(n_elements, NerfCoordinate(Vector3f::Zero(), Vector3f::Zero(), 0.f));
	launch_3d(res3d, generate_grid_samples_nerf_uniform_dir, res3d, 0u, render_aabb, m_aabb, v3(ray_dir), positions.data());
	std::vector<network_precision_t> net_out((size_t)n_elements * 16);
	inference(user, n_elements, (const float*)positions.data(), (uint16_t*)net_out.data(), 16, NRS_INTERLEAVED);
	Array4f* rgba = (Array4f*)out_rgba;
	for (uint32_t i = 0; i < n_elements; ++i)
		rgba[i] = Array4f((float)net_out[(size_t)i * 16 + 0], (float)net_out[(size_t)i * 16 + 1], (float)net_out[(size_t)i * 16 + 2], (float)net_out[(size_t)i * 16 + 3]);
	launch_linear(true, n_elements, compute_nerf_density, rgba, (ENerfActivation)desc->rgb_activation, (ENerfActivation)desc->density_activation);
}

// ---- GrowingSelection::compute_poisson_boundary, growing_selection.cu:2220-2348: the sampling loop, activate_network_output, filter_empty, the density pick
// and the SH9 fit loop compiled from the reference; the network is the callback.  The reference jitters the directions with std::rand(): the loop runs
// after srand(seed), and the same draws ((float)std::rand() / RAND_MAX, two per sample, in the loop's order) are handed back in jitter_out so that the
// oracle / the product can be given the identical jitter.
void ref_poisson_boundary(const nrs_model_desc* desc, const float* vertices_in, uint32_t n_verts_in, uint32_t sh_width, uint32_t hemisphere_width, unsigned seed, int is_inside_in,
                          const uint8_t* bitfield, ref_network_fn inference, void* user, float* density_out /*[n_verts]*/, float* sh_out /*[n_verts][27]*/, float* jitter_out /*[n][2]*/) {
	typedef Vector3f point_t;
	const bool is_inside = is_inside_in != 0;
	const BoundingBox m_aabb = box(desc->aabb_min, desc->aabb_max);
	std::vector<point_t> vertices(n_verts_in);
	for (uint32_t i = 0; i < n_verts_in; ++i) vertices[i] = v3(vertices_in + 3 * i);
	const uint32_t n_verts = n_verts_in;
	struct { int sh_sampling_width; } m_poisson_editing{(int)sh_width};
	const int m_hemisphere_width = (int)hemisphere_width;
	const uint32_t n_sh_samples = m_poisson_editing.sh_sampling_width * m_poisson_editing.sh_sampling_width; // :2227
	const uint32_t padded_output_width = 16, floats_per_coord = sizeof(NerfCoordinate) / sizeof(float), extra_stride = 0;
	const uint32_t n_samples = n_verts * n_sh_samples;
	std::vector<float> coords_host((size_t)n_samples * floats_per_coord);
	PitchedPtr<NerfCoordinate> coords_host_ptr = PitchedPtr<NerfCoordinate>((NerfCoordinate*)coords_host.data(), 1, 0, extra_stride);
	srand(seed);
	for (uint32_t s = 0; s < 2 * n_samples; ++s) jitter_out[s] = (float)std::rand() / RAND_MAX;
	srand(seed);
#include "poisson_boundary_sampling_loop.inc"
	std::vector<network_precision_t> mlp_out((size_t)n_samples * padded_output_width);
	inference(user, n_samples, coords_host.data(), (uint16_t*)mlp_out.data(), padded_output_width, NRS_INTERLEAVED);
	std::vector<Array3f> rgb_host(n_samples);
	std::vector<float> density_host(n_samples);
	launch_linear(true, n_samples, activate_network_output, (int)padded_output_width, (const network_precision_t*)mlp_out.data(), (ENerfActivation)desc->rgb_activation,
	              (ENerfActivation)desc->density_activation, rgb_host.data(), density_host.data());
	if (is_inside)
		launch_linear(true, n_samples, filter_empty, m_aabb, bitfield, PitchedPtr<NerfCoordinate>((NerfCoordinate*)coords_host.data(), 1, 0, extra_stride), density_host.data());
	std::vector<float> target_density(n_verts);
#include "poisson_boundary_density_loop.inc"
	std::vector<SH9RGB> target_shs(n_verts);
#include "poisson_boundary_fit_loop.inc"
	memcpy(density_out, target_density.data(), sizeof(float) * n_verts);
	memcpy(sh_out, (const void*)target_shs.data(), sizeof(float) * 27 * n_verts);
}

// ---- CudaRenderBuffer::accumulate, render_buffer.cu:540-560: the running mean of the spp frames (accumulate_kernel :217-254) ---------------------------
// sample_count = m_spp before the call (0: the accumulate buffer is cleared first, :545-547); color_space = EColorSpace.
void ref_accumulate(int width, int height, const float* frame, float* accumulate, uint32_t sample_count, int color_space) {
	const Vector2i res(width, height);
	if (sample_count == 0) memset(accumulate, 0, sizeof(float) * 4 * (size_t)width * height);
	blockDim.x = blockDim.y = blockDim.z = 1;
	threadIdx.x = threadIdx.y = threadIdx.z = 0;
	for (int y = 0; y < height; ++y)
		for (int x = 0; x < width; ++x) {
			blockIdx.x = (uint32_t)x; blockIdx.y = (uint32_t)y; blockIdx.z = 0;
			accumulate_kernel(res, (Array4f*)frame, (Array4f*)accumulate, (float)sample_count, (EColorSpace)color_space);
		}
}

// ---- update_density_grid_mean_and_bitfield, testbed_nerf.cu:3642-3657: grid_to_bitfield + bitfield_max_pool (the mean is the caller's) ------
void ref_grid_to_bitfield(const float* grid /*[5*128^3]*/, float mean_density, uint8_t* bitfield /*[5*128^3/8]*/) {
	const uint32_t n_elements = NERF_GRIDSIZE() * NERF_GRIDSIZE() * NERF_GRIDSIZE();
	launch_linear(true, n_elements / 8 * NERF_CASCADES(), grid_to_bitfield, grid, bitfield, (const float*)&mean_density);
	for (uint32_t level = 1; level < NERF_CASCADES(); ++level) // :3652-3656
		launch_linear(false, n_elements / 64, bitfield_max_pool, (const uint8_t*)(bitfield + grid_mip_offset(level - 1) / 8), bitfield + grid_mip_offset(level) / 8);
}

// ---- element-wise probes of the header / common_nerf.cu functions (arrays in, arrays out) -------------------------------------------------
void ref_bary_tet(uint32_t n, const float* abcd12, const float* p3, float* out4) {
	for (uint32_t i = 0; i < n; ++i) {
		const float* q = abcd12 + 12 * (size_t)i;
		Vector4f b = bary_tet(v3(q), v3(q + 3), v3(q + 6), v3(q + 9), v3(p3 + 3 * (size_t)i));
		for (int c = 0; c < 4; ++c) out4[4 * (size_t)i + c] = b[c];
	}
}
void ref_point_in_tet(uint32_t n, const float* abcd12, const float* p3, uint8_t* out) {
	for (uint32_t i = 0; i < n; ++i) {
		const float* q = abcd12 + 12 * (size_t)i;
		Vector3f p = v3(p3 + 3 * (size_t)i);
		out[i] = point_in_tet<float, Vector3f>(v3(q), v3(q + 3), v3(q + 6), v3(q + 9), p) ? 1 : 0;
	}
}
void ref_ld_random_val(uint32_t n, const uint32_t* index, const uint32_t* seed, float* out) {
	for (uint32_t i = 0; i < n; ++i) out[i] = ld_random_val(index[i], seed[i]);
}
void ref_ld_random_pixel_offset(uint32_t n, const uint32_t* spp, float* out2) {
	for (uint32_t i = 0; i < n; ++i) { Vector2f o = ld_random_pixel_offset(spp[i]); out2[2 * i] = o.x(); out2[2 * i + 1] = o.y(); }
}
void ref_sobol(uint32_t n, const uint32_t* index, uint32_t dim, uint32_t* out) {
	for (uint32_t i = 0; i < n; ++i) out[i] = sobol(index[i], dim);
}
void ref_ray_intersect(uint32_t n, const float* box6, const float* o3, const float* d3, float* out2, uint8_t* contains_o) {
	for (uint32_t i = 0; i < n; ++i) {
		BoundingBox b = box(box6 + 6 * (size_t)i, box6 + 6 * (size_t)i + 3);
		Vector2f r = b.ray_intersect(v3(o3 + 3 * (size_t)i), v3(d3 + 3 * (size_t)i));
		out2[2 * i] = r.x(); out2[2 * i + 1] = r.y();
		contains_o[i] = b.contains(v3(o3 + 3 * (size_t)i)) ? 1 : 0;
	}
}
void ref_box_intersects_triangle(uint32_t n, const float* box6, const float* tri9, uint8_t* out) {
	for (uint32_t i = 0; i < n; ++i) {
		BoundingBox b = box(box6 + 6 * (size_t)i, box6 + 6 * (size_t)i + 3);
		Triangle t{v3(tri9 + 9 * (size_t)i), v3(tri9 + 9 * (size_t)i + 3), v3(tri9 + 9 * (size_t)i + 6)};
		out[i] = b.intersects(t) ? 1 : 0;
	}
}
// step / mip / cell math of common_nerf.cu for (pos, dir, t, cone_angle, mip) tuples
void ref_grid_math(uint32_t n, const float* pos3, const float* dir3, const float* t, const float* cone, const uint32_t* mip, float* calc_dt_out, int32_t* mip_from_pos_out,
                   int32_t* mip_from_dt_out, uint32_t* cell_idx_out, float* dist_out, float* advance_out) {
	for (uint32_t i = 0; i < n; ++i) {
		const Vector3f pos = v3(pos3 + 3 * (size_t)i), dir = v3(dir3 + 3 * (size_t)i), idir = dir.cwiseInverse();
		const float dt = calc_dt(t[i], cone[i]);
		calc_dt_out[i] = dt;
		mip_from_pos_out[i] = mip_from_pos(pos);
		mip_from_dt_out[i] = mip_from_dt(dt, pos);
		cell_idx_out[i] = cascaded_grid_idx_at(pos, mip[i]);
		const uint32_t res = NERF_GRIDSIZE() >> mip[i];
		dist_out[i] = distance_to_next_voxel(pos, dir, idir, res);
		advance_out[i] = advance_to_next_voxel(t[i], cone[i], pos, dir, idir, res);
	}
}
void ref_warp(uint32_t n, const float* box6, const float* pos3, const float* dt, float* warp_pos3, float* unwarp_pos3, float* warp_dir3, float* unwarp_dir3, float* warp_dt_out,
              float* unwarp_dt_out) {
	for (uint32_t i = 0; i < n; ++i) {
		BoundingBox b = box(box6, box6 + 3);
		const Vector3f p = v3(pos3 + 3 * (size_t)i);
		const Vector3f a = warp_position(p, b), u = unwarp_position(p, b), wd = warp_direction(p), ud = unwarp_direction(p);
		for (int c = 0; c < 3; ++c) { warp_pos3[3 * (size_t)i + c] = a[c]; unwarp_pos3[3 * (size_t)i + c] = u[c]; warp_dir3[3 * (size_t)i + c] = wd[c]; unwarp_dir3[3 * (size_t)i + c] = ud[c]; }
		warp_dt_out[i] = warp_dt(dt[i]);
		unwarp_dt_out[i] = unwarp_dt(dt[i]);
	}
}
void ref_evaluate_sh9(uint32_t n, const float* sh27, const float* dir3, float* rgb3) {
	for (uint32_t i = 0; i < n; ++i) {
		SH9RGB sh; memcpy(sh.data(), sh27 + 27 * (size_t)i, 108);
		Vector3f c = evaluate_sh9(sh, v3(dir3 + 3 * (size_t)i));
		for (int k = 0; k < 3; ++k) rgb3[3 * (size_t)i + k] = c[k];
	}
}
void ref_activations(uint32_t n, const float* x, float* srgb_to_linear_out, float* rgb_logistic, float* rgb_exp, float* density_exp) {
	for (uint32_t i = 0; i < n; ++i) {
		srgb_to_linear_out[i] = srgb_to_linear(x[i]);
		rgb_logistic[i] = network_to_rgb(x[i], ENerfActivation::Logistic);
		rgb_exp[i] = network_to_rgb(x[i], ENerfActivation::Exponential);
		density_exp[i] = network_to_density(x[i], ENerfActivation::Exponential);
	}
}
void ref_pixel_to_ray(uint32_t n, const int32_t* pixel2, const nrs_render_params* p, float* origin3, float* dir3) {
	const Vector2i resolution(p->resolution[0], p->resolution[1]);
	const Vector2f focal_length(p->focal_length[0], p->focal_length[1]), screen_center(p->screen_center[0], p->screen_center[1]);
	const Matrix<float, 3, 4> cam = m34(p->camera_matrix1);
	for (uint32_t i = 0; i < n; ++i) {
		CameraDistortion cd{};
		cd.mode = (ECameraDistortionMode)p->distortion_mode;
		for (int k = 0; k < 7; ++k) cd.params[k] = p->distortion_params[k];
		Ray r = pixel_to_ray(p->spp_index, Vector2i(pixel2[2 * i], pixel2[2 * i + 1]), resolution, focal_length, cam, screen_center, (bool)p->snap_to_pixel_centers, p->slice_plane_z, p->dof, cd,
		                     p->d_distortion_map, Vector2i(p->distortion_resolution[0], p->distortion_resolution[1]));
		for (int c = 0; c < 3; ++c) { origin3[3 * (size_t)i + c] = r.o[c]; dir3[3 * (size_t)i + c] = r.d[c]; }
	}
}
void ref_upper_cell_idx(uint32_t n, const uint32_t* cell_idx, const uint32_t* target_level, uint32_t* out) { // selection_utils.cu:36-48
	for (uint32_t i = 0; i < n; ++i) out[i] = get_upper_cell_idx(cell_idx[i], target_level[i]);
}
void ref_cell_functions(uint32_t n, const uint32_t* xyz_level4, const float* pos3, float* cell_pos3, int32_t* cell_at_pos3) {
	for (uint32_t i = 0; i < n; ++i) {
		const uint32_t* q = xyz_level4 + 4 * (size_t)i;
		Vector3f c = get_cell_pos(q[0], q[1], q[2], q[3]);
		Vector3i a = get_cell_at_pos(v3(pos3 + 3 * (size_t)i), q[3]);
		for (int k = 0; k < 3; ++k) { cell_pos3[3 * (size_t)i + k] = c[k]; cell_at_pos3[3 * (size_t)i + k] = a[k]; }
	}
}

} // extern "C"
