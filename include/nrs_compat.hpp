// nrs_compat.hpp -- header-only C++ adaptor over the C-ABI of nrs.h that gives the render path the SHAPE of the reference's
// own C++ interfaces, so call sites in src/testbed.cu / src/testbed_nerf.cu map 1:1:
//
//   ngp::NerfNetwork<T>        include/neural-graphics-primitives/nerf_network.h:86        -> nrs::compat::NerfNetwork
//   ngp::CageDeformation       include/neural-graphics-primitives/editing/edit_operator.h  -> nrs::compat::CageDeformation
//   ngp::CudaRenderBuffer      include/neural-graphics-primitives/render_buffer.h:164      -> nrs::compat::RenderBuffer (view)
//   ngp::Testbed::render_nerf  src/testbed_nerf.cu:3066                                    -> nrs::compat::Testbed::render_nerf
//
// No Eigen / tiny-cuda-nn types: matrices are column-major float arrays (what Eigen::Matrix<float,3,4>::data() yields),
// streams are passed as void* (hipStream_t), errors become std::runtime_error (the reference throws from CUDA_CHECK_THROW).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "nrs.h"

namespace nrs {
namespace compat {

inline void check(int status, const char* what) {
	if (status != NRS_OK) throw std::runtime_error(std::string(what) + ": " + nrs_last_error());
}

class Context {
public:
	explicit Context(int device = 0) { check(nrs_ctx_create(device, &m_ctx), "nrs_ctx_create"); }
	~Context() { nrs_ctx_destroy(m_ctx); }
	Context(const Context&) = delete;
	Context& operator=(const Context&) = delete;
	nrs_ctx* get() const { return m_ctx; }

private:
	nrs_ctx* m_ctx = nullptr;
};

// Non-owning matrix views with tcnn's meaning: GPUMatrixDynamic<float> (column-major, m rows = floats per sample) and
// GPUMatrixDynamic<T> with a selectable layout for the fp16 output.
struct InputMatrix {
	const float* data;  // device
	uint32_t rows;      // floats per sample (7 for inference, 3..7 for density)
	uint32_t n;         // samples (columns)
};
struct OutputMatrix {
	void* data;         // device, fp16
	uint32_t rows;      // 16
	uint32_t n;         // columns allocated (n_el >= samples for the planes layout)
	nrs_layout layout;  // NRS_PLANES = tcnn RM (row-major 16 x n), NRS_INTERLEAVED = tcnn CM
};

class NerfNetwork {
public:
	NerfNetwork(Context& ctx, const nrs_model_desc& desc) : m_desc(desc) { check(nrs_model_create(ctx.get(), &desc, &m_model), "nrs_model_create"); }
	~NerfNetwork() { nrs_model_destroy(m_model); }
	NerfNetwork(const NerfNetwork&) = delete;
	NerfNetwork& operator=(const NerfNetwork&) = delete;

	// tcnn::Network interface subset used on the path (nerf_network.h:97-120)
	uint32_t padded_output_width() const { return NRS_NETWORK_OUTPUT_WIDTH; }
	uint32_t input_width() const { return NRS_NETWORK_INPUT_FLOATS; }
	uint32_t n_extra_dims() const { return 0; }
	size_t n_params() const { return nrs_model_n_params(&m_desc); }

	// set_params(params, inference_params, ...) of the reference takes device pointers into the trainer's blob; here the fp16
	// blob (density MLP | rgb MLP | hash grid, nerf_network_full.h:316-349) is handed over from the host once.
	void set_params(const void* h_params_fp16, size_t n) { check(nrs_model_set_params(m_model, h_params_fp16, n), "nrs_model_set_params"); }
	// ... or, as in the reference, from device pointers (call again after every optimiser step: nrs.h)
	void set_params_device(const void* d_params_fp16, size_t n, void* stream) { check(nrs_model_set_params_device(m_model, d_params_fp16, n, stream), "nrs_model_set_params_device"); }
	// budget of the cell-record cache (no counterpart in the reference; results do not depend on it), 0 = off
	void set_cell_cache(size_t max_bytes) { check(nrs_model_set_cell_cache(m_model, max_bytes), "nrs_model_set_cell_cache"); }

	void inference_mixed_precision(void* stream, const InputMatrix& input, OutputMatrix& output, bool /*use_inference_params*/ = true) {
		if (input.rows != NRS_NETWORK_INPUT_FLOATS) throw std::runtime_error("NerfNetwork::inference_mixed_precision: input must have 7 rows");
		check(nrs_network_inference(m_model, stream, input.n, input.data, output.data, output.n, output.layout), "nrs_network_inference");
	}
	void density(void* stream, const InputMatrix& input, OutputMatrix& output, bool /*use_inference_params*/ = true) {
		check(nrs_network_density(m_model, stream, input.n, input.data, input.rows, output.data, output.n, output.layout), "nrs_network_density");
	}
	// tcnn::DifferentiableObject::input_gradient(stream, dim, input, d_dinput) as the path calls it (dim = 3: the density; testbed_nerf.cu:2924, :4491).
	// d_grad: [n x 3] f32, the position rows of the reference's gradient matrix (the dt / direction rows are zero there).
	void input_gradient(void* stream, uint32_t dim, const InputMatrix& input, float* d_grad_nx3) {
		if (dim != 3) throw std::runtime_error("NerfNetwork::input_gradient: the render path differentiates output 3 (the density) only");
		check(nrs_network_input_gradient(m_model, stream, input.n, input.data, input.rows, d_grad_nx3), "nrs_network_input_gradient");
	}
	// tcnn::Network::visualize_activation(stream, layer, dimension, input, output): the activation itself, f32 [n] (testbed_nerf.cu:2926, :3159)
	void visualize_activation(void* stream, uint32_t layer, uint32_t dimension, const InputMatrix& input, float* d_out_n) {
		if (input.rows != NRS_NETWORK_INPUT_FLOATS) throw std::runtime_error("NerfNetwork::visualize_activation: input must have 7 rows");
		check(nrs_network_visualize_activation(m_model, stream, layer, dimension, input.n, input.data, d_out_n), "nrs_network_visualize_activation");
	}

	// Testbed::m_nerf.density_grid_bitfield / update_density_grid_mean_and_bitfield
	void set_density_bitfield(const uint8_t* h_bits, size_t n) { check(nrs_model_set_density_bitfield(m_model, h_bits, n), "nrs_model_set_density_bitfield"); }
	void set_density_grid(const float* h_grid, size_t n) { check(nrs_model_set_density_grid(m_model, h_grid, n), "nrs_model_set_density_grid"); }

	void get_density_grid(float* h_grid, size_t n) { check(nrs_model_get_density_grid(m_model, h_grid, n), "nrs_model_get_density_grid"); }

	nrs_model* get() const { return m_model; }
	const nrs_model_desc& desc() const { return m_desc; }

private:
	nrs_model_desc m_desc;
	nrs_model* m_model = nullptr;
};

// EditOperator (edit_operator.h:43-91): the interface NerfTracer::m_edit_operators holds (testbed.h:237) -- cage deformations and affine
// duplications alike.  An operator owns its device tables.
class EditOperator {
public:
	virtual ~EditOperator() { nrs_edit_destroy(m_edit); }
	EditOperator(const EditOperator&) = delete;
	EditOperator& operator=(const EditOperator&) = delete;

	void map_rays(void* stream, float* d_nerf_coords /*[n x 7]*/, uint8_t* d_empty_mask, uint32_t n_elements) const {
		check(nrs_edit_map_rays(m_edit, stream, n_elements, d_nerf_coords, d_empty_mask), "nrs_edit_map_rays");
	}
	void map_positions(void* stream, float* d_nerf_pos, uint32_t stride_floats, uint8_t* d_empty_mask, uint32_t n_elements) const {
		check(nrs_edit_map_positions(m_edit, stream, n_elements, d_nerf_pos, stride_floats, d_empty_mask), "nrs_edit_map_positions");
	}
	nrs_edit* get() const { return m_edit; }

protected:
	EditOperator() = default;
	nrs_edit* m_edit = nullptr;
};

class CageDeformation : public EditOperator {
public:
	CageDeformation(Context& ctx, const nrs_model_desc& desc, const nrs_tet_mesh& mesh) { check(nrs_edit_create(ctx.get(), &desc, &mesh, &m_edit), "nrs_edit_create"); }
	// The per-gizmo-move chain (Cage::interpolate_with_mvc -> TetMesh::post_update_vertices -> build_tet_grid ->
	// update_local_rotations), on the device.  set_mvc once after Cage::compute_mvc; update_cage per move.
	void set_mvc(const float* h_weights, uint32_t n_cage_vertices) { check(nrs_edit_set_mvc(m_edit, h_weights, n_cage_vertices), "nrs_edit_set_mvc"); }
	void update_cage(void* stream, const float* h_cage_vertices, uint32_t n_cage_vertices) {
		check(nrs_edit_update_cage(m_edit, stream, h_cage_vertices, n_cage_vertices), "nrs_edit_update_cage");
	}
	void update_vertices(void* stream, const float* h_vertices, uint32_t n_vertices) {
		check(nrs_edit_update_vertices(m_edit, stream, h_vertices, n_vertices), "nrs_edit_update_vertices");
	}
	// GrowingSelection::interpolate_poisson_boundary (growing_selection.cu:2350): cage-vertex membrane terms -> the operator's per-tet-vertex ones;
	// h_gamma = TetMesh::gamma_coordinates or nullptr for the weights given to set_mvc
	void interpolate_poisson_boundary(void* stream, const float* h_gamma, uint32_t n_cage_vertices, const float* h_inside_density, const float* h_outside_density,
	                                  const float* h_inside_shs, const float* h_outside_shs, float residual_amplitude) {
		check(nrs_edit_poisson_interpolate(m_edit, stream, h_gamma, n_cage_vertices, h_inside_density, h_outside_density, h_inside_shs, h_outside_shs, residual_amplitude),
		      "nrs_edit_poisson_interpolate");
	}
};

class AffineDuplication : public EditOperator { // editing/affine_duplication.h:23-106
public:
	AffineDuplication(Context& ctx, const nrs_model_desc& desc, const nrs_affine_duplication& op) {
		check(nrs_edit_create_affine(ctx.get(), &desc, &op, &m_edit), "nrs_edit_create_affine");
	}
};

// View over the caller's frame / depth device arrays (CudaRenderBuffer::frame_buffer() / depth_buffer() / spp()).
struct RenderBuffer {
	float* frame_buffer;   // float4 [H*W], premultiplied linear RGBA; the caller clears it (clear_frame, testbed.cu:2635)
	float* depth_buffer;   // float [H*W]
	int width, height;     // in_resolution()
	uint32_t spp;          // sample index of this frame
	float* accumulate_buffer = nullptr; // float4 [H*W], optional: CudaRenderBuffer::m_accumulate_buffer
	// void CudaRenderBuffer::accumulate(float exposure, cudaStream_t stream) -- render_buffer.cu:540-560 (exposure is unused there too): the running mean of the spp frames
	void accumulate(Context& ctx, void* stream, nrs_color_space color_space = NRS_COLOR_LINEAR) {
		if (!accumulate_buffer) throw std::runtime_error("RenderBuffer::accumulate: no accumulate buffer");
		check(nrs_accumulate(ctx.get(), stream, (uint32_t)width, (uint32_t)height, frame_buffer, accumulate_buffer, spp, (uint32_t)color_space), "nrs_accumulate");
		++spp;
	}
};

// The slice of ngp::Testbed that render_nerf reads (SURVEY 8b "implicit inputs"), with the reference's member names.
class Testbed {
public:
	struct Nerf {
		float cone_angle_constant = 0.f;            // 0 for aabb_scale 1, 1/256 otherwise (testbed_nerf.cu:3410-3425)
		float rendering_min_transmittance = 0.01f;
		bool training_linear_colors = false;
	} m_nerf;
	float m_render_aabb_min[3] = {0, 0, 0}, m_render_aabb_max[3] = {1, 1, 1};
	bool m_snap_to_pixel_centers = true;
	bool m_enable_edits = true;
	nrs_render_mode m_render_mode = NRS_RENDER_SHADE;
	int m_visualized_layer = 0, m_visualized_dimension = -1; // testbed.h: > -1 selects render mode EncodingVis (testbed_nerf.cu:3072)
	std::vector<const EditOperator*> m_edit_operators; // NerfTracer::m_edit_operators (testbed.h:237), applied last-to-first
	bool m_poisson_target = true;                     // NerfTracer::m_poisson_target (testbed.h:219: true, its checkbox is commented out; passed to composite_kernel_nerf, testbed_nerf.cu:2983)
	int m_show_accel = -1;                            // m_nerf.show_accel: >= 0 forces that cascade as the minimum while marching (:2751, :2849), makes
	                                                  // every sample opaque (:788-790) and colours the occupancy cells in render mode Positions (:911-920)
	float m_dof = 0.f;                                // aperture of pixel_to_ray's thin-lens branch (common_device.cuh:285-293)
	float m_slice_plane_z = 0.f, m_scale = 1.f;       // plane_z = m_slice_plane_z + m_scale (testbed_nerf.cu:3067): focus distance / slice plane
	float m_dataset_scale = 1.f;                      // m_nerf.training.dataset.scale: depth_scale = 1 / it (:3113)
	// camera model and background (render_nerf passes them to init_rays_from_camera, :3078-3100).  Device pointers, owned by the caller.
	uint32_t m_render_distortion_mode = 0;            // m_nerf.render_distortion.mode when m_nerf.render_with_camera_distortion (ECameraDistortionMode)
	float m_render_distortion_params[7] = {};         // m_nerf.render_distortion.params
	const float* m_distortion_map = nullptr;          // m_distortion.map->params_inference() (float2 [res.y][res.x]) when render_with_camera_distortion, else NULL
	int m_distortion_resolution[2] = {0, 0};          // m_distortion.resolution
	const float* m_envmap = nullptr;                  // m_envmap.envmap->params_inference() (float RGBA [res.y][res.x]), NULL when the snapshot has none
	int m_envmap_resolution[2] = {0, 0};              // m_envmap.resolution
	int m_glow_mode = 0;                              // m_nerf.m_glow_mode, m_nerf.m_glow_y_cutoff: composite_kernel_nerf's grid / cut-line overlay (:806-903)
	float m_glow_y_cutoff = 0.f;

	// Testbed state update_density_grid_nerf_operator advances: m_rng, m_nerf.density_grid_ema_step, density_grid_decay, max_cascade
	nrs_grid_update m_density_grid_update{};
	void seed_density_grid_update(uint32_t max_cascade, uint64_t seed = 1337, float decay = 0.95f) {
		m_density_grid_update = nrs_grid_update{};
		m_density_grid_update.n_uniform_samples = NRS_GRID_VOLUME * (max_cascade + 1);
		m_density_grid_update.max_cascade = max_cascade;
		m_density_grid_update.decay = decay;
		nrs_rng_seed(seed, &m_density_grid_update.rng_state, &m_density_grid_update.rng_inc);
	}
	// void Testbed::update_density_grid_nerf_render(uint32_t n_iterations, bool reset_grid, cudaStream_t)  -- testbed_nerf.cu:3514
	void update_density_grid_nerf_render(NerfNetwork& network, uint32_t n_iterations, bool reset_grid, void* stream) {
		std::vector<nrs_edit*> edits;
		if (m_enable_edits) for (const EditOperator* op : m_edit_operators) edits.push_back(op->get());
		for (uint32_t i = 0; i < n_iterations; ++i) {
			m_density_grid_update.reset_grid = (reset_grid && i == 0) ? 1u : 0u;
			check(nrs_model_update_density_grid(network.get(), edits.data(), (int)edits.size(), &m_density_grid_update, stream),
			      "nrs_model_update_density_grid");
		}
		m_density_grid_update.reset_grid = 0;
	}

	// void Testbed::render_nerf(NerfNetwork<precision_t>&, CudaRenderBuffer&, const Vector2i& max_res, const Vector2f& focal_length,
	//     const Matrix<float,3,4>& camera_matrix0, const Matrix<float,3,4>& camera_matrix1, const Vector4f& rolling_shutter,
	//     const Vector2f& screen_center, bool apply_operators, cudaStream_t stream)              -- testbed.h:305
	void render_nerf(NerfNetwork& network, RenderBuffer& render_buffer, const int /*max_res*/[2], const float focal_length[2],
	                 const float camera_matrix0[12], const float camera_matrix1[12], const float rolling_shutter[4], const float screen_center[2],
	                 bool apply_operators, void* stream, nrs_render_stats* stats = nullptr) {
		nrs_render_params p{}; // = NRS_RENDER_PARAMS_INIT, spelled so that -Wextra stays quiet in C++
		p.struct_size = (uint32_t)sizeof(nrs_render_params);
		p.resolution[0] = render_buffer.width;
		p.resolution[1] = render_buffer.height;
		for (int i = 0; i < 2; ++i) { p.focal_length[i] = focal_length[i]; p.screen_center[i] = screen_center[i]; }
		for (int i = 0; i < 12; ++i) { p.camera_matrix0[i] = camera_matrix0[i]; p.camera_matrix1[i] = camera_matrix1[i]; }
		for (int i = 0; i < 4; ++i) p.rolling_shutter[i] = rolling_shutter[i];
		for (int i = 0; i < 3; ++i) { p.render_aabb_min[i] = m_render_aabb_min[i]; p.render_aabb_max[i] = m_render_aabb_max[i]; }
		p.spp_index = render_buffer.spp;
		p.snap_to_pixel_centers = m_snap_to_pixel_centers;
		p.min_transmittance = m_nerf.rendering_min_transmittance;
		p.cone_angle_constant = m_nerf.cone_angle_constant;
		p.render_mode = m_visualized_dimension > -1 ? (uint32_t)NRS_RENDER_ENCODING_VIS : (uint32_t)m_render_mode; // testbed_nerf.cu:3072
		p.visualized_layer = (uint32_t)m_visualized_layer;
		p.visualized_dimension = m_visualized_dimension > -1 ? (uint32_t)m_visualized_dimension : 0u;
		p.linear_colors = m_nerf.training_linear_colors;
		p.apply_operators = apply_operators && m_enable_edits;
		p.poisson_target = m_poisson_target ? 1u : 0u;
		p.min_mip = m_show_accel >= 0 ? (uint32_t)m_show_accel : 0u;
		p.show_accel = m_show_accel >= 0 ? 1u : 0u;
		p.dof = m_dof;
		p.slice_plane_z = m_slice_plane_z + m_scale;
		p.depth_scale = 1.0f / m_dataset_scale;
		p.distortion_mode = m_render_distortion_mode;
		for (int i = 0; i < 7; ++i) p.distortion_params[i] = m_render_distortion_params[i];
		p.d_distortion_map = m_distortion_map;
		p.d_envmap = m_envmap;
		p.glow_mode = (uint32_t)m_glow_mode;
		p.glow_y_cutoff = m_glow_y_cutoff;
		for (int i = 0; i < 2; ++i) { p.distortion_resolution[i] = m_distortion_resolution[i]; p.envmap_resolution[i] = m_envmap_resolution[i]; }
		std::vector<nrs_edit*> edits;
		for (const EditOperator* op : m_edit_operators) edits.push_back(op->get());
		check(nrs_render_nerf(network.get(), &p, edits.data(), (int)edits.size(), render_buffer.frame_buffer, render_buffer.depth_buffer, nullptr, stream,
		                      stats),
		      "nrs_render_nerf");
	}
};

} // namespace compat
} // namespace nrs
