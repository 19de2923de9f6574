// ref_json.cpp -- the REFERENCE's own edit serialisation code, compiled from where it lies under /root/reference.
// TEST INFRASTRUCTURE ONLY: built by oracle/Makefile into oracle/_ref/libref_json.so where the reference is mounted; used by
// tests/golden/make_ref_edits_golden.py to write tests/golden/ref_edits_*.json (files as Testbed::save_edits writes them) and by tests/test_ref_pin.py to
// write such files live.  They pin nrs_edits_open.  Nothing of the product links or loads it.
//
// What is the reference's code here (compiled, not restated):
//   * whole headers: json_binding.h (Eigen <-> json, BoundingBox, NerfDataset, the std::vector writers / readers :27-321), editing/datastructures/mesh.h,
//     tet_mesh.h (to_json / from_json of TetMesh :136-174), cage.h (:100-145), editing/tools/affine_bounding_box.cuh (:105-143);
//   * member functions cut out of the .cu files by oracle/ref_extract.py (each fragment carries a #line into the reference):
//     GrowingSelection::to_json (growing_selection.cu:2459-2480), RegionGrowing::to_json (region_growing.cu:187-198), CageDeformation::to_json
//     (cage_deformation.cu:817-825), AffineDuplication::to_json (affine_duplication.cu:356-369), Testbed::save_edits (testbed.cu:3190-3204),
//     Testbed::save_snapshot (testbed.cu:3090-3113; its trainer and GPU-memory conversions are tiny-cuda-nn's and restated: see there).
// What is NOT the reference's code: <json/json.hpp> (nlohmann/json is vendored through the empty tiny-cuda-nn submodule: oracle/ref_stubs/json/json.hpp models
// its conversion dispatch, sorted keys and null-for-untouched values; floats are printed with 17 significant digits where nlohmann prints the shortest
// round-trip form -- same doubles), <Eigen/Dense>, and the CLASSES the cut-out members belong to: the real GrowingSelection / CageDeformation /
// AffineDuplication / Testbed drag in the GUI, the trainer and tiny-cuda-nn's networks, so they are declared here with the serialised members only, under the
// reference's names and types (growing_selection.h:105-120, :233-249; region_growing.h:69-73; affine_duplication.h:86-95; testbed.h:206-208).
typedef unsigned int GLuint; // mesh.h / tet_mesh.h / cage.h keep GL handles as private members (declared by the GL headers of an NGP_GUI build)
#include <memory> // (cage.h names std::shared_ptr without including <memory>)
#include <tiny-cuda-nn/common.h>

#include <neural-graphics-primitives/common.h>
#include <neural-graphics-primitives/json_binding.h>
#include <neural-graphics-primitives/editing/datastructures/mesh.h>
#include <neural-graphics-primitives/editing/datastructures/tet_mesh.h>
#include <neural-graphics-primitives/editing/datastructures/cage.h>
#include <neural-graphics-primitives/editing/tools/affine_bounding_box.cuh>

#include <filesystem/path.h>

#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <queue>
#include <string>
#include <vector>

namespace fs = ::filesystem; // testbed.cu:64

NGP_NAMESPACE_BEGIN

typedef float float_t;            // growing_selection.h:89-90
typedef Eigen::Vector3f point_t;

struct RegionGrowing { // region_growing.h:69-73 (the serialised members)
	std::vector<uint8_t> m_selection_grid_bitfield;
	std::vector<Eigen::Vector3f> m_selection_points;
	std::vector<uint32_t> m_selection_cell_idx;
	std::vector<float> m_density_grid_host;
	nlohmann::json to_json();
};
#include "region_growing_to_json.inc"

struct GrowingSelection { // growing_selection.h:105-120, :233-249 (the serialised members)
	Mesh<float_t, point_t> selection_mesh;
	Cage<float_t, point_t> proxy_cage;
	std::shared_ptr<TetMesh<float_t, point_t>> tet_interpolation_mesh;
	std::vector<Eigen::Vector3f> m_projected_pixels;
	std::vector<uint8_t> m_projected_labels;
	std::vector<uint32_t> m_projected_cell_idx;
	std::vector<Eigen::Vector3f> m_selection_points;
	std::vector<uint8_t> m_selection_labels;
	std::vector<uint32_t> m_selection_cell_idx;
	std::vector<uint8_t> m_selection_grid_bitfield;
	int m_growing_level = 0;
	RegionGrowing m_region_growing;
	void to_json(nlohmann::json& j);
};
#include "growing_selection_to_json.inc"

struct EditOperator { // edit_operator.h:93
	virtual ~EditOperator() {}
	virtual nlohmann::json to_json() = 0;
};
struct CageDeformation : EditOperator { // cage_deformation.h:154, :160 (m_growing_selection)
	GrowingSelection m_growing_selection;
	nlohmann::json to_json() override;
};
#include "cage_deformation_to_json.inc"

struct AffineDuplication : EditOperator { // affine_duplication.h:86-95 (the serialised members)
	AffineBoundingBox m_selection_box;
	Eigen::Vector3f m_translation;
	Eigen::Vector3f m_scale;
	Eigen::Matrix3f m_rotation_matrix;
	bool m_hide_original = false;
	bool m_correct_dir = true;
	nlohmann::json to_json() override;
};
#include "affine_duplication_to_json.inc"

// ---- Testbed::save_snapshot (testbed.cu:3090-3113) --------------------------------------------------------------------------------------------------------
// Compiled: the member function itself (key names under "snapshot", what goes where) and to_json(NerfDataset) (json_binding.h:136-160, whole header).
// Stand-ins, because tiny-cuda-nn is absent: Trainer::serialize (restated as recalled: {"n_params", "params_type": "__half", "params_binary": the half parameters as
// a binary value}) and the conversion of the density grid's GPUMemory<float> to json (tiny-cuda-nn's gpu_memory_json.h as recalled: the raw bytes as a binary value).
// The three keys of the trainer therefore stay UNPINNED; everything else in the file is written by reference code.
using json = nlohmann::json; // (testbed.cu:77: `using namespace tcnn`, whose `json` is nlohmann::json)
struct HostBytes { std::vector<uint8_t> bytes; };
inline void to_json(nlohmann::json& j, const HostBytes& b) { j = nlohmann::json::binary(b.bytes); }
struct SnapshotTrainer {
	HostBytes params_half;
	size_t n_params = 0;
	nlohmann::json serialize(bool /*serialize_optimizer*/) {
		nlohmann::json data;
		data["n_params"] = n_params;
		data["params_type"] = "__half";
		data["params_binary"] = params_half;
		return data;
	}
};

#include "merge_parent_network_config.inc" // testbed.cu:86-97: the "parent" chain of configs/nerf/*.json (json::parse with comments, merge_patch)

struct Testbed { // testbed.h:206-208 (NerfTracer::edit_operators) under Testbed::m_nerf.tracer; the members save_snapshot touches (testbed.h:507-640)
	struct Tracer {
		std::vector<std::shared_ptr<EditOperator>> m_edit_operators;
		std::vector<std::shared_ptr<EditOperator>>& edit_operators() { return m_edit_operators; }
	};
	struct Nerf {
		Tracer tracer;
		HostBytes density_grid; // tcnn::GPUMemory<float> (testbed.h): 5 * 128^3 floats
		struct Training {
			struct Counters { uint32_t rays_per_batch = 1 << 16, measured_batch_size = 0, measured_batch_size_before_compaction = 0; } counters_rgb; // testbed.h:565-573
			NerfDataset dataset;
		} training;
	} m_nerf;
	nlohmann::json m_network_config;
	fs::path m_network_config_path;
	std::shared_ptr<SnapshotTrainer> m_trainer;
	ETestbedMode m_testbed_mode = ETestbedMode::Nerf;
	uint32_t m_training_step = 0;
	float m_loss_scalar = 0.f;
	void save_edits(const std::string& filepath_string);
	void save_snapshot(const std::string& filepath_string, bool include_optimizer_state);
};
#include "testbed_save_edits.inc"
#include "testbed_save_snapshot.inc"

NGP_NAMESPACE_END

using namespace ngp;

namespace {
std::vector<Eigen::Vector3f> vec3s(const float* p, uint32_t n) {
	std::vector<Eigen::Vector3f> v;
	v.reserve(n);
	for (uint32_t i = 0; i < n; ++i) v.push_back(Eigen::Vector3f(p[3 * i], p[3 * i + 1], p[3 * i + 2]));
	return v;
}
Eigen::Matrix3f mat3_cm(const float* p) { Eigen::Matrix3f m; memcpy(m.data(), p, 36); return m; } // column-major 9 floats
thread_local std::string g_err;
} // namespace

extern "C" {

// One edit operator as the harness describes it.  kind 0 = cage_deformation, 1 = affine_duplication.
struct RefJsonOp {
	int32_t kind;
	// cage_deformation: the proxy cage (Cage<float, Vector3f>) ...
	uint32_t n_cage_vertices, n_cage_indices;
	const float* cage_vertices;          // [n_cage_vertices][3]
	const float* cage_original_vertices; // [n_cage_vertices][3]
	const uint32_t* cage_indices;        // [n_cage_indices] (triangles)
	const float* cage_inside_density;    // [n_cage_vertices] or NULL  (membrane terms of the cage, cage.h:62-65)
	const float* cage_outside_density;   // [n_cage_vertices] or NULL
	const float* cage_inside_shs;        // [n_cage_vertices][27] column-major 9 x 3 (SH9RGB) or NULL
	const float* cage_outside_shs;       // same
	// ... and the interpolation mesh (TetMesh<float, Vector3f>); n_vertices == 0: the operator has none yet (growing_selection.cu:2477)
	uint32_t n_vertices, n_tets, n_surface_indices;
	const float* vertices;               // [n_vertices][3]
	const float* original_vertices;      // [n_vertices][3]
	const uint32_t* tets;                // [n_tets][4]
	const uint32_t* surface_indices;     // [n_surface_indices] (Mesh::indices of the tet mesh: its boundary triangles) or NULL
	const float* mvc_coordinates;        // [n_vertices][n_cage_vertices] or NULL (empty: the reference writes null)
	const float* gamma_coordinates;      // same shape or NULL
	// affine_duplication
	float selection_center[3], selection_scale[3], selection_rot[9]; // AffineBoundingBox center / scale / rot_matrix (column-major)
	float selection_min[3], selection_max[3];
	float translation[3], scale[3], rotation[9];
	int32_t hide_original, correct_dir;
};

const char* refjson_last_error() { return g_err.c_str(); }

// Testbed::save_edits(path) over operators built from `ops`: every byte of the file comes out of the reference's to_json code.
int refjson_save_edits(const char* path, const RefJsonOp* ops, uint32_t n_ops) {
	try {
		Testbed tb;
		for (uint32_t k = 0; k < n_ops; ++k) {
			const RefJsonOp& o = ops[k];
			if (o.kind == 0) {
				auto op = std::make_shared<CageDeformation>();
				GrowingSelection& gs = op->m_growing_selection;
				Cage<float_t, point_t>& cage = gs.proxy_cage;
				cage.vertices = vec3s(o.cage_vertices, o.n_cage_vertices);
				cage.original_vertices = vec3s(o.cage_original_vertices, o.n_cage_vertices);
				cage.indices.assign(o.cage_indices, o.cage_indices + o.n_cage_indices);
				// what Cage's constructor sizes (cage.h:60-68)
				cage.initial_normals.resize(o.n_cage_vertices);
				cage.normals.resize(o.n_cage_vertices);
				cage.labels.resize(o.n_cage_vertices);
				cage.colors.resize(o.n_cage_vertices);
				if (o.cage_inside_density) cage.inside_density.assign(o.cage_inside_density, o.cage_inside_density + o.n_cage_vertices);
				if (o.cage_outside_density) cage.outside_density.assign(o.cage_outside_density, o.cage_outside_density + o.n_cage_vertices);
				for (int which = 0; which < 2; ++which) {
					const float* src = which ? o.cage_outside_shs : o.cage_inside_shs;
					if (!src) continue;
					std::vector<SH9RGB>& dst = which ? cage.outside_shs : cage.inside_shs;
					dst.resize(o.n_cage_vertices);
					for (uint32_t i = 0; i < o.n_cage_vertices; ++i) memcpy(dst[i].data(), src + 27 * i, 27 * sizeof(float));
				}
				if (o.n_vertices) {
					auto tm = std::make_shared<TetMesh<float_t, point_t>>();
					tm->vertices = vec3s(o.vertices, o.n_vertices);
					tm->original_vertices = vec3s(o.original_vertices, o.n_vertices);
					tm->tets.assign(o.tets, o.tets + 4 * (size_t)o.n_tets);
					if (o.surface_indices) tm->indices.assign(o.surface_indices, o.surface_indices + o.n_surface_indices);
					// what TetMesh's constructor sizes (tet_mesh.h:104-106) and post_update_vertices derives (tet_mesh.cu:12-20: the boxes are not read back by the path)
					tm->labels.resize(o.n_vertices);
					tm->colors.resize(o.n_vertices, default_tet_color);
					for (uint32_t i = 0; i < o.n_vertices; ++i) { tm->bbox.enlarge(tm->vertices[i]); tm->original_bbox.enlarge(tm->original_vertices[i]); }
					tm->warped_bbox = tm->bbox;
					tm->original_warped_bbox = tm->original_bbox;
					for (int which = 0; which < 2; ++which) {
						const float* src = which ? o.gamma_coordinates : o.mvc_coordinates;
						if (!src) continue;
						std::vector<std::vector<float>>& dst = which ? tm->gamma_coordinates : tm->mvc_coordinates;
						dst.resize(o.n_vertices);
						for (uint32_t i = 0; i < o.n_vertices; ++i) dst[i].assign(src + (size_t)i * o.n_cage_vertices, src + (size_t)(i + 1) * o.n_cage_vertices);
					}
					gs.tet_interpolation_mesh = tm;
				}
				tb.m_nerf.tracer.edit_operators().push_back(op);
			} else if (o.kind == 1) {
				auto op = std::make_shared<AffineDuplication>();
				AffineBoundingBox& b = op->m_selection_box;
				b.min = Eigen::Vector3f(o.selection_min[0], o.selection_min[1], o.selection_min[2]);
				b.max = Eigen::Vector3f(o.selection_max[0], o.selection_max[1], o.selection_max[2]);
				b.center = Eigen::Vector3f(o.selection_center[0], o.selection_center[1], o.selection_center[2]);
				b.scale = Eigen::Vector3f(o.selection_scale[0], o.selection_scale[1], o.selection_scale[2]);
				b.rot_matrix = mat3_cm(o.selection_rot);
				op->m_translation = Eigen::Vector3f(o.translation[0], o.translation[1], o.translation[2]);
				op->m_scale = Eigen::Vector3f(o.scale[0], o.scale[1], o.scale[2]);
				op->m_rotation_matrix = mat3_cm(o.rotation);
				op->m_hide_original = o.hide_original != 0;
				op->m_correct_dir = o.correct_dir != 0;
				tb.m_nerf.tracer.edit_operators().push_back(op);
			} else {
				g_err = "refjson_save_edits: unknown operator kind";
				return -1;
			}
		}
		tb.save_edits(path);
		return 0;
	} catch (const std::exception& e) {
		g_err = e.what();
		return -1;
	}
}

// Testbed::save_snapshot(path) with `config_path` (one of the reference's configs/nerf/*.json, parsed as it lies there) as m_network_config; log2_hashmap_size > 0
// overrides the encoding's table size (a smaller golden file).  params: n_params halfs; grid: 5 * 128^3 floats.
int refjson_save_snapshot(const char* path, const char* config_path, int log2_hashmap_size, const uint16_t* params, uint64_t n_params, const float* grid, uint64_t n_grid,
                          int aabb_scale, uint32_t training_step, float loss) {
	try {
		Testbed tb;
		{ // Testbed::load_network_config's .json branch (testbed.cu:177-181)
			const fs::path network_config_path = config_path;
			std::ifstream f{network_config_path.str()};
			if (!f) { g_err = "cannot open the network config"; return -1; }
			nlohmann::json result = json::parse(f, nullptr, true, true);
			tb.m_network_config = merge_parent_network_config(result, network_config_path);
		}
		if (log2_hashmap_size > 0) tb.m_network_config["encoding"]["log2_hashmap_size"] = log2_hashmap_size;
		tb.m_trainer = std::make_shared<SnapshotTrainer>();
		tb.m_trainer->n_params = (size_t)n_params;
		tb.m_trainer->params_half.bytes.assign((const uint8_t*)params, (const uint8_t*)params + 2 * n_params);
		tb.m_nerf.density_grid.bytes.assign((const uint8_t*)grid, (const uint8_t*)grid + 4 * n_grid);
		tb.m_nerf.training.dataset.aabb_scale = aabb_scale;
		tb.m_training_step = training_step;
		tb.m_loss_scalar = loss;
		tb.save_snapshot(path, false);
		return 0;
	} catch (const std::exception& e) {
		g_err = e.what();
		return -1;
	}
}

// The reference's READERS on a file (from_json of Cage / TetMesh / AffineBoundingBox, json_binding.h): what the reference itself would load from `path`,
// re-serialised by the reference's writers into `out_path` -- a round trip through reference code only, used to show that a file of the harness's own writer
// (nerfshop_amd/formats.py) is one the reference reads.  Returns the number of operators, or -1.
int refjson_reload_edits(const char* path, const char* out_path) {
	try {
		std::ifstream i(path);
		if (!i) { g_err = "cannot open file"; return -1; }
		nlohmann::json j;
		i >> j;
		Testbed tb;
		for (auto& operator_json : j["edit_operators"]) { // Testbed::load_edits' dispatch, testbed.cu:3210-3234
			if (operator_json["type"] == "affine_duplication") {
				auto op = std::make_shared<AffineDuplication>(); // AffineDuplication(json, aabb), affine_duplication.h:31-40
				from_json(operator_json["selection_box"], op->m_selection_box);
				from_json(operator_json["translation"], op->m_translation);
				from_json(operator_json["scale"], op->m_scale);
				from_json(operator_json["rotation_matrix"], op->m_rotation_matrix);
				op->m_hide_original = operator_json["hide_original"];
				op->m_correct_dir = operator_json["correct_dir"];
				tb.m_nerf.tracer.edit_operators().push_back(op);
			} else if (operator_json["type"] == "cage_deformation") {
				auto op = std::make_shared<CageDeformation>(); // GrowingSelection(json, ...), growing_selection.cu:96-115
				GrowingSelection& gs = op->m_growing_selection;
				from_json(operator_json["projected_pixels"], gs.m_projected_pixels);
				from_json(operator_json["projected_labels"], gs.m_projected_labels);
				from_json(operator_json["projected_cell_idx"], gs.m_projected_cell_idx);
				from_json(operator_json["selection_points"], gs.m_selection_points);
				from_json(operator_json["selection_labels"], gs.m_selection_labels);
				from_json(operator_json["selection_cell_idx"], gs.m_selection_cell_idx);
				from_json(operator_json["m_selection_grid_bitfield"], gs.m_selection_grid_bitfield);
				gs.m_growing_level = operator_json["growing_level"];
				from_json(operator_json["selection_mesh"], gs.selection_mesh);
				from_json(operator_json["proxy_cage"], gs.proxy_cage);
				if (operator_json.contains("interpolation_mesh")) {
					gs.tet_interpolation_mesh = std::make_shared<TetMesh<float_t, point_t>>();
					from_json(operator_json["interpolation_mesh"], *gs.tet_interpolation_mesh);
				}
				tb.m_nerf.tracer.edit_operators().push_back(op);
			} else {
				throw std::runtime_error{"Invalid edit operator!"};
			}
		}
		tb.save_edits(out_path);
		return (int)tb.m_nerf.tracer.edit_operators().size();
	} catch (const std::exception& e) {
		g_err = e.what();
		return -1;
	}
}

} // extern "C"
