// nrs_api.cpp -- C++ host code behind the C-ABI of include/nrs.h: context, model, edit operators, render call.
// Plain C++17 + the HIP runtime API; no torch, no third-party dependencies.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "nrs_internal.h"

using namespace nrs;

static thread_local std::string g_err;
namespace nrs { void set_last_error(const char* msg) { g_err = msg ? msg : ""; } }
static int fail(int code, const std::string& msg) {
	g_err = msg;
	return code;
}
static int fail_hip(hipError_t e, const char* what) {
	g_err = std::string(what) + ": " + hipGetErrorString(e);
	return NRS_ERR_HIP;
}
#define HIP_TRY(call)                                      \
	do {                                                   \
		hipError_t e_ = (call);                            \
		if (e_ != hipSuccess) return fail_hip(e_, #call);  \
	} while (0)
#define NRS_TRY(call)                                                       \
	do {                                                                    \
		int s_ = (call);                                                    \
		if (s_ != NRS_OK) return (g_err = launch_last_error(), s_);         \
	} while (0)

struct nrs_ctx {
	int lane_teams = 0; // nrs_ctx_set_lane_teams: 0 = automatic, 1 / 2 / 4 = lanes per ray for every render launch
	int handover = -1;  // nrs_ctx_set_ray_handover: -1 = default (on; NRS_STEAL=0 turns it off), 0 / 1
	unsigned long long last_handover = 0; // rays | hand-overs << 32 of the last launch that returned statistics
	int device = 0;
	int n_cus = 0;
	size_t hbm_bytes = 0;
	char name[256] = {0};
	// per-launch scratch comes from small rings so that render calls issued back to back on DIFFERENT streams (double-buffered
	// frames) do not share a packet counter or an operator table: slot = launch number % kInFlight
	static constexpr int kInFlight = 8;
	RenderCounters* d_counters = nullptr;   // [kInFlight][2]: the render kernel of a slot zeroes the OTHER block for the slot's next launch (no memset between frames)
	uint8_t counter_parity[kInFlight] = {};   // block the slot's next launch uses; counters_clean: that block is known to be zero
	bool counters_clean[kInFlight] = {};
	DeviceEdit* d_edits = nullptr;          // [kInFlight + 1][kMaxEdits]; the last table belongs to the occupancy refresh
	std::atomic<uint32_t> launch_serial{0};
	std::mutex launch_mutex; // slot acquisition .. launch enqueued: two threads rendering on one ctx must not share a slot's counter block / parity
	// A slot is reused every kInFlight launches, possibly from ANOTHER stream: the launch that used it last records slot_done, and a
	// launch on a different stream makes its stream wait on that event before it clears the counters / re-sends the operator table
	// (same-stream reuse is ordered by the stream itself).
	hipEvent_t slot_done[kInFlight] = {};
	hipStream_t slot_stream[kInFlight] = {};
	bool slot_used[kInFlight] = {};
	float* d_mean = nullptr;
	unsigned long long* d_wave_log = nullptr; // profiling only (NRS_DEBUG & 4)
	unsigned long long* h_feedback = nullptr; // pinned, device-visible: written by the last workgroup of a render launch
	unsigned long long* d_feedback = nullptr; // its device address
	std::vector<DeviceEdit> edits_shadow = std::vector<DeviceEdit>((size_t)kInFlight * 32); // what each slot of d_edits holds
	int shadow_n[kInFlight] = {-1, -1, -1, -1, -1, -1, -1, -1};
	static constexpr int kMaxEdits = 32;
};

// (the cell-record cache's measured optimum on 1080p lego is 10 GiB = levels 0..11 of base.json's table, 9.2 GB: 12 levels 8.80, 14 levels (64 GB) 8.64, 10 levels 8.52,
// none 8.05 Gsamples/s -- the budget a caller who opts in would pass: include/nrs.h MEMORY NOTE)

struct nrs_model {
	nrs_ctx* ctx = nullptr;
	nrs_model_desc desc{};
	DeviceModel dm{};
	uint32_t total_entries = 0;
	uint32_t* d_grid = nullptr;
	uint16_t* d_wfrag = nullptr;
	uint16_t* d_wfrag_src = nullptr;     // make_weight_fragments as a permutation (source index + 1, 0 = padding), for nrs_model_set_params_device
	uint8_t* d_bitfield = nullptr;
	uint32_t* d_accel_masks = nullptr;   // 2 x kCoarseWords: accel_any.mask | accel_exact.mask
	OccAccel accel_any{}, accel_exact{}; // marching shortcuts for general step parameters / for cone_angle == 0 && min_mip == 0
	float* d_density_grid = nullptr;   // m_nerf.density_grid [5*128^3], kept for the occupancy refresh
	uint32_t* d_density_tmp = nullptr; // density_grid_tmp (float bits), allocated on first refresh
	bool have_params = false, have_bitfield = false;
	// cell records of the first `cached_levels` levels (nrs_model_set_cell_cache)
	uint4* d_records = nullptr;
	size_t records_bytes = 0, cell_cache_budget = 0;
	uint32_t cached_levels = 0;
	// sparse brick records of the levels after them (nrs_model_set_sparse_cell_cache)
	uint32_t* d_bricks = nullptr;   // brick tables of the sparse levels, concatenated
	uint32_t* d_slots = nullptr;    // brick number of every allocated brick, per level (slot_first[l] .. +slot_count[l])
	uint4* d_records2 = nullptr;
	size_t sparse_bytes = 0;        // tables + slots + records
	uint32_t sparse_first = 0, sparse_levels = 0;
	uint32_t slot_first[kLevels] = {}, slot_count[kLevels] = {};
};

struct nrs_edit {
	nrs_ctx* ctx = nullptr;
	DeviceEdit de{};
	std::vector<void*> allocs;
	uint32_t n_vertices = 0, n_tets = 0;
	// device-side authoring state (nrs_cage.hip): everything a cage move rewrites
	float* d_verts = nullptr;          // == de.verts
	uint32_t* d_lut_off = nullptr;     // == de.lut_off, [5*128^3 + 1]
	uint32_t* d_lut_idx = nullptr;     // == de.lut_idx
	size_t lut_idx_cap = 0;            // entries allocated
	float* d_rot = nullptr;            // == de.rot when rotations are on
	float* d_planes = nullptr;         // == de.planes, [T x 32] one 128-byte record per tet (tet_planes_kernel), follows the deformed vertices
	uint32_t* d_counts = nullptr;      // [5*128^3], all zero between builds
	uint32_t* d_tile_sums = nullptr;
	unsigned long long* d_hit_masks = nullptr; // per (tet, cascade): the count pass's first 128 cell / tet tests (two words per item), reused by the fill pass (nrs_cage.hip tet_mark_kernel)
	uint32_t* d_scratch = nullptr;     // [0..5] bbox (float bits), [6] total entries, [7] max tets per cell, [8] long-list counter
	uint32_t* d_big_cells = nullptr;   // worklist of cells with long tet lists (sized with d_lut_idx)
	float* d_mvc = nullptr;            // [V x n_cv] weights
	float* d_cage = nullptr;           // [n_cv x 3]
	uint32_t n_cv = 0;
	uint32_t lut_n_idx = 0, lut_max_per_cell = 0;
	// fine look-up table under the LUT (DeviceEdit::fine_*, nrs_cage.hip): rebuilt with the LUT
	uint32_t* d_fine_off = nullptr;    // [fine_cells_cap + 1]
	uint32_t* d_fine_counts = nullptr; // [fine_cells_cap]
	uint32_t* d_fine_idx = nullptr;
	uint32_t* d_fine_tiles = nullptr;  // [kFineScanTiles]
	int32_t* d_fine_win = nullptr;     // [kCascades * 6] window + [30] total entries
	size_t fine_cells_cap = 0, fine_idx_cap = 0;
	uint32_t fine_n_idx = 0;
	// A cage MOVE does not rebuild the fine table (0.3 ms of a 1.0 ms move at 6 k tets): it drops it -- the kernels scan the LUT's own lists, as before round 6 -- and
	// the second nrs_render_nerf after the last move builds it (a gizmo drag renders one frame per move and never pays; a cage at rest renders 2-3 % faster)
	bool fine_stale = false;
	uint32_t renders_since_move = 0;
};

// Marching parameters every ray-marching entry point hands to the kernels: min_mip indexes the 5-cascade bitfield (min_mip > 4 reads past
// it and makes kGrid >> mip zero, i.e. a voxel walk without progress), a negative / non-finite cone angle breaks calc_dt's clamp, a
// non-positive resolution divides by zero on the device.
static int check_params_abi(const nrs_render_params* p, const char* who) {
	// (nothing zero-extends a struct built against another header -- the size is the contract)
	if (p->struct_size != (uint32_t)sizeof(nrs_render_params))
		return fail(NRS_ERR_INVALID_ARG, std::string(who) + ": nrs_render_params.struct_size is " + std::to_string(p->struct_size) + ", this library's is " +
		            std::to_string(sizeof(nrs_render_params)) + " (ABI " + std::to_string(NRS_ABI_VERSION) + "): initialise with NRS_RENDER_PARAMS_INIT from the same nrs.h");
	return NRS_OK;
}
static int check_march_params(const nrs_render_params& p, const char* who) {
	{ const int ac = check_params_abi(&p, who); if (ac != NRS_OK) return ac; }
	if (p.resolution[0] <= 0 || p.resolution[1] <= 0 || p.resolution[0] > 65535 || p.resolution[1] > 65535)
		return fail(NRS_ERR_INVALID_ARG, std::string(who) + ": resolution out of range (1..65535)");
	if (p.min_mip > kCascades - 1) return fail(NRS_ERR_INVALID_ARG, std::string(who) + ": min_mip > 4 (the occupancy grid has 5 cascades)");
	if (!(p.cone_angle_constant >= 0.f) || !std::isfinite(p.cone_angle_constant)) return fail(NRS_ERR_INVALID_ARG, std::string(who) + ": cone_angle_constant must be finite and >= 0");
	if (!(p.min_transmittance >= 0.f && p.min_transmittance <= 1.f)) return fail(NRS_ERR_INVALID_ARG, std::string(who) + ": min_transmittance outside [0, 1]");
	for (int i = 0; i < 3; ++i)
		if (!std::isfinite(p.render_aabb_min[i]) || !std::isfinite(p.render_aabb_max[i])) return fail(NRS_ERR_INVALID_ARG, std::string(who) + ": render box is not finite");
	return NRS_OK;
}
// ---------------------------------------------------------------------------------------------------------------
// NerfNetworkFull::width(layer) / num_forward_activations (nerf_network_full.h:507-521): the hash-grid output, the density network's hidden layer, the rgb
// network's input, then one layer per rgb hidden layer.  NerfNetworkNoDir (nerf_network_nodir.h:419-438): the first two (its num_forward_activations counts two
// more, which its own forward_activations cannot serve).  0 = no such layer.
// The kernels number the layers of base.json's shape: 0 grid, 1 density hidden, 2 rgb input, 3.. rgb hidden.  A density network WITHOUT hidden layer
// (configs/nerf/linear.json) has no layer 1 in the reference's numbering: its layer k >= 1 is the kernels' layer k + 1.
// (A layer number beyond every network of the family -- e.g. 0xFFFFFFFF, which `layer + 1` would wrap to the hash grid -- maps to a layer nobody has: width 0.)
static uint32_t kernel_layer(const nrs_model_desc& d, uint32_t layer) {
	if (layer >= 8u) return 0xFFu;
	return (d.density_hidden_layers == 0u && layer >= 1u) ? layer + 1u : layer;
}
static uint32_t network_layer_width(const nrs_model_desc& d, uint32_t layer) {
	const uint32_t k = kernel_layer(d, layer);
	if (k == 0u) return 32u;
	if (k == 1u) return 64u;
	if (d.sh_degree == 0u) return 0u;
	if (k == 2u) return 32u;
	return k - 3u < d.rgb_hidden_layers ? 64u : 0u;
}
// configs/nerf/base.json's family: hash grid of 16 x 2 features (any table size: base_14 / small / base / big.json), the 64-wide density network with one hidden
// layer -- or none: one [16 x 32] matrix (CutlassMLP, linear.json) --, and an rgb network of 0 (CutlassMLP, base_0layer.json), 1, 2 (base.json) or 3 hidden layers (base_{1,2,3}layer.json) on SH degree 4 -- or none at all
// (base_nodir.json -> NerfNetworkNoDir, testbed.cu:2314-2353: sh_degree == 0).
static bool desc_supported(const nrs_model_desc& d) {
	const bool trunk = d.n_levels == 16 && d.n_features_per_level == 2 && d.n_neurons == 64 && d.density_hidden_layers <= 1 && d.density_output_dims == 16 &&
	                   d.log2_hashmap_size >= 8 && d.log2_hashmap_size <= 24 && d.base_resolution >= 1;
	if (!trunk) return false;
	if (d.sh_degree == 0) return d.rgb_hidden_layers == 0;
	return d.sh_degree == 4 && d.rgb_hidden_layers <= 3;
}
// rgb network parameters in the caller's blob (tiny-cuda-nn's layouts as recalled: the submodule is absent): FullyFusedMLP with L >= 1 hidden layers =
// [64 x 32] + (L - 1) [64 x 64] + [16 x 64] (3 outputs padded to 16 rows); CutlassMLP without hidden layer = one [8 x 32] matrix (outputs padded to 8).
static uint32_t n_rgb_weights(const nrs_model_desc& d) {
	if (d.sh_degree == 0) return 0u;
	if (d.rgb_hidden_layers == 0) return 8u * 32u;
	return 64u * 32u + (d.rgb_hidden_layers - 1u) * 64u * 64u + 16u * 64u;
}
static uint32_t n_density_weights(const nrs_model_desc& d) { return d.density_hidden_layers == 0 ? 16u * 32u : kDensityW; }
static uint32_t n_mlp_weights(const nrs_model_desc& d) { return n_density_weights(d) + n_rgb_weights(d); }

// ---- lowering of the family onto the kernels' network (kCanonW entries, nrs_internal.h) -------------------------------------------------------------
// Entries are opaque 16-bit words: fp16 bit patterns (nrs_model_set_params) or weight indices (nrs_model_set_params_device's permutation); `ops` says what
// zero, +1, -1 and a negated entry look like.  Every lowered network computes the values of the network it stands for EXACTLY, in both rounding models:
//   one hidden layer:  Wr2 = I.  The second hidden layer is relu(round(1 x h)) = h (h >= 0, an fp16 value; the other addends are zeros).
//   no hidden layer:   y = W x is formed by the FIRST layer with its own k blocks ([density outputs | SH coefficients]: the roundings of the one-matrix network),
//                      once as W and once as -W (rounding is symmetric): hidden = (relu(y), relu(-y)); Wr2 = I; the output layer subtracts the two, one of which is 0.
//   no rgb network:    the same with unit rows in place of W: (r, g, b) = density-network outputs 1..3 (NerfNetworkNoDir::inference_mixed_precision_impl,
//                      nerf_network_nodir.h:47-91).
//   three hidden layers: Wr2b, the kernels' optional layer (DeviceModel::rgb_deep).
// (A value of -0 comes out as +0: equal, not bit-identical.)
struct LowerOps { uint16_t one, minus_one; uint16_t (*negate)(uint16_t); };
static void lower_weights(const nrs_model_desc& d, const uint16_t* w, uint16_t* canon, const LowerOps& ops) {
	memset(canon, 0, kCanonW * sizeof(uint16_t));
	if (d.density_hidden_layers == 0) {
		// no hidden layer in the density network (linear.json): y = W f by the first layer as (W, -W), the output layer subtracts relu(y) and relu(-y) -- as for the
		// rgb network below.  (The density's input gradient goes through the same two layers: (y > 0) W^T 128 - (y < 0) (-W)^T 128 = W^T 128, the linear layer's
		// own backward pass, for every y but an exact 0.)
		uint16_t* Wd1 = canon;           // [64 x 32]
		uint16_t* Wd2 = canon + 64 * 32; // [16 x 64]
		for (int row = 0; row < 16; ++row) {
			for (int k = 0; k < 32; ++k) { Wd1[row * 32 + k] = w[row * 32 + k]; Wd1[(16 + row) * 32 + k] = ops.negate(w[row * 32 + k]); }
			Wd2[row * 64 + row] = ops.one;
			Wd2[row * 64 + 16 + row] = ops.minus_one;
		}
	} else {
		memcpy(canon, w, kDensityW * sizeof(uint16_t));
	}
	const uint16_t* r = w + n_density_weights(d);
	uint16_t* Wr1 = canon + kDensityW;   // [64 x 32]
	uint16_t* Wr2 = Wr1 + 64 * 32;       // [64 x 64]
	uint16_t* Wr3 = Wr2 + 64 * 64;       // [16 x 64]
	uint16_t* Wr2b = Wr3 + 16 * 64;      // [64 x 64]
	auto identity = [&](uint16_t* M) { for (int i = 0; i < 64; ++i) M[i * 64 + i] = ops.one; };
	const uint32_t L = d.rgb_hidden_layers;
	if (d.sh_degree == 0 || L == 0) {
		for (int row = 0; row < 8; ++row) {
			if (d.sh_degree == 0) {
				if (row < 3) { Wr1[row * 32 + 1 + row] = ops.one; Wr1[(8 + row) * 32 + 1 + row] = ops.minus_one; }
			} else {
				for (int k = 0; k < 32; ++k) { Wr1[row * 32 + k] = r[row * 32 + k]; Wr1[(8 + row) * 32 + k] = ops.negate(r[row * 32 + k]); }
			}
			Wr3[row * 64 + row] = ops.one;
			Wr3[row * 64 + 8 + row] = ops.minus_one;
		}
		identity(Wr2);
	} else if (L == 1) {
		memcpy(Wr1, r, 64 * 32 * 2);
		identity(Wr2);
		memcpy(Wr3, r + 64 * 32, 16 * 64 * 2);
	} else if (L == 2) {
		memcpy(Wr1, r, kRgbW * 2);
	} else {
		memcpy(Wr1, r, (64 * 32 + 64 * 64) * 2);
		memcpy(Wr2b, r + 64 * 32 + 64 * 64, 64 * 64 * 2);
		memcpy(Wr3, r + 64 * 32 + 2 * 64 * 64, 16 * 64 * 2);
	}
}
static const LowerOps kLowerValues{0x3C00, 0xBC00, [](uint16_t h) -> uint16_t { return (uint16_t)(h ^ 0x8000u); }};
static const LowerOps kLowerIndices{kFragOne, kFragMinusOne, [](uint16_t i) -> uint16_t { return (uint16_t)(i | kFragNegate); }};

// tcnn GridEncoding level geometry: scale = exp2(l*log2(b))*Nmin - 1, res = ceil(scale)+1,
// entries = min(align8(res^3), 2^log2_T).  tiny-cuda-nn evaluates the scale in FLOAT -- exp2f(level * log2f(per_level_scale)) *
// base_resolution - 1.0f, in the encoding's constructor and again in kernel_grid -- so it is float here too: evaluated in double it can
// land one ulp away, and next to an integer that changes ceil(scale) + 1 and every later level offset (a real checkpoint would be
// mis-addressed).  Host libm stands in for the device's exp2f (<= 2 ulp on NVIDIA hardware: that last bit is outside anyone's control).
static uint32_t make_levels(const nrs_model_desc& d, LevelParams* lv) {
	uint32_t off = 0;
	const float l2 = log2f(d.per_level_scale);
	for (uint32_t l = 0; l < d.n_levels; ++l) {
		LevelParams& p = lv[l];
		p.scale = exp2f((float)l * l2) * (float)d.base_resolution - 1.0f;
		p.resolution = (uint32_t)ceilf(p.scale) + 1u;
		p.res2 = p.resolution * p.resolution;
		uint64_t n = (uint64_t)p.resolution * p.resolution * p.resolution;
		n = (n + 7ull) / 8ull * 8ull;
		p.count = (uint32_t)std::min<uint64_t>(n, 1ull << d.log2_hashmap_size);
		uint64_t stride = 1;
		for (int dim = 0; dim < 3 && stride <= p.count; ++dim) stride *= p.resolution;
		p.hashed = p.count < stride ? 1u : 0u;
		p.mask = p.hashed ? p.count - 1u : 0u;
		p.offset = off;
		p.tab_first = 0;
		p.cached = p.rec_first = p.rec_res = p.rec_res2 = 0;
		off += p.count;
	}
	return off;
}

// Arrange the five row-major fp16 weight matrices (tcnn FullyFusedMLP: [out x in], no biases) as MFMA A operands
// in the order nrs_mlp.cuh consumes them.  For fragment F, lane l = (i = l & 31, g = l >> 5), element e: the weight
// of output unit (32*mb + i) for the input that the B operand's element e of lane-half g carries.
// `one` = what the constant 1.0 is written as: 0x3C00 for real weights, kFragOne when the routine runs on the identity permutation (set_params_device).
static void make_weight_fragments(const uint16_t* w, uint16_t* frag, uint16_t one = 0x3C00) {
	const uint16_t* Wd1 = w;                 // [64 x 32]
	const uint16_t* Wd2 = Wd1 + 64 * 32;     // [16 x 64]
	const uint16_t* Wr1 = Wd2 + 16 * 64;     // [64 x 32]
	const uint16_t* Wr2 = Wr1 + 64 * 32;     // [64 x 64]
	const uint16_t* Wr3 = Wr2 + 64 * 64;     // [16 x 64]
	const uint16_t* Wr2b = Wr3 + 16 * 64;    // [64 x 64] (kCanonW: lower_weights' layout)
	auto hidden_row = [](int mb, int g, int r) { return 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * g; }; // D-tile row of reg r
	auto at = [&](int f, int lane, int e) -> uint16_t& { return frag[((size_t)f * 64 + lane) * 8 + e]; };
	memset(frag, 0, kWfragDeviceBytes);
	for (int lane = 0; lane < 64; ++lane) {
		const int i = lane & 31, g = lane >> 5;
		for (int e = 0; e < 8; ++e) {
			// Sel0 / Sel1: row i of the product picks the packed accumulator that came out of D register e (Sel0) / 8 + e (Sel1) of lane-half g
			at(24, lane, e) = i == hidden_row(0, g, e) ? one : (uint16_t)0;
			at(25, lane, e) = i == hidden_row(0, g, 8 + e) ? one : (uint16_t)0;
			// Bwd[ks]: dL/dfeatures[i] = sum_k W1[k][i] dL/dhidden[k] (density MLP, input gradient): output row i = feature i, the B operand of k step ks is
			// the packed dL/dhidden in the layout the hidden layer's D tiles come out in (as for D2)
			for (int ks = 0; ks < 4; ++ks) at(26 + ks, lane, e) = Wd1[hidden_row(ks >> 1, g, 8 * (ks & 1) + e) * 32 + i];
			for (int mb = 0; mb < 2; ++mb)
				for (int ks = 0; ks < 2; ++ks) {
					const int feat = 2 * (2 * (4 * ks + (e >> 1)) + g) + (e & 1); // level 2*it+g, it = 4ks + e/2
					at(mb * 2 + ks, lane, e) = Wd1[(32 * mb + i) * 32 + feat];
				}
			for (int ks = 0; ks < 4; ++ks) {
				const int k = hidden_row(ks >> 1, g, 8 * (ks & 1) + e);
				if (i < 16) at(4 + ks, lane, e) = Wd2[i * 64 + k];
				if (i < 16) at(20 + ks, lane, e) = Wr3[i * 64 + k];
				for (int mb = 0; mb < 2; ++mb) at(12 + mb * 4 + ks, lane, e) = Wr2[(32 * mb + i) * 64 + k];
				for (int mb = 0; mb < 2; ++mb) at(30 + mb * 4 + ks, lane, e) = Wr2b[(32 * mb + i) * 64 + k];
			}
			for (int mb = 0; mb < 2; ++mb) {
				const int kd = (e & 3) + 8 * (e >> 2) + 4 * g; // density-output row in element e
				at(8 + mb * 2 + 0, lane, e) = Wr1[(32 * mb + i) * 32 + kd];
				at(8 + mb * 2 + 1, lane, e) = Wr1[(32 * mb + i) * 32 + 16 + 8 * g + e]; // SH coefficient 8g+e
			}
		}
	}
}

static void box_of(const float* v, uint32_t n, Box3& b) {
	const float inf = std::numeric_limits<float>::infinity();
	for (int k = 0; k < 3; ++k) { b.mn[k] = inf; b.mx[k] = -inf; }
	for (uint32_t i = 0; i < n; ++i)
		for (int k = 0; k < 3; ++k) {
			b.mn[k] = std::fmin(b.mn[k], v[3 * i + k]);
			b.mx[k] = std::fmax(b.mx[k], v[3 * i + k]);
		}
}
static void warp_box(const Box3& b, const Box3& aabb, Box3& out) { // BoundingBox::warp_box, bounding_box.cuh:272
	for (int k = 0; k < 3; ++k) {
		const float diag = aabb.mx[k] - aabb.mn[k];
		out.mn[k] = (b.mn[k] - aabb.mn[k]) / diag;
		out.mx[k] = (b.mx[k] - aabb.mn[k]) / diag;
	}
}

// DeviceModel as one render / trace launch sees it: the marching accelerator that matches the launch's step parameters
static DeviceModel model_for_launch(const nrs_model* m, const nrs_render_params& p);

template <typename T>
static int upload(nrs_edit* e, const T* h, size_t count, const T** d_out) {
	void* d = nullptr;
	HIP_TRY(hipMalloc(&d, std::max<size_t>(count * sizeof(T), 16)));
	e->allocs.push_back(d);
	if (count) HIP_TRY(hipMemcpy(d, h, count * sizeof(T), hipMemcpyHostToDevice));
	*d_out = (const T*)d;
	return NRS_OK;
}

// Both flavours of the marching accelerator from m->d_bitfield (launch_occ_accel: bounds, box and look-ahead masks are built on the device);
// the host reads back the 2 x 12 floats a launch carries in its kernel arguments, and synchronises the stream for that.
static int refresh_accel(nrs_model* m, void* stream) {
	hipStream_t s = (hipStream_t)stream;
	float* d_out = reinterpret_cast<float*>(m->d_accel_masks + 2 * kCoarseWords);
	NRS_TRY(launch_occ_accel(m->dm.bitfield, m->d_accel_masks, d_out, m->d_accel_masks + 2 * kCoarseWords + 24, stream));
	float h[24];
	HIP_TRY(hipMemcpyAsync(h, d_out, sizeof(h), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	OccAccel* acc[2] = {&m->accel_any, &m->accel_exact};
	for (int f = 0; f < 2; ++f)
		for (int k = 0; k < 3; ++k) {
			acc[f]->box.mn[k] = h[f * 12 + k];
			acc[f]->box.mx[k] = h[f * 12 + 3 + k];
			acc[f]->cell[k] = h[f * 12 + 6 + k];
			acc[f]->inv_cell[k] = h[f * 12 + 9 + k];
		}
	m->accel_any.mask = m->d_accel_masks;
	m->accel_exact.mask = m->d_accel_masks + kCoarseWords;
	return NRS_OK;
}
static DeviceModel model_for_launch(const nrs_model* m, const nrs_render_params& p) {
	DeviceModel dm = m->dm;
	dm.occ = (p.cone_angle_constant == 0.f && p.min_mip == 0) ? m->accel_exact : m->accel_any;
	return dm;
}

// ---------------------------------------------------------------------------------------------------------------
extern "C" {

const char* nrs_last_error(void) { return g_err.c_str(); }
int nrs_abi_version(void) { return NRS_ABI_VERSION; }

int nrs_ctx_create(int device, nrs_ctx** out) {
	if (!out) return fail(NRS_ERR_INVALID_ARG, "nrs_ctx_create: out is NULL");
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) return fail(NRS_ERR_NO_DEVICE, "nrs_ctx_create: no HIP device visible (this library has no CPU fallback)");
	if (device < 0 || device >= n) return fail(NRS_ERR_INVALID_ARG, "nrs_ctx_create: device index out of range");
	HIP_TRY(hipSetDevice(device));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	nrs_ctx* c = new (std::nothrow) nrs_ctx();
	if (!c) return fail(NRS_ERR_STATE, "out of host memory");
	c->device = device;
	c->n_cus = prop.multiProcessorCount;
	c->hbm_bytes = prop.totalGlobalMem;
	snprintf(c->name, sizeof(c->name), "%s (%s)", prop.name, prop.gcnArchName);
	hipError_t he = hipMalloc((void**)&c->d_counters, sizeof(RenderCounters) * nrs_ctx::kInFlight * 2);
	if (he == hipSuccess) he = hipMemset(c->d_counters, 0, sizeof(RenderCounters) * nrs_ctx::kInFlight * 2);
	for (int i = 0; i < nrs_ctx::kInFlight; ++i) c->counters_clean[i] = he == hipSuccess;
	if (he == hipSuccess) he = hipMalloc((void**)&c->d_edits, sizeof(DeviceEdit) * nrs_ctx::kMaxEdits * (nrs_ctx::kInFlight + 1));
	for (int i = 0; i < nrs_ctx::kInFlight && he == hipSuccess; ++i) he = hipEventCreateWithFlags(&c->slot_done[i], hipEventDisableTiming);
	if (he == hipSuccess) he = hipMalloc((void**)&c->d_mean, 8 + 256 * 8); // mean + partial sums (launch_grid_to_bitfield)
	if (he == hipSuccess) he = hipMalloc((void**)&c->d_wave_log, 8192 * 4 * 8);
	if (he != hipSuccess) {
		nrs_ctx_destroy(c);
		return fail_hip(he, "nrs_ctx_create: hipMalloc");
	}
	// one device-visible host word for the launch feedback (lane-team sizing); optional: without it the estimate stays at its default
	if (hipHostMalloc((void**)&c->h_feedback, 8, hipHostMallocMapped) == hipSuccess) {
		*c->h_feedback = 0ull;
		if (hipHostGetDevicePointer((void**)&c->d_feedback, c->h_feedback, 0) != hipSuccess) c->d_feedback = nullptr;
	} else {
		(void)hipGetLastError();
		c->h_feedback = nullptr;
	}
	*out = c;
	return NRS_OK;
}
void nrs_ctx_destroy(nrs_ctx* c) {
	if (!c) return;
	(void)hipFree(c->d_counters);
	(void)hipFree(c->d_edits);
	for (int i = 0; i < nrs_ctx::kInFlight; ++i)
		if (c->slot_done[i]) (void)hipEventDestroy(c->slot_done[i]);
	(void)hipFree(c->d_mean);
	(void)hipFree(c->d_wave_log);
	if (c->h_feedback) (void)hipHostFree(c->h_feedback);
	delete c;
}
int nrs_ctx_set_lane_teams(nrs_ctx* ctx, int lanes_per_ray) {
	if (!ctx || !(lanes_per_ray == -4 || lanes_per_ray == -3 || lanes_per_ray == -2 || lanes_per_ray == -1 || lanes_per_ray == 0 || lanes_per_ray == 1 || lanes_per_ray == 2 || lanes_per_ray == 4))
		return fail(NRS_ERR_INVALID_ARG, "nrs_ctx_set_lane_teams: lanes_per_ray must be 0 (automatic), 1, 2, 4, -1 (hybrid), -2, -3 or -4 (small-launch schedule: teams sized per generation, 4x4- / 8x4- / 8x8-pixel packets)");
	ctx->lane_teams = lanes_per_ray;
	return NRS_OK;
}
int nrs_ctx_set_ray_handover(nrs_ctx* ctx, int enabled) {
	if (!ctx) return fail(NRS_ERR_INVALID_ARG, "ctx is NULL");
	ctx->handover = enabled ? 1 : 0;
	return NRS_OK;
}
int nrs_ctx_ray_handovers(const nrs_ctx* ctx, uint64_t* n_rays, uint64_t* n_handovers) {
	if (!ctx) return fail(NRS_ERR_INVALID_ARG, "ctx is NULL");
	if (n_rays) *n_rays = ctx->last_handover & 0xffffffffull;
	if (n_handovers) *n_handovers = ctx->last_handover >> 32;
	return NRS_OK;
}
int nrs_ctx_device_info(const nrs_ctx* c, char* name_out, size_t name_len, int* n_cus, size_t* hbm_bytes) {
	if (!c) return fail(NRS_ERR_INVALID_ARG, "ctx is NULL");
	if (name_out && name_len) snprintf(name_out, name_len, "%s", c->name);
	if (n_cus) *n_cus = c->n_cus;
	if (hbm_bytes) *hbm_bytes = c->hbm_bytes;
	return NRS_OK;
}

size_t nrs_model_n_params(const nrs_model_desc* d) {
	if (!d || !desc_supported(*d)) return 0;
	LevelParams lv[kLevels];
	return (size_t)n_mlp_weights(*d) + (size_t)make_levels(*d, lv) * 2;
}
int nrs_model_level_table(const nrs_model_desc* d, float* scale, uint32_t* resolution, uint32_t* entry_offset, uint32_t* entry_count,
                          uint32_t* hashed) {
	if (!d || !desc_supported(*d)) return fail(NRS_ERR_UNSUPPORTED, "model description outside configs/nerf/base.json's family (hash grid 16 x 2, 64-wide density network of 0..1 hidden layers, rgb network of 0..3 hidden layers or none)");
	LevelParams lv[kLevels];
	make_levels(*d, lv);
	for (uint32_t l = 0; l < d->n_levels; ++l) {
		if (scale) scale[l] = lv[l].scale;
		if (resolution) resolution[l] = lv[l].resolution;
		if (entry_offset) entry_offset[l] = lv[l].offset;
		if (entry_count) entry_count[l] = lv[l].count;
		if (hashed) hashed[l] = lv[l].hashed;
	}
	return NRS_OK;
}

int nrs_model_create(nrs_ctx* ctx, const nrs_model_desc* desc, nrs_model** out) {
	if (!ctx || !desc || !out) return fail(NRS_ERR_INVALID_ARG, "nrs_model_create: NULL argument");
	if (!desc_supported(*desc)) return fail(NRS_ERR_UNSUPPORTED, "model description outside configs/nerf/base.json's family (hash grid 16 x 2, 64-wide density network of 0..1 hidden layers, rgb network of 0..3 hidden layers or none)");
	for (int k = 0; k < 3; ++k)
		if (!(desc->aabb_max[k] > desc->aabb_min[k])) return fail(NRS_ERR_INVALID_ARG, "nrs_model_create: empty aabb");
	HIP_TRY(hipSetDevice(ctx->device));
	nrs_model* m = new (std::nothrow) nrs_model();
	if (!m) return fail(NRS_ERR_STATE, "out of host memory");
	m->ctx = ctx;
	m->desc = *desc;
	m->total_entries = make_levels(*desc, m->dm.levels);
	for (int k = 0; k < 3; ++k) { m->dm.aabb.mn[k] = desc->aabb_min[k]; m->dm.aabb.mx[k] = desc->aabb_max[k]; }
	m->dm.diag_pow2 = 1;
	for (int k = 0; k < 3; ++k) {
		const float diag = desc->aabb_max[k] - desc->aabb_min[k];
		int e = 0;
		if (std::frexp(diag, &e) != 0.5f) m->dm.diag_pow2 = 0;
		m->dm.inv_diag[k] = 1.0f / diag;
	}
	m->dm.rgb_deep = desc->sh_degree != 0 && desc->rgb_hidden_layers == 3 ? 1u : 0u;
	m->dm.no_dir = desc->sh_degree == 0 ? 1u : 0u;
	m->dm.rgb_activation = desc->rgb_activation;
	m->dm.density_activation = desc->density_activation;
	hipError_t he = hipMalloc((void**)&m->d_grid, (size_t)m->total_entries * 4);
	if (he == hipSuccess) he = hipMalloc((void**)&m->d_wfrag, kWfragDeviceBytes);
	if (he == hipSuccess) he = hipMalloc((void**)&m->d_bitfield, NRS_BITFIELD_BYTES);
	if (he == hipSuccess) he = hipMalloc((void**)&m->d_accel_masks, 2 * kCoarseWords * 4 + 256); // + 24 floats of OccAccel numbers + 12 words of scratch (refresh_accel)
	if (he == hipSuccess) he = hipMemset(m->d_accel_masks, 0, 2 * kCoarseWords * 4);
	if (he == hipSuccess) he = hipMalloc((void**)&m->d_density_grid, (size_t)kGridVol * kCascades * 4);
	if (he == hipSuccess) he = hipMemset(m->d_density_grid, 0, (size_t)kGridVol * kCascades * 4);
	if (he != hipSuccess) {
		nrs_model_destroy(m);
		return fail_hip(he, "nrs_model_create: device allocation");
	}
	m->dm.grid = m->d_grid;
	m->dm.wfrag = m->d_wfrag;
	m->dm.bitfield = m->d_bitfield;
	// The cell-record cache is OPT-IN since round 6 (a 24 MB model does not reserve gigabytes unasked): nrs_model_set_cell_cache(model, budget), or NRS_CELL_CACHE_GB in
	// the environment for a host that cannot be changed -- never more than a quarter of the free HBM.
	if (const char* e = getenv("NRS_CELL_CACHE_GB")) {
		size_t budget = (size_t)(atof(e) * 1073741824.0), free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, free_b / 4);
		if (nrs_model_set_cell_cache(m, budget) != NRS_OK) (void)nrs_model_set_cell_cache(m, 0); // an optimisation: render without it
	}
	*out = m;
	return NRS_OK;
}
void nrs_model_destroy(nrs_model* m) {
	if (!m) return;
	(void)hipFree(m->d_grid);
	(void)hipFree(m->d_wfrag);
	(void)hipFree(m->d_wfrag_src);
	(void)hipFree(m->d_bitfield);
	(void)hipFree(m->d_accel_masks);
	(void)hipFree(m->d_density_grid);
	(void)hipFree(m->d_density_tmp);
	(void)hipFree(m->d_records);
	(void)hipFree(m->d_bricks);
	(void)hipFree(m->d_slots);
	(void)hipFree(m->d_records2);
	delete m;
}
static void drop_sparse_cell_cache(nrs_model* m);
// Cell-record cache: plan (how many levels fit the budget), allocate, build.  Levels are cached from the coarsest up, an
// even number of them (the kernels evaluate levels in pairs), and their records share one allocation.
static uint32_t plan_cell_cache(const LevelParams* lv, size_t budget, LevelParams* out, size_t* bytes) {
	uint32_t n = 0;
	uint64_t records = 0, fit = 0;
	for (uint32_t l = 0; l < kLevels; ++l) {
		const uint64_t cells = (uint64_t)lv[l].resolution * lv[l].resolution * lv[l].resolution;
		if ((records + cells) * 32ull > budget || records + cells >= (1ull << 32) || lv[l].resolution >= 4096u) break; // (res < 4096: the gather's 24-bit index products, nrs_mlp.cuh mul24)
		records += cells;
		if (l & 1u) { n = l + 1; fit = records; }
	}
	uint64_t first = 0;
	for (uint32_t l = 0; l < kLevels; ++l) {
		out[l] = lv[l];
		out[l].cached = l < n ? 1u : 0u;
		out[l].rec_first = l < n ? (uint32_t)first : 0u;
		out[l].rec_res = l < n ? lv[l].resolution : 0u;
		out[l].rec_res2 = out[l].rec_res * out[l].rec_res;
		if (l < n) first += (uint64_t)lv[l].resolution * lv[l].resolution * lv[l].resolution;
	}
	*bytes = (size_t)fit * 32;
	return n;
}
static int rebuild_cell_cache(nrs_model* m, void* stream = nullptr, bool sync = true) {
	if (!m->have_params) return NRS_OK;
	if (m->cached_levels) NRS_TRY(launch_cell_records(m->dm, m->cached_levels, m->d_records, stream));
	for (uint32_t l = m->sparse_first; l < m->sparse_first + m->sparse_levels; ++l)
		NRS_TRY(launch_brick_fill(m->dm, m->dm.levels[l], m->d_slots + m->slot_first[l], m->slot_count[l], m->d_records2, stream));
	if (sync) HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
	return NRS_OK;
}
static void drop_sparse_cell_cache(nrs_model* m) {
	(void)hipFree(m->d_bricks); (void)hipFree(m->d_slots); (void)hipFree(m->d_records2);
	m->d_bricks = nullptr; m->d_slots = nullptr; m->d_records2 = nullptr;
	m->sparse_bytes = 0; m->sparse_first = m->sparse_levels = 0;
	for (uint32_t l = 0; l < kLevels; ++l)
		if (m->dm.levels[l].cached == 2u) { m->dm.levels[l].cached = 0; m->dm.levels[l].rec_first = m->dm.levels[l].rec_res = m->dm.levels[l].rec_res2 = m->dm.levels[l].tab_first = 0; }
	m->dm.records2 = nullptr; m->dm.bricks = nullptr;
}
// Sparse brick records for the levels after the dense ones, in pairs, while tables + records fit the budget.
int nrs_model_set_sparse_cell_cache(nrs_model* m, const uint8_t* h_mask_bitfield, size_t max_bytes) {
	if (!m) return fail(NRS_ERR_INVALID_ARG, "nrs_model_set_sparse_cell_cache: NULL model");
	HIP_TRY(hipSetDevice(m->ctx->device));
	HIP_TRY(hipDeviceSynchronize()); // launches in flight may still read the old records
	drop_sparse_cell_cache(m);
	if (!h_mask_bitfield || !max_bytes) return NRS_OK;
	const uint32_t first = m->cached_levels;
	uint8_t* d_mask = nullptr;
	uint32_t* d_counter = nullptr;
	uint32_t* d_tmp_table = nullptr;
	auto cleanup = [&]() { (void)hipFree(d_mask); (void)hipFree(d_counter); (void)hipFree(d_tmp_table); d_mask = nullptr; d_counter = nullptr; d_tmp_table = nullptr; };
	auto bail = [&](int rc) { cleanup(); drop_sparse_cell_cache(m); return rc; };
	hipError_t he = hipMalloc((void**)&d_mask, NRS_BITFIELD_BYTES);
	if (he == hipSuccess) he = hipMemcpy(d_mask, h_mask_bitfield, NRS_BITFIELD_BYTES, hipMemcpyHostToDevice);
	if (he == hipSuccess) he = hipMalloc((void**)&d_counter, 4 * kLevels);
	if (he == hipSuccess) he = hipMemset(d_counter, 0, 4 * kLevels);
	if (he != hipSuccess) { (void)hipGetLastError(); return bail(fail(NRS_ERR_HIP, "nrs_model_set_sparse_cell_cache: out of device memory")); }
	// pass A: level by level, mark into a scratch table to COUNT the bricks the mask asks for (the set is deterministic, only the slot order is
	// not); accept level pairs while tables + slot lists + records fit the budget
	LevelParams lv[kLevels];
	uint32_t counts[kLevels] = {}, nbs[kLevels] = {};
	uint64_t used = 0, table_total = 0, bricks_total = 0;
	uint32_t n_ok = 0;
	for (uint32_t l = first; l + 1 < kLevels; l += 2) {
		uint64_t pair_bytes = 0, pair_tables = 0, pair_bricks = 0;
		bool ok = true;
		for (uint32_t k = l; k < l + 2 && ok; ++k) {
			const uint32_t nb = (m->dm.levels[k].resolution + kBrick - 1) / kBrick;
			const uint64_t entries = (uint64_t)nb * nb * nb;
			if (entries * 4ull > max_bytes - std::min<uint64_t>(max_bytes, used + pair_bytes) || table_total + pair_tables + entries >= (1ull << 32)) { ok = false; break; }
			if (hipMalloc((void**)&d_tmp_table, entries * 4ull) != hipSuccess || hipMemset(d_tmp_table, 0, entries * 4ull) != hipSuccess) { (void)hipGetLastError(); ok = false; break; }
			lv[k] = m->dm.levels[k];
			lv[k].rec_res = nb; lv[k].rec_res2 = nb * nb; lv[k].tab_first = 0; lv[k].rec_first = 0;
			const int rc = launch_brick_mark(m->dm, lv[k], d_mask, d_tmp_table, d_counter + k, nullptr, 0, nullptr);
			if (rc != NRS_OK) { g_err = launch_last_error(); return bail(rc); }
			if (hipMemcpy(&counts[k], d_counter + k, 4, hipMemcpyDeviceToHost) != hipSuccess) return bail(fail(NRS_ERR_HIP, "nrs_model_set_sparse_cell_cache: read-back"));
			(void)hipFree(d_tmp_table); d_tmp_table = nullptr;
			nbs[k] = nb;
			pair_tables += entries; pair_bricks += counts[k];
			pair_bytes += entries * 4ull + (uint64_t)counts[k] * ((uint64_t)kBrickCells * 32ull + 4ull);
			if (dev_knob("NRS_SPARSE_LOG"))
				fprintf(stderr, "[nrs sparse] level %u: res %u, %u^3 bricks (table %.1f MB), %u bricks marked = %.2f GB of records\n", k, m->dm.levels[k].resolution, nb, entries * 4e-6,
				        counts[k], counts[k] * 16384e-9);
		}
		if (!ok || used + pair_bytes > max_bytes || (bricks_total + pair_bricks) * kBrickCells >= (1ull << 32)) break;
		used += pair_bytes; table_total += pair_tables; bricks_total += pair_bricks;
		n_ok += 2;
	}
	if (!n_ok || !bricks_total) { cleanup(); drop_sparse_cell_cache(m); return NRS_OK; }
	// pass B: the real tables, slot lists and records of the accepted levels
	he = hipMalloc((void**)&m->d_bricks, table_total * 4ull);
	if (he == hipSuccess) he = hipMemset(m->d_bricks, 0, table_total * 4ull);
	if (he == hipSuccess) he = hipMalloc((void**)&m->d_slots, bricks_total * 4ull);
	if (he == hipSuccess) he = hipMalloc((void**)&m->d_records2, bricks_total * kBrickCells * 32ull);
	if (he == hipSuccess) he = hipMemset(d_counter, 0, 4 * kLevels);
	if (he != hipSuccess) { (void)hipGetLastError(); return bail(fail(NRS_ERR_HIP, "nrs_model_set_sparse_cell_cache: out of device memory (records)")); }
	uint64_t tab = 0, slot0 = 0;
	for (uint32_t l = first; l < first + n_ok; ++l) {
		lv[l].tab_first = (uint32_t)tab;
		lv[l].rec_first = (uint32_t)(slot0 * kBrickCells);
		const int rc = launch_brick_mark(m->dm, lv[l], d_mask, m->d_bricks + tab, d_counter + l, m->d_slots + slot0, counts[l], nullptr);
		if (rc != NRS_OK) { g_err = launch_last_error(); return bail(rc); }
		m->slot_first[l] = (uint32_t)slot0; m->slot_count[l] = counts[l];
		lv[l].cached = 2u;
		m->dm.levels[l] = lv[l];
		tab += (uint64_t)nbs[l] * nbs[l] * nbs[l];
		slot0 += counts[l];
	}
	m->sparse_first = first; m->sparse_levels = n_ok;
	m->sparse_bytes = (size_t)used;
	m->dm.records2 = m->d_records2; m->dm.bricks = m->d_bricks;
	cleanup();
	return rebuild_cell_cache(m);
}
size_t nrs_model_sparse_cell_cache_bytes(const nrs_model* m, uint32_t* first_level, uint32_t* n_levels) {
	if (!m) return 0;
	if (first_level) *first_level = m->sparse_first;
	if (n_levels) *n_levels = m->sparse_levels;
	return m->sparse_bytes;
}
int nrs_model_set_cell_cache(nrs_model* m, size_t max_bytes) {
	if (!m) return fail(NRS_ERR_INVALID_ARG, "nrs_model_set_cell_cache: NULL model");
	HIP_TRY(hipSetDevice(m->ctx->device));
	HIP_TRY(hipDeviceSynchronize()); // launches in flight may still read the old records
	drop_sparse_cell_cache(m); // they start where the dense levels end: set them again afterwards
	LevelParams lv[kLevels];
	size_t bytes = 0;
	const uint32_t n = plan_cell_cache(m->dm.levels, max_bytes, lv, &bytes);
	if (bytes != m->records_bytes) {
		(void)hipFree(m->d_records);
		m->d_records = nullptr;
		m->records_bytes = 0;
		m->cached_levels = 0;
		for (uint32_t l = 0; l < kLevels; ++l) { m->dm.levels[l].cached = 0; }
		m->dm.records = nullptr;
		if (bytes && hipMalloc((void**)&m->d_records, bytes) != hipSuccess) {
			(void)hipGetLastError();
			return fail(NRS_ERR_HIP, "nrs_model_set_cell_cache: out of device memory for the cell records");
		}
		m->records_bytes = bytes;
	}
	m->cell_cache_budget = max_bytes;
	m->cached_levels = n;
	for (uint32_t l = 0; l < kLevels; ++l) m->dm.levels[l] = lv[l];
	m->dm.records = m->d_records;
	return rebuild_cell_cache(m);
}
size_t nrs_model_cell_cache_bytes(const nrs_model* m, uint32_t* n_levels) {
	if (!m) return 0;
	if (n_levels) *n_levels = m->cached_levels;
	return m->records_bytes;
}

int nrs_model_set_params(nrs_model* m, const void* h_params_fp16, size_t n_params) {
	if (!m || !h_params_fp16) return fail(NRS_ERR_INVALID_ARG, "nrs_model_set_params: NULL argument");
	const uint32_t n_mlp = n_mlp_weights(m->desc);
	const size_t expect = (size_t)n_mlp + (size_t)m->total_entries * 2;
	if (n_params != expect) {
		char buf[160];
		snprintf(buf, sizeof(buf), "nrs_model_set_params: got %zu params, the description implies %zu", n_params, expect);
		return fail(NRS_ERR_INVALID_ARG, buf);
	}
	HIP_TRY(hipSetDevice(m->ctx->device));
	const uint16_t* w = (const uint16_t*)h_params_fp16;
	std::vector<uint16_t> canon(kCanonW), frag(kWfragDeviceBytes / 2);
	lower_weights(m->desc, w, canon.data(), kLowerValues);
	make_weight_fragments(canon.data(), frag.data());
	HIP_TRY(hipMemcpy(m->d_wfrag, frag.data(), kWfragDeviceBytes, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(m->d_grid, w + n_mlp, (size_t)m->total_entries * 4, hipMemcpyHostToDevice));
	m->have_params = true;
	return rebuild_cell_cache(m);
}
// NerfNetworkFull::set_params hands over DEVICE pointers into the trainer's parameter blob (nerf_network_full.h:316-349): the same for a caller
// whose parameters already live on the device (a viewer that trains while it renders, src/testbed.cu:2502).  Everything is enqueued on `stream` and
// nothing waits: the hash grid (24-27 MB) is copied device-to-device (~10 us), the 20 KB of MLP weights are re-arranged into MFMA fragments by a
// small kernel (the permutation is make_weight_fragments', uploaded once per model), the cell records -- if the caller keeps any -- are rebuilt
// behind them.  Renders enqueued on the same stream afterwards see the new parameters; the blob may be overwritten once the stream has passed
// this call.  Copy semantics: call it again after every optimiser step (the reference's renderer reads the blob in place; ours is a transformed copy).
int nrs_model_set_params_device(nrs_model* m, const void* d_params_fp16, size_t n_params, void* stream) {
	if (!m || !d_params_fp16) return fail(NRS_ERR_INVALID_ARG, "nrs_model_set_params_device: NULL argument");
	const uint32_t n_mlp = n_mlp_weights(m->desc);
	const size_t expect = (size_t)n_mlp + (size_t)m->total_entries * 2;
	if (n_params != expect) {
		char buf[160];
		snprintf(buf, sizeof(buf), "nrs_model_set_params_device: got %zu params, the description implies %zu", n_params, expect);
		return fail(NRS_ERR_INVALID_ARG, buf);
	}
	HIP_TRY(hipSetDevice(m->ctx->device));
	hipStream_t s = (hipStream_t)stream;
	const uint16_t* d = (const uint16_t*)d_params_fp16;
	if (!m->d_wfrag_src) { // the fragment permutation as indices: run the host routine on the identity (index + 1; 0 stays "padding")
		static_assert(kDensityW + 64 * 32 + 2 * 64 * 64 + 16 * 64 < kFragNegate, "weight indices + 1 fit 15 bits");
		std::vector<uint16_t> ident(n_mlp), canon(kCanonW), src(kWfragDeviceBytes / 2);
		for (size_t i = 0; i < ident.size(); ++i) ident[i] = (uint16_t)(i + 1);
		lower_weights(m->desc, ident.data(), canon.data(), kLowerIndices);
		make_weight_fragments(canon.data(), src.data(), kFragOne);
		HIP_TRY(hipMalloc((void**)&m->d_wfrag_src, kWfragDeviceBytes));
		const hipError_t up = hipMemcpy(m->d_wfrag_src, src.data(), kWfragDeviceBytes, hipMemcpyHostToDevice);
		if (up != hipSuccess) { // never keep a permutation that was not uploaded: later calls would scramble the weights silently
			(void)hipFree(m->d_wfrag_src);
			m->d_wfrag_src = nullptr;
			return fail_hip(up, "nrs_model_set_params_device: upload of the weight permutation");
		}
	}
	NRS_TRY(launch_weight_fragments(d, m->d_wfrag_src, (uint16_t*)m->d_wfrag, kWfragDeviceBytes / 2, stream));
	HIP_TRY(hipMemcpyAsync(m->d_grid, d + n_mlp, (size_t)m->total_entries * 4, hipMemcpyDeviceToDevice, s));
	m->have_params = true;
	return rebuild_cell_cache(m, stream, false);
}
int nrs_model_set_numerics(nrs_model* m, uint32_t grid_acc, uint32_t mlp_acc) {
	if (!m) return fail(NRS_ERR_INVALID_ARG, "nrs_model_set_numerics: NULL model");
	if (grid_acc > NRS_GRID_ACC_NETWORK || mlp_acc > NRS_MLP_ACC_FP16) return fail(NRS_ERR_INVALID_ARG, "nrs_model_set_numerics: unknown mode");
	m->dm.numerics = (grid_acc == NRS_GRID_ACC_NETWORK ? 1u : 0u) | (mlp_acc == NRS_MLP_ACC_FP16 ? 2u : 0u);
	return NRS_OK;
}
int nrs_model_set_density_bitfield(nrs_model* m, const uint8_t* h_bitfield, size_t n_bytes) {
	if (!m || !h_bitfield) return fail(NRS_ERR_INVALID_ARG, "nrs_model_set_density_bitfield: NULL argument");
	if (n_bytes != NRS_BITFIELD_BYTES) return fail(NRS_ERR_INVALID_ARG, "nrs_model_set_density_bitfield: expected 5*128^3/8 bytes");
	HIP_TRY(hipSetDevice(m->ctx->device));
	HIP_TRY(hipMemcpy(m->d_bitfield, h_bitfield, n_bytes, hipMemcpyHostToDevice));
	NRS_TRY(refresh_accel(m, nullptr));
	m->have_bitfield = true;
	return NRS_OK;
}
// bitfield + mips from m->d_density_grid, then the marching shortcut's bounds and masks (all on the device; synchronises for 96 bytes)
static int refresh_bitfield(nrs_model* m, void* stream) {
	hipStream_t s = (hipStream_t)stream;
	HIP_TRY(hipMemsetAsync(m->d_bitfield, 0, NRS_BITFIELD_BYTES, s));
	NRS_TRY(launch_grid_to_bitfield(m->d_density_grid, m->d_bitfield, m->ctx->d_mean, stream));
	NRS_TRY(refresh_accel(m, stream));
	m->have_bitfield = true;
	return NRS_OK;
}
int nrs_model_set_density_grid(nrs_model* m, const float* h_grid, size_t n_floats) {
	if (!m || !h_grid) return fail(NRS_ERR_INVALID_ARG, "nrs_model_set_density_grid: NULL argument");
	if (n_floats != (size_t)kGridVol * kCascades) return fail(NRS_ERR_INVALID_ARG, "nrs_model_set_density_grid: expected 5*128^3 floats");
	HIP_TRY(hipSetDevice(m->ctx->device));
	HIP_TRY(hipMemcpy(m->d_density_grid, h_grid, n_floats * 4, hipMemcpyHostToDevice));
	return refresh_bitfield(m, nullptr);
}
int nrs_model_get_density_grid(nrs_model* m, float* h_out, size_t n_floats) {
	if (!m || !h_out || n_floats != (size_t)kGridVol * kCascades) return fail(NRS_ERR_INVALID_ARG, "nrs_model_get_density_grid: bad argument");
	HIP_TRY(hipSetDevice(m->ctx->device));
	HIP_TRY(hipMemcpy(h_out, m->d_density_grid, n_floats * 4, hipMemcpyDeviceToHost));
	return NRS_OK;
}

// tcnn::pcg32 on the host: only seeding and the skip-ahead the refresh needs
static const uint64_t kPcgMult = 0x5851f42d4c957f2dULL;
static uint64_t pcg_advance(uint64_t state, uint64_t inc, uint64_t delta) {
	uint64_t cur_mult = kPcgMult, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
	while (delta > 0) {
		if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
		cur_plus = (cur_mult + 1) * cur_plus;
		cur_mult *= cur_mult;
		delta >>= 1;
	}
	return acc_mult * state + acc_plus;
}
void nrs_rng_seed(uint64_t seed, uint64_t* state_out, uint64_t* inc_out) {
	const uint64_t inc = (1u << 1u) | 1u; // initseq = 1
	uint64_t state = 0u;
	state = state * kPcgMult + inc;
	state += seed;
	state = state * kPcgMult + inc;
	if (state_out) *state_out = state;
	if (inc_out) *inc_out = inc;
}

int nrs_model_update_density_grid(nrs_model* m, nrs_edit* const* edits, int n_edits, nrs_grid_update* u, void* stream) {
	if (!m || !u) return fail(NRS_ERR_INVALID_ARG, "nrs_model_update_density_grid: NULL argument");
	if (!m->have_params) return fail(NRS_ERR_STATE, "nrs_model_update_density_grid: parameters not set (nrs_model_set_params)");
	if (u->max_cascade >= kCascades) return fail(NRS_ERR_INVALID_ARG, "nrs_model_update_density_grid: max_cascade must be < 5");
	if (n_edits < 0 || n_edits > nrs_ctx::kMaxEdits) return fail(NRS_ERR_INVALID_ARG, "nrs_model_update_density_grid: too many edit operators");
	if (n_edits > 0 && !edits) return fail(NRS_ERR_INVALID_ARG, "nrs_model_update_density_grid: edits is NULL");
	if ((uint64_t)u->n_uniform_samples + u->n_nonuniform_samples > 0x40000000ull)
		return fail(NRS_ERR_INVALID_ARG, "nrs_model_update_density_grid: more than 2^30 samples");
	nrs_ctx* ctx = m->ctx;
	HIP_TRY(hipSetDevice(ctx->device));
	hipStream_t s = (hipStream_t)stream;
	const size_t grid_bytes = (size_t)kGridVol * kCascades * 4;
	if (!m->d_density_tmp) HIP_TRY(hipMalloc((void**)&m->d_density_tmp, grid_bytes));
	DeviceEdit* d_refresh_edits = ctx->d_edits + (size_t)nrs_ctx::kInFlight * nrs_ctx::kMaxEdits; // the refresh's own operator table
	if (n_edits > 0) {
		DeviceEdit host_edits[nrs_ctx::kMaxEdits];
		for (int i = 0; i < n_edits; ++i) {
			if (!edits[i]) return fail(NRS_ERR_INVALID_ARG, "nrs_model_update_density_grid: NULL edit operator");
			host_edits[i] = edits[i]->de;
		}
		HIP_TRY(hipMemcpyAsync(d_refresh_edits, host_edits, sizeof(DeviceEdit) * n_edits, hipMemcpyHostToDevice, s));
		HIP_TRY(hipStreamSynchronize(s)); // host_edits is a stack array
	}
	if (u->reset_grid) HIP_TRY(hipMemsetAsync(m->d_density_grid, 0, grid_bytes, s));
	HIP_TRY(hipMemsetAsync(m->d_density_tmp, 0, grid_bytes, s));
	const uint64_t rng_nonuniform = pcg_advance(u->rng_state, u->rng_inc, 1ull << 32); // m_rng.advance() between the two draws
	NRS_TRY(launch_grid_update(m->dm, d_refresh_edits, n_edits, *u, rng_nonuniform, m->d_density_grid, m->d_density_tmp, ctx->n_cus, stream));
	u->rng_state = pcg_advance(u->rng_state, u->rng_inc, 2ull << 32);
	u->ema_step += 1;
	return refresh_bitfield(m, stream);
}
int nrs_model_get_march_accelerator(nrs_model* m, int which, float* h_box12, uint32_t* h_mask) {
	if (!m || !h_box12 || !h_mask || which < 0 || which > 1) return fail(NRS_ERR_INVALID_ARG, "nrs_model_get_march_accelerator: bad argument");
	if (!m->have_bitfield) return fail(NRS_ERR_STATE, "nrs_model_get_march_accelerator: occupancy not set (nrs_model_set_density_bitfield/_grid)");
	HIP_TRY(hipSetDevice(m->ctx->device));
	const OccAccel& a = which ? m->accel_exact : m->accel_any;
	for (int k = 0; k < 3; ++k) { h_box12[k] = a.box.mn[k]; h_box12[3 + k] = a.box.mx[k]; h_box12[6 + k] = a.cell[k]; h_box12[9 + k] = a.inv_cell[k]; }
	HIP_TRY(hipMemcpy(h_mask, a.mask, kCoarseWords * 4, hipMemcpyDeviceToHost));
	return NRS_OK;
}
int nrs_model_get_density_bitfield(nrs_model* m, uint8_t* h_out, size_t n_bytes) {
	if (!m || !h_out || n_bytes != NRS_BITFIELD_BYTES) return fail(NRS_ERR_INVALID_ARG, "nrs_model_get_density_bitfield: bad argument");
	HIP_TRY(hipSetDevice(m->ctx->device));
	HIP_TRY(hipMemcpy(h_out, m->d_bitfield, n_bytes, hipMemcpyDeviceToHost));
	return NRS_OK;
}

// ---- NerfNetwork operator ------------------------------------------------------------------------------------
static int check_net(nrs_model* m, const void* in, const void* out, const char* who) {
	if (!m || !in || !out) return fail(NRS_ERR_INVALID_ARG, std::string(who) + ": NULL argument");
	if (!m->have_params) return fail(NRS_ERR_STATE, std::string(who) + ": parameters not set (nrs_model_set_params)");
	return NRS_OK;
}
int nrs_network_inference(nrs_model* m, void* stream, uint32_t n, const float* d_in, void* d_out, uint32_t ld_out, int layout) {
	int s = check_net(m, d_in, d_out, "nrs_network_inference");
	if (s != NRS_OK) return s;
	if (layout == NRS_PLANES && ld_out < n) return fail(NRS_ERR_INVALID_ARG, "nrs_network_inference: ld_out < n");
	HIP_TRY(hipSetDevice(m->ctx->device));
	NRS_TRY(launch_network(m->dm, 0, n, d_in, NRS_NETWORK_INPUT_FLOATS, d_out, ld_out, layout, m->ctx->n_cus, stream));
	return NRS_OK;
}
int nrs_network_density(nrs_model* m, void* stream, uint32_t n, const float* d_in, uint32_t ld_in, void* d_out, uint32_t ld_out, int layout) {
	int s = check_net(m, d_in, d_out, "nrs_network_density");
	if (s != NRS_OK) return s;
	if (ld_in < 3) return fail(NRS_ERR_INVALID_ARG, "nrs_network_density: ld_in < 3");
	if (layout == NRS_PLANES && ld_out < n) return fail(NRS_ERR_INVALID_ARG, "nrs_network_density: ld_out < n");
	HIP_TRY(hipSetDevice(m->ctx->device));
	NRS_TRY(launch_network(m->dm, 1, n, d_in, ld_in, d_out, ld_out, layout, m->ctx->n_cus, stream));
	return NRS_OK;
}
int nrs_network_input_gradient(nrs_model* m, void* stream, uint32_t n, const float* d_in, uint32_t ld_in, float* d_grad_out) {
	const int st = check_net(m, d_in, d_grad_out, "nrs_network_input_gradient");
	if (st != NRS_OK) return st;
	if (ld_in < 3) return fail(NRS_ERR_INVALID_ARG, "nrs_network_input_gradient: ld_in < 3");
	HIP_TRY(hipSetDevice(m->ctx->device));
	NRS_TRY(launch_network(m->dm, 3, n, d_in, ld_in, d_grad_out, 3, 0, m->ctx->n_cus, stream));
	return NRS_OK;
}
int nrs_network_visualize_activation(nrs_model* m, void* stream, uint32_t layer, uint32_t dimension, uint32_t n, const float* d_in, float* d_out) {
	const int st = check_net(m, d_in, d_out, "nrs_network_visualize_activation");
	if (st != NRS_OK) return st;
	if (dimension >= network_layer_width(m->desc, layer))
		return fail(NRS_ERR_INVALID_ARG, "nrs_network_visualize_activation: no such unit (layers: hash grid 32 | density hidden 64 | rgb input 32 | one of 64 per rgb hidden layer)");
	HIP_TRY(hipSetDevice(m->ctx->device));
	NRS_TRY(launch_network(m->dm, 4, n, d_in, NRS_NETWORK_INPUT_FLOATS, d_out, 1, (int)(kernel_layer(m->desc, layer) | (dimension << 8)), m->ctx->n_cus, stream));
	return NRS_OK;
}
int nrs_density_on_grid(nrs_model* m, void* stream, const uint32_t res3d[3], const float aabb_min[3], const float aabb_max[3], int mask_with_density_grid,
                        float* d_out) {
	if (!m || !res3d || !aabb_min || !aabb_max || !d_out) return fail(NRS_ERR_INVALID_ARG, "nrs_density_on_grid: NULL argument");
	if (!m->have_params) return fail(NRS_ERR_STATE, "nrs_density_on_grid: parameters not set (nrs_model_set_params)");
	if ((uint64_t)res3d[0] * res3d[1] * res3d[2] > 0x7fffffffull) return fail(NRS_ERR_INVALID_ARG, "nrs_density_on_grid: more than 2^31 grid points");
	HIP_TRY(hipSetDevice(m->ctx->device));
	NRS_TRY(launch_grid_eval(m->dm, 0, res3d, aabb_min, aabb_max, nullptr, mask_with_density_grid ? m->d_density_grid : nullptr, d_out, m->ctx->n_cus, stream));
	return NRS_OK;
}
int nrs_rgba_on_grid(nrs_model* m, void* stream, const uint32_t res3d[3], const float render_aabb_min[3], const float render_aabb_max[3],
                     const float ray_dir[3], float* d_out_rgba) {
	if (!m || !res3d || !render_aabb_min || !render_aabb_max || !ray_dir || !d_out_rgba) return fail(NRS_ERR_INVALID_ARG, "nrs_rgba_on_grid: NULL argument");
	if (!m->have_params) return fail(NRS_ERR_STATE, "nrs_rgba_on_grid: parameters not set (nrs_model_set_params)");
	if ((uint64_t)res3d[0] * res3d[1] * res3d[2] > 0x7fffffffull) return fail(NRS_ERR_INVALID_ARG, "nrs_rgba_on_grid: more than 2^31 grid points");
	HIP_TRY(hipSetDevice(m->ctx->device));
	const float dir01[3] = {(ray_dir[0] + 1.0f) * 0.5f, (ray_dir[1] + 1.0f) * 0.5f, (ray_dir[2] + 1.0f) * 0.5f}; // warp_direction, not normalised (tn:430)
	NRS_TRY(launch_grid_eval(m->dm, 1, res3d, render_aabb_min, render_aabb_max, dir01, nullptr, d_out_rgba, m->ctx->n_cus, stream));
	return NRS_OK;
}
int nrs_project_selection_pixels(nrs_model* m, void* stream, const nrs_render_params* p, const int32_t* d_pixels_xy, uint32_t n_pixels,
                                 float transmittance_threshold, float* d_positions, uint32_t* d_cells, uint8_t* d_found) {
	if (!m || !p || (n_pixels && (!d_pixels_xy || !d_positions || !d_cells || !d_found)))
		return fail(NRS_ERR_INVALID_ARG, "nrs_project_selection_pixels: NULL argument");
	if (!m->have_params) return fail(NRS_ERR_STATE, "nrs_project_selection_pixels: parameters not set (nrs_model_set_params)");
	if (!m->have_bitfield) return fail(NRS_ERR_STATE, "nrs_project_selection_pixels: occupancy not set (nrs_model_set_density_bitfield/_grid)");
	{ const int pc = check_march_params(*p, "nrs_project_selection_pixels"); if (pc != NRS_OK) return pc; }
	HIP_TRY(hipSetDevice(m->ctx->device));
	NRS_TRY(launch_selection_rays(m->dm, *p, d_pixels_xy, n_pixels, transmittance_threshold, d_positions, d_cells, d_found, stream));
	return NRS_OK;
}
// Sampling directions of compute_poisson_boundary (growing_selection.cu:2241-2261), on the host with the host's libm as the
// reference does; jitter = the two (float)std::rand() / RAND_MAX draws per sample, supplied by the caller.
void nrs_poisson_sample_coords(const float* vertices, uint32_t n_verts, uint32_t sh_width, uint32_t hemisphere_width, const float* jitter,
                               const float aabb_min[3], const float aabb_max[3], float* coords7) {
	const uint32_t n_sh = sh_width * sh_width;
	for (uint32_t k = 0; k < n_verts; ++k)
		for (uint32_t i = 0; i < sh_width; ++i)
			for (uint32_t j = 0; j < sh_width; ++j) {
				const size_t s = (size_t)n_sh * k + (size_t)i * sh_width + j;
				const float u = ((float)i + jitter[2 * s]) / (float)(int)hemisphere_width;
				const float v = ((float)j + jitter[2 * s + 1]) / (float)(int)hemisphere_width;
				const float theta = (float)(2.f * M_PI * v);
				const float phi = acosf(2.f * u - 1.f);
				const float x = cosf(theta) * sinf(phi), y = sinf(theta) * sinf(phi), z = cosf(phi);
				float* c = coords7 + s * 7;
				for (int a = 0; a < 3; ++a) c[a] = (vertices[3 * k + a] - aabb_min[a]) / (aabb_max[a] - aabb_min[a]); // warp_position
				c[3] = 0.f;
				c[4] = (x + 1.f) * 0.5f; c[5] = (y + 1.f) * 0.5f; c[6] = (z + 1.f) * 0.5f; // warp_direction
			}
}
int nrs_poisson_boundary(nrs_model* m, const float* h_vertices, uint32_t n_verts, uint32_t sh_width, uint32_t hemisphere_width, const float* h_jitter,
                         int is_inside, float* h_density_out, float* h_sh_out) {
	if (!m || (n_verts && (!h_vertices || !h_jitter || !h_density_out || !h_sh_out))) return fail(NRS_ERR_INVALID_ARG, "nrs_poisson_boundary: NULL argument");
	if (!m->have_params) return fail(NRS_ERR_STATE, "nrs_poisson_boundary: parameters not set (nrs_model_set_params)");
	if (is_inside && !m->have_bitfield) return fail(NRS_ERR_STATE, "nrs_poisson_boundary: occupancy not set (nrs_model_set_density_bitfield/_grid)");
	if (sh_width == 0 || sh_width > 64 || hemisphere_width == 0) return fail(NRS_ERR_INVALID_ARG, "nrs_poisson_boundary: sampling widths out of range (1..64)");
	if (n_verts == 0) return NRS_OK;
	HIP_TRY(hipSetDevice(m->ctx->device));
	const uint32_t n_sh = sh_width * sh_width;
	const size_t n = (size_t)n_verts * n_sh;
	if (n > 0x7fffffffull) return fail(NRS_ERR_INVALID_ARG, "nrs_poisson_boundary: too many samples");
	std::vector<float> coords(n * 7);
	nrs_poisson_sample_coords(h_vertices, n_verts, sh_width, hemisphere_width, h_jitter, m->dm.aabb.mn, m->dm.aabb.mx, coords.data());
	float *d_coords = nullptr, *d_density = nullptr, *d_sh = nullptr;
	void* d_net = nullptr;
	hipError_t he = hipMalloc((void**)&d_coords, n * 7 * 4);
	if (he == hipSuccess) he = hipMalloc(&d_net, n * 16 * 2);
	if (he == hipSuccess) he = hipMalloc((void**)&d_density, (size_t)n_verts * 4);
	if (he == hipSuccess) he = hipMalloc((void**)&d_sh, (size_t)n_verts * 27 * 4);
	int rc = NRS_OK;
	if (he != hipSuccess) rc = fail_hip(he, "nrs_poisson_boundary: device allocation");
	if (rc == NRS_OK && hipMemcpy(d_coords, coords.data(), n * 7 * 4, hipMemcpyHostToDevice) != hipSuccess) rc = fail(NRS_ERR_HIP, "nrs_poisson_boundary: upload");
	if (rc == NRS_OK && (rc = launch_network(m->dm, 0, (uint32_t)n, d_coords, NRS_NETWORK_INPUT_FLOATS, d_net, 16, NRS_INTERLEAVED, m->ctx->n_cus, nullptr)) != NRS_OK)
		g_err = launch_last_error();
	if (rc == NRS_OK && (rc = launch_poisson_fit(m->dm, n_verts, n_sh, d_coords, d_net, is_inside, (float)(4 * M_PI / n_sh), d_density, d_sh, nullptr)) != NRS_OK)
		g_err = launch_last_error();
	if (rc == NRS_OK && (hipMemcpy(h_density_out, d_density, (size_t)n_verts * 4, hipMemcpyDeviceToHost) != hipSuccess ||
	                     hipMemcpy(h_sh_out, d_sh, (size_t)n_verts * 27 * 4, hipMemcpyDeviceToHost) != hipSuccess))
		rc = fail(NRS_ERR_HIP, "nrs_poisson_boundary: download");
	(void)hipFree(d_coords); (void)hipFree(d_net); (void)hipFree(d_density); (void)hipFree(d_sh);
	return rc;
}
int nrs_hashgrid_encode(nrs_model* m, void* stream, uint32_t n, const float* d_in, uint32_t ld_in, void* d_out) {
	int s = check_net(m, d_in, d_out, "nrs_hashgrid_encode");
	if (s != NRS_OK) return s;
	if (ld_in < 3) return fail(NRS_ERR_INVALID_ARG, "nrs_hashgrid_encode: ld_in < 3");
	HIP_TRY(hipSetDevice(m->ctx->device));
	NRS_TRY(launch_network(m->dm, 2, n, d_in, ld_in, d_out, 0, NRS_INTERLEAVED, m->ctx->n_cus, stream));
	return NRS_OK;
}

// ---- edit operators ------------------------------------------------------------------------------------------------
// ---- device-side tet LUT (nrs_cage.hip) ------------------------------------------------------------------------------
#define CAGE_TRY(expr)                                                     \
	do {                                                                   \
		int st_ = (expr);                                                  \
		if (st_ != NRS_OK) { g_err = cage_last_error(); return st_; }      \
	} while (0)

static int ensure_build_scratch(nrs_edit* e) {
	const size_t n_cells = (size_t)kGridVol * kCascades;
	if (!e->d_counts) {
		HIP_TRY(hipMalloc((void**)&e->d_counts, n_cells * 4));
		HIP_TRY(hipMemset(e->d_counts, 0, n_cells * 4));
	}
	if (!e->d_tile_sums) HIP_TRY(hipMalloc((void**)&e->d_tile_sums, kLutScanTiles * 4));
	if (!e->d_hit_masks) HIP_TRY(hipMalloc((void**)&e->d_hit_masks, (size_t)e->n_tets * kCascades * 16));
	if (!e->d_scratch) HIP_TRY(hipMalloc((void**)&e->d_scratch, 64));
	return NRS_OK;
}
// cell -> tet CSR of `d_verts` into e->d_lut_off / e->d_lut_idx (grown as needed); optionally the touched-cell bitfield.
// Synchronises the stream once (the entry count decides the idx allocation), like the reference's host builder does.
static int build_lut_on_device(nrs_edit* e, const float* d_verts, uint8_t* d_bitfield_out, hipStream_t s) {
	NRS_TRY(ensure_build_scratch(e));
	// cells of cascade 0 in an average tet's bounding box, from the mesh's box and tet count (six tets share a lattice cube's box): only the kernels' team size hangs on it
	float cells0 = 0.f;
	{
		const Box3& bb = e->de.bbox;
		const double vol = (double)std::max(bb.mx[0] - bb.mn[0], 0.f) * std::max(bb.mx[1] - bb.mn[1], 0.f) * std::max(bb.mx[2] - bb.mn[2], 0.f);
		const double side = std::cbrt(vol / std::max<double>(e->n_tets / 6.0, 1.0)) * kGrid;
		cells0 = (float)((side + 1.0) * (side + 1.0) * (side + 1.0));
	}
	CAGE_TRY(launch_lut_count_scan(e->n_tets, d_verts, e->de.tets, e->d_counts, e->d_tile_sums, e->d_lut_off, e->d_scratch + 6, e->d_hit_masks, cells0, s));
	uint32_t total = 0;
	HIP_TRY(hipMemcpyAsync(&total, e->d_scratch + 6, 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	if ((size_t)total > e->lut_idx_cap || !e->d_big_cells) {
		const size_t cap = std::max<size_t>((size_t)total + total / 2, 1024);
		uint32_t *fresh = nullptr, *fresh_big = nullptr;
		HIP_TRY(hipMalloc((void**)&fresh, cap * 4));
		HIP_TRY(hipMalloc((void**)&fresh_big, (size_t)lut_big_list_capacity(cap) * 4));
		(void)hipFree(e->d_lut_idx);
		(void)hipFree(e->d_big_cells);
		e->d_lut_idx = fresh;
		e->d_big_cells = fresh_big;
		e->lut_idx_cap = cap;
		e->de.lut_idx = fresh;
	}
	CAGE_TRY(launch_lut_fill(e->n_tets, d_verts, e->de.tets, e->d_counts, e->d_lut_off, e->d_lut_idx, d_bitfield_out, e->d_scratch + 7, e->d_big_cells, lut_big_list_capacity(e->lut_idx_cap), e->d_hit_masks, cells0, s));
	e->lut_n_idx = total;
	return NRS_OK;
}
// The fine look-up table of e's CURRENT LUT and plane records (both on the device, written on stream s): DeviceEdit::fine_*.  Leaves the operator without one (the kernels
// then scan the LUT's own lists) when the mesh reaches no cell, when even one fine cell per LUT cell would exceed kFineMaxCells, or when NRS_NO_FINE_LUT is set (A/B).
// Two small read-backs (the window, the entry count), like the LUT's own build.
static int build_fine_lut(nrs_edit* e, hipStream_t s) {
	static const bool off = dev_knob("NRS_NO_FINE_LUT") != nullptr;
	DeviceEdit& de = e->de;
	de.fine_off = nullptr;
	de.fine_idx = nullptr;
	memset(de.fine_win, 0, sizeof(de.fine_win));
	e->fine_n_idx = 0;
	if (off) return NRS_OK;
	if (!e->d_fine_win) HIP_TRY(hipMalloc((void**)&e->d_fine_win, (kCascades * 8 + 8) * 4));
	if (!e->d_fine_tiles) HIP_TRY(hipMalloc((void**)&e->d_fine_tiles, kFineScanTiles * 4));
	CAGE_TRY(launch_fine_window(de.lut_off, e->d_fine_win, s));
	int32_t win[kCascades * 8];
	HIP_TRY(hipMemcpyAsync(win, e->d_fine_win, sizeof(win), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	// per cascade, finest first: 4 x 4 x 4 fine cells per LUT cell while the budget lasts (then 2 x 2 x 2, then none); a cascade whose longest list is a mesh-in-a-cell
	// (thousands of tets: the coarse cascades of a small cage) keeps the LUT's own lists -- the scene's samples hardly stand there and the build would walk them 64 times
	uint64_t budget = kFineMaxCells;
	uint32_t base = 0;
	bool any = false;
	for (uint32_t c = 0; c < kCascades; ++c) {
		const int32_t* w = win + 8 * c;
		int32_t* f = de.fine_win[c];
		f[3] = (int32_t)base;
		f[7] = kFinePlain;
		if (w[0] > w[3]) { f[7] = 0; continue; } // no tet reaches this cascade: extent 0, every look-up finds nothing (as the LUT's empty lists say)
		if (w[7] > kFineMaxList) continue;
		for (int shift = 2; shift >= 1; --shift) {
			const uint64_t cells = ((uint64_t)(w[3] - w[0] + 1) << shift) * ((uint64_t)(w[4] - w[1] + 1) << shift) * ((uint64_t)(w[5] - w[2] + 1) << shift);
			if (cells > budget) continue;
			for (int a = 0; a < 3; ++a) { f[a] = w[a] << shift; f[4 + a] = (w[3 + a] - w[a] + 1) << shift; }
			f[7] = shift;
			budget -= cells;
			base += (uint32_t)cells;
			any = true;
			break;
		}
	}
	if (!any) { memset(de.fine_win, 0, sizeof(de.fine_win)); return NRS_OK; }
	const uint32_t n_cells = base, n_padded = (n_cells + 4095u) / 4096u * 4096u;
	if (n_padded > e->fine_cells_cap) {
		(void)hipFree(e->d_fine_off); (void)hipFree(e->d_fine_counts);
		e->d_fine_off = e->d_fine_counts = nullptr;
		e->fine_cells_cap = 0;
		const size_t cap = std::min<size_t>(kFineMaxCells, (size_t)n_padded + n_padded / 4 + 4095) / 4096 * 4096;
		HIP_TRY(hipMalloc((void**)&e->d_fine_off, (cap + 1) * 4));
		HIP_TRY(hipMalloc((void**)&e->d_fine_counts, cap * 4));
		e->fine_cells_cap = cap;
	}
	CAGE_TRY(launch_fine_count_scan(de, n_cells, e->d_fine_counts, e->d_fine_tiles, e->d_fine_off, (uint32_t*)e->d_fine_win + kCascades * 8, s));
	uint32_t n_idx = 0;
	HIP_TRY(hipMemcpyAsync(&n_idx, e->d_fine_win + kCascades * 8, 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	if ((size_t)n_idx > e->fine_idx_cap || !e->d_fine_idx) {
		(void)hipFree(e->d_fine_idx);
		e->d_fine_idx = nullptr;
		e->fine_idx_cap = 0;
		const size_t cap = std::max<size_t>((size_t)n_idx + n_idx / 2, 1024);
		HIP_TRY(hipMalloc((void**)&e->d_fine_idx, cap * 4));
		e->fine_idx_cap = cap;
	}
	CAGE_TRY(launch_fine_fill(de, n_cells, e->d_fine_off, e->d_fine_idx, s));
	e->fine_n_idx = n_idx;
	de.fine_off = e->d_fine_off;
	de.fine_idx = e->d_fine_idx;
	static const bool log_fine = dev_knob("NRS_FINE_LOG") != nullptr;
	if (log_fine) fprintf(stderr, "[nrs fine lut] %u fine cells, %u entries (the LUT holds %u), subdivision per cascade %d %d %d %d %d\n", n_cells, n_idx, e->lut_n_idx, de.fine_win[0][7],
	                      de.fine_win[1][7], de.fine_win[2][7], de.fine_win[3][7], de.fine_win[4][7]);
	return NRS_OK;
}
// everything that follows new deformed vertices in e->d_verts: bbox, LUT, rotations.  Synchronous.
static int rebuild_after_vertices(nrs_edit* e, hipStream_t s, bool build_fine_now = false) {
	NRS_TRY(ensure_build_scratch(e));
	CAGE_TRY(launch_bbox(e->n_vertices, e->d_verts, (float*)e->d_scratch, s));
	NRS_TRY(build_lut_on_device(e, e->d_verts, nullptr, s));
	if (e->d_rot) CAGE_TRY(launch_local_rotations(e->n_tets, e->d_verts, e->de.orig, e->de.tets, e->d_rot, s));
	CAGE_TRY(launch_tet_planes(e->n_tets, e->d_verts, e->de.tets, e->d_planes, s));
	if (build_fine_now) NRS_TRY(build_fine_lut(e, s));
	else { // (see nrs_edit::fine_stale)
		e->de.fine_off = nullptr;
		e->de.fine_idx = nullptr;
		memset(e->de.fine_win, 0, sizeof(e->de.fine_win));
		e->fine_stale = true;
		e->renders_since_move = 0;
	}
	uint32_t host[8];
	HIP_TRY(hipMemcpyAsync(host, e->d_scratch, sizeof(host), hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	memcpy(e->de.bbox.mn, host, 12);      // post_update_vertices, tet_mesh.cu:12-20
	memcpy(e->de.bbox.mx, host + 3, 12);
	warp_box(e->de.bbox, e->de.aabb, e->de.warped_bbox);
	e->lut_max_per_cell = host[7];
	return NRS_OK;
}

int nrs_edit_create(nrs_ctx* ctx, const nrs_model_desc* desc, const nrs_tet_mesh* mesh, nrs_edit** out) {
	if (!ctx || !desc || !mesh || !out) return fail(NRS_ERR_INVALID_ARG, "nrs_edit_create: NULL argument");
	if (!mesh->h_vertices || !mesh->h_original_vertices || !mesh->h_tets) return fail(NRS_ERR_INVALID_ARG, "nrs_edit_create: missing mesh array");
	if (mesh->n_tets == 0 || mesh->n_vertices == 0) return fail(NRS_ERR_INVALID_ARG, "nrs_edit_create: empty mesh");
	if (mesh->apply_poisson && (!mesh->h_boundary_shs || !mesh->h_boundary_outside_density || !mesh->h_boundary_residual_density))
		return fail(NRS_ERR_INVALID_ARG, "nrs_edit_create: apply_poisson set without the per-vertex membrane arrays");
	for (size_t i = 0; i < 4 * (size_t)mesh->n_tets; ++i)
		if (mesh->h_tets[i] >= mesh->n_vertices) return fail(NRS_ERR_INVALID_ARG, "nrs_edit_create: tet index out of range");
	const size_t n_cells = (size_t)kGridVol * kCascades;
	const bool host_lut = mesh->h_lut_offsets != nullptr;
	const uint32_t n_idx = host_lut ? mesh->h_lut_offsets[n_cells] : 0;
	if (n_idx && !mesh->h_lut_idx) return fail(NRS_ERR_INVALID_ARG, "nrs_edit_create: h_lut_idx is NULL");
	HIP_TRY(hipSetDevice(ctx->device));
	nrs_edit* e = new (std::nothrow) nrs_edit();
	if (!e) return fail(NRS_ERR_STATE, "out of host memory");
	e->ctx = ctx;
	e->n_vertices = mesh->n_vertices;
	e->n_tets = mesh->n_tets;
	DeviceEdit& de = e->de;
	for (int k = 0; k < 3; ++k) { de.aabb.mn[k] = desc->aabb_min[k]; de.aabb.mx[k] = desc->aabb_max[k]; }
	de.diag_pow2 = 1;
	for (int k = 0; k < 3; ++k) {
		const float diag = desc->aabb_max[k] - desc->aabb_min[k];
		int ex = 0;
		if (std::frexp(diag, &ex) != 0.5f) de.diag_pow2 = 0;
		de.inv_diag[k] = 1.0f / diag;
	}
	Box3 orig_bbox;
	box_of(mesh->h_vertices, mesh->n_vertices, de.bbox);           // post_update_vertices, tet_mesh.cu:12-20
	warp_box(de.bbox, de.aabb, de.warped_bbox);
	box_of(mesh->h_original_vertices, mesh->n_vertices, orig_bbox); // ctor, tet_mesh.h:100-107
	warp_box(orig_bbox, de.aabb, de.orig_warped_bbox);
	int s = NRS_OK;
	auto chk = [&](int r) { if (s == NRS_OK) s = r; };
	auto dev_alloc = [&](void** p, size_t bytes) { return hipMalloc(p, std::max<size_t>(bytes, 16)) == hipSuccess ? NRS_OK : fail(NRS_ERR_HIP, "nrs_edit_create: hipMalloc failed"); };
	chk(upload(e, mesh->h_original_vertices, 3 * (size_t)mesh->n_vertices, &de.orig));
	chk(upload(e, mesh->h_tets, 4 * (size_t)mesh->n_tets, &de.tets));
	// the tables a cage move rewrites are owned individually (not in `allocs`)
	chk(dev_alloc((void**)&e->d_verts, 12 * (size_t)mesh->n_vertices));
	chk(dev_alloc((void**)&e->d_lut_off, (n_cells + 1) * 4));
	if (s == NRS_OK && hipMemcpy(e->d_verts, mesh->h_vertices, 12 * (size_t)mesh->n_vertices, hipMemcpyHostToDevice) != hipSuccess)
		s = fail(NRS_ERR_HIP, "nrs_edit_create: vertex upload failed");
	chk(dev_alloc((void**)&e->d_planes, 128 * (size_t)mesh->n_tets));
	de.verts = e->d_verts;
	de.lut_off = e->d_lut_off;
	de.planes = e->d_planes;
	const bool want_rot = mesh->h_local_rotations != nullptr || mesh->correct_direction != 0;
	if (want_rot) {
		chk(dev_alloc((void**)&e->d_rot, 36 * (size_t)mesh->n_tets));
		de.rot = e->d_rot;
	}
	uint8_t* d_orig_bits = nullptr;
	chk(dev_alloc((void**)&d_orig_bits, NRS_BITFIELD_BYTES));
	if (d_orig_bits) e->allocs.push_back(d_orig_bits);
	de.orig_bitfield = d_orig_bits;
	de.copy = mesh->copy;
	de.apply_poisson = mesh->apply_poisson;
	de.residual_amplitude = mesh->residual_amplitude;
	if (mesh->apply_poisson) {
		chk(upload(e, mesh->h_boundary_shs, 27 * (size_t)mesh->n_vertices, &de.shs));
		chk(upload(e, mesh->h_boundary_outside_density, (size_t)mesh->n_vertices, &de.out_density));
		chk(upload(e, mesh->h_boundary_residual_density, (size_t)mesh->n_vertices, &de.res_density));
	}
	if (s != NRS_OK) { nrs_edit_destroy(e); return s; }
	auto bail = [&](int st) { nrs_edit_destroy(e); return st; };
	// touched cells of the CANONICAL mesh (build_original_tet_grid, tet_mesh.cu:76): handed over, or built here
	if (mesh->h_original_bitfield) {
		if (hipMemcpy(d_orig_bits, mesh->h_original_bitfield, NRS_BITFIELD_BYTES, hipMemcpyHostToDevice) != hipSuccess)
			return bail(fail(NRS_ERR_HIP, "nrs_edit_create: bitfield upload failed"));
	} else {
		int st = build_lut_on_device(e, de.orig, d_orig_bits, nullptr);
		if (st != NRS_OK) return bail(st);
	}
	if (host_lut) {
		hipError_t he = hipMemcpy(e->d_lut_off, mesh->h_lut_offsets, (n_cells + 1) * 4, hipMemcpyHostToDevice);
		(void)hipFree(e->d_lut_idx); // (a canonical-mesh build above may have left its list here)
		(void)hipFree(e->d_big_cells);
		e->d_lut_idx = nullptr;
		e->d_big_cells = nullptr; // re-sized together with the list on the next device build
		if (he == hipSuccess) he = hipMalloc((void**)&e->d_lut_idx, std::max<size_t>((size_t)n_idx * 4, 16));
		if (he == hipSuccess && n_idx) he = hipMemcpy(e->d_lut_idx, mesh->h_lut_idx, (size_t)n_idx * 4, hipMemcpyHostToDevice);
		if (he != hipSuccess) return bail(fail_hip(he, "nrs_edit_create: LUT upload"));
		e->lut_idx_cap = n_idx;
		e->lut_n_idx = n_idx;
		de.lut_idx = e->d_lut_idx;
		{
			int st = launch_tet_planes(e->n_tets, e->d_verts, de.tets, e->d_planes, nullptr);
			if (st != NRS_OK) return bail((g_err = cage_last_error(), st));
			if (hipDeviceSynchronize() != hipSuccess) return bail(fail(NRS_ERR_HIP, "nrs_edit_create: tet_planes_kernel failed"));
		}
		{ // (the fine look-up table of the LUT that was handed over)
			const int st = build_fine_lut(e, nullptr);
			if (st != NRS_OK) return bail(st);
		}
		if (mesh->h_local_rotations) {
			he = hipMemcpy(e->d_rot, mesh->h_local_rotations, 36 * (size_t)mesh->n_tets, hipMemcpyHostToDevice);
			if (he != hipSuccess) return bail(fail_hip(he, "nrs_edit_create: rotation upload"));
		} else if (want_rot) {
			int st = launch_local_rotations(e->n_tets, e->d_verts, de.orig, de.tets, e->d_rot, nullptr);
			if (st != NRS_OK) return bail((g_err = cage_last_error(), st));
			if (hipDeviceSynchronize() != hipSuccess) return bail(fail(NRS_ERR_HIP, "nrs_edit_create: rotation kernel failed"));
		}
	} else {
		int st = rebuild_after_vertices(e, nullptr, true); // LUT (+ rotations) of the deformed mesh, on the device; an operator at rest: with its fine table
		if (st != NRS_OK) return bail(st);
		if (mesh->h_local_rotations && hipMemcpy(e->d_rot, mesh->h_local_rotations, 36 * (size_t)mesh->n_tets, hipMemcpyHostToDevice) != hipSuccess)
			return bail(fail(NRS_ERR_HIP, "nrs_edit_create: rotation upload failed"));
	}
	*out = e;
	return NRS_OK;
}
// AffineBoundingBox bookkeeping (affine_bounding_box.cuh:40-101) for the boxes the kernels test.  R column-major.
namespace {
struct HostAffineBox { float center[3], scale[3], rot[9]; };
void affine_finish(const HostAffineBox& b, AffineBox& out) {
	// u = rot * scale.x * e_x etc.;  min = -0.5 * rot * scale + center
	for (int i = 0; i < 3; ++i) {
		out.u[i] = b.rot[i] * b.scale[0];
		out.v[i] = b.rot[3 + i] * b.scale[1];
		out.w[i] = b.rot[6 + i] * b.scale[2];
		// Eigen's 3-term reduction order x0 + (x1 + x2) in every small product / dot (Redux.h complete unrolling; see nrs_device.cuh)
		out.mn[i] = ((-0.5f * b.rot[i]) * b.scale[0] + ((-0.5f * b.rot[3 + i]) * b.scale[1] + (-0.5f * b.rot[6 + i]) * b.scale[2])) + b.center[i];
		out.center[i] = b.center[i];
	}
	out.uu = out.u[0] * out.u[0] + (out.u[1] * out.u[1] + out.u[2] * out.u[2]);
	out.vv = out.v[0] * out.v[0] + (out.v[1] * out.v[1] + out.v[2] * out.v[2]);
	out.ww = out.w[0] * out.w[0] + (out.w[1] * out.w[1] + out.w[2] * out.w[2]);
}
void affine_warp_box(HostAffineBox& b, const Box3& aabb) { // warp_box, :90-97
	for (int i = 0; i < 3; ++i) {
		const float diag = aabb.mx[i] - aabb.mn[i];
		b.center[i] = (b.center[i] - aabb.mn[i]) / diag;
		b.scale[i] = b.scale[i] / diag;
	}
}
} // namespace

int nrs_edit_create_affine(nrs_ctx* ctx, const nrs_model_desc* desc, const nrs_affine_duplication* op, nrs_edit** out) {
	if (!ctx || !desc || !op || !out) return fail(NRS_ERR_INVALID_ARG, "nrs_edit_create_affine: NULL argument");
	for (int i = 0; i < 3; ++i)
		if (!(op->scale[i] != 0.f) || !(op->selection_scale[i] > 0.f)) return fail(NRS_ERR_INVALID_ARG, "nrs_edit_create_affine: zero scale / empty selection box");
	nrs_edit* e = new (std::nothrow) nrs_edit();
	if (!e) return fail(NRS_ERR_STATE, "out of host memory");
	e->ctx = ctx;
	DeviceEdit& de = e->de;
	de.kind = kEditAffine;
	for (int k = 0; k < 3; ++k) { de.aabb.mn[k] = desc->aabb_min[k]; de.aabb.mx[k] = desc->aabb_max[k]; }
	HostAffineBox sel, dst;
	memcpy(sel.center, op->selection_center, 12); memcpy(sel.scale, op->selection_scale, 12); memcpy(sel.rot, op->selection_rot, 36);
	// update_destination (affine_duplication.h:77-90): translate, scale_with_vector, rotate (rot_matrix = R * rot_matrix)
	dst = sel;
	for (int i = 0; i < 3; ++i) { dst.center[i] = dst.center[i] + op->translation[i]; dst.scale[i] = dst.scale[i] * op->scale[i]; }
	for (int c = 0; c < 3; ++c)
		for (int r = 0; r < 3; ++r)
			dst.rot[3 * c + r] = op->rotation[r] * sel.rot[3 * c] + (op->rotation[3 + r] * sel.rot[3 * c + 1] + op->rotation[6 + r] * sel.rot[3 * c + 2]);
	affine_warp_box(dst, de.aabb);
	affine_warp_box(sel, de.aabb);
	affine_finish(dst, de.a_dst);
	affine_finish(sel, de.a_sel);
	for (int i = 0; i < 3; ++i) {
		de.a_translation[i] = op->translation[i] / (de.aabb.mx[i] - de.aabb.mn[i]); // m_warped_translation
		de.a_scale[i] = op->scale[i];
	}
	memcpy(de.a_rot, op->rotation, 36);
	de.a_hide_original = op->hide_original ? 1u : 0u;
	de.a_correct_dir = op->correct_dir ? 1u : 0u;
	*out = e;
	return NRS_OK;
}
void nrs_edit_destroy(nrs_edit* e) {
	if (!e) return;
	for (void* p : e->allocs) (void)hipFree(p);
	(void)hipFree(e->d_verts); (void)hipFree(e->d_lut_off); (void)hipFree(e->d_lut_idx); (void)hipFree(e->d_rot); (void)hipFree(e->d_planes);
	(void)hipFree(e->d_big_cells); (void)hipFree(e->d_counts); (void)hipFree(e->d_tile_sums); (void)hipFree(e->d_hit_masks); (void)hipFree(e->d_scratch); (void)hipFree(e->d_mvc); (void)hipFree(e->d_cage);
	(void)hipFree(e->d_fine_off); (void)hipFree(e->d_fine_counts); (void)hipFree(e->d_fine_idx); (void)hipFree(e->d_fine_tiles); (void)hipFree(e->d_fine_win);
	delete e;
}

// ---- per-move updates --------------------------------------------------------------------------------------------
int nrs_edit_set_mvc(nrs_edit* e, const float* h_weights, uint32_t n_cage_vertices) {
	if (!e || !h_weights || n_cage_vertices == 0) return fail(NRS_ERR_INVALID_ARG, "nrs_edit_set_mvc: bad argument");
	if (e->de.kind != kEditCage) return fail(NRS_ERR_STATE, "nrs_edit_set_mvc: not a cage operator");
	HIP_TRY(hipSetDevice(e->ctx->device));
	(void)hipFree(e->d_mvc); (void)hipFree(e->d_cage);
	e->d_mvc = nullptr; e->d_cage = nullptr; e->n_cv = 0;
	const size_t nw = (size_t)e->n_vertices * n_cage_vertices;
	HIP_TRY(hipMalloc((void**)&e->d_mvc, nw * 4));
	HIP_TRY(hipMalloc((void**)&e->d_cage, (size_t)n_cage_vertices * 12));
	HIP_TRY(hipMemcpy(e->d_mvc, h_weights, nw * 4, hipMemcpyHostToDevice));
	e->n_cv = n_cage_vertices;
	return NRS_OK;
}
int nrs_edit_update_cage(nrs_edit* e, void* stream, const float* h_cage_vertices, uint32_t n_cage_vertices) {
	if (!e || !h_cage_vertices) return fail(NRS_ERR_INVALID_ARG, "nrs_edit_update_cage: NULL argument");
	if (!e->d_mvc) return fail(NRS_ERR_STATE, "nrs_edit_update_cage: MVC weights not set (nrs_edit_set_mvc)");
	if (n_cage_vertices != e->n_cv) return fail(NRS_ERR_INVALID_ARG, "nrs_edit_update_cage: cage vertex count differs from the MVC weights'");
	HIP_TRY(hipSetDevice(e->ctx->device));
	hipStream_t s = (hipStream_t)stream;
	HIP_TRY(hipMemcpyAsync(e->d_cage, h_cage_vertices, (size_t)n_cage_vertices * 12, hipMemcpyHostToDevice, s));
	CAGE_TRY(launch_mvc_apply(e->n_vertices, e->n_cv, e->d_mvc, e->d_cage, e->d_verts, s));
	return rebuild_after_vertices(e, s);
}
// GrowingSelection::interpolate_poisson_boundary (growing_selection.cu:2350-2395): the link between nrs_poisson_boundary (per CAGE vertex) and the
// render kernel's membrane path (per TET vertex).  The per-cage-vertex factors are prepared here with the host libm's expf (the reference does this
// on the host: std::exp(float)); the V_tet x V_cage weighted sums run on the device in the reference's order.
int nrs_edit_poisson_interpolate(nrs_edit* e, void* stream, const float* h_gamma, uint32_t n_cage_vertices, const float* h_inside_density, const float* h_outside_density,
                                 const float* h_inside_shs, const float* h_outside_shs, float residual_amplitude) {
	if (!e || !h_inside_density || !h_outside_density || !h_inside_shs || !h_outside_shs || n_cage_vertices == 0) return fail(NRS_ERR_INVALID_ARG, "nrs_edit_poisson_interpolate: bad argument");
	if (e->de.kind != kEditCage) return fail(NRS_ERR_STATE, "nrs_edit_poisson_interpolate: not a cage operator");
	if (!h_gamma && (!e->d_mvc || e->n_cv != n_cage_vertices))
		return fail(NRS_ERR_STATE, "nrs_edit_poisson_interpolate: no gamma coordinates given and the operator holds no MVC weights for this cage (nrs_edit_set_mvc)");
	HIP_TRY(hipSetDevice(e->ctx->device));
	hipStream_t s = (hipStream_t)stream;
	const float min_step = 1.73205080757f / 1024; // MIN_CONE_STEPSIZE(), common_nerf.h:31
	std::vector<float> per_cage((size_t)n_cage_vertices * 30);
	for (uint32_t j = 0; j < n_cage_vertices; ++j) {
		const float alpha_out = 1 - expf(-h_outside_density[j] * min_step), alpha_in = 1 - expf(-h_inside_density[j] * min_step);
		const float w_outside = 1.f, w_inside = std::min(alpha_in / alpha_out, 1.f);
		float* c = per_cage.data() + 30 * (size_t)j;
		c[0] = alpha_out;
		c[1] = h_outside_density[j];
		c[2] = h_outside_density[j] - h_inside_density[j];
		for (int k = 0; k < 27; ++k) c[3 + k] = w_outside * h_outside_shs[27 * (size_t)j + k] - w_inside * h_inside_shs[27 * (size_t)j + k];
	}
	float *d_per_cage = nullptr, *d_gamma = nullptr;
	HIP_TRY(hipMalloc((void**)&d_per_cage, per_cage.size() * 4));
	auto bail = [&](int rc) { (void)hipFree(d_per_cage); (void)hipFree(d_gamma); return rc; };
	if (hipMemcpyAsync(d_per_cage, per_cage.data(), per_cage.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess) return bail(fail(NRS_ERR_HIP, "nrs_edit_poisson_interpolate: upload"));
	if (h_gamma) {
		if (hipMalloc((void**)&d_gamma, (size_t)e->n_vertices * n_cage_vertices * 4) != hipSuccess ||
		    hipMemcpyAsync(d_gamma, h_gamma, (size_t)e->n_vertices * n_cage_vertices * 4, hipMemcpyHostToDevice, s) != hipSuccess)
			return bail(fail(NRS_ERR_HIP, "nrs_edit_poisson_interpolate: upload of the gamma coordinates"));
	}
	if (!e->de.shs) { // the operator was created without membrane arrays: they are the operator's from now on
		void* d[3] = {nullptr, nullptr, nullptr};
		const size_t sizes[3] = {27 * (size_t)e->n_vertices * 4, (size_t)e->n_vertices * 4, (size_t)e->n_vertices * 4};
		for (int k = 0; k < 3; ++k) {
			if (hipMalloc(&d[k], std::max<size_t>(sizes[k], 16)) != hipSuccess) return bail(fail(NRS_ERR_HIP, "nrs_edit_poisson_interpolate: device allocation"));
			e->allocs.push_back(d[k]);
		}
		e->de.shs = (const float*)d[0]; e->de.out_density = (const float*)d[1]; e->de.res_density = (const float*)d[2];
	}
	const int rc = launch_poisson_interpolate(e->n_vertices, n_cage_vertices, h_gamma ? d_gamma : e->d_mvc, d_per_cage, (float*)e->de.shs, (float*)e->de.out_density,
	                                          (float*)e->de.res_density, s);
	if (rc != NRS_OK) { g_err = cage_last_error(); return bail(rc); }
	if (hipStreamSynchronize(s) != hipSuccess) return bail(fail(NRS_ERR_HIP, "nrs_edit_poisson_interpolate: synchronise")); // the staging buffers are freed below
	e->de.apply_poisson = 1u;
	e->de.residual_amplitude = residual_amplitude;
	return bail(NRS_OK);
}
// the per-tet-vertex membrane terms an operator holds ([V*27], [V], [V]); any pointer may be NULL
int nrs_edit_download_poisson(nrs_edit* e, float* h_boundary_shs, float* h_outside_density, float* h_residual_density) {
	if (!e) return fail(NRS_ERR_INVALID_ARG, "nrs_edit_download_poisson: NULL argument");
	if (!e->de.shs) return fail(NRS_ERR_STATE, "nrs_edit_download_poisson: the operator holds no membrane terms");
	HIP_TRY(hipSetDevice(e->ctx->device));
	if (h_boundary_shs) HIP_TRY(hipMemcpy(h_boundary_shs, e->de.shs, 27 * (size_t)e->n_vertices * 4, hipMemcpyDeviceToHost));
	if (h_outside_density) HIP_TRY(hipMemcpy(h_outside_density, e->de.out_density, (size_t)e->n_vertices * 4, hipMemcpyDeviceToHost));
	if (h_residual_density) HIP_TRY(hipMemcpy(h_residual_density, e->de.res_density, (size_t)e->n_vertices * 4, hipMemcpyDeviceToHost));
	return NRS_OK;
}
int nrs_edit_update_vertices(nrs_edit* e, void* stream, const float* h_vertices, uint32_t n_vertices) {
	if (!e || !h_vertices) return fail(NRS_ERR_INVALID_ARG, "nrs_edit_update_vertices: NULL argument");
	if (e->de.kind != kEditCage) return fail(NRS_ERR_STATE, "nrs_edit_update_vertices: not a cage operator");
	if (n_vertices != e->n_vertices) return fail(NRS_ERR_INVALID_ARG, "nrs_edit_update_vertices: vertex count differs from the mesh's");
	HIP_TRY(hipSetDevice(e->ctx->device));
	hipStream_t s = (hipStream_t)stream;
	HIP_TRY(hipMemcpyAsync(e->d_verts, h_vertices, (size_t)n_vertices * 12, hipMemcpyHostToDevice, s));
	return rebuild_after_vertices(e, s);
}
int nrs_edit_lut_size(const nrs_edit* e, uint32_t* n_idx, uint32_t* max_per_cell) {
	if (!e) return fail(NRS_ERR_INVALID_ARG, "nrs_edit_lut_size: NULL argument");
	if (n_idx) *n_idx = e->lut_n_idx;
	if (max_per_cell) *max_per_cell = e->lut_max_per_cell;
	return NRS_OK;
}
int nrs_edit_download(nrs_edit* e, float* h_vertices, uint32_t* h_lut_offsets, uint32_t* h_lut_idx, float* h_rotations, uint8_t* h_original_bitfield,
                      float* h_bbox6) {
	if (!e) return fail(NRS_ERR_INVALID_ARG, "nrs_edit_download: NULL argument");
	if (e->de.kind != kEditCage) return fail(NRS_ERR_STATE, "nrs_edit_download: not a cage operator");
	HIP_TRY(hipSetDevice(e->ctx->device));
	if (h_vertices) HIP_TRY(hipMemcpy(h_vertices, e->d_verts, (size_t)e->n_vertices * 12, hipMemcpyDeviceToHost));
	if (h_lut_offsets) HIP_TRY(hipMemcpy(h_lut_offsets, e->d_lut_off, ((size_t)kGridVol * kCascades + 1) * 4, hipMemcpyDeviceToHost));
	if (h_lut_idx && e->lut_n_idx) HIP_TRY(hipMemcpy(h_lut_idx, e->d_lut_idx, (size_t)e->lut_n_idx * 4, hipMemcpyDeviceToHost));
	if (h_rotations) {
		if (!e->d_rot) return fail(NRS_ERR_STATE, "nrs_edit_download: the edit has no local rotations");
		HIP_TRY(hipMemcpy(h_rotations, e->d_rot, (size_t)e->n_tets * 36, hipMemcpyDeviceToHost));
	}
	if (h_original_bitfield) HIP_TRY(hipMemcpy(h_original_bitfield, e->de.orig_bitfield, NRS_BITFIELD_BYTES, hipMemcpyDeviceToHost));
	if (h_bbox6) { memcpy(h_bbox6, e->de.bbox.mn, 12); memcpy(h_bbox6 + 3, e->de.bbox.mx, 12); }
	return NRS_OK;
}
int nrs_edit_map_rays(nrs_edit* e, void* stream, uint32_t n, float* d_coords, uint8_t* d_empty_mask) {
	if (!e || !d_coords || !d_empty_mask) return fail(NRS_ERR_INVALID_ARG, "nrs_edit_map_rays: NULL argument");
	HIP_TRY(hipSetDevice(e->ctx->device));
	NRS_TRY(launch_map_rays(e->de, n, d_coords, NRS_NETWORK_INPUT_FLOATS, 1, d_empty_mask, stream));
	return NRS_OK;
}
int nrs_edit_map_positions(nrs_edit* e, void* stream, uint32_t n, float* d_pos, uint32_t ld, uint8_t* d_empty_mask) {
	if (!e || !d_pos || !d_empty_mask) return fail(NRS_ERR_INVALID_ARG, "nrs_edit_map_positions: NULL argument");
	if (ld < 3) return fail(NRS_ERR_INVALID_ARG, "nrs_edit_map_positions: ld < 3");
	HIP_TRY(hipSetDevice(e->ctx->device));
	NRS_TRY(launch_map_rays(e->de, n, d_pos, ld, 0, d_empty_mask, stream));
	return NRS_OK;
}

// ---- renderer --------------------------------------------------------------------------------------------------------
static int tile_geometry(const nrs_render_params& p, uint32_t team, uint32_t& tiles_x, uint32_t& owned, uint32_t& n_packets, uint32_t& ppt_x) {
	const uint32_t W = (uint32_t)p.resolution[0], H = (uint32_t)p.resolution[1];
	const uint32_t pw = team >= 16 ? 2u : (team >= 4 ? 4u : 8u), ph = 64u / team / pw; // packet_pixel<TEAM>()
	if (p.tile_size == 0) {
		tiles_x = (W + pw - 1) / pw; // packets per image row
		owned = 1;
		ppt_x = 0;
		n_packets = tiles_x * ((H + ph - 1) / ph);
		return NRS_OK;
	}
	if (p.tile_size % 8) return fail(NRS_ERR_INVALID_ARG, "tile_size must be a multiple of 8");
	tiles_x = tile_pitch(W, p.tile_size); // (odd row pitch: indices beyond the image's last tile column are virtual)
	const uint32_t tiles_y = (H + p.tile_size - 1) / p.tile_size, total = tiles_x * tiles_y;
	const uint32_t stride = p.tile_stride ? p.tile_stride : 1;
	owned = p.tile_first < total ? (total - p.tile_first + stride - 1) / stride : 0;
	ppt_x = p.tile_size / pw;
	n_packets = owned * ppt_x * (p.tile_size / ph);
	return NRS_OK;
}
uint32_t nrs_render_tile_pitch(const nrs_render_params* p) {
	if (!p || p->struct_size != (uint32_t)sizeof(nrs_render_params) || p->resolution[0] <= 0 || p->tile_size == 0) return 0;
	return tile_pitch((uint32_t)p->resolution[0], p->tile_size);
}
uint32_t nrs_render_owned_tiles(const nrs_render_params* p) {
	if (!p || p->struct_size != (uint32_t)sizeof(nrs_render_params) || p->resolution[0] <= 0 || p->resolution[1] <= 0) return 0;
	uint32_t tx, owned, np, ppt;
	if (tile_geometry(*p, 1, tx, owned, np, ppt) != NRS_OK) return 0;
	return owned;
}

int nrs_render_nerf(nrs_model* m, const nrs_render_params* p, nrs_edit* const* edits, int n_edits, float* d_frame, float* d_depth,
                    uint32_t* d_steps, void* stream, nrs_render_stats* h_stats) {
	if (!m || !p || !d_frame || !d_depth) return fail(NRS_ERR_INVALID_ARG, "nrs_render_nerf: NULL argument");
	if (!m->have_params) return fail(NRS_ERR_STATE, "nrs_render_nerf: parameters not set (nrs_model_set_params)");
	if (!m->have_bitfield) return fail(NRS_ERR_STATE, "nrs_render_nerf: occupancy not set (nrs_model_set_density_bitfield/_grid)");
	{ const int pc = check_march_params(*p, "nrs_render_nerf"); if (pc != NRS_OK) return pc; }
	if (p->render_mode == NRS_RENDER_ENCODING_VIS && p->visualized_dimension >= network_layer_width(m->desc, p->visualized_layer))
		return fail(NRS_ERR_INVALID_ARG, "nrs_render_nerf: EncodingVis: visualized_layer is hash grid 32 | density hidden 64 | rgb input 32 | one of 64 per rgb hidden layer (base.json: 0..4) and visualized_dimension a unit of it");
	if (!std::isfinite(p->glow_y_cutoff) || p->glow_mode > 31u) return fail(NRS_ERR_INVALID_ARG, "nrs_render_nerf: glow_mode is a 5-bit mask and glow_y_cutoff must be finite");
	if (p->distortion_mode > 2u) return fail(NRS_ERR_INVALID_ARG, "nrs_render_nerf: distortion_mode must be 0 (None), 1 (Iterative) or 2 (FTheta)");
	for (int i = 0; i < 7; ++i)
		if (!std::isfinite(p->distortion_params[i])) return fail(NRS_ERR_INVALID_ARG, "nrs_render_nerf: distortion parameters must be finite");
	if (p->d_envmap && (p->envmap_resolution[0] < 1 || p->envmap_resolution[1] < 1)) return fail(NRS_ERR_INVALID_ARG, "nrs_render_nerf: envmap without a resolution");
	if (p->d_distortion_map && (p->distortion_resolution[0] < 1 || p->distortion_resolution[1] < 1)) return fail(NRS_ERR_INVALID_ARG, "nrs_render_nerf: distortion map without a resolution");
	if (p->render_mode > NRS_RENDER_SLICE && p->render_mode != NRS_RENDER_ENCODING_VIS) return fail(NRS_ERR_INVALID_ARG, "nrs_render_nerf: unknown render mode");
	if (!std::isfinite(p->dof) || !std::isfinite(p->slice_plane_z) || !std::isfinite(p->depth_scale)) return fail(NRS_ERR_INVALID_ARG, "nrs_render_nerf: dof / slice_plane_z / depth_scale must be finite");
	if (p->dof != 0.f && p->slice_plane_z == 0.f) return fail(NRS_ERR_INVALID_ARG, "nrs_render_nerf: dof != 0 needs a focus distance (slice_plane_z = m_slice_plane_z + m_scale != 0)");
	if (n_edits < 0 || n_edits > nrs_ctx::kMaxEdits) return fail(NRS_ERR_INVALID_ARG, "nrs_render_nerf: too many edit operators");
	if (n_edits > 0 && !edits) return fail(NRS_ERR_INVALID_ARG, "nrs_render_nerf: edits is NULL");
	nrs_ctx* ctx = m->ctx;
	HIP_TRY(hipSetDevice(ctx->device));
	hipStream_t s = (hipStream_t)stream;
	std::unique_lock<std::mutex> launch_lock(ctx->launch_mutex); // held until the launch is enqueued and the slot's book-keeping is written (released before the statistics' sync)
	const uint32_t slot = ctx->launch_serial.fetch_add(1u) % (uint32_t)nrs_ctx::kInFlight;
	if (ctx->slot_used[slot] && ctx->slot_stream[slot] != s) HIP_TRY(hipStreamWaitEvent(s, ctx->slot_done[slot], 0));
	// The statistics / queue block of this launch.  A slot owns two: the render kernel's last workgroup zeroes the one it did NOT use, which the slot's next
	// launch takes (launches of a slot are ordered: same stream, or the event wait above) -- so a frame costs no memset (two 5-us fill kernels per frame in
	// the round-3 timeline: 2 % of a 1/8 share-frame).  The Slice path and a launch after a failed one still clear their block the plain way.
	RenderCounters* d_counters_slot = ctx->d_counters + 2 * slot + ctx->counter_parity[slot];
	RenderCounters* d_counters_other = ctx->d_counters + 2 * slot + (ctx->counter_parity[slot] ^ 1u);
	DeviceEdit* d_edits_slot = ctx->d_edits + (size_t)slot * nrs_ctx::kMaxEdits;

	RenderArgs a{};
	a.p = *p;
	if (p->render_mode == NRS_RENDER_ENCODING_VIS) a.p.visualized_layer = kernel_layer(m->desc, p->visualized_layer); // (the kernels number base.json's layers)
	uint32_t owned_tiles = 0;
	a.team = 1;
	a.fill_lanes = 4;
	int st = tile_geometry(*p, 1, a.tiles_x, owned_tiles, a.n_packets, a.packets_per_tile_x);
	if (st != NRS_OK) return st;
	a.n_edits = n_edits;
	a.any_poisson = 0;
	if (n_edits > 0) {
		DeviceEdit host_edits[nrs_ctx::kMaxEdits];
		for (int i = 0; i < n_edits; ++i) {
			if (!edits[i]) return fail(NRS_ERR_INVALID_ARG, "nrs_render_nerf: NULL edit operator");
			if (edits[i]->fine_stale && ++edits[i]->renders_since_move >= 2u) { // the cage has come to rest: its fine look-up table (nrs_edit::fine_stale)
				edits[i]->fine_stale = false;
				NRS_TRY(build_fine_lut(edits[i], s));
			}
			host_edits[i] = edits[i]->de;
			a.any_poisson |= edits[i]->de.apply_poisson;
			a.any_affine |= (edits[i]->de.kind == kEditAffine) ? 1u : 0u;
		}
		// the operator table of a slot is re-sent only when it changed (a viewer renders many frames per gizmo move)
		DeviceEdit* shadow = ctx->edits_shadow.data() + (size_t)slot * nrs_ctx::kMaxEdits;
		if (ctx->shadow_n[slot] != n_edits || memcmp(shadow, host_edits, sizeof(DeviceEdit) * n_edits) != 0) {
			HIP_TRY(hipMemcpyAsync(d_edits_slot, host_edits, sizeof(DeviceEdit) * n_edits, hipMemcpyHostToDevice, s));
			memcpy(shadow, host_edits, sizeof(DeviceEdit) * n_edits);
			ctx->shadow_n[slot] = n_edits;
		}
	}
	a.edits = d_edits_slot;
	{
		static const uint32_t dbg = []() {
			const char* e = dev_knob("NRS_DEBUG");
			const char* sk = dev_knob("NRS_SKIP_PAIRS"); // bit it: level pair (2 it, 2 it + 1) is not gathered (nrs_mlp.cuh: KIND_SKIP; measurement only)
			return (e ? (uint32_t)atoi(e) & 0xffu : 0u) | (sk ? ((uint32_t)strtoul(sk, nullptr, 0) & 0xffu) << 8 : 0u);
		}();
		a.dbg = dbg;
	}
	{ // Cone stepping is what the reference switches on for aabb_scale > 1 (tn:3410-3425): scenes whose fine hashed levels no two samples of a wave share, so that their
	  // 8 MB of table lines thrash the 4 MB L2 of an XCD -- the GATE instantiation time-multiplexes it (nrs_mlp.cuh encode_to_lds; profiles/r06_garden.md: L2 misses per
	  // sample 11.5 -> 7.2, +5 % on the garden frame).  A unit-cube scene (constant steps) never takes it: there the gate costs 18 %.
		static const int gate_on = []() { const char* e = dev_knob("NRS_L2_GATE"); return e ? atoi(e) : 1; }(); // (0: off -- A/B)
		// (only where the gate has something to separate: the two, three or four finest level PAIRS hashed without records -- one phase each -- and records below them;
		// more hashed pairs than phases would run the GATE instantiation ungated, 5 % behind the default kernel: profiles/r06/ab_gate_phases_*.txt)
		const LevelParams* lv = m->dm.levels;
		int hashed_pairs = 0;
		for (int it = 7; it >= 0 && lv[2 * it].hashed && lv[2 * it + 1].hashed && !lv[2 * it].cached && !lv[2 * it + 1].cached; --it) ++hashed_pairs;
		// (= kGateMaxPhases of the kernels.  Six phases for a model without any records: 4.36 -> 4.18 Gsamples/s, profiles/r06/ab_gate6_garden_nocache.txt -- a frame's round has
		// no time for six waits; the occupancy refresh, one gather per wave, gains with up to six: NRS_REFRESH_GATE_PHASES)
		a.gate = (gate_on && p->cone_angle_constant > 0.f && hashed_pairs >= 2 && hashed_pairs <= 4) ? 1u : 0u;
	}
	// everything of render_nerf's surface beyond Shade / Cost with a pinhole camera runs the EXTRA instantiation (one lane per ray)
	a.extra = ((p->render_mode != NRS_RENDER_SHADE && p->render_mode != NRS_RENDER_COST) || p->show_accel || p->dof != 0.f || p->distortion_mode || p->d_distortion_map ||
	           p->d_envmap || p->glow_mode) ? 1u : 0u;
	if (m->dm.rgb_deep && !a.extra) {
		// A third rgb hidden layer: the automatic schedule has a DEEP instantiation for the plain case (Shade / Cost, cage edits without the membrane correction, default
		// roundings, no forced schedule); everything else of such a network runs the DEEP twins of the catch-all (launch_render)
		static const bool env_sched = (dev_knob("NRS_TEAM") && atoi(dev_knob("NRS_TEAM")) != 0) || (dev_knob("NRS_HYBRID") && atoi(dev_knob("NRS_HYBRID")) == 0) || dev_knob("NRS_RENDER_CFG");
		const bool plain = !a.any_poisson && !a.any_affine && m->dm.numerics == 0u && !ctx->lane_teams && !env_sched && !(a.dbg & 4u);
		if (!plain) a.extra = 1u;
	}
	if (p->render_mode == NRS_RENDER_SLICE) { // tn:3109-3162: no marching at all; one network evaluation per owned pixel
		a.frame = d_frame; a.depth = d_depth; a.steps = d_steps; a.counters = d_counters_slot;
		HIP_TRY(hipMemsetAsync(d_counters_slot, 0, sizeof(RenderCounters), s));
		ctx->counters_clean[slot] = false; // (slice_kernel leaves its statistics in the block and cleans nothing: the slot's next launch clears it)
		NRS_TRY(launch_slice(m->dm, a, ctx->n_cus, s));
		HIP_TRY(hipEventRecord(ctx->slot_done[slot], s));
		ctx->slot_stream[slot] = s;
		ctx->slot_used[slot] = true;
		launch_lock.unlock();
		if (h_stats) {
			RenderCounters c;
			HIP_TRY(hipMemcpyAsync(&c, d_counters_slot, sizeof(c), hipMemcpyDeviceToHost, s));
			HIP_TRY(hipStreamSynchronize(s));
			h_stats->n_samples = c.n_samples;
			h_stats->n_rays_alive = c.n_rays_alive;
			h_stats->n_rays_hit = c.n_rays_hit;
		}
		return NRS_OK;
	}
	{ // lane teams (render_kernel's TEAM) when the launch cannot fill the GPU with one ray per lane.  Rays per lane is
	  // estimated from the share of pixels that became rays in the last finished launch (written by its last workgroup;
	  // 0.25 until one has finished).  Measured on 1080p lego (0.22 of the pixels hit), frame shares 1/1 .. 1/8, ms per
	  // launch with 1 / 2 / 4 lanes per ray: 3.09 2.89 3.31 | 2.07 1.78 1.95 | 1.56 1.25 1.21 | 1.19 0.93 0.88; 8 and 16
	  // lanes lose everywhere (1.07, 1.72 at 1/8); 2560x1440: 4.31 4.52 5.64; aabb-16 1080p (every pixel hits): 9.64 9.76 10.9
		static const bool hybrid_on = []() { const char* e = dev_knob("NRS_HYBRID"); return !e || atoi(e) != 0; }();
		static const int env_forced = []() { const char* e = dev_knob("NRS_TEAM"); return e ? atoi(e) : 0; }();
		const int forced = ctx->lane_teams ? ctx->lane_teams : env_forced;
		const unsigned long long fb = ctx->h_feedback ? __atomic_load_n(ctx->h_feedback, __ATOMIC_RELAXED) : 0ull;
		const double hit_share = (fb >> 32) ? (double)(uint32_t)fb / (double)(fb >> 32) : 0.25;
		a.pixels_owned = (uint32_t)std::min<uint64_t>((uint64_t)a.n_packets * 64ull, 0xffffffffull);
		// Launches of this context still running on OTHER streams (frames in flight: a rank of a multi-GPU job that overlaps its frames, a viewer that
		// double-buffers) share the GPU with this one: the launch gets 1 / (1 + busy) of the lanes, and once three or more overlap the GPU is full
		// whatever the size of one launch -- lane teams (a latency device) then only cost fill passes.  Measured, 1/8 share of the 1080p bench frame,
		// ms per share-frame with 1 / 2 / 4 frames in flight: automatic choice before this rule 0.69 / 0.69 / 0.46, one lane per ray 0.99 / 0.54 / 0.36,
		// two lanes 0.77 / 0.47 / 0.49 (tools/scale_probe_teams.py) -- i.e. 0.95 of the single-GPU per-GPU throughput at an eighth of the frame.
		uint32_t busy = 0; // = number of OTHER streams with an unfinished launch (launches queued behind one another on a stream do not overlap)
		hipStream_t seen[nrs_ctx::kInFlight];
		for (int k = 0; k < nrs_ctx::kInFlight; ++k) {
			if ((uint32_t)k == slot || !ctx->slot_used[k] || ctx->slot_stream[k] == s) continue;
			bool dup = false;
			for (uint32_t q = 0; q < busy; ++q) dup = dup || seen[q] == ctx->slot_stream[k];
			if (!dup && hipEventQuery(ctx->slot_done[k]) == hipErrorNotReady) seen[busy++] = ctx->slot_stream[k];
		}
		(void)hipGetLastError(); // hipErrorNotReady is an answer, not an error
		const double rays_per_lane = hit_share * (double)a.pixels_owned * (double)(1u + busy) / (64.0 * 16.0 * (double)ctx->n_cus);
		// Round 3: team rounds test the team's next positions in parallel and waves hand rays over, so the small-launch schedule (packets of 16 / 32 / 64
		// pixels = 4 / 2 / 1 lanes on a pixel during the fill, every generation sized by the rays its wave has pending) is the automatic choice up to
		// the sizes where the hybrid schedule of whole images takes over.  tools/schedule_probe.py (profiles/r03_schedules.md), ms per frame with
		// 16- / 32- / 64-pixel packets | hybrid | fixed 2 lanes per ray: 640x360 0.66 / 0.77 / 1.18 | 1.15 | 0.85, 960x540 0.93 / 0.96 / 1.16 | 1.20 | 1.12,
		// 1280x720 1.40 / 1.28 / 1.51 | 1.43 | 1.44, 1600x900 2.02 / 1.76 / 1.85 | 1.83 | 1.90, 1080p 2.77 / 2.36 / 2.37 | 2.35 | 2.48, 1440p 4.66 / 3.86 / 3.71 |
		// 3.79 | 3.96; a rank's tiles of the 1080p frame, N = 8: 0.53 / 0.57 / 0.89 | - | 0.57, N = 4: 0.85 / 0.82 / 1.01 | - | 0.94, N = 2: 1.49 / 1.32 / 1.43 | - | 1.47,
		// N = 1: 2.78 / 2.36 / 2.37 | - | 2.49.  With frames in flight the thresholds hold for the rays of ALL overlapping launches.
		uint32_t fill_lanes = rays_per_lane <= 0.6 ? 4u : (rays_per_lane <= 2.6 ? 2u : 1u);
		// the fill runs once per pixel and lane on it: keep it to ~16 passes over the GPU (an all-miss 1080p frame is 8)
		while (fill_lanes > 1 && (double)fill_lanes * (double)a.pixels_owned * (double)(1u + busy) > 17.0 * 64.0 * 16.0 * (double)ctx->n_cus) fill_lanes >>= 1;
		// The hybrid schedule (64-ray generations at one lane per ray for the bulk of the queue, lane teams for its tail) was the choice for whole images with many
		// rays per lane until the queue's chunks went from 8 to 2 packets; since then the small-launch schedule wins there too -- bench frames, Gsamples/s hybrid /
		// 64-pixel packets: lego + cage 10.8 / 12.0, varied opacity 10.2 / 10.6, aabb-16 (7.9 rays per lane) 4.79 / 4.87 -- and hybrid runs only when forced (-1).
		const bool small_launch = true;
		uint32_t team = 1; // fixed lanes per ray: only when forced
		if (forced == 1 || forced == 2 || forced == 4) team = (uint32_t)forced;
		if (forced == -1) team = 1;
		// (the EXTRA / run-time-numerics / mixed membrane + affine instantiations are built for one lane per ray; the membrane correction of cage edits
		// alone runs the automatic schedule since round 4, AffineDuplication since round 6)
		const bool poisson_teams = a.any_poisson && !a.any_affine && !a.extra && m->dm.numerics == 0u;
		const bool affine_teams = a.any_affine && !a.any_poisson && !a.extra && m->dm.numerics == 0u; // (round 6: the AFFINE instantiation of the automatic schedule)
		const bool one_lane_only = (a.any_poisson && !poisson_teams) || (a.any_affine && !affine_teams) || a.extra;
		if (one_lane_only || a.any_poisson) team = 1; // (fixed 2 / 4 lanes per ray exist for the default kernel only; a forced size leaves the membrane path on the catch-all)
		static const bool log_teams = dev_knob("NRS_TEAM_LOG") != nullptr;
		if (log_teams) fprintf(stderr, "[nrs team] pixels=%u hit_share=%.3f busy=%u rays/lane=%.3f small-launch=%d fill lanes=%u forced=%d\n", a.pixels_owned, hit_share, busy, rays_per_lane, (int)small_launch, fill_lanes, forced);
		static const uint32_t tail_target = []() { const char* e = dev_knob("NRS_TAIL_TARGET"); return e && atoi(e) >= 1 ? (uint32_t)std::min(atoi(e), 64) : 24u; }(); // (<= kRing - 64: the fill adds up to 64 rays per packet to a 128-entry ring) 8 / 16 / 24 / 32 / 48: 8.92 / 8.91 / 9.11 / 9.01 / 8.47 Gsamples/s
		a.tail_target = tail_target;
		static const uint32_t reteam = []() { const char* e = dev_knob("NRS_RETEAM"); return e ? (uint32_t)atoi(e) : 3u; }(); // bit 0: at the end of a wave's work, bit 1: whenever a tail generation has thinned out
		a.reteam = reteam;
		static const uint32_t steal = []() { const char* e = dev_knob("NRS_STEAL"); return e ? (uint32_t)atoi(e) : 1u; }();
		a.steal = ctx->handover >= 0 ? (uint32_t)ctx->handover : steal;
		if (((small_launch && !forced && hybrid_on) || forced == -2 || forced == -3 || forced == -4) && !one_lane_only) {
			// few rays for the GPU: 4x4 packets only, and every generation takes ALL the rays its wave has pending with as many
			// lanes per ray as fit (4 up to 16 rays, 2 up to 32), so that no wave is left with a second, nearly empty generation
			// (1/8 share of the bench frame: 0.88 -> 0.82 ms).
			a.team = 0;
			a.all_tail = 1;
			a.fill_lanes = forced == -4 ? 1u : (forced == -3 ? 2u : (forced == -2 ? 4u : fill_lanes));
			NRS_TRY(tile_geometry(*p, a.fill_lanes, a.tiles_x, owned_tiles, a.n_packets, a.packets_per_tile_x));
			static const uint32_t all_tail_target = []() { const char* e = dev_knob("NRS_ALLTAIL_TARGET"); return e && atoi(e) >= 1 ? (uint32_t)std::min(atoi(e), 64) : 16u; }(); // (<= kRing - 64, as above)
			a.tail_target = all_tail_target;
		} else if (team > 1 && (forced > 0 || p->tile_size != 0 || !hybrid_on)) {
			a.team = team;
			NRS_TRY(tile_geometry(*p, team, a.tiles_x, owned_tiles, a.n_packets, a.packets_per_tile_x));
		} else if (p->tile_size == 0 && !a.any_poisson && !a.any_affine && !a.extra && (forced == -1 || (!forced && hybrid_on))) { // (the hybrid schedule: forced only)
			// whole images with more rays than the small-launch schedule is for: hybrid (one lane per ray, lane teams for the tail of the queue)
			// hybrid: every 3rd packet row leaves the 8x8 list and joins the end of the queue as 4x4 tail packets (packet_pixel_bulk/_tail);
			// measured on 1080p lego + cage, every 2nd / 3rd / 4th / 6th / 8th / 16th row: 8.88 / 8.89 / 8.79 / 8.75 / 8.65 / 8.65 Gsamples/s
			static const uint32_t tail_every = []() { const char* e = dev_knob("NRS_TAIL_EVERY"); return e && atoi(e) >= 2 ? (uint32_t)atoi(e) : 3u; }();
			const uint32_t rows = ((uint32_t)p->resolution[1] + 7u) / 8u, tail_rows = rows / tail_every;
			a.tail_every = tail_every;
			if (tail_rows) {
				static const uint32_t tail_fill = []() { const char* e = dev_knob("NRS_TAIL_FILL"); return e && (atoi(e) == 1 || atoi(e) == 2 || atoi(e) == 4) ? (uint32_t)atoi(e) : 4u; }();
				a.team = 0;
				a.fill_lanes = tail_fill; // lanes on a pixel while a tail packet is filled: packets of 4x4 / 8x4 / 8x8 pixels (packet_pixel_tail)
				a.p_big = (rows - tail_rows) * a.tiles_x;
				a.n_packets = a.p_big + tail_rows * a.tiles_x * tail_fill;
			}
		}
		a.feedback = ctx->d_feedback;
	}
	a.max_steps = p->max_march_steps ? p->max_march_steps : 10000u; // MARCH_ITER, testbed_nerf.cu:56
	a.frame = d_frame;
	a.depth = d_depth;
	a.steps = d_steps;
	a.counters = d_counters_slot;
	a.counters_next = d_counters_other;
	a.wave_log = (a.dbg & 4u) ? ctx->d_wave_log : nullptr;
	if (a.wave_log) HIP_TRY(hipMemsetAsync(ctx->d_wave_log, 0, 8192 * 4 * 8, s));
	if (!ctx->counters_clean[slot]) HIP_TRY(hipMemsetAsync(d_counters_slot, 0, sizeof(RenderCounters), s));
	ctx->counters_clean[slot] = false; // (until the launch is known to be enqueued: its last workgroup cleans the other block)
	if (a.n_packets == 0) { // nothing to launch (no owned tiles): the block stays as it is -- zero
		ctx->counters_clean[slot] = true;
	} else {
		NRS_TRY(launch_render(model_for_launch(m, *p), a, ctx->n_cus, s));
		ctx->counter_parity[slot] ^= 1u;
		ctx->counters_clean[slot] = true;
	}
	HIP_TRY(hipEventRecord(ctx->slot_done[slot], s));
	ctx->slot_stream[slot] = s;
	ctx->slot_used[slot] = true;
	launch_lock.unlock();
	if (h_stats) {
		RenderCounters c;
		HIP_TRY(hipMemcpyAsync(&c, d_counters_slot, sizeof(c), hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		h_stats->n_samples = c.n_samples;
		h_stats->n_rays_alive = c.n_rays_alive;
		h_stats->n_rays_hit = c.n_rays_hit;
		ctx->last_handover = c.walk[7];
		if (a.dbg & 12u) fprintf(stderr, "[nrs hand-over] %llu rays in %llu hand-overs\n", c.walk[7] & 0xffffffffull, c.walk[7] >> 32);
		if (a.dbg & 4u) {
			static const char* names[8] = {"fill", "refill", "setup+warp", "gather", "sh+mlp", "composite+march+shade", "-", "exit"};
			unsigned long long tot = 0;
			for (int i = 0; i < 8; ++i) if (i != 6) tot += c.phase_cycles[i];
			fprintf(stderr, "[nrs phases] samples=%llu", (unsigned long long)c.n_samples);
			for (int i = 0; i < 8; ++i)
				if (c.phase_cycles[i] && i != 6) fprintf(stderr, " %s=%.1f%%", names[i], 100.0 * (double)c.phase_cycles[i] / (double)tot);
			fprintf(stderr, " | mean wave lifetime = %.1f%% of the longest (%.2f Mcycles)", 100.0 * ((double)tot / 4096.0) / (double)c.phase_cycles[6], (double)c.phase_cycles[6] / 1e6);
			fprintf(stderr, "\n");
			fprintf(stderr, "[nrs walk] fill: %llu lane iterations in %llu wave trips (%.1f lanes busy per trip); march: %llu lane iterations in %llu wave trips "
			        "(%.1f lanes/trip), %llu of %llu rounds needed > 1 trip; live lanes per round %.1f\n",
			        c.walk[0], c.walk[1], c.walk[1] ? (double)c.walk[0] / (double)c.walk[1] : 0.0, c.walk[2], c.walk[3],
			        c.walk[3] ? (double)c.walk[2] / (double)c.walk[3] : 0.0, c.walk[6], c.walk[4], c.walk[4] ? (double)c.walk[5] / (double)c.walk[4] : 0.0);
			if (c.walk[8])
				fprintf(stderr, "[nrs cage scan] %llu samples inside a deformed box (%.1f %% of the samples), %llu of them found a tet; rounds with such a sample: %llu of %llu (%.1f %%); "
				        "candidates tested %llu (%.2f per sample in the box), scan wave trips %llu (%.2f per round that scans)\n",
				        c.walk[8], 100.0 * (double)c.walk[8] / (double)std::max<unsigned long long>(c.n_samples, 1), c.walk[12], c.walk[9], c.walk[4], 100.0 * (double)c.walk[9] / (double)std::max<unsigned long long>(c.walk[4], 1),
				        c.walk[10], (double)c.walk[10] / (double)c.walk[8], c.walk[11], (double)c.walk[11] / (double)std::max<unsigned long long>(c.walk[9], 1));
			// per-wave log: when did each wave finish (wall clock), when did it first find the frame's queue empty
			std::vector<unsigned long long> wl(8192 * 4);
			HIP_TRY(hipMemcpy(wl.data(), ctx->d_wave_log, wl.size() * 8, hipMemcpyDeviceToHost));
			if (const char* dump = dev_knob("NRS_WAVE_LOG_FILE")) { // raw log of the LAST launch with statistics, for tools/wave_log_report.py
				if (FILE* f = fopen(dump, "wb")) { fwrite(wl.data(), 8, wl.size(), f); fclose(f); }
			}
			std::vector<std::array<unsigned long long, 4>> rec; // {end tick (10 ns), rounds | rounds before queue-empty << 16 | t_queue_empty << 32, packets, xcc}
			unsigned long long t0min = ~0ull;
			for (size_t i = 0; i < 8192; ++i) if (wl[4 * i]) t0min = std::min(t0min, wl[4 * i + 3] >> 32);
			for (size_t i = 0; i < 8192; ++i)
				if (wl[4 * i]) {
					const unsigned long long wall = wl[4 * i + 3] & 0xffffffffull, start = (wl[4 * i + 3] >> 32) - t0min;
					const unsigned long long rq = (wl[4 * i + 1] >> 16) & 0xffff, tq = (wl[4 * i + 1] >> 32) + start;
					rec.push_back({wall + start, (wl[4 * i + 1] & 0xffffull) | (rq << 16) | (tq << 32), wl[4 * i + 2] & 0xffffull, wl[4 * i + 2] >> 56});
				}
			std::sort(rec.begin(), rec.end());
			if (!rec.empty()) {
				auto pr = [&](const char* tag, size_t i) {
					fprintf(stderr, "   %s: end=%.1f us, queue found empty at %.1f us, rounds=%llu (%llu after that), packets=%llu, xcc=%llu\n", tag, rec[i][0] / 100.0,
					        (rec[i][1] >> 32) / 100.0, rec[i][1] & 0xffff, (rec[i][1] & 0xffff) - ((rec[i][1] >> 16) & 0xffff), rec[i][2], rec[i][3]);
				};
				double mean = 0;
				for (auto& r : rec) mean += (double)r[0];
				mean /= rec.size();
				fprintf(stderr, "[nrs waves] n=%zu, mean end = %.1f%% of the last end\n", rec.size(), 100.0 * mean / (double)rec.back()[0]);
				pr("min", 0); pr("p25", rec.size() / 4); pr("p50", rec.size() / 2); pr("p75", rec.size() * 3 / 4); pr("p95", rec.size() * 95 / 100);
				pr("p99", rec.size() * 99 / 100); pr("max", rec.size() - 1);
			}
		}
	}
	return NRS_OK;
}

int nrs_accumulate(nrs_ctx* ctx, void* stream, uint32_t width, uint32_t height, const float* d_frame, float* d_accumulate, uint32_t sample_count, uint32_t color_space) {
	if (!ctx || !d_frame || !d_accumulate) return fail(NRS_ERR_INVALID_ARG, "nrs_accumulate: NULL argument");
	if (color_space > NRS_COLOR_VISPOSNEG) return fail(NRS_ERR_INVALID_ARG, "nrs_accumulate: color_space is 0 (Linear), 1 (SRGB) or 2 (VisPosNeg)");
	if ((uint64_t)width * height > 0xffffffffull) return fail(NRS_ERR_INVALID_ARG, "nrs_accumulate: image too large");
	HIP_TRY(hipSetDevice(ctx->device));
	NRS_TRY(launch_accumulate(width * height, d_frame, d_accumulate, sample_count, (int)color_space, stream));
	return NRS_OK;
}

int nrs_detile(nrs_ctx* ctx, void* stream, const nrs_render_params* p, uint32_t n_ranks, uint32_t tiles_per_rank_padded, const float* d_tiles,
               uint32_t channels, size_t rank_stride_floats, float* d_image) {
	if (!ctx || !p || !d_tiles || !d_image) return fail(NRS_ERR_INVALID_ARG, "nrs_detile: NULL argument");
	{ const int ac = check_params_abi(p, "nrs_detile"); if (ac != NRS_OK) return ac; }
	if (p->tile_size == 0 || p->tile_size % 8 || n_ranks == 0 || channels == 0) return fail(NRS_ERR_INVALID_ARG, "nrs_detile: bad tiling");
	const size_t dense = (size_t)tiles_per_rank_padded * p->tile_size * p->tile_size * channels;
	if (rank_stride_floats == 0) rank_stride_floats = dense;
	if (rank_stride_floats < dense) return fail(NRS_ERR_INVALID_ARG, "nrs_detile: rank stride smaller than one rank's tiles");
	HIP_TRY(hipSetDevice(ctx->device));
	NRS_TRY(launch_detile(*p, n_ranks, rank_stride_floats, d_tiles, channels, d_image, stream));
	return NRS_OK;
}

int nrs_trace_samples(nrs_model* m, const nrs_render_params* p, void* stream, uint32_t n_pixels, const uint32_t* d_pixel_idx, uint32_t max_samples,
                      float* d_t, float* d_dt, uint32_t* d_count) {
	if (!m || !p || !d_pixel_idx || !d_t || !d_dt || !d_count) return fail(NRS_ERR_INVALID_ARG, "nrs_trace_samples: NULL argument");
	if (!m->have_bitfield) return fail(NRS_ERR_STATE, "nrs_trace_samples: occupancy not set");
	{ const int pc = check_march_params(*p, "nrs_trace_samples"); if (pc != NRS_OK) return pc; }
	HIP_TRY(hipSetDevice(m->ctx->device));
	NRS_TRY(launch_trace_samples(model_for_launch(m, *p), *p, n_pixels, d_pixel_idx, max_samples, d_t, d_dt, d_count, stream));
	return NRS_OK;
}

} // extern "C"
