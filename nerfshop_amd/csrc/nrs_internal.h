// nrs_internal.h -- PODs shared between the C++ host code (nrs_api.cpp) and the HIP kernels (nrs_kernels.hip).
// Nothing here is part of the public ABI (that is include/nrs.h).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "../../include/nrs.h"

#include <stdlib.h>

namespace nrs {

// Measurement knobs (NRS_DEBUG, NRS_TEAM, NRS_RENDER_CFG, NRS_TAIL_*, NRS_RETEAM, NRS_STEAL, the logs ...) are read from the environment ONLY when NRS_DEV_KNOBS is
// set: a production process ignores them.  The two documented overrides -- NRS_CELL_CACHE_GB (nrs.h) and NRS_RCCL_LIB (the RCCL test double, announced
// on stderr) -- are not knobs of this kind.  tools/*.sh and the probes set NRS_DEV_KNOBS=1.
inline const char* dev_knob(const char* name) {
	static const bool on = []() { const char* e = getenv("NRS_DEV_KNOBS"); return e && e[0] && e[0] != '0'; }();
	return on ? getenv(name) : nullptr;
}

constexpr uint32_t kGrid = 128;                  // NERF_GRIDSIZE
constexpr uint32_t kCascades = 5;                // NERF_CASCADES
constexpr uint32_t kGridVol = kGrid * kGrid * kGrid;
constexpr uint32_t kLevels = 16;
constexpr uint32_t kCoarse = 32;                // blocks per axis of the occupancy look-ahead mask
constexpr uint32_t kCoarseWords = kCoarse * kCoarse * kCoarse / 32;
static_assert((kCoarse & (kCoarse - 1)) == 0, "the block walk's bounds test (cx | cy | cz) needs a power of two");
constexpr uint32_t kDensityW = 64 * 32 + 16 * 64;           // density MLP params (base.json:30-36)
constexpr uint32_t kRgbW = 64 * 32 + 64 * 64 + 16 * 64;     // rgb MLP params (base.json:52-58)
// The kernels evaluate ONE network shape: base.json's (density 32 -> 64 -> 16, rgb 32 -> 64 -> 64 -> 16), plus an optional third rgb hidden layer.  The
// other members of configs/nerf/'s family -- rgb network with 0 / 1 / 3 hidden layers, no rgb network at all -- are LOWERED onto it when their parameters
// arrive (lower_weights, nrs_api.cpp): matrices of 0 / +-1 that reproduce the smaller network's values exactly.  The canonical blob the MFMA fragments are
// cut from: [Wd1 64x32 | Wd2 16x64 | Wr1 64x32 | Wr2 64x64 | Wr3 16x64 | Wr2b 64x64 (third hidden layer, base_3layer.json)].
constexpr uint32_t kCanonW = kDensityW + kRgbW + 64 * 64;

// One hash-grid level as the kernels consume it (staged in LDS, 48 B).
struct LevelParams {
	float    scale;       // exp2(l*log2(b))*Nmin - 1
	uint32_t resolution;  // ceil(scale) + 1
	uint32_t res2;        // resolution^2
	uint32_t offset;      // first entry of the level in the concatenated table (in entries = u32 words)
	uint32_t count;       // entries in the level
	uint32_t hashed;      // 1: spatial hash (count is a power of two), 0: dense x + y*res + z*res^2
	uint32_t mask;        // count - 1 when hashed
	uint32_t cached;      // 1: the level has dense cell records (DeviceModel::records, nrs_model_set_cell_cache); 2: sparse brick records
	                      //    (DeviceModel::records2 + bricks, nrs_model_set_sparse_cell_cache)
	uint32_t rec_first;   // record number of cell (0, 0, 0) [dense] / of the level's first brick [sparse]
	uint32_t rec_res;     // dense: cells per side with a record (= resolution: every cell a position in [0,1]^3 can fall into); sparse: bricks per side
	uint32_t rec_res2;    // rec_res^2
	uint32_t tab_first;   // sparse: first entry of the level's brick table in DeviceModel::bricks
};

// MFMA A-operand image of the five weight matrices: kNumFrags fragments of 64 lanes x 8 halfs (1 KiB each).
// Fragment order (see nrs_mlp.cuh): D1[mb][ks] (4), D2[ks] (4), R1[mb][ks] (4), R2[mb][ks] (8), R3[ks] (4).
// Then two CONSTANT 0 / 1 fragments Sel0 / Sel1 (round 4): MFMA(Sel0, lo, 0) + MFMA(Sel1, hi, .) turn a D tile that was packed to fp16 (lo = rows of registers
// 0..7, hi = 8..15) back into fp32 accumulator registers exactly -- the fp16-accumulator model (NRS_MLP_ACC_FP16) rounds the running sum after every k step,
// and the way back from packed halfs through the matrix core costs two MFMA issues (the pipe is 10 % busy) instead of sixteen VALU conversions.
// These 26 fragments are staged into LDS.  Behind them, in HBM only: Bwd[ks] (4), the A operands of dL/dfeatures = W1^T dL/dhidden (render mode Normals),
// and R2b[mb][ks] (8), the third hidden layer of an rgb network that has one (DeviceModel::rgb_deep; base_3layer.json).
constexpr uint32_t kNumFrags = 26;
constexpr uint32_t kNumFragsDevice = 38;
constexpr uint32_t kFragBytes = 64 * 8 * 2;
constexpr uint32_t kWfragBytes = kNumFrags * kFragBytes; // 26 KiB: the LDS image
constexpr uint32_t kWfragDeviceBytes = kNumFragsDevice * kFragBytes;
// make_weight_fragments on the identity permutation (nrs_model_set_params_device): an entry is a weight index + 1 (< 0x8000), bit 15 = "negated",
// 0 = padding, or one of the two constants
constexpr uint16_t kFragOne = 0xffffu, kFragMinusOne = 0xfffeu, kFragNegate = 0x8000u;

struct Box3 { float mn[3]; float mx[3]; };

// row pitch of the multi-GPU tile index (nrs.h, nrs_render_params::tile_size): odd, so that t mod (a power-of-two number of ranks) walks diagonals
inline uint32_t tile_pitch(uint32_t width, uint32_t tile_size) { return ((width + tile_size - 1u) / tile_size) | 1u; }

// Result-preserving marching accelerator derived from the occupancy bitfield (launch_occ_accel, on the device):
//   box   world-space bounds of every occupied cell that can be consulted, slightly inflated (shortcut 1)
//   mask  kCoarse^3 bits over box: bit (z*kCoarse + y)*kCoarse + x set iff such a cell (inflated alike) overlaps that block (shortcut 2)
struct OccAccel {
	Box3 box;
	float cell[3], inv_cell[3];
	const uint32_t* mask; // device, kCoarseWords words (staged into LDS by the kernels)
};

struct DeviceModel {
	const uint32_t* grid;      // fp16x2 entries
	const uint16_t* wfrag;     // kWfragBytes
	const uint8_t*  bitfield;  // NRS_BITFIELD_BYTES
	const void*     records;   // cell records of the cached levels: 2 x uint4 = the cell's 8 corner entries, x fastest
	const void*     records2;  // sparse levels: records of the allocated bricks, 512 records (8 x 8 x 8 cells, x fastest) per brick
	const uint32_t* bricks;    // sparse levels: brick tables, entry = brick slot + 1, or 0 = no records here (hashed gathers instead)
	LevelParams     levels[kLevels];
	Box3            aabb;      // train aabb (m_aabb)
	float           inv_diag[3]; // 1 / (aabb.max - aabb.min), exact when diag_pow2
	uint32_t        diag_pow2; // every aabb extent is a power of two (always so for NGP scene boxes): x / d == x * (1/d) bit for bit
	OccAccel        occ;       // marching shortcuts (filled per launch, see model_for_launch in nrs_api.cpp)
	uint32_t        rgb_activation;
	uint32_t        density_activation;
	uint32_t        numerics;  // bit 0: nrs_grid_acc NETWORK, bit 1: nrs_mlp_acc FP16 (nrs_model_set_numerics); 0 = the default roundings
	uint32_t        rgb_deep;  // the rgb network has a third hidden layer (fragments R2b, read from wfrag in HBM): base_3layer.json
	uint32_t        no_dir;    // NerfNetworkNoDir (base_nodir.json): the direction rows of caller batches are not read
};

// AffineBoundingBox as the kernels test it (affine_bounding_box.cuh:83-88): u.(p - min) in [0, u.u) etc.
struct AffineBox { float mn[3]; float u[3]; float v[3]; float w[3]; float uu, vv, ww; float center[3]; };
enum EditKind : uint32_t { kEditCage = 0, kEditAffine = 1 };

struct DeviceEdit {
	uint32_t kind;             // EditKind
	// AffineDuplication (affine_duplication.h:92-106): warped destination / selection boxes, warped translation
	uint32_t a_hide_original, a_correct_dir, a_pad;
	AffineBox a_dst, a_sel;
	float a_translation[3], a_scale[3], a_rot[9];
	Box3 aabb;                 // scene aabb
	float inv_diag[3];         // 1 / (aabb.max - aabb.min)
	uint32_t diag_pow2;        // every extent a power of two: x / d == x * (1 / d) bit for bit (DeviceModel::diag_pow2)
	Box3 bbox;                 // deformed mesh, world units           (TetMesh::bbox)
	Box3 warped_bbox;          // bbox in warped [0,1] coordinates     (TetMesh::warped_bbox)
	Box3 orig_warped_bbox;     // canonical mesh, warped               (TetMesh::original_warped_bbox)
	const uint32_t* lut_off;
	const uint32_t* lut_idx;
	const uint32_t* tets;
	const float*    verts;
	const float*    orig;
	const float*    rot;           // nullable
	// Fine look-up table under the reference's 128^3 one (round 6; nrs_cage.hip fine_*_kernel; nullable): every LUT cell of a cascade's window is cut into 4 x 4 x 4 (or 2 x 2 x 2) fine
	// cells, and a fine cell's list is its LUT cell's list FILTERED IN ORDER -- the tets whose four face tests can pass for some position of the fine cell (a conservative
	// box-against-half-spaces test on the very floats the scan compares) -- so the first containing tet is the one the reference's scan finds, after fewer candidates.
	const uint32_t* fine_off;      // CSR offsets over the windows' fine cells, cascade after cascade
	const uint32_t* fine_idx;
	int32_t         fine_win[kCascades][8]; // per cascade: first fine cell x, y, z, first offset index | extent x, y, z in fine cells (0: no tet reaches this cascade),
	                                        // log2 of the fine cells per LUT cell and axis (0..2) or kFinePlain: this cascade keeps the LUT's own lists
	const float*    planes;        // [T x 32] one 128-byte record per tet: its 4 vertices, the 4 face normals exactly as same_side_tet forms them, the 4 sign bits of dotV4 (tet_planes_kernel)
	const uint8_t*  orig_bitfield;
	const float*    shs;           // nullable unless apply_poisson
	const float*    out_density;
	const float*    res_density;
	float           residual_amplitude;
	uint32_t        copy;
	uint32_t        apply_poisson;
	uint32_t        pad;
};

struct RenderCounters {    // zeroed before every launch
	unsigned long long n_samples;
	uint32_t next_packet;
	uint32_t n_rays_alive;
	uint32_t n_rays_hit;
	uint32_t blocks_done; // workgroups that have flushed their statistics (the last one reports to RenderArgs::feedback)
	unsigned long long phase_cycles[8]; // NRS_DEBUG & 4: per-phase wave cycles (profiling build of the kernel only)
	// NRS_DEBUG & 4: voxel-walk statistics. [0]/[1] fill: lane iterations / wave trips (= max over lanes per call);
	// [2]/[3] the same for the per-sample march; [4] sample rounds, [5] live lanes summed over rounds, [6] march calls with > 1 trip
	// [8] samples inside a cage's deformed box (they scan a LUT cell), [9] rounds with such a sample, [10] candidates tested (lane iterations of the scan),
	// [11] the scan's wave trips (max over the lanes of a round), [12] samples that found their tet
	unsigned long long walk[13];
};

struct RenderArgs {
	nrs_render_params p;
	const DeviceEdit* edits;   // device array, applied last-to-first
	int32_t  n_edits;
	uint32_t any_poisson;      // some edit has apply_poisson set
	uint32_t any_affine;       // some edit is an AffineDuplication
	uint32_t extra;            // a render mode other than Shade / Cost, show_accel or dof != 0: render_kernel's EXTRA instantiation
	uint32_t gate;             // cone stepping (aabb_scale > 1 scenes): the plain kernel's GATE instantiation (L2 phase gate on the four finest hashed levels)
	uint32_t n_packets;        // pixel packets owned by this call (8x8 pixels; 8x4 / 4x4 with lane teams of 2 / 4)
	uint32_t tiles_x;          // image width in tiles (tiled mode) or in packets (whole-image mode)
	uint32_t packets_per_tile_x;
	uint32_t pixels_owned;     // pixels this launch covers (for the feedback word)
	unsigned long long* feedback; // host-mapped: rays that found an occupied cell | pixels_owned << 32, written by the last workgroup
	uint32_t tail_every;       // hybrid launches: every tail_every-th packet row is a tail row (3)
	uint32_t reteam;           // hybrid launches: spread a wave's last rays over its idle lanes (render_kernel)
	uint32_t tail_target;      // hybrid launches: rays a wave collects per generation once it works on tail packets (24)
	uint32_t all_tail;         // hybrid launches: every packet is a 4x4 tail packet (packet_pixel<4>'s geometry, whole-image or tiles)
	uint32_t steal;            // hybrid launches: waves that run out of work take half the rays of a sibling of their workgroup (render_body, "ray hand-over")
	uint32_t fill_lanes;       // small-launch schedule (all_tail): lanes that stand on one pixel during the fill (4, 2 or 1: packets of 16 / 32 / 64 pixels)
	uint32_t p_big;            // hybrid launches (team == 0): packets [0, p_big) are 8x8, the rest 4x4 tail packets; else 0
	uint32_t team;             // 0 = hybrid (full generations, lane teams for the queue's tail); lanes per ray: 1, or 2 / 4 for launches with too few rays to fill the GPU (render_kernel's TEAM)
	uint32_t max_steps;
	uint32_t dbg;              // NRS_DEBUG ablation bits (profiling only; 0 in production): 1 = all gathers hit entry 0, 2 = skip the MLPs
	float*    frame;           // f32x4
	float*    depth;
	uint32_t* steps;           // nullable
	RenderCounters* counters;
	RenderCounters* counters_next; // the block the slot's NEXT launch will use: zeroed by this launch's last workgroup (nullable)
	unsigned long long* wave_log; // NRS_DEBUG & 4: 4 words per wave (see nrs_render_nerf)
};

// kernel launchers (nrs_kernels.hip).  stream is a hipStream_t.
int launch_render(const DeviceModel& m, const RenderArgs& a, int n_cus, void* stream);
// render mode Slice (Testbed::render_nerf's branch, testbed_nerf.cu:3111-3175): one network evaluation per owned pixel on the slice plane
int launch_slice(const DeviceModel& m, const RenderArgs& a, int n_cus, void* stream);
int launch_trace_samples(const DeviceModel& m, const nrs_render_params& p, uint32_t n_pixels, const uint32_t* d_pixel_idx,
                         uint32_t max_samples, float* d_t, float* d_dt, uint32_t* d_count, void* stream);
// mode 0: full inference (16 channels, c3 = density), 1: density MLP outputs, 2: hash-grid features [n x 32]
int launch_network(const DeviceModel& m, int mode, uint32_t n, const float* d_in, uint32_t ld_in, void* d_out, uint32_t ld_out,
                   int layout, int n_cus, void* stream);
int launch_selection_rays(const DeviceModel& m, const nrs_render_params& p, const int32_t* d_pixels, uint32_t n, float threshold,
                          float* d_positions, uint32_t* d_cells, uint8_t* d_found, void* stream);
int launch_poisson_fit(const DeviceModel& m, uint32_t n_verts, uint32_t n_sh, const float* d_coords, const void* d_net, int is_inside, float scale,
                       float* d_density, float* d_sh, void* stream);
int launch_cell_records(const DeviceModel& m, uint32_t n_levels, void* d_records, void* stream);
int launch_weight_fragments(const uint16_t* d_params, const uint16_t* d_src, uint16_t* d_frag, uint32_t n, void* stream);
constexpr uint32_t kBrick = 8, kBrickCells = kBrick * kBrick * kBrick; // sparse cell records: 8^3 cells = 16 KiB of records per brick
// Order of the 512 records inside a brick.  0: x fastest (a 128-byte line = the 4 records of a 4 x 1 x 1 run of cells); 1: Morton (a line = a 2 x 2 x 1 block, two
// lines = 2 x 2 x 2): VERDICT r3 next #7's experiment -- measured on the aabb-16 scene in round 4, see HISTORY.md (round 4) / profiles/r04_garden.md.
#ifndef NRS_BRICK_MORTON
#define NRS_BRICK_MORTON 0
#endif
// position of cell (x, y, z) (each 0..7) among its brick's records -- shared by the kernel that writes the records and the gather that reads them
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t brick_slot(uint32_t x, uint32_t y, uint32_t z) {
#if NRS_BRICK_MORTON
	auto spread = [](uint32_t v) { return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4); };
	return spread(x) | (spread(y) << 1) | (spread(z) << 2);
#else
	return (z << 6) | (y << 3) | x;
#endif
}
int launch_brick_mark(const DeviceModel& m, const LevelParams& lp, const uint8_t* d_mask, uint32_t* d_table, uint32_t* d_counter, uint32_t* d_slots, uint32_t capacity, void* stream);
int launch_brick_fill(const DeviceModel& m, const LevelParams& lp, const uint32_t* d_slots, uint32_t n_bricks, void* d_records2, void* stream);
int launch_grid_eval(const DeviceModel& m, int mode, const uint32_t res[3], const float box_mn[3], const float box_mx[3], const float dir01[3],
                     const float* d_density_grid, float* d_out, int n_cus, void* stream);
int launch_map_rays(const DeviceEdit& e, uint32_t n, float* d_coords, uint32_t ld, int with_dir, uint8_t* d_empty, void* stream);
int launch_grid_to_bitfield(const float* d_grid, uint8_t* d_bitfield, float* d_scratch_mean, void* stream);
// both flavours of the marching accelerator from the bitfield: d_masks 2 x kCoarseWords words; d_out 2 x {mn[3], mx[3], cell[3], inv_cell[3]}
// (slot 0: any step parameters, slot 1: cone_angle == 0 && min_mip == 0); d_keys: 12 words of scratch
int launch_occ_accel(const uint8_t* d_bitfield, uint32_t* d_masks, float* d_out, uint32_t* d_keys, void* stream);
// one iteration of update_density_grid_nerf_operator up to (not including) mean/bitfield; d_grid_tmp must be zeroed
int launch_grid_update(const DeviceModel& m, const DeviceEdit* d_edits, int n_edits, const nrs_grid_update& u, uint64_t rng_state_nonuniform,
                       float* d_grid, uint32_t* d_grid_tmp, int n_cus, void* stream);
int launch_accumulate(uint32_t n_pixels, const float* d_frame, float* d_accum, uint32_t sample_count, int color_space, void* stream);
int launch_detile(const nrs_render_params& p, uint32_t n_ranks, size_t rank_stride_floats, const float* d_tiles, uint32_t channels,
                  float* d_image, void* stream);
const char* launch_last_error();

// cage-move chain on the device (nrs_cage.hip)
constexpr uint32_t kLutScanTiles = kGridVol * kCascades / 4096;
constexpr uint32_t kFineMaxCells = 16u << 20; // fine look-up table: at most 16 M fine cells (64 MB of offsets) per edit; the subdivision is chosen to fit
constexpr uint32_t kFineScanTiles = kFineMaxCells / 4096;
constexpr int32_t kFinePlain = 0xff;          // DeviceEdit::fine_win[c][7]: no fine table for cascade c
constexpr int32_t kFineMaxList = 96;          // cascades whose longest LUT list exceeds this keep the plain scan (the coarse cascades of a small cage: hundreds of tets inside one cell,
                                              // hardly a sample; one wave of the build would walk such a list 64 fine cells wide -- 95 us at 285 tets)
int launch_mvc_apply(uint32_t n_points, uint32_t n_cv, const float* d_weights, const float* d_cage, float* d_points, void* stream);
int launch_bbox(uint32_t n, const float* d_verts, float* d_out6, void* stream);
int launch_poisson_interpolate(uint32_t n_points, uint32_t n_cv, const float* d_gamma, const float* d_per_cage, float* d_shs, float* d_out_density, float* d_res_density, void* stream);
int launch_lut_count_scan(uint32_t n_tets, const float* d_verts, const uint32_t* d_tets, uint32_t* d_counts, uint32_t* d_tile_sums,
                          uint32_t* d_offsets, uint32_t* d_total, unsigned long long* d_hit_masks /* [2 * n_tets * kCascades]: written */,
                          float cells0 /* estimate: cells of cascade 0 in a tet's bounding box (team size) */, void* stream);
int launch_lut_fill(uint32_t n_tets, const float* d_verts, const uint32_t* d_tets, uint32_t* d_counts, const uint32_t* d_offsets, uint32_t* d_idx,
                    uint8_t* d_bitfield, uint32_t* d_scratch_u32 /* 3 words */, uint32_t* d_big_cells, uint32_t n_work /* its entries */, unsigned long long* d_hit_masks /* as the count pass left them */, float cells0, void* stream);
uint32_t lut_big_list_capacity(size_t idx_capacity);
int launch_tet_planes(uint32_t n_tets, const float* d_verts, const uint32_t* d_tets, float* d_planes, void* stream);
// fine look-up table (DeviceEdit::fine_*): d_window_out[kCascades][8] = min x, y, z / max x, y, z of the LUT cells with a non-empty list (min > max: none), their number, the longest list
int launch_fine_window(const uint32_t* d_lut_off, int32_t* d_window_out, void* stream);
// counts (FILL = false: d_counts[n_padded], then the exclusive scan into d_fine_off[n_padded + 1] and *d_total) or fills (FILL = true: d_fine_idx) the fine lists of `de`'s windows
int launch_fine_count_scan(const DeviceEdit& de, uint32_t n_fine_cells, uint32_t* d_counts, uint32_t* d_tile_sums, uint32_t* d_fine_off, uint32_t* d_total, void* stream);
int launch_fine_fill(const DeviceEdit& de, uint32_t n_fine_cells, const uint32_t* d_fine_off, uint32_t* d_fine_idx, void* stream);
int launch_local_rotations(uint32_t n_tets, const float* d_verts, const float* d_orig, const uint32_t* d_tets, float* d_out, void* stream);
const char* cage_last_error();

// the thread-local message nrs_last_error() returns (nrs_api.cpp); used by the host-only translation units
void set_last_error(const char* msg);

} // namespace nrs
