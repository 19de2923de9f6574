// nrs_cage.hip -- the per-gizmo-move chain of a cage edit, on the device.
//
// In the reference every cage move runs on the CPU and re-uploads ~45 MB (tet_mesh.cu:651-667):
//   Cage::interpolate_with_mvc        src/editing/datastructures/cage.cu:38        V_tet x V_cage weights applied to the cage
//   TetMesh::post_update_vertices     tet_mesh.cu:12-20                            bbox of the deformed vertices
//   TetMesh::build_tet_grid           tet_mesh.cu:368-673                          cell -> tet CSR over 5 x 128^3 cells
//   TetMesh::update_local_rotations   tet_mesh.cu:37-74                            per-tet rotation for the view direction
// Here the tables never leave HBM: count -> scan -> fill -> per-cell sort, all integer-exact against the host builder
// (nrs_authoring.cpp) and the oracle (oracle/nrs_oracle.cpp build_tet_lut): same float tests in the same operation order
// (-ffp-contract=off), and each cell lists its tets in ascending index, the order the reference's merge produces.
//
// Kernels are HBM/latency-bound integer work: one wave per (tet, cascade) strides over the cells of the tet's bounding
// box; the CSR scan streams the 42 MB count array twice.  Nothing here is GEMM-shaped.
#include <hip/hip_runtime.h>
#include "nrs_internal.h"
#include "nrs_device.cuh"
#include "nrs_svd3.h"

namespace nrs {

static thread_local char g_cage_err[512];
const char* cage_last_error() { return g_cage_err; }
#define NRS_CAGE_CHECK(what)                                                                          \
	do {                                                                                              \
		hipError_t e_ = hipGetLastError();                                                            \
		if (e_ != hipSuccess) {                                                                       \
			snprintf(g_cage_err, sizeof(g_cage_err), "%s: %s", what, hipGetErrorString(e_));          \
			return NRS_ERR_HIP;                                                                       \
		}                                                                                             \
	} while (0)

constexpr uint32_t kCells = kGridVol * kCascades;
constexpr uint32_t kScanTile = 4096; // cells per scan block: 256 threads x 16
static_assert(kCells % kScanTile == 0, "scan tiles must cover the cell array exactly");
constexpr uint32_t kScanTiles = kCells / kScanTile; // 2560
static_assert(kFineMaxCells % kScanTile == 0 && kFineScanTiles == kFineMaxCells / kScanTile, "fine look-up table: scan tiles");

// ---- Cage::interpolate_with_mvc (cage.cu:38-49): points[i] = sum_v w[i][v] * cage[v], v ascending, per component ----------
__global__ void mvc_apply_kernel(uint32_t n_points, uint32_t n_cv, const float* __restrict__ weights, const float* __restrict__ cage,
                                 float* __restrict__ points) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_points) return;
	f3 p = mk3(0.f, 0.f, 0.f);
	const float* w = weights + (size_t)i * n_cv;
	for (uint32_t v = 0; v < n_cv; ++v) {
		const float wv = w[v];
		p = p + mk3(wv * cage[3 * v], wv * cage[3 * v + 1], wv * cage[3 * v + 2]);
	}
	points[3 * i] = p.x; points[3 * i + 1] = p.y; points[3 * i + 2] = p.z;
}

// ---- GrowingSelection::interpolate_poisson_boundary (growing_selection.cu:2350-2395): cage-vertex membrane terms -> tet vertices ---------------
// thread per tet vertex; cage vertices in ascending order => the reference's float sums.  per_cage = [n_cv x 30]: alpha_out, outside density,
// outside - inside density, 27 x sh_diff (prepared on the host with the host libm's expf, as the reference computes them).
__global__ void poisson_interpolate_kernel(uint32_t n_points, uint32_t n_cv, const float* __restrict__ gamma, const float* __restrict__ per_cage,
                                           float* __restrict__ shs, float* __restrict__ out_density, float* __restrict__ res_density) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_points) return;
	float sh[27];
	#pragma unroll
	for (int k = 0; k < 27; ++k) sh[k] = 0.f;
	float wsum = 0.f, od = 0.f, rd = 0.f;
	const float* g = gamma + (size_t)i * n_cv;
	for (uint32_t j = 0; j < n_cv; ++j) {
		const float* c = per_cage + 30 * (size_t)j;
		const float ga = g[j] * c[0];
		wsum += ga;
		#pragma unroll
		for (int k = 0; k < 27; ++k) sh[k] += ga * c[3 + k];
		od += g[j] * c[1];
		rd += g[j] * c[2];
	}
	const float denom = (float)((double)wsum + 1e-6);
	#pragma unroll
	for (int k = 0; k < 27; ++k) shs[27 * (size_t)i + k] = sh[k] / denom;
	out_density[i] = od;
	res_density[i] = fmaxf(rd, 0.f);
}

// ---- bbox of the vertices (min/max are order-independent => exact) -> out[0..2] = min, out[3..5] = max ----------------------
__global__ void bbox_kernel(uint32_t n, const float* __restrict__ v, float* __restrict__ out) {
	__shared__ float lo[3][256], hi[3][256];
	const float inf = __builtin_huge_valf();
	float l[3] = {inf, inf, inf}, h[3] = {-inf, -inf, -inf};
	for (uint32_t i = threadIdx.x; i < n; i += 256)
		for (int k = 0; k < 3; ++k) { l[k] = fminf(l[k], v[3 * i + k]); h[k] = fmaxf(h[k], v[3 * i + k]); }
	for (int k = 0; k < 3; ++k) { lo[k][threadIdx.x] = l[k]; hi[k][threadIdx.x] = h[k]; }
	__syncthreads();
	for (int s = 128; s > 0; s >>= 1) {
		if ((int)threadIdx.x < s)
			for (int k = 0; k < 3; ++k) {
				lo[k][threadIdx.x] = fminf(lo[k][threadIdx.x], lo[k][threadIdx.x + s]);
				hi[k][threadIdx.x] = fmaxf(hi[k][threadIdx.x], hi[k][threadIdx.x + s]);
			}
		__syncthreads();
	}
	if (threadIdx.x < 3) { out[threadIdx.x] = lo[threadIdx.x][0]; out[3 + threadIdx.x] = hi[threadIdx.x][0]; }
}

// ---- cell / tet intersection tests of build_tet_grid (tet_mesh.cu:402-470) -------------------------------------------------
__device__ __forceinline__ float comp3(const f3& p, int i) { return i == 0 ? p.x : (i == 1 ? p.y : p.z); }
// (round 6: everything below is inlined and fully unrolled -- the point lists are compile-time sized, so the corner / vertex arrays live in registers instead of the
// 12 KiB of LDS the compiler parked them in, and the per-face normals of the eight point-in-tet tests are computed once: same operations on the same operands in
// the same order, a cell / tet test ~4 x shorter; the LUT pass of a cage move is a chain of such tests.  -ffp-contract=off: unrolling fuses nothing.)
template <int N>
__device__ __forceinline__ void span3(const f3 (&pts)[N], f3 axis, float& lo, float& hi) {
	lo = __builtin_huge_valf();
	hi = -lo;
	#pragma unroll
	for (int i = 0; i < N; ++i) {
		const float v = dot3(axis, pts[i]);
		if (v < lo) lo = v;
		if (v > hi) hi = v;
	}
}
// BoundingBox::intersects(Triangle), bounding_box.cuh:126-178: separating axes = 3 box normals, the triangle normal, 9 edge x axis
__device__ __forceinline__ bool cube_hits_triangle(f3 bmin, f3 bmax, f3 a, f3 b, f3 c) {
	const f3 axes[3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
	const f3 tri[3] = {a, b, c};
	float tlo, thi, blo, bhi;
	#pragma unroll
	for (int i = 0; i < 3; ++i) {
		span3(tri, axes[i], tlo, thi);
		if (thi < comp3(bmin, i) || tlo > comp3(bmax, i)) return false;
	}
	f3 n = cross3(b - a, c - a);
	const float len = sqrtf(dot3(n, n));
	n = {n.x / len, n.y / len, n.z / len};
	const f3 corners[8] = {{bmin.x, bmin.y, bmin.z}, {bmin.x, bmin.y, bmax.z}, {bmin.x, bmax.y, bmin.z}, {bmin.x, bmax.y, bmax.z},
	                       {bmax.x, bmin.y, bmin.z}, {bmax.x, bmin.y, bmax.z}, {bmax.x, bmax.y, bmin.z}, {bmax.x, bmax.y, bmax.z}};
	const float off = dot3(n, a);
	span3(corners, n, blo, bhi);
	if (bhi < off || blo > off) return false;
	const f3 edges[3] = {a - b, a - c, b - c};
	#pragma unroll
	for (int i = 0; i < 3; ++i) {
		#pragma unroll
		for (int j = 0; j < 3; ++j) {
			const f3 ax = cross3(edges[i], axes[j]);
			span3(corners, ax, blo, bhi);
			span3(tri, ax, tlo, thi);
			if (bhi < tlo || blo > thi) return false;
		}
	}
	return true;
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((uint32_t)(127 + e) << 23); } // scalbnf(1, e), |e| <= 4
__device__ __forceinline__ void cell_of(f3 p, uint32_t level, int out[3]) { // get_cell_at_pos, selection_utils.cu:70-83
	const float s = pow2f(-(int)level);
	p = p - mk3(0.5f, 0.5f, 0.5f);
	p = p * s;
	p = p + mk3(0.5f, 0.5f, 0.5f);
	out[0] = clampi_((int)(p.x * (float)kGrid), 0, kGrid - 1);
	out[1] = clampi_((int)(p.y * (float)kGrid), 0, kGrid - 1);
	out[2] = clampi_((int)(p.z * (float)kGrid), 0, kGrid - 1);
}
__device__ __forceinline__ f3 cell_centre(uint32_t x, uint32_t y, uint32_t z, uint32_t level) { // get_cell_pos, selection_utils.cu:65-68
	const float s = pow2f((int)level);
	return {(((float)x + 0.5f) / (float)kGrid - 0.5f) * s + 0.5f, (((float)y + 0.5f) / (float)kGrid - 0.5f) * s + 0.5f,
	        (((float)z + 0.5f) / (float)kGrid - 0.5f) * s + 0.5f};
}
__device__ __forceinline__ bool cell_meets_tet(f3 t0, f3 t1, f3 t2, f3 t3, uint32_t x, uint32_t y, uint32_t z, uint32_t level) { // (four values, not an array: an indexed private array is parked in LDS)
	const float kC[8][3] = {{-0.5f, -0.5f, -0.5f}, {-0.5f, -0.5f, 0.5f}, {-0.5f, 0.5f, -0.5f}, {0.5f, -0.5f, -0.5f},
	                        {0.5f, 0.5f, -0.5f}, {-0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.5f, 0.5f, 0.5f}}; // tet_mesh.h:34-43
	const float cell = pow2f((int)level) * (1.0f / (float)kGrid);
	const f3 ctr = cell_centre(x, y, z, level);
	#pragma unroll
	for (int k = 0; k < 8; ++k)
		if (point_in_tet(t0, t1, t2, t3, ctr + mk3(kC[k][0], kC[k][1], kC[k][2]) * cell)) return true;
	const f3 h = mk3(0.5f, 0.5f, 0.5f) * cell;
	const f3 a = ctr - h, b = ctr + h;
	const f3 bmin = {fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)};
	const f3 bmax = {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)};
	// faces (j, j + 1, j + 2) mod 4, j = 0..3 (tet_mesh.cu:441-452)
	if (cube_hits_triangle(bmin, bmax, t0, t1, t2)) return true;
	if (cube_hits_triangle(bmin, bmax, t1, t2, t3)) return true;
	if (cube_hits_triangle(bmin, bmax, t2, t3, t0)) return true;
	if (cube_hits_triangle(bmin, bmax, t3, t0, t1)) return true;
	return false;
}

// T lanes per (tet, cascade) item, 64 / T items per wave; a team strides over the cells of its tet's bounding box at that cascade.  Items are numbered cascade-major
// (item = cascade * n_tets + tet); a segment of a launch covers the items [item_begin, item_end) with ONE team size (launch_tet_mark chooses: 64 / 8 / 1).  Round 6 (profiles/r06/cage_move_kernels.md): one WAVE per item left 56-63 lanes
// idle, and in the coarse cascades -- where the whole mesh stands in a handful of cells -- every item's atomic went to the same address (6 000 serialised atomics per
// cascade and pass: that, not the tests, was the pass's time); now the hits of a wave that fall into its first hit's cell share ONE atomic.
// FILL == false: counts[cell] += 1, and the item's first 128 test results go to hit_masks[2 item .. 2 item + 1] (bit k = cell k of the box).
// FILL == true: idx[offsets[cell] + --counts[cell]] = tet (leaves counts zeroed; the order inside a list is free: lut_finish_kernel sorts); the first 128 cells take the
// count pass's answers instead of repeating cell_meets_tet (8 point-in-tet + 4 box / triangle tests per cell).
template <bool FILL, int T>
__device__ __forceinline__ void tet_mark_wave(uint32_t wv, uint32_t n_tets, uint32_t item_begin, uint32_t item_end, const float* __restrict__ verts, const uint32_t* __restrict__ tets,
                                              uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets,
                                              uint32_t* __restrict__ idx, unsigned long long* __restrict__ hit_masks) {
	const int lane = threadIdx.x & 63, sub = lane / T, r = lane % T;
	const uint32_t item = item_begin + wv * (64 / T) + (uint32_t)sub;
	const bool live = item < item_end;
	const uint32_t level = live ? item / n_tets : 0u, t = live ? item % n_tets : 0u;
	const uint4 tvid = reinterpret_cast<const uint4*>(tets)[t];
	const f3 t0 = ld3(verts, tvid.x), t1 = ld3(verts, tvid.y), t2 = ld3(verts, tvid.z), t3 = ld3(verts, tvid.w);
	f3 lo = t0, hi = t0;
	lo = {fminf(lo.x, t1.x), fminf(lo.y, t1.y), fminf(lo.z, t1.z)}; hi = {fmaxf(hi.x, t1.x), fmaxf(hi.y, t1.y), fmaxf(hi.z, t1.z)};
	lo = {fminf(lo.x, t2.x), fminf(lo.y, t2.y), fminf(lo.z, t2.z)}; hi = {fmaxf(hi.x, t2.x), fmaxf(hi.y, t2.y), fmaxf(hi.z, t2.z)};
	lo = {fminf(lo.x, t3.x), fminf(lo.y, t3.y), fminf(lo.z, t3.z)}; hi = {fmaxf(hi.x, t3.x), fmaxf(hi.y, t3.y), fmaxf(hi.z, t3.z)};
	int c0[3], c1[3];
	cell_of(lo, level, c0);
	cell_of(hi, level, c1);
	const uint32_t ny = (uint32_t)(c1[1] - c0[1] + 1), nz = (uint32_t)(c1[2] - c0[2] + 1);
	const uint32_t total = live ? (uint32_t)(c1[0] - c0[0] + 1) * ny * nz : 0u;
	unsigned long long mask0 = (FILL && live) ? hit_masks[2 * (size_t)item] : 0ull, mask1 = (FILL && live) ? hit_masks[2 * (size_t)item + 1] : 0ull; // cells 0..63, 64..127 of the box
	for (uint32_t it = 0;; ++it) { // (wave-uniform trip count: the longest box of the wave's items)
		const uint32_t k = it * T + (uint32_t)r;
		const bool in = k < total;
		if (!__any(in)) break;
		const uint32_t x = (uint32_t)c0[0] + k / (ny * nz), y = (uint32_t)c0[1] + (k / nz) % ny, z = (uint32_t)c0[2] + k % nz;
		bool hit = false;
		if (in) hit = (FILL && k < 128u) ? (((k < 64u ? mask0 : mask1) >> (k & 63u)) & 1ull) != 0ull : cell_meets_tet(t0, t1, t2, t3, x, y, z, level);
		const unsigned long long votes = __ballot(hit);
		if (!FILL && it < 128u / T) { // (it is wave-uniform; a team's T votes of one trip never straddle the two words: T divides 64)
			const unsigned long long mine = ((votes >> (sub * T)) & ((T == 64 ? 0ull : (1ull << T)) - 1ull)) << ((it * T) & 63u);
			if (it * T < 64u) mask0 |= mine; else mask1 |= mine;
		}
		if (votes == 0ull) continue;
		// the hits that stand in the wave's first hit's cell take one atomic together; the others their own
		const uint32_t cell = level * kGridVol + morton3D(x, y, z);
		const int leader = __ffsll((long long)votes) - 1;
		const uint32_t lead_cell = (uint32_t)__shfl((int)cell, leader, 64);
		const bool grouped = hit && cell == lead_cell;
		const unsigned long long group = __ballot(grouped);
		const uint32_t n_group = (uint32_t)__popcll(group);
		uint32_t base = 0;
		if (lane == leader) {
			if (!FILL) atomicAdd(counts + lead_cell, n_group);
			else base = atomicSub(counts + lead_cell, n_group);
		}
		if (FILL) {
			base = (uint32_t)__shfl((int)base, leader, 64);
			const uint32_t rank = (uint32_t)__popcll(group & ((1ull << lane) - 1ull));
			if (grouped) idx[offsets[cell] + (base - 1u - rank)] = t;
		}
		if (hit && !grouped) {
			if (!FILL) atomicAdd(counts + cell, 1u);
			else idx[offsets[cell] + (atomicSub(counts + cell, 1u) - 1u)] = t;
		}
	}
	if (!FILL && live && r == 0) { hit_masks[2 * (size_t)item] = mask0; hit_masks[2 * (size_t)item + 1] = mask1; }
}
// One launch per pass: up to three segments of items, each with its own team size (T0 = 64 or 8 for the first, 8 for the second, 1 for the third; an empty segment has
// no blocks), so that the segments' waves share the GPU instead of queueing behind each other (two launches of a 6 000-tet cage's count pass: 0.08 + 0.08 ms).
struct MarkSegments { uint32_t item_begin[3], item_end[3], first_block[4]; };
template <bool FILL, int T0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void tet_mark_kernel(uint32_t n_tets, MarkSegments sg, const float* __restrict__ verts, const uint32_t* __restrict__ tets,
                                                        uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets,
                                                        uint32_t* __restrict__ idx, unsigned long long* __restrict__ hit_masks) {
	const uint32_t wave = threadIdx.x >> 6;
	if (blockIdx.x < sg.first_block[1]) tet_mark_wave<FILL, T0>(blockIdx.x * 4u + wave, n_tets, sg.item_begin[0], sg.item_end[0], verts, tets, counts, offsets, idx, hit_masks);
	else if (blockIdx.x < sg.first_block[2]) tet_mark_wave<FILL, 8>((blockIdx.x - sg.first_block[1]) * 4u + wave, n_tets, sg.item_begin[1], sg.item_end[1], verts, tets, counts, offsets, idx, hit_masks);
	else tet_mark_wave<FILL, 1>((blockIdx.x - sg.first_block[2]) * 4u + wave, n_tets, sg.item_begin[2], sg.item_end[2], verts, tets, counts, offsets, idx, hit_masks);
}
// Team sizes per cascade.  A small mesh (a pass is as long as its longest wave: the GPU is not full) takes the widest teams its boxes fill -- a wave per item at cascade 0
// when a tet's box there holds >= 32 cells (`cells0`: the caller's estimate from the mesh's bounding box and tet count), eight lanes elsewhere; a large mesh (throughput)
// eight lanes at the two finest cascades and one lane per item at the coarse ones, where a tet touches one or two cells.
template <bool FILL>
static void launch_tet_mark(uint32_t n_tets, float cells0, const float* d_verts, const uint32_t* d_tets, uint32_t* d_counts, const uint32_t* d_offsets, uint32_t* d_idx,
                            unsigned long long* d_hit_masks, hipStream_t s) {
	const bool small = n_tets <= 16384u, wide0 = small && cells0 >= 32.f;
	const uint32_t n = n_tets, all = n_tets * kCascades;
	MarkSegments sg;
	const uint32_t bounds[4] = {0u, n, small ? all : 2u * n, all}; // segment k: items [bounds[k], bounds[k + 1])
	const uint32_t per_wave[3] = {wide0 ? 1u : 8u, 8u, 64u};       // items per wave (64 / team size)
	uint32_t block = 0;
	for (int k = 0; k < 3; ++k) {
		sg.item_begin[k] = bounds[k]; sg.item_end[k] = bounds[k + 1]; sg.first_block[k] = block;
		const uint32_t waves = (bounds[k + 1] - bounds[k] + per_wave[k] - 1u) / per_wave[k];
		block += (waves + 3u) / 4u;
	}
	sg.first_block[3] = block;
	if (wide0) hipLaunchKernelGGL((tet_mark_kernel<FILL, 64>), dim3(block), dim3(256), 0, s, n_tets, sg, d_verts, d_tets, d_counts, d_offsets, d_idx, d_hit_masks);
	else hipLaunchKernelGGL((tet_mark_kernel<FILL, 8>), dim3(block), dim3(256), 0, s, n_tets, sg, d_verts, d_tets, d_counts, d_offsets, d_idx, d_hit_masks);
}

// ---- exclusive scan of counts[kCells] -> offsets[kCells + 1] ----------------------------------------------------------------
__global__ __launch_bounds__(256) void scan_tile_sum_kernel(const uint32_t* __restrict__ counts, uint32_t* __restrict__ tile_sums) {
	__shared__ uint32_t part[256];
	const uint4* src = reinterpret_cast<const uint4*>(counts + (size_t)blockIdx.x * kScanTile) + threadIdx.x * 4;
	uint32_t s = 0;
	#pragma unroll
	for (int q = 0; q < 4; ++q) { const uint4 v = src[q]; s += v.x + v.y + v.z + v.w; }
	part[threadIdx.x] = s;
	__syncthreads();
	for (int st = 128; st > 0; st >>= 1) {
		if ((int)threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
		__syncthreads();
	}
	if (threadIdx.x == 0) tile_sums[blockIdx.x] = part[0];
}
// one block: tile_sums -> exclusive prefix in place; total -> offsets[kCells] and scratch_total
constexpr uint32_t kMaxScanTiles = 4096; // (the cell -> tet LUT scans 2560 tiles, the fine look-up table up to kFineScanTiles = 4096)
__global__ __launch_bounds__(1024) void scan_tile_prefix_kernel(uint32_t* __restrict__ tile_sums, uint32_t* __restrict__ offsets_last,
                                                                 uint32_t* __restrict__ total_out, uint32_t n_tiles) {
	__shared__ uint32_t part[1024];
	constexpr uint32_t per = (kMaxScanTiles + 1023) / 1024;
	uint32_t v[per], s = 0;
	const uint32_t kScanTiles = n_tiles; // (<= kMaxScanTiles)
	#pragma unroll
	for (uint32_t q = 0; q < per; ++q) {
		const uint32_t i = threadIdx.x * per + q;
		v[q] = i < kScanTiles ? tile_sums[i] : 0u;
		s += v[q];
	}
	part[threadIdx.x] = s;
	__syncthreads();
	for (uint32_t d = 1; d < 1024; d <<= 1) { // Hillis-Steele inclusive scan
		const uint32_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
		__syncthreads();
		part[threadIdx.x] += add;
		__syncthreads();
	}
	uint32_t run = part[threadIdx.x] - s;
	#pragma unroll
	for (uint32_t q = 0; q < per; ++q) {
		const uint32_t i = threadIdx.x * per + q;
		if (i < kScanTiles) tile_sums[i] = run;
		run += v[q];
	}
	if (threadIdx.x == 1023) { *offsets_last = part[1023]; *total_out = part[1023]; }
}
__global__ __launch_bounds__(256) void scan_write_kernel(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ tile_prefix,
                                                          uint32_t* __restrict__ offsets) {
	__shared__ uint32_t part[256];
	const size_t base = (size_t)blockIdx.x * kScanTile + threadIdx.x * 16;
	const uint4* src = reinterpret_cast<const uint4*>(counts + base);
	uint32_t c[16], s = 0;
	#pragma unroll
	for (int q = 0; q < 4; ++q) { const uint4 v = src[q]; c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w; }
	#pragma unroll
	for (int q = 0; q < 16; ++q) s += c[q];
	part[threadIdx.x] = s;
	__syncthreads();
	for (uint32_t d = 1; d < 256; d <<= 1) {
		const uint32_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
		__syncthreads();
		part[threadIdx.x] += add;
		__syncthreads();
	}
	uint32_t run = tile_prefix[blockIdx.x] + part[threadIdx.x] - s;
	uint4* dst = reinterpret_cast<uint4*>(offsets + base);
	#pragma unroll
	for (int q = 0; q < 4; ++q) {
		uint4 o;
		o.x = run; run += c[4 * q];
		o.y = run; run += c[4 * q + 1];
		o.z = run; run += c[4 * q + 2];
		o.w = run; run += c[4 * q + 3];
		dst[q] = o;
	}
}

// ---- per byte of the touched-cell bitfield: sort each of its 8 cells' tet lists ascending, emit the byte, track the maximum ----
// Lists of up to kSmallList tets are sorted by the thread (insertion sort); longer ones -- the few coarse-cascade cells that
// contain most of the mesh -- go to a worklist that lut_sort_big_kernel sorts with one workgroup each.
constexpr uint32_t kSmallList = 24;
constexpr uint32_t kMidList = 128; // lists of kSmallList + 1 .. kMidList tets: sorted by one wave in LDS (sixteen cells per workgroup at a time, no workgroup barrier); longer ones by a workgroup
// (measured with 1024: the longest list of a wave sets the kernel's time, 0.10 ms per move)
__global__ __launch_bounds__(256) void lut_finish_kernel(const uint32_t* __restrict__ offsets, uint32_t* __restrict__ idx,
                                                          uint8_t* __restrict__ bitfield, uint32_t* __restrict__ max_per_cell,
                                                          uint32_t* __restrict__ big_cells, uint32_t* __restrict__ n_big, uint32_t n_work) {
	// (round 6: one thread per CELL, not per byte of eight cells -- the kernel is as slow as its slowest thread, and that thread's work is a chain of dependent
	// memory trips; the list is fetched eight entries per trip into a per-thread row of LDS, sorted there and written back; the maximum is reduced per wave --
	// one atomicMax per non-empty cell was ~10^5 atomics on ONE address per cage move.  0.40 -> 0.0x ms per move: profiles/r06/cage_move_kernels.md)
	__shared__ uint32_t rows[256][kSmallList + 1]; // (+1: consecutive threads' rows start in consecutive banks)
	const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; // kCells threads, grid sized exactly
	const uint32_t o0 = offsets[c], n = offsets[c + 1] - o0;
	if (n > kMidList) {
		big_cells[atomicAdd(n_big, 1u)] = c;                      // a workgroup per cell (lut_sort_big_kernel): from the front of the worklist
	} else if (n > kSmallList) {
		big_cells[n_work - 1u - atomicAdd(n_big + 1, 1u)] = c;     // a wave per cell (lut_sort_mid_kernel): from its back (n_big[1] = their number)
	} else if (n > 1) {
		uint32_t* row = rows[threadIdx.x];
		uint32_t* a = idx + o0;
		for (uint32_t i = 0; i < n; i += 8) { // (eight independent loads per trip; indices clamped, the surplus is not stored)
			uint32_t v[8];
			#pragma unroll
			for (uint32_t q = 0; q < 8; ++q) v[q] = a[min(i + q, n - 1u)];
			#pragma unroll
			for (uint32_t q = 0; q < 8; ++q) if (i + q < n) row[i + q] = v[q];
		}
		for (uint32_t i = 1; i < n; ++i) {
			const uint32_t key = row[i];
			uint32_t j = i;
			while (j > 0 && row[j - 1] > key) { row[j] = row[j - 1]; --j; }
			row[j] = key;
		}
		for (uint32_t i = 0; i < n; ++i) a[i] = row[i];
	}
	if (bitfield) { // the touched-cell byte of eight consecutive cells: their lanes' votes
		const unsigned long long votes = __ballot(n != 0u);
		if ((threadIdx.x & 7u) == 0u) bitfield[c >> 3] = (uint8_t)((votes >> (threadIdx.x & 63u)) & 0xffull);
	}
	uint32_t mx = n;
	for (int sh = 32; sh > 0; sh >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, sh, 64));
	if ((threadIdx.x & 63) == 0 && mx) atomicMax(max_per_cell, mx);
}

// Normalised bitonic network (every comparator puts the smaller key at the lower index): valid for any n, because the
// virtual +inf keys at positions >= n never have to move.  Stage k: first step pairs i with its mirror i ^ (k - 1), the
// remaining steps pair i with i ^ j for j = k/4 .. 1.
template <typename Ptr>
__device__ __forceinline__ void bitonic_ascending(Ptr a, uint32_t n, uint32_t tid, uint32_t n_threads) {
	uint32_t p2 = 1;
	while (p2 < n) p2 <<= 1;
	for (uint32_t k = 2; k <= p2; k <<= 1) {
		for (uint32_t j = k >> 1; j > 0; j >>= 1) {
			const uint32_t mask = (j == (k >> 1)) ? (k - 1) : j;
			for (uint32_t i = tid; i < p2; i += n_threads) {
				const uint32_t l = i ^ mask;
				if (l > i && l < n) {
					const uint32_t x = a[i], y = a[l];
					if (x > y) { a[i] = y; a[l] = x; }
				}
			}
			__syncthreads();
		}
	}
}
// The same network run by ONE wave on an LDS array: the lanes of a wave execute in lockstep and a wave's LDS operations stay in order, so a step needs no
// workgroup barrier -- only the compiler must not move one step's reads above the previous step's writes of other lanes (fences at wavefront scope).
__device__ __forceinline__ void bitonic_ascending_wave(uint32_t* a, uint32_t n, uint32_t lane) {
	uint32_t p2 = 1;
	while (p2 < n) p2 <<= 1;
	for (uint32_t k = 2; k <= p2; k <<= 1) {
		for (uint32_t j = k >> 1; j > 0; j >>= 1) {
			const uint32_t mask = (j == (k >> 1)) ? (k - 1) : j;
			for (uint32_t i = lane; i < p2; i += 64) {
				const uint32_t l = i ^ mask;
				if (l > i && l < n) {
					const uint32_t x = a[i], y = a[l];
					if (x > y) { a[i] = y; a[l] = x; }
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		}
	}
}
constexpr uint32_t kSortLdsEntries = 16384; // 64 KiB of LDS per workgroup
// the middle worklist (kSmallList < n <= kMidList; it grows from the BACK of big_cells: entry n_work - 1 - i): one wave per cell
__global__ __launch_bounds__(1024) void lut_sort_mid_kernel(const uint32_t* __restrict__ offsets, uint32_t* __restrict__ idx,
                                                             const uint32_t* __restrict__ big_cells, const uint32_t* __restrict__ n_mid, uint32_t n_work) {
	__shared__ uint32_t sh[16 * kMidList]; // sixteen waves, one list each
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	uint32_t* a = sh + wave * kMidList;
	const uint32_t nm = *n_mid;
	for (uint32_t w = blockIdx.x * 16u + wave; w < nm; w += gridDim.x * 16u) {
		const uint32_t cell = big_cells[n_work - 1u - w];
		const uint32_t o = offsets[cell], n = offsets[cell + 1] - o;
		for (uint32_t i = lane; i < n; i += 64) a[i] = idx[o + i];
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		bitonic_ascending_wave(a, n, lane);
		for (uint32_t i = lane; i < n; i += 64) idx[o + i] = a[i];
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
}
constexpr uint32_t kSortScanWords = kSortLdsEntries - 1024u; // bitmap path: the bitmap's words in front, the 1024 scan partials behind them
__global__ __launch_bounds__(1024) void lut_sort_big_kernel(const uint32_t* __restrict__ offsets, uint32_t* __restrict__ idx,
                                                             const uint32_t* __restrict__ big_cells, const uint32_t* __restrict__ n_big, uint32_t n_tets) {
	__shared__ uint32_t sh[kSortLdsEntries];
	const uint32_t nb = *n_big;
	for (uint32_t w = blockIdx.x; w < nb; w += gridDim.x) {
		const uint32_t cell = big_cells[w];
		const uint32_t o = offsets[cell], n = offsets[cell + 1] - o;
		if ((n_tets + 31u) / 32u <= kSortScanWords) { // (every list of this worklist is longer than kMidList)
			// a long list (the coarse cascades' cells hold thousands of a large mesh's tets).  A list holds distinct tet numbers below n_tets: set their bits in an LDS
			// bitmap, scan the words' populations, write the set bits out in ascending order -- O(n + n_tets / 32) instead of a bitonic network's O(n log^2 n) with a
			// barrier per step (round 6: profiles/r06/cage_move_kernels.md)
			const uint32_t words = (n_tets + 31u) / 32u, per = (words + 1023u) / 1024u;
			uint32_t* part = sh + kSortScanWords;
			for (uint32_t i = threadIdx.x; i < words; i += 1024) sh[i] = 0u;
			__syncthreads();
			for (uint32_t i = threadIdx.x; i < n; i += 1024) { const uint32_t t = idx[o + i]; atomicOr(&sh[t >> 5], 1u << (t & 31u)); }
			__syncthreads();
			const uint32_t w0 = min(threadIdx.x * per, words), w1 = min(w0 + per, words);
			uint32_t mine = 0;
			for (uint32_t ww = w0; ww < w1; ++ww) mine += (uint32_t)__popc(sh[ww]);
			part[threadIdx.x] = mine;
			__syncthreads();
			for (uint32_t d = 1; d < 1024; d <<= 1) { // Hillis-Steele inclusive scan
				const uint32_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
				__syncthreads();
				part[threadIdx.x] += add;
				__syncthreads();
			}
			uint32_t pos = o + part[threadIdx.x] - mine;
			for (uint32_t ww = w0; ww < w1; ++ww) {
				uint32_t bits = sh[ww];
				while (bits) { idx[pos++] = ww * 32u + (uint32_t)__builtin_ctz(bits); bits &= bits - 1u; }
			}
			__syncthreads();
		} else if (n <= kSortLdsEntries) {
			for (uint32_t i = threadIdx.x; i < n; i += 1024) sh[i] = idx[o + i];
			__syncthreads();
			bitonic_ascending(sh, n, threadIdx.x, 1024u);
			for (uint32_t i = threadIdx.x; i < n; i += 1024) idx[o + i] = sh[i];
			__syncthreads();
		} else {
			bitonic_ascending(idx + o, n, threadIdx.x, 1024u); // in HBM/L2: a mesh of more than 491 520 tets with more than 16 K of them in ONE cell
		}
	}
}

// ---- TetMesh::update_local_rotations (tet_mesh.cu:37-74): R = U V^T from the reference's approximate SVD (nrs_svd3.h) -------------
__global__ void local_rotations_kernel(uint32_t n_tets, const float* __restrict__ def, const float* __restrict__ org,
                                       const uint32_t* __restrict__ tets, float* __restrict__ out) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_tets) return;
	const uint4 tv = reinterpret_cast<const uint4*>(tets)[i];
	const uint32_t id[4] = {tv.x, tv.y, tv.z, tv.w};
	float o[4][3], d[4][3];
	for (int j = 0; j < 4; ++j)
		for (int k = 0; k < 3; ++k) { o[j][k] = org[3 * id[j] + k]; d[j][k] = def[3 * id[j] + k]; }
	float R[9];
	svd3::tet_rotation(o, d, R);
	for (int k = 0; k < 9; ++k) out[9 * (size_t)i + k] = R[k];
}

// ---- per-tet face planes for point_in_tet_planes (nrs_device.cuh): the tet-only half of same_side_tet, selection_utils.h:33-39 ----
__global__ void tet_planes_kernel(uint32_t n_tets, const float* __restrict__ verts, const uint32_t* __restrict__ tets, float* __restrict__ planes) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_tets) return;
	const uint4 tv = reinterpret_cast<const uint4*>(tets)[i];
	const f3 v[4] = {ld3(verts, tv.x), ld3(verts, tv.y), ld3(verts, tv.z), ld3(verts, tv.w)};
	float out[32];
	uint32_t signs = 0;
	for (int f = 0; f < 4; ++f) { // point_in_tet's calls: (v1,v2,v3,v4), (v2,v3,v4,v1), (v3,v4,v1,v2), (v4,v1,v2,v3)
		const f3 v1 = v[f], v2 = v[(f + 1) & 3], v3 = v[(f + 2) & 3], v4 = v[(f + 3) & 3];
		const f3 normal = cross3(v2 - v1, v3 - v1);
		const float dotV4 = dot3(normal, v4 - v1);
		out[3 * f] = v1.x; out[3 * f + 1] = v1.y; out[3 * f + 2] = v1.z;
		out[12 + 3 * f] = normal.x; out[12 + 3 * f + 1] = normal.y; out[12 + 3 * f + 2] = normal.z;
		signs |= (__float_as_uint(dotV4) >> 31) << f;
	}
	out[24] = __uint_as_float(signs);
	for (int k = 25; k < 32; ++k) out[k] = 0.f;
	float4* dst = reinterpret_cast<float4*>(planes) + 8 * (size_t)i;
	for (int q = 0; q < 8; ++q) dst[q] = make_float4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
}

// ---- fine look-up table under the reference's 128^3 cell -> tet LUT (DeviceEdit::fine_*; round 6) -------------------------------------------
// The render path's scan of a LUT cell tests the cell's tets one after the other until one contains the sample: on the bench's cage 3.9 candidates per sample and --
// a wave is as slow as its slowest lane -- 7 dependent round trips per round that scans.  The fine table cuts every LUT cell of the mesh's window into S^3 fine
// cells (S = 4, 2 or 1 by the window's size) and gives each fine cell its LUT cell's list FILTERED IN ORDER: a tet stays unless one of its four face tests fails for EVERY
// position of the fine cell.  The test is made on the plane records -- the very floats point_in_tet_rec compares -- in double precision, with the rounding error of the
// float evaluation (four roundings per term: <= 4.0001 * 2^-24 * sum |n_i| |p_i - v_i|) doubled as slack, and the fine cell's box widened by what the index arithmetic can
// round (1e-6 / cascade scale): a tet that the float predicate accepts for some position of the fine cell is never dropped, so the first containing tet of the
// fine list is the first containing tet of the LUT cell's list -- the reference's result, bit for bit (tests/test_gpu_parity.py::test_map_rays_bit_exact and the 1080p frames).
// window: per cascade the box of LUT cells with a non-empty list (from the LUT itself: whatever built it), the number of such cells and the longest list
// win[c * 8 + 0..2] min x, y, z; [3..5] max; [6] non-empty cells; [7] longest list
__global__ __launch_bounds__(256) void fine_window_kernel(const uint32_t* __restrict__ offsets, int32_t* __restrict__ win) {
	// (a block covers 4096 consecutive cells of ONE cascade; its threads reduce in LDS and the block issues at most 8 device atomics: thousands of threads hitting
	// the same eight words took 0.28 ms of a cage move)
	__shared__ int32_t red[8];
	if (threadIdx.x < 8) red[threadIdx.x] = threadIdx.x < 3 ? 0x7fffffff : (threadIdx.x < 6 ? -1 : 0);
	__syncthreads();
	const uint32_t first = (blockIdx.x * blockDim.x + threadIdx.x) * 16u; // 16 cells in Morton order: a 4 x 2 x 2 block
	uint32_t o[17];
	#pragma unroll
	for (int q = 0; q < 4; ++q) { const uint4 v = reinterpret_cast<const uint4*>(offsets + first)[q]; o[4 * q] = v.x; o[4 * q + 1] = v.y; o[4 * q + 2] = v.z; o[4 * q + 3] = v.w; }
	o[16] = offsets[first + 16];
	if (o[0] != o[16]) {
		int32_t lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1}, cells = 0, longest = 0;
		#pragma unroll
		for (uint32_t k = 0; k < 16; ++k) {
			if (o[k + 1] != o[k]) {
				const uint32_t m = (first + k) % kGridVol;
				const int32_t c[3] = {(int32_t)morton3D_invert(m), (int32_t)morton3D_invert(m >> 1), (int32_t)morton3D_invert(m >> 2)};
				for (int a = 0; a < 3; ++a) { lo[a] = min(lo[a], c[a]); hi[a] = max(hi[a], c[a]); }
				++cells;
				longest = max(longest, (int32_t)(o[k + 1] - o[k]));
			}
		}
		for (int a = 0; a < 3; ++a) { atomicMin(&red[a], lo[a]); atomicMax(&red[3 + a], hi[a]); }
		atomicAdd(&red[6], cells);
		atomicMax(&red[7], longest);
	}
	__syncthreads();
	if (threadIdx.x < 8 && red[6] != 0) {
		int32_t* dst = win + (blockIdx.x * blockDim.x * 16u / kGridVol) * 8 + threadIdx.x;
		if (threadIdx.x < 3) atomicMin(dst, red[threadIdx.x]);
		else if (threadIdx.x < 6 || threadIdx.x == 7) atomicMax(dst, red[threadIdx.x]);
		else atomicAdd(dst, red[6]);
	}
}
// can the tet of plane record `r` contain (by the float predicate of point_in_tet_rec) a position of the box centre c, half extents h?  Conservative: false only when
// some face test fails everywhere.  dot(n, p - v) over the box is dot(n, c - v) +- sum |n_i| h_i; the predicate's own float evaluation is off by at most
// 4.0001 * 2^-24 * sum |n_i| |p_i - v_i| <= that bound with |c_i - v_i| + h_i, this function's float arithmetic by as much again: the slack is 20 * 2^-24 of it.
// (First version: double precision, min / max per axis -- 160 us per launch, the vector unit's double rate; this form is a third of the operations at four times the rate.)
// (r: wave-uniform -- the record arrives through scalar loads)
__device__ __forceinline__ bool tet_may_contain_box(const float* __restrict__ r, const float c[3], const float h[3]) {
	const uint32_t signs = __float_as_uint(r[24]);
	bool may = true;
	#pragma unroll
	for (int f = 0; f < 4; ++f) {
		float dc = 0.f, rad = 0.f, err = 0.f;
		#pragma unroll
		for (int i = 0; i < 3; ++i) {
			const float n = r[12 + 3 * f + i], d = c[i] - r[3 * f + i];
			dc = fmaf(n, d, dc);
			rad = fmaf(fabsf(n), h[i], rad);
			err = fmaf(fabsf(n), fabsf(d) + h[i], err);
		}
		const float slack = fmaf(err, 20.0f / 16777216.0f, 1e-30f);
		// the test wants the sign bit set: impossible when dot > 0 everywhere (dc - rad > slack); ... clear: impossible when dot < 0 everywhere
		may = may && (((signs >> f) & 1u) ? !(dc - rad > slack) : !(dc + rad < -slack));
	}
	return may;
}
// One WAVE per LUT cell of a cascade's window, one LANE per fine cell of it (4 x 4 x 4 = 64 with shift 2): every lane walks the LUT cell's list -- ids fetched 64 at a
// time, the plane record of a candidate is wave-uniform -- and keeps what its fine cell's box cannot exclude.  FILL == false: counts[fine cell] = survivors.
// FILL == true: writes them, in list order, at fine_off[fine cell].  (Thread-per-fine-cell, the first version, walked the coarse cascades' lists of thousands of
// tets serially with a dependent load per step: 20 ms per cage move at 48 k tets.)
template <bool FILL>
__global__ __launch_bounds__(256) void fine_lists_kernel(const DeviceEdit e, uint32_t* __restrict__ counts, const uint32_t* __restrict__ fine_off,
                                                         uint32_t* __restrict__ fine_idx) {
	uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); // one launch for every cascade: the few long lists of the coarse ones run beside the many short ones
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t level = kCascades;
	#pragma unroll
	for (uint32_t c = 0; c < kCascades; ++c) {
		const int32_t* f = e.fine_win[c];
		const uint32_t n_par = (f[4] == 0 || (uint32_t)f[7] > 2u) ? 0u : ((uint32_t)f[4] >> f[7]) * ((uint32_t)f[5] >> f[7]) * ((uint32_t)f[6] >> f[7]);
		if (level == kCascades) { if (w < n_par) level = c; else w -= n_par; }
	}
	if (level == kCascades) return;
	const uint32_t S = (uint32_t)e.fine_win[level][7], sub = 1u << S; // fine cells per LUT cell and axis
	const uint32_t ex = (uint32_t)e.fine_win[level][4], ey = (uint32_t)e.fine_win[level][5];
	const uint32_t px = ex >> S, py = ey >> S; // the window in LUT cells
	const uint32_t cx = w % px, cy = (w / px) % py, cz = w / (px * py);
	const uint32_t lx = ((uint32_t)e.fine_win[level][0] >> S) + cx, ly = ((uint32_t)e.fine_win[level][1] >> S) + cy, lz = ((uint32_t)e.fine_win[level][2] >> S) + cz;
	const uint32_t parent = level * kGridVol + morton3D(lx, ly, lz);
	const uint32_t j0 = e.lut_off[parent], j1 = e.lut_off[parent + 1];
	const bool owner = lane < sub * sub * sub;
	const uint32_t sx = lane & (sub - 1u), sy = (lane >> S) & (sub - 1u), sz = lane >> (2u * S);
	const uint32_t rx = (cx << S) + sx, ry = (cy << S) + sy, rz = (cz << S) + sz; // fine cell inside the window
	const uint32_t cell = (uint32_t)e.fine_win[level][3] + (rz * ey + ry) * ex + rx;
	if (j0 == j1) { if (!FILL && owner) counts[cell] = 0u; return; }
	// the positions u whose fine coordinate is f: q = ((u - 0.5) * 2^-level + 0.5), floor(q * res) = f  =>  u in 0.5 + ((f .. f + 1) / res - 0.5) * 2^level, widened by
	// what the three float operations can round; the outermost fine cells take everything beyond (the index clamps)
	const uint32_t res = kGrid << S;
	const double scale = (double)(1u << level), margin = 2e-6 * scale; // (1e-6: the index arithmetic; the rest: c and h below are rounded to float)
	const uint32_t f[3] = {(uint32_t)e.fine_win[level][0] + rx, (uint32_t)e.fine_win[level][1] + ry, (uint32_t)e.fine_win[level][2] + rz};
	float c[3], h[3];
	bool border = false; // an outermost fine cell of the grid takes every position beyond it: no box, nothing is filtered
	#pragma unroll
	for (int i = 0; i < 3; ++i) {
		const double lo = 0.5 + ((double)f[i] / (double)res - 0.5) * scale - margin, hi = 0.5 + ((double)(f[i] + 1) / (double)res - 0.5) * scale + margin;
		c[i] = (float)(0.5 * (lo + hi));
		h[i] = (float)(0.5 * (hi - lo));
		border = border || f[i] == 0 || f[i] >= res - 1;
	}
	uint32_t n = 0, wpos = (FILL && owner) ? fine_off[cell] : 0u;
	for (uint32_t b = j0; b < j1; b += 64u) {
		const uint32_t mine = b + lane < j1 ? e.lut_idx[b + lane] : 0u; // 64 candidates, one per lane
		const uint32_t nb = min(64u, j1 - b);
		// (the record of candidate k + 1 is requested before candidate k is tested: a list is a chain of dependent loads otherwise, 1.5 us per tet)
		float4 cur[7], nxt[7];
		{
			const float4* q = reinterpret_cast<const float4*>(e.planes) + 8 * (size_t)(uint32_t)__builtin_amdgcn_readlane((int)mine, 0);
			#pragma unroll
			for (int i = 0; i < 7; ++i) cur[i] = q[i];
		}
		for (uint32_t k = 0; k < nb; ++k) {
			const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)k); // wave-uniform
			const uint32_t t_next = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)min(k + 1u, nb - 1u));
			{
				const float4* q = reinterpret_cast<const float4*>(e.planes) + 8 * (size_t)t_next;
				#pragma unroll
				for (int i = 0; i < 7; ++i) nxt[i] = q[i];
			}
			const float* r = reinterpret_cast<const float*>(cur);
			if ((border || tet_may_contain_box(r, c, h)) && owner) {
				if (FILL) fine_idx[wpos++] = t;
				++n;
			}
			#pragma unroll
			for (int i = 0; i < 7; ++i) cur[i] = nxt[i];
		}
	}
	if (!FILL && owner) counts[cell] = n;
}

// ---- launchers ----------------------------------------------------------------------------------------------------------------
int launch_mvc_apply(uint32_t n_points, uint32_t n_cv, const float* d_weights, const float* d_cage, float* d_points, void* stream) {
	hipLaunchKernelGGL(mvc_apply_kernel, dim3((n_points + 127) / 128), dim3(128), 0, (hipStream_t)stream, n_points, n_cv, d_weights, d_cage, d_points);
	NRS_CAGE_CHECK("mvc_apply_kernel launch");
	return NRS_OK;
}
int launch_poisson_interpolate(uint32_t n_points, uint32_t n_cv, const float* d_gamma, const float* d_per_cage, float* d_shs, float* d_out_density, float* d_res_density, void* stream) {
	hipLaunchKernelGGL(poisson_interpolate_kernel, dim3((n_points + 63) / 64), dim3(64), 0, (hipStream_t)stream, n_points, n_cv, d_gamma, d_per_cage, d_shs, d_out_density, d_res_density);
	NRS_CAGE_CHECK("poisson_interpolate_kernel launch");
	return NRS_OK;
}
int launch_bbox(uint32_t n, const float* d_verts, float* d_out6, void* stream) {
	hipLaunchKernelGGL(bbox_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, n, d_verts, d_out6);
	NRS_CAGE_CHECK("bbox_kernel launch");
	return NRS_OK;
}
// counts must be all zero on entry (it is again on exit of launch_lut_fill).  Writes offsets[kCells + 1] and *d_total.
int launch_lut_count_scan(uint32_t n_tets, const float* d_verts, const uint32_t* d_tets, uint32_t* d_counts, uint32_t* d_tile_sums,
                          uint32_t* d_offsets, uint32_t* d_total, unsigned long long* d_hit_masks, float cells0, void* stream) {
	hipStream_t s = (hipStream_t)stream;
	launch_tet_mark<false>(n_tets, cells0, d_verts, d_tets, d_counts, nullptr, nullptr, d_hit_masks, s);
	hipLaunchKernelGGL(scan_tile_sum_kernel, dim3(kScanTiles), dim3(256), 0, s, d_counts, d_tile_sums);
	hipLaunchKernelGGL(scan_tile_prefix_kernel, dim3(1), dim3(1024), 0, s, d_tile_sums, d_offsets + kCells, d_total, kScanTiles);
	hipLaunchKernelGGL(scan_write_kernel, dim3(kScanTiles), dim3(256), 0, s, d_counts, d_tile_sums, d_offsets);
	NRS_CAGE_CHECK("tet LUT count/scan launch");
	return NRS_OK;
}
// d_scratch_u32[0] = max tets per cell (out), [1] / [2] = counters of the two sort worklists; all zeroed here.  d_bitfield may be NULL.
// d_big_cells: worklist of n_work >= (entries / kSmallList + 1) cells (long lists from its front, middle ones from its back).
int launch_lut_fill(uint32_t n_tets, const float* d_verts, const uint32_t* d_tets, uint32_t* d_counts, const uint32_t* d_offsets, uint32_t* d_idx,
                    uint8_t* d_bitfield, uint32_t* d_scratch_u32, uint32_t* d_big_cells, uint32_t n_work, unsigned long long* d_hit_masks, float cells0, void* stream) {
	hipStream_t s = (hipStream_t)stream;
	if (hipMemsetAsync(d_scratch_u32, 0, 12, s) != hipSuccess) { snprintf(g_cage_err, sizeof(g_cage_err), "tet LUT fill: memset failed"); return NRS_ERR_HIP; }
	launch_tet_mark<true>(n_tets, cells0, d_verts, d_tets, d_counts, d_offsets, d_idx, d_hit_masks, s);
	hipLaunchKernelGGL(lut_finish_kernel, dim3(kCells / 256), dim3(256), 0, s, d_offsets, d_idx, d_bitfield, d_scratch_u32, d_big_cells,
	                   d_scratch_u32 + 1, n_work);
	hipLaunchKernelGGL(lut_sort_mid_kernel, dim3(512), dim3(1024), 0, s, d_offsets, d_idx, d_big_cells, d_scratch_u32 + 2, n_work);
	hipLaunchKernelGGL(lut_sort_big_kernel, dim3(512), dim3(1024), 0, s, d_offsets, d_idx, d_big_cells, d_scratch_u32 + 1, n_tets);
	NRS_CAGE_CHECK("tet LUT fill launch");
	return NRS_OK;
}
uint32_t lut_big_list_capacity(size_t idx_capacity) { return (uint32_t)(idx_capacity / kSmallList + 1); }
int launch_tet_planes(uint32_t n_tets, const float* d_verts, const uint32_t* d_tets, float* d_planes, void* stream) {
	hipLaunchKernelGGL(tet_planes_kernel, dim3((n_tets + 127) / 128), dim3(128), 0, (hipStream_t)stream, n_tets, d_verts, d_tets, d_planes);
	NRS_CAGE_CHECK("tet_planes_kernel launch");
	return NRS_OK;
}
int launch_local_rotations(uint32_t n_tets, const float* d_verts, const float* d_orig, const uint32_t* d_tets, float* d_out, void* stream) {
	hipLaunchKernelGGL(local_rotations_kernel, dim3((n_tets + 63) / 64), dim3(64), 0, (hipStream_t)stream, n_tets, d_verts, d_orig, d_tets, d_out);
	NRS_CAGE_CHECK("local_rotations_kernel launch");
	return NRS_OK;
}

int launch_fine_window(const uint32_t* d_lut_off, int32_t* d_window_out, void* stream) {
	hipStream_t s = (hipStream_t)stream;
	static int32_t init[kCascades * 8];
	for (uint32_t c = 0; c < kCascades; ++c) { int32_t* w = init + 8 * c; w[0] = w[1] = w[2] = 0x7fffffff; w[3] = w[4] = w[5] = -1; w[6] = w[7] = 0; }
	if (hipMemcpyAsync(d_window_out, init, sizeof(init), hipMemcpyHostToDevice, s) != hipSuccess) { snprintf(g_cage_err, sizeof(g_cage_err), "fine look-up table: window init failed"); return NRS_ERR_HIP; }
	hipLaunchKernelGGL(fine_window_kernel, dim3(kCells / 16 / 256), dim3(256), 0, s, d_lut_off, d_window_out);
	NRS_CAGE_CHECK("fine_window_kernel launch");
	return NRS_OK;
}
template <bool FILL>
static void launch_fine_lists(const DeviceEdit& de, uint32_t* d_counts, const uint32_t* d_fine_off, uint32_t* d_fine_idx, hipStream_t s) {
	uint32_t n_parents = 0;
	for (uint32_t c = 0; c < kCascades; ++c) {
		const int32_t* f = de.fine_win[c];
		if (f[4] == 0 || (uint32_t)f[7] > 2u) continue;
		n_parents += ((uint32_t)f[4] >> f[7]) * ((uint32_t)f[5] >> f[7]) * ((uint32_t)f[6] >> f[7]);
	}
	if (n_parents) hipLaunchKernelGGL(fine_lists_kernel<FILL>, dim3((n_parents + 3) / 4), dim3(256), 0, s, de, d_counts, d_fine_off, d_fine_idx);
}
int launch_fine_count_scan(const DeviceEdit& de, uint32_t n_fine_cells, uint32_t* d_counts, uint32_t* d_tile_sums, uint32_t* d_fine_off, uint32_t* d_total, void* stream) {
	hipStream_t s = (hipStream_t)stream;
	const uint32_t n_tiles = (n_fine_cells + kScanTile - 1) / kScanTile, n_padded = n_tiles * kScanTile;
	if (n_tiles == 0 || n_tiles > kMaxScanTiles) { snprintf(g_cage_err, sizeof(g_cage_err), "fine look-up table: %u fine cells", n_fine_cells); return NRS_ERR_INVALID_ARG; }
	if (n_padded != n_fine_cells && hipMemsetAsync(d_counts + n_fine_cells, 0, (size_t)(n_padded - n_fine_cells) * 4, s) != hipSuccess) { snprintf(g_cage_err, sizeof(g_cage_err), "fine look-up table: memset failed"); return NRS_ERR_HIP; }
	launch_fine_lists<false>(de, d_counts, nullptr, nullptr, s);
	hipLaunchKernelGGL(scan_tile_sum_kernel, dim3(n_tiles), dim3(256), 0, s, d_counts, d_tile_sums);
	hipLaunchKernelGGL(scan_tile_prefix_kernel, dim3(1), dim3(1024), 0, s, d_tile_sums, d_fine_off + n_padded, d_total, n_tiles);
	hipLaunchKernelGGL(scan_write_kernel, dim3(n_tiles), dim3(256), 0, s, d_counts, d_tile_sums, d_fine_off);
	NRS_CAGE_CHECK("fine look-up table count/scan launch");
	return NRS_OK;
}
int launch_fine_fill(const DeviceEdit& de, uint32_t n_fine_cells, const uint32_t* d_fine_off, uint32_t* d_fine_idx, void* stream) {
	(void)n_fine_cells;
	launch_fine_lists<true>(de, nullptr, d_fine_off, d_fine_idx, (hipStream_t)stream);
	NRS_CAGE_CHECK("fine_lists_kernel launch");
	return NRS_OK;
}

} // namespace nrs
