// nrs_mlp.cuh -- hash-grid gather + fused MLPs on MFMA for one wavefront (gfx950).
//
// Replaces tiny-cuda-nn's kernel_grid + 2x kernel_mlp_fused + SH encoding + extract_density with one
// on-chip pipeline.  A wave owns 64 samples, processed as two 32-sample MFMA column blocks:
//
//   block b holds the samples of lanes 32b..32b+31.  Every lane gathers the 16 hash-grid levels of its OWN sample, two levels per
//   iteration (level parameters wave-uniform, in scalar registers), and writes them into the per-wave LDS slab in the order the
//   MFMA B operands want: lane l = (j = l & 31, g = l >> 5) supplies, for sample j of each block, the levels 2*it + g (it = 0..7)
//   and the SH coefficients 8g..8g+7.
//
// Every layer is computed transposed, H^T[unit][sample] = W[unit][k] * X^T[k][sample], with
// v_mfma_f32_32x32x16_f16: A = a 32x16 weight tile (LDS, pre-arranged on the host), B = 16 x 32 samples.  The D tile of
// one layer (lane = sample column, 16 rows per lane) is, after ReLU + fp16 rounding, directly the B operand of the next
// layer: the k index of an MFMA is free as long as A and B agree, so the host arranges the weight tiles in the order the
// D registers come out (make_weight_fragments in nrs_api.cpp).  No cross-lane shuffles between layers, no global
// intermediates.  The gathered features pass through a 4 KiB per-wave LDS slab only because the level loop is kept
// rolled (small code, few live registers, which buys occupancy for the gather latency).
//
// Numerics (stated; parity at the tcnn boundary is unpinned): grid entries fp16, trilinear sum in fp32 via
// fmaf in corner order 0..7, rounded to fp16; MLP products fp16 x fp16 accumulated in fp32 by the MFMA, ReLU, rounded to
// fp16 between layers; outputs rounded to fp16.
#pragma once
#include <hip/hip_runtime.h>
#include "nrs_device.cuh"

// L2 time-multiplexing of the four finest hashed levels (encode_to_lds, GATE): a phase is 2^kGateShift ticks of the 100 MHz wall clock (20.5 us); a wave waits at most
// kGateWaitCap eighths of a phase for the right one (profiles/r06_garden.md: 10 / 20 / 41 us phases, wait caps 2 / 4 / 8 eighths measured)
#ifndef NRS_GATE_SHIFT
#define NRS_GATE_SHIFT 11
#endif
#ifndef NRS_GATE_WAIT_CAP
#define NRS_GATE_WAIT_CAP 2
#endif
constexpr uint32_t kGateShift = NRS_GATE_SHIFT, kGateWaitCap = NRS_GATE_WAIT_CAP;
#ifndef NRS_GATE_MAX_PHASES
#define NRS_GATE_MAX_PHASES 4
#endif
#ifndef NRS_REFRESH_GATE_PHASES
#define NRS_REFRESH_GATE_PHASES 6 // (grid_refresh_kernel: without cell records five / six level pairs are hashed; 5.68 -> 5.04 ms at aabb 16, profiles/r06/ab_refresh_gate_phases.txt)
#endif
constexpr uint32_t kGateMaxPhases = NRS_GATE_MAX_PHASES; // 2: levels 12..15 only; 3 / 4: the pairs below too when they are hashed without records.  Measured (profiles/r06/ab_gate_phases_*.txt):
// the records budget's knee (levels 10..15 hashed) 4.59 -> 4.96 Gsamples/s with three phases, no sparse records (8..15 hashed) 4.48 -> 4.60 with four, 64 GiB (two phases) unchanged

namespace nrs {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// fragment indices inside the LDS weight image
#define NRS_FRAG_D1(mb, ks) ((mb) * 2 + (ks))
#define NRS_FRAG_D2(ks) (4 + (ks))
#define NRS_FRAG_R1(mb, ks) (8 + (mb) * 2 + (ks))
#define NRS_FRAG_R2(mb, ks) (12 + (mb) * 4 + (ks))
#define NRS_FRAG_R3(ks) (20 + (ks))
#define NRS_FRAG_R2B(mb, ks) (30 + (mb) * 4 + (ks)) // (HBM only, like NRS_FRAG_BWD: the third rgb hidden layer of base_3layer.json)

enum { KIND_DENSE = 0, KIND_HASHED = 1, KIND_MIXED = 2, KIND_RECORD = 3, KIND_SPARSE = 4, KIND_SKIP = 5 };

// Per-block model state in LDS: weight fragments, kind of each level pair (the level table itself is read with scalar loads).
struct ModelLds {
	half8 w[kNumFrags * 64];
	uint32_t kinds[8];        // kind of the level pair (2 it, 2 it + 1): both records / both hashed / both dense, else KIND_MIXED
	uint32_t kinds_native[8]; // the kinds without the cell records (samples outside [0,1]^3 take these)
	uint32_t one_line;        // NRS_DEBUG & 1
};
// Per-wave feature slab: feat[it][sel][lane], sel 0 = the lane's own sample, 1 = its partner's (lane ^ 32) sample.
struct FeatLds { uint32_t feat[8][2][64]; };

__device__ __forceinline__ void stage_model_to_lds(const DeviceModel& m, ModelLds& s, uint32_t dbg = 0) {
	const uint4* src = reinterpret_cast<const uint4*>(m.wfrag);
	uint4* dst = reinterpret_cast<uint4*>(s.w);
	for (uint32_t i = threadIdx.x; i < kWfragBytes / 16; i += blockDim.x) dst[i] = src[i];
	if (threadIdx.x < 8) {
		const uint32_t h0 = (dbg & 1u) ? 1u : m.levels[2 * threadIdx.x].hashed, h1 = (dbg & 1u) ? 1u : m.levels[2 * threadIdx.x + 1].hashed;
		const uint32_t c0 = (dbg & 1u) ? 0u : m.levels[2 * threadIdx.x].cached, c1 = (dbg & 1u) ? 0u : m.levels[2 * threadIdx.x + 1].cached;
		const uint32_t native = (h0 && h1) ? KIND_HASHED : ((!h0 && !h1) ? KIND_DENSE : KIND_MIXED);
		s.kinds[threadIdx.x] = (c0 == 1u && c1 == 1u) ? KIND_RECORD : ((c0 == 2u && c1 == 2u) ? KIND_SPARSE : ((c0 == 1u || c1 == 1u) ? KIND_MIXED : native));
		s.kinds_native[threadIdx.x] = native;
		// measurement (NRS_SKIP_PAIRS, dbg bits 8..15): level pairs that are not gathered at all (their features read 0) -- the L2 misses of a frame
		// attributed level pair by level pair (profiles/r06_garden_levels.md).  Pictures are wrong; a production process cannot set it (dev_knob).
		if ((dbg >> 8) & (1u << threadIdx.x)) s.kinds[threadIdx.x] = s.kinds_native[threadIdx.x] = KIND_SKIP;
	}
	if (threadIdx.x == 0) s.one_line = dbg & 1u;
	__syncthreads();
}

// The table is read through a buffer descriptor: one 32-bit offset per gather instead of 64-bit pointer arithmetic.
struct GridView { __amdgpu_buffer_rsrc_t rsrc; const uint4* records; const uint4* records2; const uint32_t* bricks; };
__device__ __forceinline__ GridView make_grid_view(const uint32_t* grid, uint32_t n_entries, const uint4* records = nullptr, const uint4* records2 = nullptr,
                                                   const uint32_t* bricks = nullptr) {
	GridView v;
	v.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)grid, 0, (int)(n_entries * 4u), 0x00020000);
	v.records = records;
	v.records2 = records2;
	v.bricks = bricks;
	return v;
}
__device__ __forceinline__ GridView make_grid_view(const DeviceModel& m) {
	return make_grid_view(m.grid, m.levels[kLevels - 1].offset + m.levels[kLevels - 1].count, (const uint4*)m.records, (const uint4*)m.records2, m.bricks);
}
__device__ __forceinline__ uint32_t grid_load(const GridView& v, uint32_t entry) {
	return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(v.rsrc, (int)(entry * 4u), 0, 0);
}
__device__ __forceinline__ uint32_t grid_load_bytes(const GridView& v, uint32_t byte_offset) {
	return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(v.rsrc, (int)byte_offset, 0, 0);
}
// acc + w * half(lo/hi 16 bits of a table entry): v_fma_mix_f32 converts the fp16 operand on the fly (exactly) and fuses the
// multiply-add with one rounding, i.e. fmaf(w, (float)h, acc) without the separate v_cvt_f32_f16.
__device__ __forceinline__ float fma_mix_lo(float w, uint32_t entry, float acc) {
	float r;
	asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(entry), "v"(acc));
	return r;
}
__device__ __forceinline__ float fma_mix_hi(float w, uint32_t entry, float acc) {
	float r;
	asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(entry), "v"(acc));
	return r;
}

// Exact tcnn index for any position (also far outside [0,1]^3): the rarely taken out-of-line path.
template <bool NETACC = false>
__device__ __forceinline__ void level_eval_slow(const GridView gv, const LevelParams lp, uint32_t gx, uint32_t gy, uint32_t gz,
                                                          float wx, float wy, float wz, float* out0, float* out1) {
	float acc0 = 0.f, acc1 = 0.f;
	_Float16 h0 = (_Float16)0.f, h1 = (_Float16)0.f;
	#pragma unroll 1
	for (int c = 0; c < 8; ++c) {
		const uint32_t cx = gx + (c & 1), cy = gy + ((c >> 1) & 1), cz = gz + ((c >> 2) & 1);
		uint32_t index = lp.hashed ? ((cx * 1u) ^ (cy * 2654435761u) ^ (cz * 805459861u)) : (cx + cy * lp.resolution + cz * lp.res2);
		index %= lp.count;
		float weight = 1.0f;
		weight *= (c & 1) ? wx : 1.0f - wx;
		weight *= (c & 2) ? wy : 1.0f - wy;
		weight *= (c & 4) ? wz : 1.0f - wz;
		const half2v hv = __builtin_bit_cast(half2v, grid_load(gv, lp.offset + index));
		if (NETACC) {
			h0 = h0 + (_Float16)(weight * (float)hv[0]);
			h1 = h1 + (_Float16)(weight * (float)hv[1]);
		} else {
			acc0 = fmaf(weight, (float)hv[0], acc0);
			acc1 = fmaf(weight, (float)hv[1], acc1);
		}
	}
	*out0 = NETACC ? (float)h0 : acc0;
	*out1 = NETACC ? (float)h1 : acc1;
}

// Index arithmetic on the full-rate 24-bit multiplier.  v_mul_lo_u32 is a quarter-rate instruction (16 cycles per wave64 against 4), and the gather
// issues four of them per level pair.  v_mul_u32_u24 returns the low 32 bits of the product of the operands' low 24 bits -- the same number whenever
// both operands are below 2^24 (grid coordinates, resolutions, and resolution^2 of every level that can have dense storage or records: res < 4096),
// and the same LOW 24 BITS for any operands, which is all a hashed index keeps (mask = 2^log2_T - 1, log2_T <= 24).
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 grid_load2_bytes(const GridView& v, uint32_t byte_offset) {
	return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(v.rsrc, (int)byte_offset, 0, 0));
}

// Grid cell + interpolation weights of one sample at one level.
struct CellCoords { uint32_t gx, gy, gz; float wx, wy, wz; };
__device__ __forceinline__ CellCoords cell_coords(const LevelParams& lp, f3 pos) {
	CellCoords c;
	const float px = fmaf(lp.scale, pos.x, 0.5f), py = fmaf(lp.scale, pos.y, 0.5f), pz = fmaf(lp.scale, pos.z, 0.5f);
	const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
	c.gx = (uint32_t)(int)fx; c.gy = (uint32_t)(int)fy; c.gz = (uint32_t)(int)fz;
	c.wx = px - fx; c.wy = py - fy; c.wz = pz - fz;
	return c;
}
// The same for a position inside [0,1]^3 (every caller of the record paths: a wave with a sample outside the cube gathers the native way): there
// p = scale * x + 0.5 >= 0.5, so the cell is the truncation of p (v_cvt_u32_f32: no floor, no second conversion) and the weight its fractional part
// (v_fract_f32 = p - floor(p), exact for p >= 0): the same cell and the same weight bits in 9 instructions per level instead of 12.
__device__ __forceinline__ CellCoords cell_coords_incube(const LevelParams& lp, f3 pos) {
	CellCoords c;
	typedef float f2p __attribute__((ext_vector_type(2))); // scale * (x, y) + 0.5 as one v_pk_fma_f32 (two IEEE fmas: same bits), z alone
	const f2p sc = {lp.scale, lp.scale}, xy = {pos.x, pos.y}, hf = {0.5f, 0.5f};
	const f2p pxy = __builtin_elementwise_fma(sc, xy, hf);
	const float px = pxy.x, py = pxy.y, pz = fmaf(lp.scale, pos.z, 0.5f);
	c.gx = (uint32_t)px; c.gy = (uint32_t)py; c.gz = (uint32_t)pz;
	c.wx = __builtin_amdgcn_fractf(px); c.wy = __builtin_amdgcn_fractf(py); c.wz = __builtin_amdgcn_fractf(pz);
	return c;
}
typedef float f2 __attribute__((ext_vector_type(2)));
// dense fast path precondition: no index of the cell reaches `count`, so no wrap and x-neighbours are adjacent entries
__device__ __forceinline__ bool dense_needs_slow(const LevelParams& lp, const CellCoords& c) {
	return c.gx >= lp.resolution || c.gy >= lp.resolution || c.gz >= lp.resolution ||
	       c.gx + mul24(c.gy, lp.resolution) + mul24(c.gz, lp.res2) + 1u + lp.resolution + lp.res2 >= lp.count; // (the products are only used when the three tests before passed)
}
// Issue the gathers of one sample at one level: v[2q + bx] = entry of corner (bx, q&1, q>>1).
// Dense levels fetch the two x-neighbours with ONE 8-byte load (entry(x+1) = entry(x) + 1; MUBUF needs dword alignment only).
template <bool HASHED>
__device__ __forceinline__ void issue_gathers(const GridView& gv, const LevelParams& lp, const CellCoords& c, uint32_t v[8]) {
	const uint32_t off4 = lp.offset * 4u;
	if (HASHED) {
		// (only the bits under lp.mask < 2^24 survive: the low 24 bits of the primes on the 24-bit multiplier give the same index)
		// The BYTE offset of an entry directly: 4 e = (4 x ^ 4 y P1 ^ 4 z P2) & 4 mask (shifts commute with xor and and), with 4 P < 2^24 on the 24-bit
		// multiplier (the low 24 bits of the primes, as above: 4 mask < 2^26 keeps bits 2..25 of the products, i.e. bits 0..23 of y P), and the level's
		// first entry in the instruction's scalar offset -- no shift and no add per corner.
		const uint32_t hx0 = c.gx << 2, hx1 = hx0 + 4u, hy0 = mul24(c.gy, (2654435761u & 0xffffffu) * 4u), hy1 = hy0 + (2654435761u & 0xffffffu) * 4u,
		               hz0 = mul24(c.gz, (805459861u & 0xffffffu) * 4u), hz1 = hz0 + (805459861u & 0xffffffu) * 4u;
		const uint32_t mask4 = lp.mask << 2;
		#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const uint32_t e4 = (((k & 1) ? hx1 : hx0) ^ ((k & 2) ? hy1 : hy0) ^ ((k & 4) ? hz1 : hz0)) & mask4;
			v[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(gv.rsrc, (int)e4, (int)off4, 0);
		}
	} else {
		const uint32_t base = c.gx + mul24(c.gy, lp.resolution) + mul24(c.gz, lp.res2); // dense level: res^3 <= 2^24, coordinates checked by dense_needs_slow
		#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const uint32_t i = base + ((q & 1) ? lp.resolution : 0u) + ((q & 2) ? lp.res2 : 0u);
			const u32x2 pr = grid_load2_bytes(gv, (i << 2) + off4);
			v[2 * q] = pr[0];
			v[2 * q + 1] = pr[1];
		}
	}
}
// Cached level: the 8 corner entries of the sample's cell sit in one 32-byte record (written by cell_records_kernel with
// the level's own index function, so the values are the ones the hashed / dense gather would fetch): two 16-byte loads
// and one address instead of eight gathers and eight hashes, and one cache line instead of four.
// Every cell a position in [0,1]^3 falls into has a record: floor(scale * p + 0.5) <= ceil(scale) = resolution - 1.
__device__ __forceinline__ bool outside_unit_cube(f3 p) {
	return !(p.x >= 0.f && p.x <= 1.f && p.y >= 0.f && p.y <= 1.f && p.z >= 0.f && p.z <= 1.f);
}
__device__ __forceinline__ void issue_record_loads(const GridView& gv, const LevelParams& lp, const CellCoords& c, uint32_t v[8]) {
	const uint32_t rec = lp.rec_first + c.gx + mul24(c.gy, lp.rec_res) + mul24(c.gz, lp.rec_res2); // rec_res < 4096 (plan_cell_cache), product < 2^32 records
	const uint4* p = gv.records + 2 * (size_t)rec;
	const uint4 lo = p[0], hi = p[1];
	v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
	v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
}
// Trilinear interpolation in the oracle's order (corner 0..7, x fastest; weight = (wx' * wy') * wz') -> packed fp16 pair.
// NETACC (nrs_grid_acc NETWORK): tiny-cuda-nn's kernel_grid as recalled -- every corner's fp32 product is rounded to fp16 and added in fp16.
template <bool NETACC = false>
__device__ __forceinline__ uint32_t interpolate(const CellCoords& c, const uint32_t v[8]) {
	const float ux = 1.0f - c.wx, uy = 1.0f - c.wy, uz = 1.0f - c.wz;
	const float wxy[4] = {ux * uy, c.wx * uy, ux * c.wy, c.wx * c.wy};
	if (NETACC) {
		half2v r = {(_Float16)0.f, (_Float16)0.f};
		#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const float weight = wxy[k & 3] * ((k & 4) ? c.wz : uz);
			// the two fp32 products straight from the packed fp16 entry (v_fma_mix_f32 with a zero addend: the product's own rounding), one v_cvt_pk_f16_f32,
			// one v_pk_add_f16: 4 issue slots per corner.  (v_fma_mixlo/hi_f16 -- 3 slots -- is neither bit-identical to the two-step rounding nor faster:
			// profiles/r04/ab_numerics_variants.txt.)
			const half2v pr = {(_Float16)fma_mix_lo(weight, v[k], 0.f), (_Float16)fma_mix_hi(weight, v[k], 0.f)};
			r = r + pr;
		}
		return __builtin_bit_cast(uint32_t, r);
	}
	float acc0 = 0.f, acc1 = 0.f;
	{ // the same twelve products as packed fp32 multiplies (v_pk_mul_f32: two IEEE products per issue slot; same bits)
		typedef float f2v __attribute__((ext_vector_type(2)));
		const f2v xs = {ux, c.wx};
		const f2v p01 = xs * uy, p23 = xs * c.wy;
		const f2v w01 = p01 * uz, w23 = p23 * uz, w45 = p01 * c.wz, w67 = p23 * c.wz;
		const float wk[8] = {w01.x, w01.y, w23.x, w23.y, w45.x, w45.y, w67.x, w67.y};
		#pragma unroll
		for (int k = 0; k < 8; ++k) {
			acc0 = fma_mix_lo(wk[k], v[k], acc0);
			acc1 = fma_mix_hi(wk[k], v[k], acc1);
		}
	}
	half2v r;
	r[0] = (_Float16)acc0;
	r[1] = (_Float16)acc1;
	return __builtin_bit_cast(uint32_t, r);
}
template <bool NETACC = false>
__device__ __forceinline__ uint32_t level_eval_exact(const GridView& gv, const LevelParams& lp, const CellCoords& c) {
	float a0, a1;
	level_eval_slow<NETACC>(gv, lp, c.gx, c.gy, c.gz, c.wx, c.wy, c.wz, &a0, &a1);
	half2v r;
	r[0] = (_Float16)a0;
	r[1] = (_Float16)a1;
	return __builtin_bit_cast(uint32_t, r);
}

// Sparse level (nrs_model_set_sparse_cell_cache): the level's cells are grouped in 8 x 8 x 8 bricks; a table says which bricks carry
// records (slot + 1) and which do not (0).  One table load (neighbouring samples share its lines), then the record as in a dense level.
__device__ __forceinline__ uint32_t brick_entry(const GridView& gv, const LevelParams& lp, const CellCoords& c) {
	// bricks per side <= 4097 (res 32769 of an aabb-16 level 15): rec_res2 can exceed 2^24 by a hair -- wave-uniform choice of the multiplier
	const uint32_t zy = lp.rec_res2 < (1u << 24) ? mul24(c.gz >> 3, lp.rec_res2) + mul24(c.gy >> 3, lp.rec_res) : (c.gz >> 3) * lp.rec_res2 + (c.gy >> 3) * lp.rec_res;
	return gv.bricks[lp.tab_first + zy + (c.gx >> 3)];
}
__device__ __forceinline__ void issue_brick_record_loads(const GridView& gv, const LevelParams& lp, const CellCoords& c, uint32_t brick, uint32_t v[8]) {
	const uint32_t rec = lp.rec_first + (brick - 1u) * 512u + brick_slot(c.gx & 7u, c.gy & 7u, c.gz & 7u);
	const uint4* p = gv.records2 + 2 * (size_t)rec;
	// (`nt` on both halves keeps the record's line from displacing hot table lines -- tools/probe/l2_retention_probe: hot-table misses 30 % -> 10 % beside a cold stream --
	// and is worth +2.5 % on the garden frame alone, nothing on top of the phase gate: profiles/r06/ab_garden_gate.txt)
	const uint4 lo = p[0], hi = p[1];
	v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
	v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
}

// One sample at one sparse level when not every lane of the wave has a brick: records where the lane has one, else the level's own index function.
template <bool NETACC>
__device__ __forceinline__ uint32_t sparse_fallback(const GridView& gv, const LevelParams& lp, const CellCoords& c, uint32_t brick) {
	uint32_t v[8];
	if (brick) issue_brick_record_loads(gv, lp, c, brick, v);
	else if (lp.hashed) issue_gathers<true>(gv, lp, c, v);
	else return level_eval_exact<NETACC>(gv, lp, c); // dense level: exact tcnn index incl. the wrap at `count`
	return interpolate<NETACC>(c, v);
}

// Issue the loads of one sample at one level of kind KIND (the level parameters are wave-uniform: scalar registers).
template <int KIND>
__device__ __forceinline__ void issue_level(const GridView& gv, const LevelParams& lp, const CellCoords& c, uint32_t v[8]) {
	if (KIND == KIND_RECORD) issue_record_loads(gv, lp, c, v);
	else if (KIND == KIND_HASHED) issue_gathers<true>(gv, lp, c, v);
	else issue_gathers<false>(gv, lp, c, v);
}
// (ZERO = false: the caller never looks at the features of its idle lanes -- the render kernel's rounds -- and the select per level is saved)
template <bool ZERO = true>
__device__ __forceinline__ uint32_t zero_if(bool cond, uint32_t v) { return ZERO && cond ? 0u : v; }

// TWO levels (an even one and the odd one after it) of ONE sample, both of kind KIND.  All loads of both levels are issued before the
// first is consumed: a round is a chain of dependent memory round trips, and when few waves are left on a CU (the end of a frame)
// its latency, not its throughput, sets the frame time.  Idle lanes gather for position 0 (one shared cache line) so that the code
// stays branch-free; their result is zeroed.
template <int KIND, bool NETACC = false, bool ZERO = true>
__device__ __forceinline__ void level_eval_two(const GridView& gv, const LevelParams& lp0, const LevelParams& lp1, f3 pos, bool act, uint32_t& f0, uint32_t& f1) {
	const f3 q = act ? pos : mk3(0.f, 0.f, 0.f);
	const bool incube = KIND == KIND_RECORD || KIND == KIND_SPARSE; // (record kinds are only chosen for waves whose samples all lie in [0,1]^3)
	const CellCoords c0 = incube ? cell_coords_incube(lp0, q) : cell_coords(lp0, q), c1 = incube ? cell_coords_incube(lp1, q) : cell_coords(lp1, q);
	if (KIND == KIND_SPARSE) {
		uint32_t b0 = brick_entry(gv, lp0, c0), b1 = brick_entry(gv, lp1, c1);
		if (!act) { b0 = 1u; b1 = 1u; } // idle lanes read the level's first record
		uint32_t v0[8], v1[8];
		if (__builtin_expect(__all(b0 != 0u && b1 != 0u), 1)) {
			issue_brick_record_loads(gv, lp0, c0, b0, v0);
			issue_brick_record_loads(gv, lp1, c1, b1, v1);
		} else { // some lane stands where the mask promised no lookups: that lane gathers the level's native way (same values).  A sparse level need
			// not be hashed (a small dense budget leaves dense levels to the sparse records): those lanes take the exact dense index.
			f0 = sparse_fallback<NETACC>(gv, lp0, c0, b0);
			f1 = sparse_fallback<NETACC>(gv, lp1, c1, b1);
			f0 = zero_if<ZERO>(!act, f0);
			f1 = zero_if<ZERO>(!act, f1);
			return;
		}
		f0 = zero_if<ZERO>(!act, interpolate<NETACC>(c0, v0));
		f1 = zero_if<ZERO>(!act, interpolate<NETACC>(c1, v1));
		return;
	}
	if (KIND == KIND_DENSE && __builtin_expect(__any(dense_needs_slow(lp0, c0) || dense_needs_slow(lp1, c1)), 0)) { // exact tcnn wrap for samples outside [0,1)^3: rare
		f0 = level_eval_exact<NETACC>(gv, lp0, c0);
		f1 = level_eval_exact<NETACC>(gv, lp1, c1);
	} else {
		uint32_t v0[8], v1[8];
		issue_level<KIND>(gv, lp0, c0, v0);
		issue_level<KIND>(gv, lp1, c1, v1);
		f0 = interpolate<NETACC>(c0, v0);
		f1 = interpolate<NETACC>(c1, v1);
	}
	f0 = zero_if<ZERO>(!act, f0);
	f1 = zero_if<ZERO>(!act, f1);
}
// FOUR record levels (two consecutive pairs) of one sample with all their loads in flight at once: a round is a chain of dependent memory round
// trips and the gather is eight of them; the twelve record levels then cost three trips instead of six.  The price is registers (32 loaded dwords
// instead of 16): the interpolation weights are therefore NOT kept across the loads but recomputed from the position (9 instructions per level),
// behind an opaque copy of the position so that the compiler cannot keep the first set alive.
template <bool NETACC = false, bool ZERO = true>
__device__ __forceinline__ void record_eval_four(const GridView& gv, const LevelParams& lp0, const LevelParams& lp1, const LevelParams& lp2, const LevelParams& lp3, f3 pos, bool act,
                                                 uint32_t& f0, uint32_t& f1, uint32_t& f2, uint32_t& f3_) {
	f3 q = act ? pos : mk3(0.f, 0.f, 0.f);
	uint32_t v0[8], v1[8], v2[8], v3[8];
	issue_record_loads(gv, lp0, cell_coords_incube(lp0, q), v0);
	issue_record_loads(gv, lp1, cell_coords_incube(lp1, q), v1);
	issue_record_loads(gv, lp2, cell_coords_incube(lp2, q), v2);
	issue_record_loads(gv, lp3, cell_coords_incube(lp3, q), v3);
	asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z)); // the weights below are recomputed, not carried across the loads
	f0 = zero_if<ZERO>(!act, interpolate<NETACC>(cell_coords_incube(lp0, q), v0));
	f1 = zero_if<ZERO>(!act, interpolate<NETACC>(cell_coords_incube(lp1, q), v1));
	f2 = zero_if<ZERO>(!act, interpolate<NETACC>(cell_coords_incube(lp2, q), v2));
	f3_ = zero_if<ZERO>(!act, interpolate<NETACC>(cell_coords_incube(lp3, q), v3));
}

// FOUR sparse levels (two consecutive pairs of brick-record levels: levels 8..11 of an aabb-16 scene) in TWO round trips instead of four: the four brick-table entries
// first, then -- when every lane of the wave has a brick at every one of them, the usual case inside the occupancy mask -- all eight 16-byte record loads at once (round 6:
// the garden frame's round is a chain of ~12 dependent trips of 3-4 us each).  Weights recomputed behind the loads, as above.  Returns false (nothing written) when
// some lane lacks a brick: the caller then takes the pairs one by one (level_eval_two's fallback).
template <bool NETACC = false, bool ZERO = true>
__device__ __forceinline__ bool sparse_eval_four(const GridView& gv, const LevelParams& lp0, const LevelParams& lp1, const LevelParams& lp2, const LevelParams& lp3, f3 pos, bool act,
                                                 uint32_t& f0, uint32_t& f1, uint32_t& f2, uint32_t& f3_) {
	f3 q = act ? pos : mk3(0.f, 0.f, 0.f);
	const CellCoords c0 = cell_coords_incube(lp0, q), c1 = cell_coords_incube(lp1, q), c2 = cell_coords_incube(lp2, q), c3 = cell_coords_incube(lp3, q);
	uint32_t b0 = brick_entry(gv, lp0, c0), b1 = brick_entry(gv, lp1, c1), b2 = brick_entry(gv, lp2, c2), b3 = brick_entry(gv, lp3, c3);
	if (!act) { b0 = 1u; b1 = 1u; b2 = 1u; b3 = 1u; } // idle lanes read the levels' first records
	if (!__builtin_expect(__all(b0 != 0u && b1 != 0u && b2 != 0u && b3 != 0u), 1)) return false;
	uint32_t v0[8], v1[8], v2[8], v3[8];
	issue_brick_record_loads(gv, lp0, c0, b0, v0);
	issue_brick_record_loads(gv, lp1, c1, b1, v1);
	issue_brick_record_loads(gv, lp2, c2, b2, v2);
	issue_brick_record_loads(gv, lp3, c3, b3, v3);
	asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z)); // the weights below are recomputed, not carried across the loads
	f0 = zero_if<ZERO>(!act, interpolate<NETACC>(cell_coords_incube(lp0, q), v0));
	f1 = zero_if<ZERO>(!act, interpolate<NETACC>(cell_coords_incube(lp1, q), v1));
	f2 = zero_if<ZERO>(!act, interpolate<NETACC>(cell_coords_incube(lp2, q), v2));
	f3_ = zero_if<ZERO>(!act, interpolate<NETACC>(cell_coords_incube(lp3, q), v3));
	return true;
}

// The same for FOUR hashed levels (the two pairs behind the cell records: levels 12..15 of base.json's table): 32 single-dword gathers in flight, one round trip
// instead of two.  The hashes need the cells before the loads; the weights are recomputed behind them (as above).
// INCUBE: every sample of the wave lies in [0, 1]^3 (the caller's wave-uniform test): truncation / v_fract instead of floor / subtract, as for the records
template <bool NETACC = false, bool ZERO = true, bool INCUBE = false>
__device__ __forceinline__ void hashed_eval_four(const GridView& gv, const LevelParams& lp0, const LevelParams& lp1, const LevelParams& lp2, const LevelParams& lp3, f3 pos, bool act,
                                                 uint32_t& f0, uint32_t& f1, uint32_t& f2, uint32_t& f3_) {
	f3 q = act ? pos : mk3(0.f, 0.f, 0.f);
	uint32_t v0[8], v1[8], v2[8], v3[8];
	auto cc = [](const LevelParams& lp, f3 x) { return INCUBE ? cell_coords_incube(lp, x) : cell_coords(lp, x); };
	issue_gathers<true>(gv, lp0, cc(lp0, q), v0);
	issue_gathers<true>(gv, lp1, cc(lp1, q), v1);
	issue_gathers<true>(gv, lp2, cc(lp2, q), v2);
	issue_gathers<true>(gv, lp3, cc(lp3, q), v3);
	asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z)); // the weights below are recomputed, not carried across the loads
	f0 = zero_if<ZERO>(!act, interpolate<NETACC>(cc(lp0, q), v0));
	f1 = zero_if<ZERO>(!act, interpolate<NETACC>(cc(lp1, q), v1));
	f2 = zero_if<ZERO>(!act, interpolate<NETACC>(cc(lp2, q), v2));
	f3_ = zero_if<ZERO>(!act, interpolate<NETACC>(cc(lp3, q), v3));
}

// One level of one sample, kind decided at run time (wave-uniform): the pairs whose two levels are of different kinds (the one
// dense | hashed pair of a model without cell records, a records | no-records boundary at an odd level) come here, level after level.
template <bool NETACC = false, bool ZERO = true>
__device__ __forceinline__ uint32_t level_eval_one(const GridView& gv, const LevelParams& lp, bool use_record, f3 pos, bool act) {
	const f3 q = act ? pos : mk3(0.f, 0.f, 0.f);
	const CellCoords c = cell_coords(lp, q);
	uint32_t f;
	uint32_t v[8];
	if (use_record) {
		issue_level<KIND_RECORD>(gv, lp, c, v);
		f = interpolate<NETACC>(c, v);
	} else if (lp.hashed) {
		issue_level<KIND_HASHED>(gv, lp, c, v);
		f = interpolate<NETACC>(c, v);
	} else if (__builtin_expect(__any(dense_needs_slow(lp, c)), 0)) {
		f = level_eval_exact<NETACC>(gv, lp, c);
	} else {
		issue_level<KIND_DENSE>(gv, lp, c, v);
		f = interpolate<NETACC>(c, v);
	}
	return zero_if<ZERO>(!act, f);
}

// All 16 levels of the lane's OWN sample -> the wave's feature slab, two levels (2 it, 2 it + 1) per iteration.
// A lane used to gather half the levels of its own sample and half the levels of its partner's (lane ^ 32), which made the level a
// per-lane quantity: twelve VGPRs of level parameters read from LDS per iteration, the partner's position held in registers, a
// fourth "mixed" kind.  Now every lane of the wave works on the same two levels, so their parameters are wave-uniform -- scalar
// loads straight from the kernel's argument segment (`lv` = DeviceModel::levels), no VGPR, no LDS read -- and the values are the
// same bits as before: the arithmetic per (sample, level) did not change.  The slab layout the MLP reads is unchanged too:
// feat[it][0][l] = level 2 it + g(l) of lane l's sample, feat[it][1][l] = the same level of lane (l ^ 32)'s sample; so the level
// of the lane's own parity goes to [0][lane] and the other one to [1][lane ^ 32] (a conflict-free permutation of the banks).
template <bool NETACC = false, bool QUADS = false, bool ZERO = true, int GATE = 0> // (GATE: 0, or the largest number of L2 phases -- trailing hashed level pairs -- to gate)
__device__ __forceinline__ void encode_to_lds(const GridView& gv, const LevelParams* __restrict__ lv, const ModelLds& ml, FeatLds& fl, int lane, int g, f3 pos, bool act) {
	// the records cover [0,1]^3; a wave with a sample outside it (a warped sample of an edit, rarely) gathers the native way
	const bool outside = __any(act && outside_unit_cube(pos));
	const uint32_t* kinds = outside ? ml.kinds_native : ml.kinds;
	const bool one_line = __builtin_amdgcn_readfirstlane(ml.one_line) != 0u; // profiling (NRS_DEBUG & 1): every gather of a wave hits one 128-byte line
	int it = 0, it_end = 8;
	// GATE (round 6, the instantiation for cone-stepping scenes -- aabb_scale > 1; profiles/r06_garden.md): there the four finest levels are hashed tables of 2 MB each whose
	// lines no two samples of a wave share -- 8 MB of hot lines against the 4 MB L2 of an XCD: alone, each PAIR of them hits (0.15 L2 misses per sample), together they
	// thrash (7.8 of the frame's 11.5).  So the L2 is time-multiplexed: the wall clock (100 MHz, the same on every XCD) is cut into phases of 2^kGateShift ticks; levels
	// 12-13 are gathered in even phases and 14-15 in odd ones by every wave of the GPU, whichever phase a wave arrives in first, with the record levels in between as
	// filler; a wave waits at most kGateWaitCap eighths of a phase for the other one, then goes ahead.  Results cannot change (the same loads in another order).
	// (round 6, late: the same with THREE or FOUR phases when the pairs below are hashed without records too -- the knee of the records budget gathers levels 10..15
	// from 12 MB of tables, a model without sparse records levels 8..15 from 16 MB: gate_m = the number of trailing hashed pairs, one phase each.)
	// Phase length: 2^kGateShift ticks (20.5 us) for two and three phases, half of it for four (profiles/r06/ab_gate_tune_*.txt: 10 / 20 / 41 us x wait caps of a quarter / half
	// a phase, per number of phases).
	uint32_t gate_m = 0, gate_ph = 0, gate_shift = kGateShift;
	auto gate_phase = [&](unsigned long long now) { return ((uint32_t)(now >> gate_shift)) % gate_m; };
	auto gate_wait = [&](uint32_t ph) {
		const unsigned long long t_in = wall_clock64(), cap = ((unsigned long long)kGateWaitCap << gate_shift) >> 3;
		for (;;) {
			const unsigned long long now = wall_clock64();
			if (gate_phase(now) == ph || now - t_in >= cap) break;
			__builtin_amdgcn_s_sleep(8);
		}
	};
	auto gate_pair = [&](int itp) {
		uint32_t f0, f1;
		level_eval_two<KIND_HASHED, NETACC, ZERO>(gv, lv[2 * itp], lv[2 * itp + 1], pos, act, f0, f1);
		fl.feat[itp][0][lane] = g ? f1 : f0;
		fl.feat[itp][1][lane ^ 32] = g ? f0 : f1;
	};
	if (GATE && !one_line) {
		uint32_t n_hashed = 0; // trailing hashed level pairs (wave-uniform: scalar loop)
		for (int itp = 7; itp >= 0 && __builtin_amdgcn_readfirstlane(kinds[itp]) == KIND_HASHED; --itp) ++n_hashed;
		gate_m = (n_hashed >= 2u && n_hashed <= (uint32_t)GATE) ? n_hashed : 0u; // (more hashed pairs than phases: no gate)
		if (gate_m >= 4u) gate_shift = kGateShift - 1u;
	}
	if (GATE && gate_m) {
		const unsigned long long now = wall_clock64();
		const unsigned long long kPhaseMask = (1ull << gate_shift) - 1ull;
		gate_ph = gate_phase(now);
		if ((now & kPhaseMask) > kPhaseMask * 13ull / 16ull) { gate_ph = (gate_ph + 1u) % gate_m; gate_wait(gate_ph); } // (late in a phase: this gather would run into the next one -- take that)
		gate_pair(8 - (int)gate_m + (int)gate_ph);
		it_end = 8 - (int)gate_m;
	}
	#pragma unroll 1
	while (it < it_end) {
		LevelParams lp0 = lv[2 * it], lp1 = lv[2 * it + 1];
		if (one_line) { lp0.hashed = lp1.hashed = 1u; lp0.mask = lp1.mask = 31u; lp0.offset = lp1.offset = 0u; lp0.count = lp1.count = 32u; }
		const uint32_t kind = __builtin_amdgcn_readfirstlane(kinds[it]);
		if (QUADS && kind == KIND_RECORD && it + 1 < 8 && __builtin_amdgcn_readfirstlane(kinds[it + 1]) == KIND_RECORD) {
			uint32_t f0, f1, f2, f3_;
			record_eval_four<NETACC, ZERO>(gv, lp0, lp1, lv[2 * it + 2], lv[2 * it + 3], pos, act, f0, f1, f2, f3_);
			fl.feat[it][0][lane] = g ? f1 : f0;
			fl.feat[it][1][lane ^ 32] = g ? f0 : f1;
			fl.feat[it + 1][0][lane] = g ? f3_ : f2;
			fl.feat[it + 1][1][lane ^ 32] = g ? f2 : f3_;
			it += 2;
			continue;
		}
		if (QUADS && kind == KIND_SPARSE && it + 1 < it_end && __builtin_amdgcn_readfirstlane(kinds[it + 1]) == KIND_SPARSE) {
			uint32_t f0, f1, f2, f3_;
			if (sparse_eval_four<NETACC, ZERO>(gv, lp0, lp1, lv[2 * it + 2], lv[2 * it + 3], pos, act, f0, f1, f2, f3_)) {
				fl.feat[it][0][lane] = g ? f1 : f0;
				fl.feat[it][1][lane ^ 32] = g ? f0 : f1;
				fl.feat[it + 1][0][lane] = g ? f3_ : f2;
				fl.feat[it + 1][1][lane ^ 32] = g ? f2 : f3_;
				it += 2;
				continue;
			}
		}
		if (QUADS && kind == KIND_HASHED && it + 1 < 8 && __builtin_amdgcn_readfirstlane(kinds[it + 1]) == KIND_HASHED) {
			uint32_t f0, f1, f2, f3_;
			if (!outside) hashed_eval_four<NETACC, ZERO, true>(gv, lp0, lp1, lv[2 * it + 2], lv[2 * it + 3], pos, act, f0, f1, f2, f3_);
			else hashed_eval_four<NETACC, ZERO, false>(gv, lp0, lp1, lv[2 * it + 2], lv[2 * it + 3], pos, act, f0, f1, f2, f3_);
			fl.feat[it][0][lane] = g ? f1 : f0;
			fl.feat[it][1][lane ^ 32] = g ? f0 : f1;
			fl.feat[it + 1][0][lane] = g ? f3_ : f2;
			fl.feat[it + 1][1][lane ^ 32] = g ? f2 : f3_;
			it += 2;
			continue;
		}
		uint32_t f0, f1;
		if (kind == KIND_RECORD) level_eval_two<KIND_RECORD, NETACC, ZERO>(gv, lp0, lp1, pos, act, f0, f1);
		else if (kind == KIND_HASHED) level_eval_two<KIND_HASHED, NETACC, ZERO>(gv, lp0, lp1, pos, act, f0, f1);
		else if (kind == KIND_DENSE) level_eval_two<KIND_DENSE, NETACC, ZERO>(gv, lp0, lp1, pos, act, f0, f1);
		else if (kind == KIND_SPARSE) level_eval_two<KIND_SPARSE, NETACC, ZERO>(gv, lp0, lp1, pos, act, f0, f1);
		else if (kind == KIND_SKIP) { f0 = 0u; f1 = 0u; } // (measurement only: stage_model_to_lds)
		else {
			f0 = level_eval_one<NETACC, ZERO>(gv, lp0, !outside && !one_line && lp0.cached == 1u, pos, act);
			f1 = level_eval_one<NETACC, ZERO>(gv, lp1, !outside && !one_line && lp1.cached == 1u, pos, act);
		}
		fl.feat[it][0][lane] = g ? f1 : f0;
		fl.feat[it][1][lane ^ 32] = g ? f0 : f1;
		++it;
	}
	if (GATE && gate_m) {
		#pragma unroll 1
		for (uint32_t k = 1; k < gate_m; ++k) {
			const uint32_t ph = (gate_ph + k) % gate_m;
			gate_wait(ph);
			gate_pair(8 - (int)gate_m + (int)ph);
		}
	}
	// feat[..][1][lane ^ 32] is another lane's slot: order the wave's writes before load_features' reads (no instruction: LDS operations of a wave
	// stay in order; this keeps the compiler from moving a read above the write it cannot see through the xor)
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// B operand of block b for k-step ks: levels it = 4ks..4ks+3 of sample (b, j) as this lane gathered them
__device__ __forceinline__ half8 load_features(const FeatLds& fl, int lane, int sel, int ks) {
	u32x4 v;
	v[0] = fl.feat[4 * ks + 0][sel][lane];
	v[1] = fl.feat[4 * ks + 1][sel][lane];
	v[2] = fl.feat[4 * ks + 2][sel][lane];
	v[3] = fl.feat[4 * ks + 3][sel][lane];
	return __builtin_bit_cast(half8, v);
}

// SH degree 4 (tcnn SphericalHarmonics) of a direction given as (d+1)/2: the 8 coefficients 8g..8g+7 this lane owns.
__device__ __forceinline__ half8 encode_sh4(int g, f3 dir01) {
	float x = dir01.x * 2.f - 1.f, y = dir01.y * 2.f - 1.f, z = dir01.z * 2.f - 1.f;
	float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
	float o[8];
	if (g == 0) {
		o[0] = 0.28209479177387814f;
		o[1] = -0.48860251190291987f * y;
		o[2] = 0.48860251190291987f * z;
		o[3] = -0.48860251190291987f * x;
		o[4] = 1.0925484305920792f * xy;
		o[5] = -1.0925484305920792f * yz;
		o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
		o[7] = -1.0925484305920792f * xz;
	} else {
		o[0] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
		o[1] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
		o[2] = 2.8906114426405538f * xy * z;
		o[3] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
		o[4] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
		o[5] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
		o[6] = 1.4453057213202769f * z * (x2 - y2);
		o[7] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
	}
	half8 r;
	#pragma unroll
	for (int e = 0; e < 8; ++e) r[e] = (_Float16)o[e];
	return r;
}

// The same for two directions (the lane's own sample and its partner's) in packed fp32: every product and sum is the scalar
// one (+1.7 % on the bench).  The same treatment of the trilinear weights measured -11 %, of the cell coordinates 0.
__device__ __forceinline__ void encode_sh4_2(int g, f3 dirA, f3 dirB, half8& outA, half8& outB) {
	const f2 two = {2.f, 2.f}, one = {1.f, 1.f};
	const f2 x = (f2){dirA.x, dirB.x} * two - one, y = (f2){dirA.y, dirB.y} * two - one, z = (f2){dirA.z, dirB.z} * two - one;
	const f2 xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
	f2 o[8];
	if (g == 0) {
		o[0] = (f2){0.28209479177387814f, 0.28209479177387814f};
		o[1] = -0.48860251190291987f * y;
		o[2] = 0.48860251190291987f * z;
		o[3] = -0.48860251190291987f * x;
		o[4] = 1.0925484305920792f * xy;
		o[5] = -1.0925484305920792f * yz;
		o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
		o[7] = -1.0925484305920792f * xz;
	} else {
		o[0] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
		o[1] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
		o[2] = 2.8906114426405538f * xy * z;
		o[3] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
		o[4] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
		o[5] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
		o[6] = 1.4453057213202769f * z * (x2 - y2);
		o[7] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
	}
	#pragma unroll
	for (int e = 0; e < 8; ++e) { outA[e] = (_Float16)o[e].x; outB[e] = (_Float16)o[e].y; }
}

// ReLU + fp16 rounding of 8 accumulator rows.  max(round(x), 0) == round(max(x, 0)); done on packed halfs.
__device__ __forceinline__ half8 relu_pack(const floatx16& d, int base) {
	half8 r;
	#pragma unroll
	for (int e = 0; e < 8; ++e) r[e] = (_Float16)d[base + e];
	const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
	return __builtin_elementwise_max(r, zero);
}
__device__ __forceinline__ half8 pack(const floatx16& d, int base) {
	half8 r;
	#pragma unroll
	for (int e = 0; e < 8; ++e) r[e] = (_Float16)d[base + e];
	return r;
}
__device__ __forceinline__ floatx16 zero16() {
	floatx16 z;
	#pragma unroll
	for (int i = 0; i < 16; ++i) z[i] = 0.f;
	return z;
}

#define NRS_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define NRS_FRAG_SEL(h) (24 + (h))
#define NRS_FRAG_BWD(ks) (26 + (ks)) // (HBM only: DeviceModel::wfrag + ...; see nrs_internal.h)
// ACC16 (nrs_mlp_acc FP16): the running sums are rounded to fp16 after every 16-wide k step, the model of tiny-cuda-nn's fp16 accumulator fragments.
// (round 4) the rounded sums go back into the accumulator registers THROUGH THE MATRIX CORE -- pack to fp16 (8 v_cvt_pk_f16_f32: the
// rounding), then MFMA(Sel0, lo, 0) and MFMA(Sel1, hi, .) with the two constant 0 / 1 fragments, which reproduce the packed values exactly in fp32
// (1.0 x h summed with zeros) in the D layout -- and the k step's own MFMA accumulates on top as before.  8 VALU slots per rounding instead of 24-32 on a
// kernel that is VALU-bound (the packed-conversion round trip on the vector pipe: profiles/r04/ab_numerics_variants.txt), paid with two issues on a pipe
// that is 10 % busy; same values (tests/test_gpu_numerics.py).
template <bool ACC16>
__device__ __forceinline__ floatx16 mfma_step(const half8* lds_w, int lane, half8 a, half8 b, floatx16 c) {
	if (ACC16) {
		half8 lo, hi;
		#pragma unroll
		for (int e = 0; e < 8; ++e) { lo[e] = (_Float16)c[e]; hi[e] = (_Float16)c[8 + e]; }
		floatx16 z;
		#pragma unroll
		for (int i = 0; i < 16; ++i) z[i] = 0.f;
		__builtin_amdgcn_sched_barrier(0); // (keeps the selection fragments' LDS reads of later steps from being hoisted over this one: registers)
		c = NRS_MFMA(lds_w[NRS_FRAG_SEL(0) * 64 + lane], lo, z);
		c = NRS_MFMA(lds_w[NRS_FRAG_SEL(1) * 64 + lane], hi, c);
	}
	return NRS_MFMA(a, b, c);
}
// the first k step of a layer: nothing to round yet
__device__ __forceinline__ floatx16 mfma_first(half8 a, half8 b) {
	floatx16 z;
	#pragma unroll
	for (int i = 0; i < 16; ++i) z[i] = 0.f;
	return NRS_MFMA(a, b, z);
}
// Scheduling fence between MLP stages: without it hipcc hoists all 24 weight-fragment LDS reads (96 VGPRs) to the top of
// the MLP, which costs a wave of occupancy.  The gather, not the MLP, is the phase that needs the latency hiding.
#define NRS_STAGE_FENCE() __builtin_amdgcn_sched_barrier(0)

// Density MLP 32 -> 64 (ReLU) -> 16 for one 32-sample block.  x0/x1: features (k-steps 0/1).
// Returns the fp16-rounded outputs as the next B operand: element e of lane (j, g) = output row (e&3) + 8*(e>>2) + 4g.
template <bool ACC16 = false>
__device__ __forceinline__ half8 density_mlp(const half8* lds_w, int lane, half8 x0, half8 x1) {
	// hidden rows 0..31 then 32..63, each reduced to its two packed B operands before the next accumulator is started
	// (ACC16: mfma_step rounds the sum it is handed before it adds its own k step; the last sum of a layer is rounded by relu_pack / pack themselves)
	floatx16 h = mfma_first(lds_w[NRS_FRAG_D1(0, 0) * 64 + lane], x0);
	h = mfma_step<ACC16>(lds_w, lane, lds_w[NRS_FRAG_D1(0, 1) * 64 + lane], x1, h);
	const half8 p0 = relu_pack(h, 0), p1 = relu_pack(h, 8);
	NRS_STAGE_FENCE();
	h = mfma_first(lds_w[NRS_FRAG_D1(1, 0) * 64 + lane], x0);
	h = mfma_step<ACC16>(lds_w, lane, lds_w[NRS_FRAG_D1(1, 1) * 64 + lane], x1, h);
	const half8 p2 = relu_pack(h, 0), p3 = relu_pack(h, 8);
	NRS_STAGE_FENCE();
	floatx16 o = mfma_first(lds_w[NRS_FRAG_D2(0) * 64 + lane], p0);
	o = mfma_step<ACC16>(lds_w, lane, lds_w[NRS_FRAG_D2(1) * 64 + lane], p1, o);
	o = mfma_step<ACC16>(lds_w, lane, lds_w[NRS_FRAG_D2(2) * 64 + lane], p2, o);
	o = mfma_step<ACC16>(lds_w, lane, lds_w[NRS_FRAG_D2(3) * 64 + lane], p3, o);
	NRS_STAGE_FENCE();
	return pack(o, 0);
}

// One 64 -> 64 ReLU layer of the rgb MLP whose fragments `w` (8: [mb][ks]) may live in LDS or in HBM: b0..b3 in, the packed B operands of the next layer out.
template <bool ACC16>
__device__ __forceinline__ void hidden_layer_64(const half8* lds_w, const half8* w, int lane, half8& b0, half8& b1, half8& b2, half8& b3) {
	floatx16 c = mfma_first(w[0 * 64 + lane], b0);
	c = mfma_step<ACC16>(lds_w, lane, w[1 * 64 + lane], b1, c);
	c = mfma_step<ACC16>(lds_w, lane, w[2 * 64 + lane], b2, c);
	c = mfma_step<ACC16>(lds_w, lane, w[3 * 64 + lane], b3, c);
	const half8 q0 = relu_pack(c, 0), q1 = relu_pack(c, 8);
	NRS_STAGE_FENCE();
	c = mfma_first(w[4 * 64 + lane], b0);
	c = mfma_step<ACC16>(lds_w, lane, w[5 * 64 + lane], b1, c);
	c = mfma_step<ACC16>(lds_w, lane, w[6 * 64 + lane], b2, c);
	c = mfma_step<ACC16>(lds_w, lane, w[7 * 64 + lane], b3, c);
	const half8 q2 = relu_pack(c, 0), q3 = relu_pack(c, 8);
	NRS_STAGE_FENCE();
	b0 = q0; b1 = q1; b2 = q2; b3 = q3;
}
// RGB MLP [density out 16 | SH 16] -> 64 -> 64 -> 16 (3 used) for one block.  Same output row map.
// DEEP instantiations take `deep_w` (wave-uniform; DeviceModel::wfrag when DeviceModel::rgb_deep, else null): a third hidden layer between the second and the
// output layer, its fragments read from HBM (base_3layer.json; the other members of the family are lowered onto the two-layer shape, nrs_api.cpp lower_weights).
template <bool ACC16 = false, bool DEEP = false>
__device__ __forceinline__ half8 rgb_mlp(const half8* lds_w, int lane, half8 din, half8 sh, const half8* deep_w = nullptr) {
	floatx16 a = mfma_first(lds_w[NRS_FRAG_R1(0, 0) * 64 + lane], din);
	a = mfma_step<ACC16>(lds_w, lane, lds_w[NRS_FRAG_R1(0, 1) * 64 + lane], sh, a);
	half8 b0 = relu_pack(a, 0), b1 = relu_pack(a, 8);
	NRS_STAGE_FENCE();
	a = mfma_first(lds_w[NRS_FRAG_R1(1, 0) * 64 + lane], din);
	a = mfma_step<ACC16>(lds_w, lane, lds_w[NRS_FRAG_R1(1, 1) * 64 + lane], sh, a);
	half8 b2 = relu_pack(a, 0), b3 = relu_pack(a, 8);
	NRS_STAGE_FENCE();
	hidden_layer_64<ACC16>(lds_w, lds_w + NRS_FRAG_R2(0, 0) * 64, lane, b0, b1, b2, b3);
	if (DEEP && deep_w) hidden_layer_64<ACC16>(lds_w, deep_w + NRS_FRAG_R2B(0, 0) * 64, lane, b0, b1, b2, b3);
	floatx16 o = mfma_first(lds_w[NRS_FRAG_R3(0) * 64 + lane], b0);
	o = mfma_step<ACC16>(lds_w, lane, lds_w[NRS_FRAG_R3(1) * 64 + lane], b1, o);
	o = mfma_step<ACC16>(lds_w, lane, lds_w[NRS_FRAG_R3(2) * 64 + lane], b2, o);
	o = mfma_step<ACC16>(lds_w, lane, lds_w[NRS_FRAG_R3(3) * 64 + lane], b3, o);
	NRS_STAGE_FENCE();
	return pack(o, 0);
}

// ---- network introspection: render modes EncodingVis and Normals (render_body's INTRO instantiation only) ----------------------------------------
// tiny-cuda-nn's visualize_activation / input_gradient as restated in oracle/nrs_oracle.cpp (network_activation_one, density_input_gradient_one).
// register r (wave-uniform) of a D tile
__device__ __forceinline__ float pick16(const floatx16& d, int r) {
	float v = d[0];
	#pragma unroll
	for (int i = 1; i < 16; ++i) v = (r == i) ? d[i] : v;
	return v;
}
__device__ __forceinline__ _Float16 pick8(const half8& h, int e) {
	_Float16 v = h[0];
	#pragma unroll
	for (int i = 1; i < 8; ++i) v = (e == i) ? h[i] : v;
	return v;
}
// Activation `unit` of hidden layer `layer` (1: density MLP hidden, 3 / 4 / 5: rgb MLP hidden 1 / 2 / 3 -- 5 with deep_w only) for one 32-sample block: the value of sample column j sits,
// after the call, in the lanes whose half (lane >> 5) equals tile_half(unit); the other half returns another row.  din = density_mlp's output (layers 3, 4).
__device__ __forceinline__ int tile_half(uint32_t unit) { return (int)((unit >> 2) & 1u); }
template <bool ACC16>
__device__ __forceinline__ float mlp_hidden_activation(const half8* lds_w, int lane, half8 x0, half8 x1, half8 din, half8 sh, uint32_t layer, uint32_t unit, const half8* deep_w = nullptr) {
	const int mb = (int)((unit >> 5) & 1u), R = (int)(unit & 31u), r = (R & 3) + 4 * (R >> 3); // D register of row R in its lane half
	floatx16 t;
	if (layer == 1u) {
		t = mfma_first(lds_w[NRS_FRAG_D1(mb, 0) * 64 + lane], x0);
		t = mfma_step<ACC16>(lds_w, lane, lds_w[NRS_FRAG_D1(mb, 1) * 64 + lane], x1, t);
	} else if (layer == 3u) {
		t = mfma_first(lds_w[NRS_FRAG_R1(mb, 0) * 64 + lane], din);
		t = mfma_step<ACC16>(lds_w, lane, lds_w[NRS_FRAG_R1(mb, 1) * 64 + lane], sh, t);
	} else {
		floatx16 a = mfma_first(lds_w[NRS_FRAG_R1(0, 0) * 64 + lane], din);
		a = mfma_step<ACC16>(lds_w, lane, lds_w[NRS_FRAG_R1(0, 1) * 64 + lane], sh, a);
		half8 b0 = relu_pack(a, 0), b1 = relu_pack(a, 8);
		NRS_STAGE_FENCE();
		a = mfma_first(lds_w[NRS_FRAG_R1(1, 0) * 64 + lane], din);
		a = mfma_step<ACC16>(lds_w, lane, lds_w[NRS_FRAG_R1(1, 1) * 64 + lane], sh, a);
		half8 b2 = relu_pack(a, 0), b3 = relu_pack(a, 8);
		NRS_STAGE_FENCE();
		const half8* w = lds_w + NRS_FRAG_R2(mb, 0) * 64;
		if (layer == 5u && deep_w) { // the third hidden layer: behind the whole second one
			hidden_layer_64<ACC16>(lds_w, lds_w + NRS_FRAG_R2(0, 0) * 64, lane, b0, b1, b2, b3);
			w = deep_w + NRS_FRAG_R2B(mb, 0) * 64;
		}
		t = mfma_first(w[0 * 64 + lane], b0);
		t = mfma_step<ACC16>(lds_w, lane, w[1 * 64 + lane], b1, t);
		t = mfma_step<ACC16>(lds_w, lane, w[2 * 64 + lane], b2, t);
		t = mfma_step<ACC16>(lds_w, lane, w[3 * 64 + lane], b3, t);
	}
	NRS_STAGE_FENCE();
	const _Float16 h = (_Float16)pick16(t, r);
	return (float)(h > (_Float16)0 ? h : (_Float16)0); // the stored activation: ReLU, fp16
}
// Backward of 128 * e_0 through the density MLP for one 32-sample block (input_gradient(stream, 3, ...): the one-hot lands on the density network's output
// row 0, nerf_network_full.h:188-195): dL/dhidden[k] = (hidden[k] > 0) * fp16(W2[0][k] * 128) in the layout of the hidden tiles, then
// dL/dfeatures = W1^T dL/dhidden on MFMA (A operands Bwd[ks] from HBM: DeviceModel::wfrag), accumulated as the forward pass is (ACC16).
// out[q] = features (2 L, 2 L + 1) packed, L = level_of_pair(q, lane >> 5), of sample column j.
__device__ __forceinline__ int level_of_pair(int q, int gg) { const int r = 2 * q; return ((r & 3) + 8 * (r >> 2) + 4 * gg) >> 1; }
template <bool ACC16>
__device__ __forceinline__ void density_backward_features(const half8* lds_w, const half8* __restrict__ gfrag, int lane, half8 x0, half8 x1, uint32_t out[8]) {
	floatx16 h = mfma_first(lds_w[NRS_FRAG_D1(0, 0) * 64 + lane], x0);
	h = mfma_step<ACC16>(lds_w, lane, lds_w[NRS_FRAG_D1(0, 1) * 64 + lane], x1, h);
	const half8 p0 = relu_pack(h, 0), p1 = relu_pack(h, 8);
	NRS_STAGE_FENCE();
	h = mfma_first(lds_w[NRS_FRAG_D1(1, 0) * 64 + lane], x0);
	h = mfma_step<ACC16>(lds_w, lane, lds_w[NRS_FRAG_D1(1, 1) * 64 + lane], x1, h);
	const half8 p2 = relu_pack(h, 0), p3 = relu_pack(h, 8);
	NRS_STAGE_FENCE();
	// W2[0][k] for the k's this lane's hidden registers stand for: what lane (0, lane >> 5) of the D2 fragments holds (output row 0)
	const int row0 = lane & 32;
	auto dhidden = [&](const half8& p, int ks) {
		const half8 w = lds_w[NRS_FRAG_D2(ks) * 64 + row0];
		half8 q;
		#pragma unroll
		for (int e = 0; e < 8; ++e) q[e] = p[e] > (_Float16)0 ? (_Float16)(w[e] * (_Float16)128) : (_Float16)0; // x 128 in fp16: the same value as fp16(float(w) * 128)
		return q;
	};
	const half8 q0 = dhidden(p0, 0), q1 = dhidden(p1, 1), q2 = dhidden(p2, 2), q3 = dhidden(p3, 3);
	NRS_STAGE_FENCE();
	floatx16 d = mfma_first(gfrag[NRS_FRAG_BWD(0) * 64 + lane], q0);
	d = mfma_step<ACC16>(lds_w, lane, gfrag[NRS_FRAG_BWD(1) * 64 + lane], q1, d);
	d = mfma_step<ACC16>(lds_w, lane, gfrag[NRS_FRAG_BWD(2) * 64 + lane], q2, d);
	d = mfma_step<ACC16>(lds_w, lane, gfrag[NRS_FRAG_BWD(3) * 64 + lane], q3, d);
	NRS_STAGE_FENCE();
	#pragma unroll
	for (int q = 0; q < 8; ++q) {
		half2v pr = {(_Float16)d[2 * q], (_Float16)d[2 * q + 1]};
		out[q] = __builtin_bit_cast(uint32_t, pr);
	}
}
// kernel_grid's dy_dx and kernel_grid_backward_input for ONE level of one sample (tiny-cuda-nn grid.h as restated in hashgrid_input_gradient_one):
// result[d] += (float)dL_dy[f] * sum over the 4 corner pairs along d of scale * w(other two dims) * (right - left), fp32 with fmaf.  The 8 corners are
// fetched with the level's own (exact) index function.  dl = (dL_dy[2 L], dL_dy[2 L + 1]) packed.
__device__ __forceinline__ void level_input_gradient(const GridView& gv, const LevelParams& lp, f3 pos, uint32_t dl, float result[3]) {
	const CellCoords c = cell_coords(lp, pos);
	uint32_t v[8];
	#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const uint32_t cx = c.gx + (k & 1), cy = c.gy + ((k >> 1) & 1), cz = c.gz + ((k >> 2) & 1);
		uint32_t index = lp.hashed ? ((cx * 1u) ^ (cy * 2654435761u) ^ (cz * 805459861u)) : (cx + cy * lp.resolution + cz * lp.res2);
		index %= lp.count;
		v[k] = grid_load(gv, lp.offset + index);
	}
	const float w[3] = {c.wx, c.wy, c.wz};
	const half2v dlh = __builtin_bit_cast(half2v, dl);
	float grads[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
	#pragma unroll
	for (int gd = 0; gd < 3; ++gd) {
		const int da = gd == 0 ? 1 : 0, db = gd == 2 ? 1 : 2; // the two other dimensions, ascending
		#pragma unroll
		for (int idx = 0; idx < 4; ++idx) {
			float weight = lp.scale;
			weight *= (idx & 1) ? w[da] : 1 - w[da];
			weight *= (idx & 2) ? w[db] : 1 - w[db];
			const int left = ((idx & 1) << da) | (((idx >> 1) & 1) << db), right = left | (1 << gd);
			const half2v hl = __builtin_bit_cast(half2v, v[left]), hr = __builtin_bit_cast(half2v, v[right]);
			grads[0][gd] = fmaf(weight, (float)hr[0] - (float)hl[0], grads[0][gd]);
			grads[1][gd] = fmaf(weight, (float)hr[1] - (float)hl[1], grads[1][gd]);
		}
	}
	#pragma unroll
	for (int f = 0; f < 2; ++f)
		#pragma unroll
		for (int d = 0; d < 3; ++d) result[d] = fmaf((float)dlh[f], grads[f][d], result[d]);
}

// tiny-cuda-nn's roundings as a template value: NUM >= 0 fixes them at compile time (bit 0 grid accumulation in network precision, bit 1 fp16 MLP
// accumulators), kNumRuntime reads them from `nm` (DeviceModel::numerics, wave-uniform) -- both flavours compiled in, one scalar branch.
constexpr int kNumRuntime = -1;
template <int NUM, bool QUADS = false, bool ZERO = true, int GATE = 0>
__device__ __forceinline__ void encode_num(uint32_t nm, const GridView& gv, const LevelParams* __restrict__ lv, const ModelLds& ml, FeatLds& fl, int lane, int g, f3 pos, bool act) {
	if (NUM == kNumRuntime ? (nm & 1u) != 0u : (NUM & 1) != 0) encode_to_lds<true, QUADS, ZERO, GATE>(gv, lv, ml, fl, lane, g, pos, act);
	else encode_to_lds<false, QUADS, ZERO, GATE>(gv, lv, ml, fl, lane, g, pos, act);
}
template <int NUM>
__device__ __forceinline__ half8 density_mlp_num(uint32_t nm, const half8* lds_w, int lane, half8 x0, half8 x1) {
	if (NUM == kNumRuntime ? (nm & 2u) != 0u : (NUM & 2) != 0) return density_mlp<true>(lds_w, lane, x0, x1);
	return density_mlp<false>(lds_w, lane, x0, x1);
}
template <int NUM, bool DEEP = false>
__device__ __forceinline__ half8 rgb_mlp_num(uint32_t nm, const half8* lds_w, int lane, half8 din, half8 sh, const half8* deep_w = nullptr) {
	if (NUM == kNumRuntime ? (nm & 2u) != 0u : (NUM & 2) != 0) return rgb_mlp<true, DEEP>(lds_w, lane, din, sh, deep_w);
	return rgb_mlp<false, DEEP>(lds_w, lane, din, sh, deep_w);
}

// Exchange a value with the partner lane (l ^ 32): one ds_bpermute.
__device__ __forceinline__ float xchg32(float v) { return __shfl_xor(v, 32, 64); }
__device__ __forceinline__ uint32_t xchg32u(uint32_t v) { return (uint32_t)__shfl_xor((int)v, 32, 64); }

} // namespace nrs
