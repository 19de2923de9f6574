// nrs_mlp.cuh -- hash-grid gather + fused MLPs on MFMA for one wavefront (gfx950).
//
// Replaces tiny-cuda-nn's kernel_grid + 2x kernel_mlp_fused + SH encoding + extract_density with one
// register-resident pipeline.  A wave owns 64 samples, processed as two 32-sample MFMA column blocks:
//
//   block b holds the samples of lanes 32b..32b+31.  Lane l = (j = l & 31, g = l >> 5) works for sample j of each
//   block and owns, of that sample, the hash-grid levels L(g, it) = 2*it + g (it = 0..7) -- so the two lane halves
//   split the 16 levels even/odd -- and the SH coefficients 8g..8g+7.
//
// Every layer is computed transposed, H^T[unit][sample] = W[unit][k] * X^T[k][sample], with
// v_mfma_f32_32x32x16_f16: A = a 32x16 weight tile (from LDS, pre-arranged on the host), B = 16 x 32 samples.
// The D tile of one layer (lane = sample column, 16 rows per lane) is, after ReLU + fp16 rounding, directly the B
// operand of the next layer: the k index of an MFMA is free as long as A and B agree, so the host arranges the
// weight tiles in the order the D registers come out (make_weight_fragments in nrs_api.cpp).  No cross-lane
// traffic, no LDS round trip, no global intermediates.
//
// Numerics (stated; parity at the tcnn boundary is unpinned): grid entries fp16, trilinear sum in
// fp32 via fmaf in corner order 0..7, rounded to fp16; MLP products fp16 x fp16 accumulated in fp32 by the MFMA,
// ReLU, rounded to fp16 between layers; outputs rounded to fp16.
#pragma once
#include <hip/hip_runtime.h>
#include "nrs_device.cuh"

namespace nrs {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// fragment indices inside the LDS weight image
#define NRS_FRAG_D1(mb, ks) ((mb) * 2 + (ks))
#define NRS_FRAG_D2(ks) (4 + (ks))
#define NRS_FRAG_R1(mb, ks) (8 + (mb) * 2 + (ks))
#define NRS_FRAG_R2(mb, ks) (12 + (mb) * 4 + (ks))
#define NRS_FRAG_R3(ks) (20 + (ks))

__device__ __forceinline__ uint32_t fast_wrap(uint32_t index, const LevelParams& lp) {
	if (lp.hashed) return index & lp.mask;
	// dense level: index < 2*count for every in-range position; the exact modulo is the (never taken) slow path
	if (index >= lp.count) {
		index -= lp.count;
		if (index >= lp.count) index %= lp.count;
	}
	return index;
}

// Gather + trilinear interpolation of the 8 levels this lane owns for one sample position (warped, [0,1]^3).
// Result: 16 fp16 features as two MFMA B operands (k-step 0: it 0..3, k-step 1: it 4..7; element 2*(it&3)+f).
__device__ __forceinline__ void encode_levels(const uint32_t* __restrict__ grid, const LevelParams* lds_levels, int g, f3 pos, bool active,
                                              half8& k0, half8& k1) {
	_Float16 feat[16];
	#pragma unroll
	for (int it = 0; it < 8; ++it) {
		float acc0 = 0.f, acc1 = 0.f;
		if (active) {
			const LevelParams lp = lds_levels[2 * it + g];
			float px = fmaf(lp.scale, pos.x, 0.5f), py = fmaf(lp.scale, pos.y, 0.5f), pz = fmaf(lp.scale, pos.z, 0.5f);
			float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
			uint32_t gx = (uint32_t)(int)fx, gy = (uint32_t)(int)fy, gz = (uint32_t)(int)fz;
			float wx = px - fx, wy = py - fy, wz = pz - fz;
			uint32_t vals[8];
			#pragma unroll
			for (int c = 0; c < 8; ++c) {
				uint32_t cx = gx + (c & 1), cy = gy + ((c >> 1) & 1), cz = gz + ((c >> 2) & 1);
				uint32_t index = lp.hashed ? ((cx * 1u) ^ (cy * 2654435761u) ^ (cz * 805459861u)) : (cx + cy * lp.resolution + cz * lp.res2);
				index = fast_wrap(index, lp);
				vals[c] = grid[lp.offset + index];
			}
			#pragma unroll
			for (int c = 0; c < 8; ++c) {
				float weight = 1.0f;
				weight *= (c & 1) ? wx : 1.0f - wx;
				weight *= (c & 2) ? wy : 1.0f - wy;
				weight *= (c & 4) ? wz : 1.0f - wz;
				half2v hv = __builtin_bit_cast(half2v, vals[c]);
				acc0 = fmaf(weight, (float)hv[0], acc0);
				acc1 = fmaf(weight, (float)hv[1], acc1);
			}
		}
		feat[2 * it + 0] = (_Float16)acc0;
		feat[2 * it + 1] = (_Float16)acc1;
	}
	#pragma unroll
	for (int e = 0; e < 8; ++e) { k0[e] = feat[e]; k1[e] = feat[8 + e]; }
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

__device__ __forceinline__ half8 relu_pack(const floatx16& d, int base) {
	half8 r;
	#pragma unroll
	for (int e = 0; e < 8; ++e) r[e] = (_Float16)fmaxf(d[base + e], 0.f);
	return r;
}
__device__ __forceinline__ half8 pack(const floatx16& d, int base) {
	half8 r;
	#pragma unroll
	for (int e = 0; e < 8; ++e) r[e] = (_Float16)d[base + e];
	return r;
}

#define NRS_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// Density MLP 32 -> 64 (ReLU) -> 16 for both sample blocks.  x[b][ks]: features of block b.
// dout[b] = fp16-rounded outputs as the next B operand: element e of lane (j, g) = output row (e&3) + 8*(e>>2) + 4g.
__device__ __forceinline__ void density_mlp(const half8* lds_w, int lane, const half8 x[2][2], half8 dout[2]) {
	floatx16 h[2][2]; // [block][mb]
	#pragma unroll
	for (int b = 0; b < 2; ++b)
		#pragma unroll
		for (int mb = 0; mb < 2; ++mb)
			#pragma unroll
			for (int i = 0; i < 16; ++i) h[b][mb][i] = 0.f;
	#pragma unroll
	for (int mb = 0; mb < 2; ++mb)
		#pragma unroll
		for (int ks = 0; ks < 2; ++ks) {
			half8 a = lds_w[NRS_FRAG_D1(mb, ks) * 64 + lane];
			h[0][mb] = NRS_MFMA(a, x[0][ks], h[0][mb]);
			h[1][mb] = NRS_MFMA(a, x[1][ks], h[1][mb]);
		}
	floatx16 o[2];
	#pragma unroll
	for (int b = 0; b < 2; ++b)
		#pragma unroll
		for (int i = 0; i < 16; ++i) o[b][i] = 0.f;
	#pragma unroll
	for (int ks = 0; ks < 4; ++ks) {
		half8 a = lds_w[NRS_FRAG_D2(ks) * 64 + lane];
		o[0] = NRS_MFMA(a, relu_pack(h[0][ks >> 1], 8 * (ks & 1)), o[0]);
		o[1] = NRS_MFMA(a, relu_pack(h[1][ks >> 1], 8 * (ks & 1)), o[1]);
	}
	dout[0] = pack(o[0], 0);
	dout[1] = pack(o[1], 0);
}

// RGB MLP [density out 16 | SH 16] -> 64 -> 64 -> 16 (3 used) for both blocks.  rout[b]: fp16 outputs, same row map.
__device__ __forceinline__ void rgb_mlp(const half8* lds_w, int lane, const half8 din[2], const half8 sh[2], half8 rout[2]) {
	floatx16 h1[2][2];
	#pragma unroll
	for (int b = 0; b < 2; ++b)
		#pragma unroll
		for (int mb = 0; mb < 2; ++mb)
			#pragma unroll
			for (int i = 0; i < 16; ++i) h1[b][mb][i] = 0.f;
	#pragma unroll
	for (int mb = 0; mb < 2; ++mb) {
		half8 a0 = lds_w[NRS_FRAG_R1(mb, 0) * 64 + lane];
		h1[0][mb] = NRS_MFMA(a0, din[0], h1[0][mb]);
		h1[1][mb] = NRS_MFMA(a0, din[1], h1[1][mb]);
		half8 a1 = lds_w[NRS_FRAG_R1(mb, 1) * 64 + lane];
		h1[0][mb] = NRS_MFMA(a1, sh[0], h1[0][mb]);
		h1[1][mb] = NRS_MFMA(a1, sh[1], h1[1][mb]);
	}
	half8 b1[2][4]; // B operands of layer 2: [block][ks]
	#pragma unroll
	for (int b = 0; b < 2; ++b)
		#pragma unroll
		for (int ks = 0; ks < 4; ++ks) b1[b][ks] = relu_pack(h1[b][ks >> 1], 8 * (ks & 1));
	floatx16 h2[2][2];
	#pragma unroll
	for (int b = 0; b < 2; ++b)
		#pragma unroll
		for (int mb = 0; mb < 2; ++mb)
			#pragma unroll
			for (int i = 0; i < 16; ++i) h2[b][mb][i] = 0.f;
	#pragma unroll
	for (int mb = 0; mb < 2; ++mb)
		#pragma unroll
		for (int ks = 0; ks < 4; ++ks) {
			half8 a = lds_w[NRS_FRAG_R2(mb, ks) * 64 + lane];
			h2[0][mb] = NRS_MFMA(a, b1[0][ks], h2[0][mb]);
			h2[1][mb] = NRS_MFMA(a, b1[1][ks], h2[1][mb]);
		}
	floatx16 o[2];
	#pragma unroll
	for (int b = 0; b < 2; ++b)
		#pragma unroll
		for (int i = 0; i < 16; ++i) o[b][i] = 0.f;
	#pragma unroll
	for (int ks = 0; ks < 4; ++ks) {
		half8 a = lds_w[NRS_FRAG_R3(ks) * 64 + lane];
		o[0] = NRS_MFMA(a, relu_pack(h2[0][ks >> 1], 8 * (ks & 1)), o[0]);
		o[1] = NRS_MFMA(a, relu_pack(h2[1][ks >> 1], 8 * (ks & 1)), o[1]);
	}
	rout[0] = pack(o[0], 0);
	rout[1] = pack(o[1], 0);
}

// Exchange a value with the partner lane (l ^ 32): one ds_bpermute.
__device__ __forceinline__ float xchg32(float v) { return __shfl_xor(v, 32, 64); }
__device__ __forceinline__ uint32_t xchg32u(uint32_t v) { return (uint32_t)__shfl_xor((int)v, 32, 64); }

// Cooperative copy of the weight-fragment image and the level table into LDS (all threads of the block).
__device__ __forceinline__ void stage_model_to_lds(const DeviceModel& m, half8* lds_w, LevelParams* lds_levels) {
	const uint4* src = reinterpret_cast<const uint4*>(m.wfrag);
	uint4* dst = reinterpret_cast<uint4*>(lds_w);
	for (uint32_t i = threadIdx.x; i < kWfragBytes / 16; i += blockDim.x) dst[i] = src[i];
	if (threadIdx.x < kLevels) lds_levels[threadIdx.x] = m.levels[threadIdx.x];
	__syncthreads();
}

} // namespace nrs
