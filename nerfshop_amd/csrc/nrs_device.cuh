// nrs_device.cuh -- per-ray / per-sample device math of the render path for gfx950.
//
// Everything here is plain fp32/u32 arithmetic whose operation ORDER is part of the contract: the file is
// compiled with -ffp-contract=off and the only fused multiply-adds are explicit fmaf(), so that the
// (t, dt, mip, cell) stream of every ray is bit-identical to the CPU oracle.  Citations are file:line in the
// reference checkout ("tn" = src/testbed_nerf.cu, "cn" = src/common_nerf.cu).
#pragma once
#include <hip/hip_runtime.h>
#include "nrs_internal.h"

namespace nrs {

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
// Eigen evaluates fixed-size reductions (dot, squaredNorm, the inner sums of small matrix products) with its complete unroller
// (Eigen/src/Core/Redux.h, redux_novec_unroller: split at len / 2), so a 3-term sum is x0 + (x1 + x2), not (x0 + x1) + x2.
// The reference's own sources compiled against that model pin this (oracle/ref_render.cpp, tests/test_ref_pin.py).
__device__ __forceinline__ float sum3(float a, float b, float c) { return a + (b + c); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return sum3(a.x * b.x, a.y * b.y, a.z * b.z); }
// M * v, M column-major 3x3 (Eigen's coefficient-based product: every row in the reduction order above)
__device__ __forceinline__ f3 mat3_mul(const float* M, f3 v) {
	return {sum3(M[0] * v.x, M[3] * v.y, M[6] * v.z), sum3(M[1] * v.x, M[4] * v.y, M[7] * v.z), sum3(M[2] * v.x, M[5] * v.y, M[8] * v.z)};
}
__device__ __forceinline__ f3 cross3(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

__device__ __forceinline__ bool box_contains(const Box3& b, f3 p) { // BoundingBox::contains, bounding_box.cuh:243
	return p.x >= b.mn[0] && p.x <= b.mx[0] && p.y >= b.mn[1] && p.y <= b.mx[1] && p.z >= b.mn[2] && p.z <= b.mx[2];
}

// ---- constants (common_nerf.h:16-39) ------------------------------------------------------------------------
#define NRS_SQRT3 1.73205080757f
#define NRS_MIN_STEP (NRS_SQRT3 / 1024)
#define NRS_MAX_STEP (NRS_MIN_STEP * (1 << (kCascades - 1)) * 1024 / kGrid)
#define NRS_NEAR_DISTANCE 0.05f

// ---- Morton (tcnn morton3D; x in the lowest bit) --------------------------------------------------------------
// (v * 0x00010001) & mask etc. of tcnn's expand_bits equals (v | v << s) & mask for v < 2^10: written with shifts because
// v_mul_lo_u32 is a quarter-rate instruction and this runs once per DDA step.
__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
	v = (v | (v << 16)) & 0xFF0000FFu;
	v = (v | (v << 8)) & 0x0F00F00Fu;
	v = (v | (v << 4)) & 0xC30C30C3u;
	v = (v | (v << 2)) & 0x49249249u;
	return v;
}
__device__ __forceinline__ uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) {
	return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
__device__ __forceinline__ uint32_t morton3D_invert(uint32_t x) {
	x = x & 0x49249249u;
	x = (x | (x >> 2)) & 0xc30c30c3u;
	x = (x | (x >> 4)) & 0x0f00f00fu;
	x = (x | (x >> 8)) & 0xff0000ffu;
	x = (x | (x >> 16)) & 0x0000ffffu;
	return x;
}

// ---- scrambled Sobol (random_val.cuh:159-288, 317-322); dims 0 and 1 only ---------------------------------------
// dim 0 direction numbers are 0x80000000 >> bit, i.e. sobol(index, 0) = bit reversal of the index;
// dim 1 obeys v[b] = v[b-1] ^ (v[b-1] >> 1).
__device__ __forceinline__ uint32_t sobol_dim0(uint32_t index) { return __brev(index); }
__device__ __forceinline__ uint32_t sobol_dim1(uint32_t index) {
	uint32_t X = 0, v = 0x80000000u;
	#pragma unroll
	for (uint32_t bit = 0; bit < 32; ++bit) {
		X ^= ((index >> bit) & 1u) ? v : 0u;
		v ^= v >> 1;
	}
	return X;
}
__device__ __forceinline__ uint32_t hash_combine(uint32_t seed, uint32_t v) { return seed ^ (v + (seed << 6) + (seed >> 2)); }
__device__ __forceinline__ uint32_t laine_karras_permutation(uint32_t x, uint32_t seed) {
	x += seed;
	x ^= x * 0x6c50b47cu;
	x ^= x * 0xb82f1e52u;
	x ^= x * 0xc7afe638u;
	x ^= x * 0x8d22f6e6u;
	return x;
}
__device__ __forceinline__ uint32_t nested_uniform_scramble_base2(uint32_t x, uint32_t seed) {
	x = __brev(x);
	x = laine_karras_permutation(x, seed);
	x = __brev(x);
	return x;
}
#define NRS_SOBOL_S 2.3283064365386963e-10f /* float(1.0 / 2^32) */
__device__ __forceinline__ float ld_random_val(uint32_t index, uint32_t seed) { // dim 0
	index = nested_uniform_scramble_base2(index, seed);
	return (float)nested_uniform_scramble_base2(sobol_dim0(index), hash_combine(seed, 0u)) * NRS_SOBOL_S;
}
__device__ __forceinline__ void ld_random_val_2d(uint32_t index, uint32_t seed, float& a, float& b) {
	index = nested_uniform_scramble_base2(index, seed);
	a = (float)nested_uniform_scramble_base2(sobol_dim0(index), hash_combine(seed, 0u)) * NRS_SOBOL_S;
	b = (float)nested_uniform_scramble_base2(sobol_dim1(index), hash_combine(seed, 1u)) * NRS_SOBOL_S;
}
__device__ __forceinline__ float fractf_(float x) { return x - floorf(x); }
__device__ __forceinline__ void ld_random_pixel_offset(uint32_t spp, float& ox, float& oy) {
	float a0, a1, b0, b1;
	ld_random_val_2d(0u, 0xdeadbeefu, a0, a1);
	ld_random_val_2d(spp, 0xdeadbeefu, b0, b1);
	ox = fractf_((0.5f - a0) + b0);
	oy = fractf_((0.5f - a1) + b1);
}

// ---- step / grid math (cn:80-177) -------------------------------------------------------------------------------
__device__ __forceinline__ float clampf_(float v, float lo, float hi) { return v < lo ? lo : (hi < v ? hi : v); }
// clamp(t * cone, MIN, MAX) as one v_med3_f32: the median of (v, lo, hi) is the clamp for every non-NaN v (t * cone >= 0)
__device__ __forceinline__ float calc_dt(float t, float cone_angle) { return __builtin_amdgcn_fmed3f(t * cone_angle, NRS_MIN_STEP, NRS_MAX_STEP); }
__device__ __forceinline__ float signf_(float x) { return copysignf(1.0f, x); }

// frexpf's exponent: x = m * 2^e, 0.5 <= |m| < 1; 0 for x == 0
__device__ __forceinline__ int frexp_exponent(float x) {
	uint32_t u = __float_as_uint(x) & 0x7fffffffu;
	if (u == 0) return 0;
	uint32_t ef = u >> 23;
	if (ef == 0) return -117 - __clz((int)u);
	return (int)ef - 126;
}
// res is a power of two (128 >> mip): dividing by it equals multiplying by inv_res = 1/res exactly, bit for bit
__device__ __forceinline__ float distance_to_next_voxel(f3 pos, f3 dir, f3 idir, uint32_t res, float inv_res) {
	f3 p = (float)res * pos;
	float tx = (floorf(p.x + 0.5f + 0.5f * signf_(dir.x)) - p.x) * idir.x;
	float ty = (floorf(p.y + 0.5f + 0.5f * signf_(dir.y)) - p.y) * idir.y;
	float tz = (floorf(p.z + 0.5f + 0.5f * signf_(dir.z)) - p.z) * idir.z;
	float t = fminf(fminf(tx, ty), tz);
	return fmaxf(t * inv_res, 0.0f);
}
__device__ __forceinline__ float advance_to_next_voxel(float t, float cone_angle, f3 pos, f3 dir, f3 idir, uint32_t res, float inv_res) {
	float t_target = t + distance_to_next_voxel(pos, dir, idir, res, inv_res);
	if (cone_angle == 0.f) {
		// constant step (aabb_scale 1): the same chain of float additions as the loop below, but the first eight are
		// straight-line select code -- a cell diagonal is at most sqrt(3)/128 = 8 steps of sqrt(3)/1024, so the divergent
		// loop (whose trip count is the wave's maximum) only runs for coarser cascades.
		// t += (t < t_target) ? dt : 0 without a compare/select pair (VCC hazards): m = clamp((t_target - t) * 2^100, 0, 1)
		// is exactly 1 or 0, and fmaf(m, dt, t) rounds t + dt exactly as the addition does (m * dt is exact).
		t += NRS_MIN_STEP;
		#pragma unroll
		for (int k = 0; k < 7; ++k) {
			const float m = __builtin_amdgcn_fmed3f((t_target - t) * 0x1p100f, 0.0f, 1.0f);
			t = fmaf(m, NRS_MIN_STEP, t);
		}
		while (t < t_target) t += NRS_MIN_STEP;
		return t;
	}
	do {
		t += calc_dt(t, cone_angle);
	} while (t < t_target);
	return t;
}
__device__ __forceinline__ int clampi_(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ uint32_t cascaded_grid_idx_at(f3 pos, uint32_t mip) {
	float mip_scale = ldexpf(1.0f, -(int)mip);
	pos = pos - mk3(0.5f, 0.5f, 0.5f);
	pos = pos * mip_scale;
	pos = pos + mk3(0.5f, 0.5f, 0.5f);
	int ix = (int)(pos.x * (float)kGrid), iy = (int)(pos.y * (float)kGrid), iz = (int)(pos.z * (float)kGrid);
	return morton3D((uint32_t)clampi_(ix, 0, kGrid - 1), (uint32_t)clampi_(iy, 0, kGrid - 1), (uint32_t)clampi_(iz, 0, kGrid - 1));
}
__device__ __forceinline__ bool get_bitfield_at(uint32_t cell_idx, uint32_t level, const uint8_t* __restrict__ bitfield) {
	return bitfield[cell_idx / 8 + (kGridVol * level) / 8] & (1 << (cell_idx % 8));
}
__device__ __forceinline__ bool density_grid_occupied_at(f3 pos, const uint8_t* __restrict__ bitfield, uint32_t mip) {
	return get_bitfield_at(cascaded_grid_idx_at(pos, mip), mip, bitfield);
}
// min(4, max(0, frexp_exponent(maxval) + 1)) without branches: the biased exponent alone decides (sub-normal maxval
// clamps to 0 like every maxval < 0.5), except frexpf(0) = 0 * 2^0, which the reference turns into mip 1.
__device__ __forceinline__ int mip_from_pos(f3 pos) {
	float maxval = fmaxf(fmaxf(fabsf(pos.x - 0.5f), fabsf(pos.y - 0.5f)), fabsf(pos.z - 0.5f));
	const uint32_t u = __float_as_uint(maxval); // maxval >= 0
	const int mip = min((int)kCascades - 1, max(0, (int)(u >> 23) - 125));
	return u == 0 ? 1 : mip;
}
__device__ __forceinline__ int mip_from_dt(float dt, f3 pos) {
	int mip = mip_from_pos(pos);
	dt *= 2 * kGrid;
	// dt >= 1 is a normal number: frexp exponent = biased exponent - 126; for dt < 1 that is <= 0 <= mip, so max() keeps mip
	const int exponent = (int)(__float_as_uint(dt) >> 23) - 126;
	return min((int)kCascades - 1, max(exponent, mip));
}
__device__ __forceinline__ f3 warp_position(f3 pos, const Box3& aabb) {
	return {(pos.x - aabb.mn[0]) / (aabb.mx[0] - aabb.mn[0]), (pos.y - aabb.mn[1]) / (aabb.mx[1] - aabb.mn[1]),
	        (pos.z - aabb.mn[2]) / (aabb.mx[2] - aabb.mn[2])};
}
__device__ __forceinline__ f3 unwarp_position(f3 pos, const Box3& aabb) {
	return {aabb.mn[0] + pos.x * (aabb.mx[0] - aabb.mn[0]), aabb.mn[1] + pos.y * (aabb.mx[1] - aabb.mn[1]),
	        aabb.mn[2] + pos.z * (aabb.mx[2] - aabb.mn[2])};
}
__device__ __forceinline__ f3 warp_direction(f3 d) { return {(d.x + 1.0f) * 0.5f, (d.y + 1.0f) * 0.5f, (d.z + 1.0f) * 0.5f}; }
__device__ __forceinline__ f3 unwarp_direction(f3 d) { return {d.x * 2.0f - 1.0f, d.y * 2.0f - 1.0f, d.z * 2.0f - 1.0f}; }
__device__ __forceinline__ float warp_dt(float dt) {
	float max_stepsize = NRS_MIN_STEP * (1 << (kCascades - 1));
	return (dt - NRS_MIN_STEP) / (max_stepsize - NRS_MIN_STEP);
}
__device__ __forceinline__ float unwarp_dt(float dt) {
	float max_stepsize = NRS_MIN_STEP * (1 << (kCascades - 1));
	return dt * (max_stepsize - NRS_MIN_STEP) + NRS_MIN_STEP;
}

// ---- primary rays: pixel_to_ray (common_device.cuh:245-295) + init_rays_with_payload_kernel_nerf (tn:2512-2616) -------
__device__ __forceinline__ void ray_intersect(const float* mn, const float* mx, f3 pos, f3 dir, float& tmin_out) {
	const float FMAX = 3.402823466e+38f;
	float tmin = (mn[0] - pos.x) / dir.x, tmax = (mx[0] - pos.x) / dir.x;
	if (tmin > tmax) { float s = tmin; tmin = tmax; tmax = s; }
	float tymin = (mn[1] - pos.y) / dir.y, tymax = (mx[1] - pos.y) / dir.y;
	if (tymin > tymax) { float s = tymin; tymin = tymax; tymax = s; }
	if (tmin > tymax || tymin > tmax) { tmin_out = FMAX; return; }
	if (tymin > tmin) tmin = tymin;
	if (tymax < tmax) tmax = tymax;
	float tzmin = (mn[2] - pos.z) / dir.z, tzmax = (mx[2] - pos.z) / dir.z;
	if (tzmin > tzmax) { float s = tzmin; tzmin = tzmax; tzmax = s; }
	if (tmin > tzmax || tzmin > tmax) { tmin_out = FMAX; return; }
	if (tzmin > tmin) tmin = tzmin;
	tmin_out = tmin;
}

struct Ray { f3 o, d; float t; bool alive; };

// square2disk_shirley, random_val.cuh:109-125.  sincosf is the device library's (<= 2 ulp): the thin-lens branch is the one place on the path where
// ray origins are not bit-identical to the host-compiled reference (glibc's sincosf) -- nor is CUDA's; tests/test_gpu_modes.py states the tolerance.
__device__ __forceinline__ void square2disk_shirley(float a, float b, float& ox, float& oy) {
	const float PI = 3.14159265358979323846f;
	float phi, r;
	if (a * a > b * b) { r = a; phi = (PI / 4.0f) * (b / a); }
	else { r = b; phi = (PI / 2.0f) - (PI / 4.0f) * (a / b); }
	float sin_phi, cos_phi;
	sincosf(phi, &sin_phi, &cos_phi);
	ox = r * cos_phi; oy = r * sin_phi;
}

// ---- camera model and background (LENS instantiations only) ------------------------------------------------------------------------------------
// apply_camera_distortion / iterative_camera_undistortion (common_device.cuh:146-200): OpenCV radial + tangential model, undone by Newton iterations with
// a central-difference Jacobian and Eigen's 2x2 inverse (adjugate times 1 / det) -- plain fp32 in the reference's order: bit-identical to the oracle.
__device__ __forceinline__ void apply_camera_distortion(const float* prm, float u, float v, float& du, float& dv) {
	const float k1 = prm[0], k2 = prm[1], p1 = prm[2], p2 = prm[3];
	const float u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2;
	const float radial = k1 * r2 + k2 * r2 * r2;
	du = u * radial + 2.f * p1 * uv + p2 * (r2 + 2.f * u2);
	dv = v * radial + 2.f * p2 * uv + p1 * (r2 + 2.f * v2);
}
__device__ __forceinline__ void iterative_camera_undistortion(const float* prm, float& u, float& v) {
	const float kMaxStepNorm = 1e-10f, kRelStepSize = 1e-6f, eps = 1.1920928955078125e-07f;
	const float x00 = u, x01 = v;
	float x0 = u, x1 = v;
	#pragma unroll 1
	for (uint32_t i = 0; i < 100; ++i) {
		const float step0 = fmaxf(eps, fabsf(kRelStepSize * x0)), step1 = fmaxf(eps, fabsf(kRelStepSize * x1));
		float dx0, dx1, b00, b01, f00, f01, b10, b11, f10, f11;
		apply_camera_distortion(prm, x0, x1, dx0, dx1);
		apply_camera_distortion(prm, x0 - step0, x1, b00, b01);
		apply_camera_distortion(prm, x0 + step0, x1, f00, f01);
		apply_camera_distortion(prm, x0, x1 - step1, b10, b11);
		apply_camera_distortion(prm, x0, x1 + step1, f10, f11);
		const float J00 = 1 + (f00 - b00) / (2 * step0), J01 = (f10 - b10) / (2 * step1), J10 = (f01 - b01) / (2 * step0), J11 = 1 + (f11 - b11) / (2 * step1);
		const float invdet = 1.0f / (J00 * J11 - J10 * J01);
		const float i00 = J11 * invdet, i10 = -J10 * invdet, i01 = -J01 * invdet, i11 = J00 * invdet;
		const float r0 = x0 + dx0 - x00, r1 = x1 + dx1 - x01;
		const float s0 = i00 * r0 + i01 * r1, s1 = i10 * r0 + i11 * r1;
		x0 -= s0; x1 -= s1;
		if (s0 * s0 + s1 * s1 < kMaxStepNorm) break;
	}
	u = x0; v = x1;
}
// read_image<2> (common_device.cuh:80-110): bilinear, texels clamped
__device__ __forceinline__ void read_image2(const float* __restrict__ data, const int32_t* res, float px, float py, float& o0, float& o1) {
	const float fx = px * (float)(res[0] - 1), fy = py * (float)(res[1] - 1);
	const int tx = (int)fx, ty = (int)fy;
	const float wx = fx - (float)tx, wy = fy - (float)ty;
	const int x0 = max(min(tx, res[0] - 1), 0), x1 = max(min(tx + 1, res[0] - 1), 0), y0 = max(min(ty, res[1] - 1), 0), y1 = max(min(ty + 1, res[1] - 1), 0);
	const float2* d = reinterpret_cast<const float2*>(data);
	const float2 a = d[x0 + y0 * res[0]], b = d[x1 + y0 * res[0]], c = d[x0 + y1 * res[0]], e = d[x1 + y1 * res[0]];
	const float w00 = (1 - wx) * (1 - wy), w10 = (wx) * (1 - wy), w01 = (1 - wx) * (wy), w11 = (wx) * (wy);
	o0 = ((w00 * a.x + w10 * b.x) + w01 * c.x) + w11 * e.x;
	o1 = ((w00 * a.y + w10 * b.y) + w01 * c.y) + w11 * e.y;
}
// read_envmap (envmap.cuh:30-63): spherical coordinates of the direction (acosf / atan2f: the device library's, as sincosf in the thin-lens branch --
// tolerance, not bits, against the host-compiled reference), bilinear lookup wrapping in x and clamped in y
__device__ __forceinline__ float4 read_envmap(const float* __restrict__ data, const int32_t* res, f3 dir) {
	const float PI = 3.14159265358979323846f;
	const f3 d = {dir.z, -dir.x, dir.y};
	const float theta = acosf(fminf(fmaxf(d.z, -1.0f), 1.0f));
	const float phi = atan2f(d.y, d.x);
	const float cyl_x = theta / PI, cyl_y = (phi / (2.0f * PI) + 0.5f);
	const float fx = cyl_y * (float)(res[0] - 1), fy = cyl_x * (float)(res[1] - 1);
	const int tx = (int)fx, ty = (int)fy;
	const float wx = fx - (float)tx, wy = fy - (float)ty;
	auto wrapx = [&](int x) { return x < 0 ? x + res[0] : (x >= res[0] ? x - res[0] : x); };
	const int x0 = wrapx(tx), x1 = wrapx(tx + 1), y0 = max(min(ty, res[1] - 1), 0), y1 = max(min(ty + 1, res[1] - 1), 0);
	const float4* q = reinterpret_cast<const float4*>(data);
	const float4 a = q[x0 + y0 * res[0]], b = q[x1 + y0 * res[0]], c = q[x0 + y1 * res[0]], e = q[x1 + y1 * res[0]];
	const float w00 = (1 - wx) * (1 - wy), w10 = (wx) * (1 - wy), w01 = (1 - wx) * (wy), w11 = (wx) * (wy);
	return make_float4(((w00 * a.x + w10 * b.x) + w01 * c.x) + w11 * e.x, ((w00 * a.y + w10 * b.y) + w01 * c.y) + w11 * e.y,
	                   ((w00 * a.z + w10 * b.z) + w01 * c.z) + w11 * e.z, ((w00 * a.w + w10 * b.w) + w01 * c.w) + w11 * e.w);
}

// pixel_to_ray (common_device.cuh:245-295): origin and UN-normalised direction of pixel (x, y) through the camera of its ray time
// (init_rays_with_payload_kernel_nerf, tn:2551-2567).  offset = ld_random_pixel_offset(snap ? 0 : spp), computed once per thread by the caller.
// LENS compiles in the thin-lens branch (:285-293; m_dof, focus distance focus_z = plane_z): only the instantiations that serve dof != 0 carry it.
template <bool LENS = false>
__device__ __forceinline__ void pixel_ray_raw(const nrs_render_params& p, uint32_t x, uint32_t y, float off_x, float off_y, float focus_z, f3& o, f3& d, bool use_dof = true) {
	const float W = (float)p.resolution[0], H = (float)p.resolution[1];
	const uint32_t idx = x + (uint32_t)p.resolution[0] * y;
	float u = ((float)x + 0.5f) * (1.f / W);
	float v = ((float)y + 0.5f) * (1.f / H);
	float ray_time = p.rolling_shutter[0] + p.rolling_shutter[1] * u + p.rolling_shutter[2] * v;
	float rs_rand = (p.rolling_shutter[3] != 0.f) ? ld_random_val(p.spp_index, idx * 72239731u) : 0.f; // x * 0 == 0 for finite x
	ray_time = ray_time + p.rolling_shutter[3] * rs_rand;
	float cam[12];
	#pragma unroll
	for (int i = 0; i < 12; ++i) cam[i] = p.camera_matrix0[i] * ray_time + p.camera_matrix1[i] * (1.f - ray_time);
	float uvx = ((float)x + off_x) / W;
	float uvy = ((float)y + off_y) / H;
	f3 dir = {(uvx - p.screen_center[0]) * W / p.focal_length[0], (uvy - p.screen_center[1]) * H / p.focal_length[1], 1.0f};
	if (LENS && p.distortion_mode == 2u) { // FTheta, common_device.cuh:231-243, :263-267
		const float* prm = p.distortion_params;
		const float xpix = (uvx - p.screen_center[0]) * prm[5], ypix = (uvy - p.screen_center[1]) * prm[6];
		const float norm = sqrtf(xpix * xpix + ypix * ypix);
		const float alpha = prm[0] + norm * (prm[1] + norm * (prm[2] + norm * (prm[3] + norm * prm[4])));
		float sin_alpha, cos_alpha;
		sincosf(alpha, &sin_alpha, &cos_alpha);
		if (cos_alpha <= 1.17549435e-38f || norm == 0.f) { // the error direction: a point outside the aabb so that the pixel is not rendered
			o = mk3(1000.f, 0.f, 0.f); d = mk3(0.f, 0.f, 1.f);
			return;
		}
		sin_alpha *= 1.f / norm;
		dir = {sin_alpha * xpix, sin_alpha * ypix, cos_alpha};
	} else if (LENS && p.distortion_mode == 1u) {
		iterative_camera_undistortion(p.distortion_params, dir.x, dir.y);
	}
	if (LENS && p.d_distortion_map) { // :278-280
		float d0, d1;
		read_image2(p.d_distortion_map, p.distortion_resolution, uvx, uvy, d0, d1);
		dir.x += d0; dir.y += d1;
	}
	d = mat3_mul(cam, dir); // camera_matrix.block<3, 3>(0, 0) * dir, common_device.cuh:282
	o = {cam[9], cam[10], cam[11]};
	if (LENS && use_dof && p.dof != 0.0f) {
		const f3 lookat = o + d * focus_z;
		float r0, r1, bx, by;
		ld_random_val_2d(p.spp_index, x * 19349663u + y * 96925573u, r0, r1);
		square2disk_shirley(r0 * 2.0f - 1.0f, r1 * 2.0f - 1.0f, bx, by);
		bx = p.dof * bx; by = p.dof * by;
		o = {o.x + (cam[0] * bx + cam[3] * by), o.y + (cam[1] * bx + cam[4] * by), o.z + (cam[2] * bx + cam[5] * by)};
		const f3 diff = lookat - o;
		d = {diff.x / focus_z, diff.y / focus_z, diff.z / focus_z};
	}
}
// origin and normalised direction (tn:2588)
template <bool LENS = false>
__device__ __forceinline__ void ray_origin_dir(const nrs_render_params& p, uint32_t x, uint32_t y, float off_x, float off_y, f3& o, f3& d) {
	pixel_ray_raw<LENS>(p, x, y, off_x, off_y, p.slice_plane_z, o, d);
	float n = sqrtf(dot3(d, d));
	d = {d.x / n, d.y / n, d.z / n};
}

template <bool LENS = false>
__device__ __forceinline__ Ray init_ray(const nrs_render_params& p, uint32_t x, uint32_t y, float off_x, float off_y) {
	Ray r;
	ray_origin_dir<LENS>(p, x, y, off_x, off_y, r.o, r.d);
	float tmin;
	ray_intersect(p.render_aabb_min, p.render_aabb_max, r.o, r.d, tmin);
	r.t = fmaxf(tmin, NRS_NEAR_DISTANCE) + 1e-6f;
	Box3 bb;
	#pragma unroll
	for (int i = 0; i < 3; ++i) { bb.mn[i] = p.render_aabb_min[i]; bb.mx[i] = p.render_aabb_max[i]; }
	r.alive = box_contains(bb, r.o + r.d * r.t);
	return r;
}

// Does the ray o + d*s, s >= t, still meet the box?  (slab test; NaNs from 0 * inf drop out of fminf / fmaxf)
__device__ __forceinline__ bool ray_meets_box_ahead(const Box3& b, f3 o, f3 idir, float t) {
	const float ax = (b.mn[0] - o.x) * idir.x, bx = (b.mx[0] - o.x) * idir.x;
	const float ay = (b.mn[1] - o.y) * idir.y, by = (b.mx[1] - o.y) * idir.y;
	const float az = (b.mn[2] - o.z) * idir.z, bz = (b.mx[2] - o.z) * idir.z;
	const float tnear = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), t));
	const float tfar = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
	return tnear <= tfar;
}

// Marching shortcut 2 (result-preserving).  kCoarse^3 blocks over occ_box carry "some occupied cell (any cascade,
// inflated like occ_box) overlaps this block".  coarse_safe_until walks the blocks along o + d*s, s >= t (Amanatides-Woo,
// <= 3 * kCoarse - 2 steps) and returns the parameter up to which the ray provably meets no occupied cell:
//   < 0       nothing ahead at all: no further sample can be emitted, the caller retires the ray at once;
//   otherwise the entry into the first marked block, minus a margin that dwarfs the walk's float error.
// `mask` is the LDS copy of DeviceModel::coarse_mask.
__device__ __forceinline__ float coarse_safe_until(const DeviceModel& m, const uint32_t* __restrict__ mask, f3 o, f3 d, f3 idir, float t) {
	const Box3& b = m.occ.box;
	const float ax = (b.mn[0] - o.x) * idir.x, bx = (b.mx[0] - o.x) * idir.x;
	const float ay = (b.mn[1] - o.y) * idir.y, by = (b.mx[1] - o.y) * idir.y;
	const float az = (b.mn[2] - o.z) * idir.z, bz = (b.mx[2] - o.z) * idir.z;
	const float tnear = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), t));
	const float tfar = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
	if (!(tnear <= tfar)) return -1.f;
	const f3 p = o + d * tnear;
	const float hi = (float)(kCoarse - 1);
	int cx = (int)__builtin_amdgcn_fmed3f((p.x - b.mn[0]) * m.occ.inv_cell[0], 0.f, hi);
	int cy = (int)__builtin_amdgcn_fmed3f((p.y - b.mn[1]) * m.occ.inv_cell[1], 0.f, hi);
	int cz = (int)__builtin_amdgcn_fmed3f((p.z - b.mn[2]) * m.occ.inv_cell[2], 0.f, hi);
	const int sx = d.x >= 0.f ? 1 : -1, sy = d.y >= 0.f ? 1 : -1, sz = d.z >= 0.f ? 1 : -1;
	const float inf = __builtin_huge_valf();
	float tmx = d.x != 0.f ? ((b.mn[0] + (float)(cx + (sx > 0 ? 1 : 0)) * m.occ.cell[0]) - o.x) * idir.x : inf;
	float tmy = d.y != 0.f ? ((b.mn[1] + (float)(cy + (sy > 0 ? 1 : 0)) * m.occ.cell[1]) - o.y) * idir.y : inf;
	float tmz = d.z != 0.f ? ((b.mn[2] + (float)(cz + (sz > 0 ? 1 : 0)) * m.occ.cell[2]) - o.z) * idir.z : inf;
	const float tdx = fabsf(m.occ.cell[0] * idir.x), tdy = fabsf(m.occ.cell[1] * idir.y), tdz = fabsf(m.occ.cell[2] * idir.z);
	float t_enter = tnear; // parameter at which the current block was entered
	#pragma unroll 1
	for (int it = 0; it < (int)(3 * kCoarse); ++it) {
		const uint32_t idx = ((uint32_t)cz * kCoarse + (uint32_t)cy) * kCoarse + (uint32_t)cx;
		if ((mask[idx >> 5] >> (idx & 31)) & 1u) return fmaxf(t_enter - 1e-4f, 0.f);
		const bool step_x = tmx <= tmy && tmx <= tmz, step_y = !step_x && tmy <= tmz, step_z = !step_x && !step_y;
		t_enter = step_x ? tmx : (step_y ? tmy : tmz);
		cx += step_x ? sx : 0; cy += step_y ? sy : 0; cz += step_z ? sz : 0;
		tmx += step_x ? tdx : 0.f; tmy += step_y ? tdy : 0.f; tmz += step_z ? tdz : 0.f;
		if ((uint32_t)(cx | cy | cz) > kCoarse - 1) return -1.f; // left the box through an empty block (negative indices have high bits set)
	}
	return 0.f; // unreachable (a ray crosses at most 3 * kCoarse - 2 blocks); "nothing is known" keeps the plain walk
}

// ---- the lean walk in O(1) (round 5) ----------------------------------------------------------------------------------------------------------------
// With a constant step (cone_angle == 0: dt = MIN_STEP everywhere) every parameter the walk ever stands on is a LATTICE point t_0 (+) dt (+) dt ... -- the same
// chain of float additions whatever the cells are; the cells only decide at which lattice points the walk STOPS (the first lattice point at or behind each cell
// border: advance_to_next_voxel).  Inside one binade [2^e, 2^(e+1)) the rounded addition t (+) dt adds a CONSTANT number q of ulps (dt = (I + f) ulp(t), f != 1/2:
// q = I + (f > 1/2) -- no ties, so no dependence on the parity of t), so the lattice point k steps on is the integer addition bits(t) + k q: no chain needed.
// What the chain-free jump cannot know is at which lattice points the walk stopped on the way -- and over a stretch that is known to be empty (t < t_safe,
// coarse_safe_until) it does not need to, with one exception: the walk must continue from a lattice point at which the reference's walk STOPS, because the border
// parameter t + distance_to_next_voxel(pos(t)) carries the rounding of the stop it is computed from.  So: jump to a lattice point t_k a cell and a half short of
// t_safe, take ONE ordinary step of the walk from there to u = the first lattice point behind the border g of t_k's cell, and accept u only if no lattice point
// lies within +-B of g, where B bounds the difference between the border parameters computed from any two lattice points of one cell (derivation: DESIGN.md 4,
// "lattice jump": |g - exact border| <= (1.2e-7 t + 6e-8) |1/d_axis| + 1.2e-7 t; B is ten times that).  Then every evaluation of that border -- the reference's,
// from its own stop in that cell, included -- selects the same u: u IS the reference's next stop, and the walk goes on from it bit for bit.  If a lattice point
// sits inside the band (about 1 % of the jumps), or anything else is unusual (a tie, a sub-normal, the stretch leaves the unit cube where the cascade could
// change), the function declines and the ordinary lean walk runs.  The (t, dt) streams stay bit-identical to the oracle's cell-by-cell walk (tests/test_gpu_parity.py).
constexpr uint32_t kJumpMinSteps = 14u; // (a cell is 4.6 steps; the jump costs about two cells of lean walk)
__device__ __forceinline__ bool lattice_jump(f3 o, f3 d, f3 idir, float t_safe, uint32_t mip, float& t) {
	// (Straight-line: every condition is folded into one predicate and the candidate is computed whether or not it will be taken -- nested early exits cost this
	// kernel scalar registers for the saved execution masks.  And written with VOP2-encodable literals: a literal operand of a three-source instruction needs a
	// register on gfx9, which the compiler hoists out of the frame loop -- a dozen such constants once pushed the kernel past its 128 registers.)
	const float margin = 0.02f * (float)(1u << mip); // a cell diagonal (1.35e-2 at cascade 0) + two steps + slack
	const float k_want = floorf((t_safe - t - margin) * (1.0f / NRS_MIN_STEP)); // lattice steps to take (a lattice increment differs from dt by < ulp(t) / 2: 1e-5 over a stretch)
	bool ok = k_want >= (float)kJumpMinSteps && k_want < 60000.0f; // (false for a NaN)
	float k_left = ok ? k_want : 0.0f;
	float tc = t;
	// Two binade segments with three ordinary additions behind each: a stretch of the lego-like scenes starts below t = 1 and ends above it, and a jump that stopped
	// at the binade's end would leave the rest to the cell-by-cell walk of the slowest lane.  (The additions are lattice steps like any other.)
	#pragma unroll 1
	for (int seg = 0; seg < 2; ++seg) {
		const uint32_t b = __float_as_uint(tc);
		const int e_t = (int)(b >> 23); // tc > 0: the sign bit is clear
		// dt in ulps of this binade: x = dt 2^(150 - e_t), exact (a power of two); the lattice increment is x rounded to an integer, a constant as long as x is not
		// half-way between two (round-to-even would then look at the parity of t): such a binade -- or one out of range -- is not jumped through
		const float x = ldexpf(NRS_MIN_STEP, 150 - e_t), qf = rintf(x);
		const bool seg_ok = e_t >= 24 && e_t <= 137 && fabsf(x - qf) != 0.5f && qf >= 1.0f && qf < 8388608.0f; // (e_t <= 137: ulp(t) <= 2^-9 < dt)
		const float left = (float)(0x7fffffu - (b & 0x7fffffu)); // ulps to the binade's last value
		const float k_bin = fmaxf(floorf(left * __builtin_amdgcn_rcpf(qf)) - 1.0f, 0.0f); // lattice points left in the binade (rounded down with room)
		const float k = seg_ok ? fminf(k_left, k_bin) : 0.0f;
		tc = __uint_as_float(b + (uint32_t)k * (uint32_t)qf);
		k_left -= k;
		#pragma unroll
		for (int i = 0; i < 3; ++i) {
			const bool step = k_left > 0.0f;
			tc = step ? tc + NRS_MIN_STEP : tc;
			k_left = step ? k_left - 1.0f : k_left;
		}
	}
	ok = ok && (k_want - k_left) >= (float)kJumpMinSteps + 8.0f;
	// the cascade must be the same at every stop on the way: both ends strictly inside the unit cube, where mip_from_pos is 0 (cone_angle == 0: mip_from_dt adds nothing)
	ok = ok && mip_from_pos(o + d * t) == 0 && mip_from_pos(o + d * tc) == 0;
	__builtin_amdgcn_sched_barrier(0); // (stage by stage: interleaved, the stages' temporaries add up)
	// Up to three candidates: the lattice point reached, then the ones 4 and 8 steps before it (integer subtraction inside the binade: the same lattice).  A candidate
	// that fails the band test below would fail it again on every later attempt from this walk -- the attempts all aim at the same lattice point -- and ONE lane per
	// packet walking its stretch cell by cell is what the whole wave then waits for (measured: 36 instead of 9 wave trips per packet).
	const uint32_t bk = __float_as_uint(tc);
	const float q_here = rintf(ldexpf(NRS_MIN_STEP, 150 - (int)(bk >> 23)));
	const uint32_t back = 4u * (uint32_t)q_here;
	const uint32_t res = kGrid >> mip;
	const float inv_res = ldexpf(1.0f, (int)mip - 7);
	const float B = (fmaxf(fmaxf(fabsf(idir.x), fabsf(idir.y)), fabsf(idir.z)) + 1.0f) * 2e-6f * fmaxf(t_safe, 1.0f);
	bool taken = false;
	float t_new = t;
	#pragma unroll 1
	for (uint32_t attempt = 0; attempt < 3u; ++attempt) {
		//  a back-off candidate needs one full increment of room above the binade's start: the first lattice point of a binade can sit q ulps above 2^e,
		// and bk - 4q could then land ON 2^e, which is not on the lattice
		const bool in_binade = attempt == 0u || (bk & 0x7fffffu) >= attempt * back + (uint32_t)q_here;
		const float tk = __uint_as_float(bk - (in_binade ? attempt * back : 0u));
		const f3 pk = o + d * tk;
		const float g = tk + distance_to_next_voxel(pk, d, idir, res, inv_res);
		// one ordinary step of the walk from t_k (advance_to_next_voxel's constant-step chain; the candidate's own while-loop tail cannot run: a cell of the unit cube's
		// cascade is at most 8 steps, and for coarser cells -- min_mip > 0 -- the predicate below declines when the eight steps did not reach the border)
		float u = tk + NRS_MIN_STEP;
		#pragma unroll
		for (int i = 0; i < 7; ++i) {
			const float mstep = __builtin_amdgcn_fmed3f((g - u) * 0x1p100f, 0.0f, 1.0f);
			u = fmaf(mstep, NRS_MIN_STEP, u);
		}
		bool pass = ok && in_binade && u >= g;
		pass = pass && (g + NRS_MIN_STEP + B < t_safe);             // the lean walk's own hand-over test at the stop in t_k's cell, band included
		pass = pass && (u - g > B) && (g - (u - NRS_MIN_STEP) > B); // no lattice point inside the band (also false for a non-finite B: axis-parallel rays)
		t_new = (pass && !taken) ? u : t_new;
		taken = taken || pass;
		if (!__any(ok && !taken)) break; // (wave-uniform: every lane that can jump has its stop)
	}
	t = t_new;
	return taken;
}

// The inner loop shared by advance_pos_nerf (tn:589-603) and generate_next_nerf_network_inputs (tn:668-692):
// advance t until the ray sits in an occupied cell (returns true; pos/dt valid) or leaves the render box (false).
//
// Shortcut 1: occ_box bounds all occupied cells of all cascades (inflated).  Outside it every occupancy test of the
// reference is false, so (i) the bitfield lookup is skipped -- the DDA recurrence that decides WHICH t values get tested is
// still executed verbatim, so the first accepted t is bit-identical -- and (ii) once the remaining ray cannot meet the box
// any more, no further sample can ever be emitted and the walk to the far side of the render box is cut short.
//
// Shortcut 2: when the cell the ray stands in is empty, the coarse mask is consulted ONCE (coarse_safe_until).  Rays with
// nothing ahead are retired.  For the others the stretch up to t_safe is walked by the LEAN loop below: the same float
// operations on t in the same order (position, cascade, distance to the voxel border, the chain of step additions) but
// none of the tests whose outcome is known there -- inside the render box (t_safe is clipped to it), cell empty.  The
// lean loop hands over BEFORE a step could land beyond t_safe; the full loop then re-enters at a t it would have reached
// itself, so the (t, dt) stream stays bit-identical (tests/test_gpu_parity.py compares it against the oracle).
// The look-ahead of shortcut 2 is repeated every kLookEvery empty cells of a walk, not only at its first: a ray that leaves
// the object through blocks that are marked (they touch the surface) but empty along its path used to walk on, cell by cell
// with a full test each, until it left the occupied bounds; a few cells on, nothing is marked ahead any more and it can be
// retired at once (or lean-walked to the next marked block).  Measured (Gsamples/s, lego + cage edit; ms for a 64x40-pixel
// frame): once per walk 9.0 / 0.84, every 3rd cell 9.43 / 0.58, 6th 9.92 / 0.56, 10th 9.80 / 0.55, 16th 9.82 / 0.58, 32nd 9.54 / 0.65.
constexpr float kJumpSpan = 0.06f; // remaining stretch from which the lattice jump is attempted (22 steps + the margin: nothing to gain below; 0.04 / 0.10 measured the same)
constexpr int kLookEvery = 6;
// The Morton code of an occupancy cell from a 128-entry table in LDS (spread3(v) = the bits of v at every third position; staged
// behind the look-ahead mask by stage_march_lds) -- three ds_read + two v_lshl_or instead of 28 VALU instructions per occupancy test; the cell index
// and therefore every decision stay the same.
constexpr uint32_t kMarchLdsWords = kCoarseWords + kGrid; // look-ahead mask | spread3 table
__device__ __forceinline__ void stage_march_lds(uint32_t* __restrict__ lds, const uint32_t* __restrict__ mask) { // (the caller's barrier publishes it)
	for (uint32_t i = threadIdx.x; i < kCoarseWords; i += blockDim.x) lds[i] = mask[i];
	for (uint32_t i = threadIdx.x; i < kGrid; i += blockDim.x) lds[kCoarseWords + i] = expand_bits(i);
}
// The voxel walk keeps the 64-bit word of the bitfield it last read -- in Morton order that is one 4 x 4 x 4 block of cells of one
// cascade -- and a test that falls into the same block is answered from the registers.  A ray that walks through empty cells next to the surface (the
// silhouette rays that bound small launches: DESIGN 5) then pays the bitfield's load latency once per block instead of once per cell; when no lane of
// the wave needs a new word the load is skipped altogether.  Pure caching: the decisions are the bitfield's.
struct OccWord { uint32_t tag; uint32_t lo, hi; };
// bit number of the cell of cascade `mip` that holds pos, in the concatenated bitfield (cn:117-141 through the LDS spread table)
__device__ __forceinline__ uint32_t occupancy_bit_index(f3 pos, uint32_t mip, const uint32_t* __restrict__ march_lds) {
	const float mip_scale = ldexpf(1.0f, -(int)mip);
	f3 q = pos - mk3(0.5f, 0.5f, 0.5f);
	q = q * mip_scale;
	q = q + mk3(0.5f, 0.5f, 0.5f);
	const int ix = (int)(q.x * (float)kGrid), iy = (int)(q.y * (float)kGrid), iz = (int)(q.z * (float)kGrid);
	const uint32_t* spread = march_lds + kCoarseWords;
	return (spread[clampi_(ix, 0, kGrid - 1)] | (spread[clampi_(iy, 0, kGrid - 1)] << 1) | (spread[clampi_(iz, 0, kGrid - 1)] << 2)) + mip * kGridVol;
}
__device__ __forceinline__ bool occupied_at_cached(f3 pos, const uint8_t* __restrict__ bitfield, uint32_t mip, const uint32_t* __restrict__ march_lds, OccWord& w) {
	const uint32_t idx = occupancy_bit_index(pos, mip, march_lds);
	const uint32_t tag = idx >> 6;
	if (tag != w.tag) {
		const uint2 v = reinterpret_cast<const uint2*>(bitfield)[tag];
		w.tag = tag; w.lo = v.x; w.hi = v.y;
	}
	const uint32_t bit = idx & 63u;
	return (((bit & 32u) ? w.hi : w.lo) >> (bit & 31u)) & 1u;
}
__device__ __forceinline__ bool occupied_at(f3 pos, const uint8_t* __restrict__ bitfield, uint32_t mip, const uint32_t* __restrict__ march_lds) {
	const float mip_scale = ldexpf(1.0f, -(int)mip);
	f3 q = pos - mk3(0.5f, 0.5f, 0.5f);
	q = q * mip_scale;
	q = q + mk3(0.5f, 0.5f, 0.5f);
	const int ix = (int)(q.x * (float)kGrid), iy = (int)(q.y * (float)kGrid), iz = (int)(q.z * (float)kGrid);
	const uint32_t* spread = march_lds + kCoarseWords;
	const uint32_t idx = spread[clampi_(ix, 0, kGrid - 1)] | (spread[clampi_(iy, 0, kGrid - 1)] << 1) | (spread[clampi_(iz, 0, kGrid - 1)] << 2);
	return get_bitfield_at(idx, mip, bitfield);
}
// 1 / d (three IEEE divisions, ~40 issue slots) is only needed once the ray stands in an EMPTY cell; the common call -- the next
// sample of a ray inside the object -- finds an occupied cell at once.  The first test is peeled off in front of the loop, the reciprocal formed behind it.
// JUMP: compile lattice_jump into this instance (the two hot ones: the fill's first_hit and the per-round walk of one-lane rounds; the team walks keep the plain lean
// walk -- every inlined copy costs scalar registers in a kernel that spills them)
template <bool JUMP = false>
__device__ __forceinline__ bool march_to_occupied(const nrs_render_params& p, const DeviceModel& m, const uint32_t* __restrict__ march_lds, f3 o, f3 d,
                                                  float& t, f3& pos, float& dt, uint32_t* n_iter = nullptr) {
	const uint32_t* __restrict__ coarse_mask = march_lds;
	const uint8_t* __restrict__ bitfield = m.bitfield;
	const Box3& occ_box = m.occ.box;
	Box3 bb;
	#pragma unroll
	for (int i = 0; i < 3; ++i) { bb.mn[i] = p.render_aabb_min[i]; bb.mx[i] = p.render_aabb_max[i]; }
	const float cone = p.cone_angle_constant;
	int until_look = 0; // trips until the next look-ahead
	uint32_t mip;
	bool in_occ;
	OccWord occ_word{0xffffffffu, 0u, 0u};
	#define NRS_OCCUPIED(pos_, mip_) occupied_at_cached(pos_, bitfield, mip_, march_lds, occ_word)
	if (n_iter) ++*n_iter; // profiling build only
	pos = o + d * t;
	if (!box_contains(bb, pos)) return false;
	dt = calc_dt(t, cone);
	mip = max(p.min_mip, (uint32_t)mip_from_dt(dt, pos));
	in_occ = box_contains(occ_box, pos);
	if (in_occ && NRS_OCCUPIED(pos, mip)) return true;
	const f3 idir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
	bool first = true;
	while (1) {
		if (!first) {
			if (n_iter) ++*n_iter; // profiling build only
			pos = o + d * t;
			if (!box_contains(bb, pos)) return false;
			dt = calc_dt(t, cone);
			mip = max(p.min_mip, (uint32_t)mip_from_dt(dt, pos));
			in_occ = box_contains(occ_box, pos);
			if (in_occ && NRS_OCCUPIED(pos, mip)) return true;
		}
		first = false;
		if (!in_occ && !ray_meets_box_ahead(occ_box, o, idir, t)) return false;
		if (until_look-- == 0) {
			until_look = kLookEvery;
			float t_safe = coarse_safe_until(m, coarse_mask, o, d, idir, t);
			if (t_safe < 0.f) return false;
			{ // clip to the exit of the render box: the lean loop carries no containment test
				const float ax = (bb.mn[0] - o.x) * idir.x, bx = (bb.mx[0] - o.x) * idir.x;
				const float ay = (bb.mn[1] - o.y) * idir.y, by = (bb.mx[1] - o.y) * idir.y;
				const float az = (bb.mn[2] - o.z) * idir.z, bz = (bb.mx[2] - o.z) * idir.z;
				t_safe = fminf(t_safe, fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)) - 1e-4f);
			}
			// lean walk: every position it stands on has parameter < t_safe
			while (1) {
				// (tried in EVERY pass of the lean walk, not once in front of it: a lane that declines -- a lattice point in the guard band, the end of a binade, the
				// entry point on the cube's face -- takes one ordinary step and tries again from there; a wave is as slow as its slowest lane, and one lane in a
				// hundred walking the whole stretch cell by cell would keep most waves waiting)
				if (JUMP && cone == 0.f && t_safe - t > kJumpSpan && lattice_jump(o, d, idir, t_safe, mip, t)) {
					if (n_iter) ++*n_iter;
					pos = o + d * t;
					dt = calc_dt(t, cone);
					mip = max(p.min_mip, (uint32_t)mip_from_dt(dt, pos));
				}
				const uint32_t lres = kGrid >> mip;
				const float linv = ldexpf(1.0f, (int)mip - 7);
				// one step past the border is the farthest the next position can be (first lattice point >= the border)
				const float border = t + distance_to_next_voxel(pos, d, idir, lres, linv);
				if (!(border + calc_dt(border, cone) < t_safe)) break;
				t = advance_to_next_voxel(t, cone, pos, d, idir, lres, linv);
				if (n_iter) ++*n_iter;
				pos = o + d * t;
				dt = calc_dt(t, cone);
				mip = max(p.min_mip, (uint32_t)mip_from_dt(dt, pos));
			}
		}
		uint32_t res = kGrid >> mip;
		t = advance_to_next_voxel(t, cone, pos, d, idir, res, ldexpf(1.0f, (int)mip - 7));
	}
	#undef NRS_OCCUPIED
}

// march_to_occupied's first test on its own: true when the walk from parameter t would return at once (inside the render box, in an occupied cell)
__device__ __forceinline__ bool stands_in_occupied_cell(const nrs_render_params& p, const DeviceModel& m, const uint32_t* __restrict__ march_lds, f3 o, f3 d, float t) {
	Box3 bb;
	#pragma unroll
	for (int i = 0; i < 3; ++i) { bb.mn[i] = p.render_aabb_min[i]; bb.mx[i] = p.render_aabb_max[i]; }
	const f3 pos = o + d * t;
	if (!box_contains(bb, pos)) return false;
	const float dt = calc_dt(t, p.cone_angle_constant);
	const uint32_t mip = max(p.min_mip, (uint32_t)mip_from_dt(dt, pos));
	if (!box_contains(m.occ.box, pos)) return false;
	return occupied_at(pos, m.bitfield, mip, march_lds);
}

// advance_pos_nerf, tn:557-606: jitter by one Sobol value, then skip to the first occupied cell
__device__ __forceinline__ bool first_hit(const nrs_render_params& p, const DeviceModel& m, const uint32_t* __restrict__ march_lds, uint32_t pixel_idx, Ray& r,
                                          uint32_t* n_iter = nullptr) {
	float dt = calc_dt(r.t, p.cone_angle_constant);
	r.t += ld_random_val(p.spp_index, pixel_idx * 786433u) * dt;
	f3 pos;
	return march_to_occupied<true>(p, m, march_lds, r.o, r.d, r.t, pos, dt, n_iter);
}

// ---- tet warp: selection_utils.h:10-47, cage_deformation.cu:136-269 -----------------------------------------------
__device__ __forceinline__ float scalar_tp(f3 a, f3 b, f3 c) { return dot3(a, cross3(b, c)); }
__device__ __forceinline__ bool same_side_tet(f3 v1, f3 v2, f3 v3, f3 v4, f3 p) {
	f3 normal = cross3(v2 - v1, v3 - v1);
	float dotV4 = dot3(normal, v4 - v1);
	float dotP = dot3(normal, p - v1);
	return signbit(dotV4) == signbit(dotP);
}
__device__ __forceinline__ bool point_in_tet(f3 v1, f3 v2, f3 v3, f3 v4, f3 p) {
	return same_side_tet(v1, v2, v3, v4, p) && same_side_tet(v2, v3, v4, v1, p) && same_side_tet(v3, v4, v1, v2, p) &&
	       same_side_tet(v4, v1, v2, v3, p);
}
__device__ __forceinline__ void bary_tet(f3 a, f3 b, f3 c, f3 d, f3 p, float out[4]) {
	f3 vap = p - a, vbp = p - b, vab = b - a, vac = c - a, vad = d - a, vbc = c - b, vbd = d - b;
	float va6 = scalar_tp(vbp, vbd, vbc);
	float vb6 = scalar_tp(vap, vac, vad);
	float vc6 = scalar_tp(vap, vad, vab);
	float vd6 = scalar_tp(vap, vab, vac);
	float v6 = (float)(1. / (double)scalar_tp(vab, vac, vad)); // the reference divides in double ("1. / float")
	out[0] = va6 * v6; out[1] = vb6 * v6; out[2] = vc6 * v6; out[3] = vd6 * v6;
}
// The operator tables are reached through pointers that sit in a device-memory struct (DeviceEdit): left alone, the compiler cannot tell their address
// space and emits FLAT loads (an aperture check per access, both wait counters) with one 64-bit address computation per dword.  They are global memory:
// gp() casts them so (global_load), and a vertex / matrix column is one 12-byte load.
#define NRS_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const NRS_GLOBAL T* gp(const T* p) { return (const NRS_GLOBAL T*)p; }
__device__ __forceinline__ f3 ld3(const float* __restrict__ a, uint32_t i) {
	typedef float f3a __attribute__((ext_vector_type(3), aligned(4)));
	const f3a v = *(const NRS_GLOBAL f3a*)(gp(a) + 3 * (size_t)i);
	return {v.x, v.y, v.z};
}

// point_in_tet with the per-tet part of same_side_tet hoisted out of the sample loop, and the tet's own vertices stored next to
// it: tet_planes_kernel writes one 128-byte record per tet -- vertices v_0..v_3 (12 floats), normal_f = cross(v_{f+1} - v_f,
// v_{f+2} - v_f) for f = 0..3 (12 floats), the four bits signbit(dot(normal_f, v_{f+3} - v_f)) -- the very floats same_side_tet
// computes, so dot(normal_f, p - v_f) and the sign comparison are bit-identical to the direct evaluation.  One cache line per
// candidate instead of the tets[] -> vertices[] chain, and a quarter of the arithmetic.
__device__ __forceinline__ bool point_in_tet_rec(const float* __restrict__ recs, uint32_t t, f3 p) {
	typedef float f4n __attribute__((ext_vector_type(4))); // (a native vector: HIP's float4 is a class whose copy constructor wants a generic reference)
	const NRS_GLOBAL f4n* q = gp(reinterpret_cast<const f4n*>(recs)) + 8 * (size_t)t;
	const f4n v0 = q[0], v1 = q[1], v2 = q[2], n0 = q[3], n1 = q[4], n2 = q[5], sg = q[6];
	const float d0 = dot3(mk3(n0.x, n0.y, n0.z), p - mk3(v0.x, v0.y, v0.z));
	const float d1 = dot3(mk3(n0.w, n1.x, n1.y), p - mk3(v0.w, v1.x, v1.y));
	const float d2 = dot3(mk3(n1.z, n1.w, n2.x), p - mk3(v1.z, v1.w, v2.x));
	const float d3 = dot3(mk3(n2.y, n2.z, n2.w), p - mk3(v2.y, v2.z, v2.w));
	const uint32_t got = (__float_as_uint(d0) >> 31) | ((__float_as_uint(d1) >> 31) << 1) | ((__float_as_uint(d2) >> 31) << 2) | ((__float_as_uint(d3) >> 31) << 3);
	return got == __float_as_uint(sg.x);
}
// first tet of the cell's list that contains p (0xffffffff: none); the next candidate's id is fetched while the current one is tested
__device__ __forceinline__ uint32_t scan_list_for_tet(const DeviceEdit& e, const uint32_t* __restrict__ off, const uint32_t* __restrict__ idx, uint32_t cell, f3 p, uint32_t* n_tested = nullptr) {
	const NRS_GLOBAL uint32_t* lut_off = gp(off);
	const NRS_GLOBAL uint32_t* lut_idx = gp(idx);
	const uint32_t j0 = lut_off[cell], j1 = lut_off[cell + 1];
	uint32_t found = 0xffffffffu;
	if (j0 < j1) {
		uint32_t t = lut_idx[j0];
		#pragma unroll 1
		for (uint32_t j = j0; j < j1; ++j) {
			const uint32_t t_next = lut_idx[min(j + 1, j1 - 1)];
			if (n_tested) ++*n_tested; // (profiling instantiation only)
			if (point_in_tet_rec(e.planes, t, p)) { found = t; break; }
			t = t_next;
		}
	}
	return found;
}
__device__ __forceinline__ uint32_t scan_cell_for_tet(const DeviceEdit& e, uint32_t cell, f3 p, uint32_t* n_tested = nullptr) { return scan_list_for_tet(e, e.lut_off, e.lut_idx, cell, p, n_tested); }
// The tet of e's deformed mesh that contains position u (un-warped), as the reference's scan of u's LUT cell finds it (0xffffffff: none).  With a fine look-up table
// (DeviceEdit::fine_off) the candidates are the fine cell's -- the same first hit by construction (nrs_cage.hip: fine_lists_kernel), after fewer tests: on the bench's cage
// the scan of a wave drops from 7 dependent trips per round to 2-3.  The fine coordinates are the LUT cell's arithmetic with a finer multiplier (cn:117-141), so fine >> shift
// IS the LUT cell; a position outside the window stands in a LUT cell whose list is empty.
__device__ __forceinline__ uint32_t find_tet(const DeviceEdit& e, f3 u, const uint32_t* __restrict__ march_lds, uint32_t* n_tested = nullptr) {
	const int level = mip_from_pos(u);
	if (e.fine_off) {
		typedef int32_t i4n __attribute__((ext_vector_type(4)));
		const NRS_GLOBAL i4n* win = gp(reinterpret_cast<const i4n*>(&e.fine_win[0][0])) + 2 * level;
		const i4n lo = win[0], ext = win[1];
		const float mip_scale = ldexpf(1.0f, -level);
		f3 q = u - mk3(0.5f, 0.5f, 0.5f);
		q = q * mip_scale;
		q = q + mk3(0.5f, 0.5f, 0.5f);
		if (ext.w != kFinePlain) { // (else: this cascade keeps the LUT's own lists -- below)
			const int fres_i = (int)(kGrid << ext.w);
			const float fres = (float)fres_i;
			const int hi = fres_i - 1;
			const int fx = clampi_((int)(q.x * fres), 0, hi) - lo.x, fy = clampi_((int)(q.y * fres), 0, hi) - lo.y, fz = clampi_((int)(q.z * fres), 0, hi) - lo.z;
			if ((uint32_t)fx >= (uint32_t)ext.x || (uint32_t)fy >= (uint32_t)ext.y || (uint32_t)fz >= (uint32_t)ext.z) return 0xffffffffu;
			const uint32_t cell = (uint32_t)lo.w + ((uint32_t)fz * (uint32_t)ext.y + (uint32_t)fy) * (uint32_t)ext.x + (uint32_t)fx;
			return scan_list_for_tet(e, e.fine_off, e.fine_idx, cell, u, n_tested);
		}
	}
	const uint32_t cell = march_lds ? occupancy_bit_index(u, (uint32_t)level, march_lds) : (uint32_t)level * kGridVol + cascaded_grid_idx_at(u, (uint32_t)level);
	return scan_cell_for_tet(e, cell, u, n_tested);
}

// interpolate_tet (with_dir, honours copy) / interpolate_tet_pos (!with_dir, ignores copy).  pos/dir are the warped
// [0,1] values of the NerfCoordinate; returns true if the sample must be treated as empty space.
// Two phases on purpose: the LUT scan only decides WHICH tet contains the sample; the barycentric map-back reloads that
// tet's vertices afterwards.  Fusing them keeps ~48 more VGPRs live across the scan and costs the kernel a wave of occupancy.
// march_lds (optional): the kernel's LDS copy of the Morton spread table (stage_march_lds) -- the cell index then costs three LDS reads instead of 28 VALU
// instructions (the same index: occupancy_bit_index is cascaded_grid_idx_at through the table)
// scan_out (optional): what the tet search found (a tet number or 0xffffffff), kTetNotSearched when the sample is outside the deformed mesh's box -- the membrane
// correction of the same operator looks for the same tet at the same position (poisson_residual_find) and takes it from here.
constexpr uint32_t kTetNotSearched = 0xfffffffeu;
__device__ __forceinline__ bool tet_warp(const DeviceEdit& e, bool with_dir, f3& wpos, f3& wdir, const uint32_t* __restrict__ march_lds = nullptr, uint32_t* scan_out = nullptr,
                                         uint32_t* n_tested = nullptr) {
	bool in_deformed = false;
	if (scan_out) *scan_out = kTetNotSearched;
	if (box_contains(e.warped_bbox, wpos)) {
		const f3 u = unwarp_position(wpos, e.aabb);
		if (n_tested) *n_tested |= 0x10000u; // (profiling: the sample stands inside the deformed mesh's box; low half = candidates tested)
		const uint32_t found = find_tet(e, u, march_lds, n_tested);
		if (n_tested && found != 0xffffffffu) *n_tested |= 0x20000u;
		if (scan_out) *scan_out = found;
		__builtin_amdgcn_sched_barrier(0);
		if (found != 0xffffffffu) {
			typedef uint32_t u4n __attribute__((ext_vector_type(4)));
			const u4n tv = gp(reinterpret_cast<const u4n*>(e.tets))[found];
			float bc[4];
			{
				const f3 a = ld3(e.verts, tv.x), b = ld3(e.verts, tv.y), c = ld3(e.verts, tv.z), d = ld3(e.verts, tv.w);
				bary_tet(a, b, c, d, u, bc);
			}
			__builtin_amdgcn_sched_barrier(0);
			f3 canon = bc[0] * ld3(e.orig, tv.x) + bc[1] * ld3(e.orig, tv.y);
			canon = canon + bc[2] * ld3(e.orig, tv.z);
			canon = canon + bc[3] * ld3(e.orig, tv.w);
			wpos = e.diag_pow2 ? mk3((canon.x - e.aabb.mn[0]) * e.inv_diag[0], (canon.y - e.aabb.mn[1]) * e.inv_diag[1], (canon.z - e.aabb.mn[2]) * e.inv_diag[2])
			                   : warp_position(canon, e.aabb);
			__builtin_amdgcn_sched_barrier(0);
			if (with_dir && e.rot) {
				const f3 ud = unwarp_direction(wdir);
				const f3 c0 = ld3(e.rot, 3u * found), c1 = ld3(e.rot, 3u * found + 1u), c2 = ld3(e.rot, 3u * found + 2u); // the three columns
				const float R[9] = {c0.x, c0.y, c0.z, c1.x, c1.y, c1.z, c2.x, c2.y, c2.z};
				const f3 rd = mat3_mul(R, ud);
				wdir = warp_direction(rd);
			}
			in_deformed = true;
		}
	}
	bool empty = false;
	if (!(with_dir && e.copy)) {
		if (!in_deformed && box_contains(e.orig_warped_bbox, wpos)) {
			const f3 u = unwarp_position(wpos, e.aabb);
			const int level = mip_from_pos(u);
			const uint32_t pos_idx = cascaded_grid_idx_at(u, (uint32_t)level);
			empty = gp(e.orig_bitfield)[pos_idx / 8 + (kGridVol * (uint32_t)level) / 8] & (1 << (pos_idx % 8)); // get_bitfield_at
		}
	}
	return empty;
}

// AffineDuplication::map_rays / map_positions (affine_duplication.cu:69-118): samples inside the warped destination box
// are carried back into the selection box (inverse scale, inverse rotation about the destination centre, inverse
// translation; optionally the view direction too); samples inside the selection box are emptied if hide_original.
__device__ __forceinline__ bool affine_contains(const AffineBox& b, f3 p) {
	const f3 q = p - mk3(b.mn[0], b.mn[1], b.mn[2]);
	const float du = dot3(mk3(b.u[0], b.u[1], b.u[2]), q), dv = dot3(mk3(b.v[0], b.v[1], b.v[2]), q), dw = dot3(mk3(b.w[0], b.w[1], b.w[2]), q);
	return du >= 0.f && du < b.uu && dv >= 0.f && dv < b.vv && dw >= 0.f && dw < b.ww;
}
__device__ __forceinline__ f3 mul_rt(const float* R, f3 q) { // R^T q, R column-major: (R^T q)_i = sum_k R(k, i) q_k
	return {sum3(R[0] * q.x, R[1] * q.y, R[2] * q.z), sum3(R[3] * q.x, R[4] * q.y, R[5] * q.z), sum3(R[6] * q.x, R[7] * q.y, R[8] * q.z)};
}
__device__ __forceinline__ bool affine_warp(const DeviceEdit& e, bool with_dir, f3& wpos, f3& wdir) {
	if (affine_contains(e.a_dst, wpos)) {
		const f3 c = mk3(e.a_dst.center[0], e.a_dst.center[1], e.a_dst.center[2]);
		const f3 d = wpos - c;
		const f3 q = {d.x / e.a_scale[0], d.y / e.a_scale[1], d.z / e.a_scale[2]};
		f3 p = mul_rt(e.a_rot, q) + c;
		wpos = p - mk3(e.a_translation[0], e.a_translation[1], e.a_translation[2]);
		if (with_dir && e.a_correct_dir) wdir = warp_direction(mul_rt(e.a_rot, unwarp_direction(wdir)));
		return false;
	}
	return e.a_hide_original && affine_contains(e.a_sel, wpos);
}
// EditOperator::map_rays / map_positions dispatch
__device__ __forceinline__ bool edit_warp(const DeviceEdit& e, bool with_dir, f3& wpos, f3& wdir) {
	if (e.kind == kEditAffine) return affine_warp(e, with_dir, wpos, wdir);
	return tet_warp(e, with_dir, wpos, wdir);
}

// Membrane ("Poisson") correction inputs of one sample: compute_residual_poisson_kernel's body (cage_deformation.cu:467-507)
// fused with the evaluate_sh9 (cn:218-245) that composite_kernel_nerf applies to its result (tn:800-805).  wpos0 is the
// sample position BEFORE map_rays (residuals live in deformed space), dir the un-warped view direction AFTER map_rays.
// The barycentric interpolation of the 27 SH9RGB coefficients and the dot product with the SH basis are evaluated in the
// reference's order, coefficient by coefficient, so no 27-float array is kept.  Outputs untouched when no tet contains it.
// Two steps (round 4), so that the renderer can run the un-deformed network pass between them with only three values live: _find decides which tet of the
// deformed mesh holds the sample and interpolates the two densities; _colour re-derives the barycentric weights of that tet (the same arithmetic: the same
// bits) and evaluates the SH9 colour.
// searched (optional): the result of tet_warp's search of THIS operator's mesh at THIS position (the first operator the sample met: its position was still wpos0), or
// kTetNotSearched -- the same function of the same arguments, so it is taken instead of searched for again.
__device__ __forceinline__ bool poisson_residual_find(const DeviceEdit& e, f3 wpos0, uint32_t& found_out, float& out_density, float& res_density,
                                                      const uint32_t* __restrict__ march_lds = nullptr, uint32_t searched = kTetNotSearched) {
	const f3 pos = unwarp_position(wpos0, e.aabb);
	if (!box_contains(e.bbox, pos)) return false;
	uint32_t found = searched;
	if (searched == kTetNotSearched) {
		found = find_tet(e, pos, march_lds);
	}
	if (found == 0xffffffffu) return false;
	typedef uint32_t u4n __attribute__((ext_vector_type(4)));
	const u4n tv = gp(reinterpret_cast<const u4n*>(e.tets))[found];
	float bc[4];
	{
		const f3 a = ld3(e.verts, tv.x), b = ld3(e.verts, tv.y), c = ld3(e.verts, tv.z), dd = ld3(e.verts, tv.w);
		bary_tet(a, b, c, dd, pos, bc);
	}
	const NRS_GLOBAL float* od = gp(e.out_density);
	const NRS_GLOBAL float* rd = gp(e.res_density);
	const float lo = ((bc[0] * od[tv.x] + bc[1] * od[tv.y]) + bc[2] * od[tv.z]) + bc[3] * od[tv.w];
	const float lr = ((bc[0] * rd[tv.x] + bc[1] * rd[tv.y]) + bc[2] * rd[tv.z]) + bc[3] * rd[tv.w];
	out_density = e.residual_amplitude * lo;
	res_density = e.residual_amplitude * lr;
	found_out = found;
	return true;
}
__device__ __forceinline__ void poisson_residual_colour(const DeviceEdit& e, uint32_t found, f3 wpos0, f3 dir, float rgb[3]) {
	const f3 pos = unwarp_position(wpos0, e.aabb);
	typedef uint32_t u4n __attribute__((ext_vector_type(4)));
	const u4n tv = gp(reinterpret_cast<const u4n*>(e.tets))[found];
	float bc[4];
	{
		const f3 a = ld3(e.verts, tv.x), b = ld3(e.verts, tv.y), c = ld3(e.verts, tv.z), dd = ld3(e.verts, tv.w);
		bary_tet(a, b, c, dd, pos, bc);
	}
	// The 4 x 27 coefficients are read through a buffer descriptor (wave-uniform base in scalar registers, one 32-bit offset per vertex, the coefficient in
	// the instruction's immediate) and in small groups: the registers of a wave, not the latency of a few more round trips, are what this instantiation is
	// short of (the kernel must fit the 128 VGPRs of the default launch shape).  Two terms of the dot product are in flight at once.
	// (Round 6 measured the other extreme: a colour's nine coefficients of a vertex as two 16-byte loads + one dword, the interpolation vertex by vertex over all nine --
	// 36 loads and 6 round trips per sample instead of 108 in 15 groups: 54 spilled registers instead of 21 and the frame 3 % SLOWER, 9.42 against 9.72 Gsamples/s on one
	// box; one vertex in flight at a time: 9.36.  profiles/r06/ab_membrane_sh_*.txt.)
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)e.shs, 0, 0x7fffffff, 0x00020000);
	const uint32_t o0 = tv.x * 108u, o1 = tv.y * 108u, o2 = tv.z * 108u, o3 = tv.w * 108u; // byte offset of a vertex's 27 floats
	#pragma unroll 1 // one colour at a time (the unrolled form holds all 108 coefficient loads in flight: 250 VGPRs)
	for (int c = 0; c < 3; ++c) {
		const int cb = c * 36;
		// SH basis (evaluate_sh9, cn:222-240), each coefficient formed where it is used from an opaque copy of the direction: nine basis values kept across
		// the colour loop are nine registers this instantiation does not have (the products are the reference's, term by term)
		const float dx = dir.x, dy = dir.y, dz = dir.z;
		auto pSH = [&](int k) -> float {
			switch (k) {
				case 0: return 0.2820947917738781f;
				case 1: return -0.48860251190292f * dy;                          // fTmpA * fS0
				case 2: return 0.4886025119029199f * dz;
				case 3: return -0.48860251190292f * dx;                          // fTmpA * fC0
				case 4: return 0.5462742152960395f * (dx * dy + dy * dx);         // fTmpC * fS1
				case 5: return (-1.092548430592079f * dz) * dy;                  // fTmpB * fS0
				case 6: return 0.9461746957575601f * (dz * dz) + -0.3153915652525201f;
				case 7: return (-1.092548430592079f * dz) * dx;                  // fTmpB * fC0
				default: return 0.5462742152960395f * (dx * dx - dy * dy);       // fTmpC * fC1
			}
		};
		// pSH.dot(sh.block<9, 1>(0, c)): Eigen's unrolled 9-term reduction, 4 | 5 -> (2|2) | (2|(1|2))
		auto L = [&](uint32_t ov, int k) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)ov, cb + 4 * k, 0)); };
		auto term = [&](int k) { return pSH(k) * (((bc[0] * L(o0, k) + bc[1] * L(o1, k)) + bc[2] * L(o2, k)) + bc[3] * L(o3, k)); };
		const float q0 = term(0), q1 = term(1); const float a01 = q0 + q1;
		__builtin_amdgcn_sched_barrier(0);
		const float q2 = term(2), q3 = term(3); const float lo4 = a01 + (q2 + q3);
		__builtin_amdgcn_sched_barrier(0);
		const float q4 = term(4), q5 = term(5); const float a45 = q4 + q5;
		__builtin_amdgcn_sched_barrier(0);
		const float q7 = term(7), q8 = term(8); const float a78 = q7 + q8;
		__builtin_amdgcn_sched_barrier(0);
		const float q6 = term(6);
		rgb[c] = lo4 + (a45 + (q6 + a78));
	}
}
// (the two steps in one call, for callers without anything in between)
__device__ __forceinline__ void poisson_residual_rgb(const DeviceEdit& e, f3 wpos0, f3 dir, float rgb[3], float& out_density, float& res_density) {
	uint32_t found;
	if (poisson_residual_find(e, wpos0, found, out_density, res_density)) poisson_residual_colour(e, found, wpos0, dir, rgb);
}

// ---- occupancy refresh pieces (update_density_grid_nerf_operator, tn:3533-3640) ----------------------------------------
// tcnn::pcg32 = PCG32 XSH-RR (see oracle/nrs_oracle.cpp for the provenance note); advance() is the LCG skip-ahead.
struct Pcg32 {
	uint64_t state, inc;
	__device__ __forceinline__ uint32_t next_uint() {
		const uint64_t old = state;
		state = old * 0x5851f42d4c957f2dULL + inc;
		const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
		const uint32_t rot = (uint32_t)(old >> 59u);
		return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
	}
	__device__ __forceinline__ float next_float() { return __uint_as_float((next_uint() >> 9) | 0x3f800000u) - 1.0f; }
	// state after `delta` steps = mult * state + plus (mod 2^64)
	__device__ __forceinline__ static void skip_coefficients(uint64_t inc, uint64_t delta, uint64_t& mult, uint64_t& plus) {
		uint64_t cur_mult = 0x5851f42d4c957f2dULL, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
		while (delta > 0) {
			if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
			cur_plus = (cur_mult + 1) * cur_plus;
			cur_mult *= cur_mult;
			delta >>= 1;
		}
		mult = acc_mult;
		plus = acc_plus;
	}
	__device__ __forceinline__ void advance(uint64_t delta) {
		uint64_t m, p;
		skip_coefficients(inc, delta, m, p);
		state = m * state + p;
	}
};

// generate_grid_samples_nerf_nonuniform, cn:179-208: cell index + a uniformly random warped position inside that cell
// `rng` arrives already advanced by 4 * i (the kernel splits that skip into a wave-uniform and a per-lane part).
__device__ __forceinline__ uint32_t generate_grid_sample(Pcg32 rng, uint32_t i, uint32_t n_elements, uint32_t step, const Box3& aabb,
                                                         const float* __restrict__ grid_in, uint32_t n_cascades, float thresh, f3& wpos) {
	const uint32_t level = (uint32_t)(rng.next_float() * n_cascades) % n_cascades;
	uint32_t idx = 0;
	#pragma unroll 1
	for (uint32_t j = 0; j < 10; ++j) {
		idx = ((i + step * n_elements) * 56924617u + j * 19349663u + 96925573u) % kGridVol;
		idx += level * kGridVol;
		if (grid_in[idx] > thresh) break;
	}
	const uint32_t pos_idx = idx % kGridVol;
	const float x = (float)morton3D_invert(pos_idx >> 0), y = (float)morton3D_invert(pos_idx >> 1), z = (float)morton3D_invert(pos_idx >> 2);
	const float rx = rng.next_float(), ry = rng.next_float(), rz = rng.next_float();
	const float s = __uint_as_float((127u + level) << 23); // scalbnf(1, level)
	const float inv = 1.0f / (float)kGrid;                  // exact: x / 128 == x * 2^-7
	const f3 pos = {((x + rx) * inv - 0.5f) * s + 0.5f, ((y + ry) * inv - 0.5f) * s + 0.5f, ((z + rz) * inv - 0.5f) * s + 0.5f};
	wpos = warp_position(pos, aabb);
	return idx;
}

// compute_poisson_residual_density_kernel, cage_deformation.cu:341-384.  wpos is the position AFTER map_positions; the
// look-up nevertheless uses the deformed mesh's LUT, as the reference does.
__device__ __forceinline__ bool poisson_residual_density(const DeviceEdit& e, f3 wpos, float& residual) {
	const f3 pos = unwarp_position(wpos, e.aabb);
	if (!box_contains(e.bbox, pos)) return false;
	const uint32_t t = find_tet(e, pos, nullptr);
	if (t == 0xffffffffu) return false;
	const uint4 tv = reinterpret_cast<const uint4*>(e.tets)[t];
	const f3 a = ld3(e.verts, tv.x), b = ld3(e.verts, tv.y), c = ld3(e.verts, tv.z), d = ld3(e.verts, tv.w);
	float bc[4];
	bary_tet(a, b, c, d, pos, bc);
	residual = ((bc[0] * e.res_density[tv.x] + bc[1] * e.res_density[tv.y]) + bc[2] * e.res_density[tv.z]) + bc[3] * e.res_density[tv.w];
	return true;
}

// composite_kernel_nerf's glow overlay (tn:806-903; m_glow_mode, off by default): a grid / cut-line visualisation that adds to (or, "grid only", replaces)
// the sample's colour and can scale its weight.  cosf is the device library's (the host-compiled reference uses glibc's): colour tolerance, not bits.
__device__ __forceinline__ void glow_overlay(const nrs_render_params& p, f3 pos, f3 cam_o, float& weight, float& r, float& g, float& b) {
	const uint32_t gm = p.glow_mode;
	const bool green_grid = gm & 1u, green_cutline = gm & 2u, mask_to_alpha = gm & 4u, radial_mode = gm & 8u, grid_mode = gm & 16u;
	float glow = 0.f, dist;
	if (radial_mode) {
		const f3 dv = pos - cam_o;
		dist = sqrtf(dot3(dv, dv));
		dist = fminf(dist, (4.5f - pos.y) * 0.333f);
	} else {
		dist = pos.y;
	}
	if (grid_mode) {
		glow = 1.f / fmaxf(1.f, dist);
	} else {
		float y = p.glow_y_cutoff - dist;
		float mask = 0.f;
		if (y > 0.f) {
			y *= 80.f;
			mask = fminf(1.f, y);
			if (green_cutline) glow += fmaxf(0.f, 1.f - fabsf(1.f - y)) * 4.f;
			if (y > 1.f) y = 1.f - (y - 1.f) * 0.05f;
			if (green_grid) glow += fmaxf(0.f, y / fmaxf(1.f, dist));
		}
		if (mask_to_alpha) weight *= mask;
	}
	if (glow > 0.f) {
		const float PI = 3.141592653589793f;
		float line = 0.f;
		#pragma unroll
		for (int k = 0; k < 4; ++k) { // y, x, z at 2, 4, 8, 16 times the base frequency, in the reference's order
			const float f = (float)(2 << k);
			line += fmaxf(0.f, cosf(pos.y * f * PI * 16.f) - 0.975f);
			line += fmaxf(0.f, cosf(pos.x * f * PI * 16.f) - 0.975f);
			line += fmaxf(0.f, cosf(pos.z * f * PI * 16.f) - 0.975f);
		}
		if (grid_mode) {
			glow = glow * line * 15.f;
			g = glow; b = glow * 0.5f; r = glow * 0.25f;
		} else {
			glow = glow * glow * 0.25f + glow * line * 15.f;
			g += glow; b += glow * 0.5f; r += glow * 0.25f;
		}
	}
}

// composite_kernel_nerf's per-sample render modes (tn:905-937): what replaces the network's colour.  pos = the (mapped) sample position in world
// units, origin = payload.origin, cdt = unwarp_dt(input->dt), alpha after the show_accel override.  Normals / EncodingVis are refused by the host.
__device__ __forceinline__ void render_mode_rgb(const nrs_render_params& p, f3 pos, f3 origin, f3 cam_fwd, float cdt, float alpha, float& r, float& g, float& b) {
	switch (p.render_mode) {
		case NRS_RENDER_POSITIONS:
			if (p.show_accel) { // colour of the occupancy cell the sample stands in (tn:911-920); tcnn::default_rng_t = pcg32(initstate, initseq 1)
				const uint32_t mip = (uint32_t)max((int)p.min_mip, mip_from_pos(pos));
				const float res = (float)(kGrid >> mip);
				const int ix = (int)(pos.x * res), iy = (int)(pos.y * res), iz = (int)(pos.z * res);
				Pcg32 rng{0ull, 3ull};
				rng.next_uint();
				rng.state += (uint64_t)(int64_t)(int)((uint32_t)ix + (uint32_t)iy * 232323u + (uint32_t)iz * 727272u);
				rng.next_uint();
				r = 1.f - (float)mip * (1.f / (float)(kCascades - 1));
				g = rng.next_float();
				b = rng.next_float();
			} else {
				r = (pos.x - 0.5f) / 2.0f + 0.5f; g = (pos.y - 0.5f) / 2.0f + 0.5f; b = (pos.z - 0.5f) / 2.0f + 0.5f;
			}
			break;
		case NRS_RENDER_DEPTH: r = g = b = dot3(cam_fwd, pos - origin) * p.depth_scale; break;
		case NRS_RENDER_DISTANCE: { const f3 dv = pos - origin; r = g = b = sqrtf(dot3(dv, dv)) * p.depth_scale; break; }
		case NRS_RENDER_STEPSIZE: r = g = b = warp_dt(cdt); break;
		case NRS_RENDER_AO: r = g = b = alpha; break;
		default: break; // Shade, Cost: the network's colour
	}
}

// ---- activations (cn:38-66) and shade (common_device.cuh:31-37) ---------------------------------------------------
__device__ __forceinline__ float network_to_rgb(float v, uint32_t act) {
	switch (act) {
		case NRS_ACT_RELU: return v > 0.f ? v : 0.f;
		case NRS_ACT_LOGISTIC: return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); // 1-ulp rcp: inside the stated colour tolerance
		case NRS_ACT_EXPONENTIAL: return __expf(clampf_(v, -10.f, 10.f));
		default: return v;
	}
}
__device__ __forceinline__ float network_to_density(float v, uint32_t act) {
	switch (act) {
		case NRS_ACT_RELU: return v > 0.f ? v : 0.f;
		case NRS_ACT_LOGISTIC: return 1.0f / (1.0f + __expf(-v));
		case NRS_ACT_EXPONENTIAL: return __expf(v);
		default: return v;
	}
}
__device__ __forceinline__ float network_to_density_derivative(float v, uint32_t act) { // tn:308-317
	switch (act) {
		case NRS_ACT_RELU: return v > 0.0f ? 1.0f : 0.0f;
		case NRS_ACT_LOGISTIC: { const float density = 1.0f / (1.0f + __expf(-v)); return density * (1 - density); }
		case NRS_ACT_EXPONENTIAL: return __expf(clampf_(v, -15.0f, 15.0f));
		default: return 1.0f;
	}
}
// x^2.4 as exp2(2.4 * log2 x) on the transcendental unit (v_log_f32 / v_exp_f32, ~1 ulp each): |rel. error| < 1e-5 on
// (0.04, 1], far inside the stated colour tolerance, and ~100 instructions cheaper per channel than powf.
__device__ __forceinline__ float srgb_to_linear(float s) {
	if (s <= 0.04045f) return s * (1.0f / 12.92f);
	return __builtin_amdgcn_exp2f(2.4f * __builtin_amdgcn_logf((s + 0.055f) * (1.0f / 1.055f)));
}

} // namespace nrs
