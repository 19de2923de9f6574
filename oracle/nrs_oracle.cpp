/*
 * nrs_oracle.cpp -- CPU ORACLE for the NeRFshop volumetric render path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is a plain C++ restatement (no Eigen, no tiny-cuda-nn, no GPU) of the reference's algorithm
 * for the path BASELINE.json:north_star names.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load it -- as the checker, never as the thing measured or shipped.  Nothing
 * under nerfshop_amd/ imports, links or calls it.
 *
 * PARITY: pinned to the reference's own compiled code everywhere except tiny-cuda-nn.  The reference ships no tests, golden vectors,
 * fixtures or snapshots for this path and cannot be built as a whole (nvcc, Eigen and tcnn are absent), but its
 * render-path headers, src/common_nerf.cu and the kernels of testbed_nerf.cu / cage_deformation.cu / tet_mesh.cu / affine_duplication.cu
 * compile as host code against the stand-ins of oracle/ref_stubs/ (oracle/Makefile -> oracle/_ref/libref_render.so, oracle/ref_render.cpp,
 * oracle/ref_extract.py).  tests/test_ref_pin.py holds this file to that library bit for bit: every header function on 1e5 seeded
 * inputs, the operators, the LUT builder, rotations, MVC, the membrane interpolation, 10 whole frames and 5 whole sample streams; the
 * same numbers travel as tests/golden/ref_pin_golden.npz.  To match, every dot product / small matrix product / camera transform here
 * follows Eigen's reduction order (sum3: x0 + (x1 + x2); 9 terms: ((x0+x1)+(x2+x3)) + ((x4+x5)+(x6+(x7+x8)))) and mvc.h's float / double
 * overload choices.
 * PARITY UNPINNED for tiny-cuda-nn (hash grid, fully fused MLPs, SH encoding): an EMPTY, un-pinned submodule (fork
 * gitlab.inria.fr/cjambon/tcnn-pyngp, branch pyngp-api, .gitmodules:16-19).  Its published algorithm is restated; its two
 * ambiguous roundings are switchable (orc_model_set_numerics, mirrored by nrs_model_set_numerics): grid accumulation fp32 per corner
 * (fmaf, corners 0..7) rounded to fp16 once | every corner's product rounded to fp16 and added in fp16; MLP products fp16 x fp16
 * accumulated exactly (double) then rounded fp32 -> ReLU -> fp16 | the running sum rounded to fp16 after every 16-wide k step.
 * Also checked: pcg32 against the PCG library's demo vector, the level table against instant-ngp's 12 196 240 lego parameters,
 * hand-derived known answers (tests/test_oracle_kat.py).
 *
 * Floating point: built with -ffp-contract=off; fused multiply-adds appear only as explicit fmaf().
 * The HIP kernels are built the same way so that ray/sample indexing is bit-exact oracle <-> HIP.
 *
 * Citations are file:line in the reference checkout.  "tn" = src/testbed_nerf.cu, "cn" = src/common_nerf.cu.
 */
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/nrs.h"

namespace {

// ------------------------------------------------------------------------------------------------
// fp16 <-> fp32 (software, round-to-nearest-even; gcc 11 has no _Float16 on x86)
// ------------------------------------------------------------------------------------------------
inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

inline float h2f(uint16_t h) {
	uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
	uint32_t exp = (h >> 10) & 0x1fu;
	uint32_t man = h & 0x3ffu;
	if (exp == 0) {
		if (man == 0) return u2f(sign);
		// subnormal: value = man * 2^-24
		float v = (float)man * 5.9604644775390625e-08f;
		return sign ? -v : v;
	}
	if (exp == 31) return u2f(sign | 0x7f800000u | (man << 13));
	return u2f(sign | ((exp + 112u) << 23) | (man << 13));
}

inline uint16_t f2h(float f) {
	uint32_t x = f2u(f);
	uint32_t sign = (x >> 16) & 0x8000u;
	uint32_t ax = x & 0x7fffffffu;
	if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0u)); // inf / nan
	if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); // >= 65520 rounds to inf
	if (ax < 0x33000001u) return (uint16_t)sign;              // < 2^-25 (and exactly 2^-25 ties to even -> 0)
	int e = (int)(ax >> 23) - 127;
	uint32_t m = (ax & 0x7fffffu) | 0x800000u; // 24-bit significand
	int shift = (e < -14) ? (13 + (-14 - e)) : 13; // bits to drop
	uint32_t kept = m >> shift;
	uint32_t rem = m & ((1u << shift) - 1u);
	uint32_t half = 1u << (shift - 1);
	if (rem > half || (rem == half && (kept & 1u))) kept++;
	uint32_t h;
	if (e < -14) h = kept;                       // subnormal (kept may carry into the normal range: fine)
	else h = ((uint32_t)(e + 15) << 10) + (kept - 0x400u); // kept has the implicit bit at 0x400; carry propagates
	return (uint16_t)(sign | h);
}

// ------------------------------------------------------------------------------------------------
// small vectors
// ------------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
// Eigen evaluates fixed-size reductions (dot, squaredNorm, the inner sums of small matrix products) with its complete unroller,
// Eigen/src/Core/Redux.h redux_novec_unroller: the range is split at len/2, so a 3-term sum is x0 + (x1 + x2) -- NOT (x0 + x1) + x2.
// Pinned by the reference's own code compiled against oracle/ref_stubs/Eigen (tests/test_ref_pin.py).
inline float sum3(float a, float b, float c) { return a + (b + c); }
inline float dot(V3 a, V3 b) { return sum3(a.x * b.x, a.y * b.y, a.z * b.z); }
// M * v for a column-major 3x3 (Eigen's coefficient-based product: row . v with the reduction order above)
inline V3 mat3_mul(const float* M, V3 v) {
	return {sum3(M[0] * v.x, M[3] * v.y, M[6] * v.z), sum3(M[1] * v.x, M[4] * v.y, M[7] * v.z), sum3(M[2] * v.x, M[5] * v.y, M[8] * v.z)};
}
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float comp(const V3& a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

struct Box { V3 mn, mx; };
inline bool box_contains(const Box& b, V3 p) { // bounding_box.cuh:243-248
	return p.x >= b.mn.x && p.x <= b.mx.x && p.y >= b.mn.y && p.y <= b.mx.y && p.z >= b.mn.z && p.z <= b.mx.z;
}

// ------------------------------------------------------------------------------------------------
// constants, common_nerf.h:16-39, tn:56-59
// ------------------------------------------------------------------------------------------------
constexpr uint32_t GRID = 128;
constexpr uint32_t CASCADES = 5;
constexpr uint32_t GRIDVOL = GRID * GRID * GRID;
constexpr float SQRT3 = 1.73205080757f;
constexpr float MIN_STEP = SQRT3 / 1024;                          // MIN_CONE_STEPSIZE
constexpr float MAX_STEP = MIN_STEP * (1 << (CASCADES - 1)) * 1024 / GRID; // MAX_CONE_STEPSIZE
constexpr float NEAR_DISTANCE = 0.05f;
constexpr uint32_t MARCH_ITER = 10000;
constexpr uint32_t MIN_STEPS_INBETWEEN_COMPACTION = 1, MAX_STEPS_INBETWEEN_COMPACTION = 8;

// ------------------------------------------------------------------------------------------------
// Morton code (tcnn morton3D: 10 bits per axis, x lowest)
// ------------------------------------------------------------------------------------------------
inline uint32_t expand_bits(uint32_t v) {
	v = (v * 0x00010001u) & 0xFF0000FFu;
	v = (v * 0x00000101u) & 0x0F00F00Fu;
	v = (v * 0x00000011u) & 0xC30C30C3u;
	v = (v * 0x00000005u) & 0x49249249u;
	return v;
}
inline uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) { return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2); }
inline uint32_t morton3D_invert(uint32_t x) {
	x = x & 0x49249249u;
	x = (x | (x >> 2)) & 0xc30c30c3u;
	x = (x | (x >> 4)) & 0x0f00f00fu;
	x = (x | (x >> 8)) & 0xff0000ffu;
	x = (x | (x >> 16)) & 0x0000ffffu;
	return x;
}

// ------------------------------------------------------------------------------------------------
// Scrambled Sobol, random_val.cuh:159-288, 317-322.  Only dimensions 0 and 1 are used on the path.
// Direction numbers are generated, not tabulated: dim 0 is the van der Corput sequence (bit b ->
// 0x80000000 >> b), dim 1 satisfies v[b] = v[b-1] ^ (v[b-1] >> 1); tests/golden pins both against
// the reference's table.
// ------------------------------------------------------------------------------------------------
inline uint32_t sobol_dir(uint32_t dim, uint32_t bit) {
	if (dim == 0) return 0x80000000u >> bit;
	uint32_t v = 0x80000000u;
	for (uint32_t b = 0; b < bit; ++b) v ^= v >> 1;
	return v;
}
inline uint32_t sobol(uint32_t index, uint32_t dim) { // random_val.cuh:159-216
	uint32_t X = 0;
	for (uint32_t bit = 0; bit < 32; ++bit)
		if ((index >> bit) & 1u) X ^= sobol_dir(dim, bit);
	return X;
}
inline uint32_t hash_combine(uint32_t seed, uint32_t v) { return seed ^ (v + (seed << 6) + (seed >> 2)); } // :226
inline uint32_t reverse_bits(uint32_t x) { // :230
	x = (((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1));
	x = (((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2));
	x = (((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4));
	x = (((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8));
	return (x >> 16) | (x << 16);
}
inline uint32_t laine_karras_permutation(uint32_t x, uint32_t seed) { // :238
	x += seed;
	x ^= x * 0x6c50b47cu;
	x ^= x * 0xb82f1e52u;
	x ^= x * 0xc7afe638u;
	x ^= x * 0x8d22f6e6u;
	return x;
}
inline uint32_t nested_uniform_scramble_base2(uint32_t x, uint32_t seed) { // :247
	x = reverse_bits(x);
	x = laine_karras_permutation(x, seed);
	x = reverse_bits(x);
	return x;
}
constexpr float SOBOL_S = 2.3283064365386963e-10f; // float(1.0 / 2^32)
inline float ld_random_val(uint32_t index, uint32_t seed, uint32_t dim = 0) { // :284
	index = nested_uniform_scramble_base2(index, seed);
	return (float)nested_uniform_scramble_base2(sobol(index, dim), hash_combine(seed, dim)) * SOBOL_S;
}
inline void ld_random_val_2d(uint32_t index, uint32_t seed, float out[2]) { // :278 via shuffled_scrambled_sobol2d :263
	index = nested_uniform_scramble_base2(index, seed);
	for (uint32_t i = 0; i < 2; ++i)
		out[i] = (float)nested_uniform_scramble_base2(sobol(index, i), hash_combine(seed, i)) * SOBOL_S;
}
inline float fractf(float x) { return x - floorf(x); } // :76
inline void ld_random_pixel_offset(uint32_t spp, float out[2]) { // :317-322
	float a[2], b[2];
	ld_random_val_2d(0, 0xdeadbeefu, a);
	ld_random_val_2d(spp, 0xdeadbeefu, b);
	out[0] = fractf((0.5f - a[0]) + b[0]);
	out[1] = fractf((0.5f - a[1]) + b[1]);
}

// ------------------------------------------------------------------------------------------------
// grid / step math, cn:80-177
// ------------------------------------------------------------------------------------------------
inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (hi < v ? hi : v); }
inline float calc_dt(float t, float cone_angle) { return clampf(t * cone_angle, MIN_STEP, MAX_STEP); } // cn:89
inline float signf(float x) { return copysignf(1.0f, x); }                                             // common.h:183

// exponent e such that x = m * 2^e with 0.5 <= |m| < 1 (frexpf's second result); 0 for x == 0
inline int frexp_exponent(float x) {
	uint32_t u = f2u(x) & 0x7fffffffu;
	if (u == 0) return 0;
	uint32_t ef = u >> 23;
	if (ef == 0) return -117 - __builtin_clz(u); // subnormal: value = man * 2^-149, top bit b = 31 - clz -> e = b - 148
	return (int)ef - 126;
}

inline float distance_to_next_voxel(V3 pos, V3 dir, V3 idir, uint32_t res) { // cn:93-101
	V3 p = (float)res * pos;
	float tx = (floorf(p.x + 0.5f + 0.5f * signf(dir.x)) - p.x) * idir.x;
	float ty = (floorf(p.y + 0.5f + 0.5f * signf(dir.y)) - p.y) * idir.y;
	float tz = (floorf(p.z + 0.5f + 0.5f * signf(dir.z)) - p.z) * idir.z;
	float t = fminf(fminf(tx, ty), tz);
	return fmaxf(t / (float)res, 0.0f);
}
inline float advance_to_next_voxel(float t, float cone_angle, V3 pos, V3 dir, V3 idir, uint32_t res) { // cn:103-115
	float t_target = t + distance_to_next_voxel(pos, dir, idir, res);
	do {
		t += calc_dt(t, cone_angle);
	} while (t < t_target);
	return t;
}
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline uint32_t cascaded_grid_idx_at(V3 pos, uint32_t mip) { // cn:117-136
	float mip_scale = scalbnf(1.0f, -(int)mip);
	pos = pos - v3(0.5f, 0.5f, 0.5f);
	pos = pos * mip_scale;
	pos = pos + v3(0.5f, 0.5f, 0.5f);
	int ix = (int)(pos.x * (float)GRID), iy = (int)(pos.y * (float)GRID), iz = (int)(pos.z * (float)GRID);
	return morton3D((uint32_t)clampi(ix, 0, GRID - 1), (uint32_t)clampi(iy, 0, GRID - 1), (uint32_t)clampi(iz, 0, GRID - 1));
}
inline uint32_t grid_mip_offset(uint32_t mip) { return GRIDVOL * mip; } // cn:76
inline bool get_bitfield_at(uint32_t cell_idx, uint32_t level, const uint8_t* bitfield) { // cn:153
	return bitfield[cell_idx / 8 + grid_mip_offset(level) / 8] & (1 << (cell_idx % 8));
}
inline void set_bitfield_at(uint32_t cell_idx, uint32_t level, bool value, uint8_t* bitfield) { // cn:157
	uint32_t bit = cell_idx % 8, mask = 1u << bit;
	uint8_t& b = bitfield[cell_idx / 8 + grid_mip_offset(level) / 8];
	b = (uint8_t)((b & ~mask) | ((uint32_t)value << bit));
}
inline bool density_grid_occupied_at(V3 pos, const uint8_t* bitfield, uint32_t mip) { // cn:138
	return get_bitfield_at(cascaded_grid_idx_at(pos, mip), mip, bitfield);
}
inline int mip_from_pos(V3 pos) { // cn:163-168
	float maxval = fmaxf(fmaxf(fabsf(pos.x - 0.5f), fabsf(pos.y - 0.5f)), fabsf(pos.z - 0.5f));
	int exponent = frexp_exponent(maxval);
	return std::min((int)CASCADES - 1, std::max(0, exponent + 1));
}
inline int mip_from_dt(float dt, V3 pos) { // cn:170-177
	int mip = mip_from_pos(pos);
	dt *= 2 * GRID;
	if (dt < 1.f) return mip;
	int exponent = frexp_exponent(dt);
	return std::min((int)CASCADES - 1, std::max(exponent, mip));
}
inline V3 warp_position(V3 pos, const Box& aabb) { // cn:5 -> bounding_box.cuh:96 relative_pos
	V3 d = aabb.mx - aabb.mn;
	return {(pos.x - aabb.mn.x) / d.x, (pos.y - aabb.mn.y) / d.y, (pos.z - aabb.mn.z) / d.z};
}
inline V3 unwarp_position(V3 pos, const Box& aabb) { // cn:12
	V3 d = aabb.mx - aabb.mn;
	return {aabb.mn.x + pos.x * d.x, aabb.mn.y + pos.y * d.y, aabb.mn.z + pos.z * d.z};
}
inline V3 warp_direction(V3 d) { return {(d.x + 1.0f) * 0.5f, (d.y + 1.0f) * 0.5f, (d.z + 1.0f) * 0.5f}; }   // cn:20
inline V3 unwarp_direction(V3 d) { return {d.x * 2.0f - 1.0f, d.y * 2.0f - 1.0f, d.z * 2.0f - 1.0f}; }       // cn:24
inline float warp_dt(float dt) { // cn:28
	float max_stepsize = MIN_STEP * (1 << (CASCADES - 1));
	return (dt - MIN_STEP) / (max_stepsize - MIN_STEP);
}
inline float unwarp_dt(float dt) { // cn:33
	float max_stepsize = MIN_STEP * (1 << (CASCADES - 1));
	return dt * (max_stepsize - MIN_STEP) + MIN_STEP;
}

// tcnn::pcg32 (provenance: see the note above orc_pcg32_seed)
struct Pcg32 {
	uint64_t state, inc;
	uint32_t next_uint() {
		uint64_t old = state;
		state = old * 0x5851f42d4c957f2dULL + inc;
		uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
		uint32_t rot = (uint32_t)(old >> 59u);
		return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
	}
	float next_float() {
		uint32_t u = (next_uint() >> 9) | 0x3f800000u;
		float f;
		memcpy(&f, &u, 4);
		return f - 1.0f;
	}
	void advance(uint64_t delta) {
		uint64_t cur_mult = 0x5851f42d4c957f2dULL, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
		while (delta > 0) {
			if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
			cur_plus = (cur_mult + 1) * cur_plus;
			cur_mult *= cur_mult;
			delta >>= 1;
		}
		state = acc_mult * state + acc_plus;
	}
};
inline Pcg32 pcg32_seeded(uint64_t initstate) { // pcg32(initstate, initseq = 1)
	Pcg32 r{0u, (1u << 1u) | 1u};
	r.next_uint();
	r.state += initstate;
	r.next_uint();
	return r;
}

// ------------------------------------------------------------------------------------------------
// Rays: pixel_to_ray (common_device.cuh:245-295), init_rays_with_payload_kernel_nerf (tn:2512-2616)
// ------------------------------------------------------------------------------------------------
struct Payload { // nerf.h:23
	V3 origin, dir;
	float t, max_weight;
	uint32_t idx;
	uint32_t n_steps;
	bool alive;
};

inline void ray_intersect(const Box& b, V3 pos, V3 dir, float& tmin_out, float& tmax_out) { // bounding_box.cuh:180-238
	const float FMAX = std::numeric_limits<float>::max();
	float tmin = (b.mn.x - pos.x) / dir.x, tmax = (b.mx.x - pos.x) / dir.x;
	if (tmin > tmax) std::swap(tmin, tmax);
	float tymin = (b.mn.y - pos.y) / dir.y, tymax = (b.mx.y - pos.y) / dir.y;
	if (tymin > tymax) std::swap(tymin, tymax);
	if (tmin > tymax || tymin > tmax) { tmin_out = FMAX; tmax_out = FMAX; return; }
	if (tymin > tmin) tmin = tymin;
	if (tymax < tmax) tmax = tymax;
	float tzmin = (b.mn.z - pos.z) / dir.z, tzmax = (b.mx.z - pos.z) / dir.z;
	if (tzmin > tzmax) std::swap(tzmin, tzmax);
	if (tmin > tzmax || tzmin > tmax) { tmin_out = FMAX; tmax_out = FMAX; return; }
	if (tzmin > tmin) tmin = tzmin;
	if (tzmax < tmax) tmax = tzmax;
	tmin_out = tmin; tmax_out = tmax;
}

struct Camera { float m[12]; }; // 3x4 column-major
inline V3 cam_col(const float* m, int c) { return {m[3 * c + 0], m[3 * c + 1], m[3 * c + 2]}; }

// square2disk_shirley, random_val.cuh:109-125 (sincosf = the host libm's, as in the reference compiled as host code)
inline void square2disk_shirley(float a, float b, float& ox, float& oy) {
	const float PI = 3.14159265358979323846f;
	float phi, r;
	if (a * a > b * b) { r = a; phi = (PI / 4.0f) * (b / a); }
	else { r = b; phi = (PI / 2.0f) - (PI / 4.0f) * (a / b); }
	float sin_phi, cos_phi;
	sincosf(phi, &sin_phi, &cos_phi);
	ox = r * cos_phi; oy = r * sin_phi;
}
// render_nerf's plane_z (tn:3067-3070): m_slice_plane_z + m_scale, negated in render mode Slice
inline float frame_plane_z(const nrs_render_params& p) { return p.render_mode == NRS_RENDER_SLICE ? -p.slice_plane_z : p.slice_plane_z; }

// apply_camera_distortion (OpenCV radial k1 k2 + tangential p1 p2), common_device.cuh:146-159
inline void apply_camera_distortion(const float* prm, float u, float v, float* du, float* dv) {
	const float k1 = prm[0], k2 = prm[1], p1 = prm[2], p2 = prm[3];
	const float u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2;
	const float radial = k1 * r2 + k2 * r2 * r2;
	*du = u * radial + 2.f * p1 * uv + p2 * (r2 + 2.f * u2);
	*dv = v * radial + 2.f * p2 * uv + p1 * (r2 + 2.f * v2);
}
// iterative_camera_undistortion, common_device.cuh:161-200: Newton with a central-difference Jacobian, Eigen's 2x2 inverse (adjugate times 1 / det)
inline void iterative_camera_undistortion(const float* prm, float* u, float* v) {
	const float kMaxStepNorm = 1e-10f, kRelStepSize = 1e-6f, eps = std::numeric_limits<float>::epsilon();
	const float x00 = *u, x01 = *v;
	float x0 = *u, x1 = *v;
	for (uint32_t i = 0; i < 100; ++i) {
		const float step0 = std::max(eps, std::abs(kRelStepSize * x0)), step1 = std::max(eps, std::abs(kRelStepSize * x1));
		float dx0, dx1, b00, b01, f00, f01, b10, b11, f10, f11;
		apply_camera_distortion(prm, x0, x1, &dx0, &dx1);
		apply_camera_distortion(prm, x0 - step0, x1, &b00, &b01);
		apply_camera_distortion(prm, x0 + step0, x1, &f00, &f01);
		apply_camera_distortion(prm, x0, x1 - step1, &b10, &b11);
		apply_camera_distortion(prm, x0, x1 + step1, &f10, &f11);
		const float J00 = 1 + (f00 - b00) / (2 * step0), J01 = (f10 - b10) / (2 * step1), J10 = (f01 - b01) / (2 * step0), J11 = 1 + (f11 - b11) / (2 * step1);
		const float invdet = 1.0f / (J00 * J11 - J10 * J01);              // Eigen: compute_inverse<..., 2>
		const float i00 = J11 * invdet, i10 = -J10 * invdet, i01 = -J01 * invdet, i11 = J00 * invdet;
		const float r0 = x0 + dx0 - x00, r1 = x1 + dx1 - x01;              // (x + dx - x0)
		const float s0 = i00 * r0 + i01 * r1, s1 = i10 * r0 + i11 * r1;    // J^-1 * r: 2-term rows
		x0 -= s0; x1 -= s1;
		if (s0 * s0 + s1 * s1 < kMaxStepNorm) break;
	}
	*u = x0; *v = x1;
}
// read_image<2> (common_device.cuh:80-110): bilinear lookup, texels clamped
inline void read_image2(const float* data, const int32_t res[2], float px, float py, float out[2]) {
	const float fx = px * (float)(res[0] - 1), fy = py * (float)(res[1] - 1);
	const int tx = (int)fx, ty = (int)fy;
	const float wx = fx - (float)tx, wy = fy - (float)ty;
	auto rd = [&](int x, int y, int c) {
		x = std::max(std::min(x, res[0] - 1), 0); y = std::max(std::min(y, res[1] - 1), 0);
		return data[((size_t)x + (size_t)y * res[0]) * 2 + c];
	};
	for (int c = 0; c < 2; ++c)
		out[c] = (((1 - wx) * (1 - wy) * rd(tx, ty, c) + (wx) * (1 - wy) * rd(tx + 1, ty, c)) + (1 - wx) * (wy)*rd(tx, ty + 1, c)) + (wx) * (wy)*rd(tx + 1, ty + 1, c);
}
// read_envmap (envmap.cuh:30-63): the direction in spherical coordinates (dir_to_spherical_unorm, random_val.cuh:64-69: acosf / atan2f of the host libm),
// bilinear lookup wrapping in x and clamped in y
inline void read_envmap(const float* data, const int32_t res[2], V3 dir, float out[4]) {
	const float PI = 3.14159265358979323846f;
	const V3 d = {dir.z, -dir.x, dir.y};
	const float cos_theta = fminf(fmaxf(d.z, -1.0f), 1.0f);
	const float theta = acosf(cos_theta);
	const float phi = atan2f(d.y, d.x);
	const float cyl_x = theta / PI, cyl_y = (phi / (2.0f * PI) + 0.5f);
	const float fx = cyl_y * (float)(res[0] - 1), fy = cyl_x * (float)(res[1] - 1);
	const int tx = (int)fx, ty = (int)fy;
	const float wx = fx - (float)tx, wy = fy - (float)ty;
	auto rd = [&](int x, int y, int c) {
		if (x < 0) x += res[0]; else if (x >= res[0]) x -= res[0];
		y = std::max(std::min(y, res[1] - 1), 0);
		return data[((size_t)x + (size_t)y * res[0]) * 4 + c];
	};
	for (int c = 0; c < 4; ++c)
		out[c] = (((1 - wx) * (1 - wy) * rd(tx, ty, c) + (wx) * (1 - wy) * rd(tx + 1, ty, c)) + (1 - wx) * (wy)*rd(tx, ty + 1, c)) + (wx) * (wy)*rd(tx + 1, ty + 1, c);
}

// pixel_to_ray, common_device.cuh:245-295: origin and UN-normalised direction
inline void pixel_to_ray(const nrs_render_params& p, const float* cam, uint32_t x, uint32_t y, float focus_z, float dof, V3& o, V3& d) {
	const uint32_t W = (uint32_t)p.resolution[0], H = (uint32_t)p.resolution[1];
	float offset[2];
	ld_random_pixel_offset(p.snap_to_pixel_centers ? 0 : p.spp_index, offset);
	float uvx = ((float)x + offset[0]) / (float)W;
	float uvy = ((float)y + offset[1]) / (float)H;
	V3 dir;
	if (p.distortion_mode == 2) { // FTheta, :231-243, :263-267
		const float* prm = p.distortion_params;
		const float xpix = (uvx - p.screen_center[0]) * prm[5], ypix = (uvy - p.screen_center[1]) * prm[6];
		const float norm = sqrtf(xpix * xpix + ypix * ypix);
		const float alpha = prm[0] + norm * (prm[1] + norm * (prm[2] + norm * (prm[3] + norm * prm[4])));
		float sin_alpha, cos_alpha;
		sincosf(alpha, &sin_alpha, &cos_alpha);
		if (cos_alpha <= std::numeric_limits<float>::min() || norm == 0.f) { // error direction: a point outside the aabb so the pixel is not rendered
			o = v3(1000.f, 0.f, 0.f); d = v3(0.f, 0.f, 1.f);
			return;
		}
		sin_alpha *= 1.f / norm;
		dir = {sin_alpha * xpix, sin_alpha * ypix, cos_alpha};
	} else {
		dir = {(uvx - p.screen_center[0]) * (float)W / p.focal_length[0], (uvy - p.screen_center[1]) * (float)H / p.focal_length[1], 1.0f};
		if (p.distortion_mode == 1) iterative_camera_undistortion(p.distortion_params, &dir.x, &dir.y);
	}
	if (p.d_distortion_map) { // :278-280
		float dd[2];
		read_image2(p.d_distortion_map, p.distortion_resolution, uvx, uvy, dd);
		dir.x += dd[0]; dir.y += dd[1];
	}
	d = mat3_mul(cam, dir); // camera_matrix.block<3, 3>(0, 0) * dir, common_device.cuh:282
	o = cam_col(cam, 3);
	if (dof == 0.0f) return;
	// thin lens, common_device.cuh:286-292
	V3 lookat = o + d * focus_z;
	float r2[2];
	ld_random_val_2d(p.spp_index, x * 19349663u + y * 96925573u, r2);
	float bx, by;
	square2disk_shirley(r2[0] * 2.0f - 1.0f, r2[1] * 2.0f - 1.0f, bx, by);
	bx = dof * bx; by = dof * by;
	// origin += camera_matrix.block<3, 2>(0, 0) * blur: a 2-term coefficient-based product per row
	o = {o.x + (cam[0] * bx + cam[3] * by), o.y + (cam[1] * bx + cam[4] * by), o.z + (cam[2] * bx + cam[5] * by)};
	V3 diff = lookat - o;
	d = {diff.x / focus_z, diff.y / focus_z, diff.z / focus_z};
}

inline void init_ray(const nrs_render_params& p, uint32_t x, uint32_t y, Payload& payload, float& depth_out, float* frame_px = nullptr) {
	const uint32_t W = (uint32_t)p.resolution[0], H = (uint32_t)p.resolution[1];
	const uint32_t idx = x + W * y;
	const Box aabb{v3(p.render_aabb_min[0], p.render_aabb_min[1], p.render_aabb_min[2]),
	               v3(p.render_aabb_max[0], p.render_aabb_max[1], p.render_aabb_max[2])};
	const float plane_z = frame_plane_z(p);
	float dof = p.dof;
	if (plane_z < 0) dof = 0.0f; // tn:2543-2545
	// tn:2551-2553
	float u = ((float)x + 0.5f) * (1.f / (float)W);
	float v = ((float)y + 0.5f) * (1.f / (float)H);
	float ray_time = p.rolling_shutter[0] + p.rolling_shutter[1] * u + p.rolling_shutter[2] * v +
	                 p.rolling_shutter[3] * ld_random_val(p.spp_index, idx * 72239731u);
	float cam[12];
	for (int i = 0; i < 12; ++i) cam[i] = p.camera_matrix0[i] * ray_time + p.camera_matrix1[i] * (1.f - ray_time); // tn:2559
	V3 o, d;
	pixel_to_ray(p, cam, x, y, plane_z, dof, o, d);

	payload.max_weight = 0.0f; // tn:2573
	if (plane_z < 0) { // Slice: the ray stands on the plane at distance -plane_z along the view axis, tn:2575-2585
		float n = sqrtf(dot(d, d));
		payload.origin = o;
		payload.dir = (1.0f / n) * d;
		payload.t = -plane_z * n;
		payload.idx = idx;
		payload.n_steps = 0;
		payload.alive = false;
		depth_out = -plane_z;
		return;
	}
	depth_out = 1e10f;         // tn:2586
	float n = sqrtf(dot(d, d)); // .normalized(), tn:2588
	d = {d.x / n, d.y / n, d.z / n};
	if (p.d_envmap && frame_px) read_envmap(p.d_envmap, p.envmap_resolution, d, frame_px); // tn:2590-2592: replaces the frame value
	float tmin, tmax;
	ray_intersect(aabb, o, d, tmin, tmax);
	float t = fmaxf(tmin, NEAR_DISTANCE) + 1e-6f; // tn:2594
	payload.idx = idx;
	payload.n_steps = 0;
	payload.origin = o;
	payload.dir = d;
	payload.t = t;
	if (!box_contains(aabb, o + d * t)) { // tn:2596-2600
		payload.alive = false;
		return;
	}
	if (p.render_mode == NRS_RENDER_DISTORTION) { // tn:2602-2613: the distortion map as a picture, no tracing
		if (frame_px) {
			if (p.d_distortion_map) {
				float dd[2];
				read_image2(p.d_distortion_map, p.distortion_resolution, ((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H, dd);
				frame_px[0] = dd[0] * 50.0f + 0.5f; frame_px[1] = dd[1] * 50.0f + 0.5f;
			} else {
				frame_px[0] = 0.5f; frame_px[1] = 0.5f;
			}
			frame_px[2] = 0.5f; frame_px[3] = 1.0f;
		}
		depth_out = 1.0f;
		payload.origin = o + d * 10000.0f;
		payload.alive = false;
		return;
	}
	payload.alive = true;
}

// advance_pos_nerf, tn:557-606
inline void advance_pos(const nrs_render_params& p, const uint8_t* bitfield, Payload& payload, uint32_t pixel_i) {
	if (!payload.alive) return;
	const Box aabb{v3(p.render_aabb_min[0], p.render_aabb_min[1], p.render_aabb_min[2]),
	               v3(p.render_aabb_max[0], p.render_aabb_max[1], p.render_aabb_max[2])};
	V3 origin = payload.origin, dir = payload.dir;
	V3 idir = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
	float cone_angle = p.cone_angle_constant; // calc_cone_angle returns the constant, cn:80-87
	float t = payload.t;
	float dt = calc_dt(t, cone_angle);
	t += ld_random_val(p.spp_index, pixel_i * 786433u) * dt;
	V3 pos;
	while (1) {
		pos = origin + dir * t;
		if (!box_contains(aabb, pos)) { payload.alive = false; break; }
		dt = calc_dt(t, cone_angle);
		uint32_t mip = std::max(p.min_mip, (uint32_t)mip_from_dt(dt, pos));
		if (density_grid_occupied_at(pos, bitfield, mip)) break;
		uint32_t res = GRID >> mip;
		t = advance_to_next_voxel(t, cone_angle, pos, dir, idir, res);
	}
	payload.t = t;
}

// ------------------------------------------------------------------------------------------------
// Network: hash grid + SH + two MLPs (tiny-cuda-nn semantics; wiring nerf_network_full.h:40-96)
// ------------------------------------------------------------------------------------------------
constexpr uint32_t MAX_LEVELS = 16;
struct LevelTable {
	float scale[MAX_LEVELS];
	uint32_t resolution[MAX_LEVELS], offset[MAX_LEVELS], count[MAX_LEVELS], hashed[MAX_LEVELS];
	uint32_t total_entries;
};

// The architectures of configs/nerf/: base.json and its relatives that keep the hash grid and the 64-wide density network -- rgb network with 0 (CutlassMLP,
// base_0layer.json), 1, 2 (base.json) or 3 hidden layers, and no rgb network / direction encoding at all (base_nodir.json -> NerfNetworkNoDir,
// testbed.cu:2314-2353: nrs_model_desc::sh_degree == 0).  Table sizes: log2_hashmap_size 14 (base_14.json), 15 (small.json), 19, 21 (big.json), ...
bool desc_supported(const nrs_model_desc& d) {
	const bool trunk = d.n_levels == 16 && d.n_features_per_level == 2 && d.n_neurons == 64 && d.density_hidden_layers <= 1 && d.density_output_dims == 16 &&
	                   d.log2_hashmap_size >= 8 && d.log2_hashmap_size <= 24 && d.base_resolution >= 1;
	if (!trunk) return false;
	if (d.sh_degree == 0) return d.rgb_hidden_layers == 0; // NerfNetworkNoDir
	return d.sh_degree == 4 && d.rgb_hidden_layers <= 3;
}

void make_level_table(const nrs_model_desc& d, LevelTable& lt) {
	// tcnn GridEncoding ctor: scale = exp2(l*log2(pls))*base - 1; res = ceil(scale)+1;
	// params_in_level = min(align8(res^3), 2^log2_T).  tcnn evaluates the scale in FLOAT (exp2f(level * log2f(pls)) * base - 1.0f, in the
	// encoding's constructor and in kernel_grid); so do we, with the host libm (upstream NVlabs/tiny-cuda-nn grid.h as recalled, App. B).
	uint32_t off = 0;
	const float l2 = log2f(d.per_level_scale);
	for (uint32_t l = 0; l < d.n_levels; ++l) {
		lt.scale[l] = exp2f((float)l * l2) * (float)d.base_resolution - 1.0f;
		uint32_t res = (uint32_t)ceilf(lt.scale[l]) + 1u;
		lt.resolution[l] = res;
		uint64_t n = (uint64_t)res * res * res;
		uint64_t cap = 1ull << d.log2_hashmap_size;
		n = (n + 7ull) / 8ull * 8ull;
		uint32_t cnt = (uint32_t)std::min<uint64_t>(n, cap);
		lt.count[l] = cnt;
		lt.offset[l] = off;
		// hashed iff the dense stride product exceeds the level's entry count (tcnn grid_index)
		uint64_t stride = 1;
		for (int dim = 0; dim < 3 && stride <= cnt; ++dim) stride *= res;
		lt.hashed[l] = cnt < stride ? 1u : 0u;
		off += cnt;
	}
	lt.total_entries = off;
}

constexpr uint32_t N_DENSITY_W = 64 * 32 + 16 * 64;           // 3072
// rgb network parameters (tiny-cuda-nn's layouts as recalled): FullyFusedMLP with L >= 1 hidden layers = [64 x 32] + (L - 1) [64 x 64] + [16 x 64] (the 3
// outputs padded to 16 rows); CutlassMLP with no hidden layer = one [8 x 32] matrix (outputs padded to the tensor-core width 8); NerfNetworkNoDir: none.
// base.json (L = 2): 7168.
// density network: FullyFusedMLP [64 x 32] + [16 x 64], or -- without hidden layer (configs/nerf/linear.json, CutlassMLP) -- one [16 x 32] matrix
inline uint32_t n_density_weights(const nrs_model_desc& d) { return d.density_hidden_layers == 0 ? 16u * 32u : N_DENSITY_W; }
inline uint32_t n_rgb_weights(const nrs_model_desc& d) {
	if (d.sh_degree == 0) return 0u;
	if (d.rgb_hidden_layers == 0) return 8u * 32u;
	return 64u * 32u + (d.rgb_hidden_layers - 1u) * 64u * 64u + 16u * 64u;
}

// one correctly rounded double -> binary16 conversion: the double is first brought to float with round-to-odd (so that the float -> half
// rounding cannot double-round); hadd = one binary16 addition (the exact sum of two halfs fits a double)
static uint16_t d2h(double d) {
	float f = (float)d;
	if ((double)f == d) return f2h(f);
	float other = ((double)f < d) ? nextafterf(f, INFINITY) : nextafterf(f, -INFINITY);
	return f2h((f2u(f) & 1u) ? f : other);
}
static uint16_t hadd(uint16_t a, uint16_t b) { return d2h((double)h2f(a) + (double)h2f(b)); }

struct Model {
	nrs_model_desc desc;
	// The two places where tiny-cuda-nn's rounding cannot be read off the reference checkout (the submodule is empty), switchable
	// (nrs_model_set_numerics / orc_model_set_numerics):
	//   grid_acc: NRS_GRID_ACC_FP32 (default): the trilinear sum runs in fp32 (fmaf per corner) and is rounded to fp16 once;
	//             NRS_GRID_ACC_NETWORK: kernel_grid as we recall it from NVlabs/tiny-cuda-nn of 2022 (grid.h): per corner
	//             `result[f] += (T)(weight * (float)value[f])` with T = __half -- the fp32 product is rounded to fp16 and ADDED in fp16.
	//   mlp_acc:  NRS_MLP_ACC_FP32 (default): exact products, one rounding per output (what fp32 accumulators give up to their own rounding);
	//             NRS_MLP_ACC_FP16: the fully fused MLP's wmma fragments have fp16 accumulators: modelled as ONE fp16 rounding of the running
	//             sum after every 16-wide k block (acc = fp16(acc + exact sum of 16 products)); how a tensor core sums inside a block is not
	//             specified by NVIDIA, so this is a model of it, not a pin.
	uint32_t grid_acc = 0, mlp_acc = 0;
	// CPU-baseline flavour (orc_model_set_fast; bench.py's cpu_baseline leg ONLY, never a checker): SURVEY 8(d)'s "fp32 math with fp16 rounding points
	// emulated (_Float16 / F16C)" -- hardware half conversions, MLP sums accumulated in fp32 (vectorised) instead of exactly, power-of-two table sizes
	// masked instead of divided.  Same algorithm, same rounding POINTS as the checker; sums may differ from it in the last fp16 bit
	// (tests/test_oracle_kat.py::test_fast_flavour_stays_within_the_network_tolerance).
	bool fast = false;
	LevelTable lt;
	Box aabb;
	std::vector<uint16_t> params; // tcnn order: density | rgb | grid
	std::vector<float> wf;        // MLP weights converted to float once
	std::vector<uint8_t> bitfield;
	bool no_dir() const { return desc.sh_degree == 0; }
	uint32_t n_density_w() const { return n_density_weights(desc); }
	uint32_t n_mlp_w() const { return n_density_w() + n_rgb_weights(desc); }
	const uint16_t* grid() const { return params.data() + n_mlp_w(); }
};

inline uint32_t grid_index(const LevelTable& lt, uint32_t l, uint32_t gx, uint32_t gy, uint32_t gz) {
	const uint32_t res = lt.resolution[l], cnt = lt.count[l];
	uint32_t index;
	if (lt.hashed[l]) {
		index = (gx * 1u) ^ (gy * 2654435761u) ^ (gz * 805459861u);
	} else {
		// stride loop of tcnn grid_index with N_DIMS = 3 (stride <= count holds for all three dims when not hashed)
		index = gx + gy * res + gz * res * res;
	}
	return index % cnt;
}

// one sample -> 32 fp16 features, level-major [l*2+f]
void hashgrid_encode_one(const Model& m, const float pos[3], uint16_t out[32]) {
	const uint16_t* grid = m.grid();
	for (uint32_t l = 0; l < m.desc.n_levels; ++l) {
		const float scale = m.lt.scale[l];
		float p[3], w[3];
		uint32_t g[3];
		for (int d = 0; d < 3; ++d) {
			p[d] = fmaf(scale, pos[d], 0.5f);
			float fl = floorf(p[d]);
			g[d] = (uint32_t)(int)fl;
			w[d] = p[d] - fl;
		}
		float acc0 = 0.f, acc1 = 0.f;
		uint16_t hacc0 = 0, hacc1 = 0;
		for (uint32_t c = 0; c < 8; ++c) {
			float weight = 1.0f;
			uint32_t gl[3];
			for (int d = 0; d < 3; ++d) {
				if ((c & (1u << d)) == 0) { weight *= 1.0f - w[d]; gl[d] = g[d]; }
				else { weight *= w[d]; gl[d] = g[d] + 1u; }
			}
			uint32_t e = m.lt.offset[l] + grid_index(m.lt, l, gl[0], gl[1], gl[2]);
			if (m.grid_acc == NRS_GRID_ACC_NETWORK) { // result[f] += (T)(weight * data), T = __half
				hacc0 = hadd(hacc0, f2h(weight * h2f(grid[2 * (size_t)e + 0])));
				hacc1 = hadd(hacc1, f2h(weight * h2f(grid[2 * (size_t)e + 1])));
			} else {
				acc0 = fmaf(weight, h2f(grid[2 * (size_t)e + 0]), acc0);
				acc1 = fmaf(weight, h2f(grid[2 * (size_t)e + 1]), acc1);
			}
		}
		out[2 * l + 0] = m.grid_acc == NRS_GRID_ACC_NETWORK ? hacc0 : f2h(acc0);
		out[2 * l + 1] = m.grid_acc == NRS_GRID_ACC_NETWORK ? hacc1 : f2h(acc1);
	}
}

// SH degree 4 of the direction given as (d+1)/2 in [0,1]^3 -> 16 fp16 (tcnn SphericalHarmonics encoding)
void sh4_encode_one(const float dir01[3], uint16_t out[16]) {
	float x = dir01[0] * 2.f - 1.f, y = dir01[1] * 2.f - 1.f, z = dir01[2] * 2.f - 1.f;
	float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
	float o[16];
	o[0] = 0.28209479177387814f;
	o[1] = -0.48860251190291987f * y;
	o[2] = 0.48860251190291987f * z;
	o[3] = -0.48860251190291987f * x;
	o[4] = 1.0925484305920792f * xy;
	o[5] = -1.0925484305920792f * yz;
	o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
	o[7] = -1.0925484305920792f * xz;
	o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
	o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
	o[10] = 2.8906114426405538f * xy * z;
	o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
	o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
	o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
	o[14] = 1.4453057213202769f * z * (x2 - y2);
	o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
	for (int i = 0; i < 16; ++i) out[i] = f2h(o[i]);
}

// out[j] = act( sum_k W[j*n_in + k] * in[k] ), fp16 in/out, exact accumulation
inline void dense_layer(const float* W, uint32_t n_out, uint32_t n_in, const uint16_t* in, uint16_t* out, bool relu, uint32_t mlp_acc = 0) {
	float inf[64];
	for (uint32_t k = 0; k < n_in; ++k) inf[k] = h2f(in[k]);
	for (uint32_t j = 0; j < n_out; ++j) {
		const float* w = W + (size_t)j * n_in;
		if (mlp_acc == NRS_MLP_ACC_FP16) { // fp16 accumulator fragments: one rounding of the running sum per 16-wide k block
			uint16_t hacc = 0;
			for (uint32_t kb = 0; kb < n_in; kb += 16) {
				double blk = (double)h2f(hacc);
				for (uint32_t k = kb; k < kb + 16; ++k) blk += (double)w[k] * (double)inf[k];
				hacc = d2h(blk);
			}
			if (relu && !(h2f(hacc) > 0.f)) hacc = 0;
			out[j] = hacc;
			continue;
		}
		double acc = 0.0;
		for (uint32_t k = 0; k < n_in; ++k) acc += (double)w[k] * (double)inf[k];
		float r = (float)acc;
		if (relu && !(r > 0.f)) r = 0.f;
		out[j] = f2h(r);
	}
}

// density MLP 32->64->16 (base.json:30-36).  feat: 32 fp16.  out: 16 fp16.
void density_mlp_one(const Model& m, const uint16_t feat[32], uint16_t out[16]) {
	if (m.desc.density_hidden_layers == 0) { dense_layer(m.wf.data(), 16, 32, feat, out, false, m.mlp_acc); return; } // linear.json: one matrix
	const float* W1 = m.wf.data();
	const float* W2 = W1 + 64 * 32;
	uint16_t h[64];
	dense_layer(W1, 64, 32, feat, h, true, m.mlp_acc);
	dense_layer(W2, 16, 64, h, out, false, m.mlp_acc);
}
// rgb MLP 32->64->64->16 (base.json:52-58); input = [density out 16 | SH 16] (nerf_network_full.h:65-87).  L hidden layers (base_{1,2,3}layer.json), or none:
// one linear map (base_0layer.json's CutlassMLP: 8 padded output rows; the other 8 of `out` are zero).  hidden_out, if given, receives hidden layer `want` (0-based).
void rgb_mlp_one(const Model& m, const uint16_t in32[32], uint16_t out[16], uint16_t* hidden_out = nullptr, uint32_t want = 0) {
	const float* W = m.wf.data() + m.n_density_w();
	const uint32_t L = m.desc.rgb_hidden_layers;
	if (L == 0) {
		dense_layer(W, 8, 32, in32, out, false, m.mlp_acc);
		for (int i = 8; i < 16; ++i) out[i] = 0;
		return;
	}
	uint16_t h[64], hn[64];
	dense_layer(W, 64, 32, in32, h, true, m.mlp_acc);
	W += 64 * 32;
	if (hidden_out && want == 0) memcpy(hidden_out, h, sizeof(h));
	for (uint32_t l = 1; l < L; ++l) {
		dense_layer(W, 64, 64, h, hn, true, m.mlp_acc);
		memcpy(h, hn, sizeof(h));
		W += 64 * 64;
		if (hidden_out && want == l) memcpy(hidden_out, h, sizeof(h));
	}
	dense_layer(W, 16, 64, h, out, false, m.mlp_acc);
}
// ---- the CPU-baseline flavour of the network (Model::fast) ------------------------------------------------------------------------
#if defined(__F16C__)
#include <immintrin.h>
static inline float h2f_hw(uint16_t h) { return _cvtsh_ss(h); }
static inline uint16_t f2h_hw(float f) { return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }
#else
static inline float h2f_hw(uint16_t h) { return h2f(h); }
static inline uint16_t f2h_hw(float f) { return f2h(f); }
#endif
static inline void dense_layer_fast(const float* W, uint32_t n_out, uint32_t n_in, const float* in, float* out, bool relu) {
	for (uint32_t j = 0; j < n_out; ++j) {
		const float* w = W + (size_t)j * n_in;
		float acc = 0.f;
#pragma omp simd reduction(+ : acc)
		for (uint32_t k = 0; k < n_in; ++k) acc += w[k] * in[k];
		if (relu && !(acc > 0.f)) acc = 0.f;
		out[j] = h2f_hw(f2h_hw(acc)); // the fp16 rounding point between layers
	}
}
static void network_inference_fast(const Model& m, const float coord[7], uint16_t out16[16]) {
	const uint16_t* grid = m.grid();
	float in32[32], dout[16], h1[64], h2[64], o[16];
	for (uint32_t l = 0; l < m.desc.n_levels; ++l) { // hash grid: the checker's arithmetic (fmaf per corner, one rounding) with hardware conversions
		const float scale = m.lt.scale[l];
		const uint32_t res = m.lt.resolution[l], cnt = m.lt.count[l], off = m.lt.offset[l];
		const bool pow2 = (cnt & (cnt - 1u)) == 0u;
		float w[3];
		uint32_t g[3];
		for (int d = 0; d < 3; ++d) {
			const float p = fmaf(scale, coord[d], 0.5f), fl = floorf(p);
			g[d] = (uint32_t)(int)fl;
			w[d] = p - fl;
		}
		float acc0 = 0.f, acc1 = 0.f;
		for (uint32_t c = 0; c < 8; ++c) {
			float weight = 1.0f;
			uint32_t gl[3];
			for (int d = 0; d < 3; ++d) {
				if ((c & (1u << d)) == 0) { weight *= 1.0f - w[d]; gl[d] = g[d]; }
				else { weight *= w[d]; gl[d] = g[d] + 1u; }
			}
			uint32_t index = m.lt.hashed[l] ? ((gl[0] * 1u) ^ (gl[1] * 2654435761u) ^ (gl[2] * 805459861u)) : (gl[0] + gl[1] * res + gl[2] * res * res);
			index = pow2 ? (index & (cnt - 1u)) : (index % cnt);
			const uint16_t* e = grid + 2 * (size_t)(off + index);
			acc0 = fmaf(weight, h2f_hw(e[0]), acc0);
			acc1 = fmaf(weight, h2f_hw(e[1]), acc1);
		}
		in32[2 * l] = h2f_hw(f2h_hw(acc0));
		in32[2 * l + 1] = h2f_hw(f2h_hw(acc1));
	}
	const float* W = m.wf.data();
	if (m.desc.density_hidden_layers == 0) {
		dense_layer_fast(W, 16, 32, in32, dout, false);
	} else {
		dense_layer_fast(W, 64, 32, in32, h1, true);
		dense_layer_fast(W + 64 * 32, 16, 64, h1, dout, false);
	}
	uint16_t sh[16];
	sh4_encode_one(coord + 4, sh);
	float rin[32];
	for (int i = 0; i < 16; ++i) { rin[i] = dout[i]; rin[16 + i] = h2f_hw(sh[i]); }
	const float* R = W + m.n_density_w();
	const uint32_t L = m.desc.rgb_hidden_layers;
	for (int i = 0; i < 16; ++i) o[i] = 0.f;
	if (m.no_dir()) { // NerfNetworkNoDir: (r, g, b) = density-network outputs 1..3
		o[0] = dout[1]; o[1] = dout[2]; o[2] = dout[3];
	} else if (L == 0) {
		dense_layer_fast(R, 8, 32, rin, o, false);
	} else {
		dense_layer_fast(R, 64, 32, rin, h1, true);
		R += 64 * 32;
		for (uint32_t l = 1; l < L; ++l) {
			dense_layer_fast(R, 64, 64, h1, h2, true);
			memcpy(h1, h2, sizeof(h1));
			R += 64 * 64;
		}
		dense_layer_fast(R, 16, 64, h1, o, false);
	}
	for (int i = 0; i < 16; ++i) out16[i] = f2h_hw(o[i]);
	out16[3] = f2h_hw(dout[0]);
}

// NerfNetworkFull::inference_mixed_precision_impl, nerf_network_full.h:62-96.  coord: 7 floats. out16: channels
// 0..2 rgb raw, 3 = density raw (extract_density :89-95), 4..15 rgb-net padding outputs.
void network_inference_one(const Model& m, const float coord[7], uint16_t out16[16]) {
	if (m.fast) { network_inference_fast(m, coord, out16); return; }
	uint16_t feat[32], in32[32];
	hashgrid_encode_one(m, coord, feat);
	density_mlp_one(m, feat, in32);          // rows 0..15 of rgb_network_input
	if (m.no_dir()) { // NerfNetworkNoDir::inference_mixed_precision_impl (nerf_network_nodir.h:47-91): the density network's outputs 1..3 are the colour, 0 the density
		for (int i = 0; i < 16; ++i) out16[i] = 0;
		out16[0] = in32[1]; out16[1] = in32[2]; out16[2] = in32[3]; out16[3] = in32[0];
		return;
	}
	sh4_encode_one(coord + 4, in32 + 16);    // dir_offset = 4 (testbed.cu:2328)
	rgb_mlp_one(m, in32, out16);
	out16[3] = in32[0];
}

// ---- the network's introspection entry points render_nerf uses (render modes Normals and EncodingVis) -----------------------------------------
// Both live in tiny-cuda-nn (absent from /root/reference: empty, un-pinned submodule), so like the forward pass they are RESTATED from the published
// algorithm (NVlabs/tiny-cuda-nn of 2022 as recalled: object.h `input_gradient`, network.h `visualize_activation`, grid.h `kernel_grid` /
// `kernel_grid_backward_input`, fully_fused_mlp.cu's backward) around the reference's own NerfNetworkFull::backward_impl / forward_activations
// (nerf_network_full.h:142-221, 523-534).  Parity at this boundary is "HIP == stated tcnn numerics" (DESIGN.md 2).
//
// network.input_gradient(stream, 3, positions, gradients) (tn:2924): d output[3] / d input with a backprop scale of 128 (object.h: "prevents underflows
// during half-precision backprop"), i.e. backward of the one-hot 128 * e_3:
//   * backward_impl copies rows 0..2 of dL_doutput into the rgb network's gradient (all zero here): its input gradients, the SH encoding's and hence
//     rows 4..6 of the result are zero; add_density_gradient puts row 3 on the density network's output row 0: dL_ddensity_out = 128 * e_0;
//   * fully fused backward of the density MLP: dL_dhidden[k] = (hidden[k] > 0) * fp16(W2[0][k] * 128) (one non-zero term: exact), then
//     dL_dfeatures[i] = sum_k W1[k][i] * dL_dhidden[k] in network precision -- accumulated as Model::mlp_acc says (fp32: one rounding; fp16: one
//     rounding of the running sum per 16-wide k block), the same model as the forward pass;
//   * kernel_grid with prepare_input_gradients: dy_dx[level, feature][dim] = sum over the 4 corner pairs along dim of
//     scale * w(other two dims) * (value_right - value_left), fp32, accumulated with the contraction nvcc applies (fmaf), dims 0..2, pairs in index order;
//   * kernel_grid_backward_input: dL_dx[dim] = sum over the 32 features in order of (float)dL_dy[k] * dy_dx[k][dim], fp32 (fmaf);
//   * mult by 1 / 128.
// grad_out: 3 floats (d / d warped position); feat / hidden are the forward activations of the same sample.
void hashgrid_input_gradient_one(const Model& m, const float pos[3], const uint16_t dLdy[32], float grad_out[3]) {
	const uint16_t* grid = m.grid();
	float result[3] = {0.f, 0.f, 0.f};
	for (uint32_t l = 0; l < m.desc.n_levels; ++l) {
		const float scale = m.lt.scale[l];
		float p[3], w[3];
		uint32_t g[3];
		for (int d = 0; d < 3; ++d) {
			p[d] = fmaf(scale, pos[d], 0.5f);
			float fl = floorf(p[d]);
			g[d] = (uint32_t)(int)fl;
			w[d] = p[d] - fl;
		}
		float grads[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
		for (uint32_t grad_dim = 0; grad_dim < 3; ++grad_dim) {
			for (uint32_t idx = 0; idx < 4; ++idx) {
				float weight = scale;
				uint32_t gl[3];
				for (uint32_t non_grad_dim = 0; non_grad_dim < 2; ++non_grad_dim) {
					const uint32_t dim = non_grad_dim >= grad_dim ? non_grad_dim + 1 : non_grad_dim;
					if ((idx & (1u << non_grad_dim)) == 0) { weight *= 1 - w[dim]; gl[dim] = g[dim]; }
					else { weight *= w[dim]; gl[dim] = g[dim] + 1u; }
				}
				gl[grad_dim] = g[grad_dim];
				const uint32_t e_left = m.lt.offset[l] + grid_index(m.lt, l, gl[0], gl[1], gl[2]);
				gl[grad_dim] = g[grad_dim] + 1u;
				const uint32_t e_right = m.lt.offset[l] + grid_index(m.lt, l, gl[0], gl[1], gl[2]);
				for (uint32_t f = 0; f < 2; ++f)
					grads[f][grad_dim] = fmaf(weight, h2f(grid[2 * (size_t)e_right + f]) - h2f(grid[2 * (size_t)e_left + f]), grads[f][grad_dim]); // (* pos_derivative = 1: linear interpolation)
			}
		}
		for (uint32_t f = 0; f < 2; ++f) {
			const float dl = h2f(dLdy[2 * l + f]);
			for (int d = 0; d < 3; ++d) result[d] = fmaf(dl, grads[f][d], result[d]);
		}
	}
	for (int d = 0; d < 3; ++d) grad_out[d] = result[d] * (1.0f / 128.0f);
}
void density_input_gradient_one(const Model& m, const float coord[7], float grad_out[3]) {
	if (m.desc.density_hidden_layers == 0) { // linear.json: dL_dfeatures = W^T (128 e_0) = fp16(W[0][i] * 128), one non-zero term
		uint16_t dfeat[32];
		for (uint32_t i = 0; i < 32; ++i) dfeat[i] = f2h(m.wf[i] * 128.0f);
		hashgrid_input_gradient_one(m, coord, dfeat, grad_out);
		return;
	}
	const float* W1 = m.wf.data();       // [64 x 32]
	const float* W2 = W1 + 64 * 32;      // [16 x 64]
	uint16_t feat[32], h[64];
	hashgrid_encode_one(m, coord, feat);
	dense_layer(W1, 64, 32, feat, h, true, m.mlp_acc);
	uint16_t dh[64];
	for (uint32_t k = 0; k < 64; ++k) dh[k] = h2f(h[k]) > 0.f ? f2h(W2[k] * 128.0f) : (uint16_t)0; // row 0 of W2; x 128 is exact in fp16 up to overflow
	// dL_dfeatures = W1^T dL_dhidden: a dense layer with the transposed matrix
	float W1T[32 * 64];
	for (uint32_t k = 0; k < 64; ++k)
		for (uint32_t i = 0; i < 32; ++i) W1T[i * 64 + k] = W1[k * 32 + i];
	uint16_t dfeat[32];
	dense_layer(W1T, 32, 64, dh, dfeat, false, m.mlp_acc);
	hashgrid_input_gradient_one(m, coord, dfeat, grad_out);
}
// network.visualize_activation(stream, layer, dimension, input, output) (tn:2926): a forward pass, then the activation `dimension` of
// forward_activations(layer) -- 0: the hash-grid output (32), 1: the density MLP's hidden layer (64, after ReLU), 2: the rgb network's input (16 density
// outputs | 16 SH coefficients), 3 / 4: the rgb MLP's hidden layers (64) -- written over the 7-row input matrix by extract_dimension_pos_neg_kernel:
// row 0 = max(-v, 0), row 1 = max(v, 0), row 2 = 0, rows 3..6 = 1.  The reference passes the network INPUT as the output matrix, so the sample's
// NerfCoordinate is overwritten: composite_kernel_nerf then reads pos = (max(-v, 0), max(v, 0), 0) as "warped_pos" (the colour, tn:925), dt = 1
// (unwarp_dt(1) = the largest step: tn:762) and dir = (1, 1, 1).  Restated as it is.
// NerfNetworkFull::width / num_forward_activations (nerf_network_full.h:507-521): layers 0, 1, 2 and one per rgb hidden layer; NerfNetworkNoDir: the grid output
// and the density network's hidden layer (its num_forward_activations counts two more, which its own forward_activations cannot serve: refused).  0 = no such layer.
uint32_t network_layer_width(const nrs_model_desc& d, uint32_t layer) {
	if (layer == 0) return 32u;
	const uint32_t nd = d.density_hidden_layers; // forward activations of the density network (none for linear.json's single matrix)
	if (layer <= nd) return 64u;
	if (d.sh_degree == 0) return 0u;
	if (layer == nd + 1u) return 32u;
	return layer - nd - 2u < d.rgb_hidden_layers ? 64u : 0u;
}
float network_activation_one(const Model& m, const float coord[7], uint32_t layer, uint32_t dim) {
	const float* Wd1 = m.wf.data();
	const float* Wd2 = Wd1 + 64 * 32;
	uint16_t feat[32], h[64], in32[32], h1[64];
	hashgrid_encode_one(m, coord, feat);
	if (layer == 0) return h2f(feat[dim]);
	if (m.desc.density_hidden_layers == 0) {
		density_mlp_one(m, feat, in32);
		layer += 1; // (no density hidden layer: the reference's layer k >= 1 is layer k + 1 of the numbering below)
	} else {
		dense_layer(Wd1, 64, 32, feat, h, true, m.mlp_acc);
		if (layer == 1) return h2f(h[dim]);
		dense_layer(Wd2, 16, 64, h, in32, false, m.mlp_acc);
	}
	sh4_encode_one(coord + 4, in32 + 16);
	if (layer == 2) return h2f(in32[dim]);
	uint16_t out[16];
	rgb_mlp_one(m, in32, out, h1, layer - 3u);
	return h2f(h1[dim]);
}
inline float network_to_density_derivative(float v, uint32_t act); // below

inline float logistic(float x) { return 1.0f / (1.0f + expf(-x)); }
inline float network_to_rgb(float v, uint32_t act) { // cn:38-47
	switch (act) {
		case NRS_ACT_NONE: return v;
		case NRS_ACT_RELU: return v > 0.f ? v : 0.f;
		case NRS_ACT_LOGISTIC: return logistic(v);
		case NRS_ACT_EXPONENTIAL: return expf(clampf(v, -10.f, 10.f));
	}
	return 0.f;
}
inline float network_to_density(float v, uint32_t act) { // cn:57-66
	switch (act) {
		case NRS_ACT_NONE: return v;
		case NRS_ACT_RELU: return v > 0.f ? v : 0.f;
		case NRS_ACT_LOGISTIC: return logistic(v);
		case NRS_ACT_EXPONENTIAL: return expf(v);
	}
	return 0.f;
}

inline float network_to_density_derivative(float v, uint32_t act) { // tn:308-317
	switch (act) {
		case NRS_ACT_NONE: return 1.0f;
		case NRS_ACT_RELU: return v > 0.0f ? 1.0f : 0.0f;
		case NRS_ACT_LOGISTIC: { float density = logistic(v); return density * (1 - density); }
		case NRS_ACT_EXPONENTIAL: return expf(clampf(v, -15.0f, 15.0f));
	}
	return 0.0f;
}

// ------------------------------------------------------------------------------------------------
// Cage / tet warp: selection_utils.h:10-47, cage_deformation.cu:136-269, 431-541
// ------------------------------------------------------------------------------------------------
inline float scalar_tp(V3 a, V3 b, V3 c) { return dot(a, cross(b, c)); }
inline void bary_tet(V3 a, V3 b, V3 c, V3 d, V3 p, float out[4]) { // selection_utils.h:14-31
	V3 vap = p - a, vbp = p - b, vab = b - a, vac = c - a, vad = d - a, vbc = c - b, vbd = d - b;
	float va6 = scalar_tp(vbp, vbd, vbc);
	float vb6 = scalar_tp(vap, vac, vad);
	float vc6 = scalar_tp(vap, vad, vab);
	float vd6 = scalar_tp(vap, vab, vac);
	float v6 = (float)(1. / (double)scalar_tp(vab, vac, vad)); // "1. / float" is a double division rounded to float
	out[0] = va6 * v6; out[1] = vb6 * v6; out[2] = vc6 * v6; out[3] = vd6 * v6;
}
inline bool same_side_tet(V3 v1, V3 v2, V3 v3_, V3 v4, V3 p) { // :33-39
	V3 normal = cross(v2 - v1, v3_ - v1);
	float dotV4 = dot(normal, v4 - v1);
	float dotP = dot(normal, p - v1);
	return std::signbit(dotV4) == std::signbit(dotP);
}
inline bool point_in_tet(V3 v1, V3 v2, V3 v3_, V3 v4, V3 p) { // :41-47
	return same_side_tet(v1, v2, v3_, v4, p) && same_side_tet(v2, v3_, v4, v1, p) &&
	       same_side_tet(v3_, v4, v1, v2, p) && same_side_tet(v4, v1, v2, v3_, p);
}

// AffineBoundingBox after warp_box (affine_bounding_box.cuh:83-101): what translate_in_box tests
struct ABox { V3 mn, u, v, w, center; float uu, vv, ww; };
struct Edit {
	int kind = 0;                               // 0 = CageDeformation, 1 = AffineDuplication
	ABox a_dst, a_sel;                          // m_warped_destination_box, m_warped_selection_box
	V3 a_translation, a_scale;                  // m_warped_translation, m_scale
	float a_rot[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; // m_rotation_matrix, column-major
	bool a_hide_original = false, a_correct_dir = true;
	Box aabb;                                   // scene aabb (m_scene_aabb)
	Box bbox, warped_bbox, orig_bbox, orig_warped_bbox; // tet_mesh.cu:12-20, tet_mesh.h:100-107
	std::vector<V3> verts, orig;
	std::vector<uint32_t> tets, lut_off, lut_idx;
	std::vector<uint8_t> orig_bitfield;
	std::vector<float> rot; // [T*9] col-major or empty
	bool copy = false;
	bool apply_poisson = false;
	float residual_amplitude = 1.f;
	std::vector<float> shs, out_density, res_density;
};

inline V3 ldv(const std::vector<V3>& a, uint32_t i) { return a[i]; }

Box bbox_of(const std::vector<V3>& v) {
	const float inf = std::numeric_limits<float>::infinity();
	Box b{v3(inf, inf, inf), v3(-inf, -inf, -inf)};
	for (const V3& p : v) {
		b.mn = v3(fminf(b.mn.x, p.x), fminf(b.mn.y, p.y), fminf(b.mn.z, p.z));
		b.mx = v3(fmaxf(b.mx.x, p.x), fmaxf(b.mx.y, p.y), fmaxf(b.mx.z, p.z));
	}
	return b;
}
Box warp_box(const Box& b, const Box& aabb) { return Box{warp_position(b.mn, aabb), warp_position(b.mx, aabb)}; } // bounding_box.cuh:272

// AffineDuplication: translate_in_box / translate_in_box_pos, affine_duplication.cu:69-118
inline bool abox_contains(const ABox& b, V3 p) { // affine_bounding_box.cuh:83-88
	V3 q = p - b.mn;
	float du = dot(b.u, q), dv = dot(b.v, q), dw = dot(b.w, q);
	return du >= 0.f && du < b.uu && dv >= 0.f && dv < b.vv && dw >= 0.f && dw < b.ww;
}
inline V3 mul_rt(const float* R, V3 q) { // R^T q, R column-major
	return {sum3(R[0] * q.x, R[1] * q.y, R[2] * q.z), sum3(R[3] * q.x, R[4] * q.y, R[5] * q.z), sum3(R[6] * q.x, R[7] * q.y, R[8] * q.z)};
}
static void affine_map_one(const Edit& e, float* pos, float* dir /* nullable */, uint8_t* empty) {
	V3 p = {pos[0], pos[1], pos[2]};
	if (abox_contains(e.a_dst, p)) {
		V3 d = p - e.a_dst.center;
		V3 q = {d.x / e.a_scale.x, d.y / e.a_scale.y, d.z / e.a_scale.z};
		V3 r = mul_rt(e.a_rot, q) + e.a_dst.center;
		r = r - e.a_translation;
		pos[0] = r.x; pos[1] = r.y; pos[2] = r.z;
		if (dir && e.a_correct_dir) {
			V3 wd = warp_direction(mul_rt(e.a_rot, unwarp_direction(v3(dir[0], dir[1], dir[2]))));
			dir[0] = wd.x; dir[1] = wd.y; dir[2] = wd.z;
		}
	} else if (e.a_hide_original && abox_contains(e.a_sel, p)) {
		*empty = 1;
	}
}

// interpolate_tet, cage_deformation.cu:197-269.  coord: 7 floats, in place.  returns through *empty.
void map_ray_one(const Edit& e, float coord[7], uint8_t* empty) {
	if (e.kind == 1) { affine_map_one(e, coord, coord + 4, empty); return; }
	V3 p = {coord[0], coord[1], coord[2]};
	bool in_deformed = false;
	if (box_contains(e.warped_bbox, p)) {
		V3 u = unwarp_position(p, e.aabb);
		int level = mip_from_pos(u);
		uint32_t cell = (uint32_t)level * GRIDVOL + cascaded_grid_idx_at(u, (uint32_t)level);
		for (uint32_t j = e.lut_off[cell]; j < e.lut_off[cell + 1]; ++j) {
			uint32_t t = e.lut_idx[j];
			V3 a = e.verts[e.tets[4 * t]], b = e.verts[e.tets[4 * t + 1]], c = e.verts[e.tets[4 * t + 2]], d = e.verts[e.tets[4 * t + 3]];
			if (point_in_tet(a, b, c, d, u)) {
				float bc[4];
				bary_tet(a, b, c, d, u, bc);
				V3 o0 = e.orig[e.tets[4 * t]], o1 = e.orig[e.tets[4 * t + 1]], o2 = e.orig[e.tets[4 * t + 2]], o3 = e.orig[e.tets[4 * t + 3]];
				V3 canon = ((bc[0] * o0 + bc[1] * o1) + bc[2] * o2) + bc[3] * o3;
				V3 wp = warp_position(canon, e.aabb);
				coord[0] = wp.x; coord[1] = wp.y; coord[2] = wp.z;
				if (!e.rot.empty()) {
					V3 ud = unwarp_direction(v3(coord[4], coord[5], coord[6]));
					const float* R = &e.rot[9 * (size_t)t]; // column-major
					V3 rd = mat3_mul(R, ud);
					V3 wd = warp_direction(rd);
					coord[4] = wd.x; coord[5] = wd.y; coord[6] = wd.z;
				}
				in_deformed = true;
				break;
			}
		}
	}
	if (!e.copy) {
		V3 q = {coord[0], coord[1], coord[2]};
		if (!in_deformed && box_contains(e.orig_warped_bbox, q)) {
			V3 u = unwarp_position(q, e.aabb);
			int level = mip_from_pos(u);
			uint32_t pos_idx = cascaded_grid_idx_at(u, (uint32_t)level);
			if (get_bitfield_at(pos_idx, (uint32_t)level, e.orig_bitfield.data())) *empty = 1;
		}
	}
}

// interpolate_tet_pos, cage_deformation.cu:136-192 (no direction, no copy flag)
void map_position_one(const Edit& e, float pos[3], uint8_t* empty) {
	if (e.kind == 1) { affine_map_one(e, pos, nullptr, empty); return; }
	V3 p = {pos[0], pos[1], pos[2]};
	bool in_deformed = false;
	if (box_contains(e.warped_bbox, p)) {
		V3 u = unwarp_position(p, e.aabb);
		int level = mip_from_pos(u);
		uint32_t cell = (uint32_t)level * GRIDVOL + cascaded_grid_idx_at(u, (uint32_t)level);
		for (uint32_t j = e.lut_off[cell]; j < e.lut_off[cell + 1]; ++j) {
			uint32_t t = e.lut_idx[j];
			V3 a = e.verts[e.tets[4 * t]], b = e.verts[e.tets[4 * t + 1]], c = e.verts[e.tets[4 * t + 2]], d = e.verts[e.tets[4 * t + 3]];
			if (point_in_tet(a, b, c, d, u)) {
				float bc[4];
				bary_tet(a, b, c, d, u, bc);
				V3 o0 = e.orig[e.tets[4 * t]], o1 = e.orig[e.tets[4 * t + 1]], o2 = e.orig[e.tets[4 * t + 2]], o3 = e.orig[e.tets[4 * t + 3]];
				V3 canon = ((bc[0] * o0 + bc[1] * o1) + bc[2] * o2) + bc[3] * o3;
				V3 wp = warp_position(canon, e.aabb);
				pos[0] = wp.x; pos[1] = wp.y; pos[2] = wp.z;
				in_deformed = true;
				break;
			}
		}
	}
	V3 q = {pos[0], pos[1], pos[2]};
	if (!in_deformed && box_contains(e.orig_warped_bbox, q)) {
		V3 u = unwarp_position(q, e.aabb);
		int level = mip_from_pos(u);
		uint32_t pos_idx = cascaded_grid_idx_at(u, (uint32_t)level);
		if (get_bitfield_at(pos_idx, (uint32_t)level, e.orig_bitfield.data())) *empty = 1;
	}
}

// compute_residual_poisson_kernel body for one sample, cage_deformation.cu:467-507
void poisson_residual_one(const Edit& e, const float coord[7], float sh_out[27], float* out_density, float* res_density) {
	V3 pos = unwarp_position(v3(coord[0], coord[1], coord[2]), e.aabb);
	if (!box_contains(e.bbox, pos)) return;
	int level = mip_from_pos(pos);
	uint32_t cell = (uint32_t)level * GRIDVOL + cascaded_grid_idx_at(pos, (uint32_t)level);
	for (uint32_t j = e.lut_off[cell]; j < e.lut_off[cell + 1]; ++j) {
		uint32_t t = e.lut_idx[j];
		const uint32_t* tv = &e.tets[4 * (size_t)t];
		V3 a = e.verts[tv[0]], b = e.verts[tv[1]], c = e.verts[tv[2]], d = e.verts[tv[3]];
		if (point_in_tet(a, b, c, d, pos)) {
			float bc[4];
			bary_tet(a, b, c, d, pos, bc);
			for (int k = 0; k < 27; ++k)
				sh_out[k] = ((bc[0] * e.shs[27 * (size_t)tv[0] + k] + bc[1] * e.shs[27 * (size_t)tv[1] + k]) +
				             bc[2] * e.shs[27 * (size_t)tv[2] + k]) + bc[3] * e.shs[27 * (size_t)tv[3] + k];
			float lo = ((bc[0] * e.out_density[tv[0]] + bc[1] * e.out_density[tv[1]]) + bc[2] * e.out_density[tv[2]]) + bc[3] * e.out_density[tv[3]];
			float lr = ((bc[0] * e.res_density[tv[0]] + bc[1] * e.res_density[tv[1]]) + bc[2] * e.res_density[tv[2]]) + bc[3] * e.res_density[tv[3]];
			*out_density = e.residual_amplitude * lo;
			*res_density = e.residual_amplitude * lr;
			break;
		}
	}
}

// evaluate_sh9 (SH9RGB = 9x3 column-major), cn:218-245
void evaluate_sh9(const float sh[27], V3 dir, float rgb[3]) {
	float fC0, fC1, fS0, fS1, fTmpA, fTmpB, fTmpC;
	float fZ2 = dir.z * dir.z;
	float pSH[9];
	pSH[0] = 0.2820947917738781f;
	pSH[2] = 0.4886025119029199f * dir.z;
	pSH[6] = 0.9461746957575601f * fZ2 + -0.3153915652525201f;
	fC0 = dir.x; fS0 = dir.y;
	fTmpA = -0.48860251190292f;
	pSH[3] = fTmpA * fC0; pSH[1] = fTmpA * fS0;
	fTmpB = -1.092548430592079f * dir.z;
	pSH[7] = fTmpB * fC0; pSH[5] = fTmpB * fS0;
	fC1 = dir.x * fC0 - dir.y * fS0;
	fS1 = dir.x * fS0 + dir.y * fC0;
	fTmpC = 0.5462742152960395f;
	pSH[8] = fTmpC * fC1; pSH[4] = fTmpC * fS1;
	for (int c = 0; c < 3; ++c) { // pSH.dot(sh.block<9, 1>(0, c)): 9 terms in Eigen's unrolled reduction order (Redux.h), 4 | 5 -> 2|2 | 2|(1|2)
		const float* q = sh + 9 * c;
		rgb[c] = ((pSH[0] * q[0] + pSH[1] * q[1]) + (pSH[2] * q[2] + pSH[3] * q[3])) + ((pSH[4] * q[4] + pSH[5] * q[5]) + (pSH[6] * q[6] + (pSH[7] * q[7] + pSH[8] * q[8])));
	}
}

inline float srgb_to_linear(float s) { // common_device.cuh:31-37
	if (s <= 0.04045f) return s / 12.92f;
	return powf((s + 0.055f) / 1.055f, 2.4f);
}

// ------------------------------------------------------------------------------------------------
// The render loop: Testbed::render_nerf tn:3066 = init_rays_from_camera tn:2683 + trace tn:2772 + shade tn:2448
// ------------------------------------------------------------------------------------------------
struct RayState {
	Payload payload;
	float rgba[4];
	float depth;
	bool in_hit_list;
};

struct RenderStats { uint64_t generated, composited; uint32_t n_alive0, n_hit, iterations; };

void render(const Model& m, const nrs_render_params& p, const Edit* const* edits, int n_edits,
            float* frame, float* depth_buf, uint32_t* steps_buf, RenderStats* stats, int fixed_S) {
	const uint32_t W = (uint32_t)p.resolution[0], H = (uint32_t)p.resolution[1];
	const uint32_t N = W * H;
	const Box render_aabb{v3(p.render_aabb_min[0], p.render_aabb_min[1], p.render_aabb_min[2]),
	                      v3(p.render_aabb_max[0], p.render_aabb_max[1], p.render_aabb_max[2])};
	const Box train_aabb = m.aabb;
	const uint8_t* grid = m.bitfield.data();
	const bool ops = p.apply_operators && n_edits > 0;
	const uint32_t march_iter = p.max_march_steps ? p.max_march_steps : MARCH_ITER;

	const int show_accel = p.show_accel ? (int)p.min_mip : -1; // m_nerf.show_accel; min_mip = (show_accel >= 0) ? show_accel : 0, tn:2751, :2849
	const uint32_t render_mode = p.render_mode;

	std::vector<RayState> rays(N);
	// tiles: a pixel belongs to this call iff its tile index matches (tile_first, tile_stride)
	auto owned = [&](uint32_t x, uint32_t y) -> bool {
		if (p.tile_size == 0) return true;
		uint32_t tiles_x = ((W + p.tile_size - 1) / p.tile_size) | 1u; // the odd row pitch of the tile index (nrs.h)
		uint32_t t = (y / p.tile_size) * tiles_x + (x / p.tile_size);
		uint32_t stride = p.tile_stride ? p.tile_stride : 1;
		return t >= p.tile_first && (t - p.tile_first) % stride == 0;
	};

#pragma omp parallel for schedule(dynamic, 256)
	for (int64_t i = 0; i < (int64_t)N; ++i) {
		uint32_t x = (uint32_t)i % W, y = (uint32_t)i / W;
		RayState& r = rays[i];
		memset(&r, 0, sizeof(r));
		if (!owned(x, y)) { r.payload.alive = false; r.payload.idx = (uint32_t)i; continue; }
		float d0;
		init_ray(p, x, y, r.payload, d0, frame + 4 * (size_t)i);
		depth_buf[i] = d0;
		advance_pos(p, grid, r.payload, (uint32_t)i);
		if (steps_buf) steps_buf[i] = 0;
	}

	if (render_mode == NRS_RENDER_SLICE) {
		// tn:3111-3175: every initialised ray stands on the slice plane (init_ray); one network evaluation there
		// (generate_nerf_network_inputs_at_current_position tn:616-622 -> NerfNetwork::inference -> compute_nerf_density tn:624-635) -> shade
		uint64_t n_px = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : n_px)
		for (int64_t i = 0; i < (int64_t)N; ++i) {
			uint32_t x = (uint32_t)i % W, y = (uint32_t)i / W;
			if (!owned(x, y)) continue;
			const Payload& pl = rays[i].payload;
			V3 wp = warp_position(pl.origin + pl.dir * pl.t, train_aabb), wd = warp_direction(pl.dir);
			float c7[7] = {wp.x, wp.y, wp.z, warp_dt(MIN_STEP), wd.x, wd.y, wd.z};
			uint16_t out[16];
			network_inference_one(m, c7, out);
			// network.inference(): the fp16 outputs as floats; then compute_nerf_density
			float a = clampf(1.f - expf(-network_to_density(h2f(out[3]), m.desc.density_activation) / 100.0f), 0.0f, 1.0f);
			float tmp[4] = {network_to_rgb(h2f(out[0]), m.desc.rgb_activation) * a, network_to_rgb(h2f(out[1]), m.desc.rgb_activation) * a,
			                network_to_rgb(h2f(out[2]), m.desc.rgb_activation) * a, a};
			if (!p.linear_colors) { tmp[0] = srgb_to_linear(tmp[0]); tmp[1] = srgb_to_linear(tmp[1]); tmp[2] = srgb_to_linear(tmp[2]); } // tn:2474-2477
			float* f = frame + 4 * (size_t)pl.idx;
			float one_minus = 1.0f - tmp[3];
			for (int c = 0; c < 4; ++c) f[c] = tmp[c] + f[c] * one_minus;
			++n_px; // (no depth write in Slice mode, tn:2480: the depth buffer keeps -plane_z from init_rays)
		}
		if (stats) { stats->generated = n_px; stats->composited = n_px; stats->n_alive0 = (uint32_t)n_px; stats->n_hit = (uint32_t)n_px; stats->iterations = 0; }
		return;
	}

	std::vector<uint32_t> alive, next_alive;
	alive.reserve(N);
	for (uint32_t i = 0; i < N; ++i) alive.push_back(i);
	uint32_t n_rays_initialized = N; // m_n_rays_initialized = res.x*res.y, tn:2739
	// (with tiling the reference has no counterpart; S then differs but per-pixel results do not)
	uint64_t generated = 0, composited = 0;
	uint32_t n_hit = 0, n_alive0 = 0, iterations = 0;
	std::vector<uint32_t> hit;

	uint32_t i_step = 1;
	bool first = true;
	while (i_step < march_iter) { // tn:2812
		// compact_kernel_nerf, tn:2485-2510 (order is irrelevant per pixel; we keep index order)
		next_alive.clear();
		for (uint32_t ri : alive) {
			RayState& r = rays[ri];
			if (r.payload.alive) next_alive.push_back(ri);
			else if (r.rgba[3] > 0.001f) hit.push_back(ri);
		}
		alive.swap(next_alive);
		const uint32_t n_alive = (uint32_t)alive.size();
		if (first) { n_alive0 = n_alive; first = false; }
		if (n_alive == 0) break;
		uint32_t S = std::min(std::max(n_rays_initialized / n_alive, MIN_STEPS_INBETWEEN_COMPACTION), MAX_STEPS_INBETWEEN_COMPACTION); // tn:2835
		if (fixed_S > 0) S = (uint32_t)fixed_S;
		++iterations;

		uint64_t gen_local = 0, comp_local = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : gen_local, comp_local)
		for (int64_t ai = 0; ai < (int64_t)n_alive; ++ai) {
			RayState& r = rays[alive[ai]];
			Payload& payload = r.payload;
			float coords[8][7];
			uint8_t empty[8] = {0};
			float sh_b[8][27];
			float dens_out_b[8], dens_res_b[8];

			// ---- generate_next_nerf_network_inputs, tn:637-696
			{
				V3 origin = payload.origin, dir = payload.dir;
				V3 idir = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
				float cone_angle = p.cone_angle_constant;
				float t = payload.t;
				uint32_t j = 0;
				bool exited = false;
				for (; j < S; ++j) {
					V3 pos;
					float dt = 0.0f;
					while (1) {
						pos = origin + dir * t;
						if (!box_contains(render_aabb, pos)) { exited = true; break; }
						dt = calc_dt(t, cone_angle);
						uint32_t mip = std::max(p.min_mip, (uint32_t)mip_from_dt(dt, pos));
						if (density_grid_occupied_at(pos, grid, mip)) break;
						uint32_t res = GRID >> mip;
						t = advance_to_next_voxel(t, cone_angle, pos, dir, idir, res);
					}
					if (exited) break;
					V3 wp = warp_position(pos, train_aabb);
					V3 wd = warp_direction(dir);
					coords[j][0] = wp.x; coords[j][1] = wp.y; coords[j][2] = wp.z;
					coords[j][3] = warp_dt(dt);
					coords[j][4] = wd.x; coords[j][5] = wd.y; coords[j][6] = wd.z;
					t += dt;
				}
				payload.n_steps = j;
				if (!exited) payload.t = t; // tn:675-677: on exit payload.t is NOT updated
			}
			const uint32_t actual_n_steps = payload.n_steps;
			gen_local += actual_n_steps;

			// ---- membrane residuals in deformed space, tn:2863-2883 (memset 0, then every operator last-to-first)
			for (uint32_t j = 0; j < actual_n_steps; ++j) {
				memset(sh_b[j], 0, sizeof(sh_b[j]));
				dens_out_b[j] = 0.f; dens_res_b[j] = 0.f;
			}
			if (ops)
				for (int oi = n_edits - 1; oi >= 0; --oi)
					if (edits[oi]->apply_poisson)
						for (uint32_t j = 0; j < actual_n_steps; ++j)
							poisson_residual_one(*edits[oi], coords[j], sh_b[j], &dens_out_b[j], &dens_res_b[j]);

			// ---- first network pass on un-deformed coordinates, tn:2890-2892.  The reference runs it
			// unconditionally; its result is consumed only where density_out_boundary > 1e-9 (tn:770-773).
			uint16_t out_old[8][16];
			for (uint32_t j = 0; j < actual_n_steps; ++j)
				if (dens_out_b[j] > 1e-9f) network_inference_one(m, coords[j], out_old[j]);

			// ---- map_rays, last-to-first, in place, tn:2896-2904
			if (ops)
				for (int oi = n_edits - 1; oi >= 0; --oi)
					for (uint32_t j = 0; j < actual_n_steps; ++j) map_ray_one(*edits[oi], coords[j], &empty[j]);

			// ---- second network pass, tn:2908-2913
			uint16_t out[8][16];
			for (uint32_t j = 0; j < actual_n_steps; ++j) network_inference_one(m, coords[j], out[j]);
			// ---- tn:2923-2927: the input gradient of the density (render mode Normals), or the visualised activation written OVER the network input
			float grad_pos[8][3];
			if (render_mode == NRS_RENDER_NORMALS) {
				for (uint32_t j = 0; j < actual_n_steps; ++j) density_input_gradient_one(m, coords[j], grad_pos[j]);
			} else if (render_mode == NRS_RENDER_ENCODING_VIS) {
				for (uint32_t j = 0; j < actual_n_steps; ++j) {
					const float v = network_activation_one(m, coords[j], p.visualized_layer, p.visualized_dimension);
					coords[j][0] = fmaxf(-v, 0.0f); coords[j][1] = fmaxf(v, 0.0f); coords[j][2] = 0.f;
					coords[j][3] = coords[j][4] = coords[j][5] = coords[j][6] = 1.f;
				}
			}

			// ---- composite_kernel_nerf, tn:698-979 (Shade mode)
			{
				float lr = r.rgba[0], lg = r.rgba[1], lb = r.rgba[2], la = r.rgba[3];
				float local_depth = r.depth;
				V3 cam_fwd = cam_col(p.camera_matrix1, 2), cam_o = cam_col(p.camera_matrix1, 3);
				uint32_t j = 0;
				for (; j < actual_n_steps; ++j) {
					V3 pos = unwarp_position(v3(coords[j][0], coords[j][1], coords[j][2]), train_aabb);
					float T = 1.f - la;
					float dt = unwarp_dt(coords[j][3]);
					float alpha;
					const bool has_res = dens_out_b[j] > 1e-9f;
					float sigma_raw = h2f(out[j][3]);
					if (ops && empty[j]) {
						alpha = 0.0f;
					} else if (has_res) {
						float sourceval = network_to_density(sigma_raw, m.desc.density_activation);
						float targetval = network_to_density(h2f(out_old[j][3]), m.desc.density_activation);
						float val = p.poisson_target ? fminf(fmaxf(targetval, sourceval), sourceval + dens_res_b[j]) : sourceval + dens_res_b[j];
						alpha = 1.f - expf(-(val) * dt);
					} else {
						alpha = 1.f - expf(-network_to_density(sigma_raw, m.desc.density_activation) * dt);
					}
					if (show_accel >= 0) alpha = 1.f; // tn:788-790
					float weight = alpha * T;
					float rgb[3] = {network_to_rgb(h2f(out[j][0]), m.desc.rgb_activation), network_to_rgb(h2f(out[j][1]), m.desc.rgb_activation),
					                network_to_rgb(h2f(out[j][2]), m.desc.rgb_activation)};
					// ---- the glow overlay ("random grid visualizations"), tn:806-903: adds to / replaces the colour, may scale the weight
					if (p.glow_mode) {
						const uint32_t gm = p.glow_mode;
						const bool green_grid = gm & 1, green_cutline = gm & 2, mask_to_alpha = gm & 4, radial_mode = gm & 8, grid_mode = gm & 16;
						float glow = 0.f;
						float dist;
						if (radial_mode) {
							V3 dv = pos - cam_o;
							dist = sqrtf(dot(dv, dv));
							dist = std::min(dist, (4.5f - pos.y) * 0.333f);
						} else {
							dist = pos.y;
						}
						if (grid_mode) {
							glow = 1.f / std::max(1.f, dist);
						} else {
							float y = p.glow_y_cutoff - dist;
							float mask = 0.f;
							if (y > 0.f) {
								y *= 80.f;
								mask = std::min(1.f, y);
								if (green_cutline) glow += std::max(0.f, 1.f - fabsf(1.f - y)) * 4.f;
								if (y > 1.f) y = 1.f - (y - 1.f) * 0.05f;
								if (green_grid) glow += std::max(0.f, y / std::max(1.f, dist));
							}
							if (mask_to_alpha) weight *= mask;
						}
						if (glow > 0.f) {
							const float PI = 3.141592653589793f;
							float line = std::max(0.f, cosf(pos.y * 2.f * PI * 16.f) - 0.975f);
							line += std::max(0.f, cosf(pos.x * 2.f * PI * 16.f) - 0.975f);
							line += std::max(0.f, cosf(pos.z * 2.f * PI * 16.f) - 0.975f);
							line += std::max(0.f, cosf(pos.y * 4.f * PI * 16.f) - 0.975f);
							line += std::max(0.f, cosf(pos.x * 4.f * PI * 16.f) - 0.975f);
							line += std::max(0.f, cosf(pos.z * 4.f * PI * 16.f) - 0.975f);
							line += std::max(0.f, cosf(pos.y * 8.f * PI * 16.f) - 0.975f);
							line += std::max(0.f, cosf(pos.x * 8.f * PI * 16.f) - 0.975f);
							line += std::max(0.f, cosf(pos.z * 8.f * PI * 16.f) - 0.975f);
							line += std::max(0.f, cosf(pos.y * 16.f * PI * 16.f) - 0.975f);
							line += std::max(0.f, cosf(pos.x * 16.f * PI * 16.f) - 0.975f);
							line += std::max(0.f, cosf(pos.z * 16.f * PI * 16.f) - 0.975f);
							if (grid_mode) {
								glow = glow * line * 15.f;
								rgb[1] = glow; rgb[2] = glow * 0.5f; rgb[0] = glow * 0.25f;
							} else {
								glow = glow * glow * 0.25f + glow * line * 15.f;
								rgb[1] += glow; rgb[2] += glow * 0.5f; rgb[0] += glow * 0.25f;
							}
						}
					}
					// ---- per-sample render modes, tn:905-937
					if (render_mode == NRS_RENDER_NORMALS) { // tn:905-910: the direction of decreasing density
						const float k = -network_to_density_derivative(sigma_raw, m.desc.density_activation);
						const float nx = k * grad_pos[j][0], ny = k * grad_pos[j][1], nz = k * grad_pos[j][2];
						const float z = nx * nx + (ny * ny + nz * nz); // Eigen: squaredNorm of a fixed-size 3-vector, then normalized() (z > 0 ? v / sqrt(z) : v)
						if (z > 0.f) { const float n = sqrtf(z); rgb[0] = nx / n; rgb[1] = ny / n; rgb[2] = nz / n; }
						else { rgb[0] = nx; rgb[1] = ny; rgb[2] = nz; }
					} else if (render_mode == NRS_RENDER_ENCODING_VIS) { // tn:925: rgb = warped_pos (the overwritten input)
						rgb[0] = coords[j][0]; rgb[1] = coords[j][1]; rgb[2] = coords[j][2];
					} else if (render_mode == NRS_RENDER_POSITIONS) {
						if (show_accel >= 0) {
							uint32_t mip = (uint32_t)std::max(show_accel, mip_from_pos(pos));
							uint32_t res = GRID >> mip;
							int ix = (int)(pos.x * (float)res), iy = (int)(pos.y * (float)res), iz = (int)(pos.z * (float)res);
							Pcg32 rng = pcg32_seeded((uint64_t)(int64_t)(int)((uint32_t)ix + (uint32_t)iy * 232323u + (uint32_t)iz * 727272u)); // int arithmetic, sign-extended
							rgb[0] = 1.f - (float)mip * (1.f / (CASCADES - 1));
							rgb[1] = rng.next_float();
							rgb[2] = rng.next_float();
						} else {
							rgb[0] = (pos.x - 0.5f) / 2.0f + 0.5f; rgb[1] = (pos.y - 0.5f) / 2.0f + 0.5f; rgb[2] = (pos.z - 0.5f) / 2.0f + 0.5f;
						}
					} else if (render_mode == NRS_RENDER_DEPTH) {
						float z = dot(cam_fwd, pos - payload.origin) * p.depth_scale;
						rgb[0] = rgb[1] = rgb[2] = z;
					} else if (render_mode == NRS_RENDER_DISTANCE) {
						V3 dv = pos - payload.origin;
						float z = sqrtf(dot(dv, dv)) * p.depth_scale;
						rgb[0] = rgb[1] = rgb[2] = z;
					} else if (render_mode == NRS_RENDER_STEPSIZE) {
						float warped_dt = warp_dt(dt);
						rgb[0] = rgb[1] = rgb[2] = warped_dt;
					} else if (render_mode == NRS_RENDER_AO) {
						rgb[0] = rgb[1] = rgb[2] = alpha;
					}
					if (has_res) {
						float alpha_N = 1.f - expf(-network_to_density(sigma_raw, m.desc.density_activation) * dt);
						float alpha_R = 1.f - expf(-dens_out_b[j] * dt);
						float w_N = alpha_N / (alpha_N + alpha_R), w_R = alpha_R / (alpha_N + alpha_R);
						float res_rgb[3];
						evaluate_sh9(sh_b[j], unwarp_direction(v3(coords[j][4], coords[j][5], coords[j][6])), res_rgb);
						lr += weight * (w_N * rgb[0] + w_R * res_rgb[0]);
						lg += weight * (w_N * rgb[1] + w_R * res_rgb[1]);
						lb += weight * (w_N * rgb[2] + w_R * res_rgb[2]);
					} else {
						lr += rgb[0] * weight; lg += rgb[1] * weight; lb += rgb[2] * weight;
					}
					la += weight;
					if (weight > payload.max_weight) {
						payload.max_weight = weight;
						local_depth = dot(cam_fwd, pos - cam_o);
					}
					++comp_local;
					if (steps_buf) steps_buf[payload.idx] += 1;
					if (la > (1.0f - p.min_transmittance)) {
						lr /= la; lg /= la; lb /= la; la /= la;
						break;
					}
				}
				if (j < S) { // tn:957-960
					payload.alive = false;
					payload.n_steps = j + i_step;
				}
				r.rgba[0] = lr; r.rgba[1] = lg; r.rgba[2] = lb; r.rgba[3] = la;
				r.depth = local_depth;
			}
		}
		generated += gen_local;
		composited += comp_local;
		i_step += S; // tn:2989
	}

	// shade_kernel_nerf, tn:2448-2483
	n_hit = (uint32_t)hit.size();
	for (uint32_t ri : hit) {
		RayState& r = rays[ri];
		float tmp[4] = {r.rgba[0], r.rgba[1], r.rgba[2], r.rgba[3]};
		if (p.render_mode == NRS_RENDER_NORMALS) { // tn:2466-2468
			const float z = tmp[0] * tmp[0] + (tmp[1] * tmp[1] + tmp[2] * tmp[2]);
			float n[3] = {tmp[0], tmp[1], tmp[2]};
			if (z > 0.f) { const float s = sqrtf(z); n[0] = tmp[0] / s; n[1] = tmp[1] / s; n[2] = tmp[2] / s; }
			for (int c = 0; c < 3; ++c) tmp[c] = (0.5f * n[c] + 0.5f) * tmp[3];
		} else if (p.render_mode == NRS_RENDER_COST) {
			float col = (float)r.payload.n_steps / 128;
			tmp[0] = tmp[1] = tmp[2] = col; tmp[3] = 1.0f;
		}
		if (!p.linear_colors && (p.render_mode == NRS_RENDER_SHADE || p.render_mode == NRS_RENDER_SLICE)) { // tn:2474
			tmp[0] = srgb_to_linear(tmp[0]); tmp[1] = srgb_to_linear(tmp[1]); tmp[2] = srgb_to_linear(tmp[2]);
		}
		float* f = frame + 4 * (size_t)r.payload.idx;
		float one_minus = 1.0f - tmp[3];
		for (int c = 0; c < 4; ++c) f[c] = tmp[c] + f[c] * one_minus;
		if (tmp[3] > 0.2f) depth_buf[r.payload.idx] = r.depth;
	}
	if (stats) { stats->generated = generated; stats->composited = composited; stats->n_alive0 = n_alive0; stats->n_hit = n_hit; stats->iterations = iterations; }
}

// ------------------------------------------------------------------------------------------------
// Host authoring restatements: LUT builder (tet_mesh.cu:76-237, 368-673), SAT (bounding_box.cuh:126-178),
// MVC (mvc.h:125-188), local rotations (tet_mesh.cu:37-74), grid->bitfield (tn:514-555, 3642-3657)
// ------------------------------------------------------------------------------------------------
inline void project_pts(const V3* pts, int n, V3 axis, float& mn, float& mx) { // bounding_box.cuh:25-40
	mn = std::numeric_limits<float>::infinity(); mx = -mn;
	for (int i = 0; i < n; ++i) {
		float val = dot(axis, pts[i]);
		if (val < mn) mn = val;
		if (val > mx) mx = val;
	}
}
bool box_intersects_triangle(const Box& b, V3 ta, V3 tb, V3 tc) { // bounding_box.cuh:126-178
	float tmin, tmax, bmin, bmax;
	const V3 box_normals[3] = {v3(1, 0, 0), v3(0, 1, 0), v3(0, 0, 1)};
	V3 tn = cross(tb - ta, tc - ta);
	float nn = sqrtf(dot(tn, tn));
	tn = {tn.x / nn, tn.y / nn, tn.z / nn}; // triangle.normal() is .normalized(), triangle.cuh:39
	V3 tverts[3] = {ta, tb, tc};
	for (int i = 0; i < 3; ++i) {
		project_pts(tverts, 3, box_normals[i], tmin, tmax);
		if (tmax < comp(b.mn, i) || tmin > comp(b.mx, i)) return false;
	}
	V3 verts[8] = {v3(b.mn.x, b.mn.y, b.mn.z), v3(b.mn.x, b.mn.y, b.mx.z), v3(b.mn.x, b.mx.y, b.mn.z), v3(b.mn.x, b.mx.y, b.mx.z),
	               v3(b.mx.x, b.mn.y, b.mn.z), v3(b.mx.x, b.mn.y, b.mx.z), v3(b.mx.x, b.mx.y, b.mn.z), v3(b.mx.x, b.mx.y, b.mx.z)};
	float toff = dot(tn, ta);
	project_pts(verts, 8, tn, bmin, bmax);
	if (bmax < toff || bmin > toff) return false;
	V3 edges[3] = {ta - tb, ta - tc, tb - tc};
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j) {
			V3 axis = cross(edges[i], box_normals[j]);
			project_pts(verts, 8, axis, bmin, bmax);
			project_pts(tverts, 3, axis, tmin, tmax);
			if (bmax < tmin || bmin > tmax) return false;
		}
	return true;
}
inline V3 get_cell_pos(uint32_t x, uint32_t y, uint32_t z, uint32_t level) { // selection_utils.cu:65-68
	float s = scalbnf(1.0f, (int)level);
	return {(((float)x + 0.5f) / (float)GRID - 0.5f) * s + 0.5f, (((float)y + 0.5f) / (float)GRID - 0.5f) * s + 0.5f,
	        (((float)z + 0.5f) / (float)GRID - 0.5f) * s + 0.5f};
}
inline void get_cell_at_pos(V3 pos, uint32_t level, int out[3]) { // selection_utils.cu:70-83
	float mip_scale = scalbnf(1.0f, -(int)level);
	pos = pos - v3(0.5f, 0.5f, 0.5f);
	pos = pos * mip_scale;
	pos = pos + v3(0.5f, 0.5f, 0.5f);
	out[0] = clampi((int)(pos.x * (float)GRID), 0, GRID - 1);
	out[1] = clampi((int)(pos.y * (float)GRID), 0, GRID - 1);
	out[2] = clampi((int)(pos.z * (float)GRID), 0, GRID - 1);
}
const V3 corner_offsets[8] = {{-0.5f, -0.5f, -0.5f}, {-0.5f, -0.5f, 0.5f}, {-0.5f, 0.5f, -0.5f}, {0.5f, -0.5f, -0.5f},
                              {0.5f, 0.5f, -0.5f}, {-0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.5f, 0.5f, 0.5f}}; // tet_mesh.h:34-43

struct TetLut {
	std::vector<uint32_t> offsets, idx;
	std::vector<uint8_t> bitfield;
	uint32_t max_per_cell = 0;
};
// membership rule of build_tet_grid (tet_mesh.cu:405-466): per tet, per level, per cell of the tet's bbox:
// any of the 8 cell corners inside the tet, else any of the 4 faces intersecting the cell box.
// CSR order inside a cell = ascending tet index (what the reference gets from its thread-ordered second pass).
void build_tet_lut(const V3* verts, const uint32_t* tets, uint32_t n_tets, TetLut& out) {
	const uint32_t n_elements = GRIDVOL * CASCADES;
	std::vector<std::pair<uint32_t, uint32_t>> marks; // (cell, tet)
	out.bitfield.assign(n_elements / 8, 0);
	for (uint32_t i = 0; i < n_tets; ++i) {
		V3 tv[4] = {verts[tets[4 * i]], verts[tets[4 * i + 1]], verts[tets[4 * i + 2]], verts[tets[4 * i + 3]]};
		const float inf = std::numeric_limits<float>::infinity();
		V3 mn = v3(inf, inf, inf), mx = v3(-inf, -inf, -inf);
		for (int j = 0; j < 4; ++j) {
			mn = v3(fminf(mn.x, tv[j].x), fminf(mn.y, tv[j].y), fminf(mn.z, tv[j].z));
			mx = v3(fmaxf(mx.x, tv[j].x), fmaxf(mx.y, tv[j].y), fmaxf(mx.z, tv[j].z));
		}
		for (uint32_t level = 0; level < CASCADES; ++level) {
			float scale = scalbnf(1.0f, (int)level);
			int mi[3], ma[3];
			get_cell_at_pos(mn, level, mi);
			get_cell_at_pos(mx, level, ma);
			for (int x = mi[0]; x <= ma[0]; ++x)
				for (int y = mi[1]; y <= ma[1]; ++y)
					for (int z = mi[2]; z <= ma[2]; ++z) {
						V3 ps = get_cell_pos((uint32_t)x, (uint32_t)y, (uint32_t)z, level);
						bool inside = false;
						for (int c = 0; c < 8 && !inside; ++c) {
							V3 q = ps + corner_offsets[c] * scale * (1.0f / (float)GRID) ;
							// reference: pos + corner * scale / GRIDSIZE  (Eigen: (corner*scale)/128) -- /128 == *(1/128) exactly
							if (point_in_tet(tv[0], tv[1], tv[2], tv[3], q)) inside = true;
						}
						if (!inside) {
							V3 h = v3(0.5f, 0.5f, 0.5f) * scale * (1.0f / (float)GRID);
							Box cube;
							V3 a = ps - h, b = ps + h;
							cube.mn = v3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z));
							cube.mx = v3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z));
							for (int j = 0; j < 4 && !inside; ++j)
								if (box_intersects_triangle(cube, tv[j], tv[(j + 1) % 4], tv[(j + 2) % 4])) inside = true;
						}
						if (inside) {
							uint32_t pos_idx = morton3D((uint32_t)x, (uint32_t)y, (uint32_t)z);
							marks.emplace_back(level * GRIDVOL + pos_idx, i);
							set_bitfield_at(pos_idx, level, true, out.bitfield.data());
						}
					}
		}
	}
	out.offsets.assign((size_t)n_elements + 1, 0);
	for (auto& mk : marks) out.offsets[(size_t)mk.first + 1]++;
	out.max_per_cell = 0;
	for (uint32_t c = 0; c < n_elements; ++c) {
		out.max_per_cell = std::max(out.max_per_cell, out.offsets[(size_t)c + 1]);
		out.offsets[(size_t)c + 1] += out.offsets[c];
	}
	out.idx.assign(marks.size(), 0);
	std::vector<uint32_t> fill(out.offsets.begin(), out.offsets.end() - 1);
	for (auto& mk : marks) out.idx[fill[mk.first]++] = mk.second; // marks are in ascending tet order already
}

// MVC3D::computeCoordinatesCustomCode with float_t = float, mvc.h:125-188.  Returns true on the degenerate exits.
bool mvc_one(V3 eta, const uint32_t* tris, uint32_t n_tris, const V3* cv, uint32_t n_v, float* weights, std::vector<float>& w_weights,
             std::vector<float>& d, std::vector<V3>& u) {
	typedef float T;
	T epsilon = 0.00000001;
	for (uint32_t v = 0; v < n_v; ++v) weights[v] = 0.f;
	T sumWeights = 0.0;
	d.assign(n_v, 0.f); u.resize(n_v);
	for (uint32_t v = 0; v < n_v; ++v) {
		V3 e = eta - cv[v];
		d[v] = sqrtf(dot(e, e));
		if (d[v] < epsilon) { weights[v] = 1.0; return true; }
		V3 q = cv[v] - eta;
		u[v] = {q.x / d[v], q.y / d[v], q.z / d[v]};
	}
	w_weights.assign(n_v, 0.f);
	uint32_t vid[3]; T l[3], theta[3], w[3], c[3], s[3];
	for (uint32_t t = 0; t < n_tris; ++t) {
		for (int i = 0; i < 3; ++i) vid[i] = tris[3 * t + i];
		for (int i = 0; i < 3; ++i) { V3 q = u[vid[(i + 1) % 3]] - u[vid[(i + 2) % 3]]; l[i] = sqrtf(dot(q, q)); }
		// Overloads as the reference's translation units resolve them (unqualified sin / fabs with <math.h> and CUDA's global float
		// overloads in scope): sin(float) is the FLOAT function; asin(l / 2.0) and sqrt(std::max<double>(..)) take doubles.  Pinned by
		// mvc.h itself compiled with Eigen::Vector3f points (oracle/ref_render.cpp: ref_mvc_compute; tests/test_ref_pin.py).
		for (int i = 0; i < 3; ++i) theta[i] = (T)(2.0 * asin((double)l[i] / 2.0));
		T h = (T)(((double)((theta[0] + theta[1]) + theta[2])) / 2.0);
		if (M_PI - (double)h < (double)epsilon) {
			for (int i = 0; i < 3; ++i) w[i] = (sinf(theta[i]) * l[(i + 2) % 3]) * l[(i + 1) % 3];
			sumWeights = (w[0] + w[1]) + w[2];
			weights[vid[0]] = w[0] / sumWeights; weights[vid[1]] = w[1] / sumWeights; weights[vid[2]] = w[2] / sumWeights;
			return true;
		}
		for (int i = 0; i < 3; ++i)
			c[i] = (T)((((2.0 * (double)sinf(h)) * (double)sinf(h - theta[i])) / (double)(sinf(theta[(i + 1) % 3]) * sinf(theta[(i + 2) % 3]))) - 1.0);
		T sign_basis = 1;
		if ((double)dot(cross(u[vid[0]], u[vid[1]]), u[vid[2]]) < 0.0) sign_basis = -1;
		for (int i = 0; i < 3; ++i) s[i] = (T)((double)sign_basis * sqrt(std::max<double>(0.0, 1.0 - (double)(c[i] * c[i]))));
		if (fabsf(s[0]) < epsilon || fabsf(s[1]) < epsilon || fabsf(s[2]) < epsilon) continue;
		for (int i = 0; i < 3; ++i)
			w[i] = (T)(((double)((theta[i] - c[(i + 1) % 3] * theta[(i + 2) % 3]) - c[(i + 2) % 3] * theta[(i + 1) % 3])) /
			           (((2.0 * (double)d[vid[i]]) * (double)sinf(theta[(i + 1) % 3])) * (double)s[(i + 2) % 3]));
		sumWeights += (w[0] + w[1] + w[2]);
		w_weights[vid[0]] += w[0]; w_weights[vid[1]] += w[1]; w_weights[vid[2]] += w[2];
	}
	for (uint32_t v = 0; v < n_v; ++v) weights[v] = w_weights[v] / sumWeights;
	return false;
}

// The reference's 3x3 SVD (svd3.h: McAdams/Selle/Tamstorf/Teran/Sifakis TR1690, implementation by E. Jang), restated in
// fp32 with its operation order: four cyclic Jacobi sweeps with approximate Givens rotations on A^T A (quaternion
// accumulation), column sort with sign flips, QR by three Givens rotations.  Inexact by design: U V^T is up to 1.3e-2 away
// from the exact polar factor on ordinary tets, which is why it is restated rather than replaced.  Pinned bit for bit by
// tests/golden/ref_rotations_golden.npz (the reference header itself, compiled into oracle/_ref/libref_render.so).  Matrices are row-major [r][c].
struct Quat { float x, y, z, w; };
inline float ref_rsqrt(float x) { return 1.0f / sqrtf(x); }            // rsqrt(float) of nvcc's host math
inline float ref_rsqrt1(float x) {                                    // svd3.h:54-63
	float xhalf = 0.5f * x;
	int32_t i = (int32_t)f2u(x);
	i = 0x5f37599e - (i >> 1);
	x = u2f((uint32_t)i);
	x = x * (1.5f - xhalf * x * x);
	x = x * (1.5f - xhalf * x * x);
	return x;
}
// jacobiConjugation (svd3.h:165-213) for the pivot whose quaternion component is `zc` (and x/y components xc/yc);
// sym = {s11, s21, s22, s31, s32, s33}
inline void jacobi_conjugation(int xc, int yc, int zc, float sym[6], float q[4]) {
	float s11 = sym[0], s21 = sym[1], s22 = sym[2], s31 = sym[3], s32 = sym[4], s33 = sym[5];
	float ch = 2 * (s11 - s22);                                        // approximateGivensQuaternion, :147-163
	float sh = s21;
	bool b = 5.828427124 * sh * sh < ch * ch;
	float w = ref_rsqrt(ch * ch + sh * sh);
	ch = b ? w * ch : (float)0.923879532;
	sh = b ? w * sh : (float)0.3826834323;
	float scale = ch * ch + sh * sh;
	float a = (ch * ch - sh * sh) / scale;
	float bb = (2 * sh * ch) / scale;
	float o11 = a * (a * s11 + bb * s21) + bb * (a * s21 + bb * s22);
	float o21 = a * (-bb * s11 + a * s21) + bb * (-bb * s21 + a * s22);
	float o22 = -bb * (-bb * s11 + a * s21) + a * (-bb * s21 + a * s22);
	float o31 = a * s31 + bb * s32;
	float o32 = -bb * s31 + a * s32;
	float o33 = s33;
	float tmp0 = q[0] * sh, tmp1 = q[1] * sh, tmp2 = q[2] * sh;
	float tmp[3] = {tmp0, tmp1, tmp2};
	sh *= q[3];
	q[0] *= ch; q[1] *= ch; q[2] *= ch; q[3] *= ch;
	q[zc] += sh;
	q[3] -= tmp[zc];
	q[xc] += tmp[yc];
	q[yc] -= tmp[xc];
	sym[0] = o22; sym[1] = o32; sym[2] = o33; sym[3] = o21; sym[4] = o31; sym[5] = o11; // re-arranged for the next pivot
}
inline void qr_givens_quaternion(float a1, float a2, float& ch, float& sh) { // svd3.h:270-285
	float epsilon = (float)1e-6;
	float sq = a1 * a1 + a2 * a2;
	float rho = sq * ref_rsqrt1(sq);
	sh = rho > epsilon ? a2 : 0;
	ch = fabsf(a1) + fmaxf(rho, epsilon);
	if (a1 < 0) std::swap(sh, ch);
	float w = ref_rsqrt(ch * ch + sh * sh);
	ch *= w;
	sh *= w;
}
// U, V of svd(A) (svd3.h:355-403)
void ref_svd_uv(const float A[3][3], float U[3][3], float V[3][3]) {
	float sym[6] = {
		A[0][0] * A[0][0] + A[1][0] * A[1][0] + A[2][0] * A[2][0],   // (A^T A)11
		A[0][1] * A[0][0] + A[1][1] * A[1][0] + A[2][1] * A[2][0],   // 21
		A[0][1] * A[0][1] + A[1][1] * A[1][1] + A[2][1] * A[2][1],   // 22
		A[0][2] * A[0][0] + A[1][2] * A[1][0] + A[2][2] * A[2][0],   // 31
		A[0][2] * A[0][1] + A[1][2] * A[1][1] + A[2][2] * A[2][1],   // 32
		A[0][2] * A[0][2] + A[1][2] * A[1][2] + A[2][2] * A[2][2]};  // 33
	float q[4] = {0, 0, 0, 1};
	for (int i = 0; i < 4; ++i) {
		jacobi_conjugation(0, 1, 2, sym, q);
		jacobi_conjugation(1, 2, 0, sym, q);
		jacobi_conjugation(2, 0, 1, sym, q);
	}
	{ // quatToMat3, :121-145
		float w = q[3], x = q[0], y = q[1], z = q[2];
		float qxx = x * x, qyy = y * y, qzz = z * z, qxz = x * z, qxy = x * y, qyz = y * z, qwx = w * x, qwy = w * y, qwz = w * z;
		V[0][0] = 1 - 2 * (qyy + qzz); V[0][1] = 2 * (qxy - qwz); V[0][2] = 2 * (qxz + qwy);
		V[1][0] = 2 * (qxy + qwz); V[1][1] = 1 - 2 * (qxx + qzz); V[1][2] = 2 * (qyz - qwx);
		V[2][0] = 2 * (qxz - qwy); V[2][1] = 2 * (qyz + qwx); V[2][2] = 1 - 2 * (qxx + qyy);
	}
	float B[3][3];
	for (int r = 0; r < 3; ++r)
		for (int c = 0; c < 3; ++c) B[r][c] = A[r][0] * V[0][c] + A[r][1] * V[1][c] + A[r][2] * V[2][c];
	auto rho_of = [&](int c) { return B[0][c] * B[0][c] + B[1][c] * B[1][c] + B[2][c] * B[2][c]; };
	auto neg_swap = [&](bool c, int i, int j) {                        // condNegSwap on columns i, j of B and V, :78-84
		for (int r = 0; r < 3; ++r) {
			float zb = -B[r][i]; B[r][i] = c ? B[r][j] : B[r][i]; B[r][j] = c ? zb : B[r][j];
			float zv = -V[r][i]; V[r][i] = c ? V[r][j] : V[r][i]; V[r][j] = c ? zv : V[r][j];
		}
	};
	float rho1 = rho_of(0), rho2 = rho_of(1), rho3 = rho_of(2);
	bool c = rho1 < rho2;
	neg_swap(c, 0, 1);
	if (c) std::swap(rho1, rho2);
	c = rho1 < rho3;
	neg_swap(c, 0, 2);
	if (c) std::swap(rho1, rho3);
	c = rho2 < rho3;
	neg_swap(c, 1, 2);
	float ch1, sh1, ch2, sh2, ch3, sh3, R[3][3];
	qr_givens_quaternion(B[0][0], B[1][0], ch1, sh1);
	float a = 1 - 2 * sh1 * sh1, b = 2 * ch1 * sh1;
	for (int k = 0; k < 3; ++k) { R[0][k] = a * B[0][k] + b * B[1][k]; R[1][k] = -b * B[0][k] + a * B[1][k]; R[2][k] = B[2][k]; }
	qr_givens_quaternion(R[0][0], R[2][0], ch2, sh2);
	a = 1 - 2 * sh2 * sh2; b = 2 * ch2 * sh2;
	for (int k = 0; k < 3; ++k) { B[0][k] = a * R[0][k] + b * R[2][k]; B[1][k] = R[1][k]; B[2][k] = -b * R[0][k] + a * R[2][k]; }
	qr_givens_quaternion(B[1][1], B[2][1], ch3, sh3);
	float sh12 = sh1 * sh1, sh22 = sh2 * sh2, sh32 = sh3 * sh3;        // Q = Q1 Q2 Q3, :336-352
	U[0][0] = (-1 + 2 * sh12) * (-1 + 2 * sh22);
	U[0][1] = 4 * ch2 * ch3 * (-1 + 2 * sh12) * sh2 * sh3 + 2 * ch1 * sh1 * (-1 + 2 * sh32);
	U[0][2] = 4 * ch1 * ch3 * sh1 * sh3 - 2 * ch2 * (-1 + 2 * sh12) * sh2 * (-1 + 2 * sh32);
	U[1][0] = 2 * ch1 * sh1 * (1 - 2 * sh22);
	U[1][1] = -8 * ch1 * ch2 * ch3 * sh1 * sh2 * sh3 + (-1 + 2 * sh12) * (-1 + 2 * sh32);
	U[1][2] = -2 * ch3 * sh3 + 4 * sh1 * (ch3 * sh1 * sh3 + ch1 * ch2 * sh2 * (-1 + 2 * sh32));
	U[2][0] = 2 * ch2 * sh2;
	U[2][1] = 2 * ch3 * (1 - 2 * sh22) * sh3;
	U[2][2] = (-1 + 2 * sh22) * (-1 + 2 * sh32);
}

} // namespace

// =================================================================================================
// C interface for ctypes (tests / smoke / bench cpu_baseline only)
// =================================================================================================
extern "C" {

uint16_t orc_f2h(float f) { return f2h(f); }
float orc_h2f(uint16_t h) { return h2f(h); }
uint32_t orc_morton3D(uint32_t x, uint32_t y, uint32_t z) { return morton3D(x, y, z); }
uint32_t orc_morton3D_invert(uint32_t x) { return morton3D_invert(x); }
uint32_t orc_sobol(uint32_t index, uint32_t dim) { return sobol(index, dim); }
float orc_ld_random_val(uint32_t index, uint32_t seed, uint32_t dim) { return ld_random_val(index, seed, dim); }
void orc_ld_random_pixel_offset(uint32_t spp, float* out2) { ld_random_pixel_offset(spp, out2); }
int orc_mip_from_pos(const float* pos) { return mip_from_pos(v3(pos[0], pos[1], pos[2])); }
int orc_mip_from_dt(float dt, const float* pos) { return mip_from_dt(dt, v3(pos[0], pos[1], pos[2])); }
uint32_t orc_cascaded_grid_idx_at(const float* pos, uint32_t mip) { return cascaded_grid_idx_at(v3(pos[0], pos[1], pos[2]), mip); }
float orc_calc_dt(float t, float cone) { return calc_dt(t, cone); }
float orc_min_step(void) { return MIN_STEP; }
float orc_max_step(void) { return MAX_STEP; }
float orc_warp_dt(float dt) { return warp_dt(dt); }
float orc_unwarp_dt(float dt) { return unwarp_dt(dt); }
float orc_distance_to_next_voxel(const float* pos, const float* dir, uint32_t res) {
	V3 d = v3(dir[0], dir[1], dir[2]);
	return distance_to_next_voxel(v3(pos[0], pos[1], pos[2]), d, v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z), res);
}
float orc_advance_to_next_voxel(float t, float cone, const float* pos, const float* dir, uint32_t res) {
	V3 d = v3(dir[0], dir[1], dir[2]);
	return advance_to_next_voxel(t, cone, v3(pos[0], pos[1], pos[2]), d, v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z), res);
}
int orc_frexp_exponent(float x) { return frexp_exponent(x); }
float orc_srgb_to_linear(float x) { return srgb_to_linear(x); }
void orc_evaluate_sh9(const float* sh27, const float* dir, float* rgb) { evaluate_sh9(sh27, v3(dir[0], dir[1], dir[2]), rgb); }
void orc_bary_tet(const float* abcd12, const float* p, float* out4) {
	bary_tet(v3(abcd12[0], abcd12[1], abcd12[2]), v3(abcd12[3], abcd12[4], abcd12[5]), v3(abcd12[6], abcd12[7], abcd12[8]),
	         v3(abcd12[9], abcd12[10], abcd12[11]), v3(p[0], p[1], p[2]), out4);
}
int orc_point_in_tet(const float* abcd12, const float* p) {
	return point_in_tet(v3(abcd12[0], abcd12[1], abcd12[2]), v3(abcd12[3], abcd12[4], abcd12[5]), v3(abcd12[6], abcd12[7], abcd12[8]),
	                    v3(abcd12[9], abcd12[10], abcd12[11]), v3(p[0], p[1], p[2])) ? 1 : 0;
}
int orc_box_intersects_triangle(const float* box6, const float* tri9) {
	Box b{v3(box6[0], box6[1], box6[2]), v3(box6[3], box6[4], box6[5])};
	return box_intersects_triangle(b, v3(tri9[0], tri9[1], tri9[2]), v3(tri9[3], tri9[4], tri9[5]), v3(tri9[6], tri9[7], tri9[8])) ? 1 : 0;
}

size_t orc_model_n_params(const nrs_model_desc* d) {
	if (!desc_supported(*d)) return 0;
	LevelTable lt;
	make_level_table(*d, lt);
	return (size_t)n_density_weights(*d) + n_rgb_weights(*d) + (size_t)lt.total_entries * 2;
}
int orc_model_level_table(const nrs_model_desc* d, float* scale, uint32_t* res, uint32_t* off, uint32_t* cnt, uint32_t* hashed) {
	if (!desc_supported(*d)) return -1;
	LevelTable lt;
	make_level_table(*d, lt);
	for (uint32_t l = 0; l < d->n_levels; ++l) { scale[l] = lt.scale[l]; res[l] = lt.resolution[l]; off[l] = lt.offset[l]; cnt[l] = lt.count[l]; hashed[l] = lt.hashed[l]; }
	return 0;
}
void* orc_model_create(const nrs_model_desc* d, const uint16_t* params, size_t n_params, const uint8_t* bitfield) {
	if (!desc_supported(*d)) return nullptr;
	Model* m = new Model();
	m->desc = *d;
	make_level_table(*d, m->lt);
	if (n_params != (size_t)m->n_mlp_w() + (size_t)m->lt.total_entries * 2) { delete m; return nullptr; }
	m->aabb = Box{v3(d->aabb_min[0], d->aabb_min[1], d->aabb_min[2]), v3(d->aabb_max[0], d->aabb_max[1], d->aabb_max[2])};
	m->params.assign(params, params + n_params);
	m->wf.resize(m->n_mlp_w());
	for (uint32_t i = 0; i < m->n_mlp_w(); ++i) m->wf[i] = h2f(params[i]);
	if (bitfield) m->bitfield.assign(bitfield, bitfield + NRS_BITFIELD_BYTES);
	else m->bitfield.assign(NRS_BITFIELD_BYTES, 0);
	return m;
}
void orc_model_set_bitfield(void* model, const uint8_t* bitfield) { ((Model*)model)->bitfield.assign(bitfield, bitfield + NRS_BITFIELD_BYTES); }
void orc_model_destroy(void* model) { delete (Model*)model; }
void orc_model_set_fast(void* model, int on) { ((Model*)model)->fast = on != 0; } // bench.py's cpu_baseline flavour; see Model::fast
void orc_model_set_numerics(void* model, uint32_t grid_acc, uint32_t mlp_acc) { ((Model*)model)->grid_acc = grid_acc; ((Model*)model)->mlp_acc = mlp_acc; }

// out: [n x 32] fp16 interleaved
void orc_hashgrid_encode(void* model, uint32_t n, const float* in, uint32_t ld_in, uint16_t* out) {
	const Model& m = *(Model*)model;
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t)n; ++i) hashgrid_encode_one(m, in + (size_t)i * ld_in, out + (size_t)i * 32);
}
void orc_sh4_encode(uint32_t n, const float* dir01, uint32_t ld, uint16_t* out) {
	for (uint32_t i = 0; i < n; ++i) sh4_encode_one(dir01 + (size_t)i * ld, out + (size_t)i * 16);
}
// layout: 0 = planes out[c*ld_out + s], 1 = interleaved out[s*16 + c]
void orc_network_inference(void* model, uint32_t n, const float* in7, uint16_t* out, uint32_t ld_out, int layout) {
	const Model& m = *(Model*)model;
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t)n; ++i) {
		uint16_t o[16];
		network_inference_one(m, in7 + (size_t)i * 7, o);
		for (int c = 0; c < 16; ++c)
			if (layout == NRS_PLANES) out[(size_t)c * ld_out + i] = o[c];
			else out[(size_t)i * 16 + c] = o[c];
	}
}
// CudaRenderBuffer::accumulate (render_buffer.cu:540-560) + accumulate_kernel (:217-254): the running mean over the spp frames of a view.
// sample_count = frames already accumulated (0 clears the buffer first); color_space: 0 Linear, 1 SRGB (linear_to_srgb, common_device.cuh:55-61, before
// the mean), 2 VisPosNeg (the EncodingVis picture: positive part in x, negative in y).
static inline float linear_to_srgb(float linear) { return linear < 0.0031308f ? 12.92f * linear : 1.055f * powf(linear, 0.41666f) - 0.055f; }
void orc_accumulate(int width, int height, const float* frame, float* accumulate, uint32_t sample_count, int color_space) {
	const size_t n = (size_t)width * height;
	if (sample_count == 0) memset(accumulate, 0, sizeof(float) * 4 * n);
	const float sc = (float)sample_count;
	for (size_t i = 0; i < n; ++i) {
		float color[4] = {frame[4 * i], frame[4 * i + 1], frame[4 * i + 2], frame[4 * i + 3]};
		float* tmp = accumulate + 4 * i;
		if (color_space == 2) {
			const float val = color[0] - color[1];
			float tmp_val = tmp[0] - tmp[1];
			tmp_val = (tmp_val * sc + val) / (sc + 1);
			tmp[0] = fmaxf(tmp_val, 0.0f);
			tmp[1] = fmaxf(-tmp_val, 0.0f);
		} else {
			if (color_space == 1) for (int c = 0; c < 3; ++c) color[c] = linear_to_srgb(color[c]);
			for (int c = 0; c < 3; ++c) tmp[c] = (tmp[c] * sc + color[c]) / (sc + 1);
		}
		tmp[3] = (tmp[3] * sc + color[3]) / (sc + 1);
	}
}
// tcnn input_gradient(stream, 3, ...) restated: d density_raw / d warped position of n samples [n x 7] -> [n x 3]
void orc_density_input_gradient(void* model, uint32_t n, const float* in7, float* grad3) {
	const Model& m = *(const Model*)model;
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t)n; ++i) density_input_gradient_one(m, in7 + 7 * (size_t)i, grad3 + 3 * (size_t)i);
}
// tcnn visualize_activation restated: activation `dim` of forward_activations(layer) of n samples [n x 7] -> [n]; returns 0 for an unknown layer / unit
int orc_network_activation(void* model, uint32_t n, const float* in7, uint32_t layer, uint32_t dim, float* out) {
	const Model& m = *(const Model*)model;
	if (dim >= network_layer_width(m.desc, layer)) return 0;
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t)n; ++i) out[i] = network_activation_one(m, in7 + 7 * (size_t)i, layer, dim);
	return 1;
}
// the same two in the shape the reference calls them (7-row matrices; visualize_activation writes over its input): the callbacks of oracle/ref_render.cpp
void orc_density_input_gradient7(void* model, uint32_t n, const float* in7, float* grad7) {
	const Model& m = *(const Model*)model;
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t)n; ++i) {
		float* g = grad7 + 7 * (size_t)i;
		density_input_gradient_one(m, in7 + 7 * (size_t)i, g);
		g[3] = g[4] = g[5] = g[6] = 0.f;
	}
}
void orc_visualize_activation7(void* model, uint32_t n, float* in7, uint32_t layer, uint32_t dim) {
	const Model& m = *(const Model*)model;
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t)n; ++i) {
		float* c = in7 + 7 * (size_t)i;
		const float v = network_activation_one(m, c, layer, dim);
		c[0] = fmaxf(-v, 0.0f); c[1] = fmaxf(v, 0.0f); c[2] = 0.f; // extract_dimension_pos_neg_kernel
		c[3] = c[4] = c[5] = c[6] = 1.f;
	}
}
void orc_network_density(void* model, uint32_t n, const float* in, uint32_t ld_in, uint16_t* out, uint32_t ld_out, int layout) {
	const Model& m = *(Model*)model;
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t)n; ++i) {
		uint16_t feat[32], o[16];
		hashgrid_encode_one(m, in + (size_t)i * ld_in, feat);
		density_mlp_one(m, feat, o);
		for (int c = 0; c < 16; ++c)
			if (layout == NRS_PLANES) out[(size_t)c * ld_out + i] = o[c];
			else out[(size_t)i * 16 + c] = o[c];
	}
}

// Testbed::get_density_on_grid (tn:4538-4586): generate_grid_samples_nerf_uniform (tn:406-417) -> density() -> grid_samples_half_to_float
// (tn:464-481).  density_grid: float [5*128^3] or NULL (no masking).
void orc_density_on_grid(void* model, const uint32_t* res, const float* box_mn, const float* box_mx, const float* density_grid, float* out) {
	const Model& m = *(Model*)model;
	const int64_t n = (int64_t)res[0] * res[1] * res[2];
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < n; ++i) {
		const uint32_t x = (uint32_t)(i % res[0]), y = (uint32_t)((i / res[0]) % res[1]), z = (uint32_t)(i / ((int64_t)res[0] * res[1]));
		V3 pos = {(float)x * (1.f / (float)res[0]), (float)y * (1.f / (float)res[1]), (float)z * (1.f / (float)res[2])};
		pos = {pos.x * (box_mx[0] - box_mn[0]) + box_mn[0], pos.y * (box_mx[1] - box_mn[1]) + box_mn[1], pos.z * (box_mx[2] - box_mn[2]) + box_mn[2]};
		const V3 w = warp_position(pos, m.aabb);
		const float p[3] = {w.x, w.y, w.z};
		uint16_t feat[32], o[16];
		hashgrid_encode_one(m, p, feat);
		density_mlp_one(m, feat, o);
		float v = h2f(o[0]);
		if (density_grid) {
			const V3 up = unwarp_position(w, m.aabb);
			const uint32_t mip = (uint32_t)mip_from_pos(up);
			if (density_grid[cascaded_grid_idx_at(up, mip) + (size_t)mip * GRIDVOL] < 0.01f) v = -10000.f;
		}
		out[i] = v;
	}
}
// GrowingSelection::project_selection_pixels (growing_selection.cu:1832-2035), in the reference's three steps:
// shoot_selection_rays_kernel (:1673-1766) writes every sample of every ray, NerfNetwork::density runs on all of them,
// composite_shot_rays (:1768-1829) walks each ray's samples.  Per-pixel outputs (the reference compacts in atomic order).
// A ray whose transmittance falls to the threshold only behind its LAST sample leaves the reference's output slot unwritten
// (uninitialised workspace); it is reported as not found here, as in the product.
void orc_project_selection_pixels(void* model, const nrs_render_params* p, const int32_t* pixels, uint32_t n, float threshold, float* positions,
                                  uint32_t* cells, uint8_t* found) {
	const Model& m = *(Model*)model;
	const uint8_t* grid = m.bitfield.data();
	float offset[2];
	ld_random_pixel_offset(0, offset);
#pragma omp parallel for schedule(dynamic, 16)
	for (int64_t i = 0; i < (int64_t)n; ++i) {
		// pixel_to_ray(0, pixel, ...), common_device.cuh:259-284; NOT normalised here
		const float W = (float)p->resolution[0], H = (float)p->resolution[1];
		const float uvx = ((float)pixels[2 * i] + offset[0]) / W, uvy = ((float)pixels[2 * i + 1] + offset[1]) / H;
		const V3 dir = {(uvx - p->screen_center[0]) * W / p->focal_length[0], (uvy - p->screen_center[1]) * H / p->focal_length[1], 1.0f};
		const float* cam = p->camera_matrix1;
		const V3 d = mat3_mul(cam, dir);
		const V3 o = cam_col(cam, 3);
		const V3 idir = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
		float tmin, tmax;
		ray_intersect(m.aabb, o, d, tmin, tmax);
		const float startt = fmaxf(tmin, 0.0f);
		// the samples (second pass of the kernel; the first only counts them)
		std::vector<V3> wpos;
		std::vector<float> wdt;
		float t = startt;
		V3 pos;
		while (box_contains(m.aabb, pos = o + d * t) && wpos.size() < 1024) {
			const float dt = calc_dt(t, p->cone_angle_constant);
			const uint32_t mip = (uint32_t)mip_from_dt(dt, pos);
			if (density_grid_occupied_at(pos, grid, mip)) {
				wpos.push_back(warp_position(pos, m.aabb));
				wdt.push_back(warp_dt(dt));
				t += dt;
			} else {
				t = advance_to_next_voxel(t, p->cone_angle_constant, pos, d, idir, GRID >> mip);
			}
		}
		found[i] = 0;
		cells[i] = 0;
		positions[3 * i] = m.aabb.mn.x - 1.f; positions[3 * i + 1] = m.aabb.mn.y - 1.f; positions[3 * i + 2] = m.aabb.mn.z - 1.f;
		// composite_shot_rays
		float T = 1.f;
		for (size_t c = 0; c < wpos.size(); ++c) {
			if (T <= threshold) {
				const V3 up = unwarp_position(wpos[c], m.aabb);
				positions[3 * i] = up.x; positions[3 * i + 1] = up.y; positions[3 * i + 2] = up.z;
				const uint32_t level = (uint32_t)mip_from_pos(up);
				cells[i] = level * GRIDVOL + cascaded_grid_idx_at(up, level);
				found[i] = 1;
				break;
			}
			const float in[3] = {wpos[c].x, wpos[c].y, wpos[c].z};
			uint16_t feat[32], out[16];
			hashgrid_encode_one(m, in, feat);
			density_mlp_one(m, feat, out);
			const float density = network_to_density(h2f(out[0]), m.desc.density_activation);
			const float alpha = 1.f - expf(-density * unwarp_dt(wdt[c]));
			T *= (1.f - alpha);
		}
	}
}
// GrowingSelection::compute_poisson_boundary (growing_selection.cu:2220-2348); jitter = the (float)std::rand() / RAND_MAX draws.
void orc_poisson_boundary(void* model, const float* vertices, uint32_t n_verts, uint32_t sh_width, uint32_t hemisphere_width, const float* jitter,
                          int is_inside, float* density_out, float* sh_out, float* coords_out /* nullable [n][7] */) {
	const Model& m = *(Model*)model;
	const uint32_t n_sh = sh_width * sh_width;
#pragma omp parallel for schedule(dynamic, 4)
	for (int64_t k = 0; k < (int64_t)n_verts; ++k) {
		float sh[27] = {0};
		for (uint32_t i = 0; i < sh_width; ++i)
			for (uint32_t j = 0; j < sh_width; ++j) {
				const size_t s = (size_t)n_sh * k + (size_t)i * sh_width + j;
				float u = ((float)i + jitter[2 * s]) / (float)(int)hemisphere_width;
				float v = ((float)j + jitter[2 * s + 1]) / (float)(int)hemisphere_width;
				float theta = (float)(2.f * M_PI * v);
				float phi = acosf(2.f * u - 1.f);
				const V3 dir = {cosf(theta) * sinf(phi), sinf(theta) * sinf(phi), cosf(phi)};
				const V3 wp = warp_position(v3(vertices[3 * k], vertices[3 * k + 1], vertices[3 * k + 2]), m.aabb);
				const V3 wd = {(dir.x + 1.f) * 0.5f, (dir.y + 1.f) * 0.5f, (dir.z + 1.f) * 0.5f};
				const float in[7] = {wp.x, wp.y, wp.z, 0.f, wd.x, wd.y, wd.z};
				if (coords_out) memcpy(coords_out + s * 7, in, sizeof(in));
				uint16_t out[16];
				network_inference_one(m, in, out);
				const float rgb[3] = {network_to_rgb(h2f(out[0]), m.desc.rgb_activation), network_to_rgb(h2f(out[1]), m.desc.rgb_activation),
				                      network_to_rgb(h2f(out[2]), m.desc.rgb_activation)};
				if (i == 0 && j == 0) { // target_density[k] = density_host[k * n_sh_samples]
					float density = network_to_density(h2f(out[3]), m.desc.density_activation);
					if (is_inside) {
						const V3 pos = unwarp_position(wp, m.aabb);
						if (!density_grid_occupied_at(pos, m.bitfield.data(), (uint32_t)mip_from_pos(pos))) density = 0.0f;
					}
					density_out[k] = density;
				}
				// project_sh9(unwarp_direction(dir), rgb), sh_utils.cu:30-69
				const V3 d = unwarp_direction(wd);
				const float x = d.x, y = d.y, z = d.z;
				for (int col = 0; col < 3; ++col) {
					float* c9 = sh + 9 * col;
					float c;
					c = 0.282095; c9[0] += rgb[col] * c * 1.0f;
					c = 0.488603; c9[1] += rgb[col] * (c * y) * 1.0f; c9[2] += rgb[col] * (c * z) * 1.0f; c9[3] += rgb[col] * (c * x) * 1.0f;
					c = 1.092548; c9[4] += rgb[col] * (c * x * y) * 1.0f; c9[5] += rgb[col] * (c * y * z) * 1.0f; c9[7] += rgb[col] * (c * x * z) * 1.0f;
					c = 0.315392; c9[6] += rgb[col] * (c * (3 * z * z - 1)) * 1.0f;
					c = 0.546274; c9[8] += rgb[col] * (c * (x * x - y * y)) * 1.0f;
				}
			}
		const float scale = (float)(4 * M_PI / (n_sh));
		for (int c = 0; c < 27; ++c) sh_out[27 * k + c] = sh[c] * scale;
	}
}
// Testbed::get_rgba_on_grid (tn:4588-4611): generate_grid_samples_nerf_uniform_dir (tn:419-431) -> inference() -> compute_nerf_density (tn:624-635)
void orc_rgba_on_grid(void* model, const uint32_t* res, const float* box_mn, const float* box_mx, const float* ray_dir, float* out4) {
	const Model& m = *(Model*)model;
	const int64_t n = (int64_t)res[0] * res[1] * res[2];
	const V3 wd = warp_direction(v3(ray_dir[0], ray_dir[1], ray_dir[2]));
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < n; ++i) {
		const uint32_t x = (uint32_t)(i % res[0]), y = (uint32_t)((i / res[0]) % res[1]), z = (uint32_t)(i / ((int64_t)res[0] * res[1]));
		V3 pos = {(float)x * (1.f / (float)res[0]), (float)y * (1.f / (float)res[1]), (float)z * (1.f / (float)res[2])};
		pos = {pos.x * (box_mx[0] - box_mn[0]) + box_mn[0], pos.y * (box_mx[1] - box_mn[1]) + box_mn[1], pos.z * (box_mx[2] - box_mn[2]) + box_mn[2]};
		const V3 w = warp_position(pos, m.aabb);
		const float in7[7] = {w.x, w.y, w.z, warp_dt(MIN_STEP), wd.x, wd.y, wd.z};
		uint16_t o[16];
		network_inference_one(m, in7, o);
		const float a = clampf(1.f - expf(-network_to_density(h2f(o[3]), m.desc.density_activation) / 100.0f), 0.0f, 1.0f);
		out4[4 * i] = network_to_rgb(h2f(o[0]), m.desc.rgb_activation) * a;
		out4[4 * i + 1] = network_to_rgb(h2f(o[1]), m.desc.rgb_activation) * a;
		out4[4 * i + 2] = network_to_rgb(h2f(o[2]), m.desc.rgb_activation) * a;
		out4[4 * i + 3] = a;
	}
}

void* orc_edit_create(const nrs_model_desc* d, const nrs_tet_mesh* mesh) {
	Edit* e = new Edit();
	e->aabb = Box{v3(d->aabb_min[0], d->aabb_min[1], d->aabb_min[2]), v3(d->aabb_max[0], d->aabb_max[1], d->aabb_max[2])};
	e->verts.resize(mesh->n_vertices); e->orig.resize(mesh->n_vertices);
	for (uint32_t i = 0; i < mesh->n_vertices; ++i) {
		e->verts[i] = v3(mesh->h_vertices[3 * i], mesh->h_vertices[3 * i + 1], mesh->h_vertices[3 * i + 2]);
		e->orig[i] = v3(mesh->h_original_vertices[3 * i], mesh->h_original_vertices[3 * i + 1], mesh->h_original_vertices[3 * i + 2]);
	}
	e->tets.assign(mesh->h_tets, mesh->h_tets + 4 * (size_t)mesh->n_tets);
	const size_t n_cells = (size_t)GRIDVOL * CASCADES;
	e->lut_off.assign(mesh->h_lut_offsets, mesh->h_lut_offsets + n_cells + 1);
	e->lut_idx.assign(mesh->h_lut_idx, mesh->h_lut_idx + e->lut_off[n_cells]);
	e->orig_bitfield.assign(mesh->h_original_bitfield, mesh->h_original_bitfield + NRS_BITFIELD_BYTES);
	if (mesh->h_local_rotations) e->rot.assign(mesh->h_local_rotations, mesh->h_local_rotations + 9 * (size_t)mesh->n_tets);
	e->copy = mesh->copy != 0;
	e->apply_poisson = mesh->apply_poisson != 0;
	e->residual_amplitude = mesh->residual_amplitude;
	if (e->apply_poisson) {
		e->shs.assign(mesh->h_boundary_shs, mesh->h_boundary_shs + 27 * (size_t)mesh->n_vertices);
		e->out_density.assign(mesh->h_boundary_outside_density, mesh->h_boundary_outside_density + mesh->n_vertices);
		e->res_density.assign(mesh->h_boundary_residual_density, mesh->h_boundary_residual_density + mesh->n_vertices);
	}
	e->bbox = bbox_of(e->verts);            // post_update_vertices, tet_mesh.cu:12-20
	e->warped_bbox = warp_box(e->bbox, e->aabb);
	e->orig_bbox = bbox_of(e->orig);        // ctor, tet_mesh.h:100-107
	e->orig_warped_bbox = warp_box(e->orig_bbox, e->aabb);
	return e;
}
// AffineDuplication(selection_box, ...) + update_destination, affine_duplication.h:26-40, 77-90; box bookkeeping
// affine_bounding_box.cuh:40-101.  Only center / scale / rot_matrix of the selection box enter (warp_box rebuilds the rest).
static void abox_finish(const float center[3], const float scale[3], const float rot[9], ABox& out) {
	float u[3], v[3], w[3], mn[3];
	for (int i = 0; i < 3; ++i) {
		u[i] = rot[i] * scale[0];
		v[i] = rot[3 + i] * scale[1];
		w[i] = rot[6 + i] * scale[2];
		mn[i] = sum3((-0.5f * rot[i]) * scale[0], (-0.5f * rot[3 + i]) * scale[1], (-0.5f * rot[6 + i]) * scale[2]) + center[i];
	}
	out.u = v3(u[0], u[1], u[2]); out.v = v3(v[0], v[1], v[2]); out.w = v3(w[0], w[1], w[2]);
	out.mn = v3(mn[0], mn[1], mn[2]);
	out.center = v3(center[0], center[1], center[2]);
	out.uu = dot(out.u, out.u); out.vv = dot(out.v, out.v); out.ww = dot(out.w, out.w);
}
void* orc_edit_create_affine(const nrs_model_desc* d, const nrs_affine_duplication* op) {
	Edit* e = new Edit();
	e->kind = 1;
	e->aabb = Box{v3(d->aabb_min[0], d->aabb_min[1], d->aabb_min[2]), v3(d->aabb_max[0], d->aabb_max[1], d->aabb_max[2])};
	const float diag[3] = {d->aabb_max[0] - d->aabb_min[0], d->aabb_max[1] - d->aabb_min[1], d->aabb_max[2] - d->aabb_min[2]};
	float sc[3], ss[3], dc[3], ds[3], drot[9];
	for (int i = 0; i < 3; ++i) {
		dc[i] = op->selection_center[i] + op->translation[i];   // translate
		ds[i] = op->selection_scale[i] * op->scale[i];           // scale_with_vector
	}
	for (int c = 0; c < 3; ++c)                                   // rotate: rot_matrix = rotation * rot_matrix
		for (int r = 0; r < 3; ++r)
			drot[3 * c + r] = sum3(op->rotation[r] * op->selection_rot[3 * c], op->rotation[3 + r] * op->selection_rot[3 * c + 1], op->rotation[6 + r] * op->selection_rot[3 * c + 2]);
	for (int i = 0; i < 3; ++i) {                                 // warp_box: relative_pos(center), scale / diag
		dc[i] = (dc[i] - d->aabb_min[i]) / diag[i];
		ds[i] = ds[i] / diag[i];
		sc[i] = (op->selection_center[i] - d->aabb_min[i]) / diag[i];
		ss[i] = op->selection_scale[i] / diag[i];
	}
	abox_finish(dc, ds, drot, e->a_dst);
	abox_finish(sc, ss, op->selection_rot, e->a_sel);
	e->a_translation = v3(op->translation[0] / diag[0], op->translation[1] / diag[1], op->translation[2] / diag[2]);
	e->a_scale = v3(op->scale[0], op->scale[1], op->scale[2]);
	memcpy(e->a_rot, op->rotation, sizeof(e->a_rot));
	e->a_hide_original = op->hide_original != 0;
	e->a_correct_dir = op->correct_dir != 0;
	return e;
}
void orc_edit_destroy(void* e) { delete (Edit*)e; }
void orc_edit_map_rays(void* edit, uint32_t n, float* coords7, uint8_t* empty) {
	const Edit& e = *(Edit*)edit;
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t)n; ++i) map_ray_one(e, coords7 + (size_t)i * 7, empty + i);
}
void orc_edit_map_positions(void* edit, uint32_t n, float* pos, uint32_t ld, uint8_t* empty) {
	const Edit& e = *(Edit*)edit;
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t)n; ++i) map_position_one(e, pos + (size_t)i * ld, empty + i);
}

struct orc_render_stats { uint64_t generated, composited; uint32_t n_alive0, n_hit, iterations, pad; };
// frame must be pre-cleared by the caller; depth/steps are written for owned pixels.
// fixed_S: 0 = the reference's dynamic S, else force n_steps_between_compaction (S-invariance tests).
static int effective_cores();
void orc_render(void* model, const nrs_render_params* p, void* const* edits, int n_edits, float* frame, float* depth, uint32_t* steps,
                orc_render_stats* stats, int fixed_S, int n_threads) {
#ifdef _OPENMP
	omp_set_num_threads(n_threads > 0 ? n_threads : effective_cores()); // 0 = every core this process may use (and not whatever an earlier call left behind)
#endif
	RenderStats rs{};
	render(*(Model*)model, *p, (const Edit* const*)edits, n_edits, frame, depth, steps, &rs, fixed_S);
	if (stats) { stats->generated = rs.generated; stats->composited = rs.composited; stats->n_alive0 = rs.n_alive0; stats->n_hit = rs.n_hit; stats->iterations = rs.iterations; stats->pad = 0; }
}
// The cores this process may actually use: the affinity mask, cut to the cgroup's CPU quota (cpu.max of cgroup v2, cpu.cfs_quota_us / cpu.cfs_period_us
// of v1).  A container that sees 256 logical CPUs under a 16-core quota -- the GPU boxes of this project -- is throttled to a crawl when OpenMP starts
// 256 busy threads: every parallel region of this file runs with this count (set once when the library is loaded, and by orc_render's n_threads = 0).
static int effective_cores() {
	int n = 1;
#ifdef _OPENMP
	n = omp_get_num_procs();
#endif
	double quota = -1.0, period = 100000.0;
	if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
		char q[64] = {0};
		if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0) quota = atof(q);
		fclose(f);
	} else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
		if (fscanf(g, "%lf", &quota) != 1) quota = -1.0;
		fclose(g);
		if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lf", &period) != 1) period = 100000.0; fclose(h); }
	}
	if (quota > 0.0 && period > 0.0) n = std::min(n, std::max(1, (int)ceil(quota / period)));
	return std::max(n, 1);
}
#ifdef _OPENMP
static const int g_threads_at_load = []() { const int n = effective_cores(); if (!getenv("OMP_NUM_THREADS")) omp_set_num_threads(n); return n; }();
#endif
int orc_max_threads(void) { return effective_cores(); }

// per listed pixel: the (t, dt) stream of init -> jitter -> first hit -> successive samples, ignoring compositing
void orc_trace_samples(void* model, const nrs_render_params* p, uint32_t n_pixels, const uint32_t* pixel_idx, uint32_t max_samples,
                       float* t_out, float* dt_out, uint32_t* count_out) {
	const Model& m = *(Model*)model;
	const uint32_t W = (uint32_t)p->resolution[0];
	const Box render_aabb{v3(p->render_aabb_min[0], p->render_aabb_min[1], p->render_aabb_min[2]),
	                      v3(p->render_aabb_max[0], p->render_aabb_max[1], p->render_aabb_max[2])};
	const uint8_t* grid = m.bitfield.data();
#pragma omp parallel for schedule(dynamic, 64)
	for (int64_t k = 0; k < (int64_t)n_pixels; ++k) {
		uint32_t idx = pixel_idx[k];
		Payload pl;
		float d0;
		init_ray(*p, idx % W, idx / W, pl, d0);
		advance_pos(*p, grid, pl, idx);
		uint32_t cnt = 0;
		if (pl.alive) {
			V3 origin = pl.origin, dir = pl.dir;
			V3 idir = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
			float t = pl.t;
			while (cnt < max_samples) {
				V3 pos;
				float dt = 0.f;
				bool exited = false;
				while (1) {
					pos = origin + dir * t;
					if (!box_contains(render_aabb, pos)) { exited = true; break; }
					dt = calc_dt(t, p->cone_angle_constant);
					uint32_t mip = std::max(p->min_mip, (uint32_t)mip_from_dt(dt, pos));
					if (density_grid_occupied_at(pos, grid, mip)) break;
					t = advance_to_next_voxel(t, p->cone_angle_constant, pos, dir, idir, GRID >> mip);
				}
				if (exited) break;
				t_out[(size_t)k * max_samples + cnt] = t;
				dt_out[(size_t)k * max_samples + cnt] = dt;
				++cnt;
				t += dt;
			}
		}
		count_out[k] = cnt;
	}
}

// ---- authoring restatements -------------------------------------------------------------------------
void* orc_tet_lut_build(const float* verts, uint32_t n_vertices, const uint32_t* tets, uint32_t n_tets) {
	(void)n_vertices;
	TetLut* l = new TetLut();
	build_tet_lut((const V3*)verts, tets, n_tets, *l);
	return l;
}
uint32_t orc_tet_lut_n_idx(void* l) { return (uint32_t)((TetLut*)l)->idx.size(); }
uint32_t orc_tet_lut_max_per_cell(void* l) { return ((TetLut*)l)->max_per_cell; }
void orc_tet_lut_copy(void* l, uint32_t* offsets, uint32_t* idx, uint8_t* bitfield) {
	TetLut* t = (TetLut*)l;
	if (offsets) memcpy(offsets, t->offsets.data(), t->offsets.size() * 4);
	if (idx) memcpy(idx, t->idx.data(), t->idx.size() * 4);
	if (bitfield) memcpy(bitfield, t->bitfield.data(), t->bitfield.size());
}
void orc_tet_lut_destroy(void* l) { delete (TetLut*)l; }

void orc_mvc_compute(const float* cage_v, uint32_t n_cv, const uint32_t* tris, uint32_t n_tris, const float* pts, uint32_t n_pts,
                     float* weights, uint8_t* labels) {
	std::vector<float> ww, d;
	std::vector<V3> u;
	for (uint32_t i = 0; i < n_pts; ++i) {
		bool degenerate = mvc_one(v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), tris, n_tris, (const V3*)cage_v, n_cv, weights + (size_t)i * n_cv, ww, d, u);
		// Cage::compute_mvc marks labels[i] = 1 when the call returns false (cage.cu:19-21)
		if (labels) labels[i] = degenerate ? 0 : 1;
	}
}
void orc_mvc_apply(const float* weights, const float* cage_v, uint32_t n_cv, uint32_t n_pts, float* out) { // cage.cu:38-49
	for (uint32_t i = 0; i < n_pts; ++i) {
		V3 p = v3(0, 0, 0);
		for (uint32_t v = 0; v < n_cv; ++v) p = p + weights[(size_t)i * n_cv + v] * v3(cage_v[3 * v], cage_v[3 * v + 1], cage_v[3 * v + 2]);
		out[3 * i] = p.x; out[3 * i + 1] = p.y; out[3 * i + 2] = p.z;
	}
}
void orc_tet_local_rotations(const float* verts, const float* orig, const uint32_t* tets, uint32_t n_tets, float* out) { // tet_mesh.cu:37-74
	for (uint32_t i = 0; i < n_tets; ++i) {
		V3 cc = v3(0, 0, 0), dc = v3(0, 0, 0);
		for (int j = 0; j < 4; ++j) {
			uint32_t v = tets[4 * i + j];
			cc = cc + v3(orig[3 * v], orig[3 * v + 1], orig[3 * v + 2]);
			dc = dc + v3(verts[3 * v], verts[3 * v + 1], verts[3 * v + 2]);
		}
		cc = v3(cc.x / 4.f, cc.y / 4.f, cc.z / 4.f); dc = v3(dc.x / 4.f, dc.y / 4.f, dc.z / 4.f);
		float A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
		for (int j = 0; j < 4; ++j) {
			uint32_t v = tets[4 * i + j];
			V3 a = v3(orig[3 * v], orig[3 * v + 1], orig[3 * v + 2]) - cc, b = v3(verts[3 * v], verts[3 * v + 1], verts[3 * v + 2]) - dc;
			float av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
			for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[r][c] += av[r] * bv[c]; // corr_mat += (orig - c)(def - c)^T, fp32 like Eigen::Matrix3f
		}
		float U[3][3], V[3][3];
		ref_svd_uv(A, U, V);                                          // svd_eigen, svd3.h:405-420
		for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {     // R = U V^T, column-major out
			out[9 * (size_t)i + 3 * c + r] = sum3(U[r][0] * V[c][0], U[r][1] * V[c][1], U[r][2] * V[c][2]); // Eigen's 3-term reduction order
		}
	}
}

// update_density_grid_mean_and_bitfield, tn:3642-3657 + grid_to_bitfield tn:514-532 + bitfield_max_pool tn:534-555
void orc_density_grid_to_bitfield(const float* grid, uint8_t* bitfield) {
	double mean = 0.0;
	for (uint32_t i = 0; i < GRIDVOL; ++i) mean += (double)(fmaxf(grid[i], 0.f) / (float)GRIDVOL);
	float thresh = std::min(0.01f, (float)mean);
	for (uint32_t i = 0; i < GRIDVOL / 8 * CASCADES; ++i) {
		uint8_t bits = 0;
		for (uint32_t j = 0; j < 8; ++j) bits |= grid[(size_t)i * 8 + j] > thresh ? (uint8_t)(1u << j) : 0;
		bitfield[i] = bits;
	}
	for (uint32_t level = 1; level < CASCADES; ++level) {
		const uint8_t* prev = bitfield + grid_mip_offset(level - 1) / 8;
		uint8_t* next = bitfield + grid_mip_offset(level) / 8;
		for (uint32_t i = 0; i < GRIDVOL / 64; ++i) {
			uint8_t bits = 0;
			for (uint32_t j = 0; j < 8; ++j) bits |= prev[i * 8 + j] > 0 ? (uint8_t)(1u << j) : 0;
			uint32_t x = morton3D_invert(i >> 0) + GRID / 8, y = morton3D_invert(i >> 1) + GRID / 8, z = morton3D_invert(i >> 2) + GRID / 8;
			next[morton3D(x, y, z)] |= bits;
		}
	}
}
// ---- deformed-space occupancy refresh: Testbed::update_density_grid_nerf_operator, tn:3533-3640 -----------------------
// tcnn::pcg32 (dependencies/tiny-cuda-nn/include/tiny-cuda-nn/common_device.h at the submodule pin, ABSENT from the
// checkout): M.E. O'Neill's PCG32 XSH-RR as published (pcg-random.org, Apache-2.0 "pcg32.h" by W. Jakob), restated:
//   state' = state * 0x5851f42d4c957f2d + inc;  out = ror32(((state >> 18) ^ state) >> 27, state >> 59)
//   pcg32(seed): state = 0, inc = (1 << 1) | 1, next, state += seed, next
//   next_float = bit_cast<float>((next_uint() >> 9) | 0x3f800000) - 1
//   advance(delta) = LCG skip-ahead (Brown, "Random number generation with arbitrary strides");  tcnn's default delta is 2^32
void orc_pcg32_seed(uint64_t seed, uint64_t* state, uint64_t* inc) {
	Pcg32 r{0u, (1u << 1u) | 1u};
	r.next_uint();
	r.state += seed;
	r.next_uint();
	*state = r.state; *inc = r.inc;
}
uint32_t orc_pcg32_next_uint(uint64_t* state, uint64_t inc) { Pcg32 r{*state, inc}; uint32_t v = r.next_uint(); *state = r.state; return v; }
float orc_pcg32_next_float(uint64_t* state, uint64_t inc) { Pcg32 r{*state, inc}; float v = r.next_float(); *state = r.state; return v; }
void orc_pcg32_advance(uint64_t* state, uint64_t inc, uint64_t delta) { Pcg32 r{*state, inc}; r.advance(delta); *state = r.state; }

// __hadd: the exact sum of two halfs rounded ONCE to half.  The sum is exact in double; it is brought to float with
// round-to-odd (sticky bit) so that the final round-to-nearest-even to 11 bits cannot double-round.

// generate_grid_samples_nerf_nonuniform, cn:179-208.  Returns the cell index; pos_out = warped position.
static uint32_t generate_grid_sample(Pcg32 rng, uint32_t i, uint32_t n_elements, uint32_t step, const Box& aabb, const float* grid_in,
                                     uint32_t n_cascades, float thresh, V3& pos_out) {
	rng.advance((uint64_t)(int64_t)(uint32_t)(i * 4u));
	uint32_t level = (uint32_t)(rng.next_float() * n_cascades) % n_cascades;
	uint32_t idx = 0;
	for (uint32_t j = 0; j < 10; ++j) {
		idx = ((i + step * n_elements) * 56924617u + j * 19349663u + 96925573u) % GRIDVOL;
		idx += level * GRIDVOL;
		if (grid_in[idx] > thresh) break;
	}
	uint32_t pos_idx = idx % GRIDVOL;
	uint32_t x = morton3D_invert(pos_idx >> 0), y = morton3D_invert(pos_idx >> 1), z = morton3D_invert(pos_idx >> 2);
	float rx = rng.next_float(), ry = rng.next_float(), rz = rng.next_float();
	float s = scalbnf(1.0f, (int)level);
	V3 pos = {(((float)x + rx) / (float)GRID - 0.5f) * s + 0.5f, (((float)y + ry) / (float)GRID - 0.5f) * s + 0.5f,
	          (((float)z + rz) / (float)GRID - 0.5f) * s + 0.5f};
	pos_out = warp_position(pos, aabb);
	return idx;
}

// compute_poisson_residual_density_kernel, cage_deformation.cu:341-384: barycentric residual density of the tet of the
// DEFORMED mesh that contains the (already mapped) position; 0 contribution if none.  Returns true if a tet was found.
static bool poisson_residual_density_one(const Edit& e, V3 warped_pos, float* residual) {
	V3 pos = unwarp_position(warped_pos, e.aabb);
	if (!box_contains(e.bbox, pos)) return false;
	int level = mip_from_pos(pos);
	uint32_t cell = (uint32_t)level * GRIDVOL + cascaded_grid_idx_at(pos, (uint32_t)level);
	for (uint32_t j = e.lut_off[cell]; j < e.lut_off[cell + 1]; ++j) {
		uint32_t t = e.lut_idx[j];
		const uint32_t* tv = &e.tets[4 * (size_t)t];
		V3 a = e.verts[tv[0]], b = e.verts[tv[1]], c = e.verts[tv[2]], d = e.verts[tv[3]];
		if (point_in_tet(a, b, c, d, pos)) {
			float bc[4];
			bary_tet(a, b, c, d, pos, bc);
			*residual = ((bc[0] * e.res_density[tv[0]] + bc[1] * e.res_density[tv[1]]) + bc[2] * e.res_density[tv[2]]) + bc[3] * e.res_density[tv[3]];
			return true;
		}
	}
	return false;
}

// One call of update_density_grid_nerf_operator.  grid: [5*128^3] in/out; bitfield: [NRS_BITFIELD_BYTES] out.
// u: nrs_grid_update (include/nrs.h), rng / ema_step advanced as the reference advances m_rng / density_grid_ema_step.
// Deviation, stated: the reference launches clear_empty_space / compute_poisson_residual_density over n_elements =
// 5*128^3 threads although only n_samples positions exist (tn:3606, 3622) -- an out-of-bounds read when max_cascade < 4;
// here both run over the n_samples samples that exist.
void orc_update_density_grid(void* model, void* const* edits, int n_edits, float* grid, uint8_t* bitfield, nrs_grid_update* u) {
	const Model& m = *(Model*)model;
	const uint32_t n_elements = GRIDVOL * CASCADES;
	if (u->reset_grid) memset(grid, 0, sizeof(float) * n_elements);
	std::vector<uint32_t> tmp(n_elements, 0u); // density_grid_tmp, float bits (atomicMax on uint)
	const uint32_t n_cascades = u->max_cascade + 1;
	Pcg32 rng0{u->rng_state, u->rng_inc};
	Pcg32 rng1 = rng0;
	rng1.advance(1ull << 32);
	const uint32_t n_total = u->n_uniform_samples + u->n_nonuniform_samples;
	std::vector<uint32_t> cell(n_total);
	std::vector<uint32_t> val(n_total);
#pragma omp parallel for schedule(static)
	for (int64_t s = 0; s < (int64_t)n_total; ++s) {
		const bool uni = (uint32_t)s < u->n_uniform_samples;
		const uint32_t i = uni ? (uint32_t)s : (uint32_t)s - u->n_uniform_samples;
		V3 wpos;
		uint32_t idx = generate_grid_sample(uni ? rng0 : rng1, i, uni ? u->n_uniform_samples : u->n_nonuniform_samples, u->ema_step, m.aabb, grid,
		                                    n_cascades, uni ? -0.01f : 0.01f, wpos);
		float p[3] = {wpos.x, wpos.y, wpos.z};
		uint8_t empty = 0;
		for (int k = n_edits - 1; k >= 0; --k) map_position_one(*(const Edit*)edits[k], p, &empty);
		uint16_t feat[32], o[16];
		hashgrid_encode_one(m, p, feat);
		density_mlp_one(m, feat, o);
		uint16_t raw = o[0];
		(void)empty; // clear_empty_space (tn:2759-2770) is launched here (tn:3606) but its body is commented out in the reference: the mask changes nothing,
		             // a sample whose position falls into vacated space keeps the density of the place it stands on
		uint16_t act = f2h(network_to_density(h2f(raw), m.desc.density_activation)); // activate_network_density, tn:3522
		for (int k = n_edits - 1; k >= 0; --k) {
			const Edit& e = *(const Edit*)edits[k];
			float r;
			if (e.apply_poisson && poisson_residual_density_one(e, v3(p[0], p[1], p[2]), &r)) act = hadd(act, f2h(r)); // __half += (__half)r
		}
		float thickness = h2f(act) * MIN_STEP; // scalbnf(MIN_CONE_STEPSIZE(), 0)
		uint32_t bits;
		memcpy(&bits, &thickness, 4);
		cell[s] = idx;
		val[s] = bits;
	}
	for (uint32_t s = 0; s < n_total; ++s) tmp[cell[s]] = std::max(tmp[cell[s]], val[s]); // atomicMax on the bit pattern, tn:447-462
	for (uint32_t i = 0; i < n_elements; ++i) { // ema_grid_samples_nerf, tn:483-506
		float importance;
		memcpy(&importance, &tmp[i], 4);
		float prev = grid[i];
		grid[i] = (prev < 0.f) ? prev : fmaxf(prev * u->decay, importance);
	}
	rng0.advance(2ull << 32); // m_rng.advance() twice
	u->rng_state = rng0.state;
	u->ema_step += 1;
	orc_density_grid_to_bitfield(grid, bitfield);
}

float orc_density_grid_threshold(const float* grid) {
	double mean = 0.0;
	for (uint32_t i = 0; i < GRIDVOL; ++i) mean += (double)(fmaxf(grid[i], 0.f) / (float)GRIDVOL);
	return std::min(0.01f, (float)mean);
}

// ---- array probes with the signatures of oracle/ref_render.cpp's ref_* probes (tests/test_ref_pin.py drives both with one code path) --------
void orc_p_bary_tet(uint32_t n, const float* abcd12, const float* p3, float* out4) {
	for (uint32_t i = 0; i < n; ++i) orc_bary_tet(abcd12 + 12 * (size_t)i, p3 + 3 * (size_t)i, out4 + 4 * (size_t)i);
}
void orc_p_point_in_tet(uint32_t n, const float* abcd12, const float* p3, uint8_t* out) {
	for (uint32_t i = 0; i < n; ++i) out[i] = (uint8_t)orc_point_in_tet(abcd12 + 12 * (size_t)i, p3 + 3 * (size_t)i);
}
void orc_p_ld_random_val(uint32_t n, const uint32_t* index, const uint32_t* seed, float* out) {
	for (uint32_t i = 0; i < n; ++i) out[i] = ld_random_val(index[i], seed[i]);
}
void orc_p_ld_random_pixel_offset(uint32_t n, const uint32_t* spp, float* out2) {
	for (uint32_t i = 0; i < n; ++i) ld_random_pixel_offset(spp[i], out2 + 2 * (size_t)i);
}
void orc_p_sobol(uint32_t n, const uint32_t* index, uint32_t dim, uint32_t* out) {
	for (uint32_t i = 0; i < n; ++i) out[i] = sobol(index[i], dim);
}
void orc_p_ray_intersect(uint32_t n, const float* box6, const float* o3, const float* d3, float* out2, uint8_t* contains_o) {
	for (uint32_t i = 0; i < n; ++i) {
		const float* q = box6 + 6 * (size_t)i;
		Box b{v3(q[0], q[1], q[2]), v3(q[3], q[4], q[5])};
		V3 o = v3(o3[3 * (size_t)i], o3[3 * (size_t)i + 1], o3[3 * (size_t)i + 2]);
		ray_intersect(b, o, v3(d3[3 * (size_t)i], d3[3 * (size_t)i + 1], d3[3 * (size_t)i + 2]), out2[2 * (size_t)i], out2[2 * (size_t)i + 1]);
		contains_o[i] = box_contains(b, o) ? 1 : 0;
	}
}
void orc_p_box_intersects_triangle(uint32_t n, const float* box6, const float* tri9, uint8_t* out) {
	for (uint32_t i = 0; i < n; ++i) out[i] = (uint8_t)orc_box_intersects_triangle(box6 + 6 * (size_t)i, tri9 + 9 * (size_t)i);
}
void orc_p_grid_math(uint32_t n, const float* pos3, const float* dir3, const float* t, const float* cone, const uint32_t* mip, float* calc_dt_out, int32_t* mip_from_pos_out,
                     int32_t* mip_from_dt_out, uint32_t* cell_idx_out, float* dist_out, float* advance_out) {
	for (uint32_t i = 0; i < n; ++i) {
		const V3 pos = v3(pos3[3 * (size_t)i], pos3[3 * (size_t)i + 1], pos3[3 * (size_t)i + 2]), dir = v3(dir3[3 * (size_t)i], dir3[3 * (size_t)i + 1], dir3[3 * (size_t)i + 2]);
		const V3 idir = v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
		const float dt = calc_dt(t[i], cone[i]);
		calc_dt_out[i] = dt;
		mip_from_pos_out[i] = mip_from_pos(pos);
		mip_from_dt_out[i] = mip_from_dt(dt, pos);
		cell_idx_out[i] = cascaded_grid_idx_at(pos, mip[i]);
		const uint32_t res = GRID >> mip[i];
		dist_out[i] = distance_to_next_voxel(pos, dir, idir, res);
		advance_out[i] = advance_to_next_voxel(t[i], cone[i], pos, dir, idir, res);
	}
}
void orc_p_warp(uint32_t n, const float* box6, const float* pos3, const float* dt, float* warp_pos3, float* unwarp_pos3, float* warp_dir3, float* unwarp_dir3, float* warp_dt_out,
                float* unwarp_dt_out) {
	const Box b{v3(box6[0], box6[1], box6[2]), v3(box6[3], box6[4], box6[5])};
	for (uint32_t i = 0; i < n; ++i) {
		const V3 p = v3(pos3[3 * (size_t)i], pos3[3 * (size_t)i + 1], pos3[3 * (size_t)i + 2]);
		const V3 a = warp_position(p, b), u = unwarp_position(p, b), wd = warp_direction(p), ud = unwarp_direction(p);
		const V3 r[4] = {a, u, wd, ud};
		float* outs[4] = {warp_pos3, unwarp_pos3, warp_dir3, unwarp_dir3};
		for (int k = 0; k < 4; ++k) { outs[k][3 * (size_t)i] = r[k].x; outs[k][3 * (size_t)i + 1] = r[k].y; outs[k][3 * (size_t)i + 2] = r[k].z; }
		warp_dt_out[i] = warp_dt(dt[i]);
		unwarp_dt_out[i] = unwarp_dt(dt[i]);
	}
}
void orc_p_evaluate_sh9(uint32_t n, const float* sh27, const float* dir3, float* rgb3) {
	for (uint32_t i = 0; i < n; ++i) orc_evaluate_sh9(sh27 + 27 * (size_t)i, dir3 + 3 * (size_t)i, rgb3 + 3 * (size_t)i);
}
void orc_p_activations(uint32_t n, const float* x, float* srgb_to_linear_out, float* rgb_logistic, float* rgb_exp, float* density_exp) {
	for (uint32_t i = 0; i < n; ++i) {
		srgb_to_linear_out[i] = srgb_to_linear(x[i]);
		rgb_logistic[i] = network_to_rgb(x[i], NRS_ACT_LOGISTIC);
		rgb_exp[i] = network_to_rgb(x[i], NRS_ACT_EXPONENTIAL);
		density_exp[i] = network_to_density(x[i], NRS_ACT_EXPONENTIAL);
	}
}
// pixel_to_ray (common_device.cuh:245-295; camera_matrix1, no distortion; focus_z = slice_plane_z, dof): the un-normalised direction
void orc_p_pixel_to_ray(uint32_t n, const int32_t* pixel2, const nrs_render_params* p, float* origin3, float* dir3) {
	for (uint32_t i = 0; i < n; ++i) {
		V3 o, d;
		pixel_to_ray(*p, p->camera_matrix1, (uint32_t)pixel2[2 * (size_t)i], (uint32_t)pixel2[2 * (size_t)i + 1], p->slice_plane_z, p->dof, o, d);
		origin3[3 * (size_t)i] = o.x; origin3[3 * (size_t)i + 1] = o.y; origin3[3 * (size_t)i + 2] = o.z;
		dir3[3 * (size_t)i] = d.x; dir3[3 * (size_t)i + 1] = d.y; dir3[3 * (size_t)i + 2] = d.z;
	}
}
void orc_p_cell_functions(uint32_t n, const uint32_t* xyz_level4, const float* pos3, float* cell_pos3, int32_t* cell_at_pos3) {
	for (uint32_t i = 0; i < n; ++i) {
		const uint32_t* q = xyz_level4 + 4 * (size_t)i;
		V3 c = get_cell_pos(q[0], q[1], q[2], q[3]);
		cell_pos3[3 * (size_t)i] = c.x; cell_pos3[3 * (size_t)i + 1] = c.y; cell_pos3[3 * (size_t)i + 2] = c.z;
		int a[3];
		get_cell_at_pos(v3(pos3[3 * (size_t)i], pos3[3 * (size_t)i + 1], pos3[3 * (size_t)i + 2]), q[3], a);
		for (int k = 0; k < 3; ++k) cell_at_pos3[3 * (size_t)i + k] = a[k];
	}
}
// per listed pixel: the network-input records (warped position, warped dt, warped direction) of every generated sample and payload.t after it
void orc_p_trace_coords(const nrs_model_desc* desc, const nrs_render_params* p, const uint8_t* grid, uint32_t n_pixels, const uint32_t* pixel_idx, uint32_t max_samples,
                        float* coords_out, float* t_after_out, uint32_t* count_out, float* origin_dir_t0_out) {
	const uint32_t W = (uint32_t)p->resolution[0];
	const Box render_aabb{v3(p->render_aabb_min[0], p->render_aabb_min[1], p->render_aabb_min[2]), v3(p->render_aabb_max[0], p->render_aabb_max[1], p->render_aabb_max[2])};
	const Box train_aabb{v3(desc->aabb_min[0], desc->aabb_min[1], desc->aabb_min[2]), v3(desc->aabb_max[0], desc->aabb_max[1], desc->aabb_max[2])};
#pragma omp parallel for schedule(dynamic, 64)
	for (int64_t k = 0; k < (int64_t)n_pixels; ++k) {
		uint32_t idx = pixel_idx[k];
		Payload pl;
		memset(&pl, 0, sizeof(pl));
		float d0;
		init_ray(*p, idx % W, idx / W, pl, d0);
		advance_pos(*p, grid, pl, idx);
		if (origin_dir_t0_out) {
			float* o = origin_dir_t0_out + 7 * (size_t)k;
			o[0] = pl.origin.x; o[1] = pl.origin.y; o[2] = pl.origin.z; o[3] = pl.dir.x; o[4] = pl.dir.y; o[5] = pl.dir.z; o[6] = pl.t;
		}
		uint32_t cnt = 0;
		if (pl.alive) {
			V3 origin = pl.origin, dir = pl.dir;
			V3 idir = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
			float t = pl.t;
			while (cnt < max_samples) {
				V3 pos;
				float dt = 0.f;
				bool exited = false;
				while (1) {
					pos = origin + dir * t;
					if (!box_contains(render_aabb, pos)) { exited = true; break; }
					dt = calc_dt(t, p->cone_angle_constant);
					uint32_t mip = std::max(p->min_mip, (uint32_t)mip_from_dt(dt, pos));
					if (density_grid_occupied_at(pos, grid, mip)) break;
					t = advance_to_next_voxel(t, p->cone_angle_constant, pos, dir, idir, GRID >> mip);
				}
				if (exited) break;
				float* c = coords_out + ((size_t)k * max_samples + cnt) * 7;
				V3 wp = warp_position(pos, train_aabb), wd = warp_direction(dir);
				c[0] = wp.x; c[1] = wp.y; c[2] = wp.z; c[3] = warp_dt(dt); c[4] = wd.x; c[5] = wd.y; c[6] = wd.z;
				t += dt;
				t_after_out[(size_t)k * max_samples + cnt] = t;
				++cnt;
			}
		}
		count_out[k] = cnt;
	}
}

// compute_residual_poisson_kernel on a caller batch (one sample per call site; outputs pre-zeroed like the tracer's memsets, tn:2866-2869)
void orc_p_poisson_residuals(void* edit, uint32_t n, const float* coords7, float* sh27, float* out_density, float* res_density) {
	const Edit& e = *(Edit*)edit;
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t)n; ++i) {
		memset(sh27 + 27 * (size_t)i, 0, 108);
		out_density[i] = 0.f; res_density[i] = 0.f;
		if (e.apply_poisson) poisson_residual_one(e, coords7 + 7 * (size_t)i, sh27 + 27 * (size_t)i, out_density + i, res_density + i);
	}
}

// GrowingSelection::interpolate_poisson_boundary, growing_selection.cu:2350-2395: the MVC-weighted (gamma_coordinates) transfer of the cage
// vertices' inside / outside membrane terms (compute_poisson_boundary) to the tet vertices.  Host code in the reference: std::exp(float) is
// the host libm's expf; `boundary_shs /= sh_weights_sum + 1e-6` adds in double and divides by the sum rounded to float (Eigen converts the
// scalar to the matrix' scalar type).  Pinned by the reference's own loop (oracle/ref_render.cpp: ref_poisson_interpolate).
void orc_poisson_interpolate(const float* gamma, uint32_t n_tet_vertices, uint32_t n_cage_vertices, const float* inside_density, const float* outside_density,
                             const float* inside_shs27, const float* outside_shs27, float* boundary_shs27_out, float* outside_density_out, float* residual_density_out) {
	for (uint32_t i = 0; i < n_tet_vertices; ++i) {
		float sh[27] = {0}, sh_weights_sum = 0.f, od = 0.f, rd = 0.f;
		for (uint32_t j = 0; j < n_cage_vertices; ++j) {
			const float g = gamma[(size_t)i * n_cage_vertices + j];
			const float alpha_out = 1 - expf(-outside_density[j] * MIN_STEP);
			const float alpha_in = 1 - expf(-inside_density[j] * MIN_STEP);
			const float w_outside = 1.f, w_inside = std::min(alpha_in / alpha_out, 1.f);
			sh_weights_sum += g * alpha_out;
			for (int k = 0; k < 27; ++k) {
				const float sh_diff = w_outside * outside_shs27[27 * (size_t)j + k] - w_inside * inside_shs27[27 * (size_t)j + k];
				sh[k] += (g * alpha_out) * sh_diff;
			}
			od += g * outside_density[j];
			rd += g * (outside_density[j] - inside_density[j]);
		}
		const float denom = (float)((double)sh_weights_sum + 1e-6);
		for (int k = 0; k < 27; ++k) boundary_shs27_out[27 * (size_t)i + k] = sh[k] / denom;
		outside_density_out[i] = od;
		residual_density_out[i] = std::max(rd, 0.f);
	}
}

} // extern "C"
