// Stand-in for <tiny-cuda-nn/common.h> + the CUDA keywords, so that the reference's __host__ __device__ / __global__ render-path
// sources compile as plain host C++ (g++).  TEST INFRASTRUCTURE: used only to build oracle/_ref.
//
// tiny-cuda-nn is an EMPTY, un-pinned submodule of the reference (fork cjambon/tcnn-pyngp, .gitmodules:16-19).  What follows
// restates the few upstream helpers the render path names (NVlabs/tiny-cuda-nn common.h / common_device.h of that era, from
// memory -- SURVEY App. B): clamp, host_device_swap, logistic, morton3D(_invert), next_multiple, vector_t, PitchedPtr,
// network_precision_t = __half (TCNN_MIN_GPU_ARCH >= 70), batch_size_granularity = 128.
// A CUDA kernel runs here as a host function: threadIdx / blockIdx / blockDim are thread-local variables that the driver's
// launch helper sets before each call (one "thread" per call).
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <math.h>
#include <string>
#include <vector>

// ---- CUDA language shim ---------------------------------------------------------------------------------------------------
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define TCNN_MIN_GPU_ARCH 86

struct uint3 { unsigned int x, y, z; };
struct dim3 { unsigned int x = 1, y = 1, z = 1; };
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
typedef void* cudaStream_t;

extern thread_local uint3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

// fast-math intrinsics: approximate on NVIDIA hardware; the host build uses libm (tolerance, not bits, here)
inline float __expf(float x) { return expf(x); }
inline float __powf(float x, float y) { return powf(x, y); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline unsigned int __float_as_uint(float f) { unsigned int u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned int u) { float f; memcpy(&f, &u, 4); return f; }

// one host thread at a time touches a given element in the drivers, so plain read-modify-write is enough
template <typename T> struct nrs_same { typedef T type; };
template <typename T> inline T atomicAdd(T* p, typename nrs_same<T>::type v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicMax(T* p, typename nrs_same<T>::type v) { T o = *p; *p = o < v ? v : o; return o; }

// CUDA's global min / max overload set (crt/math_functions.hpp)
inline int min(int a, int b) { return a < b ? a : b; }
inline unsigned int min(unsigned int a, unsigned int b) { return a < b ? a : b; }
inline unsigned int min(int a, unsigned int b) { return min((unsigned int)a, b); }
inline unsigned int min(unsigned int a, int b) { return min(a, (unsigned int)b); }
inline float min(float a, float b) { return fminf(a, b); }
inline double min(double a, double b) { return fmin(a, b); }
inline double min(float a, double b) { return fmin((double)a, b); }
inline double min(double a, float b) { return fmin(a, (double)b); }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned int max(unsigned int a, unsigned int b) { return a > b ? a : b; }
inline unsigned int max(int a, unsigned int b) { return max((unsigned int)a, b); }
inline unsigned int max(unsigned int a, int b) { return max(a, (unsigned int)b); }
inline float max(float a, float b) { return fmaxf(a, b); }
inline double max(double a, double b) { return fmax(a, b); }
inline double max(float a, double b) { return fmax((double)a, b); }
inline double max(double a, float b) { return fmax(a, (double)b); }

// ---- __half: IEEE binary16 storage, round-to-nearest-even conversions (what cuda_fp16.h's __float2half_rn does) --------------
struct __half {
	uint16_t bits = 0;
	__half() {}
	__half(float f) { bits = from_float(f); }
	__half(double d) { bits = from_float((float)d); }
	__half(int i) { bits = from_float((float)i); }
	operator float() const { return to_float(bits); }
	static float to_float(uint16_t h) {
		const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
		uint32_t u;
		if (exp == 0) {
			if (man == 0) u = sign;
			else { float v = (float)man * 5.9604644775390625e-08f; memcpy(&u, &v, 4); u |= sign; }
		} else if (exp == 31) u = sign | 0x7f800000u | (man << 13);
		else u = sign | ((exp + 112u) << 23) | (man << 13);
		float f; memcpy(&f, &u, 4); return f;
	}
	static uint16_t from_float(float f) {
		uint32_t x; memcpy(&x, &f, 4);
		const uint32_t sign = (x >> 16) & 0x8000u, ax = x & 0x7fffffffu;
		if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0u));
		if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
		if (ax < 0x33000001u) return (uint16_t)sign;
		const int e = (int)(ax >> 23) - 127;
		const uint32_t m = (ax & 0x7fffffu) | 0x800000u;
		const int shift = (e < -14) ? (13 + (-14 - e)) : 13;
		uint32_t kept = m >> shift;
		const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
		if (rem > half || (rem == half && (kept & 1u))) kept++;
		const uint32_t h = (e < -14) ? kept : (((uint32_t)(e + 15) << 10) + (kept - 0x400u));
		return (uint16_t)(sign | h);
	}
};
// __hadd: one correctly rounded binary16 addition.  The exact sum of two halfs fits a double; it is brought to float with
// round-to-odd (truncate, then OR the sticky bit into the last place) so that the final float -> half rounding cannot double-round.
inline __half operator+(const __half& a, const __half& b) {
	const double s = (double)(float)a + (double)(float)b;
	float f = (float)s;
	if ((double)f != s) {
		if (fabs((double)f) > fabs(s)) f = nextafterf(f, 0.0f);
		uint32_t u; memcpy(&u, &f, 4); u |= 1u; memcpy(&f, &u, 4);
	}
	return __half(f);
}
inline __half& operator+=(__half& a, const __half& b) { a = a + b; return a; }
struct __half2 { __half x, y; };

// ---- tcnn helpers ---------------------------------------------------------------------------------------------------------
namespace tcnn {

using network_precision_t = __half;
static constexpr uint32_t batch_size_granularity = 128;

template <typename T> inline T clamp(T val, T lower, T upper) { return val < lower ? lower : (upper < val ? upper : val); }
template <typename T> inline void host_device_swap(T& a, T& b) { T c(a); a = b; b = c; }
inline float logistic(const float x) { return 1.0f / (1.0f + expf(-x)); }
inline float logit(const float x) { return -logf(1.0f / (fminf(fmaxf(x, 1e-9f), 1.0f - 1e-9f)) - 1.0f); }
template <typename T> inline T div_round_up(T val, T divisor) { return (val + divisor - 1) / divisor; }
template <typename T> inline T next_multiple(T val, T divisor) { return div_round_up(val, divisor) * divisor; }

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

template <typename T, uint32_t N_ELEMS> struct alignas(sizeof(T) * N_ELEMS) vector_t {
	T& operator[](uint32_t idx) { return data[idx]; }
	T operator[](uint32_t idx) const { return data[idx]; }
	T data[N_ELEMS];
	static constexpr uint32_t N = N_ELEMS;
};

template <typename T> struct PitchedPtr {
	PitchedPtr() : ptr{nullptr}, stride_in_bytes{sizeof(T)} {}
	PitchedPtr(T* ptr_, size_t stride_in_elements, size_t offset = 0, size_t extra_stride_bytes = 0)
	    : ptr{ptr_ + offset}, stride_in_bytes{(uint32_t)(stride_in_elements * sizeof(T) + extra_stride_bytes)} {}
	template <typename U> explicit PitchedPtr(PitchedPtr<U> other) : ptr{(T*)other.ptr}, stride_in_bytes{other.stride_in_bytes} {}
	T* operator()(uint32_t y) const { return (T*)((const char*)ptr + y * stride_in_bytes); }
	void operator+=(uint32_t y) { ptr = (T*)((const char*)ptr + y * stride_in_bytes); }
	void operator-=(uint32_t y) { ptr = (T*)((const char*)ptr - y * stride_in_bytes); }
	explicit operator bool() const { return ptr; }
	T* ptr;
	uint32_t stride_in_bytes;
};

} // namespace tcnn
