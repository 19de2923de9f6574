// mixlo_probe.hip -- does v_fma_mixlo_f16 (fp32 x fp16 -> fp16) round the exact product ONCE to fp16, or twice (to fp32, then to fp16) like
// tiny-cuda-nn's `(T)(weight * (float)value)`?  If twice, the per-corner product of the NRS_GRID_ACC_NETWORK interpolation is one instruction.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probe/mixlo_probe tools/probe/mixlo_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void probe(const float* w, const uint32_t* v, uint32_t n, uint32_t* mism_double, uint32_t* mism_single, uint32_t* first_bad) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t r = 0;
	asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(r) : "v"(w[i]), "v"(v[i]));
	const _Float16 got = __builtin_bit_cast(_Float16, (uint16_t)(r & 0xffffu));
	const _Float16 h = __builtin_bit_cast(_Float16, (uint16_t)(v[i] & 0xffffu));
	const float prod = w[i] * (float)h;                         // rounded to fp32
	const _Float16 twice = (_Float16)prod;                       // ... then to fp16: the reference's expression
	// the exact product (24 x 11 bits fit a double) rounded ONCE to fp16: through fp32 with round-to-odd (truncate, set the last bit if inexact)
	const double pd = (double)w[i] * (double)(float)h;
	float fo = (float)pd;
	if ((double)fo != pd) {
		uint32_t u = __float_as_uint(fo);
		if (fabs((double)fo) > fabs(pd)) u -= 1u;
		fo = __uint_as_float(u | 1u);
	}
	const _Float16 once = (_Float16)fo;
	if (__builtin_bit_cast(uint16_t, got) != __builtin_bit_cast(uint16_t, twice)) { if (atomicAdd(mism_double, 1u) == 0) *first_bad = i; }
	if (__builtin_bit_cast(uint16_t, got) != __builtin_bit_cast(uint16_t, once)) atomicAdd(mism_single, 1u);
}

int main() {
	const uint32_t n = 1u << 26;
	std::vector<float> w(n);
	std::vector<uint32_t> v(n);
	uint64_t s = 0x9E3779B97F4A7C15ull;
	auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
	for (uint32_t i = 0; i < n; ++i) {
		const uint64_t a = next();
		w[i] = (float)((a >> 40) & 0xffffff) / 16777216.0f;                // trilinear weights: [0, 1)
		if ((i & 7) == 0) w[i] = (float)((a >> 40) & 0xffffff) / 16777216.0f * 1e-3f; // small weights as well
		uint16_t h = (uint16_t)(a & 0xffff);
		if (((h >> 10) & 31) == 31) h &= 0x3fff;                          // no inf / nan
		v[i] = h | ((uint32_t)(a >> 16) & 0xffff0000u);
	}
	float* dw; uint32_t *dv, *dc;
	(void)hipMalloc(&dw, n * 4); (void)hipMalloc(&dv, n * 4); (void)hipMalloc(&dc, 12);
	(void)hipMemcpy(dw, w.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemset(dc, 0, 12);
	probe<<<n / 256, 256>>>(dw, dv, n, dc, dc + 1, dc + 2);
	uint32_t c[3];
	(void)hipMemcpy(c, dc, 12, hipMemcpyDeviceToHost);
	printf("{\"n\": %u, \"differs_from_double_rounding\": %u, \"differs_from_single_rounding\": %u, \"first_bad\": %u}\n", n, c[0], c[1], c[2]);
	return 0;
}
