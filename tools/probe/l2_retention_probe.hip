// l2_retention_probe: can a cache policy on STREAMING gathers keep a small HOT table resident in the L2 of gfx950?  (round 6, garden scene: the four hashed levels of an
// aabb-16 frame -- 2 MB each -- compete for the 4 MB L2 of an XCD with 56 GB of record lines per frame that are used once.)
// Every lane issues, per round, HOT_PER loads of 4 B at random offsets of a hot table of HOT_MB megabytes (plain loads) and COLD_PER loads of 32 B at random offsets of a
// cold table of COLD_MB megabytes with policy P: 0 plain, 1 nt, 2 sc1 (agent-scope atomic 8 B x 4), 3 sc0 sc1, 4 = `buffer_load ... nt` through a buffer descriptor,
// 5 = `buffer_load ... sc1 nt`, 6 = `buffer_load ... sc0 sc1 nt`, 7 = first 16 B plain + second 16 B nt (the line is allocated by the first load; does the nt HIT demote it?),
// 8 = first 16 B nt + second 16 B plain, 9 = ONE 16-byte nt load per gather (half a record: what a no-allocate policy costs without the second fetch).
//   l2_retention_probe <hot MB> <cold MB> <policy> [rounds = 64] [hot loads per round = 8] [cold per round = 8] [cold alloc: 0 hipMalloc, 1 uncached, 2 fine-grained]
// Run under rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum: misses beyond the cold gathers' own (lanes x rounds x COLD_PER) are hot-table misses.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
	x += 0x9e3779b97f4a7c15ull;
	x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
	x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
	return x ^ (x >> 31);
}
typedef uint32_t u32x4n __attribute__((ext_vector_type(4)));

template <int POLICY>
__device__ __forceinline__ uint32_t cold_load(const uint8_t* base, uint64_t slot, __amdgpu_buffer_rsrc_t rsrc, uint64_t window_mask) {
	const uint8_t* p = base + slot * 32;
	if (POLICY == 0) { const u32x4n a = reinterpret_cast<const u32x4n*>(p)[0], b = reinterpret_cast<const u32x4n*>(p)[1]; return a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w; }
	if (POLICY == 1) { const u32x4n a = __builtin_nontemporal_load(reinterpret_cast<const u32x4n*>(p)), b = __builtin_nontemporal_load(reinterpret_cast<const u32x4n*>(p) + 1); return a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w; }
	if (POLICY == 7) { const u32x4n a = reinterpret_cast<const u32x4n*>(p)[0], b = __builtin_nontemporal_load(reinterpret_cast<const u32x4n*>(p) + 1); return a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w; }
	if (POLICY == 8) { const u32x4n a = __builtin_nontemporal_load(reinterpret_cast<const u32x4n*>(p)), b = reinterpret_cast<const u32x4n*>(p)[1]; return a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w; }
	if (POLICY == 9) { const u32x4n a = __builtin_nontemporal_load(reinterpret_cast<const u32x4n*>(p)); return a.x ^ a.y ^ a.z ^ a.w; }
	if (POLICY == 2 || POLICY == 3) {
		uint64_t v = 0;
		#pragma unroll
		for (int k = 0; k < 4; ++k) v ^= __hip_atomic_load(reinterpret_cast<const uint64_t*>(p) + k, __ATOMIC_RELAXED, POLICY == 2 ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_SYSTEM);
		return (uint32_t)v ^ (uint32_t)(v >> 32);
	}
	// buffer loads: the descriptor covers the first 2 GB of the cold table (32-bit offsets): slots are folded into it
	const uint32_t off = (uint32_t)((slot * 32) & window_mask);
	constexpr int aux = POLICY == 4 ? 2 : (POLICY == 5 ? (2 | 16) : (1 | 2 | 16)); // bit 0 sc0, bit 1 nt, bit 4 sc1 (gfx940+)
	const u32x4n a = __builtin_bit_cast(u32x4n, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, aux));
	const u32x4n b = __builtin_bit_cast(u32x4n, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off + 16, 0, aux));
	return a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
}

template <int POLICY>
__global__ __launch_bounds__(256) void probe_kernel(const uint32_t* __restrict__ hot, uint64_t hot_slots, const uint8_t* __restrict__ cold, uint64_t cold_slots, uint32_t rounds,
                                                    uint32_t hot_per, uint32_t cold_per, uint32_t* __restrict__ out) {
	const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	const uint64_t window = cold_slots * 32 < (1ull << 31) ? cold_slots * 32 : (1ull << 31);
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)cold, 0, (int)window, 0x00020000);
	uint32_t acc = 0;
	for (uint32_t r = 0; r < rounds; ++r) {
		for (uint32_t k = 0; k < hot_per; ++k) acc += hot[mix(tid * 0x10001ull + (uint64_t)r * 64 + k) % hot_slots];
		for (uint32_t k = 0; k < cold_per; ++k) acc += cold_load<POLICY>(cold, mix(tid * 0x20003ull + (uint64_t)r * 64 + 32 + k) % cold_slots, rsrc, window - 1);
	}
	out[tid] = acc;
}

template <int POLICY>
static int run(size_t hot_mb, size_t cold_mb, uint32_t rounds, uint32_t hot_per, uint32_t cold_per, int cold_alloc) {
	const uint32_t lanes = 256u * 2048u;
	uint32_t* d_hot = nullptr; uint8_t* d_cold = nullptr; uint32_t* d_out = nullptr;
	CHECK(hipMalloc((void**)&d_hot, hot_mb << 20));
	CHECK(hipMemset(d_hot, 1, hot_mb << 20));
	if (cold_alloc == 0) CHECK(hipMalloc((void**)&d_cold, cold_mb << 20));
	else CHECK(hipExtMallocWithFlags((void**)&d_cold, cold_mb << 20, cold_alloc == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained));
	CHECK(hipMemset(d_cold, 1, cold_mb << 20));
	CHECK(hipMalloc((void**)&d_out, (size_t)lanes * 4));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL((probe_kernel<POLICY>), dim3(lanes / 256), dim3(256), 0, 0, d_hot, (uint64_t)(hot_mb << 20) / 4, d_cold, (uint64_t)(cold_mb << 20) / 32, rounds, hot_per, cold_per, d_out); // (warm-up = a full launch: every dispatch of a PMC pass is the same)
	CHECK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 3; ++rep) {
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL((probe_kernel<POLICY>), dim3(lanes / 256), dim3(256), 0, 0, d_hot, (uint64_t)(hot_mb << 20) / 4, d_cold, (uint64_t)(cold_mb << 20) / 32, rounds, hot_per, cold_per, d_out);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms = 0;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	printf("{\"cold_alloc\": %d, \"policy\": %d, \"hot_mb\": %zu, \"cold_mb\": %zu, \"rounds\": %u, \"hot_per\": %u, \"cold_per\": %u, \"hot_gathers\": %.0f, \"cold_gathers\": %.0f, \"ms\": %.3f}\n", cold_alloc, POLICY, hot_mb, cold_mb, rounds,
	       hot_per, cold_per, (double)lanes * rounds * hot_per, (double)lanes * rounds * cold_per, best);
	return 0;
}

int main(int argc, char** argv) {
	if (argc < 4) { fprintf(stderr, "usage: l2_retention_probe <hot MB> <cold MB> <policy 0..6> [rounds] [hot per round] [cold per round]\n"); return 2; }
	const size_t hot = strtoull(argv[1], nullptr, 10), cold = strtoull(argv[2], nullptr, 10);
	const int pol = atoi(argv[3]);
	const uint32_t rounds = argc > 4 ? (uint32_t)atoi(argv[4]) : 64u, hp = argc > 5 ? (uint32_t)atoi(argv[5]) : 8u, cp = argc > 6 ? (uint32_t)atoi(argv[6]) : 8u;
	const int ca = argc > 7 ? atoi(argv[7]) : 0;
	switch (pol) {
		case 0: return run<0>(hot, cold, rounds, hp, cp, ca);
		case 1: return run<1>(hot, cold, rounds, hp, cp, ca);
		case 2: return run<2>(hot, cold, rounds, hp, cp, ca);
		case 3: return run<3>(hot, cold, rounds, hp, cp, ca);
		case 4: return run<4>(hot, cold, rounds, hp, cp, ca);
		case 5: return run<5>(hot, cold, rounds, hp, cp, ca);
		case 6: return run<6>(hot, cold, rounds, hp, cp, ca);
		case 7: return run<7>(hot, cold, rounds, hp, cp, ca);
		case 8: return run<8>(hot, cold, rounds, hp, cp, ca);
		case 9: return run<9>(hot, cold, rounds, hp, cp, ca);
	}
	return 2;
}
