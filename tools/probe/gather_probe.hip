// gather_probe: what the MI355X memory system delivers for RANDOM small gathers (the access pattern of hash-grid levels whose samples do not
// share lines: the aabb-16 render path, the occupancy refresh, random-sample inference).  Every lane issues ILP independent loads of BYTES bytes
// at pseudo-random BYTES-aligned offsets of a table of TABLE_MB megabytes, ROUNDS times; the sum is written so nothing is optimised away.
//   gather_probe <table MB> <bytes per gather: 4 | 8 | 16 | 32> <rounds> [lanes = 256 CUs x 2048] [policy]
// policy (round 6, 32-byte gathers only): 0 plain global_load_dwordx4 x 2, 1 `nt` (non-temporal), 2 `sc1` (agent-scope relaxed atomic loads, 4 x 8 B),
// 3 `sc0 sc1` (system scope, 4 x 8 B) -- does any cache policy make the L2 ask the fabric for less than a 128-byte line?
// prints: gathers/s, and the byte rates that correspond to 32 / 64 / 128 B per gather.  Run under rocprofv3 --pmc for the counters.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { // splitmix64
	x += 0x9e3779b97f4a7c15ull;
	x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
	x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
	return x ^ (x >> 31);
}

typedef uint32_t u32x4n __attribute__((ext_vector_type(4)));
template <int POLICY>
__device__ __forceinline__ uint32_t load32(const uint8_t* p) {
	if (POLICY == 1) {
		const u32x4n a = __builtin_nontemporal_load(reinterpret_cast<const u32x4n*>(p)), b = __builtin_nontemporal_load(reinterpret_cast<const u32x4n*>(p) + 1);
		return a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
	}
	uint64_t v = 0;
	#pragma unroll
	for (int k = 0; k < 4; ++k)
		v ^= __hip_atomic_load(reinterpret_cast<const uint64_t*>(p) + k, __ATOMIC_RELAXED, POLICY == 2 ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_SYSTEM);
	return (uint32_t)v ^ (uint32_t)(v >> 32);
}

template <int BYTES, int ILP, int POLICY = 0>
__global__ __launch_bounds__(256) void gather_kernel(const uint8_t* __restrict__ table, uint64_t n_slots, uint32_t rounds, uint32_t* __restrict__ out) {
	const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	uint32_t acc = 0;
	for (uint32_t r = 0; r < rounds; ++r) {
		uint64_t idx[ILP];
		#pragma unroll
		for (int k = 0; k < ILP; ++k) idx[k] = mix(tid * 0x10001ull + (uint64_t)r * ILP + k) % n_slots;
		#pragma unroll
		for (int k = 0; k < ILP; ++k) {
			const uint8_t* p = table + idx[k] * BYTES;
			if (BYTES == 4) acc += *reinterpret_cast<const uint32_t*>(p);
			else if (BYTES == 8) { const uint2 v = *reinterpret_cast<const uint2*>(p); acc += v.x ^ v.y; }
			else if (BYTES == 16) { const uint4 v = *reinterpret_cast<const uint4*>(p); acc += v.x ^ v.y ^ v.z ^ v.w; }
			else if (POLICY != 0) acc += load32<POLICY>(p);
			else { const uint4 a = reinterpret_cast<const uint4*>(p)[0], b = reinterpret_cast<const uint4*>(p)[1]; acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w; }
		}
	}
	out[tid] = acc;
}

template <int BYTES, int POLICY = 0>
static int run(size_t table_mb, uint32_t rounds, uint32_t lanes) {
	constexpr int ILP = 8;
	const size_t bytes = table_mb << 20;
	uint8_t* d_table = nullptr;
	uint32_t* d_out = nullptr;
	CHECK(hipMalloc((void**)&d_table, bytes));
	CHECK(hipMemset(d_table, 1, bytes));
	CHECK(hipMalloc((void**)&d_out, (size_t)lanes * 4));
	const uint64_t n_slots = bytes / BYTES;
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL((gather_kernel<BYTES, ILP, POLICY>), dim3(lanes / 256), dim3(256), 0, 0, d_table, n_slots, 2u, d_out); // warm-up
	CHECK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 3; ++rep) {
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL((gather_kernel<BYTES, ILP, POLICY>), dim3(lanes / 256), dim3(256), 0, 0, d_table, n_slots, rounds, d_out);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms = 0;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	const double gathers = (double)lanes * rounds * ILP, gps = gathers / (best * 1e-3);
	printf("{\"policy\": %d, \"table_mb\": %zu, \"bytes_per_gather\": %d, \"gathers\": %.0f, \"ms\": %.3f, \"ggathers_per_s\": %.2f, \"tb_per_s_if_32B\": %.2f, \"tb_per_s_if_64B\": %.2f, "
	       "\"tb_per_s_if_128B\": %.2f}\n", POLICY, table_mb, BYTES, gathers, best, gps / 1e9, gps * 32 / 1e12, gps * 64 / 1e12, gps * 128 / 1e12);
	(void)hipFree(d_table);
	(void)hipFree(d_out);
	return 0;
}

int main(int argc, char** argv) {
	if (argc < 4) { fprintf(stderr, "usage: gather_probe <table MB> <bytes 4|8|16|32> <rounds> [lanes]\n"); return 2; }
	const size_t mb = strtoull(argv[1], nullptr, 10);
	const int b = atoi(argv[2]);
	const uint32_t rounds = (uint32_t)atoi(argv[3]);
	const uint32_t lanes = argc > 4 ? (uint32_t)atoi(argv[4]) : 256u * 2048u;
	switch (b) {
		case 4: return run<4>(mb, rounds, lanes);
		case 8: return run<8>(mb, rounds, lanes);
		case 16: return run<16>(mb, rounds, lanes);
		case 32: {
			const int policy = argc > 5 ? atoi(argv[5]) : 0;
			if (policy == 1) return run<32, 1>(mb, rounds, lanes);
			if (policy == 2) return run<32, 2>(mb, rounds, lanes);
			if (policy == 3) return run<32, 3>(mb, rounds, lanes);
			return run<32>(mb, rounds, lanes);
		}
	}
	fprintf(stderr, "bytes per gather must be 4, 8, 16 or 32\n");
	return 2;
}
