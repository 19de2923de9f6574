// Stand-in for <tiny-cuda-nn/random.h> (test infrastructure): tcnn's default_rng_t is Wenzel Jakob's pcg32.h (PCG32, M. O'Neill),
// restated from the published algorithm: state * 0x5851f42d4c957f2d + inc, XSH-RR output, next_float from the top 23 bits,
// advance() by the O(log n) LCG skip of Brown, "Random Number Generation with Arbitrary Stride".
#pragma once
#include <tiny-cuda-nn/common.h>
namespace tcnn {
struct pcg32 {
	uint64_t state, inc;
	pcg32() : state(0x853c49e6748fea9bULL), inc(0xda3e39cb94b95bdbULL) {}
	pcg32(uint64_t initstate, uint64_t initseq = 1u) { seed(initstate, initseq); }
	void seed(uint64_t initstate, uint64_t initseq = 1) {
		state = 0U; inc = (initseq << 1u) | 1u; next_uint(); state += initstate; next_uint();
	}
	uint32_t next_uint() {
		uint64_t oldstate = state;
		state = oldstate * 0x5851f42d4c957f2dULL + inc;
		uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
		uint32_t rot = (uint32_t)(oldstate >> 59u);
		return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
	}
	float next_float() {
		union { uint32_t u; float f; } x;
		x.u = (next_uint() >> 9) | 0x3f800000u;
		return x.f - 1.0f;
	}
	void advance(int64_t delta_ = (1ll << 32)) {
		uint64_t cur_mult = 0x5851f42d4c957f2dULL, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
		uint64_t delta = (uint64_t)delta_;
		while (delta > 0) {
			if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
			cur_plus = (cur_mult + 1) * cur_plus;
			cur_mult *= cur_mult;
			delta /= 2;
		}
		state = acc_mult * state + acc_plus;
	}
};
using default_rng_t = pcg32;
} // namespace tcnn
