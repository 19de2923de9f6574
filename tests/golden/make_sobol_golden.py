"""Generates tests/golden/sobol_golden.json from the REFERENCE's own Sobol direction-number table.

The reference cannot be compiled here (CUDA), but include/neural-graphics-primitives/random_val.cuh carries the
5 x 32 direction numbers as plain data.  This script parses that table out of /root/reference (run it in the
build container only; /root/reference does not exist on the GPU box) and evaluates, in pure Python integer
arithmetic, the functions the render path uses -- sobol(), the Burley/Laine-Karras nested uniform scramble,
ld_random_val(), ld_random_pixel_offset() -- following random_val.cuh:159-288,317-322.  The committed JSON pins
the oracle's *generated* direction numbers (dims 0 and 1) and its scramble against the reference's table.

    python tests/golden/make_sobol_golden.py
"""
import json
import os
import re
import struct

REF = "/root/reference/include/neural-graphics-primitives/random_val.cuh"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sobol_golden.json")
M32 = 0xFFFFFFFF


def parse_directions():
    src = open(REF).read()
    body = src[src.index("directions[5][32]"):]
    body = body[body.index("{") + 1: body.index("};")]
    nums = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]{8}", body)]
    assert len(nums) == 160, len(nums)
    return [nums[32 * d: 32 * (d + 1)] for d in range(5)]


def f32(x):
    return struct.unpack("f", struct.pack("f", x))[0]


def main():
    D = parse_directions()

    def sobol(index, dim):
        X = 0
        for bit in range(32):
            if (index >> bit) & 1:
                X ^= D[dim][bit]
        return X

    def hash_combine(seed, v):
        return (seed ^ ((v + ((seed << 6) & M32) + (seed >> 2)) & M32)) & M32

    def reverse_bits(x):
        return int("{:032b}".format(x)[::-1], 2)

    def lk(x, seed):
        x = (x + seed) & M32
        for c in (0x6c50b47c, 0xb82f1e52, 0xc7afe638, 0x8d22f6e6):
            x ^= (x * c) & M32
        return x

    def scramble(x, seed):
        return reverse_bits(lk(reverse_bits(x), seed))

    S = f32(1.0 / (1 << 32))

    def to_unit(u):  # (float)u * S with float32 rounding of the conversion and of the product
        return f32(f32(float(u)) * S)

    def ld_random_val(index, seed, dim=0):
        index = scramble(index, seed)
        return to_unit(scramble(sobol(index, dim), hash_combine(seed, dim)))

    def ld_random_val_2d(index, seed):
        index = scramble(index, seed)
        return [to_unit(scramble(sobol(index, i), hash_combine(seed, i))) for i in range(2)]

    def fractf(x):
        import math
        return f32(x - math.floor(x))

    def pixel_offset(spp):
        a, b = ld_random_val_2d(0, 0xdeadbeef), ld_random_val_2d(spp, 0xdeadbeef)
        return [fractf(f32(f32(0.5 - a[i]) + b[i])) for i in range(2)]

    idxs = [0, 1, 2, 3, 5, 17, 255, 256, 65535, 0x12345678, 0xFFFFFFFF]
    seeds = [0, 1, 786433, 786433 * 1000 & M32, 0xdeadbeef, 72239731 * 777 & M32]
    golden = {
        "source": "include/neural-graphics-primitives/random_val.cuh:160-205 (direction numbers), :159-288, :317-322",
        "directions_dim0": D[0], "directions_dim1": D[1],
        "sobol": [[i, d, sobol(i, d)] for i in idxs for d in (0, 1)],
        "ld_random_val": [[i, s, ld_random_val(i, s)] for i in idxs[:6] for s in seeds],
        "pixel_offset": [[spp, pixel_offset(spp)] for spp in (0, 1, 2, 7, 100)],
    }
    json.dump(golden, open(OUT, "w"), indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
