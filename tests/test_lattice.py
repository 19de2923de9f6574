"""The arithmetic fact behind lattice_jump (nrs_device.cuh; DESIGN.md 4, "Lattice jump"), checked on the CPU with IEEE float32 (numpy): with a constant step dt the
parameters a voxel walk stands on are t (+) dt (+) dt ...; inside one binade [2^e, 2^(e+1)) every such rounded addition adds the SAME number q of ulps, q = dt / ulp(t)
rounded to the nearest integer (no ties for this dt: the increment then does not depend on the parity of t), so the lattice point k steps on is the integer addition
bits(t) + k q -- what the kernel computes instead of the chain.  The formulas here are the kernel's (x = dt 2^(150 - e_t), q = rint(x), ulps left in the binade)."""
import numpy as np

MIN_STEP = np.float32(1.7320508 / 1024.0)  # NRS_MIN_STEP = sqrt(3) / 1024 (common_nerf.h:29-32), as the float the kernels use


def _q_for(t):
    e_t = int(np.float32(t).view(np.uint32) >> 23)
    x = np.ldexp(np.float64(MIN_STEP), 150 - e_t)           # exact: a power-of-two scaling of a 24-bit number
    q = np.rint(x)
    return e_t, x, int(q), abs(x - q) != 0.5


def test_min_step_is_the_kernels_constant():
    assert MIN_STEP.view(np.uint32) == 0x3ADDB3D7


def test_no_binade_of_interest_has_a_tie():
    """t between 2^-6 and 2^7: every binade a ray parameter of these scenes can fall into."""
    for e in range(-6, 8):
        _, x, q, no_tie = _q_for(np.float32(2.0 ** e))
        assert no_tie and q >= 1, (e, x)


def test_integer_lattice_equals_the_chain_of_additions():
    rng = np.random.default_rng(5)
    checked = 0
    for e in range(-4, 4):
        lo, hi = np.float32(2.0 ** e), np.float32(2.0 ** (e + 1))
        for t0 in rng.uniform(lo, hi, size=200).astype(np.float32):
            e_t, x, q, no_tie = _q_for(t0)
            b = int(t0.view(np.uint32))
            k_bin = (0x7FFFFF - (b & 0x7FFFFF)) // q        # lattice points left in t0's binade
            if k_bin < 2:
                continue
            k = int(rng.integers(1, min(k_bin, 700) + 1))
            t = t0
            for _ in range(k):                              # the reference's chain: t += dt, one rounding per step
                t = np.float32(t + MIN_STEP)
            assert int(t.view(np.uint32)) == b + k * q, (float(t0), k, q)
            assert t < hi
            checked += 1
    assert checked > 1000


def test_crossing_a_binade_by_ordinary_additions_then_jumping_again():
    """The kernel's two segments: integer steps to the end of the binade, three ordinary additions, integer steps in the next binade -- equal to the plain chain."""
    t0 = np.float32(0.93)
    total = 180                                             # 0.93 + 180 dt = 1.23: crosses t = 1
    chain = t0
    for _ in range(total):
        chain = np.float32(chain + MIN_STEP)
    t, left = t0, total
    for seg in range(2):
        e_t, x, q, no_tie = _q_for(t)
        b = int(t.view(np.uint32))
        k_bin = max((0x7FFFFF - (b & 0x7FFFFF)) // q - 1, 0)
        k = min(left, k_bin) if no_tie else 0
        t = np.uint32(b + k * q).view(np.float32)
        left -= k
        for _ in range(3):
            if left > 0:
                t = np.float32(t + MIN_STEP)
                left -= 1
    assert left == 0 and t.view(np.uint32) == chain.view(np.uint32)
