import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import Scene, GpuRig
import test_gpu_cell_cache as t
sc = Scene(16, True, 5); rig = GpuRig(sc)
mask = sc.bitfield | sc.edited_bitfield
c = t._coords(200000, 5, 0.0, 1.0)
# half of the samples inside occupied cells: take cell centres of marked cells
from nerfshop_amd import synth
bits = np.unpackbits(mask, bitorder='little').reshape(5, -1)
rng = np.random.default_rng(3)
for lvl in range(5):
    idx = np.nonzero(bits[lvl])[0]
    pick = rng.choice(idx, 20000)
    x, y, z = synth._cell_coords()
    s = 2.0**lvl
    p = (np.stack([x[pick], y[pick], z[pick]], 1) + rng.uniform(0, 1, (20000, 3))) / 128.0
    p = (p - 0.5) * s + 0.5
    w = (p - np.array(sc.desc.aabb_min[:])) / (np.array(sc.desc.aabb_max[:]) - np.array(sc.desc.aabb_min[:]))
    c[lvl*20000:(lvl+1)*20000, :3] = w.astype(np.float32)
rig.net.set_sparse_cell_cache(None, 0)
base = t._encode(rig, c)
print("dense cache", rig.net.cell_cache())
for gb in (2, 16, 64):
    rig.net.set_sparse_cell_cache(mask, gb << 30)
    print("budget", gb, "GB ->", rig.net.sparse_cell_cache())
    got = t._encode(rig, c)
    print("   identical:", np.array_equal(got, base), "differing samples", int((got != base).any(1).sum()))
ref = sc.oracle_model.hashgrid_encode(c[:3000])
print("vs oracle", np.array_equal(base[:3000], ref))
