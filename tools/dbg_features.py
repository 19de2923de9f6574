import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import Scene, GpuRig
import test_gpu_cell_cache as t
sc = Scene(1, True, 6); rig = GpuRig(sc)
inside = t._coords(50000, 5, 0.0, 1.0)
inside[:8, :3] = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], np.float32)
inside[8:2008, :3] = np.round(inside[8:2008, :3] * 64) / 64
around = t._coords(50000, 6, -0.75, 1.75)
around[::7, :3] = t._coords(50000, 8, 0.0, 1.0)[::7, :3]
for name, c in (("inside", inside), ("around", around)):
    ref = sc.oracle_model.hashgrid_encode(c[:3000])
    for budget in (10 << 30, 0):
        rig.net.set_cell_cache(budget)
        got = t._encode(rig, c)[:3000]
        bad = (got != ref)
        print(name, budget, "bad values", int(bad.sum()), "bad samples", int(bad.any(1).sum()), "per level", bad.reshape(3000, 16, 2).any(2).sum(0).tolist(), "first bad samples", np.nonzero(bad.any(1))[0][:10].tolist())
