import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import Scene, GpuRig
sc = Scene(1, True, 6); rig = GpuRig(sc)
rig.use_edit(True)
p = sc.params_for(1920, 1080, 30.0)
frame, depth, steps, stats = rig.render(p)
rf, rd, rs, rst = sc.oracle_model.render(p, [sc.oracle_edit])
d = np.abs(frame - rf)
print("max", d.max(), "mean", d.mean(), "p99.9", np.quantile(d, 0.999), "p99.99", np.quantile(d, 0.9999), "p99.999", np.quantile(d, 0.99999), "count>6e-3", int((d > 6e-3).sum()), "pixels>6e-3", int((d > 6e-3).any(-1).sum()))
ds = np.abs(steps.astype(np.int64) - rs.astype(np.int64))
print("steps max", ds.max(), "equal frac", (ds == 0).mean(), "n differing", int((ds != 0).sum()))
big = (d > 6e-3).any(-1)
print("of the big ones: steps differ", int((ds[big] != 0).sum()), "of", int(big.sum()))
print("alpha of big", rf[big][:, 3][:10], frame[big][:, 3][:10])
same = ds == 0
hit = (rf[..., 3] > 0.2) & (frame[..., 3] > 0.2) & same
print("depth max diff", np.abs(depth[hit] - rd[hit]).max())
print("samples", stats.n_samples, rst.composited, "alive", stats.n_rays_alive, rst.n_alive0)
