"""ms per deformed-space occupancy refresh (nrs_model_update_density_grid = Testbed::update_density_grid_nerf_operator, tn:3533-3657) at aabb 1 / 16, with and without cell
records, one cage operator each -- bench.py's next_rows code on its own (A/B of library builds: NRS_LIB_PATH).  usage: python tools/refresh_driver.py"""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from nerfshop_amd import runtime as rt, synth
ctx = rt.Context(0)
rows = {}
for aabb_scale, max_cascade, records in ((1, 0, False), (16, 4, False), (1, 0, True), (16, 4, True)):
    d = synth.model_desc(aabb_scale)
    tb = rt.Testbed(ctx, d, aabb_scale)
    if not records:
        tb.nerf_network.set_cell_cache(0)
    tb.nerf_network.set_params(synth.make_params(d, sigma_raw=synth.default_sigma_raw(aabb_scale), shaped=True, aabb_scale=aabb_scale))
    e = synth.make_cage_edit(lattice_n=10, scene_scale=1.0 if aabb_scale == 1 else 6.0)
    tb.add_edit_operator(rt.CageDeformation(ctx, d, e))
    u = tb.new_grid_update(max_cascade=max_cascade)
    u.reset_grid = 1
    tb.update_density_grid_nerf_operator(u)
    u.reset_grid = 0
    tb.update_density_grid_nerf_operator(u); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        tb.update_density_grid_nerf_operator(u)
    torch.cuda.synchronize()
    rows[f"aabb{aabb_scale}" + ("_records" if records else "")] = round((time.perf_counter() - t0) * 1e3 / 8, 3)
    del tb
print(json.dumps(rows))
