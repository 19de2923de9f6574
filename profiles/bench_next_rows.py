"""Timings of the SURVEY 8(f) "next" rows on the device, with the CPU path beside them (one JSON line each):
  cage_move        nrs_edit_update_cage  (MVC apply + bbox + cell->tet LUT + rotations)   vs  libnrs's threaded host builder
  occupancy_refresh nrs_model_update_density_grid (128^3 * (max_cascade+1) samples)        vs  the oracle (OpenMP) on the host
  selection_rays   nrs_project_selection_pixels (scribble pixels -> surface points)         vs  the oracle's three-step restatement
  poisson_boundary nrs_poisson_boundary (cage vertices x 100 directions -> density + SH9)   vs  the oracle
Run on the GPU box:  python profiles/bench_next_rows.py [--lattice 10 20] [--reps 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lattice", type=int, nargs="*", default=[10, 20])
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    import torch
    from nerfshop_amd import runtime, synth, _abi
    from oracle import oracle as orc
    ctx = runtime.Context(0)
    desc = synth.model_desc(1)
    for n in a.lattice:
        e = synth.make_cage_edit(lattice_n=n)
        op = runtime.CageDeformation(ctx, desc, e, device_authoring=True)
        op.set_mvc(e.mvc_weights)
        poses = [synth.deform_cage(e.cage_vertices, (0.10 * k / a.reps, 0.05, 0.0), 20.0 * k / a.reps) for k in range(1, a.reps + 1)]
        op.update_cage(None, poses[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in poses:
            op.update_cage(None, c)
        torch.cuda.synchronize()
        gpu_ms = (time.perf_counter() - t0) * 1e3 / len(poses)
        cpu_ms = None
        if not a.no_cpu:
            t0 = time.perf_counter()
            for c in poses[:3]:
                v = synth.mvc_apply(e.mvc_weights, c)
                synth.build_tet_lut(v, e.tets)
                synth.local_rotations(v, e.original_vertices, e.tets)
            cpu_ms = (time.perf_counter() - t0) * 1e3 / 3
        n_idx, mx = op.lut_size()
        print(json.dumps({"row": "cage_move", "tets": int(e.tets.shape[0]), "vertices": int(e.vertices.shape[0]), "lut_entries": n_idx,
                          "max_tets_per_cell": mx, "gpu_ms_per_move": round(gpu_ms, 3), "cpu_host_builder_ms_per_move": cpu_ms and round(cpu_ms, 1),
                          "cpu_threads": os.cpu_count()}))
        op.close()
    for aabb_scale, max_cascade in ((1, 0), (16, 4)):
        d = synth.model_desc(aabb_scale)
        params = synth.make_params(d, sigma_raw=synth.default_sigma_raw(aabb_scale), shaped=True, aabb_scale=aabb_scale)
        tb = runtime.Testbed(ctx, d, aabb_scale)
        tb.nerf_network.set_params(params)
        e = synth.make_cage_edit(lattice_n=10, scene_scale=1.0 if aabb_scale == 1 else 6.0)
        tb.add_edit_operator(runtime.CageDeformation(ctx, d, e))
        u = tb.new_grid_update(max_cascade=max_cascade)
        u.reset_grid = 1
        tb.update_density_grid_nerf_operator(u)
        u.reset_grid = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            tb.update_density_grid_nerf_operator(u)
        torch.cuda.synchronize()
        gpu_ms = (time.perf_counter() - t0) * 1e3 / a.reps
        cpu_ms = None
        if not a.no_cpu:
            om = orc.Model(d, params)
            oe = orc.Edit(d, e.tet_mesh_struct(), keepalive=e)
            grid = np.zeros(5 * 128 ** 3, np.float32)
            u2 = tb.new_grid_update(max_cascade=max_cascade)
            u2.reset_grid = 1
            t0 = time.perf_counter()
            om.update_density_grid(grid, u2, [oe])
            cpu_ms = (time.perf_counter() - t0) * 1e3
        n = 128 ** 3 * (max_cascade + 1)
        print(json.dumps({"row": "occupancy_refresh", "aabb_scale": aabb_scale, "samples": n, "gpu_ms_per_iteration": round(gpu_ms, 3),
                          "gpu_msamples_per_s": round(n / gpu_ms / 1e3, 1), "cpu_oracle_ms_per_iteration": cpu_ms and round(cpu_ms, 1),
                          "cpu_threads": orc.load().orc_max_threads()}))

    # ---- selection rays / membrane boundary ----
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest
    scene = conftest.Scene(aabb_scale=1, with_edit=False)
    tb = runtime.Testbed(ctx, scene.desc, 1)
    tb.nerf_network.set_params(scene.params)
    tb.nerf_network.set_density_bitfield(scene.bitfield)
    for n_px in (3000, 100000):
        w, h = 1920, 1080
        p = scene.params_for(w, h, 50.0)
        rng = np.random.default_rng(1)
        px = np.stack([rng.integers(w // 4, 3 * w // 4, n_px), rng.integers(h // 4, 3 * h // 4, n_px)], 1).astype(np.int32)
        tb.project_selection_pixels(p, px)
        t0 = time.perf_counter()
        for _ in range(5):
            (pos, cells, found), _sel = tb.project_selection_pixels(p, px)
        gpu_ms = (time.perf_counter() - t0) * 1e3 / 5
        cpu_ms = None
        if not a.no_cpu:
            t0 = time.perf_counter()
            scene.oracle_model.project_selection_pixels(p, px)
            cpu_ms = (time.perf_counter() - t0) * 1e3
        print(json.dumps({"row": "selection_rays", "pixels": n_px, "found": int(found.sum()), "gpu_ms_end_to_end": round(gpu_ms, 3),
                          "cpu_ms": None if cpu_ms is None else round(cpu_ms, 1), "cpu_threads": orc.max_threads() if hasattr(orc, "max_threads") else None}), flush=True)
    for n_v in (300, 3000):
        rng = np.random.default_rng(2)
        v = rng.uniform(0.3, 0.7, size=(n_v, 3)).astype(np.float32)
        jitter = rng.uniform(0, 1, size=(n_v * 100, 2)).astype(np.float32)
        tb.compute_poisson_boundary(v, True, jitter)
        t0 = time.perf_counter()
        for _ in range(5):
            tb.compute_poisson_boundary(v, True, jitter)
        gpu_ms = (time.perf_counter() - t0) * 1e3 / 5
        cpu_ms = None
        if not a.no_cpu:
            t0 = time.perf_counter()
            scene.oracle_model.poisson_boundary(v, 10, 10, jitter, True)
            cpu_ms = (time.perf_counter() - t0) * 1e3
        print(json.dumps({"row": "poisson_boundary", "vertices": n_v, "samples": n_v * 100, "gpu_ms_end_to_end": round(gpu_ms, 3),
                          "cpu_ms": None if cpu_ms is None else round(cpu_ms, 1)}), flush=True)


if __name__ == "__main__":
    main()
