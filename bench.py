#!/usr/bin/env python3
"""bench.py -- the render path's headline benchmark: Msamples/s (+ FPS) at 1920x1080 on the lego-like snapshot.

    python bench.py --gpus 1 --steps 8 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full frame: clear -> nrs_render_nerf (one persistent HIP launch) [-> RCCL gather of the image tiles to
rank 0 + de-tile when N > 1].  The camera orbits (azimuth = 45 deg * step) so steps are not replays of one view.
All inputs (parameters, bitfield, cage tables) are resident in HBM before the timed region.  With N > 1 the SAME frame
is cut into 32x32 tiles dealt round-robin to the ranks ("scaling": "strong": total work per step is fixed).

Workloads (BASELINE.json configs): lego_cage (default; config[2]/[4]: one active cage edit -- the configuration the
north-star target is quoted on), lego (config[1], no edit; also reported as `noedit` in the default line), garden_cage
(config[3]: aabb_scale 16, cone stepping, one cage edit).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BYTES_PER_SAMPLE = 512          # 16 levels x 8 corners x (2 x fp16): SURVEY 8(d)
FLOP_PER_SAMPLE = 20480         # density MLP 6144 + rgb MLP 14336 (configs/nerf/base.json; flop_per_sample() for the other members of the family)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
TILE = int(os.environ.get("NRS_BENCH_TILE", "32"))  # image tiles dealt round-robin to the ranks (multiple of 8): 32 halves the load imbalance of 64 (profiles/r03_scaling.md)


def flop_per_sample(desc):
    """2 x the multiply-adds of the two MLPs of `desc` (20 480 for base.json)."""
    density = 2 * (32 * 64 + 64 * 16) if desc.density_hidden_layers else 2 * (32 * 16)
    L = int(desc.rgb_hidden_layers)
    if not desc.sh_degree:
        return density
    rgb = 2 * (32 * 64 + (L - 1) * 64 * 64 + 64 * 16) if L >= 1 else 2 * (32 * 8)
    return density + rgb


def build_scene(workload, rt, synth, ctx, torch):
    aabb_scale = 16 if workload.startswith("garden") else 1
    # garden_cage = the KNEE of the record-budget curve (profiles/r06_garden.md: 0 / 3.2 / 60.7 GB of sparse brick records -> 4.51 / 4.62 / 4.97 Gsamples/s without the L2
    # phase gate): a 4 GiB budget, levels 8..9 (three hashed pairs left: the three-phase gate, 4.97); garden_cage_records64 = the 64 GiB budget (levels 8..11: two phases, 5.3)
    sparse_gb = os.environ.get("NRS_SPARSE_GB", "64" if workload.endswith("records64") else "4")
    with_edit = "cage" in workload
    # (configs/nerf/base_1layer.json / base_3layer.json: the rgb network with one hidden layer is lowered onto the kernels' network and runs the default
    # instantiations; the third hidden layer has its own: DESIGN.md 4 "Instantiations")
    desc = synth.model_desc(aabb_scale, rgb_hidden_layers=1 if workload.endswith("base_1layer") else (3 if workload.endswith("base_3layer") else 2))
    if workload.endswith("varied"):
        # non-uniform opacity: geometry inside the network (shaped) and a strong density noise, so that per-sample alpha -- and with
        # it the number of samples a ray needs -- varies widely, as in a trained snapshot
        noise = float(os.environ.get("NRS_BENCH_NOISE", "1.5"))  # (profiles/r02_noise_sweep.md sweeps it; the bench line is always 1.5)
        params = synth.make_params(desc, sigma_raw=synth.default_sigma_raw(aabb_scale), shaped=True, density_noise=noise, aabb_scale=aabb_scale)
    else:
        params = synth.make_params(desc, sigma_raw=synth.default_sigma_raw(aabb_scale))
    grid = synth.density_grid(aabb_scale)
    tb = rt.Testbed(ctx, desc, aabb_scale)
    if aabb_scale == 1 and workload.endswith("norecords"):
        tb.nerf_network.set_cell_cache(0)  # what a caller gets who keeps no cell records (a training viewer: nrs_model_set_cell_cache(0))
    if workload.endswith("tcnn_numerics"):
        # tiny-cuda-nn's roundings as recalled: per-corner fp16 grid accumulation + fp16 MLP accumulators (nrs_model_set_numerics) -- the
        # pair a parity-minded integrator switches on; a compile-time instantiation of the automatic schedule since round 4
        g_acc, m_acc = (int(v) for v in os.environ.get("NRS_BENCH_NUMERICS", "1,1").split(","))  # (A/B of the single-rounding instantiations: "1,0" / "0,1")
        tb.nerf_network.set_numerics(g_acc, m_acc)
    tb.nerf_network.set_params(params)
    edit = None
    if with_edit:
        edit = synth.make_cage_edit(lattice_n=10, scene_scale=1.0 if aabb_scale == 1 else 6.0)
        if workload.endswith("membrane"):  # "one extra run with it on" -- the membrane ("Poisson") correction of cage_deformation.h:163
            edit = edit.with_membrane(residual_amplitude=0.8)
        op = rt.CageDeformation(ctx, desc, edit)
        tb.add_edit_operator(op)

        def map_positions(warped):  # deformed-space occupancy refresh through the HIP operator
            d = torch.from_numpy(np.ascontiguousarray(warped, np.float32)).cuda()
            m = torch.zeros(d.shape[0], dtype=torch.uint8, device=d.device)
            op.map_positions(None, d, m)
            torch.cuda.synchronize()
            return d.cpu().numpy(), m.cpu().numpy()

        grid = synth.deformed_density_grid(grid, desc, map_positions, aabb_scale)
    affine = None
    if workload == "lego_affine":
        # one AffineDuplication (editing/affine_duplication.h: the solid's arm shown again translated, scaled and rotated); since round 6 on the
        # automatic schedule (the AFFINE instantiation with lane teams, re-teaming, hand-over and four levels per round trip)
        affine = synth.make_affine_edit()
        aop = rt.AffineDuplication(ctx, desc, affine)
        tb.add_edit_operator(aop)

        def map_positions_a(warped):
            d = torch.from_numpy(np.ascontiguousarray(warped, np.float32)).cuda()
            m = torch.zeros(d.shape[0], dtype=torch.uint8, device=d.device)
            aop.map_positions(None, d, m)
            torch.cuda.synchronize()
            return d.cpu().numpy(), m.cpu().numpy()

        grid = synth.deformed_density_grid(grid, desc, map_positions_a, aabb_scale)
    tb.nerf_network.set_density_grid(grid)  # threshold + mip pooling on the device
    if aabb_scale > 1 and sparse_gb != "0" and not workload.endswith("norecords"):
        # aabb-16 scenes: the dense cell records end at level 7 (7.3 GB); levels 8.. get occupancy-sparse brick records wherever lookups can
        # happen: the occupancy of the edited scene OR the un-edited one (the cage carries samples back to canonical space)
        mask = synth.grid_to_bitfield(grid) | synth.grid_to_bitfield(synth.density_grid(aabb_scale))
        tb.nerf_network.set_sparse_cell_cache(mask, int(float(sparse_gb) * (1 << 30)))
    return dict(desc=desc, params=params, grid=grid, edit=edit, affine=affine, tb=tb, aabb_scale=aabb_scale)


def camera_for(step, synth, aabb_scale):
    scale = 0.33 if aabb_scale == 1 else 0.33 * 6.0
    return synth.orbit_camera(45.0 * (step % 8) + 30.0, 30.0, scale=scale)


def cpu_baseline(synth, scene=None):
    """The CPU leg: the render path on the host cores of the GPU box through oracle/'s restatement in its CPU-baseline flavour (Model.set_fast: F16C half
    conversions, fp32-accumulated MLP sums -- "fp32 math with fp16 rounding points emulated"; same algorithm and rounding points as the checker).  The reference itself has
    no CPU path; this is a port (`kind`), a baseline only.  `value` = a BOUNDED SAMPLE OF THE BENCH'S OWN WORKLOAD: the same scene, the same cage edit and
    occupancy, bench view 0 at 960x540 -- a quarter of the frame's pixels, about 6 M samples, work for every core -- best of 3; `config1` = BASELINE config #1 as it is named (one
    256x256 frame, no edits, best of 5) with the checker flavour's figure next to it; `one_thread` on a 64x64 view."""
    from oracle import oracle as orc
    desc = synth.model_desc(1)
    params = synth.make_params(desc, sigma_raw=synth.default_sigma_raw(1))
    model = orc.Model(desc, params, synth.grid_to_bitfield(synth.density_grid(1)))
    cores = int(orc.load().orc_max_threads())
    cam = synth.orbit_camera(30.0, 30.0, scale=0.33)

    def best_of(m, p, n, edits=(), threads=0):
        best, samples = None, 0
        for _ in range(n):
            t0 = time.perf_counter()
            _, _, _, st = m.render(p, list(edits), n_threads=threads)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            samples = int(st.composited)
        return best, samples

    p256 = synth.render_params(256, 256, cam, aabb_scale=1, apply_operators=False)
    t_checker, s256 = best_of(model, p256, 2)
    model.set_fast(True)
    t256, _ = best_of(model, p256, 5)
    p1 = synth.render_params(64, 64, cam, aabb_scale=1, apply_operators=False)
    t1, s1 = best_of(model, p1, 1, threads=1)
    # the bench's own workload, bounded: its scene, edit and occupancy (`scene` = build_scene's), view 0, 960x540
    if scene is not None and scene.get("edit") is not None and scene["aabb_scale"] == 1:
        wl_model = orc.Model(scene["desc"], scene["params"], synth.grid_to_bitfield(scene["grid"]))
        wl_model.set_fast(True)
        edits = [orc.Edit(scene["desc"], scene["edit"].tet_mesh_struct(), keepalive=scene["edit"])]
        what = "the bench's lego-like scene with its cage edit (BASELINE configs[2])"
    else:
        wl_model, edits, what = model, [], "the lego-like scene, no edits"
    pbig = synth.render_params(960, 540, camera_for(0, synth, 1), aabb_scale=1, apply_operators=bool(edits))
    tbig, sbig = best_of(wl_model, pbig, 3, edits=edits)
    # `value` = the bounded sample of the N = 1 workload, as the bench contract defines it; BASELINE config #1 (the 256 x 256 CPU frame the reference's config list names) sits
    # next to it at the top level (`config1_value`, `config1_ms_per_frame_256`: VERDICT r4 #9) and in full under `config1`
    return {"value": round(sbig / tbig / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "config1_value": round(s256 / t256 / 1e6, 3), "config1_ms_per_frame_256": round(t256 * 1e3, 1),
            "note": "oracle/ restatement in its CPU-baseline flavour (F16C conversions, fp32-accumulated MLP sums, OpenMP over rays); the reference has no CPU path",
            "sample": f"bounded sample of the N = 1 workload: {what}, bench view 0 at 960x540 (a quarter of the pixels), best of 3: {sbig} samples in {tbig * 1e3:.0f} ms ({1.0 / tbig:.2f} FPS)",
            "config1": {"sample": f"BASELINE config #1: one 256x256 frame, no edits, best of 5: {s256} samples in {t256 * 1e3:.0f} ms ({1.0 / t256:.2f} FPS)",
                        "value": round(s256 / t256 / 1e6, 3), "ms_per_frame_256": round(t256 * 1e3, 1),
                        "checker_flavour": {"value": round(s256 / t_checker / 1e6, 4), "ms_per_frame_256": round(t_checker * 1e3, 1),
                                            "note": "software fp16, exact double accumulation: the flavour the parity tests use"}},
            "one_thread": {"value": round(s1 / t1 / 1e6, 4), "unit": "Msamples/s", "sample": f"64x64 view of the same camera, {s1} samples in {t1:.2f} s"}}


def next_rows(rt, synth, ctx, torch):
    """SURVEY 8(f)'s "next" rows re-measured on THIS build (the code of profiles/bench_next_rows.py and tools/op_driver.py): the per-gizmo-move
    chain `nrs_edit_update_cage` (MVC apply + bbox + cell -> tet LUT + rotations + plane records; tet_mesh.cu:368-673) at 6 k and 48 k tets, the deformed-space occupancy
    refresh `nrs_model_update_density_grid` (tn:3533-3657) at aabb 1 / 16, and `network_kernel` (NerfNetwork::inference_mixed_precision on a caller's batch) on 2^22
    ray-ordered and random samples.  ms host-timed over back-to-back calls; `frac` = the row's algorithmic bytes over the time over the 8 TB/s HBM peak (gathers:
    512 B per sample + the batch's own 28 B in / 32 B out; the cage move: the bytes the chain must write -- LUT offsets, ids, plane records, vertices, rotations)."""
    rows = {}

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / reps

    desc = synth.model_desc(1)
    for n in (10, 20):
        e = synth.make_cage_edit(lattice_n=n)
        op = rt.CageDeformation(ctx, desc, e, device_authoring=True)
        op.set_mvc(e.mvc_weights)
        poses = [synth.deform_cage(e.cage_vertices, (0.10 * k / 10, 0.05, 0.0), 20.0 * k / 10) for k in range(1, 11)]
        it = iter(range(1 << 30))
        ms = timed(lambda: op.update_cage(None, poses[next(it) % 10]), 20)
        n_idx, mx = op.lut_size()
        n_t, n_v = int(e.tets.shape[0]), int(e.vertices.shape[0])
        alg = 4 * (5 * 128 ** 3 + 1) + 4 * n_idx + 128 * n_t + 12 * n_v + 36 * n_t  # offsets + ids + plane records + vertices + rotations
        rows[f"cage_move_{n_t // 1000}k_tets"] = {"ms": round(ms, 3), "tets": n_t, "lut_entries": int(n_idx), "algorithmic_bytes": alg, "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        op.close()
    N = 1 << 22
    params = synth.make_params(desc, sigma_raw=synth.default_sigma_raw(1), shaped=True)
    net = rt.NerfNetwork(ctx, desc)
    net.set_params(params)
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = torch.rand((N, 7), generator=g, device="cuda", dtype=torch.float32)
    W, H, S = 2048, 512, 4  # ray-ordered: a 2048 x 512 pinhole image, 4 consecutive steps; sample index = step * n_rays + ray (tn:1023)
    xs = (torch.arange(W, device="cuda", dtype=torch.float32) + 0.5) / W - 0.5
    ys = ((torch.arange(H, device="cuda", dtype=torch.float32) + 0.5) / H - 0.5) * (H / W)
    dirs = torch.stack([xs[None, :].expand(H, W), ys[:, None].expand(H, W), torch.full((H, W), 0.9, device="cuda")], -1).reshape(-1, 3)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    o = torch.tensor([0.5, 0.5, -0.6], device="cuda")
    t = 0.9 + torch.arange(S, device="cuda", dtype=torch.float32) * (3 ** 0.5 / 1024)
    ordered = torch.zeros((N, 7), device="cuda", dtype=torch.float32)
    ordered[:, :3] = (o[None, None, :] + t[:, None, None] * dirs[None, :, :]).reshape(-1, 3).clamp(0.0, 1.0)
    ordered[:, 3] = 3 ** 0.5 / 1024
    ordered[:, 4:] = ((dirs + 1) * 0.5).repeat(S, 1)
    out = torch.zeros((N, 16), device="cuda", dtype=torch.float16)
    for key, batch in (("network_ray_ordered", ordered), ("network_random", rnd)):
        ms = timed(lambda: net.inference_mixed_precision(None, batch, out), 6)
        rows[key] = {"ms": round(ms, 3), "samples": N, "msamples_per_s": round(N / ms / 1e3, 1), "frac": round(N * (BYTES_PER_SAMPLE + 28 + 32) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    net.close()
    del rnd, ordered, out
    for aabb_scale, max_cascade, records in ((1, 0, False), (16, 4, False), (1, 0, True), (16, 4, True)):
        d = synth.model_desc(aabb_scale)
        tb = rt.Testbed(ctx, d, aabb_scale)
        # without records: a refresh that follows a parameter change (a training viewer keeps none); with the harness's default 10 GiB of cell records: the refresh that
        # follows a cage move on a frozen model (the editing viewer; the kernel then runs the L2 phase gate over the trailing hashed level pairs: profiles/r06/ab_refresh_gate.txt)
        if not records:
            tb.nerf_network.set_cell_cache(0)
        tb.nerf_network.set_params(synth.make_params(d, sigma_raw=synth.default_sigma_raw(aabb_scale), shaped=True, aabb_scale=aabb_scale))
        e = synth.make_cage_edit(lattice_n=10, scene_scale=1.0 if aabb_scale == 1 else 6.0)
        tb.add_edit_operator(rt.CageDeformation(ctx, d, e))
        u = tb.new_grid_update(max_cascade=max_cascade)
        u.reset_grid = 1
        tb.update_density_grid_nerf_operator(u)
        u.reset_grid = 0
        ms = timed(lambda: tb.update_density_grid_nerf_operator(u), 6)
        n_s = 128 ** 3 * (max_cascade + 1)
        rows[f"occupancy_refresh_aabb{aabb_scale}" + ("_records" if records else "")] = {"ms": round(ms, 3), "samples": n_s, "msamples_per_s": round(n_s / ms / 1e3, 1),
                                                       "frac": round(n_s * BYTES_PER_SAMPLE / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        del tb
    torch.cuda.empty_cache()
    return rows


TRAFFIC_FILE = "profiles/r06_traffic.json"


def measured_traffic(workload):
    """HBM bytes per render_kernel launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, corrected as
    MI355X_MICROARCH.md prescribes).  PMC counters cannot be sampled from inside this process, so this is a RECORDED number from
    TRAFFIC_FILE (same workload, same kernel; `traffic_source` in the line says so); null for the other workloads."""
    path = os.path.join(ROOT, TRAFFIC_FILE)
    if not os.path.exists(path):
        return None, None
    try:
        j = json.load(open(path))
        if workload != "lego_cage":  # (other workloads with a committed PMC pass sit under their own key)
            j = j.get(workload)
            if not j:
                return None, None
        return int(j["traffic_bytes_per_launch"]), f"recorded: {TRAFFIC_FILE} ({j.get('source', 'rocprofv3 --pmc passes')}), not measured by this run"
    except Exception:
        return None, None


def render_kernel_counter_values(csv_paths, counter):
    """Per-dispatch values of one PMC counter for the render kernel from rocprofv3's `*_counter_collection.csv` files (columns Kernel_Name, Counter_Name,
    Counter_Value; one row per dispatch and counter)."""
    import csv
    vals = []
    for path in csv_paths:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if "render_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                    vals.append(float(row["Counter_Value"]))
    return vals


def live_traffic(workload, width, height):
    """HBM bytes per render_kernel launch measured NOW: two rocprofv3 PMC passes (FETCH_SIZE, then WRITE_SIZE -- separate passes, --kernel-trace only, as
    MI355X_MICROARCH.md prescribes) over a short child run of this script (8 steps = the 8 bench views; its render_kernel dispatches are averaged), corrected
    as that guide says for gfx950 (FETCH_SIZE tallies the 128-byte requests at 64 bytes: doubled).  Returns (bytes, description) or (None, why)."""
    import glob
    import shutil
    import subprocess
    import tempfile
    if os.environ.get("NRS_BENCH_LIVE_TRAFFIC", "1") == "0":
        return None, "switched off (NRS_BENCH_LIVE_TRAFFIC=0)"
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="nrs_traffic_", dir="/tmp")
    kb, n_disp = {}, 0
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "t", "--", sys.executable, os.path.abspath(__file__),
                   "--workload", workload, "--steps", "8", "--warmup", "0", "--no-cpu-baseline", "--no-extra", "--width", str(width), "--height", str(height)]
            env = dict(os.environ, NRS_BENCH_LIVE_TRAFFIC="0", TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} pass failed (rc {r.returncode})"
            vals = render_kernel_counter_values(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True), counter)
            if not vals:
                return None, f"no render_kernel rows in the {counter} pass"
            kb[counter] = sum(vals) / len(vals)
            n_disp = len(vals)
    except Exception as e:  # (a measurement aid must never take the benchmark down)
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    traffic = int(2 * kb["FETCH_SIZE"] * 1024 + kb["WRITE_SIZE"] * 1024)
    return traffic, (f"measured by this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over a child `bench.py --workload {workload} --steps 8`, "
                     f"mean of its {n_disp} render_kernel dispatches; 2 x FETCH_SIZE ({kb['FETCH_SIZE']:.0f} KB) + WRITE_SIZE ({kb['WRITE_SIZE']:.0f} KB), gfx950 correction of MI355X_MICROARCH.md")


def time_gather(torch, dist, ctx, sharder, stream, p, frame, depth, reps, world, stats_dev):
    """The frame's exchange step alone (VERDICT r4 next #4c): `reps` x nrs_gather_tiles (ncclSend / ncclRecv of every rank's tile block to rank 0 + de-tile there) on
    `stream`, each bracketed by HIP events recorded on that stream; plus the wall clock of the `reps` calls back to back between barriers.  Every rank takes part.
    Returns {"ms_event_mean", "ms_event_min", "ms_event_max_rank_mean", "ms_wall_per_gather", "bytes_to_root", "impl"} (rank 0's events; the slowest rank's mean)."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    with torch.cuda.stream(stream):
        for _ in range(3):
            sharder.gather(ctx, p, frame, depth)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for a, b in ev:
            a.record(stream)
            sharder.gather(ctx, p, frame, depth)
            b.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = (time.perf_counter() - t0) / reps * 1e3
    ms = [a.elapsed_time(b) for a, b in ev]
    mine = torch.tensor([sum(ms) / len(ms), wall], dtype=torch.float64, device=stats_dev)
    if world > 1:
        dist.all_reduce(mine, op=dist.ReduceOp.MAX)
    n_px = sharder.padded * sharder.tile * sharder.tile
    return {"ms_event_mean": round(sum(ms) / len(ms), 4), "ms_event_min": round(min(ms), 4), "ms_event_max_rank_mean": round(float(mine[0]), 4),
            "ms_wall_per_gather": round(float(mine[1]), 4), "reps": reps, "bytes_to_root": int((world - 1) * n_px * 5 * 4),
            "payload_gbs_at_event_mean": round((world - 1) * n_px * 20 / (sum(ms) / len(ms) * 1e-3) / 1e9, 1) if world > 1 else None,
            "impl": sharder.gather_impl, "note": "rank 0's HIP events around nrs_gather_tiles on the frame's stream (send/recv of (N - 1) tile blocks [frame | depth] + de-tile kernel); "
                                                 "`ms_event_max_rank_mean` = the slowest rank's mean, `ms_wall_per_gather` = host wall clock of the calls back to back, max over ranks"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="lego_cage", choices=["lego_cage", "lego", "garden_cage", "garden_cage_records64", "garden", "lego_cage_varied", "lego_cage_membrane", "garden_cage_norecords", "lego_cage_tcnn_numerics", "lego_cage_norecords", "lego_cage_base_1layer", "lego_cage_base_3layer", "lego_affine"])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary no-edit measurement")
    ap.add_argument("--frames-in-flight", type=int, default=1, help="frames rendered concurrently (double-buffered streams when > 1)")
    ap.add_argument("--gather-only", action="store_true", help="N > 1: time nrs_gather_tiles (the frame's one exchange step: RCCL send/recv to rank 0 + de-tile) alone, with HIP events; no rendering")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from nerfshop_amd import runtime as rt
    from nerfshop_amd import synth, tiles

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # NRS_BENCH_DIST=gloo: the control plane (barriers, the statistics' all-reduce, the communicator id's broadcast) over gloo, and the ranks may
    # SHARE a GPU (device = local rank modulo the visible devices) -- with NRS_RCCL_LIB=tests/fake_rccl/libfake_rccl.so the whole N > 1 branch of this
    # script, nrs_gather_tiles included, then runs on a one-GPU box (tests/test_gpu_bench_multiproc.py).  Default: nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("NRS_BENCH_DIST", "nccl")
    if backend not in ("nccl", "gloo"):
        raise SystemExit(f"NRS_BENCH_DIST={backend}: nccl or gloo")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1) if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    stats_dev = torch.device("cpu") if backend == "gloo" else dev  # where the tensors of the control-plane collectives live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    ctx = rt.Context(dev_index)
    if args.gather_only:
        # the exchange alone: no model, no rendering -- the tile buffers hold zeros (the transfer does not care)
        sh = tiles.TileSharder(args.width, args.height, TILE, rank, world, dev)
        aabb_scale = 16 if args.workload.startswith("garden") else 1
        p = sh.fill(synth.render_params(args.width, args.height, camera_for(0, synth, aabb_scale), aabb_scale=aabb_scale))
        frame = torch.zeros((args.height, args.width, 4), dtype=torch.float32, device=dev)
        depth = torch.zeros((args.height, args.width), dtype=torch.float32, device=dev)
        g = time_gather(torch, dist, ctx, sh, torch.cuda.Stream(device=dev), p, frame, depth, max(args.steps, 1), world, stats_dev)
        if rank == 0:
            print(json.dumps({"metric": "gather_tiles_ms", "value": g["ms_event_mean"], "unit": "ms", "n_gpus": world, "steps": max(args.steps, 1), "warmup": 3, "higher_is_better": False,
                              "config": {"workload": f"the frame's exchange step alone: {args.width}x{args.height} RGBA + depth, {TILE}x{TILE} tiles round-robin over {world} GPU(s), gather to rank 0 + de-tile"},
                              "gather": g}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    scene = build_scene(args.workload, rt, synth, ctx, torch)
    tb = scene["tb"]
    W, H = args.width, args.height

    # Frames are rendered strictly one at a time by default (frames_in_flight = 1, at every N): per-launch durations stay clean
    # for the roofline and the per-N values compare like with like.  The secondary "pipelined" measurement below double-buffers
    # frames over two HIP streams (two tile buffers, two output images): the last generation of rays of frame k -- one ray's
    # latency, during which the GPU drains -- then overlaps the start of frame k + 1, and the RCCL gather hides behind rendering.
    n_buf = max(1, args.frames_in_flight)
    tiled = world > 1 or n_buf > 1
    max_buf = max(n_buf, 1 if args.no_extra else 4)
    all_sharders = [tiles.TileSharder(W, H, TILE, rank, world, dev) for _ in range(max_buf)]
    all_streams = [torch.cuda.Stream(device=dev) for _ in range(max_buf)]
    frames = [torch.zeros((H, W, 4), dtype=torch.float32, device=dev) for _ in range(max_buf)]
    depths = [torch.zeros((H, W), dtype=torch.float32, device=dev) for _ in range(max_buf)]
    frame, depth = frames[0], depths[0]
    sharders = all_sharders if tiled else None
    streams = all_streams if tiled else None
    sharder = all_sharders[0] if tiled else None

    def make_params(step, apply_ops=True, force_tiled=False):
        p = synth.render_params(W, H, camera_for(step, synth, scene["aabb_scale"]), aabb_scale=scene["aabb_scale"], apply_operators=apply_ops)
        p.poisson_target = 1 if args.workload.endswith("membrane") else 0  # NerfTracer::m_poisson_target = true (testbed.h:219) wherever the membrane correction is on
        if sharder is not None or force_tiled:
            all_sharders[0].fill(p)
        return p

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def one_step(step, timed_idx=None, apply_ops=True, want_stats=False):
        p = make_params(step, apply_ops)
        if sharder is None:
            frame.zero_()  # clear_frame (testbed.cu:2635)
            if timed_idx is not None:
                ev0[timed_idx].record()
            st = tb.render_with_params(tb.nerf_network, p, frame, depth, None, None, want_stats=want_stats)
            if timed_idx is not None:
                ev1[timed_idx].record()
        else:
            b = step % n_buf
            sh, stream = sharders[b], streams[b]
            with torch.cuda.stream(stream):
                sh.clear()
                if timed_idx is not None:
                    ev0[timed_idx].record(stream)
                st = tb.render_with_params(tb.nerf_network, p, sh.local_frame, sh.local_depth, None, stream, want_stats=want_stats)
                if timed_idx is not None:
                    ev1[timed_idx].record(stream)
                sh.gather(ctx, p, frames[b], depths[b])  # RCCL gather to rank 0 + de-tile, ordered after this stream's render
        return st

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # sample counts per view (untimed; the stats read-back synchronises, so it stays out of the timed region)
    samples_per_step = []
    for s in range(8):
        st = one_step(s, want_stats=True)
        samples_per_step.append(int(st.n_samples))
    for s in range(args.warmup):
        one_step(s)
    sync_all()
    t0 = time.perf_counter()
    for s in range(args.steps):
        one_step(s, timed_idx=s)
    sync_all()
    elapsed = time.perf_counter() - t0

    local_samples = sum(samples_per_step[s % 8] for s in range(args.steps))
    kernel_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / args.steps
    stats = torch.tensor([elapsed, float(local_samples), kernel_ms], dtype=torch.float64, device=stats_dev)
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, total_samples, kernel_ms = float(mx[0]), float(sm[1]), float(mx[2])
    else:
        total_samples = float(local_samples)

    extra = {}
    if rank == 0 and world == 1 and not args.no_extra and args.workload == "lego_cage":
        # secondary: BASELINE config[1] (no edit operators; same occupancy so the sample set is comparable)
        for s in range(2):
            one_step(s, apply_ops=False)
        torch.cuda.synchronize()
        ns = sum(int(one_step(s, apply_ops=False, want_stats=True).n_samples) for s in range(8))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for s in range(8):
            one_step(s, apply_ops=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        extra["noedit"] = {"msamples_per_s": round(ns / dt / 1e6, 2), "fps": round(8 / dt, 2)}
        # BASELINE configs[1] as scripts/run.py renders it offline: 8 spp per view -- 8 frames with the Sobol pixel offsets of spp_index 0..7, each joined to the
        # running mean by nrs_accumulate (CudaRenderBuffer::accumulate); one "image" = 8 x (clear + render + accumulate)
        from nerfshop_amd import _abi as abi
        lib = abi.load()
        accum = torch.zeros_like(frame)

        def spp8_view(view, count=False):
            n = 0
            for k in range(8):
                p = synth.render_params(W, H, camera_for(view, synth, scene["aabb_scale"]), aabb_scale=scene["aabb_scale"], apply_operators=False, spp_index=k, snap=False)
                frame.zero_()
                st = tb.render_with_params(tb.nerf_network, p, frame, depth, None, None, want_stats=count)
                if count:
                    n += int(st.n_samples)
                abi.check(lib.nrs_accumulate(ctx.h, None, W, H, frame.data_ptr(), accum.data_ptr(), k, 0))
            return n
        ns_spp8 = sum(spp8_view(v, count=True) for v in range(4))  # (untimed pass over the SAME 32 jittered frames: ADVICE r4)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for v in range(4):
            spp8_view(v)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        extra["noedit_spp8"] = {"images_per_s": round(4 / dt, 2), "ms_per_8spp_image": round(dt / 4 * 1e3, 2), "msamples_per_s": round(ns_spp8 / dt / 1e6, 2),
                                "note": "BASELINE configs[1] offline: 8 frames (Sobol offsets of spp_index 0..7) + nrs_accumulate per 1920x1080 image, no edits"}

    if not args.no_extra:
        # secondary: the same frames with 2 and with 4 in flight (all ranks take part; `value` stays the one-at-a-time figure).  The drain of
        # frame k -- one ray's latency, during which a GPU that owns only 1/N of the picture runs nearly empty -- overlaps frames k+1..;
        # a viewer that accepts that latency (or an offline render) gets this rate, DESIGN 5.
        def pipelined_step(step, k):
            b = step % k
            p = make_params(step, force_tiled=True)
            with torch.cuda.stream(all_streams[b]):
                all_sharders[b].clear()
                tb.render_with_params(tb.nerf_network, p, all_sharders[b].local_frame, all_sharders[b].local_depth, None, all_streams[b])
                all_sharders[b].gather(ctx, p, frames[b], depths[b])
        for k in (2, 4):
            for s in range(2 * k):
                pipelined_step(s, k)
            sync_all()
            t1 = time.perf_counter()
            for s in range(args.steps):
                pipelined_step(s, k)
            sync_all()
            dtp = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=stats_dev)
            if world > 1:
                dist.all_reduce(dtp, op=dist.ReduceOp.MAX)
            rec = {"frames_in_flight": k, "msamples_per_s": round(total_samples / float(dtp[0]) / 1e6, 2), "fps": round(args.steps / float(dtp[0]), 2)}
            extra["pipelined" if k == 2 else "pipelined4"] = rec

    if rank == 0 and world == 1 and not args.no_extra and args.workload == "lego_cage":
        # secondary workloads, one frame at a time like `value`: BASELINE configs[3] (garden-style: aabb_scale 16, cone stepping, 5 cascades, one
        # cage edit) and the lego-like scene with non-uniform opacity (a wide distribution of ray lengths, as a trained snapshot has)
        # plus the membrane correction on (SURVEY 8d's "one extra run with it on") and the garden scene WITHOUT the 64 GB of sparse brick records
        # (they are an option of the boundary, INTEGRATION.md: the figure a caller gets who does not install them)
        for name in ("garden_cage", "garden_cage_records64", "garden_cage_norecords", "lego_cage_varied", "lego_cage_membrane", "lego_cage_tcnn_numerics", "lego_cage_norecords", "lego_cage_base_1layer",
                     "lego_cage_base_3layer", "lego_affine"):
            sc2 = build_scene(name, rt, synth, ctx, torch)
            tb2 = sc2["tb"]

            def step2(step, want_stats=False):
                p2 = synth.render_params(W, H, camera_for(step, synth, sc2["aabb_scale"]), aabb_scale=sc2["aabb_scale"], apply_operators=True)
                p2.poisson_target = 1 if name.endswith("membrane") else 0  # (the reference's default, testbed.h:219)
                frame.zero_()
                return tb2.render_with_params(tb2.nerf_network, p2, frame, depth, None, None, want_stats=want_stats)
            ns = sum(int(step2(s2, want_stats=True).n_samples) for s2 in range(8))
            rays = int(step2(0, want_stats=True).n_rays_alive)
            for s2 in range(2):
                step2(s2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for s2 in range(8):
                step2(s2)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            extra[name] = {"msamples_per_s": round(ns / dt / 1e6, 2), "fps": round(8 / dt, 2), "samples_per_frame": ns // 8, "rays_view0": rays,
                           "roofline_frac": round(ns / dt * BYTES_PER_SAMPLE / 1e9 / HBM_PEAK_GBS, 4),
                           "cell_records_gb": round((tb2.nerf_network.cell_cache()[0] + tb2.nerf_network.sparse_cell_cache()[0]) / 1e9, 1)}
            del sc2, tb2
            torch.cuda.empty_cache()
            # the HBM rate the frame really runs at: PMC traffic of this workload's kernel, measured NOW for the garden scene (two more child runs,
            # like the headline's), else the figure profiles/ holds, if any (`traffic_source` says which)
            tr, src = (None, None)
            if name in ("garden_cage", "garden_cage_records64"):
                tr, src = live_traffic(name, W, H)
            if tr is None:
                why = src
                tr, src = measured_traffic(name)
                if src and why:
                    src += f" (live measurement unavailable: {why})"
            if tr:
                extra[name]["traffic"] = int(tr)
                extra[name]["traffic_rate_frac"] = round(tr / (dt / 8) / 1e9 / HBM_PEAK_GBS, 4)
                extra[name]["l2_misses_per_sample"] = round(tr / 128.0 / (ns / 8), 2)  # every L2 miss is one 128-byte fabric request (profiles/r02_gather_probe.md)
                extra[name]["traffic_source"] = src

    if rank == 0 and world == 1 and not args.no_extra and args.workload == "lego_cage":
        try:
            extra["next_rows"] = next_rows(rt, synth, ctx, torch)
        except Exception as e:  # (a secondary measurement must not take the headline down)
            extra["next_rows"] = {"error": f"{type(e).__name__}: {e}"}

    n1_ref, gather_leg = None, None
    if world > 1 and not args.no_extra:
        # (VERDICT r4 next #4b) the same run's N = 1 figure: rank 0 renders the WHOLE frames alone, one at a time, while the other ranks wait at the barrier --
        # `retention_k` below = per-GPU sample throughput of the N-GPU job with k frames in flight relative to this
        sync_all()
        if rank == 0:
            whole_f = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
            whole_d = torch.zeros((H, W), dtype=torch.float32, device=dev)

            def whole_step(step, want_stats=False):
                pw = synth.render_params(W, H, camera_for(step, synth, scene["aabb_scale"]), aabb_scale=scene["aabb_scale"], apply_operators=True)
                pw.poisson_target = 1 if args.workload.endswith("membrane") else 0
                whole_f.zero_()
                return tb.render_with_params(tb.nerf_network, pw, whole_f, whole_d, None, None, want_stats=want_stats)
            ns1 = sum(int(whole_step(s1, want_stats=True).n_samples) for s1 in range(8))
            for s1 in range(3):
                whole_step(s1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for s1 in range(16):
                whole_step(s1)
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
            n1_ref = {"msamples_per_s": round(ns1 * 2 / dt1 / 1e6, 2), "fps": round(16 / dt1, 2), "ms_per_frame": round(dt1 / 16 * 1e3, 3),
                      "note": "this run's own N = 1 figure: rank 0 renders the whole frames alone, one at a time (16 frames over the 8 views), the other ranks idle"}
        sync_all()
        # (next #4c) the exchange step alone, so that a multi-GPU record separates exchange from render
        p_g = make_params(0)
        gather_leg = time_gather(torch, dist, ctx, all_sharders[0], all_streams[0], p_g, frames[0], depths[0], 20, world, stats_dev)
        sync_all()

    gather_check = None
    if world > 1:
        # the exchanged frame against the same view rendered WHOLE on rank 0 (every rank takes part in the exchange; outside the timed region): what the
        # N > 1 branch delivers is the single-GPU picture, bit for bit
        p_chk = make_params(0)
        with torch.cuda.stream(all_streams[0]):
            all_sharders[0].clear()
            tb.render_with_params(tb.nerf_network, p_chk, all_sharders[0].local_frame, all_sharders[0].local_depth, None, all_streams[0])
            all_sharders[0].gather(ctx, p_chk, frames[0], depths[0])
        sync_all()
        if rank == 0:
            p_whole = synth.render_params(W, H, camera_for(0, synth, scene["aabb_scale"]), aabb_scale=scene["aabb_scale"], apply_operators=True)
            whole, whole_d = torch.zeros_like(frames[0]), torch.zeros_like(depths[0])
            st_whole = tb.render_with_params(tb.nerf_network, p_whole, whole, whole_d, None, None, want_stats=True)
            torch.cuda.synchronize()
            hit = whole[..., 3] > 0
            gather_check = {"frame_equal": bool(torch.equal(frames[0].view(torch.int32), whole.view(torch.int32))),
                            "depth_equal": bool(torch.equal(depths[0][hit], whole_d[hit])), "pixels_hit": int(hit.sum()),
                            "whole_frame_samples": int(st_whole.n_samples), "tiled_samples_all_ranks": int(total_samples / args.steps) if args.steps % 8 == 0 else None,
                            "view": "step 0"}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_samples / elapsed / 1e6
        # roofline of the dominant kernel (render_kernel): algorithmic bytes per launch / mean launch duration (HIP events)
        per_launch_samples = total_samples / args.steps / world
        traffic, traffic_source = (None, None)
        if world == 1 and not args.no_extra:  # (the default run; the child runs it spawns carry --no-extra)
            traffic, traffic_source = live_traffic(args.workload, W, H)
            if traffic is None:
                why = traffic_source
                traffic, traffic_source = measured_traffic(args.workload)
                if traffic_source:
                    traffic_source += f" (live measurement unavailable: {why})"
        else:
            traffic, traffic_source = measured_traffic(args.workload)
        ach = per_launch_samples * BYTES_PER_SAMPLE / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "render_msamples_per_s_1080p",
            "value": round(value, 2),
            "unit": "Msamples/s",
            "fps": round(args.steps / elapsed, 2),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f16 storage / f32 accumulate (MFMA), f32 marching+compositing",
            "data": "synthetic",
            "config": {"workload": {"lego_cage": "lego-like snapshot 1920x1080, one cage edit (BASELINE configs[2]/[4])",
                                    "lego": "lego-like snapshot 1920x1080, no edits (BASELINE configs[1])",
                                    "garden_cage": "garden-style aabb_scale 16 1920x1080, one cage edit (BASELINE configs[3]); sparse brick records at the knee of the budget curve (4 GiB)",
                                    "garden_cage_records64": "garden-style aabb_scale 16 1920x1080, one cage edit, 64 GiB budget of sparse brick records (levels 8..11; the L2 phase gate applies)",
                                    "garden": "garden-style aabb_scale 16 1920x1080, no edits",
                                    "lego_cage_varied": "lego-like snapshot with non-uniform opacity (geometry in the network, density noise 1.5) 1920x1080, one cage edit",
                                    "lego_cage_membrane": "lego-like snapshot 1920x1080, one cage edit with the membrane (Poisson) correction on",
                                    "garden_cage_norecords": "garden-style aabb_scale 16 1920x1080, one cage edit, no sparse brick records",
                                    "lego_cage_tcnn_numerics": "lego-like snapshot 1920x1080, one cage edit, tiny-cuda-nn's roundings (fp16 per-corner grid accumulation, fp16 MLP accumulators)",
                                    "lego_cage_norecords": "lego-like snapshot 1920x1080, one cage edit, no cell records (nrs_model_set_cell_cache(0))",
                                    "lego_cage_base_1layer": "lego-like snapshot of configs/nerf/base_1layer.json (rgb network with one hidden layer) 1920x1080, one cage edit",
                                    "lego_cage_base_3layer": "lego-like snapshot of configs/nerf/base_3layer.json (rgb network with three hidden layers) 1920x1080, one cage edit",
                                    "lego_affine": "lego-like snapshot 1920x1080, one AffineDuplication edit (no cage)"}[args.workload],
                       "resolution": [W, H], "samples_per_frame": int(total_samples / args.steps),
                       "sharding": f"{TILE}x{TILE} image tiles round-robin over {world} GPU(s)" + (f", gather to rank 0 by {all_sharders[0].gather_impl}" if world > 1 else ""),
                       "frames_in_flight": n_buf,
                       "cell_records": "levels 0..%d, %.1f GB (nrs_model_set_cell_cache default)" % (tb.nerf_network.cell_cache()[1] - 1, tb.nerf_network.cell_cache()[0] / 1e9) +
                                       ("; sparse brick records for levels %d..%d, %.1f GB" % (tb.nerf_network.sparse_cell_cache()[1], sum(tb.nerf_network.sparse_cell_cache()[1:]) - 1,
                                                                                               tb.nerf_network.sparse_cell_cache()[0] / 1e9) if tb.nerf_network.sparse_cell_cache()[2] else "")},
            "roofline": {"bound": "hbm",
                         #  what `bound` / `frac` are: the contract's ALGORITHMIC figure -- 512 B per sample over the HBM peak -- not the kernel's limiter
                         "bound_kind": "algorithmic-hbm (figure of merit: algorithmic bytes / kernel time / HBM peak; the kernel's own limiter is in `limiter`, its real HBM rate in `traffic_rate_frac`)",
                         "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                         "traffic": traffic if world == 1 else None, "traffic_source": traffic_source if world == 1 else None,
                         #  `frac` is the ALGORITHMIC fraction the contract asks for (512 B per sample / kernel time / peak); the HBM rate the
                         # kernel really runs at is the measured traffic over the same time -- about half of it on this scene: the kernel is not HBM-bound
                         "traffic_rate_frac": round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (traffic and world == 1) else None,
                         "limiter": "VALU issue (DESIGN.md 4): algorithmic bytes over the HBM peak is the contract's figure of merit, not the kernel's bound",
                         "kernel": "render_kernel", "kernel_ms": round(kernel_ms, 3),
                         "algorithmic_bytes_per_launch": int(per_launch_samples * BYTES_PER_SAMPLE),
                         "mfma_tflops": round(per_launch_samples * flop_per_sample(scene["desc"]) / (kernel_ms * 1e-3) / 1e12, 2)},
        }
        if world > 1 and all_sharders[0].comm is not None:  # what the exchange actually ran on: ranks RCCL connected, its version, the library loaded
            import ctypes as C
            from nerfshop_amd import _abi
            ci = [C.c_int(), C.c_int(), C.c_int()]
            lp = C.create_string_buffer(256)
            _abi.check(_abi.load().nrs_comm_info(all_sharders[0].comm, C.byref(ci[0]), C.byref(ci[1]), C.byref(ci[2]), lp, 256))
            line["config"]["comm"] = {"n_ranks": ci[1].value, "rccl_version": ci[2].value, "library": lp.value.decode(), "control_plane": backend}
        if gather_check is not None:
            line["config"]["gather_check"] = gather_check
        line.update(extra)
        if "lego_cage_varied" in extra:
            #  the second headline: the varied-opacity scene -- the workload that resembles a trained snapshot (ray lengths spread widely) -- one frame at a
            # time like `value`, on the same 8 views; `value` keeps its meaning (the uniform lego-like scene BASELINE's target is quoted on)
            line["value_varied"] = extra["lego_cage_varied"]["msamples_per_s"]
            line["fps_varied"] = extra["lego_cage_varied"]["fps"]
            line["roofline_frac_varied"] = extra["lego_cage_varied"]["roofline_frac"]
        if n1_ref is not None:
            # per-GPU sample throughput retained at N GPUs (north_star: >= 0.9 at 8): whole-job Msamples/s / N over this run's own N = 1 figure.  `value` is and stays the
            # one-frame-at-a-time figure (retention_1); retention_2 / _4 are the same frames with 2 / 4 in flight per rank (the `pipelined` / `pipelined4` keys)
            line["n1_reference"] = n1_ref
            line["retention_1"] = round(value / world / n1_ref["msamples_per_s"], 4)
            #  what retention_1 can read AT MOST with this design, so that a SCALE record is not misread: a rank's 1/N share of this frame measured on
            # ONE MI355X with a free exchange (profiles/r06_scaling.md: 2.076 ms whole frame; shares 1.173 / 0.671 / 0.449 ms one frame at a time).  The share's launch is
            # launch + fill + first hits (0.37 / 0.33 / 0.26 ms, the intercept of the cut-off probe) + the rounds of its longest rays; north_star's 0.9 at N = 8 would need
            # the whole share in 0.288 ms -- the non-round part alone takes 0.26.  Frames in flight are what reaches it (retention_2 / _4).
            line["retention_floor_1"] = {2: 0.885, 4: 0.773, 8: 0.578}.get(world)
            line["retention_floor_note"] = "one-GPU share measurement of the same frame, exchange free (profiles/r06_scaling.md): the most `retention_1` can read with one frame in flight"
            for k, key in ((2, "pipelined"), (4, "pipelined4")):
                if key in extra:
                    line[f"retention_{k}"] = round(extra[key]["msamples_per_s"] / world / n1_ref["msamples_per_s"], 4)
        if gather_leg is not None:
            line["gather"] = gather_leg
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(synth, scene)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
