"""Inputs and cases of the reference pin (tests/test_ref_pin.py, tests/golden/make_ref_pin_golden.py).

Every case is evaluated twice with the same seeded inputs: by the REFERENCE's own code compiled as host code (oracle/_ref/libref_render.so,
which="ref"; only where /root/reference is mounted) and by the oracle's restatement (which="orc").  The generator stores the reference's
outputs (small arrays whole, 10^5-input probes as SHA-256 of the output bytes plus their first rows) in tests/golden/ref_pin_golden.npz, so
the GPU box -- where the reference does not exist -- still checks the oracle against the reference's code.
"""
import hashlib

import numpy as np

N = 100_000


def _rng(tag):
    return np.random.Generator(np.random.PCG64([1337, sum(ord(c) * (i + 1) for i, c in enumerate(tag))]))


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode())
        h.update(a.tobytes())
    return np.frombuffer(h.digest(), np.uint8).copy()


# ---------------------------------------------------------------------------------------------------------------------------------
# element-wise probes of the header / common_nerf.cu functions
# ---------------------------------------------------------------------------------------------------------------------------------
def run_probes(which):
    """-> {name: [output arrays]} for 10^5 seeded inputs per function.  which = "ref" | "orc"."""
    from oracle import ref
    out = {}
    r = _rng("tets")
    abcd = r.uniform(0, 1, (N, 12)).astype(np.float32)
    p = r.uniform(0, 1, (N, 3)).astype(np.float32)
    w = r.dirichlet(np.ones(4), N).astype(np.float32)
    p[:N // 2] = (abcd[:N // 2].reshape(-1, 4, 3) * w[:N // 2, :, None]).sum(1)  # half the points inside their tet
    p[N // 2:N // 2 + 4000] = abcd[N // 2:N // 2 + 4000, :3]                     # on a vertex
    out["bary_tet"] = [ref.bary_tet(abcd, p, which)]                               # selection_utils.h:14-31
    out["point_in_tet"] = [ref.point_in_tet(abcd, p, which)]                       # selection_utils.h:33-47

    r = _rng("sobol")
    idx = r.integers(0, 2 ** 32, N, dtype=np.uint64).astype(np.uint32)
    idx[:4096] = np.arange(4096)
    seed = r.integers(0, 2 ** 32, N, dtype=np.uint64).astype(np.uint32)
    seed[:4096] = (np.arange(4096) * 786433).astype(np.uint32)
    out["ld_random_val"] = [ref.ld_random_val(idx, seed, which)]                   # random_val.cuh:284-288
    out["ld_random_pixel_offset"] = [ref.ld_random_pixel_offset(idx, which)]       # random_val.cuh:317-322
    out["sobol"] = [ref.sobol(idx, 0, which), ref.sobol(idx, 1, which)]            # random_val.cuh:159-216 (dims 0, 1: the ones the path uses)

    r = _rng("box")
    box = np.concatenate([r.uniform(-1, 0.4, (N, 3)), r.uniform(0.6, 2, (N, 3))], 1).astype(np.float32)
    o = r.uniform(-3, 3, (N, 3)).astype(np.float32)
    d = r.normal(size=(N, 3)).astype(np.float32)
    d[::7, 0] = 0.0; d[::11, 1] = 0.0; d[::13, 2] = -0.0                            # axis-parallel rays: divisions by +-0
    a, inside = ref.ray_intersect(box, o, d, which)                                # bounding_box.cuh:180-238, :243-248
    out["ray_intersect"] = [a, inside]
    lo = r.uniform(0, 1, (N, 3)).astype(np.float32)
    box2 = np.concatenate([lo, lo + r.uniform(0.01, 0.5, (N, 3)).astype(np.float32)], 1)
    tri = r.uniform(-1, 2, (N, 9)).astype(np.float32)
    tri[:N // 2] = (box2[:N // 2, None, :3] + r.normal(scale=0.3, size=(N // 2, 3, 3))).reshape(-1, 9).astype(np.float32)
    out["box_intersects_triangle"] = [ref.box_intersects_triangle(box2, tri, which)]  # bounding_box.cuh:126-178, triangle.cuh:39

    r = _rng("grid")
    mip = r.integers(0, 5, N).astype(np.uint32)
    pos = (0.5 + (r.uniform(0, 1, (N, 3)) - 0.5) * (2.0 ** mip)[:, None] * 0.999).astype(np.float32)  # inside cascade `mip`
    dg = r.normal(size=(N, 3))
    dg = np.where(np.abs(dg) < 1e-2, 0.1, dg)
    dg = (dg / np.linalg.norm(dg, axis=1, keepdims=True)).astype(np.float32)
    t = r.uniform(0, 20, N).astype(np.float32)
    cone = np.where(r.uniform(size=N) < 0.5, 0.0, 1.0 / 256.0).astype(np.float32)
    g = ref.grid_math(pos, dg, t, cone, mip, which)                                # common_nerf.cu:89-177
    out["grid_math"] = [g[k] for k in ("calc_dt", "mip_from_pos", "mip_from_dt", "cell_idx", "distance_to_next_voxel", "advance_to_next_voxel")]
    bx = np.array([-7.5] * 3 + [8.5] * 3, np.float32)
    dt = r.uniform(0, 0.3, N).astype(np.float32)
    wv = ref.warp(bx, pos, dt, which)                                              # common_nerf.cu:5-36
    out["warp"] = [wv[k] for k in ("warp_position", "unwarp_position", "warp_direction", "unwarp_direction", "warp_dt", "unwarp_dt")]
    q = np.concatenate([r.integers(0, 128, (N, 3)), mip[:, None]], 1).astype(np.uint32)
    cp, ca = ref.cell_functions(q, pos, which)                                     # selection_utils.cu:65-83
    out["cell_functions"] = [cp, ca]

    r = _rng("sh")
    shc = r.normal(size=(N, 27)).astype(np.float32)
    dn = r.normal(size=(N, 3))
    dn = (dn / np.linalg.norm(dn, axis=1, keepdims=True)).astype(np.float32)
    out["evaluate_sh9"] = [ref.evaluate_sh9(shc, dn, which)]                       # common_nerf.cu:218-245
    x = r.uniform(-12, 12, N).astype(np.float32)
    x[:2000] = np.linspace(0, 1, 2000)
    av = ref.activations(x, which)                                                 # common_device.cuh:31-37, common_nerf.cu:38-66 (libm exp on the host)
    out["activations"] = [av[k] for k in ("srgb_to_linear", "rgb_logistic", "rgb_exponential", "density_exponential")]
    return out


def pixel_to_ray_case(scene, which):
    """pixel_to_ray (common_device.cuh:245-295) for every pixel of two views, snapped and jittered."""
    from oracle import ref
    outs = []
    for az, snap, spp, dof, focus in ((30.0, 1, 0, 0.0, 1.0), (200.0, 0, 7, 0.0, 1.0), (75.0, 0, 5, 0.03, 1.4), (310.0, 1, 0, 0.1, 0.8)):  # the last two: thin-lens branch (:285-293)
        p = scene.params_for(160, 90, az)
        p.snap_to_pixel_centers, p.spp_index, p.dof, p.slice_plane_z = snap, spp, dof, focus
        ys, xs = np.mgrid[0:90, 0:160]
        px = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
        o, d = ref.pixel_to_ray(px, p, which)
        outs += [o, d]
    # lens distortion (:262-280): OpenCV iterative undistortion, f-theta, the distortion map
    for az, mode, prm, dmap in ((45.0, 1, (0.2, -0.08, 0.006, -0.004, 0, 0, 0), None), (45.0, 1, (-0.25, 0.1, 0.0, 0.0, 0, 0, 0), (20, 10, 3, 0.01)),
                                (130.0, 2, (0.0, 0.8, 0.03, -0.02, 0.004, 1.0, 0.5625), None), (130.0, 2, (0.0, 2.9, 0.0, 0.0, 0.0, 1.0, 0.5625), None)):  # the last: rays beyond 90 degrees -> error direction
        p = scene.params_for(160, 90, az)
        p.snap_to_pixel_centers, p.spp_index = 0, 4
        p.distortion_mode = mode
        p.distortion_params[:] = prm
        camera_extras(p, {"_distmap": dmap} if dmap else {})
        ys, xs = np.mgrid[0:90, 0:160]
        px = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
        o, d = ref.pixel_to_ray(px, p, which)
        outs += [o, d]
    return outs


# ---------------------------------------------------------------------------------------------------------------------------------
# whole frames: Testbed::render_nerf with the reference's kernels (network = the oracle's, tcnn is outside the reference checkout)
# ---------------------------------------------------------------------------------------------------------------------------------
FRAME_CASES = [
    # name, scene key, (W, H, azimuth), params overrides, edit kind
    ("lego_noedit", "lego", (64, 36, 30.0), {}, None),
    ("lego_edit", "lego", (64, 36, 60.0), {}, "cage"),
    ("lego_edit_jitter", "lego", (64, 36, 100.0), {"snap_to_pixel_centers": 0, "spp_index": 5}, "cage"),
    ("lego_edit_cost", "lego", (64, 36, 200.0), {"render_mode": 8}, "cage"),
    ("lego_edit_linear", "lego", (64, 36, 250.0), {"linear_colors": 1}, "cage"),
    ("lego_membrane", "lego", (64, 36, 60.0), {}, "membrane"),
    ("lego_membrane_target", "lego", (64, 36, 60.0), {"poisson_target": 1}, "membrane"),
    ("lego_over_background", "lego", (64, 36, 30.0), {"_background": 0.25}, "cage"),
    ("aabb16_edit", "aabb16", (64, 36, 30.0), {}, "cage"),
    ("aabb16_edit_jitter", "aabb16", (64, 36, 120.0), {"snap_to_pixel_centers": 0, "spp_index": 3}, "cage"),
    # round 3: the rest of render_nerf's surface -- composite_kernel_nerf's per-sample modes (tn:905-937), show_accel (tn:788-790, 911-920),
    # depth of field (common_device.cuh:285-293), the Slice path (tn:3111-3175)
    ("lego_edit_ao", "lego", (64, 36, 60.0), {"render_mode": 0}, "cage"),
    # render modes Normals / EncodingVis (round 4): the reference's composite / shade branches (:905-910, :925, :2466-2468) around the oracle's restatements of
    # tiny-cuda-nn's input_gradient / visualize_activation -- incl. the overwritten network input of EncodingVis (dt = 1, direction (1, 1, 1): :762, :803)
    ("lego_edit_normals", "lego", (64, 36, 60.0), {"render_mode": 2}, "cage"),
    ("aabb16_normals", "aabb16", (64, 36, 30.0), {"render_mode": 2}, None),
    ("lego_edit_encoding_vis_hidden", "lego", (64, 36, 100.0), {"render_mode": 11, "visualized_layer": 1, "visualized_dimension": 20}, "cage"),
    ("lego_encoding_vis_grid", "lego", (64, 36, 30.0), {"render_mode": 11, "visualized_layer": 0, "visualized_dimension": 3}, None),
    ("lego_membrane_encoding_vis", "lego", (64, 36, 60.0), {"render_mode": 11, "visualized_layer": 2, "visualized_dimension": 17, "poisson_target": 1}, "membrane"),
    ("lego_edit_positions", "lego", (64, 36, 100.0), {"render_mode": 3}, "cage"),
    ("lego_positions_accel0", "lego", (64, 36, 30.0), {"render_mode": 3, "show_accel": 1, "min_mip": 0}, None),
    ("aabb16_positions_accel1", "aabb16", (64, 36, 30.0), {"render_mode": 3, "show_accel": 1, "min_mip": 1}, "cage"),
    ("lego_edit_depth", "lego", (64, 36, 200.0), {"render_mode": 4, "depth_scale": 0.7}, "cage"),
    ("lego_edit_distance", "lego", (64, 36, 250.0), {"render_mode": 5, "depth_scale": 1.3}, "cage"),
    ("aabb16_stepsize", "aabb16", (64, 36, 120.0), {"render_mode": 6}, "cage"),
    ("lego_membrane_ao", "lego", (64, 36, 60.0), {"render_mode": 0}, "membrane"),
    ("lego_show_accel_shade", "lego", (64, 36, 60.0), {"show_accel": 1, "min_mip": 0}, "cage"),
    ("lego_dof", "lego", (64, 36, 60.0), {"dof": 0.02, "slice_plane_z": 1.2, "snap_to_pixel_centers": 0, "spp_index": 3}, "cage"),
    ("aabb16_dof_snapped", "aabb16", (64, 36, 30.0), {"dof": 0.05, "slice_plane_z": 2.5, "spp_index": 11}, "cage"),
    ("lego_slice", "lego", (64, 36, 30.0), {"render_mode": 9, "slice_plane_z": 1.3}, None),
    ("lego_slice_linear_bg", "lego", (64, 36, 140.0), {"render_mode": 9, "slice_plane_z": 1.2, "linear_colors": 1, "_background": 0.25}, "cage"),
    ("aabb16_slice", "aabb16", (64, 36, 30.0), {"render_mode": 9, "slice_plane_z": 2.0}, None),
    # camera model and background: OpenCV / f-theta lens distortion, the distortion map, the environment map, render mode Distortion
    # (init_rays_with_payload_kernel_nerf tn:2523-2613, pixel_to_ray common_device.cuh:262-280, envmap.cuh:30-63)
    ("lego_envmap", "lego", (64, 36, 60.0), {"_envmap": (32, 16, 5)}, "cage"),
    ("lego_opencv_distortion", "lego", (64, 36, 100.0), {"distortion_mode": 1, "distortion_params": (0.12, -0.05, 0.004, -0.003, 0, 0, 0), "snap_to_pixel_centers": 0, "spp_index": 2}, "cage"),
    ("lego_ftheta", "lego", (64, 36, 30.0), {"distortion_mode": 2, "distortion_params": (0.0, 0.7, 0.02, -0.01, 0.002, 1.0, 0.5625)}, None),
    ("aabb16_distortion_map_envmap", "aabb16", (64, 36, 120.0), {"_distmap": (24, 12, 9, 0.02), "_envmap": (40, 20, 6)}, "cage"),
    ("lego_mode_distortion_map", "lego", (64, 36, 60.0), {"render_mode": 7, "_distmap": (24, 12, 9, 0.004)}, "cage"),
    ("lego_mode_distortion_nomap", "lego", (64, 36, 60.0), {"render_mode": 7, "_background": 0.25}, None),
    # composite_kernel_nerf's glow overlay (tn:806-903): green grid + cut line, radial grid-only, mask to alpha
    ("lego_glow_grid_cutline", "lego", (64, 36, 60.0), {"glow_mode": 3, "glow_y_cutoff": 0.55}, "cage"),
    ("lego_glow_radial_gridonly", "lego", (64, 36, 100.0), {"glow_mode": 24, "glow_y_cutoff": 0.5}, None),
    ("aabb16_glow_mask_to_alpha", "aabb16", (64, 36, 30.0), {"glow_mode": 5, "glow_y_cutoff": 0.6}, "cage"),
    ("lego_slice_distorted_lens", "lego", (64, 36, 30.0), {"render_mode": 9, "slice_plane_z": 1.3, "distortion_mode": 1, "distortion_params": (0.15, -0.05, 0.003, 0.002, 0, 0, 0), "dof": 0.05,
                                                          "_distmap": (24, 12, 9, 0.01)}, None),
]


def camera_extras(p, over):
    """fill the pointer fields of an nrs_render_params from the '_envmap' / '_distmap' keys of a case (host arrays, kept alive on the struct)"""
    import ctypes as C
    keep = []
    if "_envmap" in over:
        w, h, seed = over["_envmap"]
        env = _rng(f"envmap{seed}").uniform(0, 1, (h, w, 4)).astype(np.float32)
        env[..., 3] = _rng(f"envalpha{seed}").uniform(0.2, 1.0, (h, w)).astype(np.float32)
        p.d_envmap = env.ctypes.data_as(C.c_void_p)
        p.envmap_resolution[:] = (w, h)
        keep.append(env)
    if "_distmap" in over:
        w, h, seed, amp = over["_distmap"]
        dm = _rng(f"distmap{seed}").uniform(-amp, amp, (h, w, 2)).astype(np.float32)
        p.d_distortion_map = dm.ctypes.data_as(C.c_void_p)
        p.distortion_resolution[:] = (w, h)
        keep.append(dm)
    p._keep = keep
    return keep


class Scenes:
    """the conftest.Scene objects the cases use, built on demand"""

    def __init__(self):
        self._s = {}

    def get(self, key):
        if key not in self._s:
            from conftest import Scene
            if key == "lego_shaped":  # geometry in the network: the occupancy refresh has something to find
                self._s[key] = Scene(aabb_scale=1, with_edit=True, lattice_n=6, shaped=True)
            else:
                self._s[key] = Scene(aabb_scale=1, with_edit=True, lattice_n=6) if key == "lego" else Scene(aabb_scale=16, with_edit=True, lattice_n=5)
        return self._s[key]


def _case_setup(scenes, case):
    from oracle import oracle as orc
    name, key, (W, H, az), over, kind = case
    sc = scenes.get(key)
    p = sc.params_for(W, H, az)
    background = None
    for k, v in over.items():
        if k == "_background":
            background = v
        elif k.startswith("_"):
            continue
        elif k == "distortion_params":
            p.distortion_params[:] = v
        else:
            setattr(p, k, v)
    camera_extras(p, over)
    edit = None
    if kind == "cage":
        edit = sc.edit
    elif kind == "membrane":
        edit = sc.edit.with_membrane(residual_amplitude=0.8)
    p.apply_operators = 1 if edit is not None else 0
    bitfield = sc.edited_bitfield if edit is not None else sc.bitfield
    frame0 = np.full((H, W, 4), background, np.float32) if background is not None else None
    return sc, p, edit, bitfield, frame0


def render_case(scenes, case, which):
    """-> frame, depth, steps, (n_hit, composited, generated)"""
    from oracle import oracle as orc
    from oracle import ref
    sc, p, edit, bitfield, frame0 = _case_setup(scenes, case)
    sc.oracle_model.set_bitfield(bitfield)
    if which == "ref":
        meshes = [edit.tet_mesh_struct()] if edit is not None else []
        f, d, s, st = ref.render_frame(sc.desc, p, bitfield, meshes, sc.oracle_model, frame=frame0)
    else:
        edits = [orc.Edit(sc.desc, edit.tet_mesh_struct(), keepalive=edit)] if edit is not None else []
        W, H = p.resolution[0], p.resolution[1]
        if frame0 is not None:  # orc.Model.render allocates a cleared frame: render, then composite over the background as shade_kernel_nerf does
            import ctypes as C
            lib = orc.load()
            f, d, s = frame0.copy(), np.zeros((H, W), np.float32), np.zeros((H, W), np.uint32)
            st = orc.OrcRenderStats()
            arr = (C.c_void_p * max(len(edits), 1))(*[e.h for e in edits])
            lib.orc_render(sc.oracle_model.h, C.byref(p), arr, len(edits), f.ctypes.data, d.ctypes.data, s.ctypes.data, C.byref(st), 0, 0)
        else:
            f, d, s, st = sc.oracle_model.render(p, edits)
    return f, d, s, np.array([st.n_hit, st.composited, st.generated], np.uint64)


# ---------------------------------------------------------------------------------------------------------------------------------
# sample streams: per pixel the network-input records of every generated sample (init_rays -> advance_pos -> generate_next...)
# ---------------------------------------------------------------------------------------------------------------------------------
STREAM_CASES = [
    ("lego", (96, 54, 30.0), {}),
    ("lego", (96, 54, 140.0), {"snap_to_pixel_centers": 0, "spp_index": 9}),
    ("lego", (96, 54, 300.0), {"min_mip": 1}),
    ("aabb16", (96, 54, 30.0), {}),
    ("aabb16", (96, 54, 210.0), {"snap_to_pixel_centers": 0, "spp_index": 2}),
]


def stream_case(scenes, case, which):
    from oracle import ref
    key, (W, H, az), over = case
    sc = scenes.get(key)
    p = sc.params_for(W, H, az)
    for k, v in over.items():
        setattr(p, k, v)
    px = np.arange(W * H, dtype=np.uint32)
    coords, t_after, cnt, odt = ref.trace_coords(sc.desc, p, sc.edited_bitfield, px, 160, which)
    return coords, t_after, cnt, odt


# ---------------------------------------------------------------------------------------------------------------------------------
# edit operators on caller batches, LUT builder, rotations, MVC
# ---------------------------------------------------------------------------------------------------------------------------------
def edit_coords(sc, n, seed):
    """samples concentrated in the deformed and canonical boxes of the scene's cage edit, warped to [0,1]^3 of the scene box"""
    r = np.random.default_rng(seed)
    e = sc.edit
    lo = np.minimum(e.vertices.min(0), e.original_vertices.min(0)) - 0.05
    hi = np.maximum(e.vertices.max(0), e.original_vertices.max(0)) + 0.05
    pos = r.uniform(lo, hi, size=(n, 3))
    mn, mx = np.array(sc.desc.aabb_min[:]), np.array(sc.desc.aabb_max[:])
    c = np.zeros((n, 7), np.float32)
    c[:, :3] = ((pos - mn) / (mx - mn)).astype(np.float32)
    c[:, 3] = r.uniform(0, 1, n)
    d = r.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    c[:, 4:] = ((d + 1) * 0.5).astype(np.float32)
    return c


def operator_cases(scenes, which):
    """-> {name: [arrays]}: map_rays / map_positions / membrane residuals of the cage edits, AffineDuplication, on 10^5 samples"""
    from oracle import oracle as orc
    from oracle import ref
    out = {}
    for key in ("lego", "aabb16"):
        sc = scenes.get(key)
        c = edit_coords(sc, N, 11)
        for copy in (0, 1):
            e = sc.edit
            mesh = e.tet_mesh_struct()
            mesh.copy = copy
            if which == "ref":
                rc, re = ref.edit_map_rays(sc.desc, mesh, c)
                rp, rpe = ref.edit_map_positions(sc.desc, mesh, np.ascontiguousarray(c[:, :3]))
            else:
                o = orc.Edit(sc.desc, mesh, keepalive=e)
                rc, re = o.map_rays(c)
                rp, rpe = o.map_positions(np.ascontiguousarray(c[:, :3]))
            out[f"{key}_map_rays_copy{copy}"] = [rc, re]
            out[f"{key}_map_positions_copy{copy}"] = [rp, rpe]
        em = sc.edit.with_membrane(residual_amplitude=0.8)
        mesh = em.tet_mesh_struct()
        if which == "ref":
            sh, od, rd = ref.edit_poisson_residuals(sc.desc, mesh, c)
        else:
            sh, od, rd = orc_poisson_residuals(sc.desc, mesh, em, c)
        out[f"{key}_poisson_residuals"] = [sh, od, rd]
    # AffineDuplication with a general rotation (all three Euler angles) of operator and selection box
    from nerfshop_amd import synth
    sc = scenes.get("lego")
    for hide, cd in ((0, 1), (1, 1), (1, 0)):
        op = synth.make_affine_edit(hide_original=bool(hide), correct_dir=bool(cd))
        op.rotation[:] = [float(v) for v in _euler(0.3, -0.5, 0.2).T.reshape(-1)]
        op.selection_rot[:] = [float(v) for v in _euler(-0.1, 0.25, 0.4).T.reshape(-1)]
        r = np.random.default_rng(5)
        c = r.uniform(0, 1, size=(N, 7)).astype(np.float32)
        sel = np.array(op.selection_center[:], np.float32)
        ext = np.array(op.selection_scale[:], np.float32)
        c[:N // 3, :3] = sel + r.uniform(-0.75, 0.75, size=(N // 3, 3)).astype(np.float32) * ext
        c[N // 3:2 * (N // 3), :3] = sel + np.array(op.translation[:], np.float32) + r.uniform(-0.9, 0.9, size=(N // 3, 3)).astype(np.float32) * ext
        if which == "ref":
            rc, re = ref.affine_map_rays(sc.desc, op, c)
            rp, rpe = ref.affine_map_positions(sc.desc, op, np.ascontiguousarray(c[:, :3]))
        else:
            o = orc.AffineEdit(sc.desc, op)
            rc, re = o.map_rays(c)
            rp, rpe = o.map_positions(np.ascontiguousarray(c[:, :3]))
        out[f"affine_map_rays_hide{hide}_dir{cd}"] = [rc, re]
        out[f"affine_map_positions_hide{hide}_dir{cd}"] = [rp, rpe]
    return out


def _euler(a, b, c):
    ca, sa, cb, sb, cc, sc_ = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
    rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
    ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
    rz = np.array([[cc, -sc_, 0], [sc_, cc, 0], [0, 0, 1]])
    return (rz @ ry @ rx).astype(np.float32)


def orc_poisson_residuals(desc, mesh, keepalive, coords7):
    """compute_residual_poisson_kernel through the oracle (one sample per call site, as ref_edit_poisson_residuals)"""
    import ctypes as C
    from oracle import oracle as orc
    lib = orc.load()
    e = orc.Edit(desc, mesh, keepalive=keepalive)
    c = np.ascontiguousarray(coords7, np.float32)
    n = c.shape[0]
    sh, od, rd = np.zeros((n, 27), np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    lib.orc_p_poisson_residuals.restype = None
    lib.orc_p_poisson_residuals(C.c_void_p(e.h), C.c_uint32(n), C.c_void_p(c.ctypes.data), C.c_void_p(sh.ctypes.data), C.c_void_p(od.ctypes.data), C.c_void_p(rd.ctypes.data))
    return sh, od, rd


def authoring_cases(scenes, which):
    """-> {name: [arrays]}: TetMesh::build_tet_grid tables, update_local_rotations, Cage::compute_mvc / interpolate_with_mvc"""
    from oracle import oracle as orc
    from oracle import ref
    import importlib.util
    import os
    out = {}
    for key in ("lego", "aabb16"):
        e = scenes.get(key).edit
        if which == "ref":
            off, idx, bf, mx = ref.build_tet_grid(e.vertices, e.original_vertices, e.tets)
            rot = ref.local_rotations(e.vertices, e.original_vertices, e.tets)
        else:
            off, idx, _, mx = orc.tet_lut_build(e.vertices, e.tets)
            _, _, bf, _ = orc.tet_lut_build(e.original_vertices, e.tets)
            rot = orc.local_rotations(e.vertices, e.original_vertices, e.tets)
        out[f"{key}_tet_grid"] = [off, idx, bf, np.array([mx], np.uint32)]
        out[f"{key}_local_rotations"] = [rot]
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_ref_mvc_golden", os.path.join(here, "golden", "make_ref_mvc_golden.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    cv, ct, pts = g.inputs()
    if which == "ref":
        w, lab = ref.mvc_compute(cv, ct, pts)
        moved = ref.mvc_apply(w, cv * np.float32(1.07) + np.float32(0.013))
    else:
        w, lab = orc.mvc_compute(cv, ct, pts)
        moved = orc.mvc_apply(w, cv * np.float32(1.07) + np.float32(0.013))
    out["mvc"] = [w, lab, moved]
    # GrowingSelection::interpolate_poisson_boundary: cage-vertex membrane terms -> tet vertices, with the MVC weights above as gamma coordinates
    r = np.random.default_rng(17)
    n_cv = cv.shape[0]
    inside_d = r.uniform(0.0, 60.0, n_cv).astype(np.float32)
    outside_d = r.uniform(0.5, 80.0, n_cv).astype(np.float32)
    inside_d[::5] = 0.0                                            # empty inside: w_inside = 0
    outside_d[1::7] = inside_d[1::7] * np.float32(0.5) + np.float32(0.01)  # outside thinner than inside: the min(.., 1) clamp and a negative residual
    inside_s = r.normal(scale=0.5, size=(n_cv, 27)).astype(np.float32)
    outside_s = r.normal(scale=0.5, size=(n_cv, 27)).astype(np.float32)
    fn = ref.poisson_interpolate if which == "ref" else orc.poisson_interpolate
    out["poisson_interpolate"] = list(fn(w, inside_d, outside_d, inside_s, outside_s))
    return out


def bitfield_case(scenes, which):
    """grid_to_bitfield + bitfield_max_pool (testbed_nerf.cu:514-555) on the scenes' density grids with the reference's threshold rule"""
    from oracle import oracle as orc
    from oracle import ref
    outs = []
    for key in ("lego", "aabb16"):
        sc = scenes.get(key)
        grid = np.ascontiguousarray(sc.edited_grid, np.float32)
        if which == "ref":
            # update_density_grid_mean_and_bitfield (:3642-3651): mean over the FIRST cascade of max(v, 0) / 128^3, summed by reduce_sum
            # (a tcnn reduction, order unspecified) -- here in double, like the oracle
            mean = float(np.sum(np.maximum(grid[:128 ** 3], 0).astype(np.float32) / np.float32(128 ** 3), dtype=np.float64))
            outs.append(ref.grid_to_bitfield(grid, np.float32(mean)))
        else:
            outs.append(orc.density_grid_to_bitfield(grid))
    return outs


# ---------------------------------------------------------------------------------------------------------------------------------
# deformed-space occupancy refresh: Testbed::update_density_grid_nerf_operator (testbed_nerf.cu:3533-3640)
# ---------------------------------------------------------------------------------------------------------------------------------
def _grid_update(max_cascade, seed=1337):
    import ctypes as C
    from nerfshop_amd import _abi
    from oracle import oracle as orc
    u = _abi.GridUpdate()
    u.n_uniform_samples = _abi.GRID_VOLUME * (max_cascade + 1)
    u.n_nonuniform_samples = 0
    u.reset_grid, u.max_cascade, u.decay, u.ema_step = 0, max_cascade, 0.95, 0
    st, inc = C.c_uint64(), C.c_uint64()
    orc.load().orc_pcg32_seed(C.c_uint64(seed), C.byref(st), C.byref(inc))  # m_rng = default_rng_t{m_seed}, testbed.cu:2220
    u.rng_state, u.rng_inc = st.value, inc.value
    return u


def refresh_cases(scenes, which):
    """-> {name: [grid after each iteration ..., (rng state, ema step)]}: the lego-like scene (geometry in the network) under a cage edit with
    membrane terms, 2 iterations as update_density_grid_nerf_render runs them (all cells with reset, then a mixed uniform / non-uniform
    draw), and one iteration over the 5 cascades of the aabb-16 scene.  The density network is the oracle's in both runs (tiny-cuda-nn)."""
    from nerfshop_amd import _abi
    from oracle import oracle as orc
    from oracle import ref
    out = {}
    for name, key, max_cascade, iters in (("lego_membrane", "lego_shaped", 0, 2), ("aabb16", "aabb16", 4, 1)):
        sc = scenes.get(key)
        e = sc.edit.with_membrane(residual_amplitude=1.0) if name == "lego_membrane" else sc.edit
        mesh = e.tet_mesh_struct()
        grid = np.zeros(5 * 128 ** 3, np.float32)
        u = _grid_update(max_cascade)
        oe = orc.Edit(sc.desc, mesh, keepalive=e) if which == "orc" else None
        res = []
        for it in range(iters):
            u.reset_grid = 1 if it == 0 else 0
            if it == 1 or name == "aabb16":  # outside the first 256 training steps the reference mixes the two draws (testbed_nerf.cu:3508-3511)
                u.n_uniform_samples = _abi.GRID_VOLUME * (max_cascade + 1) // 4
                u.n_nonuniform_samples = _abi.GRID_VOLUME * (max_cascade + 1) // 4
            if which == "ref":
                ref.update_density_grid(sc.desc, [mesh], grid, u, sc.oracle_model)
            else:
                sc.oracle_model.update_density_grid(grid, u, [oe])
            res.append(grid.copy())
        res.append(np.array([u.rng_state, u.ema_step], np.uint64))
        out[name] = res
        sc.oracle_model.set_bitfield(sc.bitfield)  # (the oracle's refresh installs its bitfield in the model)
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# the selection tool: GrowingSelection::project_selection_pixels (growing_selection.cu:1673-1960), get_upper_cell_idx (selection_utils.cu:36)
# ---------------------------------------------------------------------------------------------------------------------------------
def selection_cases(scenes, which):
    """-> {name: [positions, cells, found]} for 6000 scribbled pixels of a 480x270 view, two transmittance thresholds, lego (geometry in the
    network) and aabb 16 (cone stepping, 5 cascades); and get_upper_cell_idx on 10^5 (cell, level) pairs (the product's host function)."""
    from oracle import ref
    out = {}
    for key, az in (("lego_shaped", 50.0), ("aabb16", 200.0)):
        sc = scenes.get(key)
        sc.oracle_model.set_bitfield(sc.bitfield)
        W, H = 480, 270
        p = sc.params_for(W, H, az)
        r = _rng("selection" + key)
        px = np.stack([r.integers(0, W, 6000), r.integers(0, H, 6000)], 1).astype(np.int32)
        for thr in (0.1, 0.5):
            if which == "ref":
                res = ref.project_selection_pixels(sc.desc, p, sc.bitfield, px, sc.oracle_model, thr)
            else:
                res = sc.oracle_model.project_selection_pixels(p, px, thr)
            out[f"{key}_thr{thr}"] = list(res)
    r = _rng("upper")
    level = r.integers(0, 5, N).astype(np.uint32)
    cell = (level * np.uint32(128 ** 3) + r.integers(0, 128 ** 3, N).astype(np.uint32)).astype(np.uint32)
    target = np.minimum(level + r.integers(0, 5, N).astype(np.uint32), 4).astype(np.uint32)
    if which == "ref":
        up = ref.upper_cell_idx(cell, target)
    else:
        from nerfshop_amd import _abi
        lib = _abi.load()
        up = np.array([lib.nrs_upper_cell_idx(int(c), int(t)) for c, t in zip(cell, target)], np.uint32)
    out["upper_cell_idx"] = [up]
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# the network on a regular grid: Testbed::get_density_on_grid / get_rgba_on_grid (testbed_nerf.cu:4538-4613)
# ---------------------------------------------------------------------------------------------------------------------------------
def grid_eval_cases(scenes, which):
    """-> {name: [array]}: raw density on a 48 x 40 x 33 grid over a sub-box (masked by the density grid, and unmasked), premultiplied RGBA for a
    fixed view direction; lego (geometry in the network) and aabb 16"""
    from oracle import ref
    out = {}
    for key in ("lego_shaped", "aabb16"):
        sc = scenes.get(key)
        mn, mx = np.array(sc.desc.aabb_min[:]), np.array(sc.desc.aabb_max[:])
        lo, hi = mn + 0.21 * (mx - mn), mn + 0.83 * (mx - mn)
        res, d = (48, 40, 33), (0.3, -0.5, 0.81)
        if which == "ref":
            out[key] = [ref.density_on_grid(sc.desc, res, lo, hi, sc.oracle_model, sc.grid), ref.density_on_grid(sc.desc, res, lo, hi, sc.oracle_model, None),
                        ref.rgba_on_grid(sc.desc, res, lo, hi, d, sc.oracle_model)]
        else:
            out[key] = [sc.oracle_model.density_on_grid(res, lo, hi, sc.grid), sc.oracle_model.density_on_grid(res, lo, hi, None), sc.oracle_model.rgba_on_grid(res, lo, hi, d)]
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# membrane boundary values: GrowingSelection::compute_poisson_boundary (growing_selection.cu:2220-2348)
# ---------------------------------------------------------------------------------------------------------------------------------
def libc_rand_jitter(seed, n):
    """(float)std::rand() / RAND_MAX after srand(seed), n draws (glibc; the reference jitters its directions this way)"""
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(ctypes.c_uint(seed))
    r = np.array([libc.rand() for _ in range(n)], np.int64)
    return (r.astype(np.float32) / np.float32(2147483647)).astype(np.float32)


def poisson_boundary_cases(scenes, which):
    """-> {name: [density, sh9rgb]} at the cage vertices (outside pass) and at the cage vertices pulled half way to their centroid (inside
    pass, with filter_empty), 10 x 10 directions per vertex, jitter = glibc rand() after srand(77) in both runs"""
    from oracle import ref
    sc = scenes.get("lego_shaped")
    sc.oracle_model.set_bitfield(sc.bitfield)
    cv = np.ascontiguousarray(sc.edit.cage_vertices, np.float32)
    inner = (cv.mean(0) + np.float32(0.5) * (cv - cv.mean(0))).astype(np.float32)
    out = {}
    for name, verts, inside in (("outside", cv, False), ("inside", inner, True)):
        if which == "ref":
            d, sh, jit = ref.poisson_boundary(sc.desc, verts, 10, 10, 77, inside, sc.bitfield, sc.oracle_model)
            assert np.array_equal(jit.reshape(-1), libc_rand_jitter(77, jit.size))
        else:
            jit = libc_rand_jitter(77, verts.shape[0] * 100 * 2).reshape(-1, 2)
            d, sh = sc.oracle_model.poisson_boundary(verts, 10, 10, jit, inside)[:2]
        out[name] = [d, sh]
    return out


def accumulate_cases(which):
    """-> {colour space: [accumulate buffer after 1, 2, 5 frames]}: CudaRenderBuffer::accumulate over seeded frames (values above 1, tiny values below the sRGB
    toe, negative VisPosNeg differences), 48 x 27 pixels"""
    from oracle import oracle as orc
    from oracle import ref
    out = {}
    for name, cs in (("linear", 0), ("srgb", 1), ("visposneg", 2)):
        rng = _rng(f"accumulate{cs}")
        acc = np.full((27, 48, 4), 7.0, np.float32)  # garbage that frame 0 must overwrite
        snaps = []
        for k in range(5):
            f = rng.uniform(0.0, 1.5, (27, 48, 4)).astype(np.float32)
            f[::3, ::5, :3] *= np.float32(1e-3)
            (ref if which == "ref" else orc).accumulate(f, acc, k, cs)
            if k in (0, 1, 4):
                snaps.append(acc.copy())
        out[name] = snaps
    return out
