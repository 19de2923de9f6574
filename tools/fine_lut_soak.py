"""Soak of the fine look-up table under the cage's cell -> tet LUT (round 6: nrs_cage.hip fine_lists_kernel, nrs_device.cuh find_tet): random cages -- lattice
sizes, translations, twists up to folded tets, both scene scales -- and, per cage, 2^20 positions (a fifth of them ON vertices / edge midpoints / faces of tets, where
two tets pass the containment test and the ORDER of the cell's list decides), map_rays and map_positions bit for bit against the oracle (interpolate_tet,
cage_deformation.cu:197-269 as restated in oracle/nrs_oracle.cpp and pinned to the reference's compiled code by tests/test_ref_pin.py).  Too long for the test tier
(tests/test_gpu_fine_lut.py and test_gpu_parity.py::test_map_rays_bit_exact are its short forms); run through gpurun:
    python tools/fine_lut_soak.py [n_cages] [device]
With `device` the operators are created with device authoring (nrs_edit_create builds the cell -> tet LUT, the canonical bitfield, rotations and plane records with the
kernels of nrs_cage.hip -- the LUT passes rewritten in round 6) and their tables are held to the host builder's first (offsets, ascending lists, bitfield: bit for bit).
Prints one line per cage and a summary line; exit code 1 when any bit differs.
Round 6's runs (profiles/r06/fine_lut_soak.txt, device_lut_soak.txt): 30 cages, 31 457 280 positions (16.3 M carried back by a tet, 2.5 M emptied), cages with any
differing bit: 0 -- with host-built tables, and again with the tables built on the device by the rewritten LUT passes (lists of up to 2 115 tets)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa: E402
from nerfshop_amd import runtime, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

n_cages = int(sys.argv[1]) if len(sys.argv) > 1 else 36
device = len(sys.argv) > 2 and sys.argv[2] == "device"
rng = np.random.default_rng(606)
ctx = runtime.Context(0)
bad_cages, total, moved_total, empty_total = 0, 0, 0, 0
for k in range(n_cages):
    aabb_scale = 16 if k % 4 == 3 else 1
    scale = 1.0 if aabb_scale == 1 else 6.0
    lattice = int(rng.choice([4, 5, 6, 8, 10, 12]))
    translate = tuple(float(v) for v in rng.uniform(-0.15, 0.15, 3))
    twist = float(rng.choice([0.0, 10.0, 33.0, 75.0, 140.0]))  # (140 degrees folds tets over each other: overlapping candidates)
    inflate = float(rng.choice([0.05, 0.2]))
    desc = synth.model_desc(aabb_scale)
    edit = synth.make_cage_edit(lattice_n=lattice, translate=translate, twist_deg=twist, inflate=inflate, scene_scale=scale, copy=bool(k % 5 == 4))
    ref = orc.Edit(desc, edit.tet_mesh_struct(), keepalive=edit)
    op = runtime.CageDeformation(ctx, desc, edit, device_authoring=device)
    table_diff = 0
    if device:
        got = op.download(rotations=False)
        table_diff = int(not np.array_equal(got["lut_offsets"], edit.lut_offsets)) + int(not np.array_equal(got["lut_idx"], edit.lut_idx[: got["lut_idx"].size])) + \
            int(not np.array_equal(got["original_bitfield"], edit.original_bitfield)) + int(op.lut_size() != (int(edit.lut_offsets[-1]), int(edit.max_per_cell)))
    mn, mx = synth.scene_aabb(aabb_scale)
    mn, mx = np.array(mn, np.float32), np.array(mx, np.float32)
    n = 1 << 20
    verts = edit.vertices
    lo, hi = verts.min(0), verts.max(0)
    ext = hi - lo
    world = rng.uniform(lo - 0.03 * ext, hi + 0.03 * ext, size=(n, 3)).astype(np.float32)
    t = edit.tets[rng.integers(0, edit.tets.shape[0], n // 5)]
    w = rng.dirichlet([0.4, 0.4, 0.4, 0.4], size=n // 5).astype(np.float32)
    w[: n // 15] = np.eye(4, dtype=np.float32)[rng.integers(0, 4, n // 15)]
    w[n // 15: 2 * n // 15] = 0.5 * (np.eye(4, dtype=np.float32)[rng.integers(0, 4, n // 15)] + np.eye(4, dtype=np.float32)[rng.integers(0, 4, n // 15)])
    world[: n // 5] = np.einsum("nk,nkd->nd", w, verts[t]).astype(np.float32)
    world[n // 10: n // 5] += rng.normal(0, 2e-7, size=(n // 5 - n // 10, 3)).astype(np.float32) * np.float32(scale)
    # a tenth around the canonical mesh (the empty-mask branch, cage_deformation.cu:254-266)
    olo, ohi = edit.original_vertices.min(0), edit.original_vertices.max(0)
    world[n // 5: n // 5 + n // 10] = rng.uniform(olo - 0.02 * (ohi - olo), ohi + 0.02 * (ohi - olo), size=(n // 10, 3)).astype(np.float32)
    c = np.zeros((n, 7), np.float32)
    c[:, :3] = (world - mn) / (mx - mn)
    c[:, 3] = 1e-3
    c[:, 4:] = rng.uniform(0, 1, size=(n, 3)).astype(np.float32)
    ref_c, ref_empty = ref.map_rays(c)
    dc = torch.from_numpy(c).cuda()
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    op.map_rays(None, dc, mask)
    got, got_mask = dc.cpu().numpy(), mask.cpu().numpy()
    pos = np.ascontiguousarray(c[:, :3])
    ref_p, ref_e2 = ref.map_positions(pos)
    dp = torch.from_numpy(pos).cuda()
    mask2 = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    op.map_positions(None, dp, mask2)
    diff = int((got.view(np.uint32) != ref_c.view(np.uint32)).any(axis=1).sum()) + int((got_mask != ref_empty).sum())
    diff += int((dp.cpu().numpy().view(np.uint32) != ref_p.view(np.uint32)).any(axis=1).sum()) + int((mask2.cpu().numpy() != ref_e2).sum())
    diff += table_diff
    moved = int((ref_c[:, :3] != c[:, :3]).any(axis=1).sum())
    print(f"cage {k:3d}: aabb {aabb_scale:2d} lattice {lattice:2d} twist {twist:5.1f} inflate {inflate:.2f} copy {int(k % 5 == 4)}: {n} positions, {moved} carried back, "
          f"{int(ref_empty.sum())} emptied, {int(edit.lut_offsets[-1])} LUT entries (longest list {int(edit.max_per_cell)}), rows / tables with ANY differing bit: {diff}", flush=True)
    bad_cages += 1 if diff else 0
    total += n; moved_total += moved; empty_total += int(ref_empty.sum())
    del op, ref
print(f"fine-lut soak: {n_cages} cages, {total} positions ({moved_total} carried back by a tet, {empty_total} emptied), cages with ANY differing bit: {bad_cages}")
sys.exit(1 if bad_cages else 0)
