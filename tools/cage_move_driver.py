"""The per-gizmo-move chain (nrs_edit_update_cage: MVC apply + bbox + cell -> tet LUT + rotations + plane records; tet_mesh.cu:368-673) called back to back, host-timed;
run under `rocprofv3 --kernel-trace --stats` for the kernels' own share of a move (profiles/r06/cage_move_kernels.md).  usage: python tools/cage_move_driver.py [lattice_n] [reps]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from nerfshop_amd import runtime as rt, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ctx = rt.Context(0)
desc = synth.model_desc(1)
e = synth.make_cage_edit(lattice_n=n)
op = rt.CageDeformation(ctx, desc, e, device_authoring=True)
op.set_mvc(e.mvc_weights)
poses = [synth.deform_cage(e.cage_vertices, (0.10 * k / 10, 0.05, 0.0), 20.0 * k / 10) for k in range(1, 11)]
op.update_cage(None, poses[0]); torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(reps):
    op.update_cage(None, poses[k % 10])
torch.cuda.synchronize()
print(f"cage move, {e.tets.shape[0]} tets: {(time.perf_counter() - t0) * 1e3 / reps:.3f} ms per move (host-timed, {reps} moves)")
