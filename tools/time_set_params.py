"""ms per nrs_model_set_params_device (grid copy + weight fragments + rebuild of the cell records) on the bench scene.  usage: python tools/time_set_params.py [workload]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import bench
    from nerfshop_amd import runtime as rt, synth
    ctx = rt.Context(0)
    scene = bench.build_scene(sys.argv[1] if len(sys.argv) > 1 else "lego_cage", rt, synth, ctx, torch)
    net = scene["tb"].nerf_network
    blob = torch.from_numpy(scene["params"].view(np.int16)).cuda()
    for _ in range(3):
        net.set_params_device(blob)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        net.set_params_device(blob)
    torch.cuda.synchronize()
    print(f"set_params_device: {(time.perf_counter() - t0) * 1e3 / K:.3f} ms per call, cell records {net.cell_cache()[0] / 1e9:.1f} GB")


if __name__ == "__main__":
    main()
