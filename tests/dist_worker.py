"""One rank of the CPU (gloo) rehearsal of the multi-GPU frame path: renders its row
stripes with the CPU oracle and runs the same FrameGather that bench.py uses on GPUs."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as op  # noqa: E402

tiles = importlib.import_module("ray-tracing_amd.tiles")
scenes = importlib.import_module("ray-tracing_amd.scenes")
capi = op.capi


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out = sys.argv[1]
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # all ranks are on this host
    w, h, nl, seed, stripe = 160, 90, 3, 5, 8
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = op.OracleBuilder(6)
    d = scenes.assemble(b, "dino_spheres")
    p = capi.default_params(w, h, nl, seed)
    p.stripe_rows, p.n_parts, p.part = stripe, world, rank
    rgb, _, _, _ = op.render(d, p, want_hits=False)
    g = tiles.FrameGather(torch, dist, h, w, stripe, world, rank, torch.device("cpu"))
    g.tile[:rgb.shape[0]] = torch.from_numpy(rgb)
    frame = g.run()
    if rank == 0:
        full, _, _, _ = op.render(d, capi.default_params(w, h, nl, seed), want_hits=False)
        np.save(out, np.stack([frame.numpy(), full]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
