#!/usr/bin/env python3
"""Frames per second of ONE GPU's share of a frame split N ways, with 1..3 frames in flight
(own scene replica, stream and host thread each, as bench.py does on every rank): what a rank
of an N-GPU run sustains before the gather.  usage: share_rate.py [workload] [n_parts]"""
import importlib
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

rtx = importlib.import_module("ray-tracing_amd")
scenes = importlib.import_module("ray-tracing_amd.scenes")
host = importlib.import_module("ray-tracing_amd.host_py")
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "buddha"
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg, w, h, nl, _ = bench.WORKLOADS[name]
b = host.HostBuilder(10)
d = scenes.assemble(b, cfg)
p = rtx.default_params(w, h, nl, 1)
if parts > 1:
    p.stripe_rows, p.n_parts, p.part = 8, parts, 0
for slots in (1, 2, 3, 4):
    replicas = [rtx.Scene(d) for _ in range(slots)]
    streams = [torch.cuda.Stream() for _ in range(slots)]
    outs = [torch.zeros((rtx.part_rows(p), w, 3), dtype=torch.uint8, device="cuda") for _ in range(slots)]
    n = 60

    def work(k, count):
        torch.cuda.set_device(0)
        for _ in range(count):
            replicas[k].render_device(p, outs[k].data_ptr(), streams[k].cuda_stream)

    for k in range(slots):
        work(k, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    threads = [threading.Thread(target=work, args=(k, n)) for k in range(slots)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s 1/%d of the frame, %d in flight: %.3f ms per frame" % (name, parts, slots, dt / (n * slots) * 1e3))
    for r in replicas:
        r.close()
