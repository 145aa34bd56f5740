#!/usr/bin/env python3
"""Wall time per rtx_render_device call against the time its kernels take (GPU box):
the difference is what the host adds per frame (launches, counter read-back, sync).
usage: step_overhead.py [workload] [n_parts]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

rtx = importlib.import_module("ray-tracing_amd")
scenes = importlib.import_module("ray-tracing_amd.scenes")
host = importlib.import_module("ray-tracing_amd.host_py")
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "buddha"
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg, w, h, nl, _ = bench.WORKLOADS[name]
b = host.HostBuilder(10)
d = scenes.assemble(b, cfg)
sc = rtx.Scene(d)
for timing in (True, False):
    p = rtx.default_params(w, h, nl, 1)
    if timing:
        p.flags |= rtx.capi.RTX_FLAG_TIMING
    if parts > 1:
        p.stripe_rows, p.n_parts, p.part = 8, parts, 0
    out = torch.zeros((rtx.part_rows(p), w, 3), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        st = sc.render_device(p, out.data_ptr(), stream)
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    ksum = 0.0
    for _ in range(n):
        st = sc.render_device(p, out.data_ptr(), stream)
        ksum += st.kernel_ms[3]
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    print("%s parts %d timing events %s: wall %.3f ms per call, kernels (trace..merge) %.3f ms" %
          (name, parts, timing, wall, ksum / n))
