#!/usr/bin/env python3
"""Benchmark of the render hot path on 1..8 MI355X of one node.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

A step is one frame of the workload: the image rows are dealt to the N ranks in stripes
(one process per GPU, scene replicated), every rank renders its stripes through the C
ABI (rtx_render_device) into device memory, and the stripes are gathered to rank 0 over
RCCL.  The scene is uploaded once before the timed region.  --inflight frames are in flight
at a time per GPU (default: 1 on one GPU, so that the per-kernel times of the roofline
block are those of a kernel that has the GPU to itself; 2 on two or three GPUs and 3 on four
or more, where a rank's share of the frame is small and the gather and the kernels' last
waves overlap with the next frames' kernels on separate HIP streams); every timed frame is complete, gathered and
assembled before the clock stops.  Rank 0 prints one JSON line:

  value     Mrays/s = (primary rays + shadow rays) of the whole frame / frame time, both
            ray counts as counted by the renderer (pixels; hit records x light samples)
  roofline  the shadow/shade kernel (the dominant one): algorithmic FLOP of one launch
            divided by its HIP-event duration, against the FP32 vector peak
  cpu_baseline  the CPU oracle (a port of the reference's renderer with the same light
            sampler) timed on this host's cores on a bounded sample of the same frame
"""
import argparse
from collections import deque
from concurrent.futures import ThreadPoolExecutor
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# benchmark workloads (BASELINE.json configs) -> (scene, width, height, light samples)
WORKLOADS = {
    "buddha": ("buddha", 3840, 2160, 500,
               "buddha.off, box depth 10, 3840x2160, 500 light samples (BASELINE.json configs[3], the north-star target)"),
    "bunny": ("bunny", 1920, 1080, 100,
              "bunny.off, box depth 10, 1920x1080, 100 light samples (BASELINE.json configs[1])"),
    "dino_spheres": ("dino_spheres", 1920, 1080, 500,
                     "dino.off + sphere, box depth 10, 1920x1080, 500 light samples (BASELINE.json configs[2])"),
    "multi": ("multi", 3840, 2160, 1000,
              "bunny + dino + buddha + sphere, 3840x2160, 1000 light samples (BASELINE.json configs[4])"),
    "default": ("default", 1000, 1000, 1, "scene.h as committed, 1000x1000, 1 light sample"),
    "default_500": ("default", 1000, 1000, 500,
                    "scene.h as committed, 1000x1000, 500 light samples (the frame README.md:11 quotes at '~ 1 minute')"),
}
STRIPE_ROWS = 8

# Algorithmic work per unit (FLOP; a multiply and an add count one each, a division or
# a square root one).
FLOP_PER_PRIM_TEST = 51      # Moeller-Trumbore run to completion (intersect.h:28-67)
FLOP_PER_SHADOW_RAY = 16     # origin step, direction, normalisation (raytracing.h:40-43)
FLOP_PER_SHADED_SAMPLE = 92  # shade(): ~80 + exp + 3 sqrt + 8 div (shade.h:80-90)
BYTES_PER_PRIM_TEST = 36     # nine floats per ray-triangle test, as if fetched each time
PEAK_FP32_VECTOR_TFLOPS = 157.3   # MI355X FP32 vector peak (an FMA counted as 2 FLOP)
PEAK_FP32_UNFUSED_TFLOPS = 78.6   # the same issue rate without FMA contraction
PEAK_HBM_GBS = 8000.0


def counters_file(rtx, workload):
    """What the rocprofv3 PMC passes of a separate run measured for one shade_kernel launch of
    this workload (profiles/hbm_traffic.json, written by tools/hbm_traffic.py): HBM bytes and
    the share of the kernel's cycles in which its SIMDs issue vector instructions.  The file
    names the kernel sources it was measured on; for other sources the figures would describe
    another kernel and are not quoted.  Returns (traffic, valu_issue_utilization, note)."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            data = json.load(f)
    except (OSError, ValueError):
        return None, None, "profiles/hbm_traffic.json not found"
    measured_on = data.get("kernel_source_sha256")
    if measured_on != rtx.kernel_source_hash():
        return None, None, ("profiles/hbm_traffic.json was measured on other kernel sources (%s...), not quoted; "
                            "tools/profile_round.sh refreshes it" % str(measured_on)[:12])
    return (data.get("shade_kernel", {}).get(workload), data.get("valu_issue_utilization", {}).get(workload),
            "rocprofv3 --pmc passes of a separate run on the same kernel sources (sha256 %s...)" % measured_on[:12])


def rt_main_cold(scenes):
    """The reference's own use: one process, one frame of the committed scene (main.cpp:14-31),
    through the command-line renderer: wall time of the process, and the compute time it
    prints (init of the runtime, scene conversion, upload, frame, read-back, PNG)."""
    import re
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "ray-tracing_amd", "host", "rt_main")
    if not os.path.exists(exe):
        return None
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "meshs"))
        for name in ("bunny", "dino"):
            v, f = scenes.load_mesh(name)
            with open(os.path.join(d, "meshs", name + ".off"), "w") as fh:
                fh.write("OFF\n%d %d 0\n" % (len(v), len(f)))
                fh.write("".join("%.9g %.9g %.9g\n" % tuple(float(x) for x in p) for p in v))
                fh.write("".join("3 %d %d %d \n" % tuple(int(x) for x in t) for t in f))
        t0 = time.perf_counter()
        out = subprocess.run([exe], cwd=d, capture_output=True, text=True, timeout=300)
        wall = (time.perf_counter() - t0) * 1e3
        m = re.search(r"Compute time \(ms\): ([0-9.]+)", out.stdout)
        if out.returncode != 0 or not m:
            return {"error": (out.stderr or out.stdout)[-200:]}
        return {"process_ms": wall, "compute_ms": float(m.group(1)),
                "what": "rt_main, scene.h as committed, 1000x1000, 1 light sample, out.png written; compute_ms is "
                        "what it prints for render_image (main.cpp:17-31), process_ms the whole process"}


def build_scene(rtx, scenes, name):
    """Scene description through the product's host layer (C++), not the oracle."""
    host = importlib.import_module("ray-tracing_amd.host_py")
    builder = host.HostBuilder(10)
    desc = scenes.assemble(builder, name)
    return builder, desc


def cpu_baseline(scenes, name, w, h, nl):
    """The CPU oracle on this host's cores, on every second stripe of the same frame."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    b = oracle_py.OracleBuilder(10)
    d = scenes.assemble(b, name)
    p = oracle_py.capi.default_params(w, h, nl, 1)
    sample = "every 2nd stripe of %d rows (half the frame)" % STRIPE_ROWS
    p.stripe_rows, p.n_parts, p.part = STRIPE_ROWS, 2, 0
    # keep it bounded: thin the sample until a pilot says it fits in ~20 s
    pilot = oracle_py.capi.default_params(w, h, nl, 1)
    pilot.stripe_rows, pilot.n_parts, pilot.part = STRIPE_ROWS, 32, 15
    t0 = time.perf_counter()
    oracle_py.render(d, pilot, want_hits=False, want_rgb=False)
    t_pilot = time.perf_counter() - t0
    if t_pilot * 16 > 25.0:
        parts = 32 if t_pilot > 12.0 else max(2, int(round(t_pilot * 32 / 20.0)))
        p.n_parts, p.part = parts, parts // 2
        sample = "one stripe of %d rows in every %d" % (STRIPE_ROWS, parts)
    t0 = time.perf_counter()
    _, _, _, st = oracle_py.render(d, p, want_hits=False, want_rgb=False)
    dt = time.perf_counter() - t0
    rays = st.primary_rays + st.shadow_rays
    threads = oracle_py.oracle().orc_max_threads()
    # one thread, on a thinner sample of the same frame sized to ~8 s from the rate just measured
    parts1 = max(2, min(256, int(round(p.n_parts * threads * dt / 8.0))))
    p1 = oracle_py.capi.default_params(w, h, nl, 1)
    p1.stripe_rows, p1.n_parts, p1.part = STRIPE_ROWS, parts1, parts1 // 2
    t0 = time.perf_counter()
    _, _, _, st1 = oracle_py.render(d, p1, want_hits=False, want_rgb=False, threads=1)
    dt1 = time.perf_counter() - t0
    rays1 = st1.primary_rays + st1.shadow_rays
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "%s of %s, %d rays in %.2f s" % (sample, name, rays, dt),
            "one_thread": {"value": rays1 / dt1 / 1e6, "unit": "Mrays/s", "cores": 1,
                           "sample": "one stripe of %d rows in every %d of %s, %d rays in %.2f s"
                                     % (STRIPE_ROWS, parts1, name, rays1, dt1)}}


def spawn_ranks(n):
    """One child process per GPU (the same command line, RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in its environment), as `python -m torch.distributed.run --nproc-per-node n`
    would start them.  Rank 0 prints the JSON line on the inherited stdout.  Returns the
    exit code: 0 when every rank succeeded, else the first failure's (the other ranks are
    then terminated by their process ids)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    children = []
    # a rank that hangs (a peer that never joins the rendezvous, a collective one rank never
    # enters) must end the job with an error, not block the caller for ever: every rank dumps
    # its Python stacks to stderr shortly before the limit, then all are terminated
    limit = float(os.environ.get("RTX_BENCH_TIMEOUT", "1500"))
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RTX_BENCH_STACKS_AFTER=str(max(limit - 10.0, 1.0)))
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, live = 0, list(children)
    t_start = time.monotonic()
    while live:
        time.sleep(0.05)
        if time.monotonic() - t_start > limit:
            print("bench.py: ranks still running after %.0f s, terminating them" % limit, file=sys.stderr, flush=True)
            for c in live:
                c.terminate()
            for c in live:
                try:
                    c.wait(10)
                except subprocess.TimeoutExpired:
                    c.kill()
            return 124
        for c in list(live):
            code = c.poll()
            if code is None:
                continue
            live.remove(c)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code
                for other in live:          # a rank failed: the others would wait for it forever
                    other.terminate()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="buddha", choices=sorted(WORKLOADS))
    ap.add_argument("--inflight", type=int, default=0,
                    help="frames in flight per GPU; 0 (default) = 1 on one GPU, 2 on two or three, 3 on more (own scene replica, stream and output tile each); with 2 "
                         "the kernels of consecutive frames overlap: more frames per second, but a kernel's "
                         "HIP-event duration then includes the time it shares the GPU, so the roofline "
                         "figures are only meaningful with 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-steps", type=int, default=5,
                    help="frames rendered through rtx_render (host output, PCIe read-back included) after "
                         "the timed region, reported as ms_per_step_host; 0 = skip")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed region compare the assembled frames with a full-frame render")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: this process becomes the launcher of
        # the N ranks.  It has not touched the GPU (nor imported torch) at this point.
        raise SystemExit(spawn_ranks(args.gpus))

    if "RTX_BENCH_STACKS_AFTER" in os.environ:
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["RTX_BENCH_STACKS_AFTER"]), exit=False)
    backend = os.environ.get("RTX_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        # all ranks are on this node: bind gloo to the loopback interface instead of whatever
        # the host name resolves to (it may not resolve at all, or only after a long time-out)
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    # Rehearsal of the N > 1 path on a box with fewer GPUs than ranks (not a measurement):
    # RTX_BENCH_SHARE_GPU=1 puts rank r on GPU r % device_count, RTX_BENCH_BACKEND=gloo
    # replaces RCCL, which refuses two ranks on one device.
    if os.environ.get("RTX_BENCH_SHARE_GPU") == "1":
        local_rank %= torch.cuda.device_count()
    elif world > 1 and torch.cuda.device_count() == 1:
        local_rank = 0   # the launcher shows every rank its own GPU only
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    rtx = importlib.import_module("ray-tracing_amd")
    scenes = importlib.import_module("ray-tracing_amd.scenes")
    name, w, h, nl, label = WORKLOADS[args.workload]

    builder, desc = build_scene(rtx, scenes, name)
    p = rtx.default_params(w, h, nl, 1)
    p.flags |= rtx.capi.RTX_FLAG_TIMING
    if world > 1:
        p.stripe_rows, p.n_parts, p.part = STRIPE_ROWS, world, rank
    tiles = importlib.import_module("ray-tracing_amd.tiles")

    # Frames in flight: every slot has its own scene replica (uploaded once, resident in HBM),
    # HIP stream, output tile and gather buffers.  A worker thread per slot sits in the
    # blocking render call; the exchange step of the finished frames runs on the main thread
    # in frame order, so all ranks issue their collectives in the same order.  With more than
    # one slot a frame's low-occupancy phases (the trace kernel's heaviest tiles, the last
    # records of the shading kernel, the gather) overlap with the next frame's kernels.
    # (measured on one GPU rendering its share of the buddha frame, tools/share_rate.py: half the
    # frame 1.53 / 1.41 / 1.40 ms per frame with 1 / 2 / 3 in flight, a quarter 0.92 / 0.84 / 0.78, an
    # eighth 0.58 / 0.48 / 0.45)
    n_slots = args.inflight if args.inflight > 0 else (1 if world == 1 else 2 if world < 4 else 3)
    # cold path: the first scene of the process on this device (code object load, sampled
    # arithmetic self-test, conversion, upload) and a second one (conversion and upload only)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    first_scene = rtx.Scene(desc, local_rank)
    create_first_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    replicas = [rtx.Scene(desc, local_rank) for _ in range(n_slots)]
    create_again_ms = (time.perf_counter() - t0) * 1e3 / n_slots
    first_scene.close()
    # (one slot: everything on the current stream, which also orders tile reuse)
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_slots)] if n_slots > 1 else [torch.cuda.current_stream()]
    gathers = [tiles.FrameGather(torch, dist, h, w, STRIPE_ROWS, world, rank, dev) for _ in range(n_slots)]
    tile_free = [None] * n_slots      # event after which a slot's tile may be overwritten
    pool = ThreadPoolExecutor(max_workers=n_slots)

    def render(slot):
        torch.cuda.set_device(local_rank)          # (the current device is per thread)
        if tile_free[slot] is not None:
            streams[slot].wait_event(tile_free[slot])
        return replicas[slot].render_device(p, gathers[slot].tile.data_ptr(), streams[slot].cuda_stream)

    gather_events = []

    def exchange(slot):
        if world > 1:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        gathers[slot].run()   # RCCL gather to rank 0 + scatter of the stripes into the frame
        if world > 1:
            ev1.record()
            gather_events.append((ev0, ev1))
        if n_slots > 1:
            tile_free[slot] = torch.cuda.Event()
            tile_free[slot].record()

    def run_frames(n):
        """n steps; returns the stats of every frame."""
        pending, done = deque(), []
        if n_slots == 1:                      # nothing to overlap: no thread hand-off either
            for _ in range(n):
                done.append(render(0))
                exchange(0)
            return done
        for f in range(n):
            slot = f % n_slots
            pending.append((slot, pool.submit(render, slot)))
            if len(pending) == n_slots:
                s_done, fut = pending.popleft()
                done.append(fut.result())
                exchange(s_done)
        while pending:
            s_done, fut = pending.popleft()
            done.append(fut.result())
            exchange(s_done)
        return done

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the first frame of a scene handle (work buffers allocated, perhaps grown and the frame
    # rendered again), then the second
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st_first = render(0)
    exchange(0)
    fence()
    first_frame_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    render(0)
    exchange(0)
    fence()
    second_frame_ms = (time.perf_counter() - t0) * 1e3
    # the other slots' first frames (their replicas allocate their work buffers), one at a time
    for slot in range(1, n_slots):
        render(slot)
        exchange(slot)
        fence()

    run_frames(args.warmup)
    fence()
    del gather_events[:]
    t0 = time.perf_counter()
    frames = run_frames(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    # the exchange step on the stream it runs on (gather + scatter of the stripes into the frame)
    gather_ms = sum(a.elapsed_time(b) for a, b in gather_events) / max(len(gather_events), 1)
    st = frames[-1]
    trace_ms = sum(f.kernel_ms[0] for f in frames)
    shade_ms = sum(f.kernel_ms[1] for f in frames)
    merge_ms = sum(f.kernel_ms[2] for f in frames)
    screen_ms = sum(f.screen_ms for f in frames)
    counts = torch.tensor([elapsed, float(st.primary_rays), float(st.shadow_rays), float(st.shadow_tests),
                           float(st.hits), float(st.lit_samples), float(st.shadow_tests_run)],
                          dtype=torch.float64, device=dev)
    # per rank, for the slowest-rank columns of an N > 1 line
    per_rank = torch.tensor([trace_ms / args.steps, screen_ms / args.steps, shade_ms / args.steps, merge_ms / args.steps,
                             gather_ms, float(st.hits)], dtype=torch.float64, device=dev)
    rank_max, rank_min = per_rank.clone(), per_rank.clone()
    if world > 1:
        mx = counts.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        counts[0] = mx[0]
        dist.all_reduce(rank_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(rank_min, op=dist.ReduceOp.MIN)
    elapsed, primary, shadow, tests, hits, lit, tests_run = [float(x) for x in counts.tolist()]
    rank_max, rank_min = rank_max.tolist(), rank_min.tolist()

    ranks_seen = dist.get_world_size() if world > 1 else 1
    # the same frames through rtx_render (host output buffer: the PCIe read-back is part of
    # the call), outside the timed region; never `value`
    host_ms = None
    if args.host_steps > 0:
        host_frame = rtx.HostFrame(rtx.part_rows(p), w)     # page-locked output buffer
        replicas[0].render_into(p, host_frame)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.host_steps):
            replicas[0].render_into(p, host_frame)
        host_ms = (time.perf_counter() - t0) / args.host_steps * 1e3
        host_frame.close()
        if world > 1:
            hm = torch.tensor([host_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(hm, op=dist.ReduceOp.MAX)
            host_ms = float(hm.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        rays = primary + shadow
        # roofline of the shadow/shade kernel on this rank (rank 0's share of the frame):
        # the work the kernel EXECUTED (tests left after its screens, rays, shaded samples)
        # over its average HIP-event duration, against the FP32 vector peak
        k = args.steps
        shade_avg_ms = shade_ms / k
        sec = shade_avg_ms * 1e-3
        flop_alg = (st.shadow_tests * FLOP_PER_PRIM_TEST + st.shadow_rays * FLOP_PER_SHADOW_RAY +
                    st.lit_samples * FLOP_PER_SHADED_SAMPLE)
        flop_run = (st.shadow_tests_run * FLOP_PER_PRIM_TEST + st.shadow_rays * FLOP_PER_SHADOW_RAY +
                    st.lit_samples * FLOP_PER_SHADED_SAMPLE)
        achieved = flop_run / sec / 1e12 if sec > 0 else 0.0
        traffic, valu_issue, counters_note = counters_file(rtx, args.workload)
        out = {
            "metric": "Mrays/s (primary+shadow)",
            "value": rays / (elapsed / args.steps) / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_step_host": host_ms,
            # the same frame timed through rtx_render: host output buffer, PCIe read-back included
            "value_host": (rays / (host_ms * 1e-3) / 1e6) if host_ms else None,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "reference mesh geometry (%s), counter-based synthetic light samples, seed 1" % name,
            "config": {"workload": label, "stripe_rows": STRIPE_ROWS, "frames_in_flight": n_slots,
                       "parallelism": "row stripes dealt to %d GPU(s), scene replicated, RCCL gather" % world,
                       "primary_rays": int(primary), "shadow_rays": int(shadow), "hit_records": int(hits)},
            "roofline": {
                "kernel": "shade_kernel", "bound": "valu",
                "bound_note": "FP32 vector ALUs (no MFMA is issued: the path has no dense contraction). peak "
                              "counts a fused multiply-add as 2 FLOP; bit-exact hit ids forbid contraction, so "
                              "packed un-fused arithmetic tops out at peak_unfused",
                "achieved": achieved, "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_VECTOR_TFLOPS,
                "peak_unfused": PEAK_FP32_UNFUSED_TFLOPS,
                "frac_unfused": achieved / PEAK_FP32_UNFUSED_TFLOPS,
                "traffic": traffic,
                "launch_ms": shade_avg_ms,
                "launch_ms_note": ("kernel alone on the GPU" if n_slots == 1 else
                                   "%d frames in flight: the kernel shares the GPU with the other frame's kernels, "
                                   "its duration (and frac) is not that of the kernel alone" % n_slots),
                "executed_flop_per_launch": flop_run,
                "flop_per_unit": {"prim_test": FLOP_PER_PRIM_TEST, "shadow_ray": FLOP_PER_SHADOW_RAY,
                                  "shaded_sample": FLOP_PER_SHADED_SAMPLE},
                "units_per_launch": {"prim_tests_run": int(st.shadow_tests_run), "shadow_rays": int(st.shadow_rays),
                                     "shaded_samples": int(st.lit_samples)},
                "tests_run_fraction": st.shadow_tests_run / max(st.shadow_tests, 1),
                # measured, not derived from the FLOP count: SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES of the
                # kernel (tools/hbm_traffic.py; good to a few per cent).  Near 1 = the kernel is bound by
                # vector issue slots; frac then measures how many of them are the counted FLOP
                "valu_issue_utilization": valu_issue,
                "counters_note": counters_note,
                # what the reference's algorithm would have done for the same rays (every leaf
                # primitive tested for every shadow ray): a speed-up equivalent, not a hardware rate
                "algorithmic_equivalent": {
                    "flop_per_launch": flop_alg,
                    "tflops": flop_alg / sec / 1e12 if sec > 0 else 0.0,
                    "bytes_per_launch": st.shadow_tests * BYTES_PER_PRIM_TEST,
                    "note": "work of the reference's loop nest / kernel time; the kernel proves most of those "
                            "tests unnecessary per record, so this can exceed the hardware peak"},
                "hbm": {"bound": "hbm", "achieved": (traffic / sec / 1e9) if (traffic and sec > 0) else None,
                        "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": (traffic / sec / 1e9 / PEAK_HBM_GBS) if (traffic and sec > 0) else None,
                        "note": "measured HBM bytes of one launch (rocprofv3 PMC passes, profiles/hbm_traffic.json) "
                                "/ launch time: the scene is cache resident, HBM is not the bound"},
            },
            # trace: tile list, primary rays, work lists; screen: the records' candidate lists
            # (screen_kernel); shade: the shadow-ray + BSDF kernel the roofline block describes; merge
            "kernel_ms": {"trace": trace_ms / k, "screen": screen_ms / k, "shade": shade_avg_ms, "merge": merge_ms / k},
            # cold path of this process (rank 0): scene creation and the first frames of a handle
            "cold": {"scene_create_first_ms": create_first_ms, "scene_create_again_ms": create_again_ms,
                     "first_minus_again_ms": create_first_ms - create_again_ms,
                     "first_minus_again_is": "code object load, sampled arithmetic self-test, address of the static work area",
                     "first_frame_ms": first_frame_ms, "first_frame_attempts": int(st_first.render_attempts),
                     "second_frame_ms": second_frame_ms, "record_capacity": int(st_first.record_capacity)},
        }
        if world > 1:
            # slowest and fastest rank, per frame: a poor scaling point shows here whether it is the
            # kernels of the heaviest share, the spread between the shares, or the exchange step
            names = ("trace_ms", "screen_ms", "shade_ms", "merge_ms", "gather_ms", "hit_records")
            out["ranks"] = {"max": dict(zip(names, rank_max)), "min": dict(zip(names, rank_min)),
                            "note": "per-frame averages over the timed steps; kernel times are HIP events on the "
                                    "rank's stream (with several frames in flight they include time shared with "
                                    "other frames' kernels); gather_ms = gather + scatter into the frame on rank 0's stream"}
        if world == 1 and not args.no_cpu_baseline:
            out["cold"]["rt_main"] = rt_main_cold(scenes)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scenes, name, w, h, nl)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if args.verify and rank == 0:
        # the assembled frames of all slots against one full-frame render on this GPU
        full, _ = replicas[0].render(rtx.default_params(w, h, nl, 1))
        for k, g in enumerate(gathers[:min(n_slots, args.warmup + args.steps)]):
            same = bool((g.frame.cpu().numpy() == full).all())
            print("verify slot %d: assembled frame %s the single-GPU frame" % (k, "equals" if same else "DIFFERS FROM"),
                  file=sys.stderr, flush=True)
            if not same:
                raise SystemExit(1)
    pool.shutdown()
    for scene in replicas:
        scene.close()


if __name__ == "__main__":
    main()
