#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh.

    python tools/hbm_traffic.py gpurun_out/round profiles/hbm_traffic.json

Per workload: the last dispatch of the shading kernel in pmc_fetch_<w>.txt and
pmc_write_<w>.txt (summaries by tools/pmc_summary.py; counters in KB)."""
import importlib
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


N_SIMDS = 1024  # 256 CUs x 4
N_SQ = 32       # 8 XCDs x 4 shader engines, one SQ_BUSY_CYCLES counter each


def last_shade(path, counter):
    val = None
    for line in open(path):
        if "shade_kernel" in line or "shade_few_kernel" in line:
            m = re.search(counter + r"=([0-9.e+]+)", line)
            if m:
                val = float(m.group(1))
    return val


def main():
    src, dst = sys.argv[1], sys.argv[2]
    out = {"note": "HBM bytes per launch of the shading kernel = (FETCH_SIZE + WRITE_SIZE) * 1024, raw rocprofv3 "
                   "counters from separate --pmc runs (profiles/r<round>_pmc_hbm_<workload>.txt).  On gfx950 FETCH_SIZE "
                   "reports half the bytes of wide coalesced streaming reads; this kernel reads through scalar "
                   "loads and dword scratch reloads (uncalibrated widths), so the true figure lies between "
                   "shade_kernel[w] and shade_kernel_upper[w] = (2 * FETCH_SIZE + WRITE_SIZE) * 1024",
           "shade_kernel": {}, "shade_kernel_upper": {}, "fetch_kb": {}, "write_kb": {}}
    # the kernels these counters were taken on: bench.py quotes them only for the same sources
    out["kernel_source_sha256"] = importlib.import_module("ray-tracing_amd").kernel_source_hash()
    try:   # workloads measured earlier on the same kernel sources stay
        old = json.load(open(dst))
        if old.get("kernel_source_sha256") == out["kernel_source_sha256"]:
            for k, v in old.items():
                if isinstance(v, dict):
                    out[k] = dict(v)
    except (OSError, ValueError):
        pass
    for w in ("buddha", "bunny", "dino_spheres", "multi", "default", "default_500"):
        try:
            f = last_shade("%s/pmc_fetch_%s.txt" % (src, w), "FETCH_SIZE")
            wr = last_shade("%s/pmc_write_%s.txt" % (src, w), "WRITE_SIZE")
        except OSError:
            continue
        if f is None or wr is None:
            continue
        out["fetch_kb"][w], out["write_kb"][w] = f, wr
        out["shade_kernel"][w] = int((f + wr) * 1024)
        out["shade_kernel_upper"][w] = int((2 * f + wr) * 1024)
        # vector-issue utilisation of the same kernel from the SQ pass: SQ_ACTIVE_INST_VALU (quad-cycles,
        # summed over the SIMDs) against SQ_BUSY_CYCLES (cycles, summed over the 32 shader engines' SQs)
        try:
            act = last_shade("%s/pmc_sq_%s.txt" % (src, w), "SQ_ACTIVE_INST_VALU")
            busy = last_shade("%s/pmc_sq_%s.txt" % (src, w), "SQ_BUSY_CYCLES")
            dur = last_shade("%s/pmc_sq_%s.txt" % (src, w), "_dur_ms")
        except OSError:
            act = busy = dur = None
        if act and busy:
            out.setdefault("valu_issue_utilization", {})[w] = round((act * 4 / N_SIMDS) / (busy / N_SQ), 4)
            if dur:
                out.setdefault("shader_clock_ghz", {})[w] = round(busy / N_SQ / (dur * 1e-3) / 1e9, 3)
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["shade_kernel"]))


main()
