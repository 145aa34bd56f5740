"""`python bench.py --gpus 2` on a one-GPU box: the launcher path of the multi-GPU bench.

In a file of its own, collected after the parity tests: it starts three processes that share
one GPU and talk over gloo, and a stall there must not hide the results of the other tests."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with nothing else: the process starts its two ranks itself
    (one per GPU on a multi-GPU node; here both share the box's GPU and talk over gloo,
    a rehearsal of the path, not a measurement) and rank 0 prints the line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RTX_BENCH_BACKEND="gloo", RTX_BENCH_SHARE_GPU="1", RTX_BENCH_TIMEOUT="240",
               GLOO_SOCKET_IFNAME="lo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "bunny",
                          "--steps", "3", "--warmup", "1", "--verify"], capture_output=True, text=True,
                         timeout=400, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["value"] > 0
    assert "verify slot 0: assembled frame equals" in out.stderr
