"""Row tiling and the multi-process gather (world size 2 and 3 over gloo on the CPU)."""
import importlib
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tiles():
    return importlib.import_module("ray-tracing_amd.tiles")


def test_rows_partition_the_frame(tiles, capi):
    for h, stripe, parts in [(1080, 8, 8), (2160, 8, 4), (90, 8, 3), (7, 8, 2), (1000, 1, 8)]:
        seen = np.concatenate([tiles.rows_of_part(h, stripe, parts, k) for k in range(parts)])
        assert sorted(seen) == list(range(h))
        for k in range(parts):
            p = capi.default_params(4, h)
            p.stripe_rows, p.n_parts, p.part = stripe, parts, k
            assert capi.part_rows(p) == len(tiles.rows_of_part(h, stripe, parts, k))
        assert tiles.max_part_rows(h, stripe, parts) == len(tiles.rows_of_part(h, stripe, parts, 0))


def test_assemble_numpy(tiles):
    h, w = 45, 7
    frame = np.random.default_rng(1).integers(0, 256, (h, w, 3), dtype=np.uint8)
    parts = [frame[tiles.rows_of_part(h, 4, 3, k)] for k in range(3)]
    assert np.array_equal(tiles.assemble_numpy(parts, h, 4), frame)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_gather_over_gloo(tmp_path, world):
    out = str(tmp_path / "frames.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world),
               OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), out],
                              env=dict(env, RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    got, want = np.load(out)
    assert np.array_equal(got, want)


def test_bench_launcher_reports_failing_ranks():
    """`python bench.py --gpus 2` starts its own ranks; without a GPU every rank refuses to
    run, and the launcher must end with a non-zero exit code (not hang, not print a line)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "bunny"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and out.stdout.strip() == ""
    assert out.stderr.count("bench.py needs a GPU") == 2
