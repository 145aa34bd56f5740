"""Frames of the CPU oracle against the golden frames that make_golden.py rendered by
composing the reference's own intersect_box / intersect_scene / shade /
coord_out_of_scene / clamp_color on the reference's own box tree.  Bit for bit: RGB,
float colours, and every (pixel, leaf) hit record (leaf, primitive id, distance, colour)."""
import hashlib
import json
import os

import numpy as np
import pytest


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


with open(os.path.join(os.path.dirname(__file__), "golden", "frames.json")) as _f:
    FRAMES = json.load(_f)


def params_of(capi, fr):
    p = capi.default_params(fr["width"], fr["height"], fr["n_lights"], fr["seed"])
    p.stripe_rows, p.n_parts, p.part = fr["stripe_rows"], fr["n_parts"], fr["part"]
    p.max_depth = fr.get("max_depth", 0)
    return p


@pytest.mark.parametrize("name", sorted(FRAMES))
def test_frame_hashes(op, scenes, capi, name):
    fr = FRAMES[name]
    b = op.OracleBuilder(10)
    d = scenes.assemble(b, fr["config"])
    rgb, fcol, hits, st = op.render(d, params_of(capi, fr))
    assert st.slab_tests == fr["slab_tests"] and st.leaf_visits == fr["leaf_visits"]
    assert st.hits == fr["hits"] == len(hits)
    assert st.shadow_rays == fr["shadow_rays"] and st.shaded_samples == fr["shaded_samples"]
    assert st.nan_channels == 0
    assert sha(hits) == fr["sha256_hits"]
    assert sha(fcol) == fr["sha256_fcolor"]
    assert sha(rgb) == fr["sha256_rgb"]


@pytest.mark.parametrize("name", ["default_250", "spheres_200", "multi_384"])
def test_frame_data(op, scenes, capi, golden_dir, name):
    fr = FRAMES[name]
    g = np.load(os.path.join(golden_dir, "frame_%s.npz" % name))
    b = op.OracleBuilder(10)
    d = scenes.assemble(b, fr["config"])
    rgb, fcol, hits, _ = op.render(d, params_of(capi, fr))
    assert np.array_equal(rgb, g["rgb"])
    assert np.array_equal(fcol.view(np.uint32), g["fcolor_bits"])
    assert hits.tobytes() == g["hits"].tobytes()


def test_reference_counts_of_the_committed_scene():
    """Work counts of scene.h as committed at 1000 x 1000, 1 sample."""
    fr = FRAMES["default_1000"]
    assert fr["slab_tests"] == 14086054 and fr["hits"] == 573189 and fr["leaf_visits"] == 1474339


def test_threads_do_not_change_the_frame(op, scenes, capi):
    b = op.OracleBuilder(10)
    d = scenes.assemble(b, "default")
    p = capi.default_params(160, 160, 3, 5)
    a = op.render(d, p, threads=1)
    c = op.render(d, p, threads=4)
    assert np.array_equal(a[0], c[0]) and a[2].tobytes() == c[2].tobytes()


def test_row_tiles_assemble_to_the_full_frame(op, scenes, capi):
    b = op.OracleBuilder(10)
    d = scenes.assemble(b, "dino_spheres")
    p = capi.default_params(200, 150, 2, 9)
    full, _, fh, _ = op.render(d, p)
    out = np.zeros_like(full)
    nh = 0
    for part in range(4):
        q = capi.default_params(200, 150, 2, 9)
        q.stripe_rows, q.n_parts, q.part = 8, 4, part
        rgb, _, h, _ = op.render(d, q)
        rows = [j for j in range(150) if (j // 8) % 4 == part]
        assert rgb.shape[0] == len(rows) == capi.part_rows(q)
        out[rows] = rgb
        nh += len(h)
    assert np.array_equal(out, full) and nh == len(fh)


def test_nested_fold_is_not_a_flat_fold(op, capi):
    """raytracing.h:85-93: each nesting level folds its own colours.  Three unit-ish
    spheres behind one another in three leaf boxes, two of them inside an inner box:
    the result must be the per-level fold, which here differs from folding all leaves
    into one running colour."""
    from handmade import nested_fold_scene
    d, _keep = nested_fold_scene(capi)
    p = capi.default_params(64, 64, 4, 1)
    rgb, fcol, hits, _ = op.render(d, p)
    centre = hits[hits["pixel"] == 32 * 64 + 32]
    assert len(centre) == 3
    a, b, c = (centre["color"][k] for k in range(3))

    def less(x, y):
        return bool((x < y).all())
    zero = np.zeros(3, np.float32)
    inner = zero
    for col in (b, c):
        inner = col if less(inner, col) else inner
    nested = a if less(zero, a) else zero
    nested = inner if less(nested, inner) else nested
    flat = zero
    for col in (a, b, c):
        flat = col if less(flat, col) else flat
    assert not np.array_equal(nested, flat), "scene does not separate the two folds"
    assert np.array_equal(fcol[32, 32], nested)
