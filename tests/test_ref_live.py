"""The CPU oracle against the reference harness run live (oracle/_ref/libref.so, built
from the reference's headers by oracle/Makefile).  Complements the committed fixtures
with other seeds, sizes and tilings.  Skipped where the harness has not been built."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def have_ref(op):
    if not op.have_ref():
        pytest.skip("oracle/_ref/libref.so not built")
    return True


CASES = [("default", 180, 180, 5, 11, None), ("spheres", 96, 96, 7, 2, None),
         ("multi", 256, 144, 2, 3, None), ("dino_spheres", 200, 120, 9, 4, (4, 3, 2)),
         ("buddha", 240, 135, 3, 5, None), ("bunny", 1000, 1000, 1, 6, (16, 8, 5))]


@pytest.mark.parametrize("cfg,w,h,nl,seed,tile", CASES)
def test_oracle_equals_reference_functions_composed(op, scenes, capi, have_ref, cfg, w, h, nl, seed, tile):
    ob = op.OracleBuilder(10)
    d = scenes.assemble(ob, cfg)
    rb = op.RefBuilder()
    scenes.assemble(rb, cfg)
    p = capi.default_params(w, h, nl, seed)
    if tile:
        p.stripe_rows, p.n_parts, p.part = tile
    rgb1, f1, h1, st = op.render(d, p)
    rgb2, f2, h2, counts = rb.render(p)
    assert np.array_equal(rgb1, rgb2)
    assert np.array_equal(f1.view(np.uint32), f2.view(np.uint32))
    assert h1.tobytes() == h2.tobytes()
    assert [st.slab_tests, st.leaf_visits, st.hits, st.shadow_rays, st.shaded_samples] == counts


def test_scene_trees_are_identical(op, scenes, have_ref):
    for cfg in scenes.CONFIGS:
        ob = op.OracleBuilder(10)   # the builders own the arrays: keep them alive
        a = op.desc_arrays(scenes.assemble(ob, cfg))
        rb = op.RefBuilder()
        b = op.desc_arrays(scenes.assemble(rb, cfg))
        for k in a:
            assert np.array_equal(a[k], b[k]), (cfg, k)
