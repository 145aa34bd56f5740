"""The C-ABI library loads and exports every function include/rtx.h declares; argument
checks that need no GPU.  (No compute calls here: those are the -m gpu tests.)"""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtx\w*)\s*\(", text)) - {"rtx_status"})


def test_header_functions_are_exported(rtx):
    names = declared_functions(os.path.join(ROOT, "include", "rtx.h"))
    assert "rtx_render" in names and "rtx_scene_create" in names and len(names) >= 10
    L = rtx.lib()
    for n in names:
        assert hasattr(L, n), "librtx_amd.so does not export " + n


def test_struct_sizes_match_the_header(capi):
    assert C.sizeof(capi.Material) == 32 and C.sizeof(capi.Light) == 24
    assert C.sizeof(capi.Prim) == 48 and C.sizeof(capi.Node) == 32 and C.sizeof(capi.Leaf) == 8
    assert C.sizeof(capi.Hit) == 32 and C.sizeof(capi.Params) == 52 and C.sizeof(capi.Stats) == 80


def test_defaults_are_the_reference_settings(rtx, capi):
    p = capi.Params()
    rtx.lib().rtx_params_default(C.byref(p))
    q = capi.default_params()
    assert bytes(p) == bytes(q)
    assert (p.width, p.height, p.n_lights, p.max_depth) == (1000, 1000, 1, 0)  # raytracing.h:12-19
    assert abs(p.acne - 0.01) < 1e-9 and list(p.camera) == [500.0, 500.0, 3000.0]
    assert rtx.lib().rtx_abi_version() == 2


def test_part_rows(rtx, capi):
    L = rtx.lib()
    for h, stripe, parts in [(1080, 8, 8), (1000, 16, 3), (7, 8, 2), (2160, 8, 8), (100, 1, 7)]:
        total = 0
        for part in range(parts):
            p = capi.default_params(64, h)
            p.stripe_rows, p.n_parts, p.part = stripe, parts, part
            r = L.rtx_part_rows(C.byref(p))
            assert r == capi.part_rows(p)
            total += r
        assert total == h


def test_invalid_scenes_are_rejected_without_a_gpu(rtx, capi):
    """Validation happens before any device is touched."""
    L = rtx.lib()
    h = C.c_void_p()
    assert L.rtx_scene_create(None, 0, C.byref(h)) == 1
    d = capi.SceneDesc()
    nodes = (capi.Node * 1)()
    nodes[0].skip, nodes[0].leaf = 0, 0      # skip must exceed the own index
    leaves = (capi.Leaf * 1)()
    d.n_nodes, d.n_leaves, d.nodes, d.leaves = 1, 1, nodes, leaves
    assert L.rtx_scene_create(C.byref(d), 0, C.byref(h)) == 1
    assert b"skip" in L.rtx_last_error()
    nodes[0].skip = 1
    leaves[0].first, leaves[0].count = 0, 5  # range beyond the (empty) primitive array
    assert L.rtx_scene_create(C.byref(d), 0, C.byref(h)) == 1


def test_no_gpu_is_an_error_not_a_fallback(rtx, capi, op, scenes):
    """Without a usable device the product must fail loudly."""
    if rtx.device_count() > 0:
        pytest.skip("a GPU is visible")
    b = op.OracleBuilder(10)
    d = scenes.assemble(b, "spheres")
    with pytest.raises(rtx.RtxError) as e:
        rtx.Scene(d)
    assert e.value.status == 2


def test_host_library_exports_the_reference_entry_points():
    """librtx_host.so must export the C++ functions of include/rtx_dropin.hpp with the
    reference's signatures (scene.h:4, raytracing.h:100, meshs.h:1, :60, bvh.h:5)."""
    import subprocess
    lib = os.path.join(ROOT, "ray-tracing_amd", "host", "librtx_host.so")
    out = subprocess.check_output(["nm", "-D", "--demangle", "--defined-only", lib], text=True)
    for sig in ["init_scene(Light&, std::vector<Box", "render_image(Light&, std::vector<Box",
                "parse(std::", "create_mesh(Mesh&, std::vector<Box", "create_boxs(std::vector<Box",
                "rtx_host::flatten(", "rtx_host::render_to_memory(", "rtx_host::write_png("]:
        assert sig in out, sig
