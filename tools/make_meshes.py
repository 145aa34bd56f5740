#!/usr/bin/env python3
"""Convert the reference's OFF meshes into the binary .mesh files the tests and the
benchmark load (ray-tracing_amd/data/).  The vertices and indices are the ones the
REFERENCE's own parse() (meshs.h:60-125, through oracle/_ref/libref.so) yields, so the
files double as golden vectors for the OFF readers.  Run in the build container only
(needs /root/reference and `make -C oracle`).
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py  # noqa: E402

scenes = importlib.import_module("ray-tracing_amd.scenes")
REF = os.environ.get("RTX_REFERENCE", "/root/reference")

for name in ("bunny", "dino", "buddha"):
    rc, v, f = oracle_py.parse_off(os.path.join(REF, "meshs", name + ".off"), which="ref")
    assert rc == 0, name
    out = os.path.join(scenes.MESH_DIR, name + ".mesh")
    scenes.save_mesh(out, v, f)
    print(name, v.shape, f.shape, "->", out, os.path.getsize(out), "bytes")
