"""Scene assembly of the CPU oracle (meshs.h, bvh.h, scene.h restated) against the
structure fixtures produced by the reference's own create_mesh / create_boxs."""
import hashlib
import json
import os

import numpy as np
import pytest


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def structure(golden_dir):
    with open(os.path.join(golden_dir, "structure.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("cfg", ["default", "spheres", "bunny", "dino_spheres", "buddha", "multi"])
def test_flattened_tree_matches_reference(op, scenes, structure, cfg):
    b = op.OracleBuilder(10)
    d = scenes.assemble(b, cfg)
    want = structure[cfg]
    assert (d.n_nodes, d.n_leaves, d.n_prims, d.n_materials) == (
        want["n_nodes"], want["n_leaves"], want["n_prims"], want["n_materials"])
    for k, v in op.desc_arrays(d).items():
        assert sha(v) == want["sha256_" + k], k


def test_default_config_is_init_scene(structure):
    """CONFIGS['default'] reproduces scene.h:4-108 as the reference assembles it."""
    a, b = structure["default"], structure["init_scene"]
    assert a == b


def test_tree_shape(op, scenes):
    """bvh.h:5-61: every mesh becomes a complete tree of 2047 boxes / 1024 leaves; the
    sphere box stays a single top-level leaf (scene.h:52)."""
    b = op.OracleBuilder(10)
    d = scenes.assemble(b, "default")
    assert d.n_nodes == 1 + 2 * 2047 and d.n_leaves == 1 + 2 * 1024
    nodes = np.frombuffer(op.desc_arrays(d)["nodes"], dtype=np.uint32).reshape(-1, 8)
    assert nodes[0, 7] == 0 and nodes[0, 6] == 1          # sphere box: leaf 0, skip 1
    assert nodes[1, 7] == 0xFFFFFFFF and nodes[1, 6] == 2048  # bunny root spans 2047 boxes
    # z is never split (bvh.h:16-53): all boxes of a mesh share the root's z range
    f = nodes.view(np.float32)
    assert np.all(f[1:2048, 2] == f[1, 2]) and np.all(f[1:2048, 5] == f[1, 5])


def test_other_box_depths(op, scenes):
    for depth in (0, 1, 4):
        b = op.OracleBuilder(depth)
        d = scenes.assemble(b, "dino_spheres")
        assert d.n_leaves == 1 + 2 ** depth
        assert d.n_nodes == 1 + 2 ** (depth + 1) - 1


def _write_off(path, verts, faces, header="OFF"):
    with open(path, "w") as f:
        f.write("%s\n%d %d 0\n" % (header, len(verts), len(faces)))
        for v in verts:
            f.write("%.9g %.9g %.9g\n" % tuple(float(x) for x in v))
        for t in faces:
            f.write("3 %d %d %d \n" % tuple(int(x) for x in t))


def test_off_reader_round_trip(op, scenes, tmp_path):
    """The .mesh fixtures hold what the reference's parse() returns for its OFF files
    (tools/make_meshes.py); an OFF text written from them must read back identically."""
    verts, faces = scenes.load_mesh("dino")
    path = str(tmp_path / "dino.off")
    _write_off(path, verts, faces)
    rc, v, f = op.parse_off(path)
    assert rc == 0
    assert np.array_equal(v.view(np.uint32), verts.view(np.uint32)) and np.array_equal(f, faces)


def test_off_reader_errors(op, tmp_path):
    bad = str(tmp_path / "bad.off")
    _write_off(bad, np.zeros((1, 3)), np.zeros((0, 3)), header="COFF")
    assert op.parse_off(bad)[0] == 1                      # meshs.h:79-83
    assert op.parse_off(str(tmp_path / "missing.off"))[0] == 2


@pytest.mark.skipif(not os.path.isdir("/root/reference/meshs"), reason="reference tree not present")
@pytest.mark.parametrize("name", ["bunny", "dino", "buddha"])
def test_off_reader_on_reference_files(op, scenes, name):
    rc, v, f = op.parse_off("/root/reference/meshs/%s.off" % name)
    verts, faces = scenes.load_mesh(name)
    assert rc == 0
    assert np.array_equal(v.view(np.uint32), verts.view(np.uint32)) and np.array_equal(f, faces)
