"""Scene definitions: the values of the reference's scene.h as plain data, and the
benchmark configurations built from them.

A *builder* is any object with add_material / add_leaf_box / add_mesh / set_light /
build (the host library's, or a test-side one); `assemble` drives it.
"""
import os
import struct

import numpy as np

from . import capi

# scene.h:8-33
MATERIALS = {
    "red": dict(ior=1.0771, roughness=0.0589, specular=(1.0, 0.852, 1.172), diffuse=(0.26, 0.036, 0.014)),
    "green": dict(ior=1.1481, roughness=0.0625, specular=(1.0, 1.056, 1.146), diffuse=(0.016, 0.073, 0.04)),
    "blue": dict(ior=1.1153, roughness=0.068, specular=(1.0, 0.965, 1.07), diffuse=(0.012, 0.036, 0.106)),
    "white": dict(ior=1.1022, roughness=0.0579, specular=(1.0, 0.766, 0.762), diffuse=(0.2, 0.2, 0.2)),
    "nickel": dict(ior=30.0, roughness=0.01, specular=(1.0, 0.882, 0.786), diffuse=(0.014, 0.012, 0.012)),
}

# scene.h:37-47
SPHERE_NICKEL = dict(centre=(250.0, 200.0, 500.0), radius=175.0, material="nickel")
SPHERE_WHITE = dict(centre=(250.0, 750.0, 500.0), radius=175.0, material="white")

# scene.h:58-76 (constructed by the reference but never pushed into a box)
_R = 130000.0
WALLS = [
    dict(centre=(500.0, 500.0, _R + 1001.0), radius=_R, material="white"),   # back
    dict(centre=(500.0, 500.0, -_R - 1.0), radius=_R, material="white"),     # front
    dict(centre=(500.0, _R + 1001.0, 500.0), radius=_R, material="white"),   # up
    dict(centre=(500.0, -_R - 1.0, 500.0), radius=_R, material="white"),     # down
    dict(centre=(_R + 1001.0, 500.0, 500.0), radius=_R, material="red"),     # right
    dict(centre=(-_R - 1.0, 500.0, 500.0), radius=_R, material="blue"),      # left
]

LIGHT = dict(position=(250.0, 250.0, 800.0), color=(255.0, 255.0, 255.0))  # scene.h:80-81


def _sphere_box(s):
    """Bounding box of one sphere as scene.h:49-52 computes it (binary32 adds)."""
    c = np.float32(s["centre"])
    r = np.float32(s["radius"])
    return (c - r).astype(np.float32), (c + r).astype(np.float32)


# name -> description.  "boxes" are top-level boxes in push order.
CONFIGS = {
    # scene.h as committed: white sphere + red bunny + blue dino
    "default": dict(
        width=1000, height=1000, n_lights=1,
        boxes=[("spheres", [SPHERE_WHITE], None),
               ("mesh", "bunny", dict(position=(725.0, 725.0, 500.0), taille=500.0, material="red")),
               ("mesh", "dino", dict(position=(575.0, 250.0, 500.0), taille=500.0, material="blue"))]),
    # spheres only, one unsplit box spanning everything (no hierarchy)
    "spheres": dict(
        width=1000, height=1000, n_lights=4,
        boxes=[("spheres", [SPHERE_WHITE, SPHERE_NICKEL] + WALLS,
                ((-1e6, -1e6, -1e6), (1e6, 1e6, 1e6)))]),
    "bunny": dict(
        width=1920, height=1080, n_lights=100,
        boxes=[("mesh", "bunny", dict(position=(500.0, 500.0, 500.0), taille=500.0, material="red"))]),
    "dino_spheres": dict(
        width=1920, height=1080, n_lights=500,
        boxes=[("spheres", [SPHERE_WHITE], None),
               ("mesh", "dino", dict(position=(575.0, 250.0, 500.0), taille=500.0, material="blue"))]),
    "buddha": dict(
        width=3840, height=2160, n_lights=500,
        boxes=[("mesh", "buddha", dict(position=(500.0, 500.0, 500.0), taille=500.0, material="white"))]),
    "multi": dict(
        width=3840, height=2160, n_lights=1000,
        boxes=[("spheres", [SPHERE_WHITE], None),
               ("mesh", "bunny", dict(position=(725.0, 725.0, 500.0), taille=500.0, material="red")),
               ("mesh", "dino", dict(position=(575.0, 250.0, 500.0), taille=500.0, material="blue")),
               ("mesh", "buddha", dict(position=(250.0, 300.0, 500.0), taille=500.0, material="green"))]),
}

MESH_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def load_mesh(name, mesh_dir=None):
    """Read <name>.mesh: 'RTXM', u32 nv, u32 nf, f32 verts[nv][3], u32 faces[nf][3].

    The files hold exactly the binary32 vertices and the indices that the reference's
    parse() (meshs.h:60-125) yields for meshs/<name>.off.
    """
    path = os.path.join(mesh_dir or MESH_DIR, name + ".mesh")
    with open(path, "rb") as f:
        magic, nv, nf = struct.unpack("<4sII", f.read(12))
        if magic != b"RTXM":
            raise ValueError("%s: not a mesh file" % path)
        verts = np.frombuffer(f.read(nv * 12), dtype="<f4").reshape(nv, 3).copy()
        faces = np.frombuffer(f.read(nf * 12), dtype="<u4").reshape(nf, 3).copy()
    return verts, faces


def save_mesh(path, verts, faces):
    verts = np.ascontiguousarray(verts, dtype="<f4")
    faces = np.ascontiguousarray(faces, dtype="<u4")
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"RTXM", len(verts), len(faces)))
        f.write(verts.tobytes())
        f.write(faces.tobytes())


def make_material(name):
    m = MATERIALS[name]
    out = capi.Material()
    out.ior, out.roughness = m["ior"], m["roughness"]
    out.specular[:] = m["specular"]
    out.diffuse[:] = m["diffuse"]
    return out


def assemble(builder, config, mesh_dir=None):
    """Drive `builder` through the boxes of CONFIGS[config] (or a config dict)."""
    cfg = CONFIGS[config] if isinstance(config, str) else config
    mat_index = {}

    def mat(name):
        if name not in mat_index:
            mat_index[name] = builder.add_material(make_material(name))
        return mat_index[name]

    light = capi.Light()
    light.position[:] = cfg.get("light", LIGHT)["position"]
    light.color[:] = cfg.get("light", LIGHT)["color"]
    builder.set_light(light)
    for kind, what, where in cfg["boxes"]:
        if kind == "spheres":
            prims = (capi.Prim * len(what))()
            for k, s in enumerate(what):
                prims[k].type = capi.RTX_PRIM_SPHERE
                prims[k].g[0:3] = s["centre"]
                prims[k].g[3] = s["radius"]
                prims[k].material = mat(s["material"])
            if where is None:
                lb, rt = _sphere_box(what[0])
            else:
                lb, rt = np.float32(where[0]), np.float32(where[1])
            builder.add_leaf_box(lb, rt, prims)
        else:
            verts, faces = load_mesh(what, mesh_dir)
            builder.add_mesh(verts, faces, where["position"], where["taille"], mat(where["material"]))
    return builder.build()
