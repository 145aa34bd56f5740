"""ctypes mirror of include/rtx.h (plain data only) and loader of the HIP library.

The structures here are shared by the product wrapper and by the test-side
loaders of the CPU oracle, which speaks the same plain-data scene format.
"""
import ctypes as C
import os

RTX_NONE = 0xFFFFFFFF
RTX_PRIM_SPHERE = 1
RTX_PRIM_TRIANGLE = 2
RTX_FLAG_TIMING = 1
RTX_FLAG_WAVE_PER_HIT = 2
RTX_FLAG_IEEE_SHADE = 4
RTX_FLAG_NO_SORT = 8
RTX_FLAG_NO_SCREEN = 16

STATUS_NAMES = {0: "RTX_OK", 1: "RTX_ERR_INVALID", 2: "RTX_ERR_NO_DEVICE", 3: "RTX_ERR_HIP",
                4: "RTX_ERR_NOMEM", 5: "RTX_ERR_LIMIT"}


class Material(C.Structure):
    _fields_ = [("ior", C.c_float), ("roughness", C.c_float),
                ("specular", C.c_float * 3), ("diffuse", C.c_float * 3)]


class Light(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("color", C.c_float * 3)]


class Prim(C.Structure):
    _fields_ = [("type", C.c_uint32), ("g", C.c_float * 9),
                ("material", C.c_uint32), ("id", C.c_uint32)]


class Node(C.Structure):
    _fields_ = [("lb", C.c_float * 3), ("rt", C.c_float * 3),
                ("skip", C.c_uint32), ("leaf", C.c_uint32)]


class Leaf(C.Structure):
    _fields_ = [("first", C.c_uint32), ("count", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_leaves", C.c_uint32),
                ("n_prims", C.c_uint32), ("n_materials", C.c_uint32),
                ("nodes", C.POINTER(Node)), ("leaves", C.POINTER(Leaf)),
                ("prims", C.POINTER(Prim)), ("materials", C.POINTER(Material)),
                ("light", Light)]


class Params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32),
                ("n_lights", C.c_uint32), ("max_depth", C.c_uint32),
                ("seed", C.c_uint32), ("acne", C.c_float), ("camera", C.c_float * 3),
                ("stripe_rows", C.c_uint32), ("n_parts", C.c_uint32),
                ("part", C.c_uint32), ("flags", C.c_uint32)]


class Hit(C.Structure):
    _fields_ = [("pixel", C.c_uint32), ("ordinal", C.c_uint32), ("leaf", C.c_uint32),
                ("prim", C.c_uint32), ("t", C.c_float), ("color", C.c_float * 3)]


class Stats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("hits", C.c_uint64),
                ("shadow_rays", C.c_uint64), ("shadow_tests", C.c_uint64),
                ("lit_samples", C.c_uint64), ("shadow_tests_run", C.c_uint64),
                ("kernel_ms", C.c_float * 4),
                ("render_attempts", C.c_uint32), ("record_capacity", C.c_uint32),
                ("screen_ms", C.c_float), ("reserved", C.c_uint32)]


assert C.sizeof(Material) == 32 and C.sizeof(Prim) == 48 and C.sizeof(Node) == 32
assert C.sizeof(Hit) == 32 and C.sizeof(Stats) == 80


def default_params(width=1000, height=1000, n_lights=1, seed=1):
    """The reference's compile-time settings (raytracing.h:12-21, :112)."""
    p = Params()
    p.width, p.height, p.n_lights, p.max_depth, p.seed = width, height, n_lights, 0, seed
    p.acne = 0.01
    p.camera[:] = (500.0, 500.0, 3000.0)
    p.stripe_rows, p.n_parts, p.part, p.flags = 0, 0, 0, 0
    return p


def part_rows(p):
    if p.n_parts <= 1 or p.stripe_rows == 0:
        return p.height
    return sum(1 for j in range(p.height) if (j // p.stripe_rows) % p.n_parts == p.part)


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTX_AMD_LIB") or os.path.join(_HERE, "csrc", "librtx_amd.so")
HOST_LIB_PATH = os.path.join(_HERE, "host", "librtx_host.so")
