"""ctypes access to the C++ host layer (librtx_host.so): the drop-in parse / create_mesh /
create_boxs / init_scene / render_image of include/rtx_dropin.hpp, driven from Python."""
import ctypes as C
import os

import numpy as np

from . import capi

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(capi.HOST_LIB_PATH):
            raise RuntimeError("%s is not built (run __graft_entry__.build())" % capi.HOST_LIB_PATH)
        L = C.CDLL(capi.HOST_LIB_PATH)
        L.rtxh_builder_new.restype = C.c_void_p
        L.rtxh_builder_new.argtypes = [C.c_uint32]
        L.rtxh_builder_free.argtypes = [C.c_void_p]
        L.rtxh_error.restype = C.c_char_p
        L.rtxh_error.argtypes = [C.c_void_p]
        L.rtxh_set_light.argtypes = [C.c_void_p, C.POINTER(capi.Light)]
        L.rtxh_add_material.restype = C.c_uint32
        L.rtxh_add_material.argtypes = [C.c_void_p, C.POINTER(capi.Material)]
        L.rtxh_add_leaf_box.argtypes = [C.c_void_p, C.c_float * 3, C.c_float * 3,
                                        C.POINTER(capi.Prim), C.c_uint32]
        L.rtxh_add_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                    C.c_float * 3, C.c_float, C.c_uint32]
        L.rtxh_parse_off.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint32),
                                     C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint32)]
        L.rtxh_free.argtypes = [C.c_void_p]
        L.rtxh_init_scene.argtypes = [C.c_void_p, C.c_char_p]
        L.rtxh_build.argtypes = [C.c_void_p, C.POINTER(capi.SceneDesc)]
        L.rtxh_render_image.argtypes = [C.c_void_p, C.POINTER(capi.Params), C.POINTER(C.c_int), C.c_uint32,
                                        C.c_char_p]
        L.rtxh_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.rtxh_replica_stats.argtypes = [C.c_uint64 * 2, C.c_int]
        _lib = L
    return _lib


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class HostBuilder:
    """Assembles a scene with the C++ host layer and flattens it for rtx_scene_create."""

    def __init__(self, box_depth=10):
        self.L = lib()
        self.h = self.L.rtxh_builder_new(box_depth)
        self.desc = None

    def set_light(self, light):
        self.L.rtxh_set_light(self.h, C.byref(light))

    def add_material(self, m):
        return self.L.rtxh_add_material(self.h, C.byref(m))

    def add_leaf_box(self, lb, rt, prims):
        if self.L.rtxh_add_leaf_box(self.h, _f3(lb), _f3(rt), prims, len(prims)):
            raise RuntimeError("add_leaf_box: bad material index")

    def add_mesh(self, verts, faces, position, taille, material):
        v = np.ascontiguousarray(verts, dtype=np.float32)
        f = np.ascontiguousarray(faces, dtype=np.uint32)
        rc = self.L.rtxh_add_mesh(self.h, v.ctypes.data, len(v), f.ctypes.data, len(f),
                                  _f3(position), float(taille), material)
        if rc:
            raise RuntimeError("add_mesh failed: %d" % rc)

    def init_scene(self, directory):
        """The scene the reference's init_scene assembles; reads <directory>/meshs/*.off."""
        rc = self.L.rtxh_init_scene(self.h, directory.encode())
        if rc:
            raise RuntimeError("init_scene failed: %d" % rc)

    def build(self):
        d = capi.SceneDesc()
        if self.L.rtxh_build(self.h, C.byref(d)):
            raise RuntimeError("flatten: " + self.L.rtxh_error(self.h).decode())
        self.desc = d
        return d

    def render_image(self, p, path, devices=(0,)):
        """render_image() (raytracing.h:100): renders on the GPU(s) and writes the file.
        More than one entry in `devices` splits the frame into row stripes, one host
        thread and scene replica per entry (a device id may repeat)."""
        devs = (C.c_int * len(devices))(*devices)
        if self.L.rtxh_render_image(self.h, C.byref(p), devs, len(devices), path.encode()):
            raise RuntimeError("render_image failed")

    def close(self):
        if self.h:
            self.L.rtxh_builder_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def replica_stats(release=False):
    """(created, reused) scene replicas of render_image since the process started;
    release=True also frees the replicas it keeps between calls."""
    out = (C.c_uint64 * 2)()
    lib().rtxh_replica_stats(out, 1 if release else 0)
    return int(out[0]), int(out[1])


def parse_off(path):
    L = lib()
    pv, pf = C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)()
    nv, nf = C.c_uint32(), C.c_uint32()
    rc = L.rtxh_parse_off(path.encode(), C.byref(pv), C.byref(nv), C.byref(pf), C.byref(nf))
    if rc:
        return rc, None, None
    v = np.ctypeslib.as_array(pv, shape=(max(nv.value, 1) * 3,))[:nv.value * 3].reshape(-1, 3).copy()
    f = np.ctypeslib.as_array(pf, shape=(max(nf.value, 1) * 3,))[:nf.value * 3].reshape(-1, 3).copy()
    L.rtxh_free(pv)
    L.rtxh_free(pf)
    return 0, v, f


def write_png(path, rgb):
    a = np.ascontiguousarray(rgb, np.uint8)
    if lib().rtxh_write_png(path.encode(), a.ctypes.data, a.shape[1], a.shape[0]):
        raise RuntimeError("cannot write " + path)
