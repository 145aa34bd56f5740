"""Python binding (ctypes) of the MI355X render library, librtx_amd.so.

The product is the C-ABI library (include/rtx.h) and the C++ host layer; this module
only exposes them to the tests and to bench.py.  There is no CPU fallback: if the
HIP library is missing or no gfx950 device is usable, calls raise RtxError.
"""
import ctypes as C
import os

import numpy as np

from . import capi
from .capi import Params, SceneDesc, Stats, Hit, default_params, part_rows  # noqa: F401

HIT_DTYPE = np.dtype([("pixel", "<u4"), ("ordinal", "<u4"), ("leaf", "<u4"), ("prim", "<u4"),
                      ("t", "<f4"), ("color", "<f4", 3)])


class RtxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (capi.STATUS_NAMES.get(status, status), message))
        self.status = status


_lib = None


def kernel_source_hash():
    """SHA-256 over the sources librtx_amd.so is built from (csrc/*.h, *.hip, Makefile, in
    name order).  Measurements kept in files carry it, so that a reader can tell whether they
    were taken on these kernels."""
    import hashlib
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")) or name == "Makefile":
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def lib():
    """Load librtx_amd.so (built in-tree by `make -C ray-tracing_amd/csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(capi.LIB_PATH):
            raise RtxError(2, "%s is not built (run __graft_entry__.build())" % capi.LIB_PATH)
        L = C.CDLL(capi.LIB_PATH)
        L.rtx_abi_version.restype = C.c_uint32
        L.rtx_device_count.restype = C.c_int
        L.rtx_last_error.restype = C.c_char_p
        L.rtx_params_default.argtypes = [C.POINTER(Params)]
        L.rtx_part_rows.restype = C.c_uint32
        L.rtx_part_rows.argtypes = [C.POINTER(Params)]
        L.rtx_selftest.argtypes = [C.c_int, C.c_uint64 * 7]
        L.rtx_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]
        L.rtx_scene_destroy.argtypes = [C.c_void_p]
        L.rtx_render.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.POINTER(Stats)]
        L.rtx_render_device.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p,
                                        C.POINTER(Stats)]
        L.rtx_render_device_static.argtypes = L.rtx_render_device.argtypes
        L.rtx_host_alloc.argtypes = [C.c_uint64, C.POINTER(C.c_void_p)]
        L.rtx_host_free.argtypes = [C.c_void_p]
        L.rtx_render_hits.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint64,
                                      C.POINTER(C.c_uint64), C.c_void_p]
        _lib = L
    return _lib


def _check(status):
    if status != 0:
        raise RtxError(status, lib().rtx_last_error().decode(errors="replace"))


def device_count():
    return lib().rtx_device_count()


def selftest(device=0):
    """rtx_selftest: dict of the checked / mismatch counts (mismatches must be 0)."""
    out = (C.c_uint64 * 7)()
    _check(lib().rtx_selftest(device, out))
    return dict(rcp_checked=int(out[0]), rcp_mismatches=int(out[1]), chain_checked=int(out[2]),
                chain_mismatches=int(out[3]), chain_declined=int(out[4]), div_checked=int(out[5]),
                div_mismatches=int(out[6]))


class HostFrame:
    """Page-locked host memory for a frame (rtx_host_alloc), viewed as a numpy array."""

    def __init__(self, rows, width):
        self._p = C.c_void_p()
        self.nbytes = rows * width * 3
        _check(lib().rtx_host_alloc(self.nbytes, C.byref(self._p)))
        buf = (C.c_uint8 * max(self.nbytes, 1)).from_address(self._p.value)
        self.array = np.frombuffer(buf, np.uint8, self.nbytes).reshape(rows, width, 3)

    def close(self):
        if self._p:
            self.array = None
            lib().rtx_host_free(self._p)
            self._p = C.c_void_p()


class Scene:
    """A scene uploaded to one GPU (rtx_scene_create)."""

    def __init__(self, desc, device=0):
        self._h = C.c_void_p()
        _check(lib().rtx_scene_create(C.byref(desc), device, C.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            lib().rtx_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, p, timing=False):
        """Full frame (or the rows of one part) as uint8 [rows, width, 3], plus Stats."""
        rgb = np.empty((part_rows(p), p.width, 3), np.uint8)
        st = Stats()
        if timing:
            p.flags |= capi.RTX_FLAG_TIMING
        _check(lib().rtx_render(self._h, C.byref(p), rgb.ctypes.data, C.byref(st)))
        return rgb, st

    def render_into(self, p, frame):
        """rtx_render into a HostFrame (page-locked: the read-back runs at the link's rate)."""
        st = Stats()
        _check(lib().rtx_render(self._h, C.byref(p), frame.array.ctypes.data, C.byref(st)))
        return st

    def render_device(self, p, d_rgb, stream=0, stats=True):
        """Render into device memory at address d_rgb on HIP stream `stream`."""
        st = Stats() if stats else None
        _check(lib().rtx_render_device(self._h, C.byref(p), C.c_void_p(d_rgb), C.c_void_p(stream),
                                       C.byref(st) if stats else None))
        return st

    def render_device_static(self, p, d_rgb, stream=0):
        """rtx_render_device_static: the same frame with every work buffer taken from the
        library's static device array (the call allocates nothing)."""
        st = Stats()
        _check(lib().rtx_render_device_static(self._h, C.byref(p), C.c_void_p(d_rgb), C.c_void_p(stream),
                                              C.byref(st)))
        return st

    def render_hits(self, p):
        """(rgb, hits) where hits is a structured array sorted by (pixel, ordinal)."""
        rgb = np.empty((part_rows(p), p.width, 3), np.uint8)
        n = C.c_uint64(0)
        _check(lib().rtx_render_hits(self._h, C.byref(p), None, 0, C.byref(n), None))
        hits = np.zeros(max(int(n.value), 1), HIT_DTYPE)
        _check(lib().rtx_render_hits(self._h, C.byref(p), hits.ctypes.data, len(hits), C.byref(n),
                                     rgb.ctypes.data))
        return rgb, hits[:int(n.value)]
