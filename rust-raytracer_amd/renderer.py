"""Host-side mirror of the reference's seam: `Renderer::render(camera, Arc<Scene>) -> Surface`
(src/raytracer/renderer.rs:24-65) on top of the C ABI.  `Scene` owns an rtx_scene handle (the
flattened scene resident in HBM); `Renderer` carries the RenderOptions and calls rtx_render.
Everything here goes through librtx.so; if it is missing, `abi.load()` raises.
"""
import ctypes as C

import numpy as np

from . import abi
from .desc import options as make_options


class Scene:
    """Arc<Scene> counterpart: a scene flattened, octree-indexed and uploaded to one GPU."""

    def __init__(self, flat, device=0):
        self._lib = abi.load()
        self.flat = flat
        h = C.c_void_p()
        abi.check(self._lib.rtx_scene_create(C.byref(flat.desc), device, C.byref(h)), "rtx_scene_create")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self._lib.rtx_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        inf = abi.SceneInfo()
        abi.check(self._lib.rtx_scene_get_info(self.h, C.byref(inf)), "rtx_scene_get_info")
        return inf

    def last_kernel_names(self):
        """rtx_last_kernel_names as a dict: {"closest": "k_closest<5>", "shade": ..., "shadow0": ..., ...}."""
        buf = C.create_string_buffer(1024)
        abi.check(self._lib.rtx_last_kernel_names(self.h, buf, 1024), "rtx_last_kernel_names")
        return dict(kv.split("=", 1) for kv in buf.value.decode().split(";") if "=" in kv)

    def hierarchy_dump(self):
        """rtx_hierarchy_dump: the SAH hierarchy as it lies on the device: (wide, bvh4) structured numpy arrays
        (wide[n, 16] of DWideChild, bvh4[n] of DNode4; empty arrays where a cut was not built)."""
        wide_t = np.dtype([("lo", "<f4", 3), ("hi", "<f4", 3), ("ref", "<u4"), ("first_node", "<u4")])
        bvh4_t = np.dtype([("lo", "<f4", (3, 4)), ("hi", "<f4", (3, 4)), ("ref", "<u4", 4)])
        out = []
        for which, dt in ((0, wide_t), (1, bvh4_t)):
            nb = C.c_int64()
            abi.check(self._lib.rtx_hierarchy_dump(self.h, which, None, 0, C.byref(nb)), "rtx_hierarchy_dump")
            arr = np.zeros(nb.value // dt.itemsize, dtype=dt)
            if nb.value:
                abi.check(self._lib.rtx_hierarchy_dump(self.h, which, arr.ctypes.data_as(C.c_void_p), nb.value, C.byref(nb)), "rtx_hierarchy_dump")
            out.append(arr)
        return out[0].reshape(-1, 16), out[1]

    def render(self, cam, opt, want_ids=True, out=None):
        """rtx_render into host arrays: returns (rgba[H,W,4] u8, prim_id[H,W] i32 or None, Stats)."""
        H, W = cam.image_height, cam.image_width
        rgba = out if out is not None else np.zeros((H, W, 4), dtype=np.uint8)
        ids = np.full((H, W), -2, dtype=np.int32) if want_ids else None
        st = abi.Stats()
        abi.check(self._lib.rtx_render(self.h, C.byref(cam), C.byref(opt), rgba.ctypes.data_as(C.c_void_p),
                                       ids.ctypes.data_as(C.c_void_p) if want_ids else None, C.byref(st)),
                  "rtx_render")
        return rgba, ids, st

    def render_device(self, cam, opt, rgba_ptr, ids_ptr=None, stream=None, want_stats=True):
        """rtx_render_device: outputs are device pointers (ints), e.g. torch tensors' data_ptr()."""
        st = abi.Stats() if want_stats else None
        abi.check(self._lib.rtx_render_device(self.h, C.byref(cam), C.byref(opt), C.c_void_p(rgba_ptr),
                                              C.c_void_p(ids_ptr) if ids_ptr else None,
                                              C.c_void_p(stream) if stream else None,
                                              C.byref(st) if want_stats else None), "rtx_render_device")
        return st

    def render_device_noalloc(self, cam, opt, rgba_ptr, ids_ptr=None, stream=None, want_stats=True):
        """rtx_render_device_noalloc: the same frame out of the library's static device scratch (allocates nothing)."""
        st = abi.Stats() if want_stats else None
        abi.check(self._lib.rtx_render_device_noalloc(self.h, C.byref(cam), C.byref(opt), C.c_void_p(rgba_ptr),
                                                      C.c_void_p(ids_ptr) if ids_ptr else None,
                                                      C.c_void_p(stream) if stream else None,
                                                      C.byref(st) if want_stats else None), "rtx_render_device_noalloc")
        return st

    def nearest_hits(self, origin, direction, flags=0):
        """rtx_nearest_hits_ex: flags 0 = the renderer's closest-hit walk, FLAG_REFERENCE_ORDER = the all-f64 walk."""
        o = np.ascontiguousarray(origin, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(direction, dtype=np.float64).reshape(-1, 3)
        n = o.shape[0]
        pid = np.zeros(n, dtype=np.int32)
        t = np.zeros(n, dtype=np.float64)
        abi.check(self._lib.rtx_nearest_hits_ex(self.h, n, o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                                                pid.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), flags),
                  "rtx_nearest_hits_ex")
        return pid, t

    def shade_probe(self, which, rows):
        """rtx_shade_probe: `rows` is a list of argument rows (<= 16 numbers each); returns out[n, 4]."""
        a = np.zeros((len(rows), 16), dtype=np.float64)
        for i, r in enumerate(rows):
            a[i, :len(r)] = r
        out = np.zeros((len(rows), 4), dtype=np.float64)
        abi.check(self._lib.rtx_shade_probe(self.h, which, len(rows), a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)),
                  "rtx_shade_probe")
        return out

    def shadow_products(self, origin, target, flags=0):
        """rtx_shadow_products_ex: flags 0 = the renderer's product walk, FLAG_REFERENCE_ORDER = the all-f64 walk."""
        o = np.ascontiguousarray(origin, dtype=np.float64).reshape(-1, 3)
        s = np.ascontiguousarray(target, dtype=np.float64).reshape(-1, 3)
        out = np.zeros_like(o)
        abi.check(self._lib.rtx_shadow_products_ex(self.h, o.shape[0], o.ctypes.data_as(C.c_void_p),
                                                   s.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), flags),
                  "rtx_shadow_products_ex")
        return out


def octree_dump(flat):
    """rtx_octree_dump (no device needed): (depth, bbox[n,6], shape[n,2], entries)."""
    lib = abi.load()
    depth, nn, ne = C.c_int32(), C.c_int64(), C.c_int64()
    abi.check(lib.rtx_octree_dump(C.byref(flat.desc), C.byref(depth), C.byref(nn), C.byref(ne), None, None, None),
              "rtx_octree_dump")
    bbox = np.zeros((nn.value, 6), dtype=np.float64)
    shape = np.zeros((nn.value, 2), dtype=np.int32)
    entries = np.zeros(max(1, ne.value), dtype=np.int32)
    abi.check(lib.rtx_octree_dump(C.byref(flat.desc), None, None, None, bbox.ctypes.data_as(C.c_void_p),
                                  shape.ctypes.data_as(C.c_void_p), entries.ctypes.data_as(C.c_void_p)),
              "rtx_octree_dump")
    return depth.value, bbox, shape, entries[:ne.value]


class Renderer:
    """renderer.rs:24-27: `tasks` is kept for API parity (the GPU path has no thread pool)."""

    def __init__(self, options=None, tasks=0, **kw):
        self.tasks = tasks
        self.options = options if options is not None else make_options(**kw)

    def render(self, camera, scene):
        rgba, _, _ = scene.render(camera, self.options, want_ids=False)
        return rgba
