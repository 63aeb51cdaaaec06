"""ctypes binding of librtx_host.so: the C++ mirror of the reference's Scene/Camera API
(host/rt.h), its scene registry (src/my_scene/mod.rs:27-45) and loaders/writers (src/util)."""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librtx_host.so")
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("%s is missing: run __graft_entry__.build()" % LIB_PATH)
    abi.load()  # librtx.so first (rpath $ORIGIN resolves it too)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rtxh_last_error.restype = C.c_char_p
    L.rtxh_scene_by_name.argtypes = [C.c_char_p, C.c_char_p, C.c_double]
    L.rtxh_scene_by_name.restype = vp
    L.rtxh_scene_free.argtypes = [vp]
    L.rtxh_scene_desc.argtypes = [vp]
    L.rtxh_scene_desc.restype = C.POINTER(abi.SceneDesc)
    L.rtxh_scene_camera.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_double, C.c_int, C.c_double, C.POINTER(abi.Camera)]
    L.rtxh_from_obj.argtypes = [C.c_char_p, C.c_int]
    L.rtxh_from_obj.restype = vp
    L.rtxh_mesh_size.argtypes = [vp]
    L.rtxh_mesh_size.restype = C.c_int64
    L.rtxh_mesh_copy.argtypes = [vp, vp, vp, vp]
    L.rtxh_mesh_transform.argtypes = [vp, vp]
    L.rtxh_mesh_free.argtypes = [vp]
    L.rtxh_from_image.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), vp]
    L.rtxh_to_ppm.argtypes = [vp, C.c_int32, C.c_int32, C.c_char_p]
    L.rtxh_mat4_inverse.argtypes = [vp, vp]
    L.rtxh_mat4_mult_p.argtypes = [vp, vp, vp]
    L.rtxh_easing_t.argtypes = [C.c_double] * 5
    L.rtxh_easing_t.restype = C.c_double
    L.rtxh_lerp_camera.argtypes = [C.POINTER(abi.Camera), vp, C.c_int, C.c_double, C.POINTER(abi.Camera)]
    L.rtxh_parse_config.argtypes = [C.c_char_p, C.c_char_p, C.c_int, vp, vp]
    _lib = L
    return L


def last_error():
    return load().rtxh_last_error().decode()


class NamedScene:
    """scene_by_name(name).get_scene(), flattened.  `.desc` is an abi.SceneDesc owned by the C++ side
    (quacks like desc.FlatScene for rtx.Scene / the oracle)."""

    def __init__(self, name, asset_root=".", detail=1.0):
        L = load()
        self.h = L.rtxh_scene_by_name(name.encode(), asset_root.encode(), float(detail))
        if not self.h:
            raise KeyError(last_error())
        self.name = name
        self.desc = L.rtxh_scene_desc(self.h).contents
        d = self.desc
        self.lights = [d.lights[i] for i in range(d.n_lights)]
        self.n_prims = d.n_prims
        self.n_triangles = d.n_triangles

    def camera(self, width, height, fov, time=None):
        cam = abi.Camera()
        rc = load().rtxh_scene_camera(self.h, width, height, float(fov), 0 if time is None else 1,
                                      0.0 if time is None else float(time), C.byref(cam))
        if rc:
            raise RuntimeError(last_error())
        return cam

    def __del__(self):
        if getattr(self, "h", None):
            load().rtxh_scene_free(self.h)
            self.h = None


def from_obj(filename, flip_normals=False, transform=None):
    L = load()
    m = L.rtxh_from_obj(filename.encode(), int(flip_normals))
    if not m:
        raise IOError(last_error())
    try:
        if transform is not None:
            t = np.ascontiguousarray(transform, dtype=np.float64).reshape(4, 4)
            L.rtxh_mesh_transform(m, t.ctypes.data_as(C.c_void_p))
        n = L.rtxh_mesh_size(m)
        v, nn, uv = np.zeros((n, 3, 3)), np.zeros((n, 3, 3)), np.zeros((n, 3, 2))
        L.rtxh_mesh_copy(m, v.ctypes.data_as(C.c_void_p), nn.ctypes.data_as(C.c_void_p), uv.ctypes.data_as(C.c_void_p))
        return v, nn, uv
    finally:
        L.rtxh_mesh_free(m)


def from_image(path):
    L = load()
    w, h = C.c_int32(), C.c_int32()
    if L.rtxh_from_image(path.encode(), C.byref(w), C.byref(h), None):
        raise IOError(last_error())
    out = np.zeros((h.value, w.value, 4), dtype=np.uint8)
    L.rtxh_from_image(path.encode(), C.byref(w), C.byref(h), out.ctypes.data_as(C.c_void_p))
    return out


def to_ppm(rgba, filename):
    a = np.ascontiguousarray(rgba, dtype=np.uint8)
    if load().rtxh_to_ppm(a.ctypes.data_as(C.c_void_p), a.shape[1], a.shape[0], filename.encode()):
        raise IOError(last_error())


def mat4_inverse(m):
    a = np.ascontiguousarray(m, dtype=np.float64).reshape(4, 4)
    out = np.zeros((4, 4))
    load().rtxh_mat4_inverse(a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def mat4_mult_p(m, p):
    a = np.ascontiguousarray(m, dtype=np.float64).reshape(4, 4)
    pp = np.ascontiguousarray(p, dtype=np.float64).reshape(3)
    out = np.zeros(3)
    load().rtxh_mat4_mult_p(a.ctypes.data_as(C.c_void_p), pp.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def easing_t(t, a=0.0, b=1.0 / 3.0, c=2.0 / 3.0, d=1.0):
    return load().rtxh_easing_t(a, b, c, d, t)


def lerp_camera(base, keyframes, time):
    k = np.ascontiguousarray(keyframes, dtype=np.float64).reshape(-1, 10)
    out = abi.Camera()
    if load().rtxh_lerp_camera(C.byref(base), k.ctypes.data_as(C.c_void_p), k.shape[0], float(time), C.byref(out)):
        raise RuntimeError(last_error())
    return out


def parse_config(text):
    name = C.create_string_buffer(256)
    u = np.zeros(8, dtype=np.uint32)
    f = np.zeros(4, dtype=np.float64)
    if load().rtxh_parse_config(text.encode(), name, 256, u.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p)):
        raise ValueError(last_error())
    return {"name": name.value.decode(), "size": (int(u[0]), int(u[1])), "reflect_depth": int(u[2]),
            "refract_depth": int(u[3]), "shadow_samples": int(u[4]), "gloss_samples": int(u[5]),
            "pixel_samples": int(u[6]), "starting_frame_number": int(u[7]), "fov": float(f[0]),
            "fps": float(f[1]), "time_slice": (float(f[2]), float(f[3]))}
