"""ctypes rendering of include/rtx.h (the C ABI of the MI355X trace loop).

Only interface definitions live here: structs, enums and the prototypes of librtx.so.
There is no CPU fallback: `load()` raises if the HIP library has not been built.
"""
import ctypes as C
import os

RTX_OK = 0
RTX_ERR_INVALID_ARGUMENT = 1
RTX_ERR_NO_DEVICE = 2
RTX_ERR_HIP = 3
RTX_ERR_OUT_OF_MEMORY = 4
RTX_ERR_LIMIT = 5
RTX_ERR_QUEUE_OVERFLOW = 6

MAT_COOK_TORRANCE, MAT_PHONG, MAT_FLAT = 0, 1, 2
TEX_CHECKER, TEX_UV, TEX_IMAGE = 0, 1, 2
LIGHT_POINT, LIGHT_SPHERE = 0, 1
PRIM_SPHERE, PRIM_PLANE, PRIM_TRIANGLE = 0, 1, 2
RNG_CENTER, RNG_PHILOX = 0, 1
FLAG_REFERENCE_ORDER = 1
FLAG_NO_COOP = 2
FLAG_SERIAL = 4
FLAG_OCTREE_WALK = 8
FLAG_COMPACT_OUTPUT = 16
PROBE_FRESNEL, PROBE_REFRACT, PROBE_REFLECT, PROBE_MATERIAL, PROBE_TEXTURE, PROBE_SKYBOX, PROBE_BBOX = range(7)

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


class Material(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("texture", C.c_int32),
        ("k_a", C.c_double), ("k_d", C.c_double), ("k_s", C.c_double),
        ("k_sg", C.c_double), ("k_tg", C.c_double),
        ("ambient", C.c_double * 3), ("diffuse", C.c_double * 3),
        ("specular", C.c_double * 3), ("transmission", C.c_double * 3),
        ("shine", C.c_double), ("gauss_constant", C.c_double),
        ("glossiness", C.c_double), ("ior", C.c_double),
    ]


class Texture(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("image", C.c_int32),
        ("color1", C.c_double * 3), ("color2", C.c_double * 3), ("scale", C.c_double),
    ]


class Image(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgba", c_uint8_p)]


class Light(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("reserved", C.c_int32),
        ("position", C.c_double * 3), ("color", C.c_double * 3), ("radius", C.c_double),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("n_spheres", C.c_int32), ("sphere_center", c_double_p), ("sphere_radius", c_double_p),
        ("sphere_material", c_int32_p),
        ("n_planes", C.c_int32), ("plane_abcd", c_double_p), ("plane_material", c_int32_p),
        ("n_triangles", C.c_int32), ("tri_vertices", c_double_p), ("tri_normals", c_double_p),
        ("tri_uv", c_double_p), ("tri_material", c_int32_p),
        ("n_prims", C.c_int32), ("order_kind", c_int32_p), ("order_index", c_int32_p),
        ("n_materials", C.c_int32), ("materials", C.POINTER(Material)),
        ("n_textures", C.c_int32), ("textures", C.POINTER(Texture)),
        ("n_images", C.c_int32), ("images", C.POINTER(Image)),
        ("n_lights", C.c_int32), ("lights", C.POINTER(Light)),
        ("background", C.c_double * 3), ("skybox_image", C.c_int32 * 6),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("position", C.c_double * 3), ("look_at", C.c_double * 3), ("up", C.c_double * 3),
        ("fov_deg", C.c_double), ("image_width", C.c_uint32), ("image_height", C.c_uint32),
    ]


class Options(C.Structure):
    _fields_ = [
        ("reflect_depth", C.c_uint32), ("refract_depth", C.c_uint32),
        ("shadow_samples", C.c_uint32), ("gloss_samples", C.c_uint32),
        ("pixel_samples", C.c_uint32), ("rng_mode", C.c_uint32),
        ("seed", C.c_uint64), ("frame", C.c_uint32),
        ("shard_index", C.c_uint32), ("shard_count", C.c_uint32),
        ("batch_rays", C.c_uint32), ("pool_rays", C.c_uint32), ("flags", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("rays_primary", C.c_uint64), ("rays_reflect", C.c_uint64), ("rays_gloss", C.c_uint64),
        ("rays_refract", C.c_uint64), ("rays_shadow", C.c_uint64),
        ("ms_total", C.c_double), ("ms_device", C.c_double),
        ("ms_closest", C.c_double), ("ms_shadow", C.c_double),
        ("launches_closest", C.c_uint32), ("launches_shadow", C.c_uint32),
        ("peak_pool_rays", C.c_uint64), ("work", (C.c_uint64 * 3) * 2), ("exact_shadow_rerenders", C.c_uint64),
        ("ms_shade", C.c_double), ("ms_shadow_light", C.c_double * 16),
    ]

    @property
    def rays_total(self):
        return (self.rays_primary + self.rays_reflect + self.rays_gloss +
                self.rays_refract + self.rays_shadow)


class SceneInfo(C.Structure):
    _fields_ = [
        ("n_finite", C.c_int32), ("n_infinite", C.c_int32), ("octree_depth", C.c_int32),
        ("n_nodes", C.c_int32), ("n_interior", C.c_int32), ("n_leaf_entries", C.c_int64),
        ("max_leaf_entries", C.c_int32), ("all_opaque", C.c_int32), ("device_bytes", C.c_int64),
        ("root_min", C.c_double * 3), ("root_max", C.c_double * 3),
        ("wide_nodes", C.c_int32), ("wide_depth", C.c_int32),
        ("bvh4_nodes", C.c_int32), ("hierarchy_on_device", C.c_int32),
    ]


# every symbol include/rtx.h declares
EXPORTS = (
    "rtx_scene_create", "rtx_scene_destroy", "rtx_scene_get_info", "rtx_render",
    "rtx_render_device", "rtx_render_device_noalloc", "rtx_nearest_hits", "rtx_nearest_hits_ex", "rtx_shadow_products", "rtx_shadow_products_ex", "rtx_shade_probe", "rtx_octree_dump", "rtx_hierarchy_dump", "rtx_last_kernel_names", "rtx_strerror",
    "rtx_last_error", "rtx_version",
)

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTX_LIB=librtx_count.so selects the profiling twin (same ABI, kernels also count their work)
LIB_PATH = os.path.join(_HERE, os.environ.get("RTX_LIB", "librtx.so"))
_lib = None


def load():
    """dlopen librtx.so (HIP/gfx950). Raises if it has not been built: no fallback exists."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the trace path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.rtx_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int32, C.POINTER(vp)]
    lib.rtx_scene_create.restype = C.c_int32
    lib.rtx_scene_destroy.argtypes = [vp]
    lib.rtx_scene_destroy.restype = C.c_int32
    lib.rtx_scene_get_info.argtypes = [vp, C.POINTER(SceneInfo)]
    lib.rtx_scene_get_info.restype = C.c_int32
    lib.rtx_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(Options), vp, vp, C.POINTER(Stats)]
    lib.rtx_render.restype = C.c_int32
    lib.rtx_render_device.argtypes = [vp, C.POINTER(Camera), C.POINTER(Options), vp, vp, vp, C.POINTER(Stats)]
    lib.rtx_render_device.restype = C.c_int32
    lib.rtx_render_device_noalloc.argtypes = [vp, C.POINTER(Camera), C.POINTER(Options), vp, vp, vp, C.POINTER(Stats)]
    lib.rtx_render_device_noalloc.restype = C.c_int32
    lib.rtx_nearest_hits.argtypes = [vp, C.c_int64, vp, vp, vp, vp]
    lib.rtx_nearest_hits.restype = C.c_int32
    lib.rtx_nearest_hits_ex.argtypes = [vp, C.c_int64, vp, vp, vp, vp, C.c_uint32]
    lib.rtx_nearest_hits_ex.restype = C.c_int32
    lib.rtx_shadow_products.argtypes = [vp, C.c_int64, vp, vp, vp]
    lib.rtx_shadow_products.restype = C.c_int32
    lib.rtx_shadow_products_ex.argtypes = [vp, C.c_int64, vp, vp, vp, C.c_uint32]
    lib.rtx_shadow_products_ex.restype = C.c_int32
    lib.rtx_shade_probe.argtypes = [vp, C.c_int32, C.c_int64, vp, vp]
    lib.rtx_shade_probe.restype = C.c_int32
    lib.rtx_octree_dump.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int64), vp, vp, vp]
    lib.rtx_octree_dump.restype = C.c_int32
    lib.rtx_hierarchy_dump.argtypes = [vp, C.c_int32, vp, C.c_int64, C.POINTER(C.c_int64)]
    lib.rtx_hierarchy_dump.restype = C.c_int32
    lib.rtx_last_kernel_names.argtypes = [vp, C.c_char_p, C.c_int64]
    lib.rtx_last_kernel_names.restype = C.c_int32
    lib.rtx_strerror.argtypes = [C.c_int32]
    lib.rtx_strerror.restype = C.c_char_p
    lib.rtx_last_error.argtypes = []
    lib.rtx_last_error.restype = C.c_char_p
    lib.rtx_version.argtypes = []
    lib.rtx_version.restype = C.c_char_p
    _lib = lib
    return lib


class RtxError(RuntimeError):
    def __init__(self, status, where):
        lib = load()
        msg = lib.rtx_strerror(status).decode()
        detail = lib.rtx_last_error().decode()
        super().__init__("%s: %s (%d)%s" % (where, msg, status, (": " + detail) if detail else ""))
        self.status = status


def check(status, where):
    if status != RTX_OK:
        raise RtxError(status, where)
