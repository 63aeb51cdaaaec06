"""SceneBuilder: assembles the flattened scene (rtx_scene_desc) from Python.

Mirrors how a scene is written against the reference's types (src/my_scene/*.rs): prims are
pushed in order (the order defines the prim ids, octree.rs:19-20,38), materials are plain
parameter sets (cooktorrancematerial.rs:8-23, phongmaterial.rs:7-21, flatmaterial.rs:6-8),
lights are PointLight / SphereLight (src/light/lights).  The result is only data: numpy
arrays wrapped in the ctypes structs of abi.py.
"""
import ctypes as C

import numpy as np

from . import abi


def _vec3(v):
    a = np.asarray(v, dtype=np.float64).reshape(3)
    return (C.c_double * 3)(*a.tolist())


class SceneBuilder:
    def __init__(self):
        self._spheres = []      # (center, radius, mat)
        self._planes = []       # (abcd, mat)
        self._tri_v = []        # arrays [n,3,3]
        self._tri_n = []
        self._tri_uv = []
        self._tri_m = []
        self._n_tris = 0
        self._order = []        # (kind, index) or (kind, start, count) runs
        self.materials = []
        self.textures = []
        self.images = []
        self.lights = []
        self.background = (0.0, 0.0, 0.0)
        self.skybox = None

    # ---- materials (Default impls: cooktorrancematerial.rs:93-111, phongmaterial.rs:71-88)
    def cook_torrance(self, k_a=0.0, k_d=1.0, k_s=1.0, k_sg=0.0, k_tg=0.0, gauss_constant=1.0,
                      roughness=0.15, glossiness=0.0, ior=1.5, ambient=(1, 1, 1),
                      diffuse=(0.5, 0.5, 0.5), specular=(1, 1, 1), transmission=(0, 0, 0),
                      diffuse_texture=-1):
        m = abi.Material(kind=abi.MAT_COOK_TORRANCE, texture=diffuse_texture, k_a=k_a, k_d=k_d,
                         k_s=k_s, k_sg=k_sg, k_tg=k_tg, ambient=_vec3(ambient),
                         diffuse=_vec3(diffuse), specular=_vec3(specular),
                         transmission=_vec3(transmission), shine=roughness,
                         gauss_constant=gauss_constant, glossiness=glossiness, ior=ior)
        self.materials.append(m)
        return len(self.materials) - 1

    def phong(self, k_a=0.0, k_d=1.0, k_s=1.0, k_sg=0.0, k_tg=0.0, shininess=10.0,
              glossiness=0.0, ior=1.0, ambient=(1, 1, 1), diffuse=(0.5, 0.5, 0.5),
              specular=(1, 1, 1), transmission=(0, 0, 0), diffuse_texture=-1):
        m = abi.Material(kind=abi.MAT_PHONG, texture=diffuse_texture, k_a=k_a, k_d=k_d, k_s=k_s,
                         k_sg=k_sg, k_tg=k_tg, ambient=_vec3(ambient), diffuse=_vec3(diffuse),
                         specular=_vec3(specular), transmission=_vec3(transmission),
                         shine=shininess, gauss_constant=0.0, glossiness=glossiness, ior=ior)
        self.materials.append(m)
        return len(self.materials) - 1

    def flat(self, color=(0.5, 0.5, 0.5)):
        m = abi.Material(kind=abi.MAT_FLAT, texture=-1, diffuse=_vec3(color), ior=1.0)
        self.materials.append(m)
        return len(self.materials) - 1

    # ---- textures
    def checker(self, color1=(0, 0, 0), color2=(1, 1, 1), scale=1.0):
        self.textures.append(abi.Texture(kind=abi.TEX_CHECKER, image=-1, color1=_vec3(color1),
                                         color2=_vec3(color2), scale=scale))
        return len(self.textures) - 1

    def uv_texture(self):
        self.textures.append(abi.Texture(kind=abi.TEX_UV, image=-1, scale=1.0))
        return len(self.textures) - 1

    def image(self, rgba):
        a = np.ascontiguousarray(rgba, dtype=np.uint8)
        assert a.ndim == 3 and a.shape[2] == 4, "image must be [H,W,4] uint8"
        self.images.append(a)
        return len(self.images) - 1

    def image_texture(self, image):
        self.textures.append(abi.Texture(kind=abi.TEX_IMAGE, image=image, scale=1.0))
        return len(self.textures) - 1

    def set_skybox(self, faces):
        """faces: 6 image indices in CubeMap::load order left,right,down,up,front,back."""
        assert len(faces) == 6
        self.skybox = tuple(int(f) for f in faces)

    # ---- lights
    def point_light(self, position, color):
        self.lights.append(abi.Light(kind=abi.LIGHT_POINT, position=_vec3(position), color=_vec3(color)))

    def sphere_light(self, position, color, radius):
        self.lights.append(abi.Light(kind=abi.LIGHT_SPHERE, position=_vec3(position),
                                     color=_vec3(color), radius=radius))

    # ---- prims
    def sphere(self, center, radius, material):
        self._order.append((abi.PRIM_SPHERE, len(self._spheres), 1))
        self._spheres.append((np.asarray(center, dtype=np.float64).reshape(3), float(radius), int(material)))

    def plane(self, a, b, c, d, material):
        self._order.append((abi.PRIM_PLANE, len(self._planes), 1))
        self._planes.append((np.array([a, b, c, d], dtype=np.float64), int(material)))

    def triangles(self, vertices, material, normals=None, uv=None):
        """vertices [n,3,3]; normals [n,3,3] or None (auto: un-normalised face cross product,
        triangle.rs:39-42); uv [n,3,2] or None (default (0.5,1),(0,0),(1,0), triangle.rs:23-29)."""
        v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 3, 3)
        n = v.shape[0]
        if normals is None:
            e1 = v[:, 1] - v[:, 0]
            e2 = v[:, 2] - v[:, 0]
            fn = np.stack([e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1],
                           e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2],
                           e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]], axis=1)
            nn = np.repeat(fn[:, None, :], 3, axis=1)
        else:
            nn = np.ascontiguousarray(normals, dtype=np.float64).reshape(n, 3, 3)
        if uv is None:
            t = np.tile(np.array([[0.5, 1.0], [0.0, 0.0], [1.0, 0.0]], dtype=np.float64), (n, 1, 1))
        else:
            t = np.ascontiguousarray(uv, dtype=np.float64).reshape(n, 3, 2)
        m = np.broadcast_to(np.asarray(material, dtype=np.int32), (n,)).copy()
        self._order.append((abi.PRIM_TRIANGLE, self._n_tris, n))
        self._tri_v.append(v)
        self._tri_n.append(np.ascontiguousarray(nn))
        self._tri_uv.append(t)
        self._tri_m.append(m)
        self._n_tris += n

    def triangle(self, v0, v1, v2, material, normals=None, uv=None):
        self.triangles(np.array([[v0, v1, v2]], dtype=np.float64), material,
                       None if normals is None else np.array([normals], dtype=np.float64),
                       None if uv is None else np.array([uv], dtype=np.float64))

    @property
    def n_prims(self):
        return len(self._spheres) + len(self._planes) + self._n_tris

    # ---- flatten
    def build(self):
        """Returns a FlatScene whose .desc is an abi.SceneDesc (arrays kept alive by the object)."""
        return FlatScene(self)


class FlatScene:
    def __init__(self, b):
        f64, i32 = np.float64, np.int32
        ns, npl, nt = len(b._spheres), len(b._planes), b._n_tris
        self.sphere_center = np.array([s[0] for s in b._spheres], dtype=f64).reshape(ns, 3)
        self.sphere_radius = np.array([s[1] for s in b._spheres], dtype=f64).reshape(ns)
        self.sphere_material = np.array([s[2] for s in b._spheres], dtype=i32).reshape(ns)
        self.plane_abcd = np.array([p[0] for p in b._planes], dtype=f64).reshape(npl, 4)
        self.plane_material = np.array([p[1] for p in b._planes], dtype=i32).reshape(npl)
        cat = lambda parts, shape, dt: (np.ascontiguousarray(np.concatenate(parts, axis=0), dtype=dt)
                                        if parts else np.zeros(shape, dtype=dt))
        self.tri_vertices = cat(b._tri_v, (0, 3, 3), f64)
        self.tri_normals = cat(b._tri_n, (0, 3, 3), f64)
        self.tri_uv = cat(b._tri_uv, (0, 3, 2), f64)
        self.tri_material = cat(b._tri_m, (0,), i32)
        kinds, idxs = [], []
        for kind, start, count in b._order:
            kinds.append(np.full(count, kind, dtype=i32))
            idxs.append(np.arange(start, start + count, dtype=i32))
        self.order_kind = cat(kinds, (0,), i32)
        self.order_index = cat(idxs, (0,), i32)
        self.n_prims = ns + npl + nt
        self.materials = (abi.Material * max(1, len(b.materials)))(*b.materials)
        self.textures = (abi.Texture * max(1, len(b.textures)))(*b.textures)
        self.image_arrays = list(b.images)
        imgs = [abi.Image(width=a.shape[1], height=a.shape[0],
                          rgba=a.ctypes.data_as(abi.c_uint8_p)) for a in self.image_arrays]
        self.images = (abi.Image * max(1, len(imgs)))(*imgs)
        self.lights = (abi.Light * max(1, len(b.lights)))(*b.lights)
        self.n_materials, self.n_textures = len(b.materials), len(b.textures)
        self.n_images, self.n_lights = len(b.images), len(b.lights)

        dp = lambda a: a.ctypes.data_as(abi.c_double_p)
        ip = lambda a: a.ctypes.data_as(abi.c_int32_p)
        d = abi.SceneDesc()
        d.n_spheres, d.sphere_center, d.sphere_radius, d.sphere_material = (
            ns, dp(self.sphere_center), dp(self.sphere_radius), ip(self.sphere_material))
        d.n_planes, d.plane_abcd, d.plane_material = npl, dp(self.plane_abcd), ip(self.plane_material)
        d.n_triangles = nt
        d.tri_vertices, d.tri_normals, d.tri_uv, d.tri_material = (
            dp(self.tri_vertices), dp(self.tri_normals), dp(self.tri_uv), ip(self.tri_material))
        d.n_prims, d.order_kind, d.order_index = self.n_prims, ip(self.order_kind), ip(self.order_index)
        d.n_materials, d.materials = self.n_materials, self.materials
        d.n_textures, d.textures = self.n_textures, self.textures
        d.n_images, d.images = self.n_images, self.images
        d.n_lights, d.lights = self.n_lights, self.lights
        d.background = _vec3(b.background)
        sky = b.skybox if b.skybox is not None else (-1,) * 6
        d.skybox_image = (C.c_int32 * 6)(*sky)
        self.desc = d

    # finite prims in insertion order first, then planes: the prim-id convention of rtx.h
    def prim_id_table(self):
        """Returns (kind, index) arrays indexed by prim id."""
        fin = self.order_kind != abi.PRIM_PLANE
        kind = np.concatenate([self.order_kind[fin], self.order_kind[~fin]])
        idx = np.concatenate([self.order_index[fin], self.order_index[~fin]])
        return kind, idx


def camera(position, look_at, up, fov_deg, width, height):
    return abi.Camera(position=_vec3(position), look_at=_vec3(look_at), up=_vec3(up),
                      fov_deg=float(fov_deg), image_width=int(width), image_height=int(height))


def options(reflect_depth=1, refract_depth=1, shadow_samples=1, gloss_samples=1, pixel_samples=1,
            rng_mode=abi.RNG_CENTER, seed=0, frame=0, shard_index=0, shard_count=1,
            batch_rays=0, pool_rays=0, flags=0):
    return abi.Options(reflect_depth=reflect_depth, refract_depth=refract_depth,
                       shadow_samples=shadow_samples, gloss_samples=gloss_samples,
                       pixel_samples=pixel_samples, rng_mode=rng_mode, seed=seed, frame=frame,
                       shard_index=shard_index, shard_count=shard_count, batch_rays=batch_rays,
                       pool_rays=pool_rays, flags=flags)
