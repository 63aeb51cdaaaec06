"""Small deterministic test scenes (inputs only), built with the product's SceneBuilder."""
import math

import numpy as np


def procedural_mesh(n_u, n_v, radius=1.0, center=(0, 0, 0), bump=0.15, seed=0):
    """A displaced UV-sphere: 2*n_u*(n_v-1) triangles with smooth (un-normalised) vertex normals."""
    rng = np.random.RandomState(seed)
    ph = rng.uniform(0, 2 * math.pi, size=6)
    us = np.linspace(0.0, 2.0 * math.pi, n_u, endpoint=False)
    vs = np.linspace(0.0, math.pi, n_v + 1)
    U, V = np.meshgrid(us, vs, indexing="ij")
    r = radius * (1.0 + bump * (np.sin(3 * U + ph[0]) * np.sin(4 * V + ph[1]) + 0.5 * np.sin(7 * U + ph[2]) * np.sin(5 * V + ph[3])))
    P = np.stack([r * np.sin(V) * np.cos(U), r * np.cos(V), r * np.sin(V) * np.sin(U)], axis=-1) + np.asarray(center, dtype=np.float64)
    N = P - np.asarray(center, dtype=np.float64)
    tris, norms, uvs = [], [], []
    for i in range(n_u):
        i2 = (i + 1) % n_u
        for j in range(n_v):
            a, b, c, d = (i, j), (i2, j), (i2, j + 1), (i, j + 1)
            quads = []
            if j > 0:
                quads.append((a, b, c))
            if j < n_v - 1:
                quads.append((a, c, d))
            for tri in quads:
                tris.append([P[p] for p in tri])
                norms.append([N[p] for p in tri])
                uvs.append([(p[0] / n_u, p[1] / n_v) for p in tri])
    return np.array(tris), np.array(norms), np.array(uvs)


def skybox_faces(size=32, seed=1):
    """Six smooth RGBA8 faces (the reference's skybox PNGs are not in the container)."""
    rng = np.random.RandomState(seed)
    faces = []
    yy, xx = np.meshgrid(np.linspace(0, 1, size), np.linspace(0, 1, size), indexing="ij")
    for f in range(6):
        c0, c1 = rng.uniform(0.1, 0.9, size=3), rng.uniform(0.1, 0.9, size=3)
        img = np.zeros((size, size, 4), dtype=np.uint8)
        for ch in range(3):
            val = c0[ch] * (1 - yy) + c1[ch] * yy + 0.1 * np.sin(6.0 * xx + f)
            img[..., ch] = np.clip(val * 255.0, 0, 255).astype(np.uint8)
        img[..., 3] = 255
        faces.append(img)
    return faces


def sphere_scene(rtx):
    """my_scene/sphere.rs:83-109 with a procedural skybox."""
    b = rtx.SceneBuilder()
    b.sphere_light((3, 10, 6), (1, 1, 1), 5.0)
    shiny = b.cook_torrance(k_a=0.0, k_d=0.2, k_s=1.0, k_sg=1.0, k_tg=0.0, gauss_constant=5.0, roughness=0.01,
                            glossiness=0.0, ior=0.05, diffuse=(1, 1, 1), specular=(0.9, 0.9, 0.9))
    b.sphere((0, 0, 0), 2.0, shiny)
    b.background = (0.3, 0.5, 0.8)
    b.set_skybox([b.image(f) for f in skybox_faces()])
    cam = lambda w, h, fov=30.0: rtx.camera((0, 0, 10), (0, 0, 0), (0, 1, 0), fov, w, h)
    return b.build(), cam


def cornell_scene(rtx, glossy=False):
    """my_scene/cornell.rs:27-75 (all prim kinds, CT + Phong, checker, refraction)."""
    b = rtx.SceneBuilder()
    b.sphere_light((50, 80, 50), (1, 1, 1), 10.0)
    checker = b.checker((1, 1, 1), (0.8, 0.1, 0.1), 16.0)
    checker_grey = b.cook_torrance(k_a=0.0, k_d=1.0, k_s=0.0, gauss_constant=1.0, roughness=0.15, ior=0.7,
                                   diffuse=(0.6, 0.6, 0.6), diffuse_texture=checker)
    grey = b.cook_torrance(diffuse=(0.6, 0.6, 0.6))
    blue = b.cook_torrance(k_d=0.3, k_s=0.7, gauss_constant=50.0, roughness=0.1, ior=1.3, diffuse=(0.1, 0.1, 1.0))
    red = b.phong(k_d=0.6, k_s=0.4, k_sg=0.8, shininess=10.0, ior=0.5, diffuse=(1, 0, 0))
    green = b.phong(k_d=0.9, k_s=0.1, k_sg=0.5, shininess=10.0, ior=0.7, diffuse=(0, 1, 0))
    shiny = b.cook_torrance(k_d=0.2, k_s=1.0, k_sg=0.8, gauss_constant=5.0, roughness=0.01, ior=0.25,
                            diffuse=(1, 1, 1), specular=(0.9, 0.9, 0.9))
    shiny_glossy = b.cook_torrance(k_d=0.7, k_s=1.0, k_sg=0.4, gauss_constant=5.0, roughness=0.01,
                                   glossiness=0.2 if glossy else 0.0, ior=0.25, diffuse=(0.3, 0.3, 1.0),
                                   specular=(0.3, 0.3, 1.0))
    refract = b.cook_torrance(k_d=0.0, k_s=1.0, k_sg=1.0, k_tg=1.0, gauss_constant=5.0, roughness=0.01, ior=3.0,
                              diffuse=(1, 1, 1), specular=(0.9, 0.9, 0.9), transmission=(0.8, 0.8, 0.8))
    b.plane(0, 0, 1, 0, grey)
    b.plane(0, 1, 0, 0, checker_grey)
    b.plane(0, -1, 0, 100, grey)
    b.plane(1, 0, 0, 0, red)
    b.plane(-1, 0, 0, 100, green)
    b.sphere((30, 15, 20), 15.0, shiny)
    b.sphere((70, 17, 60), 17.0, refract)
    b.sphere((50, 50, 20), 10.0, shiny_glossy)
    b.sphere((20, 13, 90), 13.0, blue)
    b.triangle((20, 95, 20), (15, 50, 40), (35, 50, 35), blue, uv=[(0.5, 1.0), (0.0, 0.0), (1.0, 0.0)])
    b.background = (1, 1, 1)
    cam = lambda w, h, fov=30.0: rtx.camera((50, 25, 150), (50, 50, 50), (0, 1, 0), fov, w, h)
    return b.build(), cam


def mesh_scene(rtx, n_u=24, n_v=16, skybox=True, transmissive=False, point_light=False, third_light=False):
    """A bunny.rs-style scene: ground plane + two mirror spheres + a displaced-sphere mesh."""
    b = rtx.SceneBuilder()
    b.sphere_light((200, -200, 100), (1, 1, 1), 40.0)
    if third_light:
        b.sphere_light((60, -120, 220), (0.2, 0.3, 0.4), 25.0)
    if point_light:
        b.point_light((-95, 20, 170), (0.5, 0.5, 0.3))
    else:
        b.sphere_light((-95, 20, 170), (0.5, 0.5, 0.3), 15.0)
    red = b.cook_torrance(k_a=0.1, k_d=0.4, k_s=0.5, k_sg=0.5, gauss_constant=5.0, roughness=0.05, ior=0.98,
                          diffuse=(1.0, 0.25, 0.1), k_tg=0.6 if transmissive else 0.0,
                          transmission=(0.8, 0.6, 0.4) if transmissive else (0, 0, 0))
    green = b.cook_torrance(k_d=0.4, k_s=0.6, k_sg=0.7, gauss_constant=50.0, roughness=0.3, ior=1.5, diffuse=(0.2, 0.7, 0.2))
    shiny = b.cook_torrance(k_d=0.2, k_s=0.7, k_sg=1.0, gauss_constant=25.0, roughness=0.01, ior=0.2,
                            diffuse=(0.9, 0.9, 0.1), specular=(0.9, 0.9, 0.1))
    b.plane(0, 0, 1, -10, green)
    b.sphere((-75, 60, 50), 40.0, shiny)
    b.sphere((-75, 60, 140), 40.0, shiny)
    v, n, uv = procedural_mesh(n_u, n_v, radius=45.0, center=(20, 60, 50), seed=2)
    # z-up scene: swap y/z of the generated mesh around its centre
    c = np.array([20.0, 60.0, 50.0])
    v = (v - c)[..., [0, 2, 1]] + c
    n = n[..., [0, 2, 1]]
    b.triangles(v, red, normals=n, uv=uv)
    b.background = (0.3, 0.5, 0.8)
    if skybox:
        b.set_skybox([b.image(f) for f in skybox_faces()])
    cam = lambda w, h, fov=45.0: rtx.camera((0, -150, 30), (0, 60, 50), (0, 0, 1), fov, w, h)
    return b.build(), cam


def cube_obj_scene(rtx):
    """test/res/cube.obj of the reference: 8 v, 7 vn, 12 f `a//n` (data transcribed as a fixture)."""
    V = [(1, -1, -1), (1, -1, 1), (-1, -1, 1), (-1, -1, -1), (1, 1, -1), (1, 1, 1.000001), (-1, 1, 1), (-1, 1, -1)]
    VN = [(0, -1, 0), (0, 1, -0.0), (1, 0, 0), (-0.0, -0.0, 1), (-1, -0.0, -0.0), (0, 0, -1), (1, -0.0, 0)]
    F = [(2, 3, 4, 1), (5, 8, 7, 2), (5, 6, 2, 3), (2, 6, 7, 4), (7, 8, 4, 5), (1, 4, 8, 6),
         (1, 2, 4, 1), (6, 5, 7, 2), (1, 5, 2, 7), (3, 2, 7, 4), (3, 7, 4, 5), (5, 1, 8, 6)]
    b = rtx.SceneBuilder()
    b.sphere_light((3, 4, 5), (1, 1, 1), 0.5)
    mat = b.cook_torrance()
    verts = np.array([[V[a - 1], V[bb - 1], V[c - 1]] for a, bb, c, _ in F], dtype=np.float64)
    norms = np.array([[VN[nn - 1]] * 3 for _, _, _, nn in F], dtype=np.float64)
    uv = np.zeros((12, 3, 2))     # no `vt`: all zero (util/import.rs:94-97)
    b.triangles(verts, mat, normals=norms, uv=uv)
    b.background = (0.1, 0.2, 0.3)
    cam = lambda w, h, fov=30.0: rtx.camera((3, 2.5, 4), (0, 0, 0), (0, 1, 0), fov, w, h)
    return b.build(), cam


def kitchen_sink_scene(rtx, lights="mixed"):
    """Every material/texture/light kind the path has in one small scene: UV, image and checker textures on
    Cook-Torrance and Phong, a FlatMaterial, a glossy-capable mirror, a transmissive sphere, point + sphere
    lights (or none: `lights="none"`, or three point lights: `lights="points"`), finite prims of all kinds."""
    b = rtx.SceneBuilder()
    if lights == "mixed":
        b.point_light((40, 90, 120), (0.7, 0.7, 0.6))
        b.sphere_light((-30, 60, 80), (0.4, 0.5, 0.6), 12.0)
    elif lights == "points":
        b.point_light((40, 90, 120), (0.5, 0.5, 0.4))
        b.point_light((-60, 40, 90), (0.3, 0.2, 0.2))
        b.point_light((0, 150, 10), (0.2, 0.3, 0.4))
    rng = np.random.RandomState(7)
    img = np.zeros((9, 13, 4), dtype=np.uint8)          # odd, non-square image (bilinear x+1 / y+1 taps)
    img[..., :3] = rng.randint(0, 256, size=(9, 13, 3))
    img[..., 3] = 255
    tex_img = b.image_texture(b.image(img))
    tex_uv = b.uv_texture()
    tex_chk = b.checker((0.9, 0.9, 0.2), (0.1, 0.1, 0.5), 0.25)
    m_img = b.cook_torrance(k_a=0.05, k_d=0.8, k_s=0.3, gauss_constant=10.0, roughness=0.2, ior=1.4, diffuse=(1, 1, 1), diffuse_texture=tex_img)
    m_uv = b.phong(k_a=0.1, k_d=0.8, k_s=0.5, shininess=20.0, ior=1.2, diffuse=(1, 1, 1), diffuse_texture=tex_uv)
    m_chk = b.phong(k_d=0.7, k_s=0.2, k_sg=0.3, shininess=5.0, ior=0.6, diffuse=(0.8, 0.8, 0.8), diffuse_texture=tex_chk)
    m_flat = b.flat((0.2, 0.9, 0.4))
    m_mirror = b.cook_torrance(k_d=0.1, k_s=0.9, k_sg=0.9, gauss_constant=5.0, roughness=0.02, ior=0.3, diffuse=(0.9, 0.9, 1.0))
    m_glass = b.cook_torrance(k_d=0.0, k_s=1.0, k_sg=1.0, k_tg=1.0, gauss_constant=5.0, roughness=0.01, ior=1.5,
                              diffuse=(1, 1, 1), specular=(0.9, 0.9, 0.9), transmission=(0.7, 0.8, 0.9))
    b.plane(0, 1, 0, 20, m_chk)                          # floor y = -20
    b.plane(0, 0, 1, 60, m_img)                          # back wall z = -60
    b.sphere((-35, 0, 0), 18.0, m_uv)
    b.sphere((35, -2, 10), 16.0, m_glass)
    b.sphere((0, 28, -20), 14.0, m_mirror)
    v, n, uv = procedural_mesh(12, 8, radius=14.0, center=(0, -6, 25), seed=4)
    b.triangles(v, m_img, normals=n, uv=uv)
    b.triangle((-60, -20, 40), (-40, 25, 30), (-20, -20, 45), m_flat)
    b.triangle((20, -20, 50), (40, 10, 45), (60, -20, 35), m_uv, uv=[(0.1, 0.9), (0.5, 0.1), (0.9, 0.8)])
    b.background = (0.05, 0.07, 0.12)
    cam = lambda w, h, fov=35.0: rtx.camera((0, 10, 140), (0, 0, 0), (0, 1, 0), fov, w, h)
    return b.build(), cam


def axis_aligned_scene(rtx, transmissive=False):
    """Everything on integer coordinates and aligned with the axes: a tessellated floor, two boxes, a point light and a camera
    that looks straight down from above the centre.  With an odd image width the centre column of primary rays has a direction
    component that is exactly zero and origins on octree planes; its hit points share a coordinate with the light, so their
    shadow feelers are degenerate too, and the mirror-like floor sends reflection rays straight back up."""
    b = rtx.SceneBuilder()
    b.point_light((0, 6, 1), (1.0, 0.9, 0.8))
    b.sphere_light((2, 7, 0), (0.3, 0.3, 0.5), 1.0)
    floor = b.cook_torrance(k_a=0.1, k_d=0.6, k_s=0.4, k_sg=0.4, gauss_constant=5.0, roughness=0.1, ior=1.3, diffuse=(0.7, 0.7, 0.7))
    # transmissive: the boxes let light through (shadow product instead of any-hit, the closest-hit walk with dealt exact tests)
    box = b.cook_torrance(k_a=0.1, k_d=0.7, k_s=0.3, k_sg=0.2, gauss_constant=5.0, roughness=0.2, ior=1.3, diffuse=(0.8, 0.3, 0.2),
                          **(dict(k_tg=0.5, transmission=(0.8, 0.6, 0.4)) if transmissive else {}))
    tris = []
    for i in range(-4, 4):
        for j in range(-4, 4):
            a, bb, c, d = (i, 0, j), (i + 1, 0, j), (i + 1, 0, j + 1), (i, 0, j + 1)
            tris += [(a, c, bb), (a, d, c)]
    b.triangles(np.array(tris, dtype=np.float64), floor)
    def cuboid(lo, hi):
        x0, y0, z0 = lo
        x1, y1, z1 = hi
        v = [(x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)]
        q = [(0, 1, 2, 3), (5, 4, 7, 6), (4, 0, 3, 7), (1, 5, 6, 2), (3, 2, 6, 7), (4, 5, 1, 0)]
        out = []
        for a, bb, c, d in q:
            out += [(v[a], v[bb], v[c]), (v[a], v[c], v[d])]
        return np.array(out, dtype=np.float64)
    b.triangles(cuboid((-1, 0, -1), (1, 2, 0)), box)      # straddles x = 0, ends at z = 0
    b.triangles(cuboid((2, 0, 1), (3, 1, 3)), box)
    b.background = (0.2, 0.3, 0.4)
    cam = lambda w, h, fov=25.0: rtx.camera((0, 10, 0), (0, 0, 0), (0, 0, -1), fov, w, h)
    return b.build(), cam
