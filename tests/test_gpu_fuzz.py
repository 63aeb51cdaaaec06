"""Seeded random scenes against the oracle: triangle soups, spheres and planes of very different sizes, all material and light kinds,
cameras inside and far outside the scene's cube (the f32 frame of the fast walks is re-based where a ray enters it), opaque scenes (the
4-wide hierarchy walks, the cooperative shadow walk) and scenes with a transmissive material (the shadow product over the octree).
Nothing here is tuned to pass: prim ids, RGB within 1 and the ray count must match the oracle's."""
import numpy as np
import pytest


def random_scene(rtx, seed, transmissive):
    rng = np.random.default_rng(seed)
    b = rtx.SceneBuilder()
    for _ in range(int(rng.integers(1, 4))):
        pos = rng.uniform(-150, 150, 3) + np.array([0.0, 200.0, 0.0])
        col = rng.uniform(0.2, 0.8, 3)
        if rng.random() < 0.3:
            b.point_light(tuple(pos), tuple(col))
        else:
            b.sphere_light(tuple(pos), tuple(col), float(rng.uniform(2.0, 40.0)))
    chk = b.checker((0.9, 0.9, 0.9), (0.2, 0.3, 0.5), float(rng.uniform(0.05, 4.0)))
    mats = []
    for i in range(5):
        k_sg = float(rng.choice([0.0, 0.0, 0.4, 0.9]))
        if i == 0 and transmissive:
            mats.append(b.cook_torrance(k_d=0.1, k_s=0.9, k_sg=0.8, k_tg=0.9, gauss_constant=5.0, roughness=0.02, ior=float(rng.uniform(1.1, 2.0)),
                                        diffuse=(1, 1, 1), transmission=tuple(rng.uniform(0.3, 0.95, 3))))
        elif rng.random() < 0.4:
            mats.append(b.phong(k_a=0.05, k_d=0.7, k_s=0.4, k_sg=k_sg, shininess=float(rng.uniform(2.0, 40.0)), ior=float(rng.uniform(0.3, 1.6)),
                                diffuse=tuple(rng.uniform(0.1, 1.0, 3)), diffuse_texture=chk if rng.random() < 0.5 else -1))
        elif rng.random() < 0.15:
            mats.append(b.flat(tuple(rng.uniform(0.1, 1.0, 3))))
        else:
            mats.append(b.cook_torrance(k_a=0.05, k_d=0.6, k_s=0.5, k_sg=k_sg, gauss_constant=float(rng.uniform(1.0, 20.0)),
                                        roughness=float(rng.uniform(0.02, 0.4)), ior=float(rng.uniform(0.3, 1.6)),
                                        diffuse=tuple(rng.uniform(0.1, 1.0, 3)), diffuse_texture=chk if rng.random() < 0.3 else -1))
    b.plane(0, 1, 0, 60, mats[1])                                   # floor y = -60
    if rng.random() < 0.5:
        b.plane(0, 0, 1, 220, mats[2])                              # back wall z = -220
    # clusters of small triangles, a few huge ones, some slivers, some exactly axis-aligned quads
    n_clusters = int(rng.integers(3, 9))
    for _ in range(n_clusters):
        c = rng.uniform(-100, 100, 3)
        size = float(10.0 ** rng.uniform(-0.5, 1.5))
        n = int(rng.integers(20, 400))
        v0 = c + rng.normal(0, 12.0, (n, 3))
        tri = np.stack([v0, v0 + rng.normal(0, size, (n, 3)), v0 + rng.normal(0, size, (n, 3))], axis=1)
        b.triangles(tri.reshape(-1, 9), mats[int(rng.integers(0, 5))])
    for _ in range(int(rng.integers(0, 4))):
        c = rng.uniform(-120, 120, 3)
        b.triangles(np.stack([c, c + rng.normal(0, 150.0, 3), c + rng.normal(0, 150.0, 3)]).reshape(1, 9), mats[int(rng.integers(0, 5))])
    x0, y0, z0 = rng.uniform(-80, 80, 3)
    quad = np.array([[x0, y0, z0, x0 + 40, y0, z0, x0 + 40, y0, z0 + 30], [x0, y0, z0, x0 + 40, y0, z0 + 30, x0, y0, z0 + 30]], dtype=np.float64)
    b.triangles(quad, mats[3])
    for _ in range(int(rng.integers(0, 5))):
        b.sphere(tuple(rng.uniform(-100, 100, 3)), float(rng.uniform(1.0, 35.0)), mats[int(rng.integers(0, 5))])
    b.background = tuple(rng.uniform(0.0, 1.0, 3))
    far = rng.random() < 0.4                                        # camera far outside the scene's cube
    pos = rng.uniform(-60, 60, 3) + (np.array([0.0, 300.0, 2500.0]) if far else np.array([0.0, 40.0, 260.0]))
    cam = rtx.camera(tuple(pos), tuple(rng.uniform(-30, 30, 3)), (0, 1, 0), 8.0 if far else 35.0, 96, 64)
    return b.build(), cam


OPTIONS = [
    dict(reflect_depth=4, refract_depth=4, shadow_samples=1, gloss_samples=1, pixel_samples=1, rng_mode=0),
    dict(reflect_depth=3, refract_depth=3, shadow_samples=16, gloss_samples=1, pixel_samples=1, rng_mode=1, seed=5),
    dict(reflect_depth=2, refract_depth=3, shadow_samples=4, gloss_samples=1, pixel_samples=2, rng_mode=1, seed=9),
    dict(reflect_depth=3, refract_depth=2, shadow_samples=3, gloss_samples=1, pixel_samples=1, rng_mode=1, seed=1),
]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(24))
def test_random_scene_matches_the_oracle(rtx, orc, seed):
    flat, cam = random_scene(rtx, 1000 + seed, transmissive=seed % 3 == 2)
    sc = rtx.Scene(flat)
    oscene = orc.Scene(flat)
    for kw in (OPTIONS[seed % 4], OPTIONS[(seed + 1) % 4]):
        opt = rtx.options(**kw)
        rgba, ids, st = sc.render(cam, opt)
        orgba, oids, ocn = oscene.render(cam, opt)
        assert st.rays_total == ocn.rays_total, (seed, kw)
        assert (ids == oids).mean() >= 0.9999, (seed, kw, float((ids == oids).mean()))
        close = (np.abs(rgba[..., :3].astype(int) - orgba[..., :3].astype(int)).max(axis=-1) <= 1).mean()
        assert close >= 0.999, (seed, kw, float(close))
    # closest hits of random rays, origins inside and far outside the scene's cube: t bit for bit, prim ids but for exact-t ties
    rng = np.random.default_rng(seed)
    n = 6000
    o = rng.uniform(-140, 140, (n, 3))
    o[n // 2:] *= rng.uniform(3.0, 40.0, (n - n // 2, 1))
    tgt = rng.uniform(-110, 110, (n, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    gi, gt = sc.nearest_hits(o, d)
    oi, ot = oscene.nearest_hits(o, d)
    assert np.array_equal(gt.view(np.uint64), ot.view(np.uint64)), seed
    assert (gi == oi).mean() >= 0.9999, (seed, float((gi == oi).mean()))
    sc.close()
