"""Pins the CPU oracle to every known answer the reference's own #[test]s hold for the
trace path.  Each test names the reference test it restates."""
import math

import numpy as np
import pytest

INF = float("inf")


def flat_scene(rtx, build):
    b = rtx.SceneBuilder()
    build(b)
    return b.build()


# geometry/bbox.rs:250-271  it_intersects_with_a_ray
def test_bbox_intersects_with_a_ray(orc):
    mn, mx = (0, 0, 0), (1, 1, 1)
    assert orc.bbox_intersects(mn, mx, (0.5, 1.5, 0.5), (0, -1, 0))          # out of the box
    assert orc.bbox_intersects(mn, mx, (0.5, 0.5, 0.5), (0, -1, 0))          # in the box
    assert not orc.bbox_intersects(mn, mx, (0.5, 1.5, 0.5), (0, 1, 0))       # away from box
    assert not orc.bbox_intersects(mn, mx, (0.5, 1.5, 0.5), (1000, -1, 1000))  # to the side


# geometry/bbox.rs:321-340  it_checks_for_bbox_overlap (+ inclusive touch, octree.rs:124 relies on it)
def test_bbox_overlaps(orc):
    assert orc.bbox_overlaps((0, 0, 0), (1, 1, 1), (0.5, 0.5, 0.5), (1.5, 1.5, 1.5))
    assert not orc.bbox_overlaps((0, 0, 0), (1, 1, 1), (1.5, 1.5, 1.5), (2.5, 2.5, 2.5))
    assert orc.bbox_overlaps((0, 0, 0), (1, 1, 1), (1, 1, 1), (2, 2, 2))


TRI_V = [(-1, 0, 0), (1, 0, 0), (0, 1, 0)]
TRI_UV = [(0, 0), (1, 0), (0, 1)]


# geometry/prims/triangle.rs:178-216  it_intersects_and_interpolates
def test_triangle_intersects_and_interpolates(orc):
    n = [(-1, 0, 0), (1, 0, 0), (0, 1, 0)]
    hit = orc.triangle_intersects(TRI_V, n, TRI_UV, (0, 0.5, -1), (0, 0, 1), 0.0, 10.0)
    assert hit is not None
    assert hit["position"].tolist() == [0.0, 0.5, 0.0]
    assert hit["u"] == 0.25 and hit["v"] == 0.5
    assert hit["n"].tolist() == [0.0, 0.5, 0.0]          # interpolated, NOT normalised
    assert orc.triangle_intersects(TRI_V, n, TRI_UV, (0, 0.5, -1), (100, 100, 1), 0.0, 10.0) is None
    assert orc.triangle_intersects(TRI_V, n, TRI_UV, (0, 0.5, -1), (-100, -100, 1), 0.0, 10.0) is None
    assert orc.triangle_intersects(TRI_V, n, TRI_UV, (0, 0.5, -1), (0, 0, -1), 0.0, 10.0) is None


# geometry/prims/triangle.rs:219-237  it_intersects_only_in_tmin_tmax
def test_triangle_tmin_tmax(orc):
    n = [(0, 0, 0), (0, 0, 0), (1, 1, 1)]
    assert orc.triangle_intersects(TRI_V, n, TRI_UV, (0, 0.5, -1), (0, 0, 1), 1000.0, 10000.0) is None
    assert orc.triangle_intersects(TRI_V, n, TRI_UV, (0, 0.5, -1), (0, 0, 1), 0.0, 0.0001) is None


# geometry/prims/triangle.rs:39-42  get_auto_normals: un-normalised face cross product
def test_triangle_auto_normal(orc):
    assert orc.triangle_auto_normal(TRI_V).tolist() == [0.0, 0.0, 2.0]


# geometry/prims/sphere.rs:90-120  it_intersects
def test_sphere_intersects(orc):
    hit = orc.sphere_intersects((0, 0, 0), 1.0, (0, 0, -2), (0, 0, 1), 0.0, 10.0)
    assert hit["position"].tolist() == [0.0, 0.0, -1.0]
    assert hit["n"].tolist() == [0.0, 0.0, -1.0]
    assert orc.sphere_intersects((0, 0, 0), 1.0, (0, 0, -2), (100, 100, 0.1), 0.0, 10.0) is None
    assert orc.sphere_intersects((0, 0, 0), 1.0, (0, 0, -2), (-100, -100, 0.1), 0.0, 10.0) is None
    assert orc.sphere_intersects((0, 0, 0), 1.0, (0, 0, -2), (0, 0, -1), 0.0, 10.0) is None


# geometry/prims/sphere.rs:123-138  it_intersects_only_in_tmin_tmax
def test_sphere_tmin_tmax(orc):
    assert orc.sphere_intersects((0, 0, 0), 1.0, (0, 0, -2), (0, 0, 1), 1000.0, 10000.0) is None
    assert orc.sphere_intersects((0, 0, 0), 1.0, (0, 0, -2), (0, 0, 1), 0.0, 0.0001) is None


# sphere.rs:51-54: the root with the smaller |t| is returned even when it is the out-of-window one
def test_sphere_returns_smaller_abs_root(orc):
    # origin just inside the surface: t1 ~ 2 (in window), t2 ~ -1e-9 (out of window, smaller |t|)
    hit = orc.sphere_intersects((0, 0, 0), 1.0, (0, 0, -1 + 1e-9), (0, 0, 1), 1e-6, 10.0)
    assert hit is not None and hit["t"] < 0.0


# geometry/prims/plane.rs:76-98  it_intersects
def test_plane_intersects(orc):
    hit = orc.plane_intersects((0, 1, 0, 0), (0, 1, 0), (0, -1, 0), 0.0, 10.0)
    assert hit["position"].tolist() == [0.0, 0.0, 0.0]
    assert hit["n"].tolist() == [0.0, 1.0, 0.0]
    assert orc.plane_intersects((0, 1, 0, 0), (0, 1, 0), (1, 0, 1), 0.0, 10000.0) is None   # parallel
    assert orc.plane_intersects((0, 1, 0, 0), (0, 1, 0), (0, 1, 0), 0.0, 10.0) is None      # behind


# geometry/prims/plane.rs:101-112  it_intersects_only_in_tmin_tmax
def test_plane_tmin_tmax(orc):
    assert orc.plane_intersects((0, 1, 0, 0), (0, 1, 0), (0, -1, 0), 1000.0, 10000.0) is None
    assert orc.plane_intersects((0, 1, 0, 0), (0, 1, 0), (0, -1, 0), 0.0, 0.0001) is None


# raytracer/ray.rs:85-134  it_gets_the_nearest_hit
def test_nearest_hit_three_spheres(rtx, orc):
    def build(b):
        m = b.flat((1, 1, 1))
        b.sphere((0, 0, 0), 1.0, m)
        b.sphere((-1, 0, 0), 1.0, m)
        b.sphere((-2, 0, 0), 1.0, m)
        b.background = (1, 1, 1)
    sc = orc.Scene(flat_scene(rtx, build))
    pid, t, hit = sc.nearest_hits([(10, 0, 0)], [(-1, 0, 0)], want_hit=True)
    assert pid[0] == 0 and hit[0, 6] == 1.0            # intersection.position.x == 1.0
    pid, t = sc.nearest_hits([(10, 0, 0)], [(1, 0, 0)])
    assert pid[0] == -1 and t[0] == INF


# raytracer/renderer.rs:251-291  it_renders_the_background_of_an_empty_scene
def test_renders_background_of_empty_scene(rtx, orc):
    def build(b):
        b.background = (1.0, 0.0, 0.0)
    sc = orc.Scene(flat_scene(rtx, build))
    info = sc.octree_info()
    assert info["n_finite"] == 0 and info["depth"] == -2147483648   # ln(0) -> -inf -> i32::MIN
    cam = rtx.camera((0, 0, 0), (0, 1, 0), (0, 0, 1), 45.0, 32, 32)
    rgba, ids, cn = sc.render(cam, rtx.options(1, 1, 1, 1, 1), threads=2)
    assert (rgba[..., 0] == 255).all() and (rgba[..., 1] == 0).all() and (rgba[..., 2] == 0).all()
    assert (ids == -1).all() and cn.rays[0] == 32 * 32


# material/textures/imagetexture.rs:58-78  it_bilinearly_filters
def test_bilinear_filter(orc):
    img = np.zeros((2, 2, 4), dtype=np.uint8)
    img[..., 3] = 255
    img[0, 0, :3] = (255, 0, 0)   # surface[(0,0)]
    img[1, 0, :3] = (0, 255, 0)   # surface[(0,1)]: x=0, y=1
    img[0, 1, :3] = (0, 0, 255)   # surface[(1,0)]
    left = orc.image_color(img, 0.0, 0.5)
    assert left[:3].tolist() == [0.5, 0.5, 0.0]
    center = orc.image_color(img, 0.5, 0.5)
    assert center[:3].tolist() == [0.25, 0.25, 0.25]


# octree.rs:29 depth rule on the survey's sizes (a-5): N=1 -> 0, 12 -> 1
def test_octree_depth_rule(rtx, orc):
    def one(b):
        b.sphere((0, 0, 0), 1.0, b.flat())
    assert orc.Scene(flat_scene(rtx, one)).octree_info()["depth"] == 0

    def twelve(b):
        m = b.flat()
        for k in range(12):
            b.sphere((k * 3.0, 0, 0), 1.0, m)
    info = orc.Scene(flat_scene(rtx, twelve)).octree_info()
    assert info["depth"] == 1 and info["n_nodes"] == 9 and info["n_interior"] == 1
    for n, d in ((70000, 6), (114688, 7), (260000, 7)):
        assert round(1.2 * (math.log(n) / math.log(8.0))) == d


# vec3.rs tests :250-430 that touch the path: reflect / refract identities
def test_reflect_refract(orc):
    s = math.sqrt(0.5)
    assert np.allclose(orc.reflect((s, s, 0), (0, 1, 0)), (-s, s, 0), atol=1e-15)
    # straight-on refraction keeps direction (v points away from the surface)
    assert np.allclose(orc.refract((0, 1, 0), (0, 1, 0), 1.5, False), (0, -1, 0), atol=1e-15)
    # total internal reflection from inside a dense medium at a grazing angle -> None
    g = np.array([math.sin(1.4), math.cos(1.4), 0.0])
    assert orc.refract(g, (0, -1, 0), 1.5, True) is None


# compositor/colorrgba.rs:61-76 quantisation: round half away from zero, clamp
@pytest.mark.parametrize("c,q", [(0.0, 0), (1.0, 255), (0.5, 128), (0.5 / 255.0, 1), (0.4999 / 255, 0),
                                 (2.0, 255), (-1.0, 0), (float("nan"), 0)])
def test_quantize(orc, c, q):
    assert orc.quantize(c) == q


# Random123 known-answer vectors for Philox4x32-10 (the build's own RNG)
def test_philox_kat(orc):
    assert [hex(x) for x in orc.philox4x32_10([0, 0, 0, 0], [0, 0])] == \
        ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(x) for x in orc.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2)] == \
        ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(x) for x in orc.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                              [0xa4093822, 0x299f31d0])] == \
        ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_center_mode_draws_are_half(orc):
    assert orc.draw2(0, 123, 1, 2, 3).tolist() == [0.5, 0.5]
    d = orc.draw2(1, 123, 1, 2, 3)
    assert 0.0 <= d[0] < 1.0 and 0.0 <= d[1] < 1.0 and d[0] != d[1]
    assert orc.draw3(0, 123, 1, 2, 3).tolist() == [0.5, 0.5, 0.5]


def test_draw3_is_one_philox_block(orc):
    """(word + 0.5) / 2^32 of the first three words of philox(ctr = (stream, i, j, 'DRAW'), key)."""
    key = 0x0123456789ABCDEF
    words = orc.philox4x32_10([1, 2, 3, 0x44524157], [key & 0xFFFFFFFF, key >> 32])
    d = orc.draw3(1, key, 1, 2, 3)
    assert d.tolist() == [(int(w) + 0.5) / 4294967296.0 for w in words[:3]]
    assert all(0.0 < x < 1.0 for x in d)


def test_u8_over_255_without_the_division():
    """ColorRGBA channels are p / 255.0 (colorrgba.rs:116-124).  k_shade computes them as q = p * RN(1 / 255), q + fma(-255, q, p) * RN(1 / 255)
    (rtx_kernels.hip: u8_over_255): for every one of the 256 arguments that is the correctly rounded quotient, bit for bit.  The two
    FMAs are evaluated exactly here (rationals, one rounding each)."""
    from fractions import Fraction
    c = 1.0 / 255.0
    assert c.hex() == "0x1.0101010101010p-8"
    for p in range(256):
        q = float(p) * c
        r = float(Fraction(p) - 255 * Fraction(q))                    # fma(-255, q, p): one rounding
        got = float(Fraction(q) + Fraction(r) * Fraction(c))           # fma(r, c, q)
        assert got == p / 255.0 and got.hex() == (p / 255.0).hex(), p
