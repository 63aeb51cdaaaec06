"""Parity of the HIP path (through the C ABI) with the CPU oracle, on a real MI355X.

Bars (BASELINE.json north_star): in deterministic mode (rng_mode=center) primary-ray hit prim
ids bit-exact on >= 99.99 % of pixels and 8-bit RGB within +-1 on >= 99.9 % of pixels.
Closest-hit ids and t from rtx_nearest_hits must be bit-exact (integer/index work and IEEE f64
with the reference's operation order; only libm transcendentals may differ by ulps, and those
only feed shading)."""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


def _scene(rtx, which):
    if which == "sphere":
        return scenes.sphere_scene(rtx)
    if which == "cornell":
        return scenes.cornell_scene(rtx)
    if which == "mesh":
        return scenes.mesh_scene(rtx)
    if which == "mesh_transmissive":
        return scenes.mesh_scene(rtx, transmissive=True, point_light=True)
    if which == "mesh_big":
        return scenes.mesh_scene(rtx, n_u=96, n_v=64)
    if which == "cube":
        return scenes.cube_obj_scene(rtx)
    raise KeyError(which)


def _rays(cam_fn, orc, n, seed):
    """Random camera-like rays plus rays from random points inside the scene."""
    rng = np.random.RandomState(seed)
    cam = cam_fn(64, 64)
    o, d = [], []
    for _ in range(n // 2):
        r = orc.camera_get_ray(cam, rng.uniform(0, 63), rng.uniform(0, 63))
        o.append(r["origin"]); d.append(r["direction"])
    c = np.array(list(cam.look_at))
    ext = np.linalg.norm(np.array(list(cam.position)) - c)
    for _ in range(n - n // 2):
        p = c + rng.uniform(-0.5, 0.5, size=3) * ext
        v = rng.normal(size=3)
        o.append(p); d.append(v / np.linalg.norm(v))
    return np.array(o), np.array(d)


@pytest.mark.parametrize("which", ["sphere", "cornell", "mesh", "mesh_big", "cube"])
def test_nearest_hits_bit_exact(rtx, orc, which):
    flat, cam_fn = _scene(rtx, which)
    o, d = _rays(cam_fn, orc, 4096, seed=11)
    opid, ot = orc.Scene(flat).nearest_hits(o, d)
    sc = rtx.Scene(flat)
    pid, t = sc.nearest_hits(o, d)
    sc.close()
    assert (opid >= 0).sum() > 100, "test rays should hit something"
    assert np.array_equal(pid, opid)
    assert np.array_equal(t.view(np.uint64), ot.view(np.uint64))


@pytest.mark.parametrize("transmissive", [False, True])
@pytest.mark.parametrize("size", [(65, 64), (64, 64), (33, 40)])
def test_axis_aligned_scene_parity(rtx, orc, size, transmissive):
    """A scene on integer coordinates seen straight from above: with an odd width the centre column of primary rays, its
    reflections and its shadow feelers towards a light that shares their x have a direction component that is exactly zero, and
    start on octree planes: the rays for which the reference's own walk loses subtrees (NaN slab distances).  They go through
    k_closest_retry / k_shadow_retry, and the frame must still match the oracle: prim ids everywhere, colours to +-1.
    Rays through the shared edge of two triangles of a quad hit both at exactly the same t: the winner must be the reference's
    (the first it visits), also in the walk that deals the exact tests out over the wave (transmissive = True)."""
    flat, cam_fn = scenes.axis_aligned_scene(rtx, transmissive=transmissive)
    cam = cam_fn(*size)
    sc = rtx.Scene(flat)
    assert sc.info().all_opaque == (0 if transmissive else 1)
    opt = rtx.options(reflect_depth=3, refract_depth=3, shadow_samples=1, pixel_samples=1, rng_mode=0)
    rgba, ids, st = sc.render(cam, opt)
    rgba2, ids2, _ = sc.render(cam, rtx.options(reflect_depth=3, refract_depth=3, shadow_samples=16, pixel_samples=1, rng_mode=0))
    sc.close()
    osc = orc.Scene(flat)
    orgba, oids, _ = osc.render(cam, opt)
    orgba2, oids2, _ = osc.render(cam, rtx.options(reflect_depth=3, refract_depth=3, shadow_samples=16, pixel_samples=1, rng_mode=0))
    assert (ids >= 0).mean() > 0.5
    assert np.array_equal(ids, oids) and np.array_equal(ids2, oids2)
    assert (np.abs(rgba[..., :3].astype(int) - orgba[..., :3].astype(int)) <= 1).all()
    assert (np.abs(rgba2[..., :3].astype(int) - orgba2[..., :3].astype(int)) <= 1).all()      # 16 tries: the cooperative kernel


def _cell_points(rtx, flat, rng, n):
    """Points on faces / edges / corners of octree cells: each coordinate is a cell's min, max, mid or free inside it."""
    depth, bbox, shape, entries = rtx.octree_dump(flat)
    b = bbox[rng.randint(0, bbox.shape[0], size=n)]
    lo, hi = b[:, :3], b[:, 3:]
    k = rng.randint(0, 4, size=(n, 3))
    u = rng.uniform(size=(n, 3))
    return np.where(k == 0, lo, np.where(k == 1, hi, np.where(k == 2, 0.5 * (lo + hi), lo + u * (hi - lo))))


@pytest.mark.parametrize("which", ["cornell", "mesh", "mesh_big", "cube"])
def test_nearest_hits_on_degenerate_rays(rtx, orc, which):
    """Rays built to lie IN face planes of octree cells: origins on faces, edges and corners of cells, directions exactly
    parallel to an axis (0 * inf = NaN slab distances), along cell faces, or through other cell corners.  There the
    reference's own walk loses subtrees: BBox::intersects rejects a box whose near distance is a NaN (bbox.rs:124-164),
    and a ray within an ulp of a cell face can fail the f64 test of the cell that holds the prim it hits.
      * the all-f64 restatement (RTX_FLAG_REFERENCE_ORDER) reproduces the oracle bit for bit, prim and t;
      * the renderer's walk (k_closest + k_closest_retry) sends every ray with a direction component below 1e-9 through that
        same all-f64 walk and everything else through the conservative near-to-far walk with the exact entry-bbox and prim
        tests: t AND prim must equal the oracle's on EVERY ray.  The rays aim at cell corners, which in these scenes are mesh
        vertices and edges: hundreds of them hit two or more prims at a bit-identical t, where ray.rs:52-60 keeps the prim its
        octree walk visits first; the ordered walks notice the tie and hand the ray to the all-f64 walk in the reference's order."""
    flat, cam_fn = _scene(rtx, which)
    rng = np.random.RandomState(23)
    n = 6000
    o = _cell_points(rtx, flat, rng, n)
    t = _cell_points(rtx, flat, rng, n)
    d = t - o
    par = rng.uniform(size=n) < 0.35
    axis = np.eye(3)[rng.randint(0, 3, size=n)] * rng.choice([-1.0, 1.0], size=(n, 1))
    d = np.where(par[:, None], axis, d)
    keep = np.linalg.norm(d, axis=1) > 0
    o, d = o[keep], d[keep]
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    opid, ot = orc.Scene(flat).nearest_hits(o, d)
    sc = rtx.Scene(flat)
    pid, tt = sc.nearest_hits(o, d)
    rpid, rt = sc.nearest_hits(o, d, flags=rtx.abi.FLAG_REFERENCE_ORDER)
    sc2_pid_t = sc.nearest_hits(o, d, flags=rtx.abi.FLAG_OCTREE_WALK)
    sc.close()
    assert (opid >= 0).sum() > 100
    assert np.array_equal(rpid, opid) and np.array_equal(rt.view(np.uint64), ot.view(np.uint64))
    assert np.array_equal(tt.view(np.uint64), ot.view(np.uint64))
    assert np.array_equal(pid, opid), "%d rays resolved to another prim than the reference's" % int((pid != opid).sum())
    wpid, wt = sc2_pid_t
    assert np.array_equal(wt.view(np.uint64), ot.view(np.uint64)) and np.array_equal(wpid, opid)   # the octree walk (RTX_FLAG_OCTREE_WALK) too


@pytest.mark.parametrize("which", ["cornell", "mesh", "mesh_transmissive", "cube"])
def test_shadow_products_match(rtx, orc, which):
    flat, cam_fn = _scene(rtx, which)
    o, d = _rays(cam_fn, orc, 2048, seed=5)
    osc = orc.Scene(flat)
    pid, t, hit = osc.nearest_hits(o, d, want_hit=True)
    keep = pid >= 0
    origins = hit[keep, 6:9]                      # shadow feelers start ON surfaces, un-offset (renderer.rs:204)
    light = np.array(list(flat.lights[0].position))
    targets = np.broadcast_to(light, origins.shape).copy()
    want = osc.shadow_products(origins, targets)
    sc = rtx.Scene(flat)
    got = sc.shadow_products(origins, targets)
    sc.close()
    assert origins.shape[0] > 100
    # products of identical factors in a different order: equal up to the last ulp
    assert np.allclose(got, want, rtol=1e-15, atol=0.0)
    assert np.array_equal(got == 0.0, want == 0.0)


@pytest.mark.parametrize("which", ["mesh_transmissive", "cornell", "cube"])
def test_shadow_product_walks_agree_on_adversarial_feelers(rtx, orc, which):
    """The renderer's product walk decides "reached or not" once per leaf, from the leaf's own box (nested boxes + monotone slab
    test), and skips known misses through a mailbox; the all-f64 walk tests every child box like the reference.  Both must
    give the same product ray by ray, also where the f32 classification cannot decide: feelers that start on faces, edges
    and corners of octree cells, run along cell faces or exactly parallel to an axis (0 * inf slab distances: these take the
    all-f64 path), or graze cell corners; and both must agree with the oracle."""
    flat, cam_fn = _scene(rtx, which)
    depth, bbox, shape, entries = rtx.octree_dump(flat)
    rng = np.random.RandomState(17)
    boxes = bbox[rng.randint(0, bbox.shape[0], size=1500)]
    lo, hi = boxes[:, :3], boxes[:, 3:]
    def on_cell(n):   # points on cell faces / edges / corners: each coordinate is min, max, mid or free
        k = rng.randint(0, 4, size=(n, 3))
        u = rng.uniform(size=(n, 3))
        b = boxes[rng.randint(0, boxes.shape[0], size=n)]
        l_, h_ = b[:, :3], b[:, 3:]
        return np.where(k == 0, l_, np.where(k == 1, h_, np.where(k == 2, 0.5 * (l_ + h_), l_ + u * (h_ - l_))))
    n = 6000
    o = on_cell(n)
    t = on_cell(n)
    ax = rng.randint(0, 3, size=n)
    par = rng.uniform(size=n) < 0.3          # exactly axis-parallel: the target differs in one coordinate only
    step = (rng.uniform(0.5, 3.0, size=n) * (hi - lo).max()) * rng.choice([-1.0, 1.0], size=n)
    t_par = o.copy()
    t_par[np.arange(n), ax] += step
    t = np.where(par[:, None], t_par, t)
    keep = np.linalg.norm(t - o, axis=1) > 0
    o, t = o[keep], t[keep]
    sc = rtx.Scene(flat)
    fast = sc.shadow_products(o, t)
    exact = sc.shadow_products(o, t, flags=rtx.abi.FLAG_REFERENCE_ORDER)
    sc.close()
    want = orc.Scene(flat).shadow_products(o, t)
    assert o.shape[0] > 4000
    assert np.array_equal(fast == 0.0, exact == 0.0) and np.allclose(fast, exact, rtol=1e-15, atol=0.0)
    assert np.array_equal(exact == 0.0, want == 0.0) and np.allclose(exact, want, rtol=1e-15, atol=0.0)
    assert (fast != 1.0).any() and (fast == 1.0).any()


def _compare(rgba, ids, orgba, oids, id_bar=0.9999, rgb_bar=0.999):
    id_ok = (ids == oids).mean()
    diff = np.abs(rgba[..., :3].astype(np.int32) - orgba[..., :3].astype(np.int32)).max(axis=-1)
    rgb_ok = (diff <= 1).mean()
    assert id_ok >= id_bar, "prim-id parity %.6f" % id_ok
    assert rgb_ok >= rgb_bar, "rgb +-1 parity %.6f (max diff %d)" % (rgb_ok, diff.max())
    assert (rgba[..., 3] == 255).all()
    return id_ok, rgb_ok


DET = dict(reflect_depth=4, refract_depth=4, shadow_samples=1, gloss_samples=1, pixel_samples=1, rng_mode=0)


@pytest.mark.parametrize("which,size", [("sphere", (64, 64)), ("cornell", (128, 128)), ("mesh", (128, 96)),
                                        ("mesh_transmissive", (96, 64)), ("cube", (64, 64)), ("mesh_big", (160, 120))])
def test_deterministic_render_parity(rtx, orc, which, size):
    flat, cam_fn = _scene(rtx, which)
    cam = cam_fn(*size)
    opt = rtx.options(**DET)
    orgba, oids, ocn = orc.Scene(flat).render(cam, opt)
    sc = rtx.Scene(flat)
    rgba, ids, st = sc.render(cam, opt)
    sc.close()
    _compare(rgba, ids, orgba, oids)
    # ray accounting is identical on both sides: 1 per get_nearest_hit, 1 per shadow try
    assert st.rays_primary == ocn.rays[0] == size[0] * size[1]
    for got, want in ((st.rays_reflect, ocn.rays[1]), (st.rays_gloss, ocn.rays[2]), (st.rays_refract, ocn.rays[3]),
                      (st.rays_shadow, ocn.rays[4])):
        assert abs(int(got) - int(want)) <= max(2, 1e-3 * want), (got, want)


def test_deep_refraction_recursion(rtx, orc):
    """cornell's refractive sphere (ior 3, k_sg = k_tg = 1) with depth 8/8: a branching ray tree."""
    flat, cam_fn = scenes.cornell_scene(rtx)
    cam = cam_fn(96, 96)
    opt = rtx.options(reflect_depth=8, refract_depth=8, shadow_samples=1, gloss_samples=1, pixel_samples=1, rng_mode=0)
    orgba, oids, ocn = orc.Scene(flat).render(cam, opt)
    sc = rtx.Scene(flat)
    rgba, ids, st = sc.render(cam, opt)
    # a small batch forces many wavefront steps and exercises the segment stack of the ray pool
    opt2 = rtx.options(reflect_depth=8, refract_depth=8, shadow_samples=1, gloss_samples=1, pixel_samples=1, rng_mode=0,
                       batch_rays=1024)
    rgba2, ids2, st2 = sc.render(cam, opt2)
    sc.close()
    _compare(rgba, ids, orgba, oids)
    _compare(rgba2, ids2, orgba, oids)
    assert st.rays_refract > 0 and abs(int(st.rays_refract) - int(ocn.rays[3])) <= 1e-3 * ocn.rays[3] + 2
    assert st2.rays_total == st.rays_total


def test_stochastic_render_same_rng(rtx, orc):
    """Philox mode: oracle and kernels implement the same counter-based RNG spec independently,
    so supersampled, soft-shadowed, glossy images agree to the same +-1 bar."""
    flat, cam_fn = scenes.cornell_scene(rtx, glossy=True)
    cam = cam_fn(64, 64)
    opt = rtx.options(reflect_depth=3, refract_depth=3, shadow_samples=4, gloss_samples=2, pixel_samples=2, rng_mode=1, seed=7)
    orgba, oids, ocn = orc.Scene(flat).render(cam, opt)
    sc = rtx.Scene(flat)
    rgba, ids, st = sc.render(cam, opt)
    sc.close()
    _compare(rgba, ids, orgba, oids, rgb_bar=0.995)
    assert st.rays_gloss > 0 and st.rays_primary == 64 * 64 * 4
    a, b = rgba[..., :3].astype(np.float64) / 255.0, orgba[..., :3].astype(np.float64) / 255.0
    assert np.sqrt(((a - b) ** 2).mean()) < 0.005     # RMSE < 0.5 %


def test_stochastic_converges_to_high_spp_reference(rtx, orc):
    """north_star: converged-image RMSE against a high-spp reference render below 0.5 %."""
    flat, cam_fn = scenes.mesh_scene(rtx, n_u=16, n_v=12)
    cam = cam_fn(48, 32)
    ref_opt = rtx.options(reflect_depth=2, refract_depth=2, shadow_samples=16, pixel_samples=4, rng_mode=1, seed=1)
    ref = orc.Scene(flat).render_linear(cam, ref_opt)
    sc = rtx.Scene(flat)
    opt = rtx.options(reflect_depth=2, refract_depth=2, shadow_samples=16, pixel_samples=4, rng_mode=1, seed=99)
    rgba, _, _ = sc.render(cam, opt, want_ids=False)
    sc.close()
    got = rgba[..., :3].astype(np.float64) / 255.0
    # different seeds: only the distribution matches; compare means (noise-free up to 1/sqrt(16*16) of the soft edges)
    assert abs(got.mean() - np.clip(ref, 0, 1).mean()) < 0.005


def test_empty_scene_background(rtx):
    """raytracer/renderer.rs:251-291 it_renders_the_background_of_an_empty_scene, on the GPU."""
    b = rtx.SceneBuilder()
    b.background = (1.0, 0.0, 0.0)
    sc = rtx.Scene(b.build())
    cam = rtx.camera((0, 0, 0), (0, 1, 0), (0, 0, 1), 45.0, 32, 32)
    rgba, ids, st = sc.render(cam, rtx.options(1, 1, 1, 1, 1))
    sc.close()
    assert (rgba[..., 0] == 255).all() and (rgba[..., 1] == 0).all() and (rgba[..., 2] == 0).all()
    assert (ids == -1).all() and st.rays_primary == 1024 and st.rays_shadow == 0


def test_zero_depth_is_black(rtx):
    """renderer.rs:106: reflect_depth or refract_depth of 0 returns zero before any ray."""
    flat, cam_fn = scenes.sphere_scene(rtx)
    sc = rtx.Scene(flat)
    cam = cam_fn(32, 32)
    _, ids_hit, _ = sc.render(cam, rtx.options(reflect_depth=2, refract_depth=2))      # (leaves real ids in the handle's buffers)
    assert (ids_hit >= 0).any()
    rgba, ids, st = sc.render(cam, rtx.options(reflect_depth=0, refract_depth=4))
    assert (rgba[..., :3] == 0).all() and st.rays_total == 0
    assert (ids == -1).all()              # no primary ray is traced: every pixel is a miss, not the last frame's ids
    # ... also for the blocks of one shard only (the others stay as they were)
    sc.render(cam, rtx.options(reflect_depth=2, refract_depth=2))
    _, ids1, _ = sc.render(cam, rtx.options(reflect_depth=3, refract_depth=0, shard_index=1, shard_count=3))   # (Scene.render pre-fills -2)
    sc.close()
    assert (ids1[8:16] == -1).all() and (ids1[:8] == -2).all() and (ids1[16:] == -2).all()


def test_shards_tile_the_frame(rtx):
    """8-row blocks dealt round-robin; shards are disjoint and partition-invariant."""
    flat, cam_fn = scenes.mesh_scene(rtx)
    cam = cam_fn(96, 64)
    sc = rtx.Scene(flat)
    kw = dict(reflect_depth=3, refract_depth=3, shadow_samples=2, pixel_samples=2, rng_mode=1, seed=3)
    full, ids_full, st = sc.render(cam, rtx.options(**kw))
    out = np.zeros_like(full)
    rays = 0
    for k in range(3):
        _, _, stk = sc.render(cam, rtx.options(shard_index=k, shard_count=3, **kw), want_ids=False, out=out)
        rays += stk.rays_total
    sc.close()
    assert np.array_equal(out, full)
    assert rays == st.rays_total


def test_compact_output_holds_the_shards_blocks(rtx):
    """RTX_FLAG_COMPACT_OUTPUT: the call's own 8-row blocks, packed, equal the same rows of the full frame; the
    StillGather de-interleave of the three shards' compact buffers is the full frame (ids too); the host-buffer call rejects it."""
    import torch
    flat, cam_fn = scenes.mesh_scene(rtx)
    W, H = 96, 72                                     # 9 blocks over 3 and over 4 shards (ragged: 3,2,2,2)
    cam = cam_fn(W, H)
    sc = rtx.Scene(flat)
    kw = dict(reflect_depth=3, refract_depth=3, shadow_samples=2, pixel_samples=2, rng_mode=1, seed=3)
    full, ids_full, st = sc.render(cam, rtx.options(**kw))
    stream = torch.cuda.current_stream().cuda_stream
    for world in (3, 4):
        final = None
        for k in range(world):
            g = rtx.sharding.StillGather(W, H, k, world, torch.device("cuda", 0))
            g.target.fill_(77)
            ids = torch.full((g.max_blocks * 8, W), -7, dtype=torch.int32, device="cuda")
            sc.render_device(cam, rtx.options(shard_index=k, shard_count=world, flags=rtx.abi.FLAG_COMPACT_OUTPUT, **kw),
                             g.target.data_ptr(), ids.data_ptr(), stream)
            torch.cuda.synchronize()
            got, got_ids = g.target.cpu().numpy(), ids.cpu().numpy()
            mine = rtx.sharding.blocks_of_rank(H, k, world)
            for j, b in enumerate(mine):
                assert np.array_equal(got[j * 8:j * 8 + 8], full[b * 8:b * 8 + 8])
                assert np.array_equal(got_ids[j * 8:j * 8 + 8], ids_full[b * 8:b * 8 + 8])
            assert (got[len(mine) * 8:] == 77).all() and (got_ids[len(mine) * 8:] == -7).all()   # nothing written past the shard's blocks
    with pytest.raises(Exception):
        sc.render(cam, rtx.options(flags=rtx.abi.FLAG_COMPACT_OUTPUT, **kw))
    sc.close()


def test_reference_order_flag_agrees(rtx, orc):
    """RTX_FLAG_REFERENCE_ORDER walks the octree exactly as the reference's unordered DFS does;
    both orders give the same image (ties between different prims are vanishingly rare)."""
    flat, cam_fn = scenes.mesh_scene(rtx)
    cam = cam_fn(64, 48)
    sc = rtx.Scene(flat)
    a, ia, _ = sc.render(cam, rtx.options(**DET))
    b, ib, _ = sc.render(cam, rtx.options(flags=rtx.abi.FLAG_REFERENCE_ORDER, **DET))
    sc.close()
    orgba, oids, _ = orc.Scene(flat).render(cam, rtx.options(**DET))
    assert np.array_equal(ib, oids)
    assert (ia == ib).mean() >= 0.9999
    assert (np.abs(a.astype(int) - b.astype(int)) <= 1).mean() >= 0.999


def test_octree_walk_flag_agrees(rtx, orc):
    """RTX_FLAG_OCTREE_WALK: the per-lane closest-hit and any-hit walks over the reference's octree (f32 twins) instead of the 4-wide SAH
    hierarchy.  Both reach a superset of the prims a ray truly hits and decide each with the same exact f64 tests: the same t bit for
    bit, the same prim but for exact-t ties, the same frames (1 shadow try: per-lane any-hit walk; 4: what the cooperative walk leaves)."""
    flat, cam_fn = scenes.mesh_scene(rtx, n_u=48, n_v=32)
    sc = rtx.Scene(flat)
    assert sc.info().wide_nodes > 0
    rng = np.random.default_rng(5)
    n = 20000
    o = np.tile(np.array([[0.0, -180.0, 60.0]]), (n, 1)) + rng.normal(0, 40.0, (n, 3))
    tgt = np.array([[12.0, 58.0, 52.0]]) + rng.normal(0, 45.0, (n, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ia, ta = sc.nearest_hits(o, d)
    ib, tb = sc.nearest_hits(o, d, rtx.abi.FLAG_OCTREE_WALK)
    io, to = orc.Scene(flat).nearest_hits(o, d)
    assert np.array_equal(ta.view(np.uint64), tb.view(np.uint64)) and np.array_equal(ta.view(np.uint64), to.view(np.uint64))
    assert (ia >= 0).mean() > 0.3
    assert (ia == ib).mean() >= 0.9999 and (ia == io).mean() >= 0.9999
    cam = cam_fn(160, 96)
    for kw in (DET, dict(reflect_depth=3, refract_depth=3, shadow_samples=4, gloss_samples=1, pixel_samples=2, rng_mode=1, seed=2)):
        a, ida, sta = sc.render(cam, rtx.options(**kw))
        b, idb, stb = sc.render(cam, rtx.options(flags=rtx.abi.FLAG_OCTREE_WALK, **kw))
        assert sta.rays_total == stb.rays_total
        assert (ida == idb).mean() >= 0.9999
        assert (np.abs(a.astype(int) - b.astype(int)) <= 1).mean() >= 0.999
    sc.close()


def test_bad_arguments(rtx):
    flat, cam_fn = scenes.sphere_scene(rtx)
    sc = rtx.Scene(flat)
    with pytest.raises(rtx.abi.RtxError):
        sc.render(cam_fn(32, 30), rtx.options())            # height not a multiple of 8
    with pytest.raises(rtx.abi.RtxError):
        sc.render(cam_fn(32, 32), rtx.options(shard_index=2, shard_count=2))
    sc.close()


def test_pool_exhaustion_is_an_error_and_leaves_the_handle_usable(rtx):
    """A ray pool too small for the recursion fails with RTX_ERR_QUEUE_OVERFLOW (no path is dropped silently), with the
    shadow launches that were already running on the library's streams drained; the same handle then renders the
    frame correctly with the default pool."""
    flat, cam_fn = scenes.mesh_scene(rtx, n_u=48, n_v=32, transmissive=True)
    cam = cam_fn(160, 96)
    sc = rtx.Scene(flat)
    kw = dict(reflect_depth=8, refract_depth=8, shadow_samples=2, pixel_samples=2, rng_mode=1, seed=3)
    good, ids, st = sc.render(cam, rtx.options(**kw))
    with pytest.raises(rtx.abi.RtxError) as err:
        sc.render(cam, rtx.options(batch_rays=16384, pool_rays=16384, **kw))
    assert err.value.status == rtx.abi.RTX_ERR_QUEUE_OVERFLOW
    again, ids2, st2 = sc.render(cam, rtx.options(**kw))
    sc.close()
    assert np.array_equal(ids, ids2) and st.rays_total == st2.rays_total
    assert (np.abs(good.astype(int) - again.astype(int)) <= 1).all() and (good != again).mean() < 1e-3


@pytest.mark.parametrize("depth", [2, 3, 5])
@pytest.mark.parametrize("glass", [False, True])
def test_every_ray_spawns_at_every_level(rtx, orc, depth, glass):
    """A camera inside a closed box of six mirror planes (and, `glass`, transmissive ones: two children per hit): EVERY ray hits and
    spawns its full fan-out at EVERY level, so each generation fills its share of the automatically sized ray pool exactly.  The
    last generation spawns nothing and must not be refused for want of room for children (found by tools/exp/fuzz_campaign.py: the
    library returned RTX_ERR_QUEUE_OVERFLOW with the default pool).  Ray counts: the full tree."""
    b = rtx.SceneBuilder()
    b.point_light((0.3, 0.4, -0.2), (0.9, 0.9, 0.9))
    m = b.cook_torrance(k_a=0.05, k_d=0.5, k_s=0.4, k_sg=0.8, k_tg=0.7 if glass else 0.0, gauss_constant=5.0, roughness=0.1, ior=1.4,
                        diffuse=(0.8, 0.7, 0.6), transmission=(0.6, 0.7, 0.8) if glass else (0.0, 0.0, 0.0))
    for a in range(3):
        for sgn in (-1.0, 1.0):
            n = [0.0, 0.0, 0.0]
            n[a] = sgn
            b.plane(n[0], n[1], n[2], 5.0, m)            # the planes x, y, z = -+5
    b.sphere((2.0, 0.5, 2.5), 0.8, m)
    b.background = (0.1, 0.2, 0.3)
    flat = b.build()
    cam = rtx.camera((0.5, 0.2, 0.1), (3.0, 1.0, 4.0), (0, 1, 0), 40.0, 64, 40)
    opt = rtx.options(reflect_depth=depth, refract_depth=depth, shadow_samples=1, pixel_samples=2, rng_mode=0)
    sc = rtx.Scene(flat)
    rgba, ids, st = sc.render(cam, opt)                  # default pool
    sc.close()
    orgba, oids, ocn = orc.Scene(flat).render(cam, opt)
    assert np.array_equal(ids, oids) and (ids >= 0).all()
    assert st.rays_total == ocn.rays_total
    assert (np.abs(rgba[..., :3].astype(int) - orgba[..., :3].astype(int)) <= 1).all()
    if not glass:
        assert st.rays_reflect == (depth - 1) * st.rays_primary     # one mirror child per ray and level


@pytest.mark.parametrize("tries", [1, 2, 4, 8, 16, 32, 64, 48, 3])
def test_cooperative_shadows_equal_per_lane_shadows(rtx, tries):
    """The cooperative 16-lane box-list traversal and the per-lane octree walk are two exact any-hit
    searches over the same primitives: for every group size the images must be identical.  16, 32, 48 and 64 tries:
    one hit point per group; 4 and 8: groups of neighbouring hit points, the compact ones walked cooperatively
    and the rest left to the per-lane kernel; 1, 2 and 3 (not a power of two): per-lane only."""
    flat, cam_fn = scenes.mesh_scene(rtx, n_u=48, n_v=32)
    cam = cam_fn(160, 96)
    sc = rtx.Scene(flat)
    kw = dict(reflect_depth=3, refract_depth=3, shadow_samples=tries, pixel_samples=2, rng_mode=1, seed=5)
    a, ia, sa = sc.render(cam, rtx.options(**kw))
    b, ib, sb = sc.render(cam, rtx.options(flags=rtx.abi.FLAG_NO_COOP, **kw))
    sc.close()
    assert np.array_equal(ia, ib)
    assert (np.abs(a.astype(int) - b.astype(int)) <= 1).all()      # identical up to the order of f64 atomic adds
    assert (a != b).mean() < 1e-3
    assert sa.rays_total == sb.rays_total


@pytest.mark.parametrize("which", ["mesh", "mesh_transmissive", "kitchen"])
def test_overlapped_shadow_launches_equal_serial_launches(rtx, which):
    """By default the shadow launches of a wavefront step run on the library's own stream and overlap the next
    step's closest-hit launch (two alternating shadow queues); RTX_FLAG_SERIAL keeps every launch on the caller's
    stream.  Same rays, same sums: the images may differ only by the order of the f64 atomic adds.  A small batch
    makes many steps, so the queues alternate often."""
    if which == "kitchen":
        flat, cam_fn = scenes.kitchen_sink_scene(rtx)
    else:
        flat, cam_fn = scenes.mesh_scene(rtx, n_u=48, n_v=32, transmissive=(which == "mesh_transmissive"))
    cam = cam_fn(160, 96)
    sc = rtx.Scene(flat)
    kw = dict(reflect_depth=4, refract_depth=4, shadow_samples=4, pixel_samples=2, rng_mode=1, seed=7, batch_rays=4096)
    a, ia, sa = sc.render(cam, rtx.options(**kw))
    b, ib, sb = sc.render(cam, rtx.options(flags=rtx.abi.FLAG_SERIAL, **kw))
    c, ic, sc_ = sc.render(cam, rtx.options(**kw))     # and again: the workspace (events, queues) is reused
    sc.close()
    assert sa.launches_closest > 8
    assert np.array_equal(ia, ib) and np.array_equal(ia, ic)
    assert (np.abs(a.astype(int) - b.astype(int)) <= 1).all() and (a != b).mean() < 1e-3
    assert (np.abs(a.astype(int) - c.astype(int)) <= 1).all() and (a != c).mean() < 1e-3
    assert sa.rays_total == sb.rays_total == sc_.rays_total


@pytest.mark.parametrize("name,fov,opt", [
    ("bunny", 45.0, dict(reflect_depth=4, refract_depth=4, shadow_samples=1, pixel_samples=2)),
    ("sponza", 67.5, dict(reflect_depth=3, refract_depth=3, shadow_samples=16, pixel_samples=4)),   # config 3 as stated: 16 spp
    ("sibenik", 45.0, dict(reflect_depth=3, refract_depth=3, shadow_samples=4, pixel_samples=1)),
])
def test_baseline_scenes_full_size_properties(rtx, orc, name, fov, opt):
    """BASELINE.json's scenes at 1920x1080 with their full-size stand-in meshes.  The oracle takes
    minutes per frame here, so parity is checked (a) against the oracle on two 8-row tile blocks and
    (b) through size-independent properties: the two independent shadow traversals agree on the
    whole frame, and interleaved shards reassemble the unsharded frame."""
    named = rtx.host.NamedScene(name)
    cam = named.camera(1920, 1080, fov)
    sc = rtx.Scene(named)
    kw = dict(rng_mode=1, seed=0, **opt)
    full, ids, st = sc.render(cam, rtx.options(**kw))
    alt, ids2, _ = sc.render(cam, rtx.options(flags=rtx.abi.FLAG_NO_COOP, **kw))
    assert np.array_equal(ids, ids2)
    assert (np.abs(full.astype(int) - alt.astype(int)) <= 1).all() and (full != alt).mean() < 1e-3
    out = np.zeros_like(full)
    for k in range(2):
        sc.render(cam, rtx.options(shard_index=k, shard_count=2, **kw), want_ids=False, out=out)
    sc.close()
    assert (np.abs(out.astype(int) - full.astype(int)) <= 1).all() and (out != full).mean() < 1e-3
    osc = orc.Scene(named)
    for b in (40, 100):
        o_rgba, o_ids, _ = osc.render(cam, rtx.options(**kw), threads=16, rows=(b * 8, b * 8 + 8))
        rows = slice(b * 8, b * 8 + 8)
        assert (ids[rows] == o_ids[rows]).mean() >= 0.9999
        assert (np.abs(full[rows, :, :3].astype(int) - o_rgba[rows, :, :3].astype(int)).max(axis=-1) <= 1).mean() >= 0.999


def test_heptoroid_refractive_deep_recursion(rtx, orc):
    """BASELINE config 4's material (k_sg = k_tg = 1, transmission 0.8): the branching ray tree and the
    multiset shadow product against the oracle.  The tree grows ~2x per level, so the oracle-checked
    case uses depth 7/7 at reduced size; depth 16/16 is exercised by bench.py --workload heptoroid."""
    named = rtx.host.NamedScene("heptoroid-refractive", detail=0.25)
    cam = named.camera(96, 64, 45.0)
    opt = rtx.options(reflect_depth=7, refract_depth=7, shadow_samples=1, pixel_samples=1, rng_mode=0)
    orgba, oids, ocn = orc.Scene(named).render(cam, opt, threads=16)
    sc = rtx.Scene(named)
    rgba, ids, st = sc.render(cam, opt)
    sc.close()
    _compare(rgba, ids, orgba, oids)
    assert st.rays_refract > 10000 and abs(int(st.rays_total) - int(ocn.rays_total)) <= 1e-3 * ocn.rays_total


@pytest.mark.parametrize("lights,size", [("mixed", (104, 72)), ("points", (72, 56)), ("none", (40, 24))])
def test_kitchen_sink_parity(rtx, orc, lights, size):
    """UV / image / checker textures on both BRDFs, FlatMaterial, transmissive + mirror spheres, point and sphere
    lights (also none at all, renderer.rs:114 folds over an empty list), widths that are not multiples of the
    8-pixel wave tile or of the reference's 128-pixel tile (edge tiles over-render and merge clips, surface.rs:84-85)."""
    flat, cam_fn = scenes.kitchen_sink_scene(rtx, lights=lights)
    cam = cam_fn(*size)
    opt = rtx.options(**DET)
    orgba, oids, ocn = orc.Scene(flat).render(cam, opt)
    sc = rtx.Scene(flat)
    rgba, ids, st = sc.render(cam, opt)
    sopt = rtx.options(reflect_depth=3, refract_depth=3, shadow_samples=4, gloss_samples=1, pixel_samples=2, rng_mode=1, seed=11)
    srgba, sids, sst = sc.render(cam, sopt)
    sc.close()
    _compare(rgba, ids, orgba, oids)
    assert st.rays_total == ocn.rays_total or abs(int(st.rays_total) - int(ocn.rays_total)) <= 1e-3 * ocn.rays_total + 2
    if lights == "none":
        assert st.rays_shadow == 0
    o2, oi2, _ = orc.Scene(flat).render(cam, sopt)
    _compare(srgba, sids, o2, oi2, rgb_bar=0.995)


@pytest.mark.parametrize("name,fov", [("cow", 45.0), ("easing", 30.0), ("fresnel", 30.0), ("lucy", 45.0), ("tachikoma", 45.0),
                                      ("teapot", 45.0), ("heptoroid-shiny", 45.0), ("heptoroid-white", 45.0)])
def test_registry_scenes_parity(rtx, orc, name, fov):
    """The scenes of my_scene/mod.rs:28-44 that are not BASELINE configs, at reduced tessellation, against the oracle."""
    named = rtx.host.NamedScene(name, detail=0.25)
    cam = named.camera(96, 64, fov)
    opt = rtx.options(**DET)
    orgba, oids, ocn = orc.Scene(named).render(cam, opt, threads=16)
    sc = rtx.Scene(named)
    rgba, ids, st = sc.render(cam, opt)
    sc.close()
    _compare(rgba, ids, orgba, oids)
    assert abs(int(st.rays_total) - int(ocn.rays_total)) <= 1e-3 * ocn.rays_total + 2


@pytest.mark.parametrize("which", ["mesh_transmissive", "cornell", "kitchen"])
def test_fast_shadow_product_equals_exact_walk(rtx, orc, which):
    """Scenes with transmissive materials: the f32-classified walk (traverse_product_fast) must reach exactly the
    reference's candidate multiset, i.e. give the images of the all-f64 restatement (RTX_FLAG_REFERENCE_ORDER),
    for soft coloured shadows through up to several transmissive layers, and both must match the oracle."""
    if which == "kitchen":
        flat, cam_fn = scenes.kitchen_sink_scene(rtx)
    elif which == "cornell":
        flat, cam_fn = scenes.cornell_scene(rtx)
    else:
        flat, cam_fn = scenes.mesh_scene(rtx, n_u=48, n_v=32, transmissive=True)
    cam = cam_fn(144, 96)
    kw = dict(reflect_depth=4, refract_depth=4, shadow_samples=8, pixel_samples=2, rng_mode=1, seed=21)
    sc = rtx.Scene(flat)
    assert sc.info().all_opaque == 0
    a, ia, sa = sc.render(cam, rtx.options(**kw))
    b, ib, sb = sc.render(cam, rtx.options(flags=rtx.abi.FLAG_REFERENCE_ORDER, **kw))
    sc.close()
    assert (ia == ib).mean() >= 0.9999
    assert (np.abs(a.astype(int) - b.astype(int)) <= 1).all() and (a != b).mean() < 1e-3   # up to the order of f64 atomic adds
    assert sa.rays_total == sb.rays_total
    orgba, oids, _ = orc.Scene(flat).render(cam, rtx.options(**kw), threads=16)
    _compare(a, ia, orgba, oids, rgb_bar=0.995)
