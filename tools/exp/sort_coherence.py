#!/usr/bin/env python3
"""Would sorting a launch's shadow feelers by the cell of their origin pay on heptoroid (k_shadow<MODE_PRODUCT_BVH> runs at 0.37 of its
lanes)?  Feelers of a synthetic ray tree (camera grid, then two children per hit for LEVELS levels: one that goes nearly straight on, one
mirrored about a random normal; children adjacent, as in the ray pool) to the scene's light, through rtx_shadow_products (the same walk as
the frame's kernel), in three orders: the tree's own, random, sorted by a 30-bit Morton code of the origin.  Run under
`rocprofv3 --kernel-trace` and read the k_shadow_products launches in order (tools/rocpd_launches.py): 2 launches per order.
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def unit(v):
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def morton30(p, lo, hi):
    q = np.clip(((p - lo) / (hi - lo) * 1024.0).astype(np.int64), 0, 1023)
    code = np.zeros(len(p), dtype=np.int64)
    for b in range(10):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return code


def main():
    levels = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    scene_name = sys.argv[2] if len(sys.argv) > 2 else "heptoroid-refractive"
    pkg = importlib.import_module("rust-raytracer_amd")
    named = pkg.host.NamedScene(scene_name, detail=1.0)
    scene = pkg.Scene(named, device=0)
    light = np.array(list(named.lights[0].position), dtype=np.float64)
    rng = np.random.default_rng(1)
    physical = os.environ.get("PHYSICAL", "1") == "1"
    nt = int(named.desc.n_triangles)
    verts = np.ctypeslib.as_array(named.desc.tri_vertices, shape=(nt, 3, 3)).copy()
    assert int(named.desc.n_prims) == nt, "the experiment takes prim ids for triangle indices"
    W, H = (int(x) for x in os.environ.get("GRID", "480x272").split("x"))
    cam = named.camera(W, H, 45.0)
    eye = np.array(list(cam.position), dtype=np.float64) if hasattr(cam, "position") else None
    if eye is None:
        raise SystemExit("camera has no position field")
    look = np.array(list(cam.look_at), dtype=np.float64) if hasattr(cam, "look_at") else np.zeros(3)
    fwd = unit((look - eye)[None])[0]
    right = unit(np.cross(fwd, np.array([0.0, 1.0, 0.0]))[None])[0]
    up = np.cross(right, fwd)
    ys, xs = np.mgrid[0:H, 0:W]
    t = np.tan(np.radians(45.0) / 2)
    d = (fwd[None] + ((xs.reshape(-1, 1) + 0.5) / W * 2 - 1) * t * W / H * right[None] + (1 - (ys.reshape(-1, 1) + 0.5) / H * 2) * t * up[None])
    o = np.repeat(eye[None], len(d), 0)
    d = unit(d)
    for lv in range(levels + 1):
        pid, tt = scene.nearest_hits(o, d)
        hit = pid >= 0
        print("level %d: %d rays, %d hits" % (lv, len(o), int(hit.sum())), flush=True)
        hp = (o + d * tt[:, None])[hit]
        dh = d[hit]
        if lv == levels:
            # the same rays once more in other orders: k_nearest launches after the tree-order one above
            lo_, hi_ = o.min(0), o.max(0)
            octant = ((d[:, 0] > 0).astype(np.int64) | ((d[:, 1] > 0).astype(np.int64) << 1) | ((d[:, 2] > 0).astype(np.int64) << 2))
            m30 = morton30(o, lo_, hi_)
            q4 = np.clip(((d + 1.0) * 2.0).astype(np.int64), 0, 3)      # 2 bits per direction component (sign + half)
            dir6 = q4[:, 0] | (q4[:, 1] << 2) | (q4[:, 2] << 4)
            q8 = np.clip(((d + 1.0) * 4.0).astype(np.int64), 0, 7)
            dir9 = q8[:, 0] | (q8[:, 1] << 3) | (q8[:, 2] << 6)
            keys = {"random": rng.permutation(len(o)), "origin30": np.argsort(m30, kind="stable"),
                    "origin30_then_octant": np.argsort(m30 * 8 + octant, kind="stable"),
                    "octant_then_origin30": np.argsort((octant << 30) | m30, kind="stable"),
                    "octant_then_origin21": np.argsort((octant << 21) | (m30 >> 9), kind="stable"),
                    "dir6_then_origin18": np.argsort((dir6 << 18) | (m30 >> 12), kind="stable"),
                    "dir9_then_origin15": np.argsort((dir9 << 15) | (m30 >> 15), kind="stable"),
                    "origin9_octant_origin12": np.argsort(((m30 >> 21) << 15) | (octant << 12) | ((m30 >> 9) & 0xFFF), kind="stable"),
                    "origin6_octant_origin15": np.argsort(((m30 >> 24) << 18) | (octant << 15) | ((m30 >> 9) & 0x7FFF), kind="stable"),
                    "origin18": np.argsort(m30 >> 12, kind="stable"),
                    "origin18_then_octant": np.argsort((m30 >> 12) * 8 + octant, kind="stable")}
            for name, perm in keys.items():
                pid2, tt2 = scene.nearest_hits(o[perm], d[perm])
                ok = bool(np.array_equal(pid2, pid[perm]) and np.array_equal(tt2, tt[perm]))
                print("nearest order %s: same hits %s" % (name, ok), flush=True)
            break
        if physical:
            # the scene's own mirror and transmission directions (flat normals of the triangles hit, ior 1.5, total internal
            # reflection -> the mirror direction, as renderer.rs:148-186): the ray tree of the frame, two children per hit
            tri = verts[pid[hit]]
            n = unit(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]))
            cosi = np.sum(dh * n, axis=1, keepdims=True)
            nf = np.where(cosi < 0, n, -n)
            c = -np.sum(dh * nf, axis=1, keepdims=True)
            refl = dh + 2.0 * c * nf
            eta = np.where(cosi < 0, 1.0 / 1.5, 1.5)
            k = 1.0 - eta * eta * (1.0 - c * c)
            straight = np.where(k < 0, refl, eta * dh + (eta * c - np.sqrt(np.maximum(k, 0.0))) * nf)
            straight = unit(straight)
        else:
            n = unit(rng.normal(size=hp.shape))
            refl = dh - 2.0 * np.sum(dh * n, axis=1, keepdims=True) * n
            straight = unit(dh + 0.15 * rng.normal(size=hp.shape))
        o = np.repeat(hp, 2, axis=0)
        d = np.empty_like(o)
        d[0::2] = refl
        d[1::2] = straight
        o = o + 1e-6 * d
    n = len(hp)
    lo, hi = hp.min(0), hp.max(0)
    m = morton30(hp, lo, hi)
    orders = {"tree": np.arange(n), "random": rng.permutation(n), "sorted30": np.argsort(m, kind="stable"),
              "sorted24": np.argsort(m >> 6, kind="stable"), "sorted18": np.argsort(m >> 12, kind="stable"),
              "sorted12": np.argsort(m >> 18, kind="stable")}
    tgt = np.repeat(light[None], n, 0)
    ref = None
    for name, perm in orders.items():
        out = scene.shadow_products(hp[perm], tgt)
        back = np.empty_like(out)
        back[perm] = out
        if ref is None:
            ref = back
        print("%s: %d feelers, mean product %.6f, equal to the first order: %s" % (name, n, float(back.mean()), bool(np.array_equal(back, ref))), flush=True)
    scene.close()


if __name__ == "__main__":
    main()
