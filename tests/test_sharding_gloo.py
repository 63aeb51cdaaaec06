"""The N > 1 path on CPU: two `gloo` ranks run the product's gather code (sharding.py) on synthetic
per-rank frames.  On the GPU box the same code runs over RCCL with rtx_render_device filling the
rows (bench.py --gpus N)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pattern(W, H):
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    return np.stack([(xx * 7 + yy * 3) % 256, (xx + yy * 11) % 256, (xx * yy) % 256, np.full_like(xx, 255)], axis=-1).astype(np.uint8)


def _worker(rank, world, port, W, H, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = importlib.import_module("rust-raytracer_amd.sharding")
    full = torch.from_numpy(_pattern(W, H))
    # this rank "renders" only the 8-row blocks it owns (what rtx_options.shard_index/count does)
    mine = torch.zeros_like(full)
    for b in sh.blocks_of_rank(H, rank, world):
        mine[b * 8:(b + 1) * 8] = full[b * 8:(b + 1) * 8]
    g = sh.StillGather(W, H, rank, world, torch.device("cpu"))
    out = g.gather(dist, rgba=mine)
    ok_still = bool(torch.equal(out, full)) if rank == 0 else out is None
    # the same with the rank's blocks written straight into the compact target (what RTX_FLAG_COMPACT_OUTPUT does on the device)
    g2 = sh.StillGather(W, H, rank, world, torch.device("cpu"))
    assert g2.target.shape == (g2.max_blocks * 8, W, 4)
    for j, b in enumerate(sh.blocks_of_rank(H, rank, world)):
        g2.target[j * 8:(j + 1) * 8] = full[b * 8:(b + 1) * 8]
    out2 = g2.gather(dist)
    ok_still = ok_still and (bool(torch.equal(out2, full)) if rank == 0 else out2 is None)
    # animation: 5 frames over 2 ranks, rounds of `world` frames
    total = 5
    got = {}
    rounds = (total + world - 1) // world
    for rd in range(rounds):
        k = rd * world + rank
        have = k < total
        frame = torch.full((H, W, 4), k % 251 if have else 0, dtype=torch.uint8)
        parts = sh.gather_frames(frame, have, rank, world, dist)
        if rank == 0:
            for r, p in enumerate(parts):
                if p is not None:
                    got[rd * world + r] = int(p[0, 0, 0])
    ok_anim = (got == {k: k % 251 for k in range(total)}) if rank == 0 else True
    assert sh.frames_of_rank(total, rank, world) == list(range(rank, total, world))
    q.put((rank, ok_still, ok_anim))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("W,H", [(96, 64), (50, 136)])
def test_two_rank_gather(W, H):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == [0, 1]
    assert all(r[1] and r[2] for r in results), results


def _anim_worker(rank, world, port, q):
    """bench.py --animate's driver (sharding.animate) with a stub renderer: 7 frames at 2 fps from t = 1 s, starting_frame_number 40."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = importlib.import_module("rust-raytracer_amd.sharding")
    frames = sh.animation_frames(2.0, 1.0, 4.5)          # floor(2 * 3.5) = 7 frames
    rendered_here, written = [], []

    def render_frame(k, t):
        rendered_here.append(k)
        return torch.full((16, 8, 4), (k * 10 + int(t * 2)) % 251, dtype=torch.uint8)

    def on_frame(k, t, frame):
        written.append((sh.frame_name("out/anim", 40, k), t, int(frame[0, 0, 0])))

    n = sh.animate(frames, rank, world, dist, render_frame, on_frame)
    q.put((rank, n, rendered_here, written))
    dist.barrier()
    dist.destroy_process_group()


def _short_anim_worker(rank, world, port, q):
    """Fewer frames than ranks: the idle rank takes part in the gather with shape=/device=; without them EVERY rank raises before
    any collective (nobody is left waiting in one)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = importlib.import_module("rust-raytracer_amd.sharding")
    frames = [(0, 0.25)]
    written = []
    raised = False
    try:
        sh.animate(frames, rank, world, dist, lambda k, t: torch.full((8, 4, 4), 9, dtype=torch.uint8), None)
    except RuntimeError:
        raised = True
    n = sh.animate(frames, rank, world, dist, lambda k, t: torch.full((8, 4, 4), 9, dtype=torch.uint8),
                   lambda k, t, f: written.append((k, t, int(f[7, 3, 2]))), shape=(8, 4), device=torch.device("cpu"))
    q.put((rank, raised, n, written))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_animation_with_fewer_frames_than_ranks():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_short_anim_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r for r in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1]                 # both ranks raised, neither hung
    assert (res[0][2], res[1][2]) == (1, 0)
    assert res[0][3] == [(0, 0.25, 9)] and res[1][3] == []


def test_frame_gather_single_rank_pipeline(rtx):
    """FrameGather at world 1: submit returns the PREVIOUS round's frame, flush the last one; buffers alternate."""
    sh = rtx.sharding
    fg = sh.FrameGather(8, 4, 0, 1, torch.device("cpu"), None)
    assert fg.submit(torch.full((8, 4, 4), 1, dtype=torch.uint8)) is None
    d = fg.submit(torch.full((8, 4, 4), 2, dtype=torch.uint8))
    assert len(d) == 1 and int(d[0][0, 0, 0]) == 1
    d = fg.submit(None)
    assert int(d[0][3, 3, 3]) == 2
    assert fg.flush() == [None]
    assert fg.flush() is None


def test_two_rank_animation_driver():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_anim_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r for r in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == 4 and res[1][1] == 3                       # frame k on rank k % 2
    assert res[0][2] == [0, 2, 4, 6] and res[1][2] == [1, 3, 5]
    assert res[1][3] == []                                         # only rank 0 receives frames
    want = [("out/anim%06d.ppm" % (40 + k), 1.0 + k / 2.0, (k * 10 + int((1.0 + k / 2.0) * 2)) % 251) for k in range(7)]
    assert res[0][3] == want                                       # in frame order, numbered from starting_frame_number


def test_block_ownership_partitions_the_frame(rtx):
    sh = rtx.sharding
    for H in (8, 64, 1080):
        for world in (1, 2, 3, 8):
            owned = sorted(b for r in range(world) for b in sh.blocks_of_rank(H, r, world))
            assert owned == list(range(H // 8))
    assert sh.frames_of_rank(175, 3, 8) == list(range(3, 175, 8))     # sibenik: 175 frames over 8 GPUs
