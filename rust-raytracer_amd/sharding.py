"""Multi-GPU sharding of the trace path: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Stills:     the frame is cut into blocks of 8 rows (the reference's tile height, renderer.rs:42);
            block b belongs to rank b % world (interleaved for load balance).  Each rank renders its
            blocks with rtx_options.shard_index/shard_count, packs them into a compact buffer and
            ONE gather brings them to rank 0, which de-interleaves them into the final Surface —
            the counterpart of Surface::merge (compositor/surface.rs:83-94).
Animations: frame k belongs to rank k % world (animator.rs:44-48 renders frames independently);
            each finished frame is gathered to rank 0 in rounds of `world` frames.

The RNG is keyed on absolute pixel / frame numbers, so the image does not depend on `world`.
Nothing here touches the oracle; the functions work on CPU tensors too (that is what the gloo
tests exercise).
"""
import torch


def blocks_of_rank(height, rank, world):
    """8-row block indices owned by `rank` (block b -> rank b % world)."""
    return list(range(rank, height // 8, world))


def frames_of_rank(total_frames, rank, world):
    """Animation frame numbers rendered by `rank` (frame k -> rank k % world)."""
    return list(range(rank, total_frames, world))


class StillGather:
    """Pre-allocated buffers for gathering an interleaved still to rank 0."""

    def __init__(self, width, height, rank, world, device):
        assert height % 8 == 0, "image_height must be a multiple of 8 (renderer.rs:73)"
        self.W, self.H, self.rank, self.world = width, height, rank, world
        self.n_blocks = height // 8
        self.row_bytes = 8 * width * 4
        self.mine = blocks_of_rank(height, rank, world)
        self.max_blocks = (self.n_blocks + world - 1) // world
        self.compact = torch.zeros((self.max_blocks, self.row_bytes), dtype=torch.uint8, device=device)
        self.idx = torch.tensor(self.mine, dtype=torch.long, device=device)
        self.parts = ([torch.zeros_like(self.compact) for _ in range(world)]
                      if (world > 1 and rank == 0) else None)
        self.final = torch.zeros((height, width, 4), dtype=torch.uint8, device=device) if rank == 0 else None

    def gather(self, rgba, dist=None):
        """rgba: this rank's [H, W, 4] uint8 frame (only its own blocks need to be valid).
        Returns the assembled frame on rank 0, None elsewhere."""
        if self.world == 1:
            self.final.copy_(rgba)
            return self.final
        blocks = rgba.view(self.n_blocks, self.row_bytes)
        if self.mine:
            self.compact[:len(self.mine)] = blocks.index_select(0, self.idx)
        dist.gather(self.compact, self.parts, dst=0)
        if self.rank != 0:
            return None
        out = self.final.view(self.n_blocks, self.row_bytes)
        for r in range(self.world):
            n = len(range(r, self.n_blocks, self.world))
            if n:
                out[r::self.world] = self.parts[r][:n]
        return self.final


def gather_frames(frame, have_frame, rank, world, dist):
    """One round of the animation gather: every rank passes its finished frame ([H, W, 4] uint8) or
    a dummy of the same shape with have_frame=False; rank 0 gets the list of the round's frames
    in rank order (None where a rank had none)."""
    if world == 1:
        return [frame if have_frame else None]
    flag = torch.tensor([1 if have_frame else 0], dtype=torch.uint8, device=frame.device)
    parts = [torch.zeros_like(frame) for _ in range(world)] if rank == 0 else None
    flags = [torch.zeros_like(flag) for _ in range(world)] if rank == 0 else None
    dist.gather(frame, parts, dst=0)
    dist.gather(flag, flags, dst=0)
    if rank != 0:
        return None
    return [p if int(f.item()) else None for p, f in zip(parts, flags)]


def animation_frames(fps, time_from, time_to):
    """Animator::animate's frame list (animator.rs:24,45): total_frames = floor(fps * (to - from)), frame k at from + k / fps."""
    total = int(fps * (time_to - time_from))
    return [(k, time_from + k / fps) for k in range(total)]


def frame_name(output_file, starting_frame_number, frame_number):
    """animator.rs:35-37: `{output_file}{:06}.ppm`, numbered from starting_frame_number (restart of a partial animation)."""
    return "%s%06d.ppm" % (output_file, starting_frame_number + frame_number)


def animate(frames, rank, world, dist, render_frame, on_frame=None):
    """N-rank animation driver: frame k of `frames` (a list of (frame_number, time)) is rendered by rank k % world with
    render_frame(frame_number, time) -> [H, W, 4] uint8 tensor; after every round of `world` frames ONE gather brings the
    round's frames to rank 0, which hands them to on_frame(frame_number, time, tensor) in frame order (the reference's
    writer thread, animator.rs:27-42).  Returns the number of frames this rank rendered.  No other collective."""
    rendered = 0
    blank = None
    for base in range(0, len(frames), world):
        k = base + rank
        have = k < len(frames)
        if have:
            frame = render_frame(*frames[k])
            rendered += 1
            blank = frame
        else:
            if blank is None:       # this rank has rendered nothing yet: it needs the frame's shape from a peer-independent source
                raise RuntimeError("animate(): fewer frames than ranks; give every rank at least one frame")
            frame = blank
        parts = gather_frames(frame, have, rank, world, dist)
        if rank == 0 and on_frame is not None:
            for r, part in enumerate(parts):
                if part is not None:
                    on_frame(frames[base + r][0], frames[base + r][1], part)
    return rendered
