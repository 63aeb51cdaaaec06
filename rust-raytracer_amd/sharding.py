"""Multi-GPU sharding of the trace path: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Stills:     the frame is cut into blocks of 8 rows (the reference's tile height, renderer.rs:42);
            block b belongs to rank b % world (interleaved for load balance).  Each rank renders its
            blocks with rtx_options.shard_index/shard_count and RTX_FLAG_COMPACT_OUTPUT straight into a
            compact buffer of ceil(H/8/world) blocks (StillGather.target), and ONE gather brings the
            compact buffers to rank 0, which de-interleaves them into the final Surface — the
            counterpart of Surface::merge (compositor/surface.rs:83-94).
Animations: frame k belongs to rank k % world (animator.rs:44-48 renders frames independently);
            the finished frames of a round of `world` frames travel to rank 0 in ONE gather (the
            "this rank had a frame" flag rides in a 16-byte trailer of the frame buffer), and the
            gather of round r runs while round r + 1 renders (FrameGather: two send and two receive
            buffers, allocated once) — as the reference overlaps its PPM writer thread with the next
            frame's render (animator.rs:27-42).

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
    """Pre-allocated buffers for gathering an interleaved still to rank 0.

    `target` ([max_blocks * 8, W, 4] uint8 on `device`) is what the rank renders into with
    RTX_FLAG_COMPACT_OUTPUT: its own blocks, packed.  gather(dist) sends it as it is; gather(dist, rgba=full_frame)
    packs the rank's blocks out of a full-size frame first (callers that rendered without the flag)."""

    def __init__(self, width, height, rank, world, device, always_collective=False):
        assert height % 8 == 0, "image_height must be a multiple of 8 (renderer.rs:73)"
        self.W, self.H, self.rank, self.world = width, height, rank, world
        # always_collective: issue the gather on a group of ONE rank too (a one-GPU rehearsal of the RCCL calls, tools/gpu_rccl_rehearsal.py)
        self.collective = world > 1 or always_collective
        self.n_blocks = height // 8
        self.row_bytes = 8 * width * 4
        self.mine = blocks_of_rank(height, rank, world)
        self.max_blocks = (self.n_blocks + world - 1) // world
        self.idx = torch.tensor(self.mine, dtype=torch.long, device=device)
        self.recv = self.parts = self.final = self.dst = None
        if rank == 0:
            # rank 0 receives every rank's compact buffer into ONE array and de-interleaves it with ONE indexed copy: slot (r, j) holds
            # block r + j * world; the slots past a rank's last block go to a spare block behind the frame
            self.recv = torch.zeros((world, self.max_blocks, self.row_bytes), dtype=torch.uint8, device=device)
            self.parts = [self.recv[r] for r in range(world)]
            dst = [(r + j * world) if (r + j * world) < self.n_blocks else self.n_blocks for r in range(world) for j in range(self.max_blocks)]
            self.dst = torch.tensor(dst, dtype=torch.long, device=device)
            self._padded = torch.zeros((self.n_blocks + 1, self.row_bytes), dtype=torch.uint8, device=device)
            self.final = self._padded[:self.n_blocks].view(height, width, 4)
            self.compact = self.recv[0] if not self.collective else torch.zeros((self.max_blocks, self.row_bytes), dtype=torch.uint8, device=device)
        else:
            self.compact = torch.zeros((self.max_blocks, self.row_bytes), dtype=torch.uint8, device=device)
        self.target = self.compact.view(self.max_blocks * 8, width, 4)

    def gather(self, dist=None, rgba=None):
        """Returns the assembled frame on rank 0, None elsewhere."""
        if rgba is not None and self.mine:
            self.compact[:len(self.mine)] = rgba.view(self.n_blocks, self.row_bytes).index_select(0, self.idx)
        if self.collective:
            dist.gather(self.compact, self.parts, dst=0)
            if self.rank != 0:
                return None
        self._padded.index_copy_(0, self.dst, self.recv.view(self.world * self.max_blocks, self.row_bytes))
        return self.final


class FrameGather:
    """One gather per round of an animation, pipelined one round deep.

    Every rank owns two send buffers, rank 0 two sets of `world` receive buffers (allocated once: 2 x world x (H*W*4 + 16) bytes).
    submit(frame or None) copies the rank's finished frame into the round's send buffer (its last 16 bytes carry the "have a
    frame" flag) and starts the round's gather asynchronously; it first completes the PREVIOUS round's gather and returns that
    round's frames on rank 0 ([tensor or None] per rank, valid until the round after next overwrites them), so the transfer of
    round r overlaps the rendering of round r + 1.  flush() completes the last round."""

    TRAILER = 16

    def __init__(self, height, width, rank, world, device, dist, always_collective=False):
        self.H, self.W, self.rank, self.world, self.dist = height, width, rank, world, dist
        self.collective = world > 1 or always_collective   # (as StillGather)
        self.nbytes = height * width * 4
        self.send = [torch.zeros(self.nbytes + self.TRAILER, dtype=torch.uint8, device=device) for _ in range(2)]
        self.recv = ([[torch.zeros_like(self.send[0]) for _ in range(world)] for _ in range(2)]
                     if (self.collective and rank == 0) else None)
        self.round = 0
        self.pending = None      # (work handle or None, buffer index)

    def _complete(self):
        if self.pending is None:
            return None
        work, b = self.pending
        self.pending = None
        if work is not None:
            work.wait()
        if self.rank != 0:
            return None
        bufs = self.recv[b] if self.collective else [self.send[b]]
        # one host read for the round's flags (the frames stay where they are)
        flags = torch.stack([p[self.nbytes] for p in bufs]).cpu().tolist()
        return [p[:self.nbytes].view(self.H, self.W, 4) if f else None for p, f in zip(bufs, flags)]

    def submit(self, frame):
        done = self._complete()
        b = self.round & 1
        buf = self.send[b]
        if frame is not None:
            buf[:self.nbytes].copy_(frame.reshape(-1))
        buf[self.nbytes:].fill_(1 if frame is not None else 0)
        work = None
        if self.collective:
            work = self.dist.gather(buf, self.recv[b] if self.rank == 0 else None, dst=0, async_op=True)
        self.pending = (work, b)
        self.round += 1
        return done

    def flush(self):
        return self._complete()


def gather_frames(frame, have_frame, rank, world, dist):
    """One blocking round of the animation gather (the un-pipelined form, kept for callers with a single round): every rank
    passes its finished frame ([H, W, 4] uint8) or anything of the same shape with have_frame=False; rank 0 gets the list of the
    round's frames in rank order (None where a rank had none).  One collective: the flag rides in the frame buffer's trailer."""
    fg = FrameGather(frame.shape[0], frame.shape[1], rank, world, frame.device, dist)
    fg.submit(frame if have_frame else None)
    return fg.flush()


def animation_frames(fps, time_from, time_to):
    """Animator::animate's frame list (animator.rs:24,45): total_frames = floor(fps * (to - from)), frame k at from + k / fps."""
    total = int(fps * (time_to - time_from))
    return [(k, time_from + k / fps) for k in range(total)]


def frame_name(output_file, starting_frame_number, frame_number):
    """animator.rs:35-37: `{output_file}{:06}.ppm`, numbered from starting_frame_number (restart of a partial animation)."""
    return "%s%06d.ppm" % (output_file, starting_frame_number + frame_number)


def animate(frames, rank, world, dist, render_frame, on_frame=None, shape=None, device=None, always_collective=False):
    """N-rank animation driver: frame k of `frames` (a list of (frame_number, time)) is rendered by rank k % world with
    render_frame(frame_number, time) -> [H, W, 4] uint8 tensor; ONE gather per round of `world` frames brings the round's frames
    to rank 0, overlapped with the next round's rendering (FrameGather), and rank 0 hands them to
    on_frame(frame_number, time, tensor) in frame order (the reference's writer thread, animator.rs:27-42).
    shape = (H, W) and device: needed only by a rank that renders nothing in the first round (fewer frames than ranks); every rank
    takes the same decision about that before any collective, so nobody is left waiting in one.
    Returns the number of frames this rank rendered.  No other collective."""
    if not frames:
        return 0
    if len(frames) < world and shape is None:   # (every rank evaluates the same condition: all raise, none enters a collective)
        raise RuntimeError("animate(): fewer frames than ranks; pass shape=(H, W) and device= so that idle ranks can take part in the gather")
    rendered = 0
    fg = None
    first = None

    def deliver(base, parts):
        if rank == 0 and on_frame is not None and parts is not None:
            for r, part in enumerate(parts):
                if part is not None:
                    on_frame(frames[base + r][0], frames[base + r][1], part)

    prev_base = None
    for base in range(0, len(frames), world):
        k = base + rank
        frame = None
        if k < len(frames):
            frame = render_frame(*frames[k])
            rendered += 1
        if fg is None:
            if frame is not None:
                first = (frame.shape[0], frame.shape[1], frame.device)
            else:
                first = (shape[0], shape[1], device if device is not None else torch.device("cpu"))
            fg = FrameGather(first[0], first[1], rank, world, first[2], dist, always_collective)
        done = fg.submit(frame)          # starts this round's gather; completes the previous round's
        deliver(prev_base, done)
        prev_base = base
    deliver(prev_base, fg.flush())
    return rendered
