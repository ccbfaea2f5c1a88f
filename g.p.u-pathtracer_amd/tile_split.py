"""Framebuffer tile split across the GPUs of one node + gather of the tiles.

The reference is single-GPU (utilfun.cpp:72-99 hard-codes device 0); this is the new
multi-GPU axis of BASELINE.json.  Pixels are independent and the RNG is keyed by the GLOBAL
pixel index, so the path needs NO collective while rendering: every rank holds the whole
scene and renders the row stripes it owns (stripe s -> rank s % world, `rows` rows each,
pt_params.part_*).  The only exchange is the gather of finished stripes on rank 0 when an
image is wanted (display words every step, the float accumulator at the end) — RCCL over xGMI
on GPUs (backend "nccl"), gloo in the CPU tests.

Buffers are full-frame with the height padded to a whole number of stripes per rank, so a
rank's stripes are a strided view [k, rank, :] of the frame and the gather needs one
contiguous staging copy per rank and no index arithmetic.
"""
import torch
import torch.distributed as dist


class StripeLayout:
    def __init__(self, width, height, world, rank, rows=8):
        if rows <= 0 or rows % 8:
            raise ValueError("stripe rows must be a positive multiple of 8 (the wave tile)")
        self.W, self.H, self.world, self.rank, self.rows = width, height, world, rank, rows
        self.n_stripes = -(-height // rows)
        self.stripes_per_rank = -(-self.n_stripes // world)
        self.padded_height = self.stripes_per_rank * world * rows

    def owned_rows(self, rank=None):
        """Global row indices (inside the real image) owned by `rank`."""
        rank = self.rank if rank is None else rank
        out = []
        for s in range(rank, self.n_stripes, self.world):
            out.extend(range(s * self.rows, min((s + 1) * self.rows, self.H)))
        return out

    def apply(self, params):
        """Fill pt_params.part_* for this rank."""
        params.part_index, params.part_count, params.part_rows = self.rank, self.world, self.rows
        return params

    def frame_view(self, frame):
        """frame: [padded_height, W, C...] tensor -> [stripes_per_rank, world, rows*W*C] view."""
        per = frame[0].numel() * self.rows
        return frame.view(self.stripes_per_rank, self.world, per)


def gather_stripes(frame, layout, dst=0, group=None, staging=None, force=False):
    """Collect every rank's stripes of `frame` into rank `dst`'s copy of `frame` (in place) — SURVEY 8(e)'s collective: a
    GATHER TO ROOT.  `frame` is any full-frame buffer: the display words uint32[H][W] (gathered every step, 1 MB per
    rank at 8 GPUs) or the float accumulator [H][W][3] (north_star's "accumulated tiles", 3.1 MB per rank at 8 GPUs,
    gathered when the image is wanted).

    torch.distributed.gather on every backend: over RCCL (backend "nccl") it is one group of ncclSend / ncclRecv — each
    peer sends its 1/world of the frame over its own xGMI link to the root, which receives on 7 links at once; nothing
    goes to ranks that do not need it (an all-gather would move world x as many bytes for the same result).  gloo in the
    CPU tests / rehearsal.  Runs on the CURRENT stream: call it under `torch.cuda.stream(side)` to overlap it with the
    next frame's render.  Returns the staging tensors so callers in a timed loop reuse them."""
    if layout.world == 1 and not force:   # force: a one-rank group still goes through the collective (bench --force-dist)
        return staging
    view = layout.frame_view(frame)
    if staging is None:
        send = torch.empty_like(view[:, 0, :])
        recv = [torch.empty_like(send) for _ in range(layout.world)] if layout.rank == dst else None
        staging = (send, recv)
    send, recv = staging
    send.copy_(view[:, layout.rank, :])
    dist.gather(send, recv, dst=dst, group=group)
    if layout.rank == dst:
        for r in range(layout.world):
            if r != dst:
                view[:, r, :].copy_(recv[r])
    return staging
