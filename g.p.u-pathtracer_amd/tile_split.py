"""Framebuffer tile split across the GPUs of one node + gather of the tiles.

The reference is single-GPU (utilfun.cpp:72-99 hard-codes device 0); this is the new
multi-GPU axis of BASELINE.json.  Pixels are independent and the RNG is keyed by the GLOBAL
pixel index, so the path needs NO collective while rendering: every rank holds the whole
scene and renders the row stripes it owns (stripe s -> rank s % world, `rows` rows each,
pt_params.part_*).  The only exchange is the gather of finished stripes on rank 0 when an
image is wanted — RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU tests.

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
    """Collect every rank's stripes of `frame` into rank `dst`'s copy of `frame` (in place).

    GPU tensors: one all_gather_into_tensor (RCCL all-gather: 1/world of the frame per rank,
    every rank's xGMI links carry one slice each way) + ONE permuted copy on `dst`.
    CPU tensors (gloo tests / rehearsal): dist.gather + per-rank copies.
    Runs on the CURRENT stream: call it under `torch.cuda.stream(side)` to overlap it with the
    next frame's render.  Returns the staging tensors so callers in a timed loop reuse them."""
    if layout.world == 1 and not force:   # force: a one-rank group still goes through the collective (bench --force-dist)
        return staging
    view = layout.frame_view(frame)
    if frame.is_cuda:
        if staging is None:
            send = torch.empty_like(view[:, 0, :])
            recv = torch.empty((layout.world,) + tuple(send.shape), dtype=send.dtype, device=send.device)
            staging = (send, recv)
        send, recv = staging
        send.copy_(view[:, layout.rank, :])
        dist.all_gather_into_tensor(recv, send, group=group)
        if layout.rank == dst:
            view.copy_(recv.permute(1, 0, 2))
        return staging
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
