"""Multi-GPU frame rendering: one process per GPU, image columns partitioned across ranks, the
framebuffer gathered to rank 0 with torch.distributed (RCCL over xGMI on GPU, gloo in CPU tests).

Pixels are independent in the reference (`for px .. for py .. renderPass`, Scene.fs:119-121); the
only cross-pixel state is `maxFrameBufferDepth` (Scene.fs:47-48).  Rank r renders the interleaved
columns px = r, r+N, r+2N, ... (rt_params.x_begin/x_step): interleaving balances the load, because
cost per column varies smoothly across the image (floor columns fan out 4^k, light columns stop
at depth 0).  The frame buffer is x-major (`Array2D.create width height`, index px*H+py), so each
rank's compact output is a [ncols_r, H, ...] block and the gather is one contiguous message per
rank followed by a strided placement on the root.

Exchange per frame: gather(rgba 16 B/px + depth 4 B/px [+ ids 8 B/px]) and one 2-element
all-reduce (ray count: sum, max depth: max).  Nothing is exchanged while rendering.
"""
import torch
import torch.distributed as dist


def columns_of_rank(width, rank, world):
    """(x_begin, x_end, x_step, ncols) of the interleaved partition."""
    ncols = (width - rank + world - 1) // world if rank < width else 0
    return rank, width, world, ncols


def max_columns(width, world):
    return (width + world - 1) // world


class FrameGatherer:
    """Preallocated buffers for gathering [ncols_r, H, C] column blocks into a [W, H, C] frame on rank 0."""

    def __init__(self, width, height, device, rank=None, world=None, with_ids=False):
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.width, self.height, self.device = width, height, device
        self.mc = max_columns(width, self.world)
        self.with_ids = with_ids
        # one packed f32 message per rank: rgba (4) + depth (1) [+ ids as 2 f32 words]
        self.chan = 5 + (2 if with_ids else 0)
        self.part = torch.zeros((self.mc, height, self.chan), dtype=torch.float32, device=device)
        self.recv = ([torch.zeros_like(self.part) for _ in range(self.world)] if self.rank == 0 else None)
        if self.rank == 0:
            self.rgba = torch.zeros((width, height, 4), dtype=torch.float32, device=device)
            self.depth = torch.zeros((width, height), dtype=torch.float32, device=device)
            self.ids = torch.zeros((width, height), dtype=torch.int64, device=device) if with_ids else None
        self.stat = torch.zeros(2, dtype=torch.float64, device=device)

    def gather(self, rgba, depth, ids, rays, max_depth):
        """rgba [ncols,H,4] f32, depth [ncols,H] f32, ids [ncols,H] int64 or None (this rank's columns).
        Returns (total_rays, global_max_depth); rank 0's self.rgba/self.depth/self.ids hold the frame."""
        n = rgba.shape[0]
        self.part[:n, :, 0:4] = rgba
        self.part[:n, :, 4] = depth
        if self.with_ids:
            self.part[:n, :, 5:7] = ids.contiguous().view(torch.float32).view(n, self.height, 2)
        if self.world > 1:
            dist.gather(self.part, self.recv, dst=0)
        elif self.rank == 0:
            self.recv[0].copy_(self.part)
        if self.rank == 0:
            for r in range(self.world):
                _, _, _, nr = columns_of_rank(self.width, r, self.world)
                blk = self.recv[r][:nr]
                self.rgba[r::self.world] = blk[:, :, 0:4]
                self.depth[r::self.world] = blk[:, :, 4]
                if self.with_ids:
                    self.ids[r::self.world] = blk[:, :, 5:7].contiguous().view(torch.int64).view(nr, self.height)
        # ray count (sum) and maxFrameBufferDepth (max): two tiny all-reduces
        self.stat[0] = float(rays)
        self.stat[1] = float(max_depth)
        if self.world > 1:
            s = self.stat[0:1].clone()
            m = self.stat[1:2].clone()
            dist.all_reduce(s, op=dist.ReduceOp.SUM)
            dist.all_reduce(m, op=dist.ReduceOp.MAX)
            return int(s.item()), float(m.item())
        return int(rays), float(max_depth)
