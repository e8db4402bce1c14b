"""Multi-GPU frame rendering: one process per GPU, image columns partitioned across ranks, the
framebuffer gathered to rank 0 with torch.distributed (RCCL over xGMI on GPU, gloo in CPU tests).

Pixels are independent in the reference (`for px .. for py .. renderPass`, Scene.fs:119-121); the
only cross-pixel state is `maxFrameBufferDepth` (Scene.fs:47-48).  Rank r renders the interleaved
columns px = r, r+N, r+2N, ... (rt_params.x_begin/x_step): interleaving balances the load, because
cost per column varies smoothly across the image (floor columns fan out 4^k, light columns stop
at depth 0).  The frame buffer is x-major (`Array2D.create width height`, index px*H+py), so each
rank's compact output is a [ncols_r, H, ...] block.

Exchange per frame: one gather per output (rgba 16 B/px, depth 4 B/px [, ids 8 B/px]) straight from the buffers the
kernel rendered into (`FrameGatherer.part_*` ARE the render targets: no packing pass) into one [world, mc, H, ...]
receive tensor on rank 0, ONE strided placement pass there (frame[k*N + r] = recv[r, k]), and one 2-element
all-gather of the scalars (ray count: summed, max depth: maxed on the host).
Nothing is exchanged while rendering.
"""
import torch
import torch.distributed as dist


def columns_of_rank(width, rank, world):
    """(x_begin, x_end, x_step, ncols) of the interleaved partition."""
    ncols = (width - rank + world - 1) // world if rank < width else 0
    return rank, width, world, ncols


def max_columns(width, world):
    return (width + world - 1) // world


def place_blocks(dst, src, width, world):
    """dst [W, ...] <- src [world, mc, ...]: column k of rank r goes to frame column k*world + r.  One pass over the
    data: the `width // world` full rounds as one transposed copy, the ragged last round as a second, small one."""
    full = width // world
    if full:
        dst[:full * world].view(full, world, *dst.shape[1:]).copy_(src[:, :full].transpose(0, 1))
    rem = width - full * world
    if rem:
        dst[full * world:].copy_(src[:rem, full])


class FrameGatherer:
    """Preallocated buffers for gathering [ncols_r, H, C] column blocks into a [W, H, C] frame on rank 0.

    `part_rgba` / `part_depth` / `part_ids` ([mc, H, ...], mc = columns of the widest share) are this rank's send
    buffers; render straight into them (`render_targets()`) and `gather()` moves nothing before the collective."""

    def __init__(self, width, height, device, rank=None, world=None, with_ids=False):
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.width, self.height, self.device = width, height, device
        self.mc = max_columns(width, self.world)
        self.ncols = columns_of_rank(width, self.rank, self.world)[3]
        self.with_ids = with_ids
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)
        self.part_rgba = z((self.mc, height, 4), torch.float32)
        self.part_depth = z((self.mc, height), torch.float32)
        self.part_ids = z((self.mc, height), torch.int64) if with_ids else None
        root = self.rank == 0
        many = root and self.world > 1
        self.recv_rgba = z((self.world, self.mc, height, 4), torch.float32) if many else None
        self.recv_depth = z((self.world, self.mc, height), torch.float32) if many else None
        self.recv_ids = z((self.world, self.mc, height), torch.int64) if (many and with_ids) else None
        self.rgba = z((width, height, 4), torch.float32) if root else None
        self.depth = z((width, height), torch.float32) if root else None
        self.ids = z((width, height), torch.int64) if (root and with_ids) else None
        self.stat = torch.zeros(2, dtype=torch.float64, device=device)
        self.stat_all = torch.zeros((self.world, 2), dtype=torch.float64, device=device)
        self.last_scalars = None          # [world, 2] on the host after gather(): every rank's ray count and max depth

    def render_targets(self):
        """(rgba, depth, ids) views of this rank's columns inside the send buffers."""
        n = self.ncols
        return self.part_rgba[:n], self.part_depth[:n], (self.part_ids[:n] if self.with_ids else None)

    def _stage(self, part, t):
        n = t.shape[0]
        if n and t.data_ptr() != part.data_ptr():
            part[:n].copy_(t.view(part[:n].shape) if t.dtype == part.dtype else t.to(part.dtype).view(part[:n].shape))

    def _one(self, part, recv, dst):
        if self.world > 1:
            dist.gather(part, list(recv.unbind(0)) if self.rank == 0 else None, dst=0)
            if self.rank == 0:
                place_blocks(dst, recv, self.width, self.world)
        elif self.rank == 0:
            place_blocks(dst, part.unsqueeze(0), self.width, 1)

    def gather(self, rgba, depth, ids, rays, max_depth):
        """rgba [ncols,H,4] f32, depth [ncols,H] f32, ids [ncols,H] int64 or None: this rank's columns (the
        `render_targets()` views themselves, or any other tensors, which are then copied into the send buffers).
        Returns (total_rays, global_max_depth); rank 0's self.rgba/self.depth/self.ids hold the frame."""
        self._stage(self.part_rgba, rgba)
        self._stage(self.part_depth, depth)
        self._one(self.part_rgba, self.recv_rgba, self.rgba)
        self._one(self.part_depth, self.recv_depth, self.depth)
        if self.with_ids:
            self._stage(self.part_ids, ids)
            self._one(self.part_ids, self.recv_ids, self.ids)
        # ray count (sum) and maxFrameBufferDepth (max, Scene.fs:47-48): one small all-gather, reduced on the host
        self.stat.copy_(torch.tensor([float(rays), float(max_depth)], dtype=torch.float64))
        if self.world > 1:
            dist.all_gather(list(self.stat_all.unbind(0)), self.stat)
            s = self.stat_all.cpu()
        else:
            s = self.stat.cpu().view(1, 2)
        self.last_scalars = s
        return int(round(float(s[:, 0].sum().item()))), float(s[:, 1].max().item())
