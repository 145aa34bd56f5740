"""Row tiling of a frame over several GPUs (one process each) and the gather of the
rendered stripes to rank 0 with torch.distributed (RCCL on GPUs, gloo in the CPU tests).

Rows are grouped in stripes of `stripe_rows`; stripe s belongs to part s % n_parts
(rtx_params in include/rtx.h).  Interleaving the stripes spreads the rows that cover a
mesh evenly over the GPUs; the light sampler is keyed by the global pixel index, so the
assembled frame is identical for every split.
"""
import numpy as np


def rows_of_part(height, stripe_rows, n_parts, part):
    """Global row indices, increasing, that `part` renders."""
    j = np.arange(height)
    if n_parts <= 1 or stripe_rows == 0:
        return j
    return j[(j // stripe_rows) % n_parts == part]


def max_part_rows(height, stripe_rows, n_parts):
    return max(len(rows_of_part(height, stripe_rows, n_parts, k)) for k in range(max(n_parts, 1)))


def assemble_numpy(parts, height, stripe_rows):
    """parts[k]: uint8 [rows_k (or more, padded), W, 3] -> frame [height, W, 3]."""
    n = len(parts)
    frame = np.zeros((height,) + parts[0].shape[1:], parts[0].dtype)
    for k, t in enumerate(parts):
        rows = rows_of_part(height, stripe_rows, n, k)
        frame[rows] = t[:len(rows)]
    return frame


class FrameGather:
    """Gathers the equally padded tiles of all ranks to rank 0 and scatters their rows
    into one frame tensor there.  Buffers and the row index are allocated once; the
    scatter is a single index_copy_ (the padding rows of the tiles go to a spare row
    behind the frame).  With one rank the tile IS the frame: nothing to gather or move."""

    def __init__(self, torch, dist, height, width, stripe_rows, world, rank, device):
        self.torch, self.dist = torch, dist
        self.world, self.rank = world, rank
        self.height = height
        self.pad_rows = max_part_rows(height, stripe_rows, world)
        self.tile = torch.zeros((self.pad_rows, width, 3), dtype=torch.uint8, device=device)
        self.frame = None
        self.bufs = None
        if rank == 0:
            self._store = torch.zeros((height + 1, width, 3), dtype=torch.uint8, device=device)
            self.frame = self._store[:height]
            index = np.full((world, self.pad_rows), height, np.int64)
            for k in range(world):
                rows = rows_of_part(height, stripe_rows, world, k)
                index[k, :len(rows)] = rows
            self.index = torch.as_tensor(index.reshape(-1), dtype=torch.long, device=device)
            if world > 1:
                self._stack = torch.zeros((world, self.pad_rows, width, 3), dtype=torch.uint8, device=device)
                self.bufs = [self._stack[k] for k in range(world)]
            else:
                self.tile = self.frame

    def run(self):
        """Call on every rank after its tile has been rendered into self.tile."""
        if self.world == 1:
            return self.frame
        self.dist.gather(self.tile, self.bufs if self.rank == 0 else None, dst=0)
        if self.rank != 0:
            return None
        self._store.index_copy_(0, self.index, self._stack.view(-1, self._stack.shape[2], 3))
        return self.frame
