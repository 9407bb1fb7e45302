"""Multi-GPU layout of the hot path: one process per GPU, the frame's samples split into contiguous, tile-aligned
slabs of the global ray index, no data-path collective while rays are generated (every ray depends only on its own
sample and on read-only tables).  Per-ray retry streams are keyed by the GLOBAL ray index, so a sharded frame is
bit-identical to the single-GPU frame.  The one optional exchange is a gather of the finished ray slabs on a root
rank (torch.distributed: RCCL over xGMI on GPUs, gloo on CPU in the tests).
"""

TILE = 256  # rays per workgroup tile; slabs are aligned to it


def slab_for_rank(n_total, rank, world, tile=TILE):
    """[begin, end) of rank's slab: contiguous, tile aligned, sizes differ by at most one tile."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    tiles = (n_total + tile - 1) // tile
    lo = (tiles * rank) // world
    hi = (tiles * (rank + 1)) // world
    return min(lo * tile, n_total), min(hi * tile, n_total)


def all_slabs(n_total, world, tile=TILE):
    return [slab_for_rank(n_total, r, world, tile) for r in range(world)]


def gather_planes(planes, flags, n_total, dist, dst=0, tile=TILE):
    """Gather every rank's (7, n_r) ray planes and (n_r,) flags on `dst`; returns ((7, n_total), (n_total,)) there,
    (None, None) elsewhere.  Slabs may differ in size by one tile, so the exchange is grouped point-to-point
    (the RCCL-friendly form of a gatherv: each peer->root transfer rides its own xGMI link)."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    slabs = all_slabs(n_total, world, tile)
    if rank == dst:
        full = torch.empty((7, n_total), dtype=planes.dtype, device=planes.device)
        fl = torch.empty((n_total,), dtype=flags.dtype, device=flags.device)
        lo, hi = slabs[dst]
        full[:, lo:hi] = planes
        fl[lo:hi] = flags
        reqs, bufs = [], []
        for r, (a, b) in enumerate(slabs):
            if r == dst or b <= a:
                continue
            pb = torch.empty((7, b - a), dtype=planes.dtype, device=planes.device)
            fb = torch.empty((b - a,), dtype=flags.dtype, device=flags.device)
            reqs.append(dist.irecv(pb, src=r))
            reqs.append(dist.irecv(fb, src=r))
            bufs.append((a, b, pb, fb))
        for q in reqs:
            q.wait()
        for a, b, pb, fb in bufs:
            full[:, a:b] = pb
            fl[a:b] = fb
        return full, fl
    lo, hi = slabs[rank]
    if hi > lo:
        dist.send(planes.contiguous(), dst=dst)
        dist.send(flags.contiguous(), dst=dst)
    return None, None
