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


def gather_rays(rays, n_total, dist, dst=0, tile=TILE):
    """Gather every rank's (n_r, k) ray-record tensor on `dst`; returns (n_total, k) there, None elsewhere.
    Slabs may differ in size by one tile, so the exchange is grouped point-to-point (the RCCL-friendly form of a
    gatherv: each peer->root transfer rides its own xGMI link)."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    slabs = all_slabs(n_total, world, tile)
    if rank == dst:
        full = torch.empty((n_total,) + tuple(rays.shape[1:]), dtype=rays.dtype, device=rays.device)
        lo, hi = slabs[dst]
        full[lo:hi] = rays
        reqs = []
        for r, (a, b) in enumerate(slabs):
            if r == dst or b <= a:
                continue
            reqs.append(dist.irecv(full[a:b], src=r))  # contiguous row range: received in place
        for q in reqs:
            q.wait()
        return full
    lo, hi = slabs[rank]
    if hi > lo:
        dist.send(rays.contiguous(), dst=dst)
    return None
