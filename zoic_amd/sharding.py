"""Multi-GPU layout of the hot path: one process per GPU, the frame's samples split into contiguous, tile-aligned
slabs of the global ray index, no data-path collective while rays are generated (every ray depends only on its own
sample and on read-only tables).  Per-ray retry streams are keyed by the GLOBAL ray index, so a sharded frame is
bit-identical to the single-GPU frame.  The one exchange is the gather of the finished ray slabs on a root rank
(torch.distributed: RCCL over xGMI on GPUs, gloo on CPU in the tests).

The gather is pipelined against the trace: a rank cuts its slab into sub-launches of about `chunk_bytes`
of payload, queued alternately on two compute streams (the end of one sub-launch -- its drain and the listed kernel over
its work list -- runs under the next one's trace); as soon as sub-launch k is queued, its 28-byte payload (origin, dir,
weight -- the flag word stays home) is posted to the root on the communication stream while sub-launch k+1 traces.  xGMI is point to point: every
peer -> root transfer rides its own link, so the root ingests on up to 7 links at once; all sends/recvs of one round
are posted as ONE batch_isend_irecv group (a single ncclGroupStart/End on RCCL).
"""

TILE = 256  # rays per workgroup tile; slabs are aligned to it
PAYLOAD_FLOATS = 7  # ox oy oz dx dy dz weight: what SURVEY 8(e) gathers (28 B/ray); column 7 of a record is the flag word


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


def chunks_of_slab(lo, hi, chunk_rays, tile=TILE):
    """Cut [lo, hi) into pieces of about chunk_rays rays (tile aligned, the last one ragged)."""
    chunk_rays = max(tile, (chunk_rays // tile) * tile)
    out, a = [], lo
    while a < hi:
        b = min(a + chunk_rays, hi)
        out.append((a, b))
        a = b
    return out


def gather_rays(rays, n_total, dist, dst=0, tile=TILE):
    """Blocking whole-slab gather (round-1 form; kept for the CPU tests and as the un-overlapped baseline):
    every rank's (n_r, k) tensor lands in an (n_total, k) tensor on `dst`; None elsewhere."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    slabs = all_slabs(n_total, world, tile)
    if rank == dst:
        full = torch.empty((n_total,) + tuple(rays.shape[1:]), dtype=rays.dtype, device=rays.device)
        lo, hi = slabs[dst]
        full[lo:hi] = rays
        ops = [dist.P2POp(dist.irecv, full[a:b], r) for r, (a, b) in enumerate(slabs) if r != dst and b > a]
        for q in (dist.batch_isend_irecv(ops) if ops else []):
            q.wait()
        return full
    lo, hi = slabs[rank]
    if hi > lo:
        for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, rays.contiguous(), dst)]):
            q.wait()
    return None


class ShardedFrame:
    """One frame of n_total samples rendered by `world` ranks in ray-index slabs and gathered on `dst`.

    generate(chunk)        -> callable that produces the (m, 8) ray records of global rays [a, b) on this rank
                              (the HIP path on a GPU; the tests plug the oracle in on CPU)
    The root's `full` tensor is (n_total, 7) payload.  Reusable across steps (buffers are allocated once)."""

    def __init__(self, n_total, dist, device, generate, dst=0, chunk_bytes=None, tile=TILE, payload_floats=PAYLOAD_FLOATS,
                 min_chunk_bytes=64 << 20, chunks_per_slab=4, sparse=False):
        import torch
        self.torch, self.dist, self.device, self.generate, self.dst = torch, dist, device, generate, dst
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self.n_total, self.k = n_total, payload_floats
        # sparse=True (= ZOIC_FRAME_PAYLOAD_SPARSE of the C-ABI): only the rays with weight != 0 travel -- per chunk a live bit per
        # ray and the compacted rows; rows of weight-0 rays arrive as zeros (their origin / direction, the reference's partial state
        # of the last try, and their try counts stay on the rank that traced them).  A chunk's size is then only known to its
        # sender: every round starts with the peers' live counts (one small message each), the root sizes its receives from them.
        # sparse="auto": the layout is chosen from the camera -- the first gathered run is dense, the root then looks at the frame it holds
        # and every later run ships sparse only if at least AUTO_SPARSE_ZERO_WEIGHT of the rays had weight 0 (C5: 79 % -> sparse, 4.5x fewer
        # bytes into the root; C3: 0.07 % -> dense: the sparse layout's per-chunk count round and expansion pass cost more than its
        # headers save -- bench.py measured 12.1 against 23.3 Grays/s on one GPU).  One 8-byte broadcast, once.
        self.sparse_auto = sparse == "auto"
        self.auto_decided = not self.sparse_auto
        self.zero_weight_fraction = None      # what the root measured (auto mode; every rank holds it after the decision)
        self.sparse = bool(sparse) and not self.sparse_auto
        self.root_bytes = 0          # bytes received by the root in the last run(gather=True)
        self.slabs = all_slabs(n_total, self.world, tile)
        if chunk_bytes is None:
            # A sub-launch has ~0.1 ms of fixed cost (start-up + the tail of its unluckiest rays) against ~0.03 ms of
            # trace per 64 MB of payload, so chunks are as large as overlap allows: a slab is cut into `chunks_per_slab`
            # pieces, never smaller than 64 MB of payload (SURVEY 8e's floor).
            slab_bytes = 4 * payload_floats * max(b - a for a, b in self.slabs)
            chunk_bytes = max(min_chunk_bytes, (slab_bytes + chunks_per_slab - 1) // chunks_per_slab)
        self.chunk_bytes = chunk_bytes
        chunk_rays = max(tile, chunk_bytes // (4 * payload_floats))
        if self.world == 1:
            # nothing to gather, hence nothing to overlap: the slab is ONE launch.  Sub-launches are not free -- each pays its
            # start and its drain, and a contiguous quarter of a frame is more uniform than the frame (C5: the corner rows are
            # dead pixels, bound by HBM, the middle rows by instruction issue; one launch interleaves them through its eight
            # partition cursors, four row-wise sub-launches run them one after the other: 70 -> 61 Grays/s on one GPU)
            chunk_rays = max(chunk_rays, n_total)
        # every rank cuts its slab the same way, so the root knows each peer's message sizes without a handshake
        self.chunks = [chunks_of_slab(a, b, chunk_rays, tile) for a, b in self.slabs]
        self.rounds = max(len(c) for c in self.chunks)
        lo, hi = self.slabs[self.rank]
        self.full = torch.empty((n_total, self.k), dtype=torch.float32, device=device) if self.rank == dst else None
        # staging for the payload of the chunks in flight (two buffers: chunk k is on the wire while k+1 is packed)
        biggest = max((b - a for a, b in self.chunks[self.rank]), default=0)
        self.stage = [torch.empty((biggest, self.k), dtype=torch.float32, device=device) for _ in range(2)] if self.rank != dst else None
        self.cuda = device.type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=device) if self.cuda else None
        # Sub-launches alternate between TWO compute streams: a Kolb launch ends with a drain (the pool's last rays at a few
        # lanes per pass) and, in the decision-safe mode, with the listed kernel over its work list (a latency floor of
        # ~0.1 ms); back to back on one stream, four sub-launches pay that four times (13-14 % of a C4 / C5 slab on one GPU in
        # round 2).  On alternating streams sub-launch k + 1 traces under the end of sub-launch k (every launch owns its
        # work cursors and work list: capi.cpp launch slots).
        self.compute_streams = [torch.cuda.Stream(device=device) for _ in range(2)] if self.cuda else None
        self.slots = 3          # generate() may rotate this many record buffers: chunk k may only overwrite chunk k - 3's
        self.slot_free = [None] * self.slots

    def _run_sparse(self):
        """The gather with only the live rays on the wire.  Per round: every rank queues its chunk's trace; peers pack (live bits +
        compacted rows), send the count, then the packed chunk; the root receives the counts, sizes and posts the receives, expands."""
        import contextlib
        torch, dist = self.torch, self.dist
        mine = self.chunks[self.rank]
        caller = torch.cuda.current_stream(self.device) if self.cuda else None
        if self.cuda:
            for cs in self.compute_streams:
                cs.wait_stream(caller)
        self.root_bytes = 0
        if self.rank == self.dst:
            self.full.zero_()      # rows of weight-0 rays
        for k in range(self.rounds):
            cs = self.compute_streams[k & 1] if self.cuda else None
            with (torch.cuda.stream(cs) if self.cuda else contextlib.nullcontext()):
                rec = None
                if k < len(mine):
                    a, b = mine[k]
                    rec = self.generate(a, b)
                if self.rank == self.dst:
                    if rec is not None:
                        pay = rec[:, :self.k]
                        # the root's own slab in the same convention: a weight-0 row is seven exact +0 (a product with the mask
                        # would leave NaN for a NaN / Inf component of a dead ray's partial state and -0 for a negative one)
                        self.full[a:b].copy_(torch.where(pay[:, 6:7] != 0, pay, torch.zeros_like(pay)))
                    peers = [r for r in range(self.world) if r != self.dst and k < len(self.chunks[r])]
                    counts = {r: torch.zeros(1, dtype=torch.int64, device=self.device) for r in peers}
                    for q in (dist.batch_isend_irecv([dist.P2POp(dist.irecv, counts[r], r) for r in peers]) if peers else []):
                        q.wait()
                    bufs, ops = {}, []
                    for r in peers:
                        ra, rb = self.chunks[r][k]
                        live = int(counts[r].item())
                        bufs[r] = (torch.empty(((rb - ra + 7) // 8,), dtype=torch.uint8, device=self.device),
                                   torch.empty((live, self.k), dtype=torch.float32, device=self.device))
                        ops.append(dist.P2POp(dist.irecv, bufs[r][0], r))
                        if live:
                            ops.append(dist.P2POp(dist.irecv, bufs[r][1], r))
                        self.root_bytes += bufs[r][0].numel() + 4 * self.k * live + 8
                    for q in (dist.batch_isend_irecv(ops) if ops else []):
                        q.wait()
                    for r in peers:
                        ra, rb = self.chunks[r][k]
                        bits, rows = bufs[r]
                        live_mask = ((bits.to(torch.int32)[:, None] // (2 ** torch.arange(8, device=self.device, dtype=torch.int32))[None, :]) % 2).reshape(-1)[: rb - ra].bool()
                        self.full[ra:rb][live_mask] = rows
                elif rec is not None:
                    pay = rec[:, :self.k]
                    live_mask = pay[:, 6] != 0
                    rows = pay[live_mask].contiguous()
                    m = b - a
                    padded = torch.zeros(((m + 7) // 8) * 8, dtype=torch.uint8, device=self.device)
                    padded[:m] = live_mask.to(torch.uint8)
                    bits = (padded.reshape(-1, 8).to(torch.int32) * (2 ** torch.arange(8, device=self.device, dtype=torch.int32))[None, :]).sum(1).to(torch.uint8)
                    count = torch.tensor([rows.shape[0]], dtype=torch.int64, device=self.device)
                    for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, count, self.dst)]):
                        q.wait()
                    ops = [dist.P2POp(dist.isend, bits, self.dst)]
                    if rows.shape[0]:
                        ops.append(dist.P2POp(dist.isend, rows, self.dst))
                    for q in dist.batch_isend_irecv(ops):
                        q.wait()
        if self.cuda:
            for cs in self.compute_streams:
                caller.wait_stream(cs)
        return self.full

    AUTO_SPARSE_ZERO_WEIGHT = 0.25

    def _decide_layout(self):
        """sparse="auto", after the first dense gather: the root holds the whole frame -- its zero-weight fraction picks the layout of
        every later run, for every rank (one broadcast of two numbers)."""
        torch, dist = self.torch, self.dist
        t = torch.zeros(2, dtype=torch.float64, device=self.device)
        if self.rank == self.dst:
            if self.cuda:
                torch.cuda.current_stream(self.device).synchronize()
            zero = float((self.full[:, 6] == 0).to(torch.float64).mean())
            t[0], t[1] = zero, 1.0 if zero >= self.AUTO_SPARSE_ZERO_WEIGHT else 0.0
        dist.broadcast(t, src=self.dst)
        self.zero_weight_fraction, self.sparse = float(t[0].item()), bool(t[1].item() != 0.0)
        self.auto_decided = True

    def run(self, gather=True):
        """Render this rank's slab chunk by chunk; with gather=True the payload of chunk k travels while chunk k+1 is
        traced.  Returns the root's full (n_total, 7) tensor (None on the other ranks, or when gather=False)."""
        if gather and self.world > 1:
            if not self.auto_decided:
                full = self._run_dense(True)
                self._decide_layout()
                return full
            if self.sparse:
                return self._run_sparse()
        return self._run_dense(gather)

    def _run_dense(self, gather):
        import contextlib
        torch, dist = self.torch, self.dist
        mine = self.chunks[self.rank]
        pending = []
        stage_busy = [None, None]   # the sends still reading each staging buffer
        caller = torch.cuda.current_stream(self.device) if self.cuda else None
        if self.cuda:
            for cs in self.compute_streams:
                cs.wait_stream(caller)           # the samples (and whatever else the caller queued) come first
        for k in range(self.rounds):
            cs = self.compute_streams[k & 1] if self.cuda else None
            with (torch.cuda.stream(cs) if self.cuda else contextlib.nullcontext()):
                rec = None
                if k < len(mine):
                    a, b = mine[k]
                    if self.cuda and self.slot_free[k % self.slots] is not None:
                        cs.wait_event(self.slot_free[k % self.slots])   # the record buffer's previous chunk has been packed
                    rec = self.generate(a, b)           # queued on this chunk's compute stream, asynchronous on a GPU
                ops = []
                if gather and self.rank == self.dst:
                    if rec is not None:
                        self.full[a:b].copy_(rec[:, :self.k])
                    for r in range(self.world):
                        if r != self.dst and k < len(self.chunks[r]):
                            ra, rb = self.chunks[r][k]
                            ops.append(dist.P2POp(dist.irecv, self.full[ra:rb], r))
                elif gather and rec is not None:
                    buf = self.stage[k & 1][: b - a]
                    if stage_busy[k & 1] is not None:
                        for q in stage_busy[k & 1]:         # the send that used this staging buffer two rounds ago
                            q.wait()                        # (on a GPU: this chunk's compute stream waits for it)
                        pending.remove(stage_busy[k & 1])   # a gloo send request must be waited for exactly once
                        stage_busy[k & 1] = None
                    buf.copy_(rec[:, :self.k])             # 32-byte records -> 28-byte payload, on the chunk's compute stream
                    ops.append(dist.P2POp(dist.isend, buf, self.dst))
                if self.cuda and rec is not None:
                    ev = self.slot_free[k % self.slots] or torch.cuda.Event()
                    ev.record(cs)
                    self.slot_free[k % self.slots] = ev
            if ops:
                if self.cuda:
                    # post the round on the communication stream, behind this chunk's compute work: RCCL moves chunk k
                    # while the next sub-launch runs on the other compute stream
                    self.comm_stream.wait_stream(cs)
                    with torch.cuda.stream(self.comm_stream):
                        reqs = dist.batch_isend_irecv(ops)
                else:
                    reqs = dist.batch_isend_irecv(ops)
                pending.append(reqs)
                if self.rank != self.dst:
                    stage_busy[k & 1] = reqs
        for reqs in pending:
            for q in reqs:
                q.wait()
        if gather and self.rank == self.dst:
            self.root_bytes = 4 * self.k * (self.n_total - (self.slabs[self.dst][1] - self.slabs[self.dst][0]))
        if self.cuda:
            for cs in self.compute_streams:
                caller.wait_stream(cs)
            if gather:
                caller.wait_stream(self.comm_stream)
        return self.full if gather else None
