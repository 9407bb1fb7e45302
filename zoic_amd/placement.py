"""Where in HBM a frame's two streams lie (MI355X; DESIGN.md section 5, `profiles/ab_r06/placement_*.txt`).

A frame reads one stream (16-byte samples) and writes another (32-byte zoic_ray records).  On an MI355X the rate of the
image-sampler configuration depends on WHICH device allocations hold the two -- not on their addresses inside an allocation:
every offset of one 240 GB allocation falls into one of a few classes of multi-GB regions (the driver's physical placement),
and a frame whose sample buffer and ray buffer lie in the SAME class runs at 40.9 Grays/s, in different classes at 45.5-46.2:
the DRAM read latency behind the L2 is 13 % higher when reads and writes mix in one class (TCC_EA0_RDREQ_LEVEL / RDREQ 1638
against 1444 cycles).  User space cannot ask where an allocation lies, but it can measure: a renderer allocates its frame
buffers once, so `pick_frame_buffers` allocates a few candidates of each, times the camera's own kernel over every pair for
a few frames and keeps the fastest pair (the others are freed).  The rays are the same bits whichever pair is kept.

This is host-side set-up around the C-ABI's device-pointer call (zoic_create_rays_device): nothing in the kernels changes.
"""
import time


def _pair_rate(torch, camera, samples, out, ray_index_base, steps, warmup):
    n = samples.shape[0]
    for _ in range(warmup):
        camera.create_rays(samples, ray_index_base=ray_index_base, out=out)
    torch.cuda.synchronize(samples.device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        camera.create_rays(samples, ray_index_base=ray_index_base, out=out)
    e1.record()
    torch.cuda.synchronize(samples.device)
    return n * steps / max(e0.elapsed_time(e1) * 1e-3, 1e-9)


def pick_frame_buffers(camera, samples, candidates=3, steps=4, warmup=2, ray_index_base=0, memory_fraction=0.5, min_samples=1 << 22):
    """samples: the frame's (n, 4) float32 device tensor.  Returns (samples, out, info): the pair of (sample buffer, ray buffer
    = dict(rays=(n, 8) float32)) on which `camera` ran fastest out of `candidates` allocations of each, and what was measured
    (`info["rates_mrays_s"][i][j]`: sample buffer i x ray buffer j; pair (0, 0) is what a plain torch.empty would have given).

    Falls back to one candidate -- no probe -- when the frame is too short to time (`min_samples`) or when `candidates` copies of
    both buffers would not fit into `memory_fraction` of the free device memory."""
    import torch
    if not samples.is_cuda:
        raise ValueError("pick_frame_buffers works on device tensors")
    dev = samples.device
    n = samples.shape[0]
    per_pair = samples.numel() * samples.element_size() + n * 32
    k = max(1, int(candidates))
    free = torch.cuda.mem_get_info(dev)[0]
    while k > 1 and (k * per_pair - samples.numel() * samples.element_size()) > memory_fraction * free:
        k -= 1
    if n < min_samples:
        k = 1
    t0 = time.perf_counter()
    pads = []
    sbufs, obufs = [samples], []
    for i in range(k):
        if i > 0:
            # odd-sized spacers: consecutive large allocations of one size tend to come out of one physical region
            pads.append(torch.empty((7 + 11 * i) * (1 << 20), dtype=torch.uint8, device=dev))
            sbufs.append(samples.clone())
            pads.append(torch.empty((5 + 3 * i) * (1 << 20), dtype=torch.uint8, device=dev))
        obufs.append(dict(rays=torch.empty((n, 8), dtype=torch.float32, device=dev)))
    info = {"candidates": k, "pairs": k * k, "steps": steps}
    if k == 1:
        info["note"] = "no probe (short frame or not enough free memory for candidates)"
        return samples, obufs[0], info
    rates = [[_pair_rate(torch, camera, s, o, ray_index_base, steps, warmup) for o in obufs] for s in sbufs]
    rates[0][0] = max(rates[0][0], _pair_rate(torch, camera, sbufs[0], obufs[0], ray_index_base, steps, warmup))   # the first pair was timed cold: once more, warm
    best = max(((rates[i][j], i, j) for i in range(k) for j in range(k)))
    _, bi, bj = best
    info.update(rates_mrays_s=[[round(r / 1e6, 1) for r in row] for row in rates], chosen=[bi, bj],
                first_pair_mrays_s=round(rates[0][0] / 1e6, 1), chosen_pair_mrays_s=round(best[0] / 1e6, 1),
                slowest_pair_mrays_s=round(min(min(row) for row in rates) / 1e6, 1))
    s_keep, o_keep = sbufs[bi], obufs[bj]
    del sbufs, obufs, pads
    torch.cuda.empty_cache()          # hand the losing candidates back to the driver
    info["seconds"] = round(time.perf_counter() - t0, 3)
    return s_keep, o_keep, info
