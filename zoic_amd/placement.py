"""Where in HBM a frame's two streams lie (MI355X; DESIGN.md section 5, `profiles/ab_r06/placement_*.txt`).

A frame reads one stream (16-byte samples) and writes another (32-byte zoic_ray records).  On an MI355X the rate of the
image-sampler configuration depends on WHICH device allocations hold the two -- not on their addresses inside an allocation:
every offset of one 240 GB allocation falls into one of a few classes of multi-GB regions (the driver's physical placement),
and a frame whose sample buffer and ray buffer lie in the SAME class runs at 40.9 Grays/s, in different classes at 45.5-46.2:
the DRAM read latency behind the L2 is 13 % higher when reads and writes mix in one class (TCC_EA0_RDREQ_LEVEL / RDREQ 1638
against 1444 cycles).  User space cannot ask where an allocation lies, but it can measure: a renderer allocates its frame
buffers once, so `pick_frame_buffers` allocates candidate ray buffers one after another, times the camera's own kernel on each for a few
frames until it has seen both rates, and keeps the fastest (the others are freed).  The rays are the same bits on any buffer.

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


def pick_frame_buffers(camera, samples, candidates=8, steps=4, warmup=2, ray_index_base=0, memory_fraction=0.5, min_samples=1 << 22,
                       spread_stop=1.09):
    """samples: the frame's (n, 4) float32 device tensor.  Returns (samples, out, info): `out` = dict(rays=(n, 8) float32) is the ray
    buffer on which `camera` ran this frame fastest out of up to `candidates` allocations, and `info` what was measured
    (`info["rates_mrays_s"][j]`: ray buffer j against the sample buffer; j = 0 is what a plain torch.empty would have given).

    The relation is between the two buffers' classes of regions, so only the ray buffer is searched.  Candidates are allocated one
    after another and ALL kept until the end -- a fresh process's allocations come out of the device memory region after region,
    and adjacent ones usually share a class (one box: 40.9-41.5 on all sixteen pairs of four adjacent candidates of each; in one
    scan the first 28 GB of an allocation were one class, in another all of 72 GB) -- and the search stops as soon as two rates differ
    by `spread_stop` (the rates are two-valued, 12 % apart on the image-sampler frame, with values in between for a buffer that straddles two classes: the faster class has been seen), at `candidates`, or when the candidates fill
    `memory_fraction` of the device memory that was free.  From the fifth candidate on a growing spacer allocation (4, 8, 16 ... ray buffers, within the same budget) is put in front of each.  The losers are freed.

    One candidate -- no probe -- when the frame is too short to time (`min_samples`) or a second ray buffer does not fit."""
    import torch
    if not samples.is_cuda:
        raise ValueError("pick_frame_buffers works on device tensors")
    dev = samples.device
    n = samples.shape[0]
    ray_bytes = n * 32
    budget = memory_fraction * torch.cuda.mem_get_info(dev)[0]
    kmax = max(1, int(candidates))
    if n < min_samples:
        kmax = 1
    t0 = time.perf_counter()
    obufs, pads, rates = [], [], []
    info = {"steps": steps}
    held = 0                                  # bytes this search holds: candidates + spacers
    while len(obufs) < kmax and held + ray_bytes <= max(budget, ray_bytes):
        j = len(obufs)
        if j > 0:   # an odd-sized allocation in between: consecutive allocations of one size tend to be carved out of one block
            pads.append(torch.empty((5 + 6 * j) * (1 << 20), dtype=torch.uint8, device=dev))
        if j >= 4:
            # four adjacent candidates alike: reach further -- a spacer of 4, 8, 16, ... ray buffers (kept until the end, cut to the budget) before
            # the next one.  One class of regions was seen to run for 28 GB of an allocation, on another box for all of 72 GB.
            spacer = int(min((1 << (j - 2)) * ray_bytes, budget - held - ray_bytes)) >> 21 << 21
            if spacer > 0:
                pads.append(torch.empty(spacer, dtype=torch.uint8, device=dev))
                held += spacer
        obufs.append(dict(rays=torch.empty((n, 8), dtype=torch.float32, device=dev)))
        held += ray_bytes
        if kmax == 1:
            break
        rates.append(_pair_rate(torch, camera, samples, obufs[j], ray_index_base, steps, warmup))
        if j == 1:   # the first candidate was timed cold: once more, warm
            rates[0] = max(rates[0], _pair_rate(torch, camera, samples, obufs[0], ray_index_base, steps, warmup))
        if j >= 1 and max(rates) >= spread_stop * min(rates):
            break
    info["candidates"] = len(obufs)
    info["held_gb"] = round(held / 2 ** 30, 1)          # candidates + spacers at the search's end
    if len(rates) < 2:
        info["note"] = "no probe (short frame, one candidate asked for, or no room for a second ray buffer)"
        return samples, obufs[0], info
    bj = max(range(len(rates)), key=lambda j: rates[j])
    info.update(rates_mrays_s=[round(r / 1e6, 1) for r in rates], chosen=bj, first_pair_mrays_s=round(rates[0] / 1e6, 1),
                chosen_pair_mrays_s=round(rates[bj] / 1e6, 1), slowest_pair_mrays_s=round(min(rates) / 1e6, 1),
                both_classes_seen=bool(max(rates) >= spread_stop * min(rates)))
    o_keep = obufs[bj]
    del obufs, pads
    torch.cuda.empty_cache()          # hand the losing candidates back to the driver
    info["seconds"] = round(time.perf_counter() - t0, 3)
    return samples, o_keep, info
