#!/usr/bin/env python3
"""bench.py -- camera rays/sec of zoic's lens hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--precision fast|strict]

One "step" = one pass of camera_create_ray over one full frame of synthetic samples (config C3 by default:
F_2.0_DOUBLE_GAUSS + image-based bokeh sampler, 3840x2160x16spp = 132,710,400 samples), samples already resident
in HBM, rays written to HBM.  Rank 0 prints ONE JSON line:

  value / ms_per_step   the headline workload, K timed steps between barriers.  N>1 (torch.distributed.run, one rank per
                        GPU): every rank renders its own frame of the headline size -- the path shards by independent
                        samples with no data-path collective, so this is weak scaling;
  roofline              the dominant kernel against the HBM roofline (per the bench contract) + valu_roofline, the bound
                        that actually binds the Kolb kernels (instruction issue);
  configs               the other BASELINE.json configs at their true sizes (C1, C2, C4, C5 fast + C3 strict), each
                        with rate, kernel time, roofline and a parity block -- parity cases, measured so that every
                        number quoted in DESIGN.md is a driver record;
  sharded_frame         BASELINE.json configs 4/5 as north_star states them: ONE C4 / C5 frame cut into ray-index slabs
                        over the N ranks, compute-only and with the RCCL gather of the 28-byte payload on rank 0
                        (chunked, overlapped with the trace);
  host_path             the PCIe-inclusive rate of the host-buffer entry point (never `value`);
  cpu_baseline          the oracle timed on this box's host cores (N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_RAY = 48  # 16 B sample in + one 32 B ray record out (28 B origin/dir/weight of SURVEY 8(d) + the 4 B flag word)
SURVEY_BYTES_PER_RAY = 44  # SURVEY 8(d)'s figure (16 + 28): reported next to it
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E peak (MI355X_MICROARCH.md)
VALU_PEAK_ARCH_TWIPS = 1.2288   # 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md "Wave scheduling")
VALU_PEAK_MEASURED_TWIPS = 0.95  # plain-f32 / mixed streams at >= 4 waves/SIMD, clocks as they sag under load (tools/ubench/op_rate.hip, profiles/ubench_r01.txt)
CPU_SLAB_RAYS = 16_588_800       # the fixed slab (= config 2's full size) the CPU legs are quoted on


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--precision", default=os.environ.get("ZOIC_BENCH_PRECISION", "fast"), choices=["fast", "unchecked", "strict"],
                    help="fast = ZOIC_PRECISION_FAST (decision-safe), unchecked = ZOIC_PRECISION_FAST_UNCHECKED (round 1's fast), strict = bit-exact")
    ap.add_argument("--rays", type=int, default=0, help="override the per-GPU sample count of the headline workload (experiments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config array")
    ap.add_argument("--no-sharded", action="store_true", help="skip the sharded-frame (north-star configs 4/5) measurement")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--only-headline", action="store_true", help="= --no-configs --no-sharded --no-host-path --no-cpu-baseline --no-parity")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target wall time of each CPU baseline leg")
    ap.add_argument("--gather-chunk-mb", type=int, default=0, help="payload MB per gather chunk (0: a quarter of a slab, at least 64 MB)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------- cameras
def make_camera(cfg_name, precision, device):
    from zoic_amd import PRECISION_FAST, PRECISION_FAST_UNCHECKED, PRECISION_STRICT, ZoicCamera
    from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh
    cam = ZoicCamera(device=device)
    if CONFIGS[cfg_name]["bokeh"]:
        cam.set_bokeh_image(hexagon_bokeh())
    cam.update(**camera_params(cfg_name))
    cam.set_precision({"fast": PRECISION_FAST, "unchecked": PRECISION_FAST_UNCHECKED, "strict": PRECISION_STRICT}[precision])
    return cam


def make_oracle(cfg_name):
    import oracle
    from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh
    oc = oracle.OracleCamera()
    if CONFIGS[cfg_name]["bokeh"]:
        oc.set_bokeh_image(hexagon_bokeh())
    oc.update(**camera_params(cfg_name))
    return oc


# ------------------------------------------------------------------------------------------------- CPU legs
def cpu_baseline(cfg_name, seconds):
    """The oracle (plain-C restatement of zoic.cpp) timed on this box's host cores.  Two legs:
    (i) ONE thread drawing retries from the sequential process-global xor128 -- the configuration the reference is
    validated in; (ii) all host cores with per-ray retry streams (the reference's shared stream is a data race).
    Both on the first rays of the fixed 16,588,800-ray slab of this config, bounded to about `seconds` each.
    Baseline only -- never part of `value`."""
    from zoic_amd.workloads import CONFIGS, ray_rng_states, synthetic_samples
    c = CONFIGS[cfg_name]
    oc = make_oracle(cfg_name)
    cores = os.cpu_count() or 1
    slab = min(CPU_SLAB_RAYS, c["width"] * c["height"] * c["spp"])
    gen = lambda n: synthetic_samples(n, c["width"], c["height"], c["spp"], seed=1, ray_index_base=0)  # noqa: E731
    probe_n = 1 << 17
    s = gen(probe_n)
    t0 = time.perf_counter()
    oc.create_rays(s)
    probe_rate = probe_n / (time.perf_counter() - t0)
    n1 = int(min(slab, max(probe_n, probe_rate * seconds)))
    s = gen(n1)
    oc.reset_rng()
    t0 = time.perf_counter()
    oc.create_rays(s)                                            # rng_states=None: the sequential global stream
    one = n1 / (time.perf_counter() - t0)
    n_all = int(min(slab, max(probe_n, one * cores * seconds * 0.6)))
    s = gen(n_all)
    st = ray_rng_states(n_all, seed=1, ray_index_base=0)
    t0 = time.perf_counter()
    oc.create_rays(s, rng_states=st, threads=cores)
    dt = time.perf_counter() - t0
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(n_all / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "first %d rays of %s's %d-ray slab, all %d host threads, per-ray retry streams" % (n_all, cfg_name, slab, cores),
            "one_thread_value": round(one / 1e6, 4), "one_thread_rays": n1, "one_thread_rng": "sequential process-global xor128 (zoic.cpp:647-652)",
            "slab_rays": slab, "all_cores_rays": n_all, "cpu_model": model,
            "calibration": "BASELINE.md section 2 (true reference, survey container, 1 thread): DOUBLE_GAUSS+LUT 0.75-0.8, TESSAR 1.0, "
                           "FISHEYE 0.8, PETZVAL 0.6-0.7, thin lens 20 Mrays/s"}


def parity_probe(cam, cfg_name, precision):
    """Direction RMSE / decision flips of the benchmarked mode against the oracle on a 256K-sample slab from the middle
    of the frame."""
    import numpy as np
    from zoic_amd.workloads import CONFIGS, ray_rng_states, synthetic_samples
    c = CONFIGS[cfg_name]
    n = 1 << 18
    base = (c["width"] * (c["height"] // 2)) * c["spp"]
    s = synthetic_samples(n, c["width"], c["height"], c["spp"], seed=1, ray_index_base=base)
    st = ray_rng_states(n, seed=1, ray_index_base=base)
    ref = make_oracle(cfg_name).create_rays(s, rng_states=st, threads=os.cpu_count() or 1)
    got = cam.create_rays(s, ray_index_base=base)
    same = (got["flags"] == ref["flags"])
    live = same & (ref["weight"] != 0)
    dd = (got["dir"][:, live].astype(np.float64) - ref["dir"][:, live].astype(np.float64))
    do = (got["origin"][:, live].astype(np.float64) - ref["origin"][:, live].astype(np.float64))
    return {"vs": "oracle (CPU restatement of zoic.cpp)", "samples": n, "mode": precision,
            "dir_rmse": float(np.sqrt((dd ** 2).sum(0).mean())) if live.any() else 0.0,
            "origin_rmse": float(np.sqrt((do ** 2).sum(0).mean())) if live.any() else 0.0,
            "decision_flip_frac": float((~same).mean()),
            "bit_exact": bool(np.array_equal(got["planes"].view(np.uint32), ref["planes"].view(np.uint32))
                              and np.array_equal(got["flags"], ref["flags"]))}


# ------------------------------------------------------------------------------------------------- GPU legs
def pmc_entry(cfg_name, precision):
    """HBM bytes and VALU instruction counts of the last committed rocprofv3 PMC run of this (config, mode)."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(tpath)).get("%s_%s" % (cfg_name, precision)) or None
    except Exception:
        return None


def roofline_block(cfg_name, precision, n, kernel_ms, thin):
    achieved = ALGO_BYTES_PER_RAY * n / (kernel_ms * 1e-3) / 1e9
    ent = pmc_entry(cfg_name, precision)
    kernel = "thin_rays_kernel" if thin else {"fast": "kolb_refill_guard_kernel (+ heavy-list and strict redo kernels of the launch)",
                                              "unchecked": "kolb_refill_fast_kernel", "strict": "kolb_refill_strict_kernel"}[precision]
    roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": ent.get("hbm_bytes_per_launch") if ent else None,
            "traffic_source": (ent.get("source", "committed rocprofv3 run") + " -- replayed from the committed PMC run, not measured in this process")
            if ent else "no committed PMC run for this (config, mode)",
            "kernel": kernel, "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_ray": ALGO_BYTES_PER_RAY,
            "frac_at_survey_44B": round(SURVEY_BYTES_PER_RAY * n / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "note": "HBM fraction per the bench contract; the Kolb kernels are bound by VALU instruction issue (valu_roofline)" if not thin
            else "thin lens is HBM-bound"}
    valu = None
    if ent and ent.get("lane_instr_per_ray"):
        rate = ent["lane_instr_per_ray"] / 64.0 * n / (kernel_ms * 1e-3) / 1e12
        valu = {"bound": "valu-issue", "achieved": round(rate, 4), "unit": "T wave64-instr/s",
                "peak": VALU_PEAK_ARCH_TWIPS, "frac": round(rate / VALU_PEAK_ARCH_TWIPS, 4),
                "peak_measured": VALU_PEAK_MEASURED_TWIPS, "frac_of_measured": round(rate / VALU_PEAK_MEASURED_TWIPS, 4),
                "lane_instr_per_ray": round(ent["lane_instr_per_ray"], 1), "lane_utilisation": round(ent.get("valu_thread_util", 0.0), 3),
                "note": "peak = 1024 SIMDs x 2.4 GHz / 2 cycles; peak_measured = tools/ubench/op_rate.hip on MI355X (clocks sag to ~1.8 GHz "
                        "under dense FMA); instruction count replayed from the committed PMC run"}
    return roof, valu


def time_frame(torch, cam, cfg, n, base, steps, warmup, dev, dist=None, local_rank=0):
    """`steps` launches of one frame of n samples (resident in HBM), bracketed by barrier + synchronize on both sides.
    Returns (elapsed seconds: max over ranks, mean kernel ms by HIP events on the launch stream)."""
    samples = cam.generate_samples(n, cfg["width"], cfg["height"], cfg["spp"], seed=1, ray_index_base=base)
    out = dict(rays=torch.empty((n, 8), dtype=torch.float32, device=dev))
    for _ in range(warmup):
        cam.create_rays(samples, ray_index_base=base, out=out)
    torch.cuda.synchronize()
    if dist:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
        ev[k][0].record()                      # torch's current stream == the stream the kernel is launched on
        cam.create_rays(samples, ray_index_base=base, out=out)
        ev[k][1].record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(steps, 1)
    del samples, out
    return elapsed, kernel_ms


def config_entry(torch, cfg_name, precision, dev, local_rank, steps, warmup, parity):
    from zoic_amd.workloads import CONFIGS, ray_count
    cfg = CONFIGS[cfg_name]
    n = ray_count(cfg_name)
    cam = make_camera(cfg_name, precision, local_rank)
    elapsed, kernel_ms = time_frame(torch, cam, cfg, n, 0, steps, warmup, dev)
    counters = cam.counters()
    done = counters["succesRays"] + counters["vignettedRays"]
    thin = cfg["params"]["lensModel"] == 0
    roof, valu = roofline_block(cfg_name, precision, n, kernel_ms, thin)
    ent = {"config": cfg_name, "workload": cfg["desc"], "precision_mode": precision, "rays": n, "steps": steps,
           "value": round(n * steps / elapsed / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(elapsed / steps * 1e3, 4),
           "roofline": roof, "zero_weight_frac": round(counters["vignettedRays"] / max(done, 1), 5)}
    if valu:
        ent["valu_roofline"] = valu
    if parity:
        ent["parity"] = parity_probe(cam, cfg_name, precision)
    cam.close()
    torch.cuda.empty_cache()
    return ent


def sharded_frame_entry(torch, dist, cfg_name, dev, rank, world, local_rank, steps, chunk_mb):
    """north_star configs 4/5: ONE frame in ray-index slabs over the ranks; compute-only and gather-inclusive rates."""
    from zoic_amd.sharding import PAYLOAD_FLOATS, ShardedFrame
    from zoic_amd.workloads import CONFIGS, ray_count
    cfg = CONFIGS[cfg_name]
    n_total = ray_count(cfg_name)
    cam = make_camera(cfg_name, "fast", local_rank)
    chunk_bytes = (chunk_mb << 20) if chunk_mb else None
    frame = ShardedFrame(n_total, dist if world > 1 else None, dev, None, dst=0, chunk_bytes=chunk_bytes)
    lo, hi = frame.slabs[rank]
    samples = cam.generate_samples(hi - lo, cfg["width"], cfg["height"], cfg["spp"], seed=1, ray_index_base=lo)
    biggest = max((b - a for a, b in frame.chunks[rank]), default=0)
    recs = [dict(rays=torch.empty((biggest, 8), dtype=torch.float32, device=dev)) for _ in range(3)]
    turn = [0]

    def generate(a, b):   # sub-launch over global rays [a, b) of this rank's slab; three record buffers rotate
        o = recs[turn[0] % 3]
        turn[0] += 1
        view = dict(rays=o["rays"][: b - a])
        cam.create_rays(samples[a - lo:b - lo], ray_index_base=a, out=view)
        return view["rays"]
    frame.generate = generate

    def timed(gather):
        frame.run(gather=gather)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            frame.run(gather=gather)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el
    t_compute = timed(False)
    t_gather = timed(True) if world > 1 else None
    ent = {"config": cfg_name, "workload": cfg["desc"], "rays": n_total, "n_gpus": world, "steps": steps, "scaling": "strong",
           "parallelism": "one frame in %d ray-index slabs, %d sub-launches per slab" % (world, len(frame.chunks[rank])),
           "compute_only": {"value": round(n_total * steps / t_compute / 1e6, 2), "unit": "Mrays/s", "ms_per_frame": round(t_compute / steps * 1e3, 4)}}
    if t_gather is not None:
        payload = 4 * PAYLOAD_FLOATS
        ent["with_gather"] = {"value": round(n_total * steps / t_gather / 1e6, 2), "unit": "Mrays/s", "ms_per_frame": round(t_gather / steps * 1e3, 4),
                              "gather": "28-byte payload of every peer slab -> rank 0, %d MB chunks, batch_isend_irecv on a second stream under the trace" % (frame.chunk_bytes >> 20),
                              "bytes_into_root_per_frame": payload * (n_total - (frame.slabs[0][1] - frame.slabs[0][0])),
                              "root_ingest_gb_s": round(payload * (n_total - (frame.slabs[0][1] - frame.slabs[0][0])) * steps / t_gather / 1e9, 1)}
        # rank 0 holds the gathered frame: a peer's chunk must equal what this GPU computes for the same global rays
        full = frame.run(gather=True)
        torch.cuda.synchronize()
        if rank == 0:
            a, b = frame.chunks[world - 1][-1]
            s = cam.generate_samples(b - a, cfg["width"], cfg["height"], cfg["spp"], seed=1, ray_index_base=a)
            mine = cam.create_rays(s, ray_index_base=a)["rays"][:, :PAYLOAD_FLOATS]
            ent["with_gather"]["bit_identical_to_single_gpu"] = bool(torch.equal(mine.contiguous().view(torch.int32), full[a:b].view(torch.int32)))
    else:
        ent["with_gather"] = None
        ent["note"] = "one GPU: nothing to gather"
    cam.close()
    del samples, recs, frame
    torch.cuda.empty_cache()
    return ent


def host_path_entry(cam, cfg):
    """zoic_create_rays_host end to end (H2D + trace + D2H over PCIe), pageable and page-locked caller buffers."""
    import numpy as np
    from zoic_amd import PinnedArray, _capi
    from zoic_amd.workloads import synthetic_samples
    n = 1 << 24
    s = synthetic_samples(n, cfg["width"], cfg["height"], cfg["spp"], seed=1)
    res = {"rays": n, "note": "PCIe-inclusive; never reported as `value`"}

    def timed(sp, rp):
        for _ in range(4):     # the PCIe link and the copy engines take a few transfers to reach their steady rate
            cam._check(cam._lib.zoic_create_rays_host(cam._h, n, sp, None, 0, rp))
        t0 = time.perf_counter()
        for _ in range(3):
            cam._check(cam._lib.zoic_create_rays_host(cam._h, n, sp, None, 0, rp))
        dt = (time.perf_counter() - t0) / 3
        return {"value": round(n / dt / 1e6, 1), "unit": "Mrays/s", "pcie_gb_s_both_directions": round(48 * n / dt / 1e9, 1)}
    rays = np.empty(n, dtype=_capi.RAY_DTYPE)
    res["pageable"] = timed(s.ctypes.data, rays.ctypes.data)
    ps, pr = PinnedArray((n, 4), np.float32), PinnedArray((n,), _capi.RAY_DTYPE)
    ps.array[:] = s
    res["pinned"] = timed(ps.array.ctypes.data, pr.array.ctypes.data)
    ps.free()
    pr.free()
    return res


def main():
    args = parse_args()
    if args.only_headline:
        args.no_configs = args.no_sharded = args.no_host_path = args.no_cpu_baseline = args.no_parity = True
    import torch
    from zoic_amd.workloads import CONFIGS, ray_count

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libzoic_amd has no CPU path")
    torch.cuda.set_device(local_rank)      # before the process group: RCCL binds the communicator to the current device
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("ZOIC_FORCE_DIST"):   # ZOIC_FORCE_DIST: exercise the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if world == 1:
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=dev)

    cfg = CONFIGS[args.config]
    frame = args.rays or ray_count(args.config)
    n, base, n_total = frame, rank * frame, frame * world        # rank r renders frame r (distinct global ray indices)
    cam = make_camera(args.config, args.precision, local_rank)
    elapsed, kernel_ms = time_frame(torch, cam, cfg, n, base, args.steps, args.warmup, dev, dist, local_rank)

    line = None
    if rank == 0:
        counters = cam.counters()
        done = counters["succesRays"] + counters["vignettedRays"]
        thin = cfg["params"]["lensModel"] == 0
        roof, valu = roofline_block(args.config, args.precision, n, kernel_ms, thin) if not args.rays else \
            roofline_block("none", args.precision, n, kernel_ms, thin)
        line = {
            "metric": "camera rays/sec (Mrays/s), 4K x 16spp Kolb lens trace; ray-dir RMSE vs CPU ref",
            "value": round(n_total * args.steps / elapsed / 1e6, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.config, cfg["desc"]), "rays_per_gpu_per_step": n,
                       "precision_mode": args.precision + (" (decision-safe: try counts, weights, flags and counters are the reference's)" if args.precision == "fast" else ""),
                       "parallelism": "independent frames per GPU (dp%d), no data-path collective" % world},
            "roofline": roof,
            "zero_weight_frac": round(counters["vignettedRays"] / max(done, 1), 5),
        }
        if valu:
            line["valu_roofline"] = valu
        if not args.no_parity:
            line["parity"] = parity_probe(cam, args.config, args.precision)
        if not args.no_host_path:
            line["host_path"] = host_path_entry(cam, cfg)
    cam.close()
    torch.cuda.empty_cache()

    if not args.no_configs and world == 1:
        ents = []
        for cname, prec, st, wu in (("C1", "fast", 20, 3), ("C2", "fast", 20, 3), ("C4", "fast", 10, 2), ("C5", "fast", 4, 1), ("C3", "strict", 8, 2)):
            if cname == args.config and prec == args.precision:
                continue
            ents.append(config_entry(torch, cname, prec, dev, local_rank, st, wu, not args.no_parity))
        line["configs"] = ents

    if not args.no_sharded:
        sh = []
        for cname, st in (("C4", 5), ("C5", 2)):
            e = sharded_frame_entry(torch, dist, cname, dev, rank, world, local_rank, st, args.gather_chunk_mb)
            if rank == 0:
                sh.append(e)
        if rank == 0:
            line["sharded_frame"] = sh

    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.config, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
