#!/usr/bin/env python3
"""bench.py -- camera rays/sec of zoic's lens hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--precision fast|strict]

One "step" = one pass of camera_create_ray over one full frame of synthetic samples (config C3 by default:
F_2.0_DOUBLE_GAUSS + image-based bokeh sampler, 3840x2160x16spp = 132,710,400 samples), samples already resident
in HBM, rays written to HBM.  N>1 (launched by torch.distributed.run, one rank per GPU): every rank renders its
own frame of the same size (weak scaling; the path shards by independent samples, no data-path collective);
`--gather` additionally collects every rank's ray slab on rank 0 over RCCL inside the timed region.

Rank 0 prints ONE JSON line (see DESIGN.md "measurement" for every field).
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
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E peak (MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3
VALU_PEAK_TWIPS = 0.95   # T wave64 VALU instr/s, whole chip: plain f32 ops / mixed streams at >= 4 waves/SIMD (tools/ubench/op_rate.hip on MI355X, profiles/ubench_r01.txt)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--precision", default=os.environ.get("ZOIC_BENCH_PRECISION", "fast"), choices=["fast", "strict"])
    ap.add_argument("--rays", type=int, default=0, help="override the per-GPU sample count (default: the config's full frame)")
    ap.add_argument("--gather", action="store_true", help="gather all ray slabs on rank 0 (RCCL) inside the timed region")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank renders a full frame; strong: ONE frame split into per-rank slabs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline leg")
    return ap.parse_args()


def cpu_baseline(cfg_name, seconds):
    """The oracle (plain-C restatement of zoic.cpp) timed on this box's host cores on a bounded sample of the same
    workload.  Baseline only -- never part of `value`."""
    import numpy as np
    import oracle
    from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh, ray_rng_states, synthetic_samples
    c = CONFIGS[cfg_name]
    oc = oracle.OracleCamera()
    if c["bokeh"]:
        oc.set_bokeh_image(hexagon_bokeh())
    oc.update(**camera_params(cfg_name))
    cores = os.cpu_count() or 1
    # strided slab of the frame: every 64th pixel row block, so the sample sees the whole field
    probe_n = 1 << 18
    base = (c["width"] * (c["height"] // 2)) * c["spp"]
    s = synthetic_samples(probe_n, c["width"], c["height"], c["spp"], seed=1, ray_index_base=base)
    st = ray_rng_states(probe_n, seed=1, ray_index_base=base)
    t0 = time.perf_counter()
    oc.create_rays(s, rng_states=st)
    one = probe_n / (time.perf_counter() - t0)
    n = int(min(max(one * cores * seconds * 0.7, probe_n), 64e6))
    s = synthetic_samples(n, c["width"], c["height"], c["spp"], seed=1, ray_index_base=base)
    st = ray_rng_states(n, seed=1, ray_index_base=base)
    t0 = time.perf_counter()
    oc.create_rays(s, rng_states=st, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": round(n / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d samples of %s starting at the frame's middle row, all %d host threads, per-ray retry streams; "
                      "1 thread: %.3f Mrays/s" % (n, cfg_name, cores, one / 1e6)}


def parity_probe(cam, cfg_name, precision):
    """Direction RMSE / decision flips of the benchmarked mode against the oracle on a 256K-sample slab."""
    import numpy as np
    import oracle
    from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh, ray_rng_states, synthetic_samples
    c = CONFIGS[cfg_name]
    n = 1 << 18
    base = (c["width"] * (c["height"] // 2)) * c["spp"]
    s = synthetic_samples(n, c["width"], c["height"], c["spp"], seed=1, ray_index_base=base)
    st = ray_rng_states(n, seed=1, ray_index_base=base)
    oc = oracle.OracleCamera()
    if c["bokeh"]:
        oc.set_bokeh_image(hexagon_bokeh())
    oc.update(**camera_params(cfg_name))
    ref = oc.create_rays(s, rng_states=st, threads=os.cpu_count() or 1)
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


def main():
    args = parse_args()
    import torch
    from zoic_amd import PRECISION_FAST, PRECISION_STRICT, ZoicCamera
    from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh, ray_count

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
        dist.init_process_group(backend="nccl", device_id=dev)

    from zoic_amd.sharding import gather_rays, slab_for_rank
    cfg = CONFIGS[args.config]
    frame = args.rays or ray_count(args.config)
    if args.scaling == "strong":
        lo, hi = slab_for_rank(frame, rank, world)
        n, base, n_total = hi - lo, lo, frame
    else:
        n, base, n_total = frame, rank * frame, frame * world   # rank r renders frame r (distinct ray indices)
    cam = ZoicCamera(device=local_rank)
    if cfg["bokeh"]:
        cam.set_bokeh_image(hexagon_bokeh())
    cam.update(**camera_params(args.config))
    cam.set_precision(PRECISION_FAST if args.precision == "fast" else PRECISION_STRICT)

    # inputs resident in HBM before the timed region
    samples = cam.generate_samples(n, cfg["width"], cfg["height"], cfg["spp"], seed=1, ray_index_base=base)
    out = dict(rays=torch.empty((n, 8), dtype=torch.float32, device=dev))
    def step():
        cam.create_rays(samples, ray_index_base=base, out=out)
        if args.gather and world > 1:
            gather_rays(out["rays"], n_total, dist, dst=0)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        cam.create_rays(samples, ray_index_base=base, out=out)
        ev[k][1].record()
        if args.gather and world > 1:
            gather_rays(out["rays"], n_total, dist, dst=0)
    torch.cuda.synchronize()
    if dist:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1)

    if rank == 0:
        total_rays = n_total * args.steps
        value = total_rays / elapsed / 1e6
        counters = cam.counters()
        done = counters["succesRays"] + counters["vignettedRays"]
        achieved = ALGO_BYTES_PER_RAY * n / (kernel_ms * 1e-3) / 1e9
        traffic, valu = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and not args.rays:
            try:  # PMC numbers of the last committed rocprofv3 run of this (config, mode): bytes and VALU instructions
                ent = json.load(open(tpath)).get("%s_%s" % (args.config, args.precision), {})
                traffic = ent.get("hbm_bytes_per_launch")
                if ent.get("lane_instr_per_ray"):
                    rate = ent["lane_instr_per_ray"] / 64.0 * n / (kernel_ms * 1e-3) / 1e12
                    valu = {"bound": "valu-issue", "achieved": round(rate, 4), "peak": VALU_PEAK_TWIPS, "unit": "T wave64-instr/s",
                            "frac": round(rate / VALU_PEAK_TWIPS, 4), "lane_instr_per_ray": round(ent["lane_instr_per_ray"], 1),
                            "lane_utilisation": round(ent.get("valu_thread_util", 0.0), 3),
                            "note": "peak = measured VALU issue rate of plain-f32 / mixed streams (tools/ubench/op_rate.hip); instruction count from profiles/ PMC"}
            except Exception:
                traffic, valu = None, None
        line = {
            "metric": "camera rays/sec (Mrays/s), 4K x 16spp Kolb lens trace; ray-dir RMSE vs CPU ref",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.config, cfg["desc"]), "rays_per_gpu_per_step": n,
                       "precision_mode": args.precision,
                       "parallelism": ("independent frames per GPU (dp%d)" if args.scaling == "weak" else "one frame in %d ray-index slabs") % world,
                       "gather": bool(args.gather and world > 1)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel": ("kolb_refill_%s_kernel" % args.precision) if cfg["params"]["lensModel"] == 1 else "thin_rays_kernel",
                         "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_ray": ALGO_BYTES_PER_RAY,
                         "note": "Kolb path is FP32-VALU bound (DESIGN.md); HBM fraction reported per the bench contract"},
            "zero_weight_frac": round(counters["vignettedRays"] / max(done, 1), 5),
        }
        if valu:
            line["valu_roofline"] = valu
        if not args.no_parity:
            line["parity"] = parity_probe(cam, args.config, args.precision)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.config, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
