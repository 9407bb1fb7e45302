#!/usr/bin/env python3
"""bench.py -- camera rays/sec of zoic's lens hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--precision fast|strict]

One "step" = one pass of camera_create_ray over one full frame of synthetic samples (config C3 by default:
F_2.0_DOUBLE_GAUSS + image-based bokeh sampler, 3840x2160x16spp = 132,710,400 samples), samples already resident
in HBM, rays written to HBM.  Rank 0 prints ONE JSON line (kept under 6 KB: what a field means is said once, in `notes`):

  value / ms_per_step   N=1: the headline frame, K timed steps between barriers.  N>1 (one rank per GPU; `python bench.py --gpus N`
                        without a launcher re-executes itself under torch.distributed.run): north_star's number -- the SAME
                        frame rendered ONCE per step in N ray-index slabs INCLUDING the RCCL gather of the 28-byte payload on
                        rank 0 (strong scaling); compute_only, the weak-scaling figure (every rank its own frame, no
                        collective), the root's ingest rate against its (N-1) x 153 GB/s of xGMI and the single-process
                        zoic_frame_* path (rank 0 alone driving all N devices through the C-ABI) are printed beside it;
  roofline              bound = "valu" for the Kolb configs ("hbm" for the thin lens); achieved / peak / frac = the launch against
                        the HBM roofline at SURVEY 8(d)'s 44 B/ray (kernel time by HIP events on the launch stream), flop_frac =
                        oracle-counted FLOP/ray x rays/s against 157 TFLOP/s, valu_frac = VALU issue;
  parity                direction RMSE / decision flips of the benchmarked mode against the oracle;
  configs               the other BASELINE.json configs at their true sizes (C1, C2, C4, C5 fast + C3 strict);
  sharded_frame         BASELINE.json configs 4/5 as north_star states them: ONE C4 / C5 frame in ray-index slabs over the N
                        ranks, compute-only and with the RCCL gather of the 28-byte payload on rank 0;
  host_path             PCIe-inclusive rates of the host-buffer entry points (never `value`) and the per-sample call latency;
  cpu_baseline          the oracle timed on this box's host cores (N=1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RECORD_BYTES_PER_RAY = 48  # what the kernels move: 16 B sample in + one 32 B ray record out (SURVEY's 28 B + the 4 B flag word): frac48
ALGO_BYTES_PER_RAY = 44    # SURVEY 8(d)'s algorithmic figure (16 B in + 28 B origin/dir/weight out): roofline.achieved / frac
FP32_PEAK_TFLOPS = 157.0   # MI355X FP32 vector peak, FMA counted as 2
XGMI_LINK_GBS = 153.0      # one xGMI link, one direction (MI355X_MICROARCH.md): the root of a gather ingests on N-1 of them
NORTH_STAR_MRAYS_1GPU = 1000.0   # north_star: >= 1 Grays/s on one MI355X, >= 6x that at 8 GPUs
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E peak (MI355X_MICROARCH.md)
VALU_PEAK_ARCH_TWIPS = 1.2288   # 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md "Wave scheduling")
CPU_SLAB_RAYS = 16_588_800       # the fixed slab (= config 2's full size) the CPU legs are quoted on

NOTES = {
    "roofline": "two candidate bounds (SURVEY 8d): HBM at 44 B/ray (16 B sample + 28 B origin/dir/weight) and FP32 VALU at the FLOP/ray the "
                "oracle counts for the work the kernels cannot avoid (profiles/flop_model_r06.json `executed`: 106 x interface visits + 130 x tries, "
                "minus the tries proved away for dead pixels and retry-dead rays; flop_per_ray_as_written = the reference's loop as SURVEY prices it) "
                "against 157 TFLOP/s; bound = the lower ceiling in rays/s (ceilings_grays), achieved / peak / frac = binding_frac are ITS figures "
                "(<= 1 by construction), the other bound's are hbm_gb_s / hbm_frac / flop_frac; kernel_ms = the launch (main kernel + the listed kernel over its work list) by HIP events on the "
                "launch stream; frac48 = HBM fraction with the 32 B record the kernels really write; traffic, lane_instr, lane_util = the "
                "committed rocprofv3 PMC run of this (config, mode), profiles/pmc_traffic.json (FETCH_SIZE x2 + WRITE_SIZE) -- null when "
                "the kernel sources have changed since (csrc_sha16); valu_frac = wave64 VALU instr/s over 1024 SIMDs x 2.4 GHz / 2",
    "multi_gpu": "value = ONE headline frame per step in N ray-index slabs incl. the gather of the 28 B/ray payload on rank 0 (RCCL "
                 "batch_isend_irecv, chunks overlapped with the trace); root_ingest_frac = bytes into rank 0 per second over (N-1) x 153 "
                 "GB/s: a gather to ONE root is bounded by its links, 3.25 GB of a C3 frame >= 3.0 ms at 8 GPUs against 3.3 ms to render "
                 "the whole frame on one GPU, so speed-up over one GPU cannot exceed ~1x with the gather in -- the north-star target is "
                 "absolute (>= 6x 1 Grays/s at 8 GPUs = target_mrays_s); weak = every rank its own frame, no collective; "
                 "single_process_frame = rank 0 alone driving all N devices through zoic_frame_* (hipMemcpyPeerAsync gather)",
    "mode": "fast = ZOIC_PRECISION_FAST: f32 without the reference's scattered f64 intermediates, decisions at ill-conditioned "
            "interfaces re-taken in STRICT (flips = rays whose try count / weight differs from the oracle's); strict = bit-exact",
    "parity": "256 Ki samples from the middle of the frame against the oracle (CPU restatement of zoic.cpp); rmse over rays with "
              "identical history and weight != 0; north-star tolerance 1e-5",
    "sharded_frame": "one frame in ray-index slabs; on one GPU there is nothing to gather and a slab is ONE launch",
    "host_path": "zoic_create_rays_host / _arnold end to end over PCIe, 16.8 M samples; per_sample = zoic_camera_create_ray "
                 "(resident mailbox kernel) timed by tools/native/sample_latency.c; tile = zoic_tile_submit + _wait (resident tile server: no "
                 "launch) from render threads with a page-locked tile each, tools/native/tile_latency.c: aggregate Mrays/s over the wall clock, "
                 "p50 / p99 us per call; pcie_floor_us = 112 B/sample (28 in, 84 out) at 55 GB/s; launch_* = zoic_create_rays_arnold on "
                 "page-locked arrays, the call a tile replaces",
    "cpu_baseline": "the oracle on this box's host cores, page-touched buffers, >= 3 repetitions of >= 3 s; one_thread = the "
                    "sequential process-global xor128 (the configuration the reference is validated in); scaling_efficiency = all / "
                    "(one x cores); BASELINE.md: the true reference does 0.6-1.0 Mrays/s per thread",
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: WORLD_SIZE if a launcher set it, else 1)")
    ap.add_argument("--dry-launch", action="store_true", help="CPU self-test of the launcher: ranks meet over gloo, rank 0 prints the world it saw")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10, help="untimed launches before the timed ones (the first ~25 ms of launches after an idle gap run 5-15 %% slow: DESIGN section 5)")
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--precision", default="fast", choices=["fast", "unchecked", "strict"],
                    help="fast = ZOIC_PRECISION_FAST (decision-safe), unchecked = ZOIC_PRECISION_FAST_UNCHECKED, strict = bit-exact")
    ap.add_argument("--rays", type=int, default=0, help="override the per-GPU sample count of the headline workload (experiments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config array")
    ap.add_argument("--no-sharded", action="store_true", help="skip the sharded-frame (north-star configs 4/5) measurement")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--placement-candidates", type=int, default=8, help="ray-buffer allocations to choose the frame's from, at most (1: the plain first allocation; zoic_amd/placement.py)")
    ap.add_argument("--no-device-state", action="store_true", help="skip the second of frames with the clock / power sensors read beside it")
    ap.add_argument("--only-headline", action="store_true", help="= --no-configs --no-sharded --no-host-path --no-cpu-baseline --no-parity --no-device-state")
    ap.add_argument("--cpu-seconds", type=float, default=3.0, help="minimum wall time of one repetition of a CPU baseline leg")
    ap.add_argument("--no-sparse-leg", action="store_true", help="N > 1: skip the sparse gather's timing (it runs last, behind everything else)")
    ap.add_argument("--no-single-process", action="store_true", help="N > 1: skip the zoic_frame_* measurement (rank 0 driving all devices)")
    ap.add_argument("--sharded-timeout", type=int, default=240, help="N > 1: seconds everything behind the weak-scaling leg (gathers included) may take before the line is printed with what has been measured")
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
def usable_cores():
    """Threads this process can really run at once: the affinity mask, capped by the container's CPU quota (cgroup cpu.max;
    the GPU boxes of this pool grant 16 CPUs of a 256-thread host -- 256 threads would only time the throttle)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except Exception:  # noqa: BLE001
            continue
    return n


def cpu_baseline(cfg_name, seconds):
    """The oracle (plain-C restatement of zoic.cpp) timed on this box's host cores.  Two legs: ONE thread drawing
    retries from the sequential process-global xor128, and all host cores with per-ray retry streams (the reference's shared
    stream is a data race), dynamic 4096-ray chunks.  Both on the first rays of the fixed 16,588,800-ray slab of this config,
    into buffers whose pages have been touched, 3 repetitions of >= `seconds` each (best one reported).  Baseline only."""
    import numpy as np
    from zoic_amd.workloads import CONFIGS, ray_rng_states, synthetic_samples
    c = CONFIGS[cfg_name]
    oc = make_oracle(cfg_name)
    hw = os.cpu_count() or 1
    cores = usable_cores()
    slab = min(CPU_SLAB_RAYS, c["width"] * c["height"] * c["spp"])
    gen = lambda n: synthetic_samples(n, c["width"], c["height"], c["spp"], seed=1, ray_index_base=0)  # noqa: E731

    def leg(n, threads):
        s = gen(n)
        st = ray_rng_states(n, seed=1, ray_index_base=0) if threads > 1 else None
        out = (np.zeros((7, n), np.float32), np.zeros(n, np.uint8))
        run = lambda: oc.create_rays(s, rng_states=st, threads=threads, out=out)  # noqa: E731
        run()                                                   # touches every page, warms the caches, starts the clocks
        best, reps = 0.0, []
        for _ in range(3):
            calls, t0 = 0, time.perf_counter()
            while True:
                if threads == 1:
                    oc.reset_rng()
                run()
                calls += 1
                dt = time.perf_counter() - t0
                if dt >= seconds:
                    break
            reps.append(n * calls / dt)
            best = max(best, reps[-1])
        return best, reps
    probe_n = 1 << 16
    s = gen(probe_n)
    t0 = time.perf_counter()
    oc.create_rays(s)
    rate1 = probe_n / (time.perf_counter() - t0)
    n1 = int(min(slab, max(probe_n, rate1 * min(seconds, 1.0))))
    one, _ = leg(n1, 1)
    n_all = int(min(slab, max(1 << 20, one * cores * 0.25)))     # about a quarter of a second of the whole machine per call
    allc, reps = leg(n_all, cores)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(allc / 1e6, 2), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "first %d rays of %s's %d-ray slab per call, %d threads (CPU quota of this container; the host has %d)" % (n_all, cfg_name, slab, cores, hw),
            "one_thread": round(one / 1e6, 4), "one_thread_rays": n1, "scaling_efficiency": round(allc / (one * cores), 3),
            "repetitions": [round(r / 1e6, 1) for r in reps], "cpu": model}


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
    ref = make_oracle(cfg_name).create_rays(s, rng_states=st, threads=usable_cores())
    got = cam.create_rays(s, ray_index_base=base)
    same = (got["flags"] == ref["flags"])
    live = same & (ref["weight"] != 0)
    dd = (got["dir"][:, live].astype(np.float64) - ref["dir"][:, live].astype(np.float64))
    return {"rmse": float("%.3g" % np.sqrt((dd ** 2).sum(0).mean())) if live.any() else 0.0,
            "flips": float("%.3g" % (~same).mean()),
            "bit_exact": bool(np.array_equal(got["planes"].view(np.uint32), ref["planes"].view(np.uint32)) and np.array_equal(got["flags"], ref["flags"]))}


# ------------------------------------------------------------------------------------------------- GPU legs
def csrc_sha16():
    """Fingerprint of the kernel sources (every .hip / .hpp of zoic_amd/csrc): a PMC entry of profiles/pmc_traffic.json is only
    replayed when it was taken on these sources (tools/collect_profiles.py stamps it)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "zoic_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_entry(cfg_name, precision):
    """HBM bytes and VALU instruction counts of the committed rocprofv3 PMC run of this (config, mode) -- None (=> traffic null)
    when the entry is missing or was taken on other kernel sources than the ones this tree holds."""
    try:
        ent = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("%s_%s" % (cfg_name, precision)) or None
    except Exception:
        return None
    if ent is None or ent.get("csrc_sha16") != csrc_sha16():
        return None
    return ent


def flop_per_ray(cfg_name):
    """(as written, executed) algorithmic FLOP per ray, measured with the oracle's counters (tools/flop_model.py -> profiles/flop_model_r06.json):
    `as written` = every interface visit and try of the reference's loop (SURVEY 8d's model); `executed` = the same minus the tries the
    kernels PROVE away instead of running (dead pixels: 27 identical tries are one; retry-dead rays: 26 retries that die at interface 0,
    DESIGN 4.2).  The VALU bound is priced with `executed`: with `as written` a kernel that skips four fifths of C5's tries showed 245 % of
    the FP32 peak -- a fraction above 1 is a wrong model, not a fast machine.  SURVEY 8(d)'s probe figures when the file is missing."""
    for name in ("flop_model_r06.json", "flop_model_r04.json"):
        try:
            e = json.load(open(os.path.join(ROOT, "profiles", name)))[cfg_name]
            return float(e["flop_per_ray"]), float(e.get("executed_flop_per_ray", e["flop_per_ray"]))
        except Exception:
            continue
    f = {"C2": 1500.0, "C3": 1700.0, "C4": 1550.0, "C5": 3900.0}.get(cfg_name)
    return f, f


def roofline_block(cfg_name, precision, n, kernel_ms, thin):
    """two candidate bounds -- HBM at 44 B/ray and FP32 VALU at the oracle-counted FLOP/ray the kernels cannot avoid (`executed`,
    flop_per_ray above) -- both reported, and the block's achieved / peak / frac are those of the BINDING one (the lower ceiling in rays/s):
    the thin lens and C5 (four fifths of its rays are dead pixels the kernels settle in one try) are bound by HBM, C2-C4 by VALU.
    binding_frac == frac <= 1 by construction of both models; the other bound's figures sit beside it (hbm_* / flop_*), and the
    reference's as-written FLOP count (`flop_per_ray_as_written`, what round 5 priced) for comparison."""
    secs = kernel_ms * 1e-3
    hbm_gbs = ALGO_BYTES_PER_RAY * n / secs / 1e9
    hbm_frac = hbm_gbs / HBM_PEAK_GBS
    ent = pmc_entry(cfg_name, precision)
    fl_written, fl = (None, None) if thin else flop_per_ray(cfg_name)
    tf = fl * n / secs / 1e12 if fl else None
    flop_frac = tf / FP32_PEAK_TFLOPS if tf else None
    ceil_hbm = HBM_PEAK_GBS * 1e9 / ALGO_BYTES_PER_RAY / 1e9                      # Grays/s
    ceil_valu = FP32_PEAK_TFLOPS * 1e12 / fl / 1e9 if fl else None
    valu_binds = ceil_valu is not None and ceil_valu < ceil_hbm
    if valu_binds:
        roof = {"bound": "valu", "achieved": round(tf, 1), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(flop_frac, 4)}
    else:
        roof = {"bound": "hbm", "achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_frac, 4)}
    roof.update(binding_frac=roof["frac"], ceilings_grays={"hbm": round(ceil_hbm, 1), "valu": round(ceil_valu, 1) if ceil_valu else None},
                traffic=round(ent["hbm_bytes_per_launch"]) if ent else None, kernel_ms=round(kernel_ms, 4), bytes_per_ray=ALGO_BYTES_PER_RAY,
                hbm_gb_s=round(hbm_gbs, 1), hbm_frac=round(hbm_frac, 4),
                frac48=round(RECORD_BYTES_PER_RAY * n / secs / 1e9 / HBM_PEAK_GBS, 4),
                kernel="thin_rays_kernel" if thin else {"fast": "kolb_pool_guard_kernel + kolb_listed_kernel", "unchecked": "kolb_pool_fast_kernel",
                                                        "strict": "kolb_pool_strict_kernel"}[precision])
    if fl:
        roof.update(flop_per_ray=round(fl), flop_model="executed (oracle-counted visits and tries minus the tries the kernels prove away: profiles/flop_model_r06.json)",
                    tflops=round(tf, 1), flop_frac=round(flop_frac, 3), flop_per_ray_as_written=round(fl_written))
    if ent and ent.get("lane_instr_per_ray") and not thin:
        rate = ent["lane_instr_per_ray"] / 64.0 * n / secs / 1e12
        roof.update(lane_instr=round(ent["lane_instr_per_ray"]), lane_util=round(ent.get("valu_thread_util", 0.0), 3),
                    valu_frac=round(rate / VALU_PEAK_ARCH_TWIPS, 3))
        if ent.get("trans_per_ray") == ent.get("trans_per_ray") and ent.get("trans_per_ray") is not None:   # not NaN
            roof["trans_per_ray"] = round(ent["trans_per_ray"], 1)
    if ent:
        roof["pmc_source"] = "committed rocprofv3 PMC run of these kernel sources on the profile box (profiles/pmc_traffic.json, csrc_sha16-guarded): traffic, lane_instr, lane_util"
    else:
        roof["pmc"] = "no committed PMC run of these kernel sources (csrc_sha16 %s)" % csrc_sha16()
    assert roof["frac"] <= 1.0 + 1e-9, "a roofline fraction above 1 means the model is wrong: %r" % (roof,)
    return roof


def rank_barrier(dist, local_rank):
    """Barrier over the ranks: RCCL wants to know the device; the gloo rehearsal (ZOIC_BENCH_SAME_GPU) does not."""
    if dist.get_backend() == "nccl":
        dist.barrier(device_ids=[local_rank])
    else:
        dist.barrier()


def max_over_ranks(torch, dist, seconds, dev):
    t = torch.tensor([seconds], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def time_frame(torch, cam, cfg, n, base, steps, warmup, dev, dist=None, local_rank=0, keep=None, candidates=8):
    """`steps` launches of one frame of n samples (resident in HBM), bracketed by barrier + synchronize on both sides.
    Returns (elapsed seconds: max over ranks, mean kernel ms by HIP events on the launch stream).

    The frame's ray buffer is allocated the way a renderer would allocate it once per session: up to `candidates` allocations,
    the one the camera runs fastest on kept (zoic_amd/placement.py -- which allocations hold the sample stream and the ray
    stream is worth up to 12 % on the headline; `keep["placement"]` has every candidate's rate, the plain first one's included).
    `keep` (a dict) also receives the buffers instead of their being freed (the device_state leg runs on the same pair)."""
    from zoic_amd.placement import pick_frame_buffers
    samples = cam.generate_samples(n, cfg["width"], cfg["height"], cfg["spp"], seed=1, ray_index_base=base)
    samples, out, placement = pick_frame_buffers(cam, samples, candidates=candidates, ray_index_base=base)
    if keep is not None:
        keep["placement"] = placement
    for _ in range(warmup):
        cam.create_rays(samples, ray_index_base=base, out=out)
    torch.cuda.synchronize()
    if dist:
        rank_barrier(dist, local_rank)
    torch.cuda.synchronize()
    # ONE pair of HIP events around the K launches (on torch's current stream == the stream the kernels are launched on): a
    # timing event is a barrier packet of its own -- a pair per step put 15 us between consecutive frames (rocprofv3 kernel
    # trace, round 4), 2.7 % of a C2 frame
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for k in range(steps):
        cam.create_rays(samples, ray_index_base=base, out=out)
    ev1.record()
    torch.cuda.synchronize()
    if dist:
        rank_barrier(dist, local_rank)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        elapsed = max_over_ranks(torch, dist, elapsed, dev)
    kernel_ms = ev0.elapsed_time(ev1) / max(steps, 1)
    if keep is not None and keep.get("hold"):
        keep["samples"], keep["out"] = samples, out
    del samples, out
    return elapsed, kernel_ms


def device_sensor_files(torch, device_index):
    """hwmon files (shader clock, socket power) of the card whose PCI address is the HIP device's -- a box shows every card of
    its host, so `card0` is usually another GPU.  {} when sysfs has none (the leg is then skipped)."""
    import glob
    pr = torch.cuda.get_device_properties(device_index)
    addr = "%04x:%02x:%02x." % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        if addr not in os.path.realpath(card):
            continue
        for name, pat in (("sclk_mhz", "hwmon/hwmon*/freq1_input"), ("power_w", "hwmon/hwmon*/power1_input"), ("power_w", "hwmon/hwmon*/power1_average")):
            for f in glob.glob(os.path.join(card, pat)):
                out.setdefault(name, f)
    return out


def device_state_entry(torch, cam, cfg, n, base, dev, seconds=1.0, buffers=None):
    """Shader clock and socket power WHILE the headline kernel runs (round 6: every batch kernel runs against the socket's power
    limit, 1.31-1.34 kW, at 2.06-2.34 GHz instead of the 2.4 GHz the architectural peaks are quoted at -- DESIGN section 5).
    About `seconds` of back-to-back frames OUTSIDE the timed region, the sensors read every 20 ms from a second thread."""
    import threading
    files = device_sensor_files(torch, dev.index or 0)
    if not files:
        return None
    if buffers:
        samples, out = buffers
    else:
        samples = cam.generate_samples(n, cfg["width"], cfg["height"], cfg["spp"], seed=1, ray_index_base=base)
        out = dict(rays=torch.empty((n, 8), dtype=torch.float32, device=dev))
    rows, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            row = {}
            for k, f in files.items():
                try:
                    row[k] = float(open(f).read().split()[0]) * 1e-6
                except (OSError, ValueError, IndexError):
                    pass
            rows.append((time.perf_counter(), row))
            time.sleep(0.02)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.perf_counter()
    launches = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            cam.create_rays(samples, ray_index_base=base, out=out)
        launches += 10
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    stop.set()
    th.join()
    del samples, out
    loaded = [r for t, r in rows if t0 + 0.4 * (t1 - t0) <= t <= t1]      # the clock settles within ~0.3 s of load
    ent = {"launches": launches, "mrays_s": round(n * launches / (t1 - t0) / 1e6, 1), "samples": len(loaded)}
    for k in ("sclk_mhz", "power_w"):
        v = [r[k] for r in loaded if k in r]
        if v:
            ent[k] = round(sum(v) / len(v))
    return ent


def frame_stats(counters, n_done):
    return round(counters["vignettedRays"] / max(n_done, 1), 5)


def config_entry(torch, cfg_name, precision, dev, local_rank, steps, warmup, parity, device_state=False, candidates=8):
    from zoic_amd.workloads import CONFIGS, ray_count
    cfg = CONFIGS[cfg_name]
    n = ray_count(cfg_name)
    cam = make_camera(cfg_name, precision, local_rank)
    keep = {"hold": bool(device_state)}
    elapsed, kernel_ms = time_frame(torch, cam, cfg, n, 0, steps, warmup, dev, keep=keep, candidates=candidates)
    counters = cam.counters()
    thin = cfg["params"]["lensModel"] == 0
    roof = roofline_block(cfg_name, precision, n, kernel_ms, thin)
    ent = {"config": cfg_name, "mode": precision, "rays": n, "steps": steps, "value": round(n * steps / elapsed / 1e6, 1),
           "ms_per_step": round(elapsed / steps * 1e3, 4), "kernel_ms": roof["kernel_ms"], "hbm_frac": roof["hbm_frac"], "binding_frac": roof["frac"], "bound": roof["bound"],
           "zero_weight": frame_stats(counters, counters["succesRays"] + counters["vignettedRays"])}
    for k in ("traffic", "lane_instr", "lane_util", "valu_frac"):
        if roof.get(k) is not None:
            ent[k] = roof[k]
    pl = keep.get("placement") or {}
    if pl.get("rates_mrays_s"):
        ent["placement"] = {k: pl[k] for k in ("candidates", "first_pair_mrays_s", "chosen_pair_mrays_s", "slowest_pair_mrays_s")}
    state = device_state_entry(torch, cam, cfg, n, 0, dev, seconds=0.6, buffers=(keep.pop("samples"), keep.pop("out"))) if device_state else None
    if state:
        ent.update({k: state[k] for k in ("sclk_mhz", "power_w") if k in state})
    if parity:
        ent.update(parity_probe(cam, cfg_name, precision))
    cam.close()
    torch.cuda.empty_cache()
    return ent


def sharded_frame_entry(torch, dist, cfg_name, dev, rank, world, local_rank, steps, chunk_mb, precision="fast", warmup=1, ent=None, sparse_leg=False):
    """ONE frame of `cfg_name` in ray-index slabs over the ranks (north_star): compute-only and gather-inclusive
    rates, K timed steps each between barriers, max over ranks.  `ent` is filled as results arrive (a watchdog may print it)."""
    from zoic_amd.sharding import PAYLOAD_FLOATS, ShardedFrame
    from zoic_amd.workloads import CONFIGS, ray_count
    cfg = CONFIGS[cfg_name]
    n_total = ray_count(cfg_name)
    ent = {} if ent is None else ent
    cam = make_camera(cfg_name, precision, local_rank)
    chunk_bytes = (chunk_mb << 20) if chunk_mb else None
    frame = ShardedFrame(n_total, dist if world > 1 else None, dev, None, dst=0, chunk_bytes=chunk_bytes)
    lo, hi = frame.slabs[rank]
    samples = cam.generate_samples(hi - lo, cfg["width"], cfg["height"], cfg["spp"], seed=1, ray_index_base=lo)
    biggest = max((b - a for a, b in frame.chunks[rank]), default=0)
    recs = [dict(rays=torch.empty((biggest, 8), dtype=torch.float32, device=dev)) for _ in range(min(frame.slots, max(1, len(frame.chunks[rank]))))]
    turn = [0]

    def generate(a, b):   # sub-launch over global rays [a, b) of this rank's slab; the record buffers rotate (ShardedFrame.slots)
        o = recs[turn[0] % len(recs)]
        turn[0] += 1
        view = dict(rays=o["rays"][: b - a])
        cam.create_rays(samples[a - lo:b - lo], ray_index_base=a, out=view)
        return view["rays"]
    frame.generate = generate

    def timed(gather):
        for _ in range(max(1, warmup)):
            turn[0] = 0
            frame.run(gather=gather)
        torch.cuda.synchronize()
        if world > 1:
            rank_barrier(dist, local_rank)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            turn[0] = 0
            frame.run(gather=gather)
        torch.cuda.synchronize()
        if world > 1:
            rank_barrier(dist, local_rank)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            el = max_over_ranks(torch, dist, el, dev)
        return el
    ent.update(config=cfg_name, mode=precision, rays=n_total, n_gpus=world, steps=steps, scaling="strong", sub_launches_per_slab=len(frame.chunks[rank]))
    t_compute = timed(False)
    ent.update(compute_only=round(n_total * steps / t_compute / 1e6, 1), compute_ms=round(t_compute / steps * 1e3, 3))
    if world > 1:
        t_gather = timed(True)
        payload = 4 * PAYLOAD_FLOATS
        into_root = payload * (n_total - (frame.slabs[0][1] - frame.slabs[0][0]))
        ingest = into_root * steps / t_gather / 1e9
        ent.update(with_gather=round(n_total * steps / t_gather / 1e6, 1), gather_ms=round(t_gather / steps * 1e3, 3), chunk_mb=frame.chunk_bytes >> 20,
                   root_ingest_gb_s=round(ingest, 1), root_ingest_frac=round(ingest / ((world - 1) * XGMI_LINK_GBS), 3), root_bytes=into_root)
        # rank 0 holds the gathered frame: a peer's chunk must equal what this GPU computes for the same global rays
        full = frame.run(gather=True)
        torch.cuda.synchronize()
        if rank == 0:
            a, b = frame.chunks[world - 1][-1]
            s = cam.generate_samples(b - a, cfg["width"], cfg["height"], cfg["spp"], seed=1, ray_index_base=a)
            mine = cam.create_rays(s, ray_index_base=a)["rays"][:, :PAYLOAD_FLOATS]
            ent["bit_identical_to_single_gpu"] = bool(torch.equal(mine.contiguous().view(torch.int32), full[a:b].view(torch.int32)))
            # what ShardedFrame(sparse="auto") / ZOIC_FRAME_PAYLOAD_AUTO would pick for this camera (sparse iff >= 25 % of the frame's rays have weight 0)
            zero = float((full[:, 6] == 0).to(torch.float64).mean())
            ent.update(zero_weight_fraction=round(zero, 4), auto_layout="sparse" if zero >= ShardedFrame.AUTO_SPARSE_ZERO_WEIGHT else "dense")
        # LAST (everything above is safe if this leg misbehaves on its first meeting with real peers):
        # the gather with only the rays of weight != 0 on the wire (ShardedFrame(sparse=True): counts first, then bits + rows)
        try:
            if not sparse_leg:
                raise RuntimeError("skipped (--no-sparse-leg)")
            if dist.get_backend() != "nccl":   # the gloo rehearsal moves device tensors through the host at ~25 MB/s: one dense leg is rehearsal enough
                raise RuntimeError("skipped: not an RCCL run")
            frame.sparse = True
            t_sparse = timed(True)
            ent.update(with_sparse_gather=round(n_total * steps / t_sparse / 1e6, 1), sparse_gather_ms=round(t_sparse / steps * 1e3, 3), root_bytes_sparse=int(frame.root_bytes))
        except Exception as e:  # noqa: BLE001
            ent["sparse_failed"] = str(e)[:120]
        frame.sparse = False
    cam.close()
    del samples, recs, frame
    torch.cuda.empty_cache()
    return ent


def single_process_frame_entry(torch, cfg_name, precision, devices, steps, warmup, ent=None):
    """The same frame through the C-ABI's zoic_frame_* (csrc/frame.cpp): THIS process alone drives every device -- the form a
    C++ plug-in can call (the reference is one process).  Gather = hipMemcpyPeerAsync of the 28-byte payload to devices[0]."""
    from zoic_amd import FRAME_PAYLOAD, FRAME_PAYLOAD_SPARSE, PRECISION_FAST, PRECISION_FAST_UNCHECKED, PRECISION_STRICT, ZoicFrame
    from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh, ray_count
    cfg = CONFIGS[cfg_name]
    n = ray_count(cfg_name)
    ent = {} if ent is None else ent
    ent.update(config=cfg_name, devices=len(devices))
    frame = ZoicFrame(devices)
    if cfg["bokeh"]:
        frame.set_bokeh_image(hexagon_bokeh())
    frame.update(**camera_params(cfg_name))
    frame.set_precision({"fast": PRECISION_FAST, "unchecked": PRECISION_FAST_UNCHECKED, "strict": PRECISION_STRICT}[precision])
    frame.generate_samples(n, cfg["width"], cfg["height"], cfg["spp"], seed=1)
    root = torch.device("cuda", devices[0])
    out = torch.empty((n, 7), dtype=torch.float32, device=root)

    def timed(call):
        for _ in range(max(1, warmup)):
            call()
        frame.synchronize()
        torch.cuda.synchronize(root)
        t0 = time.perf_counter()
        for _ in range(steps):
            call()
        torch.cuda.synchronize(root)     # a render is ordered on the root's current stream ...
        frame.synchronize()              # ... and render_local on the frame's own
        return time.perf_counter() - t0
    with torch.cuda.device(root):
        t_local = timed(lambda: frame.render_local(n))
        ent.update(compute_only=round(n * steps / t_local / 1e6, 1), compute_ms=round(t_local / steps * 1e3, 3))
        t_gather = timed(lambda: frame.render(n, out=out, layout=FRAME_PAYLOAD))
        into_root = 28 * (n - (frame.slab(n, 0)[1] - frame.slab(n, 0)[0]))
        ent.update(with_gather=round(n * steps / t_gather / 1e6, 1), gather_ms=round(t_gather / steps * 1e3, 3),
                   root_bytes=sum(frame.lane_info(i)["bytes_to_root"] for i in range(len(devices))),
                   peer_access=[int(frame.lane_info(i)["peer_access_to_root"] and frame.lane_info(i)["peer_access_from_root"]) for i in range(len(devices))])
        if len(set(devices)) > 1:   # (a device listed twice copies to itself: no link involved)
            ingest = into_root * steps / t_gather / 1e9
            ent.update(root_ingest_gb_s=round(ingest, 1), root_ingest_frac=round(ingest / ((len(devices) - 1) * XGMI_LINK_GBS), 3))
        # the whole frame against ONE camera on the root device
        cam = make_camera(cfg_name, precision, devices[0])
        s = cam.generate_samples(n, cfg["width"], cfg["height"], cfg["spp"], seed=1)
        ref = cam.create_rays(s)["rays"]
        torch.cuda.synchronize(root)
        same = True
        step = 1 << 24
        for a in range(0, n, step):     # in pieces: no second whole-frame temporary
            same = same and bool(torch.equal(ref[a:a + step, :7].contiguous().view(torch.int32), out[a:a + step].view(torch.int32)))
        ent["bit_identical_to_single_gpu"] = same
        cam.close()
        del ref, s
        # the same gather with only the rays of weight != 0 on the wire (ZOIC_FRAME_PAYLOAD_SPARSE)
        t_sparse = timed(lambda: frame.render(n, out=out, layout=FRAME_PAYLOAD_SPARSE))
        ent.update(with_sparse_gather=round(n * steps / t_sparse / 1e6, 1), sparse_gather_ms=round(t_sparse / steps * 1e3, 3),
                   root_bytes_sparse=sum(frame.lane_info(i)["bytes_to_root"] for i in range(len(devices))))
    frame.close()
    del out
    torch.cuda.empty_cache()
    return ent


def frame_layout_entry(torch, cfg_name, precision, devices, steps=3):
    """what ZOIC_FRAME_PAYLOAD_AUTO picks for a camera, and what the two gather layouts move and cost there -- on an eighth of
    the config's frame taken from its middle rows (the whole C5 frame is 59 GB of payload).  With one GPU listed twice this is bytes and a
    code path, not a link measurement."""
    from zoic_amd import FRAME_PAYLOAD, FRAME_PAYLOAD_AUTO, FRAME_PAYLOAD_SPARSE, PRECISION_FAST, PRECISION_FAST_UNCHECKED, PRECISION_STRICT, ZoicFrame
    from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh, ray_count
    cfg = CONFIGS[cfg_name]
    n = ray_count(cfg_name) // 8 // 256 * 256
    base = ray_count(cfg_name) * 7 // 16 // 256 * 256
    ent = dict(config=cfg_name, rays=n, ray_index_base=base, devices=len(devices))
    frame = ZoicFrame(devices)
    if cfg["bokeh"]:
        frame.set_bokeh_image(hexagon_bokeh())
    frame.update(**camera_params(cfg_name))
    frame.set_precision({"fast": PRECISION_FAST, "unchecked": PRECISION_FAST_UNCHECKED, "strict": PRECISION_STRICT}[precision])
    frame.generate_samples(n, cfg["width"], cfg["height"], cfg["spp"], seed=1, ray_index_base=base)
    root = torch.device("cuda", devices[0])
    out = torch.empty((n, 7), dtype=torch.float32, device=root)
    with torch.cuda.device(root):
        for name, layout in (("dense", FRAME_PAYLOAD), ("sparse", FRAME_PAYLOAD_SPARSE)):
            frame.render(n, ray_index_base=base, out=out, layout=layout)
            torch.cuda.synchronize(root); frame.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                frame.render(n, ray_index_base=base, out=out, layout=layout)
            torch.cuda.synchronize(root); frame.synchronize()
            t = time.perf_counter() - t0
            ent[name] = dict(mrays_s=round(n * steps / t / 1e6, 1), ms=round(t / steps * 1e3, 3),
                             root_bytes=sum(frame.lane_info(i)["bytes_to_root"] for i in range(len(devices))))
        frame.update(**camera_params(cfg_name))     # AUTO decides per update: start it from scratch
        for _ in range(2):
            frame.render(n, ray_index_base=base, out=out, layout=FRAME_PAYLOAD_AUTO)
            torch.cuda.synchronize(root)
        layout, zero = frame.auto_layout()
        ent.update(auto_picks={FRAME_PAYLOAD: "dense", FRAME_PAYLOAD_SPARSE: "sparse", None: "undecided"}[layout], zero_weight_fraction=round(zero, 4) if zero is not None else None,
                   rule="sparse iff >= 25 % of the rays rendered since the update had weight 0")
    frame.close()
    del out
    torch.cuda.empty_cache()
    return ent


def per_sample_latency():
    """zoic_camera_create_ray as a render thread sees it (tools/native/sample_latency.c, built by __graft_entry__.build())."""
    exe = os.path.join(ROOT, "tools", "native", "sample_latency")
    lens = os.path.join(ROOT, "zoic_amd", "lenses", "double_gauss_f2.0.dat")
    if not os.path.exists(exe):
        return {"error": "tools/native/sample_latency not built"}
    res = {}
    for key, threads, precision in (("fast_1_thread", 1, 1), ("strict_1_thread", 1, 0), ("fast_16_threads", 16, 1)):
        try:
            out = subprocess.run([exe, lens, str(threads), "40000", str(precision), "1"], capture_output=True, text=True, timeout=120)
            j = json.loads(out.stdout.strip().splitlines()[-1])
            res[key] = {"median_us": j["median_us"], "p99_us": j["p99_us"], "calls_per_s": round(j["calls_per_s"])}
        except Exception as e:  # noqa: BLE001
            res[key] = {"error": repr(e)[:80]}
    return res


def tile_server_entry():
    """Bucket-sized calls through the resident tile server (zoic_tile_*, csrc/mailbox.hip) as render threads see them
    (tools/native/tile_latency.c, built by __graft_entry__.build()): 16 threads x 64 x 64 x 16 spp tiles (AtCameraInput / AtCameraOutput rows, and 16-byte samples in / zoic_ray records out: zoic_tile_set_inputs / _set_rows), 16 / 4 / 1 threads
    with 4096-sample tiles, FAST, double Gauss at f/2 without the image; beside them the launch-based call they replace (zoic_create_rays_arnold on
    page-locked arrays) on the same shapes."""
    exe = os.path.join(ROOT, "tools", "native", "tile_latency")
    lens = os.path.join(ROOT, "zoic_amd", "lenses", "double_gauss_f2.0.dat")
    if not os.path.exists(exe):
        return {"error": "tools/native/tile_latency not built"}
    res = {}
    legs = (("tile_16_threads_x_65536", ["16", "65536", "60", "1", "1", "0"]), ("tile_rays_16_threads_x_65536", ["16", "65536", "60", "1", "1", "0", "1", "1"]),
            ("tile_16_threads_x_4096", ["16", "4096", "600", "1", "1", "0"]),
            ("tile_4_threads_x_4096", ["4", "4096", "600", "1", "1", "0"]), ("tile_1_thread_x_4096", ["1", "4096", "1000", "1", "1", "0"]),
            ("tile_1_thread_x_256", ["1", "256", "1000", "1", "1", "0"]), ("tile_thin_lens_1_thread_x_4096", ["1", "4096", "1000", "1", "0", "0"]),
            ("launch_16_threads_x_65536", ["16", "65536", "30", "1", "1", "3"]), ("launch_1_thread_x_4096", ["1", "4096", "300", "1", "1", "3"]),
            # more render threads than this box's CPU quota admits at once: spinning (the default) against sleeping waits (zoic_camera_set_wait_mode;
            # under a quota every thread has a core, so yielding spins all the same: profiles/ab_r06/tile_threads_wait_modes.txt)
            ("tile_64_threads_x_4096_spin", ["64", "4096", "150", "1", "1", "0", "0", "0", "0"]), ("tile_64_threads_x_4096_sleep", ["64", "4096", "150", "1", "1", "0", "0", "0", "2"]),
            ("tile_128_threads_x_4096_sleep", ["128", "4096", "80", "1", "1", "0", "0", "0", "2"]),
            ("tile_rays_64_threads_x_65536_sleep", ["64", "65536", "20", "1", "1", "0", "1", "1", "2"]),
            # DEVICE buffers through the resident kernel (zoic_create_rays_device_resident: 16-byte samples in, 32-byte records out, no launch, no
            # PCIe rows) beside the launch-based device call + stream synchronise it replaces
            ("device_tile_1_thread_x_65536", ["1", "65536", "500", "1", "1", "4"]), ("device_tile_1_thread_x_4096", ["1", "4096", "1000", "1", "1", "4"]),
            ("device_tile_4_threads_x_65536", ["4", "65536", "300", "1", "1", "4"]),
            ("device_launch_1_thread_x_65536", ["1", "65536", "300", "1", "1", "5"]), ("device_launch_1_thread_x_4096", ["1", "4096", "300", "1", "1", "5"]))
    for key, a in legs:
        try:
            out = subprocess.run([exe, lens] + a, capture_output=True, text=True, timeout=180)
            j = json.loads(out.stdout.strip().splitlines()[-1])
            res[key] = {"mrays_s": j["mrays_s"], "p50_us": j["p50_us"], "p99_us": j["p99_us"], "pcie_floor_us": j["pcie_floor_us"]}
        except Exception as e:  # noqa: BLE001
            res[key] = {"error": repr(e)[:80]}
    return res


def host_path_entry(cam, cfg):
    """The host-buffer entry points end to end (H2D + trace + D2H over PCIe), pageable and page-locked caller buffers."""
    import numpy as np
    from zoic_amd import PinnedArray, _capi
    from zoic_amd.workloads import synthetic_samples
    import ctypes as C
    n = 1 << 24
    s = synthetic_samples(n, cfg["width"], cfg["height"], cfg["spp"], seed=1)
    res = {"rays": n}

    def timed(call, bytes_per_ray):
        for _ in range(4):     # the PCIe link and the copy engines take a few transfers to reach their steady rate
            cam._check(call())
        t0 = time.perf_counter()
        for _ in range(3):
            cam._check(call())
        dt = (time.perf_counter() - t0) / 3
        return {"value": round(n / dt / 1e6, 1), "pcie_gb_s": round(bytes_per_ray * n / dt / 1e9, 1)}
    rays = np.empty(n, dtype=_capi.RAY_DTYPE)
    res["pageable"] = timed(lambda: cam._lib.zoic_create_rays_host(cam._h, n, s.ctypes.data, None, 0, rays.ctypes.data), 48)
    ps, pr = PinnedArray((n, 4), np.float32), PinnedArray((n,), _capi.RAY_DTYPE)
    ps.array[:] = s
    res["pinned"] = timed(lambda: cam._lib.zoic_create_rays_host(cam._h, n, ps.array.ctypes.data, None, 0, pr.array.ctypes.data), 48)
    ps.free()
    pr.free()
    # AtCameraInput (28 B) -> AtCameraOutput (84 B) rows, page-locked caller arrays
    pi, po = PinnedArray((n, 7), np.float32), PinnedArray((n, 21), np.float32)
    pi.array[:] = 0.0
    pi.array[:, 0], pi.array[:, 1], pi.array[:, 4], pi.array[:, 5] = s[:, 0], s[:, 1], s[:, 2], s[:, 3]
    res["arnold_layout"] = timed(lambda: cam._lib.zoic_create_rays_arnold(cam._h, n, pi.array.ctypes.data_as(C.POINTER(_capi.CameraInput)),
                                                                          po.array.ctypes.data_as(C.POINTER(_capi.CameraOutput)), 0), 112)
    pi.free()
    po.free()
    return res


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def resolve_launch(args, env, device_count, argv):
    """How many ranks this invocation means, and whether it has to start them itself.

    Returns ("run", world) -- this process is one rank of `world` (a launcher exported WORLD_SIZE, or N = 1) -- or
    ("exec", command) -- `python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU.
    Raises SystemExit (non-zero) when --gpus contradicts the launcher's WORLD_SIZE or asks for more GPUs than the node has."""
    env_world = int(env["WORLD_SIZE"]) if env.get("WORLD_SIZE") else None
    want = args.gpus if args.gpus is not None else (env_world or 1)
    if want < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if env_world is not None:
        if want != env_world:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (want, env_world))
        if not args.dry_launch and device_count < env_world and env.get("ZOIC_BENCH_SAME_GPU") != "1":
            raise SystemExit("bench.py: %d ranks but only %d HIP device(s) visible: one rank per GPU" % (env_world, device_count))
        return "run", env_world
    if want == 1:
        return "run", 1
    if not args.dry_launch and device_count < want and env.get("ZOIC_BENCH_SAME_GPU") != "1":
        raise SystemExit("bench.py: --gpus %d but only %d HIP device(s) visible on this node (one rank per GPU; nothing was measured)" % (want, device_count))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(want), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return "exec", cmd


def dry_launch(world):
    """CPU self-test of the launcher (tests/test_sharding_cpu.py): the ranks torch.distributed.run started meet over gloo."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo")
    t = torch.ones(1)
    dist.all_reduce(t)
    if dist.get_rank() == 0:
        print(json.dumps({"dry_launch": True, "world": dist.get_world_size(), "expected": world, "sum": int(t.item())}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def flush_c_stdio():
    import ctypes
    ctypes.CDLL(None).fflush(None)   # RCCL prints its banner through C stdio (block-buffered on a pipe): out with it before the line
    sys.stdout.flush()


def main():
    args = parse_args()
    if args.only_headline:
        args.no_configs = args.no_sharded = args.no_host_path = args.no_cpu_baseline = args.no_parity = args.no_device_state = True
    import torch
    from zoic_amd.workloads import CONFIGS, ray_count

    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    what, plan = resolve_launch(args, os.environ, ndev, sys.argv[1:])
    if what == "exec":
        flush_c_stdio()
        os.execv(plan[0], plan)
    world = plan
    if args.dry_launch:
        return dry_launch(world)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ZOIC_BENCH_SAME_GPU=1: a REHEARSAL of the N > 1 control flow on a 1-GPU box -- every rank uses device 0 and the ranks meet
    # over gloo (RCCL refuses two ranks on one device).  The line it prints is labelled and is not a measurement.
    same_gpu = os.environ.get("ZOIC_BENCH_SAME_GPU") == "1"
    if same_gpu:
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libzoic_amd has no CPU path")
    torch.cuda.set_device(local_rank)      # before the process group: RCCL binds the communicator to the current device
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("ZOIC_FORCE_DIST"):   # ZOIC_FORCE_DIST: initialise the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if world == 1:
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if same_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
        assert dist.get_world_size() == world, (dist.get_world_size(), world)

    cfg = CONFIGS[args.config]
    frame = args.rays or ray_count(args.config)
    n, base, n_total = frame, rank * frame, frame * world        # weak leg: rank r renders frame r (distinct global ray indices)
    cam = make_camera(args.config, args.precision, local_rank)
    keep = {"hold": world == 1 and not args.no_device_state}
    elapsed, kernel_ms = time_frame(torch, cam, cfg, n, base, args.steps, args.warmup, dev, dist, local_rank, keep=keep, candidates=args.placement_candidates)

    line = None
    if rank == 0:
        counters = cam.counters()
        thin = cfg["params"]["lensModel"] == 0
        line = {
            "metric": "camera rays/sec (Mrays/s), 4K x 16spp Kolb lens trace; ray-dir RMSE vs CPU ref",
            "value": round(n_total * args.steps / elapsed / 1e6, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.config, cfg["desc"]), "rays_per_gpu_per_step": n, "precision_mode": args.precision,
                       "parallelism": "one frame on one GPU" if world == 1 else "independent frames per GPU (dp%d), no data-path collective" % world},
            "roofline": roofline_block(args.config if not args.rays else "none", args.precision, n, kernel_ms, thin),
            "zero_weight": frame_stats(counters, counters["succesRays"] + counters["vignettedRays"]),
            "target_mrays_s": round(NORTH_STAR_MRAYS_1GPU * (1.0 if world == 1 else 0.75 * world)),
        }
        # which allocation holds the frame's rays (zoic_amd/placement.py): every candidate's rate over a few frames BEFORE the timed
        # region; `first_pair_mrays_s` is what the plain first allocation would have given
        line["placement"] = keep.get("placement")
        if world == 1 and not args.no_device_state:
            # outside the timed region: what clock did the kernel get?  (the peaks above are at 2.4 GHz; the socket's power limit decides)
            state = device_state_entry(torch, cam, cfg, n, base, dev, buffers=(keep.pop("samples"), keep.pop("out")))
            if state:
                line["device_state"] = state
                if state.get("sclk_mhz"):
                    line["roofline"]["sclk_mhz_under_load"] = state["sclk_mhz"]
                    line["roofline"]["frac_at_measured_clock"] = round(line["roofline"]["frac"] * 2400.0 / state["sclk_mhz"], 4) if line["roofline"]["bound"] == "valu" else None
        if not args.no_parity:
            line["parity"] = parity_probe(cam, args.config, args.precision)
        if not args.no_host_path and world == 1:
            line["host_path"] = host_path_entry(cam, cfg)
    cam.close()
    torch.cuda.empty_cache()
    if rank == 0 and not args.no_host_path and world == 1:
        line["host_path"]["per_sample"] = per_sample_latency()
        line["host_path"]["tile"] = tile_server_entry()

    if not args.no_configs and world == 1:
        ents = []
        # sub-millisecond frames get enough warm-up launches to leave the ~25 ms after an idle gap in which launches run slow (C2: 32.1 Grays/s over
        # launches 4-23, 35.9 from launch 50 on: profiles/ab_r06/warmup_steps.txt)
        for cname, prec, st, wu in (("C1", "fast", 400, 300), ("C2", "fast", 200, 60), ("C4", "fast", 10, 5), ("C5", "fast", 4, 1), ("C3", "strict", 8, 4)):
            if cname == args.config and prec == args.precision:
                continue
            ents.append(config_entry(torch, cname, prec, dev, local_rank, st, wu, not args.no_parity, not args.no_device_state, args.placement_candidates))
        line["configs"] = ents

    if world > 1:
        # ---- north_star's multi-GPU number: the SAME frame, once per step, in N slabs, gather included.  Everything from here
        # on involves transfers between GPUs that no 1-GPU lease can rehearse: a watchdog prints the line with what has been
        # measured (the weak-scaling leg above at the least) if any of it hangs, and a failure is reported, not fatal.
        import threading
        strong, spf, sharded = {}, {}, []
        weak = {"value": line["value"], "ms_per_step": line["ms_per_step"], "scaling": "weak"} if rank == 0 else None

        def finish_line(note=None):
            line["weak"] = weak
            if strong.get("with_gather"):
                line.update(value=strong["with_gather"], ms_per_step=strong["gather_ms"], scaling="strong", compute_only=strong["compute_only"],
                            compute_ms=strong["compute_ms"], root_ingest_gb_s=strong["root_ingest_gb_s"], root_ingest_frac=strong["root_ingest_frac"],
                            bit_identical_to_single_gpu=strong.get("bit_identical_to_single_gpu"), sub_launches_per_slab=strong["sub_launches_per_slab"],
                            chunk_mb=strong["chunk_mb"])
                for k in ("root_bytes", "with_sparse_gather", "sparse_gather_ms", "root_bytes_sparse", "sparse_failed", "zero_weight_fraction", "auto_layout"):
                    if k in strong:
                        line[k] = strong[k]
                line["config"]["parallelism"] = "ONE frame per step in %d ray-index slabs (dp%d), RCCL gather of the 28 B/ray payload to rank 0 included" % (world, world)
                line["config"]["rays_per_gpu_per_step"] = n // world
            elif strong:
                line["strong"] = strong   # the gather did not finish: value stays the weak-scaling figure and says so
            if spf:
                line["single_process_frame"] = spf
            if sharded:
                line["sharded_frame"] = sharded
            line["notes"] = {k: NOTES[k] for k in ("roofline", "multi_gpu", "mode")}
            out = dict(line)
            if note:   # FIRST in the line: whoever reads only its head sees that the multi-GPU legs did not all finish
                out = {"multi_gpu_incomplete": note}
                out.update(line)
            text = json.dumps(out, separators=(",", ":"))
            if len(text) > 6000:
                del out["notes"]
                text = json.dumps(out, separators=(",", ":"))
            flush_c_stdio()
            print(text, flush=True)

        def give_up():
            # the line with what HAS been measured, then a NON-ZERO exit: a hung gather must not look like a finished run
            if rank == 0:
                finish_line("timed out after %d s behind the weak-scaling leg" % args.sharded_timeout)
            os._exit(3)
        # (rank 0 first: it holds the line; the other ranks give it 20 s to print before they leave)
        watchdog = threading.Timer(args.sharded_timeout + (0 if rank == 0 else 20), give_up)
        watchdog.daemon = True
        watchdog.start()
        note = None
        try:
            # north_star's number FIRST (dense RCCL gather); then the single-process frame; then C4 / C5; the RCCL sparse gather (C5: the frame
            # it was built for) is the LAST thing the run does -- nothing above is lost if it misbehaves on its first meeting with real peers
            if not args.rays:
                sharded_frame_entry(torch, dist, args.config, dev, rank, world, local_rank, args.steps, args.gather_chunk_mb, args.precision, args.warmup, strong)
            # rank 0 alone drives all N devices through the C-ABI (no RCCL involved); the other ranks wait on the rendezvous store,
            # not in a collective -- a barrier kernel spinning on their GPUs would share them with rank 0's launches
            if not args.no_single_process and not args.rays:
                import datetime
                store = dist.distributed_c10d._get_default_store()
                if rank == 0:
                    try:
                        if ndev >= world or same_gpu:
                            single_process_frame_entry(torch, args.config, args.precision, [0] * world if same_gpu else list(range(world)), args.steps, args.warmup, spf)
                        else:
                            spf["skipped"] = "rank 0 sees %d device(s)" % ndev
                    except Exception as e:  # noqa: BLE001
                        spf["failed"] = str(e)[:200]
                    store.set("zoic_spf_done", "1")
                else:
                    store.wait(["zoic_spf_done"], datetime.timedelta(seconds=args.sharded_timeout))
                rank_barrier(dist, local_rank)
            if not args.no_sharded:
                for cname, st in (("C4", 5), ("C5", 2)):
                    if cname != args.config:
                        e = sharded_frame_entry(torch, dist, cname, dev, rank, world, local_rank, st, args.gather_chunk_mb, sparse_leg=cname == "C5" and not args.no_sparse_leg)
                        if rank == 0:
                            sharded.append(e)
        except Exception as e:  # noqa: BLE001 -- reported in the line; the weak leg stands
            note = "failed: %s" % (str(e)[:200],)
        watchdog.cancel()
        flush_c_stdio()
        try:
            rank_barrier(dist, local_rank)
        except Exception:  # noqa: BLE001
            pass
        if rank == 0:
            if same_gpu:
                line["rehearsal"] = "ZOIC_BENCH_SAME_GPU=1: all %d ranks on ONE GPU over gloo -- control flow only, not a measurement" % world
            finish_line(note)
        try:
            rank_barrier(dist, local_rank)
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
        return

    # ---- N = 1
    if not args.no_sharded and not args.rays:
        # the C-ABI's multi-device entry points (zoic_frame_*, csrc/frame.cpp) on the hardware this line comes from: ONE GPU listed
        # twice -- slab partition, chunking, two compute streams + a copy stream per lane, the copy "gather" (same-device here) and
        # the root-stream ordering all run as they would across devices; what it cannot show is xGMI
        try:
            line["frame_api"] = dict(single_process_frame_entry(torch, args.config, args.precision, [local_rank, local_rank], 5, 1),
                                     note="zoic_frame_* with the one GPU listed twice: code path + bit-identity on hardware, not a scaling number")
        except Exception as e:  # noqa: BLE001
            line["frame_api"] = {"failed": str(e)[:200]}
        try:   # which gather layout the frame picks by itself (ZOIC_FRAME_PAYLOAD_AUTO), and what each moves: the headline camera and C5
            line["frame_layout"] = [frame_layout_entry(torch, c, args.precision, [local_rank, local_rank]) for c in (args.config, "C5") ]
        except Exception as e:  # noqa: BLE001
            line["frame_layout"] = {"failed": str(e)[:200]}
    if not args.no_sharded:
        # north_star configs 4/5 on one GPU: nothing to gather, a slab is ONE launch (= the unsharded rates)
        line["sharded_frame"] = [sharded_frame_entry(torch, None, cname, dev, 0, 1, local_rank, st, args.gather_chunk_mb) for cname, st in (("C4", 5), ("C5", 2))]
    if dist:
        flush_c_stdio()
        rank_barrier(dist, local_rank)
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args.config, args.cpu_seconds)
    line["notes"] = {k: v for k, v in NOTES.items() if k != "multi_gpu"}
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > 6000:     # the driver keeps the tail of the line: what explains goes first, the numbers stay
        del line["notes"]
        text = json.dumps(line, separators=(",", ":"))
    flush_c_stdio()
    print(text, flush=True)
    if dist:
        rank_barrier(dist, local_rank)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
