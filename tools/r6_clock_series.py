"""Round 6: is the headline's run-to-run spread (40.7 .. 45.4 Grays/s on one box) the GPU's clock / power state?

Launches the C3 decision-safe frame 600 times back to back with a HIP event every 10 launches and samples the device's
sysfs clock / power / temperature files from a second thread; prints the series.  Usage (GPU box): python tools/r6_clock_series.py [C3 [fast|strict|unchecked]]
"""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from zoic_amd.workloads import CONFIGS, ray_count

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "C3"
precision = sys.argv[2] if len(sys.argv) > 2 else "fast"
steps, group = int(os.environ.get("SERIES_STEPS", "600")), 10


def sysfs_files():
    """The hwmon files of the card whose PCI address is the HIP device's (a box shows every card of its host)."""
    out = {}
    pr = torch.cuda.get_device_properties(0)
    addr = "%04x:%02x:%02x." % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
    cards = [c for c in sorted(glob.glob("/sys/class/drm/card*/device")) if addr in os.path.realpath(c)]
    print("HIP device 0 at PCI %s -> %s" % (addr, [os.path.realpath(c) for c in cards]), flush=True)
    for card in cards:
        for name, pat in (("sclk_mhz", "hwmon/hwmon*/freq1_input"), ("mclk_mhz", "hwmon/hwmon*/freq2_input"), ("power_w", "hwmon/hwmon*/power1_average"),
                          ("power_in_w", "hwmon/hwmon*/power1_input"), ("temp_c", "hwmon/hwmon*/temp1_input"), ("temp_hbm_c", "hwmon/hwmon*/temp3_input"), ("busy", "gpu_busy_percent"),
                          ("fclk_mhz", "pp_dpm_fclk"), ("socclk_mhz", "pp_dpm_socclk"), ("dpm_mclk_mhz", "pp_dpm_mclk"), ("dpm_sclk_mhz", "pp_dpm_sclk")):
            for f in glob.glob(os.path.join(card, pat)):
                out.setdefault(name, f)
    return out


def read(f):
    try:
        txt = open(f).read()
        if "pp_dpm" in f:                        # "0: 1250Mhz *" lines: the level in use carries the star
            for ln in txt.splitlines():
                if ln.strip().endswith("*"):
                    return float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip()) * 1e6
            return None
        return float(txt.split()[0])
    except Exception:
        return None


files = sysfs_files()
print("sysfs:", files, flush=True)
scale = {"fclk_mhz": 1e-6, "socclk_mhz": 1e-6, "dpm_mclk_mhz": 1e-6, "dpm_sclk_mhz": 1e-6, "sclk_mhz": 1e-6, "mclk_mhz": 1e-6, "power_w": 1e-6, "power_in_w": 1e-6, "temp_c": 1e-3, "temp_hbm_c": 1e-3, "busy": 1.0}
samples, stop = [], False


def sampler():
    while not stop:
        samples.append((time.perf_counter(), {k: (read(f) or 0.0) * scale[k] for k, f in files.items()}))
        time.sleep(0.02)


dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cfg = CONFIGS[cfg_name]
n = ray_count(cfg_name)
cam = bench.make_camera(cfg_name, precision, 0)
s = cam.generate_samples(n, cfg["width"], cfg["height"], cfg["spp"], seed=1, ray_index_base=0)
out = dict(rays=torch.empty((n, 8), dtype=torch.float32, device=dev))
torch.cuda.synchronize()
idle = float(os.environ.get("SERIES_IDLE", "2.0"))
th = threading.Thread(target=sampler, daemon=True)
th.start()
time.sleep(idle)                                 # the device idle: what state does a first launch meet?
evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps // group + 1)]
t_start = time.perf_counter()
evs[0].record()
for k in range(steps):
    cam.create_rays(s, ray_index_base=0, out=out)
    if (k + 1) % group == 0:
        evs[(k + 1) // group].record()
torch.cuda.synchronize()
t_end = time.perf_counter()
time.sleep(0.2)
stop = True
th.join()
ms = [evs[i].elapsed_time(evs[i + 1]) / group for i in range(len(evs) - 1)]
print("%s %s: %d launches, wall %.1f ms a frame overall; per-10-launch means (ms):" % (cfg_name, precision, steps, (t_end - t_start) / steps * 1e3))
print(" ".join("%.3f" % m for m in ms))
print("Grays/s: first 10 launches %.1f, launches 10-30 %.1f, last 100 %.1f, best group %.1f, worst %.1f" % (
    n / ms[0] / 1e6, n / (sum(ms[1:3]) / 2) / 1e6, n / (sum(ms[-10:]) / 10) / 1e6, n / min(ms) / 1e6, n / max(ms) / 1e6))
print("t_ms_since_first_launch " + " ".join(files))
last = None
for t, v in samples:
    row = tuple(round(v[k], 0 if k != "busy" else 0) for k in files)
    if row != last or True:
        rel = (t - t_start) * 1e3
        if rel < -100 or int(rel / 20) % 5:
            continue
        print("%8.0f " % rel + " ".join("%7.0f" % x for x in row))
        last = row
cam.close()
