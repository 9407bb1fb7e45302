"""tools/soak_mailbox.py [seconds]: the per-sample mailbox under churn -- 24 render threads calling zoic_camera_create_ray with
random pauses (so the resident kernel retires and restarts at random), one thread reading the counters (which stops the
kernel every time), one launching batches on the same camera.  Nothing may hang; every tid's rays must equal a quiet replay."""
import os, sys, threading, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from zoic_amd import ZoicCamera, PRECISION_STRICT
from zoic_amd.workloads import CONFIGS, camera_params, synthetic_samples
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
c = CONFIGS["C2"]
s = synthetic_samples(4096, c["width"], c["height"], c["spp"], seed=3, ray_index_base=c["width"] * 300 * c["spp"])
cam = ZoicCamera(0); cam.update(**camera_params("C2")); cam.set_precision(PRECISION_STRICT)
stop = time.time() + seconds
log = {}
def render(tid):
    rnd = random.Random(tid)
    out = []
    i = 0
    while time.time() < stop:
        row = s[(tid * 131 + i) % len(s)]
        o = cam.create_ray(*[float(v) for v in row], tid=tid)
        out.append((o.dir.x, o.dir.y, o.dir.z, o.weight[0]))
        i += 1
        r = rnd.random()
        if r < 0.002: time.sleep(rnd.random() * 0.004)       # idle long enough for the kernel to retire now and then
    log[tid] = out
def counters():
    while time.time() < stop:
        cam.counters(); time.sleep(0.013)
def batches():
    while time.time() < stop:
        cam.create_rays(s); time.sleep(0.003)
tids = list(range(20)) + [64, 65, 129, 200]                    # slots shared by tids 64 apart
th = [threading.Thread(target=render, args=(t,)) for t in tids] + [threading.Thread(target=counters), threading.Thread(target=batches)]
for t in th: t.start()
for t in th: t.join(timeout=seconds + 60)
assert not any(t.is_alive() for t in th), "a thread hangs"
total = sum(len(v) for v in log.values())
print("calls", total, "per thread", min(len(v) for v in log.values()), "...", max(len(v) for v in log.values()))
quiet = ZoicCamera(0); quiet.update(**camera_params("C2")); quiet.set_precision(PRECISION_STRICT)
bad = 0
for tid in tids:
    if tid == 0: continue                                     # tid 0 shares the camera's reference stream with nothing else here, but keep it simple
    for i, want in enumerate(log[tid][:3000]):
        row = s[(tid * 131 + i) % len(s)]
        o = quiet.create_ray(*[float(v) for v in row], tid=tid)
        got = (o.dir.x, o.dir.y, o.dir.z, o.weight[0])
        if not (np.array(got, np.float32).view(np.uint32) == np.array(want, np.float32).view(np.uint32)).all():
            bad += 1
print("mismatches", bad)
assert bad == 0
print("mailbox soak passed")
