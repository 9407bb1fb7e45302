"""zoic_frame_* (include/zoic_amd.h): one camera node over several HIP devices of ONE process -- the form a C++ plug-in
can call (the reference is one process, zoic.cpp:1752).  The bar is SURVEY 8(e)'s: the sharded frame is bit-identical to the
one-device frame (retry streams are keyed by the global ray index).  A 1-GPU box lists its device more than once
(devices = [0, 0], [0, 0, 0]): slab partition, chunking, the two compute streams, the copy stream, the peer-copy gather, buffer
reuse across back-to-back calls and the root-stream ordering all run exactly as with distinct devices."""
import numpy as np
import pytest

from zoic_amd import FRAME_PAYLOAD, FRAME_RECORDS, PRECISION_FAST, PRECISION_STRICT, ZoicCamera, ZoicFrame, frame_slab
from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh, ray_rng_states, synthetic_samples

pytestmark = pytest.mark.gpu


def setup(obj, cfg, precision):
    if CONFIGS[cfg]["bokeh"]:
        obj.set_bokeh_image(hexagon_bokeh())
    obj.update(**camera_params(cfg))
    obj.set_precision(precision)
    return obj


def single(cfg, n, base, precision):
    import torch
    c = CONFIGS[cfg]
    cam = setup(ZoicCamera(0), cfg, precision)
    s = cam.generate_samples(n, c["width"], c["height"], c["spp"], seed=1, ray_index_base=base)
    rays = cam.create_rays(s, ray_index_base=base)["rays"].clone()
    torch.cuda.synchronize()
    counters = cam.counters()
    cam.close()
    return rays, counters


@pytest.mark.parametrize("cfg,precision", [("C2", PRECISION_STRICT), ("C3", PRECISION_FAST), ("C4", PRECISION_FAST), ("C1", PRECISION_STRICT)])
@pytest.mark.parametrize("devices,chunk", [([0], 0), ([0, 0], 0), ([0, 0, 0], 40_000), ([0, 0], 256)])
def test_frame_equals_the_one_device_call(gpu, cfg, precision, devices, chunk):
    import torch
    c = CONFIGS[cfg]
    n, base = 300_000 + 77, 5_000_000          # ragged: the last tile is partial
    if chunk == 256:
        n = 20_000 + 3                         # hundreds of chunks per slab
    ref, ref_counters = single(cfg, n, base, precision)
    with ZoicFrame(devices) as frame:
        setup(frame, cfg, precision)
        if chunk:
            frame.set_chunk_rays(chunk)
        frame.generate_samples(n, c["width"], c["height"], c["spp"], seed=1, ray_index_base=base)
        rec = frame.render(n, ray_index_base=base, layout=FRAME_RECORDS)
        pay = frame.render(n, ray_index_base=base, layout=FRAME_PAYLOAD)      # back to back: staging buffers are reused
        rec2 = frame.render(n, ray_index_base=base, layout=FRAME_RECORDS, out=torch.zeros_like(rec))
        torch.cuda.synchronize()               # the calls are ordered on torch's current stream of the root device
        assert torch.equal(rec.view(torch.int32), ref.view(torch.int32))
        assert torch.equal(rec2.view(torch.int32), ref.view(torch.int32))
        assert torch.equal(pay.view(torch.int32), ref[:, :7].contiguous().view(torch.int32))
        got = frame.counters()
        # three renders; node_update's TIR bumps are counted once however many devices ran it
        pre = ZoicCamera(0)
        setup(pre, cfg, precision)
        tir0 = pre.info()["precomputeTIR"]
        pre.close()
        assert got["succesRays"] == 3 * ref_counters["succesRays"] and got["vignettedRays"] == 3 * ref_counters["vignettedRays"]
        assert got["totalInternalReflection"] - tir0 == 3 * (ref_counters["totalInternalReflection"] - tir0)


def test_frame_with_caller_samples_and_local_render(gpu):
    import torch
    cfg, n, base = "C2", 200_000, 1_000_000
    c = CONFIGS[cfg]
    ref, _ = single(cfg, n, base, PRECISION_FAST)
    with ZoicFrame([0, 0]) as frame:
        setup(frame, cfg, PRECISION_FAST)
        dev = torch.device("cuda", 0)
        slabs = [frame.slab(n, i) for i in range(2)]
        assert slabs == [frame_slab(n, 2, 0), frame_slab(n, 2, 1)] and slabs[0][0] == 0 and slabs[1][1] == n and slabs[0][1] == slabs[1][0]
        samples = [torch.from_numpy(synthetic_samples(b - a, c["width"], c["height"], c["spp"], seed=1, ray_index_base=base + a)).to(dev) for a, b in slabs]
        torch.cuda.synchronize()
        out = frame.render(n, samples=samples, ray_index_base=base)
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32))
        frame.render_local(n, samples=samples, ray_index_base=base)       # compute-only leg: nothing moves, nothing breaks
        frame.synchronize()
        with pytest.raises(Exception):
            frame.render(n, ray_index_base=base)                          # no generated samples for this (n, base)


def test_frame_host_buffers_equal_the_oracle(gpu, oracle_lib):
    cfg, n, base = "C5", 150_000, 3_000_000
    c = CONFIGS[cfg]
    s = synthetic_samples(n, c["width"], c["height"], c["spp"], seed=1, ray_index_base=base)
    oc = oracle_lib.OracleCamera()
    oc.update(**camera_params(cfg))
    ref = oc.create_rays(s, rng_states=ray_rng_states(n, seed=1, ray_index_base=base), threads=8)
    with ZoicFrame([0, 0, 0]) as frame:
        setup(frame, cfg, PRECISION_STRICT)
        rays = frame.render_host(s, ray_index_base=base)
    planes = np.stack([rays[k] for k in ("ox", "oy", "oz", "dx", "dy", "dz", "weight")])
    assert np.array_equal(rays["flags"].astype(np.uint8), ref["flags"])
    assert np.array_equal(planes.view(np.uint32), ref["planes"].view(np.uint32))


def test_frame_errors(gpu):
    from zoic_amd import ZoicError
    with pytest.raises(ZoicError):
        ZoicFrame([])
    with pytest.raises(ZoicError):
        ZoicFrame([0, 99])
    with ZoicFrame([0, 0]) as frame:
        with pytest.raises(ZoicError) as e:
            frame.update(lensDataPath="/no/such/lens.dat")
        assert e.value.status_name == "ZOIC_ERR_LENS_PATH"


def test_frame_entry_points_from_plain_c(gpu):
    """tests/native/frame_from_c.c (gcc, no HIP headers): the library's multi-device entry points as a C plug-in links them."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "native", "frame_from_c")
    if not os.path.exists(exe):
        subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(root, "include"), exe + ".c", "-o", exe, "-L" + os.path.join(root, "zoic_amd"),
                               "-lzoic_amd", "-Wl,-rpath,$ORIGIN/../../zoic_amd"])
    r = subprocess.run([exe, os.path.join(root, "zoic_amd", "lenses", "tessar_f2.8.dat")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert '"identical": 1' in r.stdout and '"counters_ok": 1' in r.stdout


@pytest.mark.parametrize("cfg,devices,chunk", [("C5", [0, 0, 0], 0), ("C5", [0, 0, 0], 30_000), ("C2", [0, 0], 50_000), ("C4", [0, 0, 0, 0], 0), ("C5", [0], 0)])
def test_sparse_payload_moves_only_the_live_rays(gpu, cfg, devices, chunk):
    """ZOIC_FRAME_PAYLOAD_SPARSE: per 256-ray tile a live mask + the compacted rows of the rays with weight != 0 travel; expanded on the
    root.  Live rows bit-identical to the dense payload, rows of weight-0 rays all zero, and the bytes that moved into the root shrink
    by the frame's zero-weight share (C5's corner rows: 4.5x), lane_info says by how much.  Back-to-back with the other layouts."""
    import torch
    from zoic_amd import FRAME_PAYLOAD_SPARSE
    c = CONFIGS[cfg]
    n = 600_000 + 131
    base = {"C5": 0, "C2": 2_000_000, "C4": 5_000_000}[cfg]       # C5: the frame's first rows, mostly dead pixels
    ref, _ = single(cfg, n, base, PRECISION_FAST)
    ref7 = ref[:, :7].contiguous()
    live = ref7[:, 6] != 0
    with ZoicFrame(devices) as frame:
        setup(frame, cfg, PRECISION_FAST)
        if chunk:
            frame.set_chunk_rays(chunk)
        frame.generate_samples(n, c["width"], c["height"], c["spp"], seed=1, ray_index_base=base)
        dense = frame.render(n, ray_index_base=base, layout=FRAME_PAYLOAD)
        torch.cuda.synchronize()
        dense_bytes = sum(frame.lane_info(i)["bytes_to_root"] for i in range(len(devices)))
        sparse = frame.render(n, ray_index_base=base, layout=FRAME_PAYLOAD_SPARSE)
        again = frame.render(n, ray_index_base=base, layout=FRAME_PAYLOAD_SPARSE, out=torch.full_like(sparse, 9.0))   # staging reuse
        sparse_bytes = sum(frame.lane_info(i)["bytes_to_root"] for i in range(len(devices)))   # (of the last render call)
        info = [frame.lane_info(i) for i in range(len(devices))]
        rec = frame.render(n, ray_index_base=base, layout=FRAME_RECORDS)                                               # and another layout behind it
        torch.cuda.synchronize()
    assert torch.equal(dense.view(torch.int32), ref7.view(torch.int32))
    assert torch.equal(sparse[live].view(torch.int32), ref7[live].view(torch.int32))
    assert bool((sparse[~live] == 0).all()) and int((~live).sum()) > 0 or cfg == "C4"
    assert torch.equal(again.view(torch.int32), sparse.view(torch.int32))
    assert torch.equal(rec.view(torch.int32), ref.view(torch.int32))
    assert info[0]["bytes_to_root"] == 0 and all(i["peer_access_to_root"] and i["peer_access_from_root"] for i in info)
    if len(devices) > 1:
        lo, hi = frame_slab(n, len(devices), 0)
        peers_live = int(live[hi:].sum())
        tiles = sum((frame_slab(n, len(devices), i)[1] - frame_slab(n, len(devices), i)[0] + 255) // 256 for i in range(1, len(devices)))
        assert dense_bytes == 28 * (n - (hi - lo))
        assert 28 * peers_live <= sparse_bytes <= 28 * peers_live + 48 * (tiles + 64 * len(devices))
        if cfg == "C5":
            assert dense_bytes / sparse_bytes > 3.0


@pytest.mark.parametrize("cfg,expect_sparse", [("C5", True), ("C3", False), ("C2", False)])
def test_auto_payload_layout_follows_the_cameras_dead_ray_fraction(gpu, cfg, expect_sparse):
    """ZOIC_FRAME_PAYLOAD_AUTO.  The first AUTO render after an update gathers dense and the layout is undecided; the second
    reads the frame's counters and decides -- SPARSE for C5's first rows (dead pixels), dense for the double Gauss and for this slab of
    the TESSAR (15 % weight 0 < 25 %); the third follows the decision (bytes into the root say which); live rows are the one-device call's
    bits throughout; an update makes it decide again."""
    import torch
    from zoic_amd import FRAME_PAYLOAD_AUTO, FRAME_PAYLOAD_SPARSE
    c = CONFIGS[cfg]
    n = 400_000 + 77
    base = {"C5": 0, "C2": 2_000_000, "C3": 50_000_000}[cfg]
    ref, _ = single(cfg, n, base, PRECISION_FAST)
    ref7 = ref[:, :7].contiguous()
    live = ref7[:, 6] != 0
    devices = [0, 0, 0]
    with ZoicFrame(devices) as frame:
        setup(frame, cfg, PRECISION_FAST)
        frame.generate_samples(n, c["width"], c["height"], c["spp"], seed=1, ray_index_base=base)
        assert frame.auto_layout() == (None, None)
        first = frame.render(n, ray_index_base=base, layout=FRAME_PAYLOAD_AUTO)
        torch.cuda.synchronize()
        dense_bytes = sum(frame.lane_info(i)["bytes_to_root"] for i in range(len(devices)))
        assert frame.auto_layout()[0] is None                      # one render: nothing measured yet
        second = frame.render(n, ray_index_base=base, layout=FRAME_PAYLOAD_AUTO)
        torch.cuda.synchronize()
        layout, zero = frame.auto_layout()
        third = frame.render(n, ray_index_base=base, layout=FRAME_PAYLOAD_AUTO)
        torch.cuda.synchronize()
        third_bytes = sum(frame.lane_info(i)["bytes_to_root"] for i in range(len(devices)))
        assert torch.equal(first.view(torch.int32), ref7.view(torch.int32))      # the undecided render is the dense payload
        assert abs(zero - float((~live).float().mean())) < 1e-6
        assert (layout == FRAME_PAYLOAD_SPARSE) == expect_sparse and (layout == FRAME_PAYLOAD) == (not expect_sparse)
        for got in (second, third):
            assert torch.equal(got[live].view(torch.int32), ref7[live].view(torch.int32))
            if expect_sparse:
                assert bool((got[~live] == 0).all())
            else:
                assert torch.equal(got.view(torch.int32), ref7.view(torch.int32))
        assert (third_bytes < 0.5 * dense_bytes) if expect_sparse else (third_bytes == dense_bytes)
        setup(frame, cfg, PRECISION_FAST)                             # node_update: another camera as far as AUTO is concerned
        assert frame.auto_layout() == (None, None)
