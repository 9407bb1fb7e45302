"""The N>1 path on CPU: world_size-2 gloo processes shard a frame by global ray index, generate their slabs and
gather them on rank 0; the result must be bit-identical to the unsharded frame.  The ray generator plugged in here is
the CPU oracle (the checker) -- on GPUs bench.py plugs in ZoicCamera.create_rays; the sharding/gather code is shared."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slabs_partition_the_frame():
    from zoic_amd.sharding import all_slabs, slab_for_rank
    for n in (0, 1, 255, 256, 257, 1000, 132710400, 2123366400):
        for world in (1, 2, 3, 8):
            slabs = all_slabs(n, world)
            assert slabs[0][0] == 0 and slabs[-1][1] == n
            for (a, b), (c, d) in zip(slabs, slabs[1:]):
                assert b == c and a <= b
            assert all(a % 256 == 0 for a, _ in slabs)
            sizes = [b - a for a, b in slabs]
            assert max(sizes) - min(sizes) <= 256 or n < 256 * world
    with pytest.raises(ValueError):
        slab_for_rank(10, 2, 2)


def test_the_library_partitions_like_the_python_mirror():
    """zoic_frame_slab (the C-ABI's partition of a frame over the devices of one process, csrc/frame.cpp) is the same
    function as sharding.slab_for_rank (one process per GPU): a frame is cut the same way whichever host drives it.
    Pure host arithmetic -- no device needed."""
    import random
    from zoic_amd import ZoicError, frame_slab
    from zoic_amd.sharding import slab_for_rank
    rnd = random.Random(4)
    for _ in range(5000):
        n = rnd.choice([0, 1, 255, 256, 257, 132710400, 2123366400, rnd.randrange(1, 1 << 40)])
        world = rnd.randrange(1, 65)
        r = rnd.randrange(world)
        assert frame_slab(n, world, r) == slab_for_rank(n, r, world)
    for bad in ((10, 0, 0), (10, 2, 2), (10, 2, -1)):
        with pytest.raises(ZoicError):
            frame_slab(*bad)


def _worker(rank, world, port, n, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import oracle
    from zoic_amd.sharding import gather_rays, slab_for_rank
    from zoic_amd.workloads import CONFIGS, camera_params, ray_rng_states, synthetic_samples
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = CONFIGS["C2"]
    lo, hi = slab_for_rank(n, rank, world)
    oc = oracle.OracleCamera().update(**camera_params("C2"))
    s = synthetic_samples(hi - lo, c["width"], c["height"], c["spp"], seed=1, ray_index_base=lo)
    r = oc.create_rays(s, rng_states=ray_rng_states(hi - lo, 1, lo))
    rec = np.zeros((hi - lo, 8), np.float32)          # the (n,8) zoic_ray record layout the GPU path produces
    rec[:, :7] = r["planes"].T
    rec[:, 7] = r["flags"].astype(np.uint32).view(np.float32)
    full = gather_rays(torch.from_numpy(rec), n, dist, dst=0)
    if rank == 0:
        np.save(os.path.join(outdir, "rays.npy"), full.numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather_equals_single_process(tmp_path, oracle_lib):
    import torch.multiprocessing as mp
    from zoic_amd.workloads import CONFIGS, camera_params, ray_rng_states, synthetic_samples
    n = 3000  # not a multiple of the tile: ragged slabs (1536 + 1464)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, n, str(tmp_path)), nprocs=2, join=True)
    c = CONFIGS["C2"]
    oc = oracle_lib.OracleCamera().update(**camera_params("C2"))
    ref = oc.create_rays(synthetic_samples(n, c["width"], c["height"], c["spp"], seed=1), rng_states=ray_rng_states(n, 1, 0))
    got = np.load(tmp_path / "rays.npy")
    assert got.shape == (n, 8)
    assert np.array_equal(np.ascontiguousarray(got[:, :7].T).view(np.uint32), ref["planes"].view(np.uint32))
    assert np.array_equal(np.ascontiguousarray(got[:, 7]).view(np.uint32), ref["flags"].astype(np.uint32))


def _oracle_generate(n_total, cfg="C2"):
    """ray-record generator over global ray indices backed by the oracle (CPU tests); bench.py plugs ZoicCamera in here"""
    import torch
    import oracle
    from zoic_amd.workloads import CONFIGS, camera_params, ray_rng_states, synthetic_samples
    c = CONFIGS[cfg]
    oc = oracle.OracleCamera().update(**camera_params(cfg))

    def generate(a, b):
        s = synthetic_samples(b - a, c["width"], c["height"], c["spp"], seed=1, ray_index_base=a)
        r = oc.create_rays(s, rng_states=ray_rng_states(b - a, 1, a))
        rec = np.zeros((b - a, 8), np.float32)
        rec[:, :7] = r["planes"].T
        rec[:, 7] = r["flags"].astype(np.uint32).view(np.float32)
        return torch.from_numpy(rec)
    return generate


def _pipelined_worker(rank, world, port, n, chunk_bytes, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from zoic_amd.sharding import ShardedFrame
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frame = ShardedFrame(n, dist, torch.device("cpu"), _oracle_generate(n), dst=0, chunk_bytes=chunk_bytes)
    assert frame.run(gather=False) is None                       # compute-only leg of the bench: no exchange at all
    for step in range(2):                                        # reusable across steps
        full = frame.run(gather=True)
    if rank == 0:
        np.save(os.path.join(outdir, "payload.npy"), full.numpy())
        np.save(os.path.join(outdir, "rounds.npy"), np.array([frame.rounds, len(frame.chunks[0]), len(frame.chunks[-1])]))
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,chunk_bytes", [(2, 3000, 28 * 512), (3, 10_001, 28 * 1024), (2, 700, 64 << 20)])
def test_pipelined_chunked_gather_equals_single_process(tmp_path, oracle_lib, world, n, chunk_bytes):
    """ShardedFrame: slabs cut into sub-launch chunks, the 28-byte payload of chunk k posted (batch_isend_irecv) while
    chunk k+1 is generated; ragged slabs, ragged last chunks, more rounds on some ranks than on others."""
    import torch.multiprocessing as mp
    from zoic_amd.workloads import CONFIGS, camera_params, ray_rng_states, synthetic_samples
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_pipelined_worker, args=(world, port, n, chunk_bytes, str(tmp_path)), nprocs=world, join=True)
    c = CONFIGS["C2"]
    oc = oracle_lib.OracleCamera().update(**camera_params("C2"))
    ref = oc.create_rays(synthetic_samples(n, c["width"], c["height"], c["spp"], seed=1), rng_states=ray_rng_states(n, 1, 0))
    got = np.load(tmp_path / "payload.npy")
    assert got.shape == (n, 7)                                   # origin, dir, weight: the flag word is not gathered
    assert np.array_equal(np.ascontiguousarray(got.T).view(np.uint32), ref["planes"].view(np.uint32))
    rounds = np.load(tmp_path / "rounds.npy")
    if chunk_bytes < (1 << 20):
        assert rounds[0] > 1                                     # really chunked


def _gpu_worker(rank, world, port, n, outdir):
    """both ranks drive cuda:0 through ZoicCamera (one GPU on the test box); the exchange runs over gloo on host tensors"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from zoic_amd import PRECISION_STRICT, ZoicCamera
    from zoic_amd.sharding import ShardedFrame
    from zoic_amd.workloads import CONFIGS, camera_params
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = CONFIGS["C2"]
    cam = ZoicCamera(0)
    cam.update(**camera_params("C2"))
    cam.set_precision(PRECISION_STRICT)

    def generate(a, b):
        s = cam.generate_samples(b - a, c["width"], c["height"], c["spp"], seed=1, ray_index_base=a)
        return cam.create_rays(s, ray_index_base=a)["rays"].cpu()
    full = ShardedFrame(n, dist, torch.device("cpu"), generate, dst=0, chunk_bytes=28 * 40_000).run(gather=True)
    if rank == 0:
        np.save(os.path.join(outdir, "payload_gpu.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()
    cam.close()


@pytest.mark.gpu
def test_two_rank_sharded_frame_through_the_hip_path(gpu, tmp_path, oracle_lib):
    """The same ShardedFrame with ZoicCamera (the HIP path) as the generator: two processes, ray-index slabs, chunked
    gather; bit-identical to the oracle's unsharded frame."""
    import torch.multiprocessing as mp
    from zoic_amd.workloads import CONFIGS, camera_params, ray_rng_states, synthetic_samples
    n = 250_000
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_gpu_worker, args=(2, port, n, str(tmp_path)), nprocs=2, join=True)
    c = CONFIGS["C2"]
    oc = oracle_lib.OracleCamera().update(**camera_params("C2"))
    ref = oc.create_rays(synthetic_samples(n, c["width"], c["height"], c["spp"], seed=1), rng_states=ray_rng_states(n, 1, 0), threads=8)
    got = np.load(tmp_path / "payload_gpu.npy")
    assert np.array_equal(np.ascontiguousarray(got.T).view(np.uint32), ref["planes"].view(np.uint32))


# ---------------------------------------------------------------------------------- bench.py's launcher
def _bench():
    sys.path.insert(0, ROOT)
    import importlib
    return importlib.import_module("bench")


def test_bench_gpus_flag_means_ranks():
    """`--gpus N` is N ranks, whoever starts them: without a launcher bench.py re-executes itself under torch.distributed.run;
    under one it must agree with WORLD_SIZE; more ranks than GPUs is refused loudly (round 3: `--gpus 8` silently ran one rank)."""
    import argparse
    bench = _bench()
    ns = lambda **kw: argparse.Namespace(**dict(dict(gpus=None, dry_launch=False), **kw))  # noqa: E731
    assert bench.resolve_launch(ns(), {}, 1, []) == ("run", 1)
    assert bench.resolve_launch(ns(gpus=1), {}, 8, []) == ("run", 1)
    assert bench.resolve_launch(ns(gpus=4), {"WORLD_SIZE": "4"}, 8, []) == ("run", 4)
    assert bench.resolve_launch(ns(), {"WORLD_SIZE": "8"}, 8, []) == ("run", 8)          # a launcher without --gpus: its world
    what, cmd = bench.resolve_launch(ns(gpus=8), {}, 8, ["--gpus", "8", "--steps", "5"])
    assert what == "exec" and cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "5"] and cmd[-5].endswith("bench.py")
    for bad in (dict(a=ns(gpus=2), env={}, ndev=1),                       # a 1-GPU box asked for 2 ranks
                dict(a=ns(gpus=2), env={"WORLD_SIZE": "4"}, ndev=8),     # --gpus contradicts the launcher
                dict(a=ns(gpus=8), env={"WORLD_SIZE": "8"}, ndev=4),     # more ranks than devices
                dict(a=ns(gpus=0), env={}, ndev=1)):
        with pytest.raises(SystemExit) as e:
            bench.resolve_launch(bad["a"], bad["env"], bad["ndev"], [])
        assert e.value.code not in (0, None) and "bench.py" in str(e.value.code)


def test_bench_launches_its_own_ranks_over_gloo():
    """The real re-exec path at world 2 on CPU: `python bench.py --gpus 2 --dry-launch` (no launcher, no WORLD_SIZE) must come
    back as two ranks that met each other; and without a GPU the measuring form exits non-zero with a clear message."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    got = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert got == {"dry_launch": True, "world": 2, "expected": 2, "sum": 2}
    from zoic_amd import _capi
    if _capi.load().zoic_device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode != 0 and "--gpus 2 but only" in (r.stdout + r.stderr)


def _sparse_worker(rank, world, port, n, chunk_bytes, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from zoic_amd.sharding import ShardedFrame
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frame = ShardedFrame(n, dist, torch.device("cpu"), _oracle_generate(n), dst=0, chunk_bytes=chunk_bytes, sparse=True)
    for step in range(2):
        full = frame.run(gather=True)
    if rank == 0:
        np.save(os.path.join(outdir, "sparse.npy"), full.numpy())
        np.save(os.path.join(outdir, "bytes.npy"), np.array([frame.root_bytes]))
    dist.barrier()
    dist.destroy_process_group()


def test_sparse_gather_ships_only_the_live_rays(tmp_path, oracle_lib):
    """ShardedFrame(sparse=True) at world 3 over gloo: live rows bit-identical to the single-process frame, rows of weight-0 rays zero,
    and the root receives 28 bytes per LIVE peer ray + a bit per ray + a count per chunk (the dense gather: 28 bytes per ray)."""
    import torch.multiprocessing as mp
    from zoic_amd.sharding import all_slabs
    from zoic_amd.workloads import CONFIGS, camera_params, ray_rng_states, synthetic_samples
    world, n, chunk_bytes = 3, 10_001, 28 * 1024
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_sparse_worker, args=(world, port, n, chunk_bytes, str(tmp_path)), nprocs=world, join=True)
    c = CONFIGS["C2"]
    oc = oracle_lib.OracleCamera().update(**camera_params("C2"))
    ref = oc.create_rays(synthetic_samples(n, c["width"], c["height"], c["spp"], seed=1), rng_states=ray_rng_states(n, 1, 0))
    got = np.load(tmp_path / "sparse.npy")
    planes = np.ascontiguousarray(ref["planes"].T)
    live = planes[:, 6] != 0
    assert live.sum() > 100 and (~live).sum() > 100
    assert np.array_equal(got[live].view(np.uint32), planes[live].view(np.uint32))
    assert (got[~live] == 0).all()
    lo, hi = all_slabs(n, world)[0]
    peers_live = int(live[hi:].sum())
    moved = int(np.load(tmp_path / "bytes.npy")[0])
    assert 28 * peers_live <= moved <= 28 * peers_live + (n - hi) // 8 + 200
    assert moved < 0.95 * 28 * (n - hi)


def _auto_worker(rank, world, port, n, chunk_bytes, cfg, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from zoic_amd.sharding import ShardedFrame
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frame = ShardedFrame(n, dist, torch.device("cpu"), _oracle_generate(n, cfg), dst=0, chunk_bytes=chunk_bytes, sparse="auto")
    moved = []
    for step in range(3):
        full = frame.run(gather=True)
        moved.append(frame.root_bytes)
    if rank == 0:
        np.save(os.path.join(outdir, "auto_%s.npy" % cfg), full.numpy())
    np.save(os.path.join(outdir, "auto_%s_rank%d.npy" % (cfg, rank)), np.array([frame.zero_weight_fraction, float(frame.sparse)] + moved))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("cfg,expect_sparse", [("C4", False), ("C5", True)])
def test_auto_layout_is_chosen_from_the_cameras_dead_ray_fraction(tmp_path, oracle_lib, cfg, expect_sparse):
    """ShardedFrame(sparse="auto") gathers dense once, the root measures the frame's zero-weight fraction and every rank
    switches to the sparse layout only at >= 25 % (the first rows of C5: all dead pixels -> sparse; the fisheye has no dead ray -> dense).
    Every rank holds the same decision; the live rows are the single-process frame's either way."""
    import torch.multiprocessing as mp
    from zoic_amd.workloads import CONFIGS, camera_params, ray_rng_states, synthetic_samples
    world, n, chunk_bytes = 2, 6_001, 28 * 1024
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_auto_worker, args=(world, port, n, chunk_bytes, cfg, str(tmp_path)), nprocs=world, join=True)
    c = CONFIGS[cfg]
    oc = oracle_lib.OracleCamera().update(**camera_params(cfg))
    ref = oc.create_rays(synthetic_samples(n, c["width"], c["height"], c["spp"], seed=1), rng_states=ray_rng_states(n, 1, 0))
    planes = np.ascontiguousarray(ref["planes"].T)
    live = planes[:, 6] != 0
    got = np.load(tmp_path / ("auto_%s.npy" % cfg))
    assert np.array_equal(got[live].view(np.uint32), planes[live].view(np.uint32))
    r0, r1 = np.load(tmp_path / ("auto_%s_rank0.npy" % cfg)), np.load(tmp_path / ("auto_%s_rank1.npy" % cfg))
    assert r0[0] == r1[0] and abs(r0[0] - float((~live).mean())) < 1e-12       # the root's measurement reached every rank
    assert bool(r0[1]) == bool(r1[1]) == expect_sparse, (cfg, r0)
    if expect_sparse:
        assert (got[~live] == 0).all() and r0[4] < 0.5 * r0[2]      # the third run moved less than half of what the dense first run did
    else:
        assert np.array_equal(got.view(np.uint32), planes.view(np.uint32)) and r0[4] == r0[2]
