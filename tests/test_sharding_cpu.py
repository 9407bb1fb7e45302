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


def _oracle_generate(n_total):
    """ray-record generator over global ray indices backed by the oracle (CPU tests); bench.py plugs ZoicCamera in here"""
    import torch
    import oracle
    from zoic_amd.workloads import CONFIGS, camera_params, ray_rng_states, synthetic_samples
    c = CONFIGS["C2"]
    oc = oracle.OracleCamera().update(**camera_params("C2"))

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
