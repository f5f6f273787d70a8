"""CPU, world_size 2, gloo: the N > 1 path of bench.py — disjoint sample shards per rank + ONE all-reduce of the float
accumulator + 1/N — gives the image of one big render (<= 1e-6 relative, summation order only).
The oracle stands in for the GPU renderer here (tests may use it); the sharding/reduce code is the product's."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP_PER_RANK, B = 48, 32, 3, 5


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    from platinum_amd import scenes
    from platinum_amd.renderer import make_params
    from platinum_amd.sharding import reduce_accumulator, shard_samples
    first, n = shard_samples(rank, world, SPP_PER_RANK)
    o = oracle_lib.OracleScene(scenes.cornell_sphere_scene(), make_params(W, H, n, B, first_sample=first))
    acc = torch.from_numpy(o.render(first, n, threads=2))
    dist.barrier()
    reduce_accumulator(acc, world, dist)
    if rank == 0:
        np.save(os.path.join(out_dir, "merged.npy"), acc.numpy())
    dist.destroy_process_group()


def test_two_rank_sample_sharding_equals_single_render(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from platinum_amd import scenes
    from platinum_amd.renderer import make_params
    from platinum_amd.sharding import shard_samples
    oracle_lib.lib()  # build before forking
    assert shard_samples(0, 2, 128) == (0, 128) and shard_samples(7, 8, 128) == (896, 128)
    with pytest.raises(ValueError):
        shard_samples(2, 2, 1)
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    merged = np.load(tmp_path / "merged.npy")
    full = oracle_lib.OracleScene(scenes.cornell_sphere_scene(), make_params(W, H, 2 * SPP_PER_RANK, B)).render(0, 2 * SPP_PER_RANK)
    assert np.array_equal(merged[..., 3], np.ones((H, W), np.float32))
    np.testing.assert_allclose(merged[..., :3], full[..., :3], rtol=2e-6, atol=1e-6)
