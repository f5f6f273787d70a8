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


# ---- the launcher of `python bench.py --gpus N` itself: it must not be able to die silently --------------------------
def _bench(*argv, env=None, timeout=240):
    import subprocess
    e = dict(os.environ)
    e.update(env or {})
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=timeout)


def test_self_launch_of_a_stalled_job_times_out_and_reports_every_rank():
    """Both ranks print their start-up line and then never come back (what a stalled RCCL rendezvous looks like from outside): the parent —
    which never touched the GPU — ends the child process group at its wall limit, prints each rank's last lines and exits 124."""
    import time
    t0 = time.time()
    p = _bench("--gpus", "2", "--launch-timeout", "12", "--no-cpu-baseline", "--test-stall-after-start")
    assert p.returncode == 124, (p.returncode, p.stderr[-2000:])
    assert time.time() - t0 < 90
    assert "TIMEOUT" in p.stderr and '"metric"' not in p.stdout
    for r in (0, 1):
        assert "[rank %d] bench.py[rank %d/2" % (r, r) in p.stderr and "started: LOCAL_RANK %d" % r in p.stderr, p.stderr[-2000:]
    # nothing of the job is left behind
    import re
    pgid = int(re.search(r"ranks run in process group (\d+)", p.stderr).group(1))
    with pytest.raises(ProcessLookupError):
        os.killpg(pgid, 0)


@pytest.mark.skipif(torch.cuda.is_available(), reason="the refusal path of a box WITHOUT a GPU")
def test_self_launch_rendezvous_works_and_ranks_refuse_without_a_gpu():
    """On this CPU-only box the launcher path runs as far as it can: two fresh ranks, a process group (gloo in the rehearsal form the
    one-GPU box uses), then every rank finds no HIP device, says so and exits 3; the parent relays the lines and fails loudly."""
    p = _bench("--gpus", "2", "--rehearse-on-device0", "--launch-timeout", "120", "--no-cpu-baseline")
    assert p.returncode not in (0, 124), (p.returncode, p.stderr[-2000:])
    assert '"metric"' not in p.stdout
    for r in (0, 1):
        assert "bench.py[rank %d/2" % r in p.stderr
    assert p.stderr.count("process group up: backend gloo") == 2, p.stderr[-3000:]
    assert "no HIP device" in p.stderr and "no result line" in p.stderr


@pytest.mark.skipif(torch.cuda.is_available(), reason="the refusal path of a box WITHOUT a GPU")
def test_eight_rank_c4_launch_reaches_the_job_collective_on_cpu():
    """BASELINE.json configs[3] as the driver would start it on an 8-GPU node — `bench.py --gpus 8 --strong --spp 1024` — rehearsed with EIGHT
    fresh ranks over gloo on this CPU-only box: the launcher brings all eight up, every rank joins the process group,
    takes part in the job's one host-side collective (the MIN over the ranks' planned batch sizes: every rank must trace the same batch for the
    sample ranges [g K S, (g + 1) K S) to tile the render) and only THEN refuses for want of a HIP device — exit 3 on every rank, relayed, no
    result line, nothing left behind.  The rendering half of the same command runs on the one-GPU box (tests/test_device_group.py)."""
    p = _bench("--gpus", "8", "--strong", "--spp", "1024", "--workload", "c1", "--rehearse-on-device0", "--launch-timeout", "200", "--no-cpu-baseline",
               env={"OMP_NUM_THREADS": "1"}, timeout=300)
    assert p.returncode not in (0, 124), (p.returncode, p.stderr[-3000:])
    assert '"metric"' not in p.stdout
    assert p.stderr.count("process group up: backend gloo") == 8, p.stderr[-4000:]
    for r in range(8):
        assert "bench.py[rank %d/8" % r in p.stderr
    assert p.stderr.count("job batch: ") == 8 and len(set(ln.split("job batch: ")[1] for ln in p.stderr.splitlines() if "job batch: " in ln)) == 1
    assert p.stderr.count("rendezvous and the batch-size all-reduce were fine") == 8


def test_a_rank_watchdog_ends_a_stalled_rank_with_its_stack():
    """Under the DRIVER's own launcher there is no parent of ours: a rank that stalls dumps its Python stacks and exits by itself."""
    import subprocess
    e = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rank-timeout", "3", "--test-stall-after-start"], env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=60)
    assert p.returncode != 0
    assert "started: LOCAL_RANK 0" in p.stderr and "Timeout (0:00:03)!" in p.stderr and "bench.py" in p.stderr
