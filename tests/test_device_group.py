"""The device group behind the C ABI (include/ptamd.h pt_create_info.device_ordinals, platinum_amd/csrc/multi_device.hip;
SURVEY §8e): samples dealt to the members in contiguous ranges, ONE reduction of the float accumulator at the end, GMoN
buckets mapped to devices and resolved on the first one.

CPU: the partition arithmetic (pt_group_partition).  GPU (one device is enough: a device listed twice = two logical shards
with their own host threads, streams, scene copies and accumulators, merged by the same code path minus RCCL; with two
or more devices present the RCCL all-reduce itself is exercised)."""
import ctypes as C
import os

import numpy as np
import pytest

from platinum_amd import Renderer, abi, scenes


def _partition(spp, members, flags=abi.FLAG_MULTISCATTER_GGX, buckets=1):
    lib = abi.load_library()
    first, count = (C.c_uint64 * members)(), (C.c_uint64 * members)()
    b0, b1 = (C.c_uint32 * members)(), (C.c_uint32 * members)()
    abi.check(lib, lib.pt_group_partition(spp, members, flags, buckets, first, count, b0, b1))
    return list(first), list(count), list(b0), list(b1)


def test_partition_is_a_contiguous_cover_of_the_sample_range():
    for spp, n in [(1024, 8), (128, 2), (5, 2), (7, 3), (3, 8), (1, 4), (1000, 7)]:
        first, count, _, _ = _partition(spp, n)
        assert first[0] == 0 and sum(count) == spp
        assert all(first[g + 1] == first[g] + count[g] for g in range(n - 1))
        assert max(count) - min(count) <= 1
    # BASELINE.json configs[3]: 1024 spp = 128 spp x 8 seeds
    first, count, _, _ = _partition(1024, 8)
    assert count == [128] * 8 and first == [128 * g for g in range(8)]


def test_partition_follows_gmon_bucket_boundaries():
    """bucket b = samples [b * ceil(spp / B), ...) (renderer_pt.cpp:124-126): every bucket lives on exactly one member."""
    flags = abi.FLAG_MULTISCATTER_GGX | abi.FLAG_GMON
    for spp, buckets, n in [(128, 15, 8), (30, 15, 2), (17, 5, 3), (100, 32, 8), (10, 3, 8)]:
        first, count, b0, b1 = _partition(spp, n, flags, buckets)
        spb = -(-spp // buckets)
        assert b0[0] == 0 and b1[-1] == buckets and all(b1[g] == b0[g + 1] for g in range(n - 1))
        assert sum(count) == spp
        for g in range(n):
            assert first[g] == min(b0[g] * spb, spp) and first[g] + count[g] == min(b1[g] * spb, spp)
    lib = abi.load_library()
    assert lib.pt_group_partition(16, 2, flags, 33, (C.c_uint64 * 2)(), (C.c_uint64 * 2)(), None, None) != 0


def test_partition_with_fewer_buckets_than_members_starts_at_member_zero():
    """The reference UI allows 5..25 buckets (pt_viewport.cpp), so 5..7 buckets on 8 GPUs happens: the remainder buckets go to
    the LOWEST-numbered members — member 0, whose device resolves / post-processes / presents the image, always has samples."""
    flags = abi.FLAG_MULTISCATTER_GGX | abi.FLAG_GMON
    for spp, buckets, n in [(128, 5, 8), (128, 7, 8), (64, 6, 8), (12, 4, 8), (30, 15, 8), (9, 1, 4)]:
        first, count, b0, b1 = _partition(spp, n, flags, buckets)
        owned = [b1[g] - b0[g] for g in range(n)]
        assert sum(owned) == buckets and owned == sorted(owned, reverse=True) and max(owned) - min(owned) <= 1
        assert count[0] > 0 and owned[0] >= 1
        assert all(count[g] == 0 for g in range(n) if owned[g] == 0)


def test_rccl_library_loads_and_exports_the_entry_points_the_merge_calls():
    """multi_device.hip dlopens librccl.so on first use by a group over >= 2 distinct GPUs; that moment never comes on a
    one-GPU box, so the load + symbol binding is checked here (no GPU is touched).  The function-pointer types and the
    ncclFloat32 / ncclSum values are checked at COMPILE time against the image's <rccl/rccl.h>."""
    lib = abi.load_library()
    rc = lib.pt_rccl_probe()
    assert rc == 0, lib.pt_last_error()
    assert lib.pt_rccl_probe() == 0  # idempotent


W, H, B = 96, 64, 5


def _render(devices, scene, spp, **kw):
    r = Renderer(devices=devices) if devices is not None else Renderer(device=0)
    try:
        r.startRender(scene, (W, H), spp, max_bounces=B, **kw)
        assert r.status() == Renderer.Status_Busy
        r.render(0)
        r.wait()
        assert r.status() == (Renderer.Status_Ready | Renderer.Status_Done)
        assert r.renderProgress() == (spp, spp)
        return r.readbackAccumulator(), r.stats(), r
    except Exception:
        r.close()
        raise


@pytest.mark.gpu
@pytest.mark.parametrize("spp,shards", [(8, 2), (5, 2), (9, 3), (2, 4)])
def test_logical_shards_equal_the_single_render(spp, shards):
    """Two (three, four) members on device 0 against one renderer tracing the union of their sample ranges: the same image
    up to fp32 summation order (<= 1e-6 relative), alpha 1, ray counters add up."""
    sc = scenes.cornell_sphere_scene()
    ref, st1, r1 = _render(None, sc, spp, samples_in_flight=3)
    r1.close()
    got, stn, rn = _render([0] * shards, sc, spp, samples_in_flight=3)
    try:
        assert np.array_equal(got[..., 3], np.ones((H, W), np.float32))
        np.testing.assert_allclose(got[..., :3], ref[..., :3], rtol=2e-6, atol=1e-6)
        assert (stn.closest_rays, stn.shadow_rays, stn.shaded_hits, stn.paths) == (st1.closest_rays, st1.shadow_rays, st1.shaded_hits, st1.paths)
        # progressive use: a second render on the same group, read in the middle
        rn.startRender(sc, (W, H), spp, max_bounces=B, first_sample=3)
        rn.render(shards)             # one sample per member
        mid = rn.readbackAccumulator()
        assert np.isfinite(mid).all() and mid[..., :3].max() > 0
        rn.render(0)
        full = rn.readbackAccumulator()
    finally:
        rn.close()
    r1 = Renderer(device=0)
    try:
        r1.startRender(sc, (W, H), spp, max_bounces=B, first_sample=3)
        r1.render(0)
        np.testing.assert_allclose(full[..., :3], r1.readbackAccumulator()[..., :3], rtol=2e-6, atol=1e-6)
    finally:
        r1.close()


@pytest.mark.gpu
def test_group_writes_the_merged_image_into_an_external_accumulator():
    torch = pytest.importorskip("torch")
    sc = scenes.cornell_scene("bench")
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    r = Renderer(devices=[0, 0])
    try:
        r.startRender(sc, (W, H), 6, max_bounces=B, external_accumulator=acc.data_ptr())
        r.render(0)
        r.wait()
        assert r.accumulatorDevicePtr() == acc.data_ptr()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(acc.cpu().numpy(), r.readbackAccumulator())
    finally:
        r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("spp,buckets,shards", [(30, 15, 2), (17, 5, 3), (12, 4, 8)])
def test_gmon_buckets_map_to_members_and_resolve_bit_identically(spp, buckets, shards):
    """SURVEY §8e last sentence: buckets -> devices, bucket means gathered, k_gmon on the first device: every bucket image and
    the resolved accumulator are BIT-identical to the single-device render."""
    sc = scenes.cornell_sphere_scene()
    flags = abi.FLAG_MULTISCATTER_GGX | abi.FLAG_GMON
    ref, _, r1 = _render(None, sc, spp, gmonBuckets=buckets, flags=flags, samples_in_flight=4)
    ref_b = [r1.readGmonBucket(b) for b in range(buckets)]
    r1.close()
    got, _, rn = _render([0] * shards, sc, spp, gmonBuckets=buckets, flags=flags, samples_in_flight=4)
    try:
        for b in range(buckets):
            assert rn.readGmonBucket(b).tobytes() == ref_b[b].tobytes(), b
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        img = rn.readbackRenderTarget()
        assert img.shape == (H, W, 4) and img[..., 3].min() == 255
    finally:
        rn.close()


@pytest.mark.gpu
def test_failed_restart_leaves_the_group_without_a_render():
    """a restart that fails validation must not leave members of the PREVIOUS render 'started' beside freed merge
    images (pt_wait would then launch the merge kernel on null scratch).  Invalid parameters are refused before anything is
    torn down; a failure past that point abandons the render as a whole."""
    sc = scenes.cornell_scene("bench")
    r = Renderer(devices=[0, 0])
    try:
        r.startRender(sc, (W, H), 4, max_bounces=B)
        r.render(2)
        with pytest.raises(abi.PtamdError):
            r.startRender(sc, (W, H), 4, max_bounces=51)       # refused up front: the first render is untouched
        first = r.readbackAccumulator()
        assert np.isfinite(first).all() and first[..., :3].max() > 0
        r.render(0)
        done = r.readbackAccumulator()
        with pytest.raises(abi.PtamdError):
            r.startRender(sc, (W, H), 4, max_bounces=B, accel_structure=7)
        np.testing.assert_array_equal(r.readbackAccumulator(), done)
        # a restart that is valid still works afterwards
        r.startRender(sc, (W, H), 2, max_bounces=B)
        r.render(0)
        assert r.renderProgress() == (2, 2) and np.isfinite(r.readbackAccumulator()).all()
    finally:
        r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("buckets", [5, 7])
def test_gmon_with_fewer_buckets_than_members_presents_on_the_first_device(buckets):
    sc = scenes.cornell_sphere_scene()
    flags = abi.FLAG_MULTISCATTER_GGX | abi.FLAG_GMON
    spp = 2 * buckets
    ref, _, r1 = _render(None, sc, spp, gmonBuckets=buckets, flags=flags, samples_in_flight=4)
    ref_img = r1.readbackRenderTarget()
    r1.close()
    got, _, rn = _render([0] * 8, sc, spp, gmonBuckets=buckets, flags=flags, samples_in_flight=4)
    try:
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        assert np.array_equal(rn.readbackRenderTarget(), ref_img)
    finally:
        rn.close()


@pytest.mark.gpu
def test_rccl_calls_of_the_merge_run_on_one_device():
    """The distinct-device merge (ncclCommInitAll, grouped ncclAllReduce(sum, f32) in place on the member's stream, ncclCommDestroy) has
    never met hardware with two GPUs; with ONE rank every one of its calls still runs for real."""
    lib = abi.load_library()
    rc = lib.pt_rccl_selftest(0)
    assert rc == 0, lib.pt_last_error()
    # ... and a renderer created afterwards still works (RCCL's own streams / allocations do not disturb it)
    r = Renderer(device=0)
    try:
        r.startRender(scenes.cornell_scene("bench"), (W, H), 2, max_bounces=B)
        r.render(0)
        assert np.isfinite(r.readbackAccumulator()).all()
    finally:
        r.close()


@pytest.mark.gpu
def test_torch_distributed_nccl_backend_reduces_the_accumulator_with_one_rank():
    """bench.py's N > 1 path in miniature: process group on the `nccl` backend (= RCCL on ROCm), the library renders into a torch tensor,
    sharding.reduce_accumulator all-reduces it.  One rank: the reduction must leave the image as it is."""
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    from platinum_amd.sharding import reduce_accumulator
    import os, socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
        r = Renderer(device=0)
        try:
            r.startRender(scenes.cornell_scene("bench"), (W, H), 3, max_bounces=B, external_accumulator=acc.data_ptr())
            r.render(0)
            r.wait()
            before = acc.clone()
            # world = 1 short-circuits in reduce_accumulator; call the collective itself, as world > 1 would
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
            assert torch.equal(acc, before) and float(acc[..., :3].max()) > 0
            assert reduce_accumulator(acc, 1, dist) is acc
        finally:
            r.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_group_argument_errors():
    lib = abi.load_library()
    info = abi.CreateInfo()
    info.abi_version = abi.PT_ABI_VERSION
    info.lut_path = abi.LUT_PATH.encode()
    info.device_count = 2
    h = C.c_void_p()
    assert lib.pt_create(C.byref(info), C.byref(h)) != 0          # count without a list
    devs = (C.c_int32 * 2)(0, 9999)
    info.device_ordinals = devs
    assert lib.pt_create(C.byref(info), C.byref(h)) != 0 and b"ordinal" in lib.pt_last_error()


@pytest.mark.gpu
def test_two_physical_devices_rccl_all_reduce():
    """The RCCL path proper (needs >= 2 GPUs; the 1-GPU box skips it)."""
    torch = pytest.importorskip("torch")
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU: the all-reduce over distinct devices cannot run here")
    sc = scenes.cornell_sphere_scene()
    ref, _, r1 = _render(None, sc, 8)
    r1.close()
    got, _, rn = _render([0, 1], sc, 8)
    rn.close()
    np.testing.assert_allclose(got[..., :3], ref[..., :3], rtol=2e-6, atol=1e-6)


@pytest.mark.gpu
def test_bench_self_launch_runs_two_ranks_on_this_gpu_and_relays_one_json_line():
    """`python bench.py --gpus 2` as the driver may start it WITHOUT a launcher, rehearsed on the one GPU of this box (every rank on device 0, gloo
    for the reduce since RCCL refuses two ranks on one device): the parent starts torch.distributed.run as a child before touching the GPU, the two
    ranks shard the samples, reduce the accumulator and take the max-over-ranks time, rank 0's line comes back through the parent, and every rank's
    start-up lines (rank, device, process group, runtime) are on stderr.  What differs on an 8-GPU node is the backend name and the device ordinal."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-on-device0", "--workload", "c1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-kernel-pass", "--launch-timeout", "300"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    assert p.returncode == 0, (p.returncode, p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["config"]["logical_shards"] == 2 and d["n_gpus"] == 1 and "REHEARSAL" in d["config"]["parallelism"] and d["value"] > 0
    assert d["config"]["spp_total"] == 2 * d["config"]["spp_per_gpu"] and d["extra"]["paths"] > 0
    for r in (0, 1):
        assert "[rank %d] bench.py[rank %d/2" % (r, r) in p.stderr and "process group up: backend gloo" in p.stderr and "renderer on device 0" in p.stderr


@pytest.mark.gpu
def test_bench_c4_command_with_four_ranks_on_this_gpu_renders_1024_samples():
    """BASELINE.json configs[3]'s command line — `bench.py --gpus N --strong --spp 1024` — with as many ranks as this pool allows on one card
    (four; eight are rehearsed on CPU as far as the job's collective: tests/test_multi_gpu_gloo.py): the MIN-of-plans all-reduce, sample ranges
    [256 g, 256 g + 256), ONE reduce of the accumulator inside the timed region, ONE JSON line with `spp_total` = 1024 and `scaling` = strong."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--strong", "--spp", "1024", "--rehearse-on-device0", "--workload", "c1",
                        "--warmup", "1", "--no-cpu-baseline", "--no-kernel-pass", "--launch-timeout", "300"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=400)
    assert p.returncode == 0, (p.returncode, p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["scaling"] == "strong" and d["config"]["spp_total"] == 1024 and d["config"]["spp_per_gpu"] == 256 and d["config"]["logical_shards"] == 4
    assert d["extra"]["paths"] == 512 * 512 * 256 and d["extra"]["time_to_image_ms"] > 0
    assert p.stderr.count("job batch: ") == 4 and p.stderr.count("process group up: backend gloo") == 4
