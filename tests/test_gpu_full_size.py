"""The configurations `bench.py` TIMES, compared with the oracle at their full size.

`python bench.py` renders C3 at 1920x1080 with the batch the library plans for the image (128 samples in flight on an empty MI355X:
32 400 segments, four bands, ~53 GB of queues); `--workload c2` the same for C2, `--workload c5` the ingested atrium at 3840x2160 with
12 bounces.  A whole oracle frame of those sizes takes minutes, so each test here
  * renders exactly what one bench step renders (same scene object, same size, same bounces, samples_in_flight = the library's own
    plan for an empty device, the DEFAULT non-finite policy), reads the accumulator through the C ABI and compares ~1 000 pixels — image corners, both sides
    of 8x8 tile borders, both sides of every segment-band border, object silhouettes found in the primary-hit ids, and a seeded random
    set — with the oracle's running mean over the same samples (`orc_render_pixels`: the loop of `orc_render` for a list of pixels),
    BIT FOR BIT (kernel.metal:672-684);
  * compares the ray / shadow-ray / shaded-hit counters of a full-size two-sample render with the oracle's whole-frame counters;
  * renders a second planned batch on top and compares the probe pixels again (the running mean's weights with samples already in it);
  * requires the whole 128-in-flight image to equal, bit for bit, the image the same library produces in batches of 16 (different segment
    fill, different chunk tables, different accumulate folds — the same samples in the same order);
  * (r6) compares 24 of the probe pixels x 4 samples with the oracle's BRUTE-FORCE loop over every triangle — the contract's definition of a
    hit — so that the oracle's own tree is not the only reference at full size.
"""
import os
import sys
import time

import numpy as np
import pytest

from platinum_amd import abi, scenes
from platinum_amd.renderer import make_params

import oracle_lib

pytestmark = pytest.mark.gpu


def _group_partition(spp, members):
    import ctypes as C
    lib = abi.load_library()
    first, count = (C.c_uint64 * members)(), (C.c_uint64 * members)()
    abi.check(lib, lib.pt_group_partition(spp, members, abi.FLAG_MULTISCATTER_GGX, 1, first, count, None, None))
    return list(first), list(count), None, None


def _same_bits_or_both_nan(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())


def _probe_pixels(W, H, ids, n_random=512, seed=5):
    """Pixels where a wavefront bug would show first.  `ids` = primary-hit instance ids [H, W] (silhouettes)."""
    px = {(0, 0), (W - 1, 0), (0, H - 1), (W - 1, H - 1), (W // 2, H // 2)}
    rng = np.random.default_rng(seed)
    tx_n, ty_n = (W + 7) // 8, (H + 7) // 8
    # both sides of tile borders (a tile = one wave of k_raygen, one segment's share of the queues)
    for _ in range(48):
        tx, ty = int(rng.integers(1, tx_n)), int(rng.integers(1, ty_n))
        for dx in (-1, 0):
            for dy in (-1, 0):
                px.add((min(W - 1, 8 * tx + dx), min(H - 1, 8 * ty + dy)))
    # both sides of the segment-band borders (kernels.hip seg_first_tile: the tile list is cut into 4 contiguous bands)
    ntiles = tx_n * ty_n
    for k in range(1, 4):
        for t in (k * ntiles // 4 - 1, k * ntiles // 4):
            tx, ty = t % tx_n, t // tx_n
            for d in (0, 7):
                px.add((min(W - 1, 8 * tx + d), min(H - 1, 8 * ty + d)))
    # the last (partial) tile row / column
    px.update({(W - 1, H // 3), (W // 3, H - 1), (8 * (tx_n - 1), 8 * (ty_n - 1))})
    # silhouettes: pixels whose right or lower neighbour sees another instance
    edge = np.zeros(ids.shape, bool)
    edge[:, :-1] |= ids[:, :-1] != ids[:, 1:]
    edge[:-1, :] |= ids[:-1, :] != ids[1:, :]
    ey, ex = np.nonzero(edge)
    if len(ex):
        for i in rng.choice(len(ex), size=min(256, len(ex)), replace=False):
            px.add((int(ex[i]), int(ey[i])))
    for _ in range(n_random):
        px.add((int(rng.integers(0, W)), int(rng.integers(0, H))))
    return np.array(sorted(px), dtype=np.uint32)


def _full_size_case(r, scene, W, H, B, spp_expected=None, counters_spp=2):
    t_start = time.perf_counter()
    r.selectKernel(abi.INTEGRATOR_MIS)
    lib = abi.load_library()
    import ctypes as C
    import torch
    free_b, total_b = torch.cuda.mem_get_info(0)
    plan = abi.QueuePlan()
    # the batch bench.py gets in its own fresh process: the plan for an EMPTY device (this session's renderer still holds the queues of earlier
    # tests, which a plan against the memory free right now would count against the batch: 39 instead of 46 samples at 3840x2160)
    abi.check(lib, lib.pt_plan_queues(W, H, 1 << 20, 0, int(total_b * 0.995), 0, 4, C.byref(plan)))
    S = int(plan.samples_in_flight)            # what bench.py calls spp_per_step
    if spp_expected is not None and total_b > 250 << 30:
        assert S == spp_expected, (S, spp_expected)
    o = oracle_lib.OracleScene(scene, make_params(W, H, S, B))
    # ---- counters of a full-size render of `counters_spp` samples against the oracle's whole frame ------------------------------------
    r.startRender(scene, (W, H), counters_spp, max_bounces=B)
    ids = r.tracePrimary(0)["instance"]
    r.render(0)
    r.wait()
    st2 = r.stats()
    o.render(0, counters_spp)
    so = o.stats()
    assert (st2.paths, st2.closest_rays, st2.shadow_rays, st2.shaded_hits) == (so.paths, so.closest_rays, so.shadow_rays, so.shaded_hits)
    # ---- one bench step: the library's own batch for this image ----------------------------------------------------------------------
    r.startRender(scene, (W, H), S, max_bounces=B, samples_in_flight=S)   # as bench.py passes it; non-finite policy = the default (propagate)
    assert r.stats().samples_in_flight == S
    r.render(S)
    r.wait()
    st = r.stats()
    assert st.batches == 1 and st.paths == W * H * S
    acc = r.readbackAccumulator()
    xy = _probe_pixels(W, H, ids)
    ref = o.render_pixels(xy, 0, S)
    got = acc[xy[:, 1], xy[:, 0]]
    bad = ~((got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))).all(axis=1)
    assert not bad.any(), "pixels %s: HIP %s oracle %s" % (xy[bad][:4].tolist(), got[bad][:4].tolist(), ref[bad][:4].tolist())
    assert (acc[..., 3] == 1).all() and np.nanmean(acc[..., :3]) > 1e-3
    # ---- a SECOND planned batch on top (samples S .. 2S-1 folded into a mean that already holds S): the running-mean weights at full size --
    r.startRender(scene, (W, H), 2 * S, max_bounces=B, samples_in_flight=S)
    r.render(0)
    r.wait()
    assert r.stats().batches == 2
    acc2 = r.readbackAccumulator()
    xy2 = xy[::4]   # (a quarter of the probe pixels: the oracle walks 2 S samples for each)
    ref2 = o.render_pixels(xy2, 0, 2 * S)
    got2 = acc2[xy2[:, 1], xy2[:, 0]]
    assert (((got2.view(np.uint32) == ref2.view(np.uint32)) | (np.isnan(got2) & np.isnan(ref2))).all(axis=1)).all()
    # ---- the CONTRACT, not only the oracle's tree: a few of the probe pixels, first four samples, answered by the oracle's brute-force
    #      loop over every triangle (min t, lowest id; any-hit = exists) — the definition the tree is merely a fast way to evaluate ------------------
    xyb = xy[:: max(1, len(xy) // 24)][:24]
    ob = oracle_lib.OracleScene(scene, make_params(W, H, 4, B), use_bvh=False)
    refb = ob.render_pixels(xyb, 0, 4)
    ob.close()
    r.startRender(scene, (W, H), 4, max_bounces=B)
    r.render(0)
    r.wait()
    gotb = r.readbackAccumulator()[xyb[:, 1], xyb[:, 0]]
    assert (((gotb.view(np.uint32) == refb.view(np.uint32)) | (np.isnan(gotb) & np.isnan(refb))).all(axis=1)).all(), "HIP path != brute force"
    # ---- the same samples in batches of 16: the whole image, bit for bit --------------------------------------------------------------
    r.startRender(scene, (W, H), S, max_bounces=B, samples_in_flight=16)
    r.render(0)
    r.wait()
    assert r.stats().batches == -(-S // 16)
    assert _same_bits_or_both_nan(r.readbackAccumulator(), acc)
    o.close()
    return S, len(xy), time.perf_counter() - t_start


def test_c3_as_the_driver_times_it_equals_the_oracle(gpu_renderer):
    """BASELINE.json configs[2] — `python bench.py`: 1.04 M-triangle field, 1920x1080, 8 bounces, 128 samples in flight."""
    factory, W, H, _spp, B = scenes.CONFIGS["c3"]
    S, n, dt = _full_size_case(gpu_renderer, factory(), W, H, B, spp_expected=128)
    print("C3 full size: %d samples in flight, %d probe pixels bit-identical, %.1f s" % (S, n, dt))


def test_c2_as_the_driver_times_it_equals_the_oracle(gpu_renderer):
    """BASELINE.json configs[1] — `python bench.py --workload c2`: Cornell box + GGX dielectric sphere, 1920x1080, 8 bounces."""
    factory, W, H, _spp, B = scenes.CONFIGS["c2"]
    S, n, dt = _full_size_case(gpu_renderer, factory(), W, H, B, spp_expected=128)
    print("C2 full size: %d samples in flight, %d probe pixels bit-identical, %.1f s" % (S, n, dt))


def test_c5_as_the_driver_times_it_equals_the_oracle(gpu_renderer, tmp_path):
    """BASELINE.json configs[4] on one GPU — `python bench.py --workload c5`: the atrium written as .glb + .exr, read back by the
    product's loaders, 3840x2160, 12 bounces, the planned batch (46 samples in flight on an empty MI355X)."""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import export_gltf
    _factory, W, H, _spp, B = scenes.CONFIGS["c5"]
    sc = export_gltf.atrium_through_ingestion(str(tmp_path))
    S, n, dt = _full_size_case(gpu_renderer, sc, W, H, B, spp_expected=46, counters_spp=1)
    print("C5 full size: %d samples in flight, %d probe pixels bit-identical, %.1f s" % (S, n, dt))


def test_c4_eight_members_of_128_samples_equal_the_oracle(gpu_renderer):
    """BASELINE.json configs[3] (C4) at its stated size, as far as one GPU can show it: the C3 scene, 1920x1080, 8 bounces, 1024 spp dealt as
    8 x 128 — member g renders frameIdx in [128 g, 128 g + 128) (`pt_group_partition`; the reference's only seed is frameIdx,
    samplers.metal:154-156) into a private running mean, ONE reduction at the end.
      * every member's PRIVATE mean — one renderer with first_sample = 128 g, incl. g = 7 whose range starts 896 deep in the Halton index —
        equals the oracle's mean over that range BIT FOR BIT on the probe pixels;
      * the device group `Renderer(devices=[0] * 8)` (eight logical members: own threads, streams, scene copies, BVHs, queues; the merge path
        of multi_device.hip minus RCCL) produces, bit for bit, the fp32 fold of those private means in member order times 1/8 (what
        `ncclAllReduce(sum)` + scale computes up to its own summation order) — the WHOLE image;
      * and that merged image equals the oracle's 1024-sample running mean to 1e-5 relative.  That tolerance is summation order only, and it is
        the SEQUENTIAL mean that carries most of it: kernel.metal:672-684 rounds twice per sample, so 1024 samples leave a random walk of
        ~sqrt(2048) x 6e-8 = 2.7e-6 (observed maximum over 2 172 values: 3.4e-6), the eight 128-sample means a third of that.  (96 x 64 x <= 9
        samples hold SURVEY section 8c's 2e-6: tests/test_device_group.py.)"""
    from platinum_amd import Renderer
    t_start = time.perf_counter()
    factory, W, H, _spp, B = scenes.CONFIGS["c3"]
    N, S = 8, 128
    scene = factory()
    first, count, _, _ = _group_partition(N * S, N)
    assert count == [S] * N and first == [S * g for g in range(N)]
    r = gpu_renderer
    r.selectKernel(abi.INTEGRATOR_MIS)
    r.startRender(scene, (W, H), 1, max_bounces=B)
    ids = r.tracePrimary(0)["instance"]
    xy = _probe_pixels(W, H, ids, n_random=256)
    o = oracle_lib.OracleScene(scene, make_params(W, H, N * S, B))
    fold = None
    for g in range(N):
        r.startRender(scene, (W, H), S, max_bounces=B, first_sample=first[g])   # the library's own batch, as a rank of bench.py --strong renders it
        r.render(0)
        r.wait()
        st = r.stats()
        assert st.paths == W * H * S
        a = r.readbackAccumulator()
        ref = o.render_pixels(xy, first[g], S)
        got = a[xy[:, 1], xy[:, 0]]
        bad = ~((got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))).all(axis=1)
        assert not bad.any(), "member %d, pixels %s: HIP %s oracle %s" % (g, xy[bad][:4].tolist(), got[bad][:4].tolist(), ref[bad][:4].tolist())
        fold = a[..., :3].copy() if fold is None else fold + a[..., :3]     # k_weighted_add: o += 1.0f * a, member order, fp32
    fold *= np.float32(1.0 / N)
    grp = Renderer(devices=[0] * N)
    try:
        grp.startRender(scene, (W, H), N * S, max_bounces=B, samples_in_flight=32)   # (8 x 32 in flight = 8 x 13 GB; the image does not depend on the batching)
        grp.render(0)
        grp.wait()
        assert grp.renderProgress() == (N * S, N * S)
        stg = grp.stats()
        assert stg.paths == W * H * N * S
        merged = grp.readbackAccumulator()
    finally:
        grp.close()
    assert (merged[..., 3] == 1).all()
    assert _same_bits_or_both_nan(merged[..., :3], fold), "the merge is not the fold of the members' private means"
    ref = o.render_pixels(xy, 0, N * S)
    got = merged[xy[:, 1], xy[:, 0]]
    np.testing.assert_allclose(got[:, :3], ref[:, :3], rtol=1e-5, atol=1e-7)
    worst = float(np.max(np.abs(got[:, :3] - ref[:, :3]) / np.maximum(np.abs(ref[:, :3]), 1e-3)))
    o.close()
    print("C4 full size: 8 members x 128 samples, %d probe pixels: private means bit-identical, merge == fold of the private means (whole image), "
          "vs 1024-sample sequential oracle mean: max relative difference %.2g; %.1f s" % (len(xy), worst, time.perf_counter() - t_start))


def test_c3xl_a_structure_beyond_the_infinity_cache_equals_the_oracle(gpu_renderer):
    """`bench.py --workload c3xl` (context): the field with 128 x 128 instances = 16.6 M triangles — leaf slots + 6-wide nodes +
    shade records ~ 1.3 GB, five times the 256 MB Infinity Cache, where C3's 47 MB structure sits inside it.  Same code path as C3 (one BVH over
    the flattened triangles, built on the GPU); what changes is where the lines come from, and the index ranges (8.5 M leaf slots, 27 bits of
    triangle index in the hit word).  At a resolution the oracle (its own median-split tree over the same 16.6 M triangles: ~35 s to build) walks
    in seconds: the whole image, ray counters and primary hits bit-identical."""
    t_start = time.perf_counter()
    factory, _W, _H, _spp, B = scenes.CONFIGS["c3xl"]
    W, H, S = 480, 270, 8
    scene = factory()
    r = gpu_renderer
    r.selectKernel(abi.INTEGRATOR_MIS)
    r.startRender(scene, (W, H), S, max_bounces=B)
    prim = r.tracePrimary(0)
    r.render(0)
    r.wait()
    st = r.stats()
    assert st.triangles == 12 + 128 * 128 * 1012
    acc = r.readbackAccumulator()
    o = oracle_lib.OracleScene(scene, make_params(W, H, S, B))
    ref = o.render(0, S)
    so = o.stats()
    assert (st.paths, st.closest_rays, st.shadow_rays, st.shaded_hits) == (so.paths, so.closest_rays, so.shadow_rays, so.shaded_hits)
    oprim = o.trace_primary(0)
    for f in ("instance", "primitive"):
        assert np.array_equal(prim[f], oprim[f])
    assert np.array_equal(prim["t"].view(np.uint32), oprim["t"].view(np.uint32))
    assert _same_bits_or_both_nan(acc, ref)
    o.close()
    print("C3XL: %d triangles, %d BVH nodes, %dx%d x %d samples bit-identical to the oracle, %.1f s" % (st.triangles, st.bvh_nodes, W, H, S, time.perf_counter() - t_start))
