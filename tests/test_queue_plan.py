"""How pt_start_render sizes the wavefront queues (platinum_amd/csrc/queue_plan.h, exported as pt_plan_queues): every index
width the kernels rely on is a limit here.  VERDICT r2 / ADVICE r2: k_shade bins segment slot numbers as 16-bit values, so a
segment may hold at most 65536 path slots — an explicit samples_in_flight (or $PTAMD_TILES_PER_SEG) beyond that is halved,
never silently truncated."""
import ctypes as C

import numpy as np
import pytest

from platinum_amd import Renderer, abi, scenes


def plan(w, h, spp, sif=0, free=200 << 30, tps_override=0, bands=4):
    lib = abi.load_library()
    q = abi.QueuePlan()
    rc = lib.pt_plan_queues(w, h, spp, sif, free, tps_override, bands, C.byref(q))
    return rc, q


def tiles(w, h):
    return ((w + 7) // 8) * ((h + 7) // 8)


@pytest.mark.parametrize("w,h", [(1920, 1080), (3840, 2160), (7680, 4320), (512, 512), (16384, 16384), (33, 17), (8, 120000)])
@pytest.mark.parametrize("sif", [0, 1, 64, 65, 128, 256, 1000])
def test_every_index_limit_holds(w, h, sif):
    rc, q = plan(w, h, 4096, sif)
    assert rc == 0
    t = tiles(w, h)
    assert 1 <= q.samples_in_flight <= 256
    assert q.seg_cap == q.tiles_per_seg * q.samples_in_flight * 64 <= 65536            # uint16 slot numbers in k_shade's bins
    assert q.nseg <= 32768 and q.nseg % 4 == 0 and q.nseg * q.tiles_per_seg >= t      # 16 bits of segment id in the chunk tables
    assert q.capacity >= w * h * q.samples_in_flight and q.capacity < 2 ** 32           # 32-bit queue slots
    assert q.lbuf_entries == t * 64 * q.samples_in_flight < 2 ** 31
    assert q.seg_cap // 64 < 65536                                                       # 16 bits of chunk index in the chunk tables
    if sif:  # an explicit request is only ever lowered, by halving
        assert q.samples_in_flight <= sif and (sif > 256 or any(q.samples_in_flight == min(sif, 4096) >> k for k in range(9)))


def test_the_cases_the_advisor_named():
    # $PTAMD_TILES_PER_SEG = 32 with 64 samples in flight: 32 * 64 * 64 = 131072 slots would wrap the 16-bit bins
    rc, q = plan(1920, 1080, 256, 64, tps_override=32)
    assert rc == 0 and q.tiles_per_seg == 32 and q.samples_in_flight == 32 and q.seg_cap == 65536
    # ~131 000 tiles (tiles_per_seg 5) at 256 samples in flight
    rc, q = plan(8 * 131000, 8, 256, 256)
    assert rc == 0 and q.tiles_per_seg == 5 and q.samples_in_flight == 128 and q.seg_cap == 40960
    # an 8K image with an explicit 65 samples in flight (tiles_per_seg 16): 16 * 65 * 64 = 66560 > 65536
    rc, q = plan(7680, 4320, 256, 65)
    assert rc == 0 and q.tiles_per_seg == 16 and q.samples_in_flight == 32
    # the auto size never exceeded the limit and gives 128 samples at 1080p on an empty MI355X (53 GB of queues), 45 at 4K
    rc, q = plan(1920, 1080, 256, 0, free=280 << 30)
    assert rc == 0 and q.samples_in_flight == 128 and q.tiles_per_seg == 1 and q.nseg == 32400
    rc, q = plan(3840, 2160, 512, 0, free=280 << 30)
    assert rc == 0 and q.samples_in_flight == 45 and q.tiles_per_seg == 4


def test_refusals():
    lib = abi.load_library()
    assert plan(0, 10, 1)[0] != 0 and plan(10, 10, 0)[0] != 0
    assert plan(20000, 20000, 1)[0] != 0 and b"too large" in lib.pt_last_error()
    assert lib.pt_plan_queues(8, 8, 1, 0, 0, 0, 4, None) != 0


@pytest.mark.gpu
def test_tall_strip_at_exactly_65536_slots_per_segment_matches_small_batches():
    """One 8-pixel-wide strip, $PTAMD_TILES_PER_SEG = 16 tiles per segment, 64 samples in flight: seg_cap = 65536 exactly, the
    last slot number is 0xffff.  The image must equal the same render traced 4 samples at a time (bit for bit: the running mean
    folds samples in index order whatever the batch size), and a request for 128 in flight must come back as 64."""
    import os
    if "PTAMD_TILES_PER_SEG" in os.environ:
        pytest.skip("$PTAMD_TILES_PER_SEG is preset for this session: the test sets and removes it itself")
    sc = scenes.cornell_scene("bench")
    w, h, spp, B = 8, 512, 64, 4
    os.environ["PTAMD_TILES_PER_SEG"] = "16"
    try:
        r = Renderer(device=0)
    finally:
        del os.environ["PTAMD_TILES_PER_SEG"]
    try:
        r.startRender(sc, (w, h), spp, max_bounces=B, samples_in_flight=128)
        assert r.stats().samples_in_flight == 64
        r.render(0)
        big = r.readbackAccumulator()
        st_big = r.stats()
    finally:
        r.close()
    r = Renderer(device=0)
    try:
        r.startRender(sc, (w, h), spp, max_bounces=B, samples_in_flight=4)
        r.render(0)
        small = r.readbackAccumulator()
        st_small = r.stats()
    finally:
        r.close()
    assert np.array_equal(big.view(np.uint32), small.view(np.uint32))
    assert (st_big.closest_rays, st_big.shadow_rays, st_big.shaded_hits) == (st_small.closest_rays, st_small.shadow_rays, st_small.shaded_hits)
