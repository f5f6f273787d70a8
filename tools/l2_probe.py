#!/usr/bin/env python3
"""Host probe (tests/emu emu_l2_probe): predicted L2 misses per closest-hit ray of C3's bounce-b rays under today's chunk order
against SCENE-SPACE ray queues (rays sorted by the Morton cell of their origin, 8 contiguous cell ranges -> 8 XCDs), by replaying the 64-byte lines
the rays fetch through eight 4 MB 16-way LRU caches fed by 768 resident waves each.  Go / no-go rule: build the scheduler only if <= 1.2 (today's counters: 1.93).
    python tools/l2_probe.py [c3] [bounces=1,2,3] [T=tiles per band in the window] [s0=first tile index inside each band]"""
import ctypes as C
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.update(EMU_MORTON="1", EMU_PLOC="8", EMU_WIDE6="1", EMU_PAIRS="1")
import emu_lib  # noqa: E402
from platinum_amd import scenes  # noqa: E402
from platinum_amd.renderer import make_params  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
bounces = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3").split(",")]
T = int(sys.argv[3]) if len(sys.argv) > 3 else 0
s0 = int(sys.argv[4]) if len(sys.argv) > 4 else 4000
ns = 128
factory, W, H, spp, B = scenes.CONFIGS[wl]
e = emu_lib.EmuScene(factory(), make_params(W, H, ns, B))
e.L.emu_l2_probe.argtypes = [C.c_void_p] + [C.c_uint32] * 8 + [C.c_double * 18]
threads = len(os.sched_getaffinity(0))
names = ("today (segment order, one cursor)", "origin cells, 8 ranges -> 8 XCDs", "origin-sorted, one cursor")
for b in bounces:
    # ~1.6 M rays in the window = four generations of the 393 216 rays the chip holds in flight
    rays_per_seg = {0: 8192, 1: 8100, 2: 4460, 3: 3040, 4: 2000, 5: 1350}.get(b, 1000)
    Tb = T or max(16, int(1.6e6 / rays_per_seg / 4))
    for l2_mb, waves in ((4, 768),):
        out = (C.c_double * 18)()
        t0 = time.time()
        e.L.emu_l2_probe(e.h, b, ns, s0, Tb, waves, l2_mb << 20, 10, threads, out)
        print("%s bounce %d: window = tiles [%d, %d) of each band, %d samples; %d x %d MB L2, %d waves per XCD  (%.0f s)" % (wl, b, s0, s0 + Tb, ns, 8, l2_mb, waves, time.time() - t0))
        for m, name in enumerate(names):
            rays, touches, misses, streamed, wsteps, wmiss = [float(x) for x in out[6 * m:6 * m + 6]]
            if rays:
                print("   %-36s %9.0f rays counted: %.2f lines per ray, BVH misses %.3f per ray + %.2f queue lines streamed = %.2f L2 misses per ray; "
                      "wave-steps with >= 1 missing lane: %.1f %% of %.1f per chunk" % (name, rays, touches / rays, misses / rays, streamed / rays,
                                                                                       (misses + streamed) / rays, 100.0 * wmiss / wsteps, wsteps / (rays / 64.0)))
