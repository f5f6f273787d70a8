#!/usr/bin/env python3
"""Host probe (tests/emu, the product's traversal compiled for the host): node visits, triangle tests and leaf-slot fetches per closest-hit
ray with one triangle per leaf slot and with edge-sharing pairs (r4), on the product's tree form (Morton + PLOC 8 + SAH collapse, 6-wide).
    python tools/pairs_probe.py c3 c2 c5"""
import ctypes as C
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.update(EMU_MORTON="1", EMU_PLOC="8", EMU_WIDE6="1")
import emu_lib  # noqa: E402
from platinum_amd import scenes  # noqa: E402
from platinum_amd.renderer import make_params  # noqa: E402

for wl in sys.argv[1:] or ["c3"]:
    factory, W, H, spp, B = scenes.CONFIGS[wl]
    sc = factory()
    w, h = W // 8, H // 8
    for pairs in (False, True):
        if pairs:
            os.environ["EMU_PAIRS"] = "1"
        else:
            os.environ.pop("EMU_PAIRS", None)
        e = emu_lib.EmuScene(sc, make_params(w, h, 1, B))
        e.debug_sample(0)
        cnt = (C.c_ulonglong * 4)()
        e.L.emu_get_counts(cnt)
        n, t, r, l = [float(x) for x in cnt]
        e.L.emu_slot_count.argtypes = [C.c_void_p]; e.L.emu_slot_count.restype = C.c_uint32
        slots = e.L.emu_slot_count(e.h)
        print("%s %-6s rays %7d  nodes/ray %6.2f  tri tests/ray %5.2f  leaf fetches/ray %5.2f  64-B lines/ray %6.2f  slots %d" %
              (wl, "pairs" if pairs else "single", r, n / r, t / r, l / r, (n + l) / r, slots))
