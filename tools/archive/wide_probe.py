#!/usr/bin/env python3
"""Host probes of tree shapes (tests/emu EMU_WIDE_PROBE): nodes / triangle tests per closest-hit ray for 4/6/8-wide trees and multi-triangle leaves.
    python tools/wide_probe.py c3 c2   (profiles/r03_tcp_bound.md section 4)"""
import sys, os, ctypes as C
ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ["EMU_WIDE_PROBE"]="1"; os.environ["EMU_MORTON"]="1"; os.environ["EMU_PLOC"]="8"; os.environ["EMU_SAH_COLLAPSE"]="1"
import numpy as np, emu_lib
from platinum_amd import scenes
from platinum_amd.renderer import make_params
which = sys.argv[1:] or ["c3","c2"]
for name in which:
    sc = scenes.CONFIGS[name][0](); B = scenes.CONFIGS[name][4]
    w,h=(240,135) if name!="c5" else (192,108)
    e = emu_lib.EmuScene(sc, make_params(w,h,1,B))
    rad = np.zeros((h,w,4),np.float32)
    e.L.emu_debug_sample(e.h, 0, rad.ctypes.data_as(C.c_void_p), None)
    out=(C.c_double*26)(); e.L.emu_get_wide(out)
    cnt=(C.c_ulonglong*4)(); e.L.emu_get_counts(cnt)
    print(name, "rays %d mismatches %d | product 4-wide nodes/ray %.2f tris/ray %.2f" % (out[24], out[25], cnt[0]/cnt[2], cnt[1]/cnt[2]))
    i=0
    for N in (4,6,8):
        for rule,rn in enumerate(("sorted","nearest-first","octant")):
            if out[i] > 0: print("   N=%d %-14s nodes/ray %6.2f  tris/ray %5.2f" % (N, rn, out[i], out[i+1]))
            i+=2
    for M in (2,3,4):
        print('   N=4 leaves<=%d sorted   nodes/ray %6.2f  tris/ray %5.2f' % (M, out[i], out[i+1])); i+=2
