#!/usr/bin/env python3
"""usage (GPU box): python tools/restart_latency.py [c3|c2] — wall time of pt_start_render on an already-started renderer (the reference's
startRender + first render() rebuild, renderer_pt.cpp:199-217 / :72-110), library side only: the snapshot is built once outside the timed
region.  $PTAMD_START_PHASES=1 prints the library's own phase times on stderr."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from platinum_amd import Renderer, abi, scenes
from platinum_amd.renderer import make_params

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
sc = scenes.field_scene(32) if wl == "c3" else scenes.cornell_sphere_scene()
r = Renderer(device=0)
snap = sc.snapshot()
p = make_params(1920, 1080, 256, 8)
lib = r._lib
times = []
for i in range(6):
    t0 = time.perf_counter()
    abi.check(lib, lib.pt_start_render(r._h, C.byref(snap.struct), C.byref(p)))
    times.append((time.perf_counter() - t0) * 1e3)
    r._params = p; r.size = (1920, 1080)
    r.render(1); r.wait()
st = r.stats()
print(json.dumps({"workload": wl, "first_start_ms": round(times[0], 2), "restart_ms": [round(t, 2) for t in times[1:]], "bvh_build_ms": round(st.bvh_build_ms, 3),
                  "upload_ms": round(st.upload_ms, 3), "triangles": st.triangles}))
r.close()
