"""GPU box, debug library (tools/build_variant.sh dbg -DPT_DEBUG_PID): the traversal of one camera ray under 46 and under 16 samples in flight."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PTAMD_LIB"] = os.path.join(ROOT, "platinum_amd", "csrc", "libptamd_dbg.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from platinum_amd import Renderer
import export_gltf
W, H, B = 3840, 2160, 12
sc = export_gltf.atrium_through_ingestion(tempfile.mkdtemp())
r = Renderer(device=0)
for (x, y, s) in ((2544, 532, 38), (3062, 895, 10)):
    for b in (0, 1):
        os.environ["PTAMD_DEBUG_RAY"] = "%d,%d,%d,%d" % (x, y, s, b)
        for sif in (0, 16):
            print("=== ray", x, y, s, "bounce", b, "sif", sif, flush=True)
            r.startRender(sc, (W, H), 46, max_bounces=B, samples_in_flight=sif)
            r.render(0); r.wait()
            sys.stdout.flush()
