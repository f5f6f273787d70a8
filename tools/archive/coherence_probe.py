"""GPU probe: closest-hit cost per ray for primary (coherent) vs secondary rays."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from platinum_amd import Renderer, abi, scenes
for name, factory in (("c2", scenes.cornell_sphere_scene), ("c3", scenes.field_scene)):
    sc = factory()
    prev = None
    for b in (1, 2, 3):
        r = Renderer(device=0); r.setProfiling(True)
        r.startRender(sc, (1920, 1080), 64, max_bounces=b, nonfinite_policy=abi.NONFINITE_ZERO)
        r.render(0); r.wait(); st = r.stats()
        cur = (st.ms_closest, st.closest_rays, st.ms_shadow, st.shadow_rays)
        if prev is None:
            print(name, "bounce 0: closest %.3f ns/ray (%.1f Grays/s), shadow %.3f ns/ray" % (1e6 * cur[0] / cur[1], cur[1] / cur[0] / 1e6, 1e6 * cur[2] / max(cur[3], 1)), flush=True)
        else:
            dc, dr = cur[0] - prev[0], cur[1] - prev[1]; ds, dsr = cur[2] - prev[2], cur[3] - prev[3]
            print(name, "bounce %d: closest %.3f ns/ray (%.1f Grays/s), shadow %.3f ns/ray" % (b - 1, 1e6 * dc / dr, dr / dc / 1e6, 1e6 * ds / max(dsr, 1)), flush=True)
        prev = cur
        r.close()
