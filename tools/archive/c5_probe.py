"""GPU probe: where does the atrium's shadow-ray time go?  (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from platinum_amd import Renderer, abi, scenes

def run(tag, sc, w=1920, h=1080, spp=16, b=12):
    r = Renderer(device=0)
    r.setProfiling(True)
    r.startRender(sc, (w, h), spp, max_bounces=b, nonfinite_policy=abi.NONFINITE_ZERO)
    r.render(0); r.wait()
    st = r.stats()
    print(tag.ljust(26), "closest %.1f ms (%.2f Grays/s)  shade %.1f  shadow %.1f ms (%.2f Grays/s)  rays c/s %d %d" % (
        st.ms_closest, st.closest_rays / st.ms_closest / 1e6, st.ms_shade, st.ms_shadow, st.shadow_rays / max(st.ms_shadow, 1e-9) / 1e6,
        st.closest_rays, st.shadow_rays), flush=True)
    r.close()

base = scenes.atrium_scene(env_size=(1024, 512))
run("atrium", base)
s = scenes.atrium_scene(env_size=(1024, 512)); s.env_texture = -1
run("no env", s)
s = scenes.atrium_scene(env_size=(1024, 512))
s.nodes = [n for n in s.nodes if n.materials[0].name != "banner"]
run("no banners (no alpha)", s)
s = scenes.atrium_scene(env_size=(1024, 512))
for n in s.nodes:
    if n.materials[0].name == "banner":
        n.materials[0].base_texture_has_alpha = False
run("banners opaque", s)
s = scenes.atrium_scene(env_size=(1024, 512))
s.nodes = [n for n in s.nodes if n.materials[0].name != "lamp"]
run("no lamps (env only)", s)
s = scenes.atrium_scene(env_size=(1024, 512))
s.nodes = [n for n in s.nodes if "column" not in n.materials[0].name and n.materials[0].name not in ("marble", "bronze", "painted")]
run("no columns", s)
