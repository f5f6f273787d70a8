#!/usr/bin/env python3
"""What does lobe divergence cost k_shade?  The C3 field with its per-instance material mix against the same geometry with one
material everywhere (diffuse / rough metal / rough glass): device time per shaded hit.  (GPU box: python tools/shade_divergence_probe.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from platinum_amd import Renderer, abi, scenes

def run(name, sc):
    r = Renderer(device=0)
    W, H, S, B = 1920, 1080, 64, 8
    r.startRender(sc, (W, H), 2 * S, max_bounces=B, samples_in_flight=S, nonfinite_policy=abi.NONFINITE_ZERO)
    r.render(S); r.wait()
    r.startRender(sc, (W, H), 2 * S, max_bounces=B, samples_in_flight=S, nonfinite_policy=abi.NONFINITE_ZERO)
    r.setProfiling(True)
    r.render(0); r.wait()
    st = r.stats()
    print("%-14s shade %.2f ms, %.1f M hits, %.3f ns/hit | closest %.3f ns/ray | shadow %.3f ns/ray" % (
        name, st.ms_shade, st.shaded_hits / 1e6, st.ms_shade * 1e6 / st.shaded_hits, st.ms_closest * 1e6 / st.closest_rays,
        st.ms_shadow * 1e6 / max(1, st.shadow_rays)))
    r.close()

def uniform(mat):
    sc = scenes.field_scene()
    for n in sc.nodes[1:]:   # the spheres; the Cornell shell keeps its walls and its light
        n.materials = [mat for _ in n.materials]
    return sc

run("mixed (C3)", scenes.field_scene())
run("all diffuse", uniform(scenes.Material(base_color=(0.7, 0.7, 0.7, 1.0), roughness=1.0)))
run("all metal", uniform(scenes.Material(base_color=(0.7, 0.6, 0.5, 1.0), roughness=0.4, metallic=1.0)))
run("all glass", uniform(scenes.Material(base_color=(1, 1, 1, 1), roughness=0.2, ior=1.5, transmission=1.0)))
