import sys, os, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from platinum_amd import Renderer, abi, scenes
from platinum_amd.scenes import Scene, Material, Transform, Camera, sphere

r = Renderer(device=0)

def render(sc, w, h, spp, B, integ, flags=abi.FLAG_MULTISCATTER_GGX):
    r.selectKernel(integ)
    r.startRender(sc, (w, h), spp, max_bounces=B, flags=flags, nonfinite_policy=abi.NONFINITE_ZERO)
    r.render(0)
    return r.readbackAccumulator().astype(np.float64)[..., :3]

out = {}
# 1. MIS vs SIMPLE on scenes lit by area lights only
for name, sc in [("cornell", scenes.cornell_scene()), ("cornell_sphere", scenes.cornell_sphere_scene()),
                 ("textured_noenv", scenes.textured_scene(env=False))]:
    a = render(sc, 96, 96, 4096, 8, abi.INTEGRATOR_MIS)
    b = render(sc, 96, 96, 4096, 8, abi.INTEGRATOR_SIMPLE)
    # block means (12x12 blocks) to tame per-pixel noise
    am = a.reshape(8, 12, 8, 12, 3).mean(axis=(1, 3)); bm = b.reshape(8, 12, 8, 12, 3).mean(axis=(1, 3))
    rel = np.abs(am - bm) / np.maximum(bm, 1e-3)
    out[name] = {"mean_mis": a.mean(), "mean_simple": b.mean(), "rel_mean": (a.mean() - b.mean()) / b.mean(), "max_block_rel": rel.max(),
                 "median_block_rel": float(np.median(rel))}

# 2. white furnace: a white sphere in a constant environment of radiance 1
def furnace_scene(roughness, metallic, transmission=0.0, clearcoat=0.0):
    sc = Scene(name="furnace")
    sph = sc.add_mesh(sphere(1.0, 48, 64))
    sc.add_instance(sph, Transform(), [Material(base_color=(1, 1, 1, 1), roughness=roughness, metallic=metallic, transmission=transmission, ior=1.5,
                                                clearcoat=clearcoat)])
    env = np.ones((8, 16, 4), dtype=np.float32)
    sc.env_texture = sc.add_texture(env, abi.TEX_RGBA32F)
    sc.set_camera(Camera.with_focal_length(50.0), Transform(translation=(0, 0, 6), target=(0, 0, 0), track=True))
    return sc

for rough in (0.0, 0.2, 0.5, 1.0):
    for metal in (0.0, 1.0):
        for integ, iname in ((abi.INTEGRATOR_SIMPLE, "simple"), (abi.INTEGRATOR_MIS, "mis")):
            a = render(furnace_scene(rough, metal), 64, 64, 1024, 16, integ)
            c = a[24:40, 24:40]  # the centre of the sphere
            out["furnace_r%.1f_m%d_%s" % (rough, metal, iname)] = {"centre": c.mean(), "all": a.mean(), "min": a.min(), "max": a.max()}
a = render(furnace_scene(0.3, 0.0, transmission=1.0), 64, 64, 1024, 16, abi.INTEGRATOR_SIMPLE)
out["furnace_glass_simple"] = {"centre": a[24:40, 24:40].mean(), "all": a.mean()}
a = render(furnace_scene(0.3, 0.0, clearcoat=1.0), 64, 64, 1024, 16, abi.INTEGRATOR_SIMPLE)
out["furnace_coat_simple"] = {"centre": a[24:40, 24:40].mean(), "all": a.mean()}
a = render(furnace_scene(0.5, 0.0), 64, 64, 1024, 16, abi.INTEGRATOR_SIMPLE, flags=0)
out["furnace_r0.5_m0_simple_no_multiscatter"] = {"centre": a[24:40, 24:40].mean(), "all": a.mean()}
r.close()
for k, v in out.items():
    print(k, {kk: round(float(vv), 5) for kk, vv in v.items()})
