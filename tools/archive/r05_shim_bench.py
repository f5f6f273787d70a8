#!/usr/bin/env python3
"""GPU box: ONE throughput figure on the runtime the C++ host would use.  Writes the C3 scene as files (tools/export_gltf.py:
.glb -> pt_scene_import_gltf -> scene.json + _data.bin, the reference's own format), then times the SAME scene file
  (a) from tests/cpp/shim_bench — a C++ process without torch: the system HIP runtime (/opt/rocm), render() one sample per call, and
  (b) from this Python process through platinum_amd (torch's bundled HIP runtime), the same call pattern,
and prints both JSON lines.  (The imported scene is C3 de-instanced into 1 025 meshes with regenerated tangents: the same triangles and
materials as bench.py's C3, not its bits.)"""
import json, os, subprocess, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
steps, warmup = int(os.environ.get("STEPS", "8")), 2
out_dir = os.environ.get("TMPDIR", "/tmp") + "/ptamd_shim_bench"
os.makedirs(out_dir, exist_ok=True)
import export_gltf
from platinum_amd import scenes, scene_io, abi
sc = scenes.field_scene(32)
glb, js = os.path.join(out_dir, "c3.glb"), os.path.join(out_dir, "c3.json")
export_gltf.export_scene(sc, glb)
cam = sc.camera_transform if hasattr(sc, "camera_transform") else None
f = export_gltf.load_exported(glb, None, (0, 14, 31.5), (0, 2, 0), 28.0)
f.save(js)
print("scene file:", js, "counts:", {n: getattr(f.counts(), n) for n in ("meshes", "triangles", "cameras")}, flush=True)
W, H, B = 1920, 1080, 8
import test_cpp_shim
exe = test_cpp_shim._build(os.path.join(ROOT, "tests", "cpp", "shim_bench.cpp"), os.path.join(ROOT, "tests", "_build", "shim_bench"))
env = dict(os.environ, PTAMD_LUT_PATH=os.path.join(ROOT, "platinum_amd", "data", "ggx_luts.bin"))
p = subprocess.run([exe, js, str(W), str(H), str(B), str(steps), str(warmup)], env=env, capture_output=True, text=True, timeout=900)
print("C++ host rc", p.returncode, p.stderr[-500:], flush=True)
cpp = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else None
print(json.dumps(cpp), flush=True)
# (b) the same file from Python (torch's runtime; import torch first so that it is the one mapped)
import torch  # noqa: F401
from platinum_amd import Renderer
g = scene_io.SceneFile.load(js)
r = Renderer(device=0)
r.startRender(g, (W, H), 1 << 16, max_bounces=B)   # (nothing is rendered: the plan is made at start)
S = int(r.stats().samples_in_flight)
def run(n):
    r.startRender(g, (W, H), n * S, max_bounces=B, samples_in_flight=S, nonfinite_policy=abi.NONFINITE_ZERO)
    r.wait(); t0 = time.perf_counter()
    for _ in range(n * S):
        r.render(1)
    r.wait()
    return time.perf_counter() - t0
run(warmup)
sec = run(steps)
st = r.stats(); ri = abi.runtime_info()
py = {"host": "Python (platinum_amd, torch in the process), render(1) per call", "value": round(W * H * steps * S * B / sec / 1e6, 2), "unit": "Msamples/s",
      "steps": steps, "spp_per_step": S, "ms_per_step": round(sec / steps * 1e3, 3), "closest_rays": int(st.closest_rays), "shadow_rays": int(st.shadow_rays),
      "shaded_hits": int(st.shaded_hits), "triangles": int(st.triangles), "hip_runtime_path": ri["hip_runtime_path"], "hip_runtime_version": ri["hip_runtime_version"]}
print(json.dumps(py), flush=True)
if cpp:
    print("C++ / Python = %.4f" % (cpp["value"] / py["value"]))
