"""One GPU runtime per process.

PyTorch's ROCm wheel bundles private copies of libamdhip64 / libhsa-runtime64 / librccl whose sonames equal /opt/rocm's.  Depending on
what is loaded first the process ends up with one runtime (torch first: libptamd.so binds to torch's copy by soname) or two (libptamd.so
first: /opt/rocm's copy, then torch's beside it) - and of two, only the first to initialise sees the GPU.  The product settles this:
`platinum_amd.abi.load_library()` maps torch's bundled runtime first when the interpreter has one, `pt_create` refuses a process that holds
two, `Rccl::load` binds the librccl that is already mapped.  Each case runs in a fresh interpreter (the state under test is the process's
link map).  The reference has no counterpart (Metal is a system framework)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
order, touch_gpu = sys.argv[1], sys.argv[2] == "gpu"
out = {}
from platinum_amd import abi
import ctypes as C
def create():
    lib = abi.load_library()
    info = abi.CreateInfo(abi_version=abi.PT_ABI_VERSION, device_ordinal=0, lut_path=abi.LUT_PATH.encode())
    h = C.c_void_p()
    rc = lib.pt_create(C.byref(info), C.byref(h))
    msg = lib.pt_last_error().decode() if rc else ""
    if rc == 0:
        lib.pt_destroy(h)
    return rc, msg
if order == "lib-first":
    abi.load_library()
    if touch_gpu:
        out["create_before_torch"] = create()
    import torch
elif order == "torch-first":
    import torch
    if touch_gpu:
        torch.zeros(4, device="cuda:0")
    abi.load_library()
elif order == "raw-lib-then-torch":     # what a host that does not go through abi.load_library() gets: /opt/rocm's runtime, then torch's too
    C.CDLL(abi.LIB_PATH)
    import torch
    os.environ["PTAMD_HIP_RUNTIME"] = "system"
    try:
        abi.load_library()
        out["load_error"] = ""
    except abi.PtamdError as e:
        out["load_error"] = str(e)
    lib = C.CDLL(abi.LIB_PATH)
    lib.pt_last_error.restype = C.c_char_p
    info = abi.CreateInfo(abi_version=abi.PT_ABI_VERSION, device_ordinal=0, lut_path=abi.LUT_PATH.encode())
    h = C.c_void_p()
    out["create_rc"] = lib.pt_create(C.byref(info), C.byref(h))
    out["create_msg"] = lib.pt_last_error().decode()
    print(json.dumps(out)); sys.exit(0)
ri = abi.runtime_info()
out.update({k: ri[k] for k in ("hip_runtime_path", "hsa_runtime_path", "hip_runtimes_mapped", "hsa_runtimes_mapped", "settled", "all_mapped")})
out["torch_hip"] = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
lib = abi.load_library()
out["rccl_probe"] = lib.pt_rccl_probe()
out["rccl_path"] = abi.runtime_info()["rccl_path"]
out["torch_rccl"] = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
if touch_gpu:
    out["torch_sum"] = float(torch.arange(8, device="cuda:0").sum().item())     # the step that failed in r3: torch AFTER the library
    out["create_after_torch"] = create()
    out["rccl_selftest"] = lib.pt_rccl_selftest(0)
print(json.dumps(out))
"""


def _run(order, mode):
    env = dict(os.environ, PTAMD_QUIET="1")
    env.pop("PTAMD_HIP_RUNTIME", None)
    p = subprocess.run([sys.executable, "-c", PROBE % {"root": ROOT}, order, mode], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    return json.loads(p.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("order", ["lib-first", "torch-first"])
def test_one_hip_runtime_whichever_of_library_and_torch_is_loaded_first(order):
    o = _run(order, "cpu")
    assert o["hip_runtimes_mapped"] == 1 and o["hsa_runtimes_mapped"] == 1, o["all_mapped"]
    assert os.path.samefile(o["hip_runtime_path"], o["torch_hip"])          # the library runs on the runtime torch runs on
    assert o["rccl_probe"] == 0 and os.path.samefile(o["rccl_path"], o["torch_rccl"])   # and binds the RCCL torch.distributed uses
    assert ("preloaded" in o["settled"]) == (order == "lib-first")


def test_a_process_with_two_hip_runtimes_is_refused_with_the_paths_named():
    o = _run("raw-lib-then-torch", "cpu")
    assert "two GPU runtimes" in o["load_error"] and "/opt/rocm" in o["load_error"] and "torch/lib" in o["load_error"]
    assert o["create_rc"] == -8 and "two GPU runtimes" in o["create_msg"]      # PT_ERR_RUNTIME_CONFLICT, before any GPU call


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["lib-first", "torch-first"])
def test_library_and_torch_share_the_gpu_in_either_load_order(order):
    """The r3 failure (`torch.zeros(device="cuda:0")` -> "No HIP GPUs are available" after libptamd.so had initialised its own runtime)
    as a test: renderer created, THEN torch imported and used, then another renderer and the one-rank RCCL self-test - one process."""
    o = _run(order, "gpu")
    assert o["hip_runtimes_mapped"] == 1, o["all_mapped"]
    assert o["torch_sum"] == 28.0
    assert o["create_after_torch"][0] == 0, o["create_after_torch"]
    if order == "lib-first":
        assert o["create_before_torch"][0] == 0, o["create_before_torch"]
    assert o["rccl_selftest"] == 0
