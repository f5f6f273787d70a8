#!/usr/bin/env python3
"""What VERDICT r3 item 1 asked for: the GPU runtime objects mapped into the process, and who sees the GPU, in each load order.

    python tools/runtime_order_probe.py raw-lib-first     # ctypes.CDLL(libptamd.so) + pt_create, THEN torch: the r3 failure, reproduced
    python tools/runtime_order_probe.py raw-torch-first   # torch on the GPU, then ctypes.CDLL(libptamd.so) + pt_create
    python tools/runtime_order_probe.py abi-lib-first     # the same through platinum_amd.abi.load_library() (the fix)
Each prints the libamdhip64 / libhsa-runtime64 / librccl objects of /proc/self/maps after every step (profiles/r04_runtime_identity.md)."""
import ctypes as C
import os
import re
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def maps(tag):
    libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if re.search(r"libamdhip64|libhsa-runtime|librccl", l)})
    print("[%s]" % tag)
    for x in libs:
        print("    " + x)


def create(lib, abi):
    lib.pt_last_error.restype = C.c_char_p
    info = abi.CreateInfo(abi_version=abi.PT_ABI_VERSION, device_ordinal=0, lut_path=abi.LUT_PATH.encode())
    h = C.c_void_p()
    rc = lib.pt_create(C.byref(info), C.byref(h))
    print("    pt_create ->", rc, (lib.pt_last_error() or b"").decode()[:300])
    return h if rc == 0 else None


def torch_gpu():
    import torch
    try:
        print("    torch.zeros on cuda:0 ->", float(torch.ones(4, device="cuda:0").sum().item()))
    except Exception as e:  # noqa: BLE001
        print("    torch on the GPU FAILED:", type(e).__name__, str(e).splitlines()[0][:200])


mode = sys.argv[1]
from platinum_amd import abi  # noqa: E402  (imports ctypes structures only; loads nothing)
if mode == "raw-lib-first":
    lib = C.CDLL(abi.LIB_PATH)
    maps("after CDLL(libptamd.so)")
    h = create(lib, abi)
    import torch  # noqa: F401
    maps("after import torch")
    torch_gpu()
    print("    a second pt_create in the same process:")
    create(lib, abi)
elif mode == "raw-torch-first":
    import torch  # noqa: F401
    maps("after import torch")
    torch_gpu()
    lib = C.CDLL(abi.LIB_PATH)
    maps("after CDLL(libptamd.so)")
    create(lib, abi)
elif mode == "abi-lib-first":
    lib = abi.load_library()
    maps("after abi.load_library()")
    print("    settled:", abi.runtime_info()["settled"])
    create(lib, abi)
    import torch  # noqa: F401
    maps("after import torch")
    torch_gpu()
    create(lib, abi)
    print("    pt_rccl_selftest ->", lib.pt_rccl_selftest(0), abi.runtime_info()["rccl_path"])
