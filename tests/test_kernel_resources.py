"""The occupancy the hot kernels are designed for is a property of the BUILD: registers, scratch and LDS per kernel as LLVM's kernel-resource-usage
remarks report them for the Makefile's flags (tools/kernel_resources.sh; no GPU needed).  A change that pushes a kernel over its budget costs 3-40 %
on the MI355X (profiles/HISTORY.md) without failing any parity test — so the budgets are asserted here."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel (as tools/kernel_resources.sh prints it) -> (max VGPRs, max scratch bytes, min waves per SIMD, max LDS bytes per block)
BUDGET = {
    "void k_trace_closest<false, false, true>": (80, 0, 6, 23 * 1024),    # 6 blocks of 256 threads per CU
    "void k_trace_shadow<false, false, true>": (72, 0, 7, 23 * 1024),     # 7 blocks per CU
    "k_shade": (128, 0, 4, 32 * 1024),                                     # 4 waves per SIMD
    "k_raygen": (64, 0, 8, 0),
    "k_accumulate": (64, 0, 4, 40 * 1024),
}


def test_hot_kernels_stay_inside_their_register_scratch_and_lds_budgets():
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "kernel_resources.sh")], capture_output=True, text=True, timeout=600).stdout
    seen = {}
    for line in out.splitlines():
        m = re.match(r"(.+?) VGPRs (\d+) scratch (\d+) spill (\d+) occ (\d+) LDS (\d+)", line.strip())
        if m:
            seen[m.group(1)] = tuple(int(x) for x in m.groups()[1:])
    for name, (vgpr, scratch, occ, lds) in BUDGET.items():
        assert name in seen, (name, sorted(seen))
        v, s, _spill, o, l = seen[name]
        assert v <= vgpr and s <= scratch and o >= occ and l <= lds, "%s: %d VGPRs, %d B scratch, occupancy %d, %d B LDS (budget %s)" % (name, v, s, o, l, BUDGET[name])
