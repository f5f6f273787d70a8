#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/profile_round.sh for one workload into profiles/:

  <tag>_<wl>_kernel_stats.csv   copy of rocprofv3 --kernel-trace --stats of the default bench command
  <tag>_<wl>_summary.md         per kernel: calls / avg ms (trace pass); FETCH_SIZE, WRITE_SIZE per launch and per work item,
                                VALU wave-instructions per work item (--pmc passes of `bench.py --pmc-pass`: full-size batches only)
  <round>_pmc_<wl>.json         what bench.py reads: per-work-item HBM bytes and VALU instructions + the library's sha
                                (<round> = the tag up to its first letter suffix: r03b -> r03)

Units and corrections as MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced streaming reads.  The factor is CALIBRATED in every pass on a kernel whose pattern never changes — a
1 GiB torch elementwise pass bench.py --pmc-pass runs (r5) — and a value outside [1.9, 2.1] stops this script; k_accumulate ((S+1) * 16 B read,
16 B written per pixel) and k_raygen (64 B written per path) are reported beside it as second opinions.
The traversal kernels read 64-B nodes / triangle slots as 16-B-per-lane gathers — a pattern the guide leaves uncalibrated;
tools/archive/calib_gather.hip calibrated it (profiles/r02_calib_gather.md): FETCH_SIZE is exact for it, so those kernels get x1."""
import csv, glob, hashlib, json, os, shutil, sys
from collections import defaultdict

prof_dir, wl, tag = sys.argv[1], sys.argv[2], sys.argv[3]  # e.g. gpurun_out/prof c3 r02
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(root, "profiles")
os.makedirs(out_dir, exist_ok=True)

KEYS = ("k_shade", "k_raygen", "k_accumulate", "k_fold_counters", "k_chunk_tables", "k_refit", "k_karras", "k_emit", "k_morton", "k_flatten",
        "k_bounds", "k_hit_records", "k_weighted_add", "k_postprocess", "k_gmon")


def short(name):
    """k_trace_closest<COUNT, TWO, W6>: the render instantiation of the one-BVH kernel (4- or 6-wide nodes: whichever the build produced) is
    the bare name; the instrumented and the two-level instantiations are tagged."""
    import re
    m = re.search(r"(k_trace_closest|k_trace_shadow)<(?:\(bool\))?(\w+), (?:\(bool\))?(\w+)(?:, (?:\(bool\))?(\w+))?>", name)
    if m:
        count, two = m.group(2) in ("true", "1"), m.group(3) in ("true", "1")
        return m.group(1) + ("[count]" if count else "") + ("[two-level]" if two else "")
    for k in KEYS:
        if k in name:
            return k
    if "rocprim" in name:
        return "rocprim_radix_sort"
    return name[:40]


def newest(pattern):
    f = glob.glob(pattern, recursive=True)
    return max(f, key=os.path.getmtime) if f else None


stats = newest(os.path.join(prof_dir, f"{wl}_trace", "**", "*_kernel_stats.csv"))
summary = {}
if stats:
    shutil.copy(stats, os.path.join(out_dir, f"{tag}_{wl}_kernel_stats.csv"))
    for r in csv.DictReader(open(stats)):
        d = summary.setdefault(short(r["Name"]), {"calls": 0, "total_ms": 0.0})
        d["calls"] += int(r["Calls"]); d["total_ms"] += float(r["TotalDurationNs"]) / 1e6


RAW_ROWS = {}   # kind -> [(full kernel name, counter, value)] of that pass (the calibration copy is found by its full name)


def counters(kind):
    """{kernel: {counter: [dispatches, sum]}} of one --pmc pass, and that pass's bench JSON (item counts)."""
    f = newest(os.path.join(prof_dir, f"{wl}_{kind}", "**", "*_counter_collection.csv"))
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    RAW_ROWS[kind] = []
    if f:
        for r in csv.DictReader(open(f)):
            a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
            RAW_ROWS[kind].append((r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])))
    try:
        j = json.loads([l for l in open(os.path.join(prof_dir, f"{wl}_{kind}.json")) if l.startswith("{")][-1])
    except Exception:
        j = None
    return acc, j


fetch, jf = counters("fetch")
write, jw = counters("write")
sq, js = counters("sq")
tc, jt = counters("tc")


def items(j):
    """work items of a --pmc-pass run, per kernel"""
    if not j:
        return {}
    e, c = j["extra"], j["config"]
    npix, spp = c["width"] * c["height"], c["spp_per_gpu"]
    return {"k_trace_closest": e["closest_rays"], "k_trace_shadow": e["shadow_rays"], "k_trace_closest[two-level]": e["closest_rays"],
            "k_trace_shadow[two-level]": e["shadow_rays"], "k_shade": e["shaded_hits"], "k_raygen": e["paths"], "k_accumulate": npix * spp, "_npix": npix, "_spp_per_step": c["spp_per_step"],
            "_steps": j["steps"]}


itf, itw, its, itt = items(jf), items(jw), items(js), items(jt)

# ---- calibration on known byte counts ----
# (r5) The streaming factor comes from a kernel whose access pattern does not change between rounds: the 1 GiB `torch.add(src, 1, out=dst)` bench.py runs at
# the end of every --pmc pass (torch's vectorised copy, 16 B per lane).  r2-r4 derived it from k_accumulate, whose reads r4 rewrote: with 46
# samples per pixel (C5) a pixel's samples are not whole 128-byte lines and the "factor" came out as 1.15.  k_accumulate is still reported, as
# a second opinion where its reads ARE whole lines (samples per step a multiple of 8).  MI355X_MICROARCH.md section HBM prescribes exactly 2 for
# wide coalesced streaming reads: a measured factor outside [1.9, 2.1] is an error, not a number to use.
calib = {}


def copy_factor(kind, counter, j):
    c = (j or {}).get("calibration_copy")
    if not c:
        return None, None
    vals = [v for (name, cn, v) in RAW_ROWS.get(kind, []) if cn == counter and c["kernel_name_contains"] in name]
    # the copies are the dispatches of that name that moved the most bytes (the fill kernels of the same family read nothing)
    vals = sorted(vals, reverse=True)[: c["copies"]]
    if len(vals) < c["copies"] or min(vals) <= 0:
        return None, None
    raw = sum(vals) * 1024.0
    return c["bytes_per_copy"] * c["copies"] / raw, raw


ff_copy, raw_copy = copy_factor("fetch", "FETCH_SIZE", jf)
wf_copy, _ = copy_factor("write", "WRITE_SIZE", jw)
if ff_copy is not None:
    calib["fetch_streaming_factor"] = ff_copy
    calib["fetch_streaming_factor_source"] = "1 GiB torch elementwise pass x %d (bench.py --pmc-pass), FETCH_SIZE raw %.0f bytes" % (jf["calibration_copy"]["copies"], raw_copy)
if wf_copy is not None:
    calib["write_factor_copy"] = wf_copy
if "k_accumulate" in fetch and itf:
    S = itf["_spp_per_step"]
    expect = itf["_npix"] * (S + 1) * 16.0 * itf["_steps"]
    raw = fetch["k_accumulate"]["FETCH_SIZE"][1] * 1024.0
    calib["k_accumulate_fetch_factor"] = expect / raw if raw else None
    calib["k_accumulate_whole_lines"] = bool(S % 8 == 0)
    calib["k_accumulate_read_bytes_expected"] = expect
    calib["k_accumulate_FETCH_SIZE_bytes_raw"] = raw
    if "fetch_streaming_factor" not in calib and S % 8 == 0:
        calib["fetch_streaming_factor"] = calib["k_accumulate_fetch_factor"]
        calib["fetch_streaming_factor_source"] = "k_accumulate (no calibration copy in this pass)"
if "k_accumulate" in write and itw:
    expect = itw["_npix"] * 16.0 * itw["_steps"]
    raw = write["k_accumulate"]["WRITE_SIZE"][1] * 1024.0
    calib["write_factor_k_accumulate"] = expect / raw if raw else None
if "k_raygen" in write and itw:
    expect = itw["k_raygen"] * 64.0   # rayO 16 + rayD 16 + att 16 + Lbuf 16 per path (kernels.hip k_raygen; the separate pid word went in r2)
    raw = write["k_raygen"]["WRITE_SIZE"][1] * 1024.0
    calib["write_factor_k_raygen"] = expect / raw if raw else None
    calib["k_raygen_write_bytes_expected"] = expect
    calib["k_raygen_WRITE_SIZE_bytes_raw"] = raw
ff = calib.get("fetch_streaming_factor")
if fetch and ff is None:
    sys.exit("summarize_prof.py: no streaming calibration in the fetch pass (no calibration copy, and k_accumulate's reads are not whole lines at %s samples per step): re-run tools/profile_round.sh" % (itf.get("_spp_per_step") if itf else "?"))
if ff is not None and not (1.9 <= ff <= 2.1):
    sys.exit("summarize_prof.py: streaming-read calibration %.3f is outside [1.9, 2.1] (MI355X_MICROARCH.md section HBM: exactly 2 for wide coalesced reads) - "
             "the counter pass is not usable as it stands (%s)" % (ff, calib.get("fetch_streaming_factor_source")))
ff = ff or 2.0
# Random 64-byte gathers (what the traversal kernels fetch) are reported exactly: profiles/r02_calib_gather.md
GATHER_FACTOR = {"k_trace_closest": 1.0, "k_trace_shadow": 1.0, "k_trace_closest[two-level]": 1.0, "k_trace_shadow[two-level]": 1.0}
# k_shade reads BOTH kinds: per hit it streams its queue entry (hit 16 B + rayO 16 + rayD 16 + att 16 + the 4-byte
# class word of the scan = 68 B, coalesced dwordx4 / dword runs: reported at 1/ff) and gathers ShadeRec, vertices, material, LUT
# texels (random records: reported exactly).  true fetch = stream + (raw - stream / ff), bounded below by raw (everything x1)
# and above by ff * raw (everything xff).
STREAM_IN_BYTES = {"k_shade": 68.0}

lines = [f"# rocprofv3 summary — bench.py --workload {wl} (MI355X, {tag})", "",
         "Durations: `rocprofv3 --kernel-trace --stats -- python3 bench.py --workload %s --no-cpu-baseline` (both of bench.py's passes)." % wl,
         "Counters: separate `--pmc` passes of `bench.py --workload %s --pmc-pass --steps 2` (full batches of the library's own size only: 128 spp at 1080p)." % wl, "",
         "Calibration on known byte counts: " + json.dumps({k: (round(v, 4) if isinstance(v, float) and v < 100 else v) for k, v in calib.items()}), "",
         "| kernel | calls | total ms | avg ms | FETCH raw MiB/launch | FETCH corrected MiB/launch (x%.2f streams, x1 gathers, k_shade split: r02_calib_gather.md) | WRITE MiB/launch | HBM B/item (corr.) | VALU wave-insts/item | items |" % ff,
         "|---|---|---|---|---|---|---|---|---|---|"]
pmc = {"source": f"rocprofv3 --pmc passes of `bench.py --workload {wl} --pmc-pass --steps 2` ({tag}); tools/summarize_prof.py",
       "calibration": calib, "kernels": {}}
lib = os.environ.get("PTAMD_LIB", os.path.join(root, "platinum_amd", "csrc", "libptamd.so"))
if os.path.exists(lib):
    pmc["library_sha16"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
names = sorted(set(summary) | set(fetch) | set(write) | set(sq) | set(tc), key=lambda k: -summary.get(k, {"total_ms": 0})["total_ms"])
for k in names:
    d = summary.get(k, {"calls": 0, "total_ms": 0.0})
    fn, fv = fetch[k]["FETCH_SIZE"] if k in fetch else (0, 0.0)
    wn, wv = write[k]["WRITE_SIZE"] if k in write else (0, 0.0)
    vn, vv = sq[k]["SQ_INSTS_VALU"] if k in sq else (0, 0.0)
    fl = fv / fn / 1024.0 if fn else float("nan")
    wr = wv / wn / 1024.0 if wn else float("nan")
    ni_f, ni_w, ni_s = itf.get(k), itw.get(k), its.get(k)
    kf = GATHER_FACTOR.get(k, ff)
    per_item = (kf * fv * 1024.0 / ni_f + wv * 1024.0 / ni_w) if (ni_f and ni_w and fn and wn) else None
    split = None
    if per_item is not None and k in STREAM_IN_BYTES:
        raw_pi, wr_pi, st = fv * 1024.0 / ni_f, wv * 1024.0 / ni_w, STREAM_IN_BYTES[k]
        fetch_true = st + max(0.0, raw_pi - st / ff)
        split = {"stream_in_bytes_per_item": st, "gather_bytes_per_item": max(0.0, raw_pi - st / ff),
                 "hbm_bytes_per_item_all_x1": raw_pi + wr_pi, "hbm_bytes_per_item_all_streams": ff * raw_pi + wr_pi}
        per_item = fetch_true + wr_pi
        kf = fetch_true / raw_pi if raw_pi else kf
    valu = vv / ni_s if (ni_s and vn) else None
    avg = d["total_ms"] / d["calls"] if d["calls"] else float("nan")
    lines.append(f"| {k} | {d['calls']} | {d['total_ms']:.3f} | {avg:.4f} | {fl:.1f} | {kf*fl:.1f} | {wr:.1f} | "
                 f"{'%.1f' % per_item if per_item is not None else '-'} | {'%.1f' % valu if valu is not None else '-'} | {ni_s or ni_f or '-'} |")
    if per_item is not None or valu is not None:
        e = {"hbm_bytes_per_item": per_item, "fetch_bytes_raw_per_item": fv * 1024.0 / ni_f if (ni_f and fn) else None,
             "write_bytes_per_item": wv * 1024.0 / ni_w if (ni_w and wn) else None, "fetch_correction": kf, "valu_insts_per_item": valu}
        if split:
            e["fetch_split"] = split
        if d["calls"] and k in summary:
            e["avg_launch_ms_trace_pass"] = d["total_ms"] / d["calls"]
        if ni_f and fn:
            e["fetch_bytes_per_launch_corrected"] = kf * fv * 1024.0 / fn
            e["write_bytes_per_launch"] = wv * 1024.0 / wn if wn else None
            e["items_per_launch"] = ni_f / fn
        if k in sq:
            for c in ("SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY"):
                if c in sq[k] and ni_s:
                    e[c.lower() + "_per_item"] = sq[k][c][1] / ni_s
        if k in tc and itt.get(k):
            t = tc[k]
            e["l2_read_requests_per_item"] = t["TCP_TCC_READ_REQ_sum"][1] / itt[k] if "TCP_TCC_READ_REQ_sum" in t else None
            e["l1_accesses_per_item"] = t["TCP_TOTAL_CACHE_ACCESSES_sum"][1] / itt[k] if "TCP_TOTAL_CACHE_ACCESSES_sum" in t else None
            if "TCC_HIT_sum" in t and "TCC_MISS_sum" in t and (t["TCC_HIT_sum"][1] + t["TCC_MISS_sum"][1]) > 0:
                e["l2_hit_rate"] = t["TCC_HIT_sum"][1] / (t["TCC_HIT_sum"][1] + t["TCC_MISS_sum"][1])
        pmc["kernels"][k] = e
open(os.path.join(out_dir, f"{tag}_{wl}_summary.md"), "w").write("\n".join(lines) + "\n")
if pmc["kernels"]:
    import re
    rnd = re.match(r"(r\d+)", tag).group(1) if re.match(r"(r\d+)", tag) else tag
    json.dump(pmc, open(os.path.join(out_dir, f"{rnd}_pmc_{wl}.json"), "w"), indent=1)
print("\n".join(lines))
