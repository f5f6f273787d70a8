#!/usr/bin/env python3
"""bench.py — Msamples/s of the MI355X wavefront path tracer on BASELINE.json's configs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c1|c5] [--spp-per-step S] [--inproc]

One "step" = one pass of the hot path over one batch: S samples per pixel of the whole frame (raygen ->
{closest-hit, shade/BSDF/NEE, shadow} x bounces -> accumulate).  Default workload: C3 (BASELINE.json configs[2]: the
1.04 M-triangle instanced field, 1920x1080, 8 bounces — the largest single-GPU configuration and the one the
north-star targets are written on), S = the library's own batch size for the image (128 at 1080p), K = 4, W = 2.  `--workload c2` = configs[1].
Metric: Msamples/s = W*H*spp*B / t / 1e6, B = configured max bounces.

N > 1: one process per GPU.  Either launched by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`
(RANK / LOCAL_RANK / WORLD_SIZE in the environment) or, when started as a plain `python bench.py --gpus N`, this process
starts exactly that launcher as a CHILD process before it touches torch or the GPU, relays rank 0's JSON line and exits
with the child's code.  Rank g renders the same frame for sample indices [g*K*S, (g+1)*K*S) — independent-sample
sharding, no exchange while rendering — then ONE RCCL all-reduce (sum) of the float accumulator over xGMI inside the
timed region.  Weak scaling: per-GPU work is fixed.  `--inproc` instead drives all N devices from ONE process through the
library's own multi-device entry (pt_create with a device list; host thread + stream per device; ncclAllReduce inside).

The timed region runs with per-kernel timing OFF; kernel times come from a second, separately timed pass of the same
steps with HIP events around every launch (on the renderer's stream).  Rank 0 prints ONE JSON line with
  roofline          the kernel with the largest device time, against every ceiling that could bind it
  roofline_kernels  the same block for each hot kernel (k_shade, k_trace_closest, k_trace_shadow)
  cpu_baseline      (N = 1) the product's own stage functions compiled for the host (tests/emu) on all host cores, bounded sample of the
                    same workload (kind "port", port_of "same-kernels-host"); the scalar oracle's figure beside it under `oracle`
`roofline` is the contract's block for the dominant kernel: bound "hbm", achieved = HBM bytes the rocprofv3 counters saw per
launch (FETCH_SIZE + WRITE_SIZE, corrected as MI355X_MICROARCH.md §HBM prescribes; committed per-item figures of
profiles/<round>_pmc_<workload>.json x this run's items per launch) / this run's average launch time (HIP events), peak 8 TB/s,
frac = achieved / peak, traffic = those bytes per launch.  SURVEY §8(d)'s ALGORITHMIC bytes are served mostly by L1 / L2 /
Infinity Cache (they exceed the HBM peak), so they appear only as `algorithmic_GBs` / `algorithmic_over_counter`, never as a met
target.  What actually binds each kernel is reported under `secondary`: VALU issue (256 CU x 4 SIMD x 32 lanes x 2.4 GHz =
78.6 T lane-ops/s), the vector L1's tag-lookup rate (one per clock per CU: profiles/r03_calib_gather.md) and the rate at which
the L2s serve random requests.  Counter figures come from the committed --pmc passes of this same command
(tools/profile_round.sh), labelled with their source and the library they were taken on — they are not measured by this run.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
L2_PEAK_GBS = 34500.0       # aggregate over the 8 XCD L2s
VALU_PEAK_TLOPS = 78.6432   # 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz, in 1e12 lane-ops/s
# Measured request ceilings (profiles/r03_calib_gather.md; tools/calib_gather2.hip):
L2_HIT_REQ_PEAK_G = 250.0   # L2 read requests served per second when every request hits (2 MiB table, one dwordx4 per lane)
L2_MISS_REQ_PEAK_G = 55.0   # requests per second on the L2-miss path (Infinity Cache / HBM), any size up to 128 B, 2..16 in flight per lane
L1_TAG_PEAK_G = 833.0       # vector-L1 tag lookups per second: lane-private 64-byte records from an L2-resident table = 166.6 G records/s x 5
                            # lookups each (1.55 per clock per CU at 2.1 GHz); lane-private 128-byte records hold 0.98 per clock per CU
HBM_TARGET_FRAC = 0.40      # north_star: ">= 40 % of HBM peak on the traversal kernel"

WORKLOADS = {
    "c1": "C1 Cornell box 512x512",
    "c2": "C2 Cornell box + GGX dielectric sphere (6144 tris), 1920x1080",
    "c3": "C3 1.04M-triangle instanced sphere field, 1920x1080",
    "c3xl": "context (not a BASELINE config): the C3 field with 128x128 instances = 16.6M triangles, structure > the 256 MB Infinity Cache, 1920x1080",
    "c5": "C5 Sponza-class atrium (258k tris, JPEG/PNG textures, cut-outs, 4096x2048 EXR environment) imported from .glb + .exr, 3840x2160",
}


def algorithmic_bytes_closest(nodes_per_ray, tris_per_ray):
    # ray 32 B + hit 20 B + N_node * 64 B + N_tri * 36 B
    return 32.0 + 20.0 + 64.0 * nodes_per_ray + 36.0 * tris_per_ray


def algorithmic_bytes_shadow(nodes_per_ray, tris_per_ray):
    return 32.0 + 4.0 + 64.0 * nodes_per_ray + 36.0 * tris_per_ray


# SURVEY §8(d) per shaded hit: 368 B geometry/material gather + 64 B state in + 64 B state out + <= 112 B LUT taps + 48 B light
# record + 3 x 16 B shadow entry
ALGORITHMIC_BYTES_SHADE = 368.0 + 64.0 + 64.0 + 112.0 + 48.0 + 48.0


def available_cpus():
    """Host cores THIS process may use: the scheduler affinity mask, cut down to the cgroup's CPU quota when there is one (a one-GPU
    box of the pool reports 256 logical CPUs and grants 16: 256 threads on 16 cores measured the oracle at half its speed)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, int(round(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--spp-per-step", type=int, default=0,
                    help="samples per pixel of one step = one batch (samples in flight); 0 = what the library would choose for this image "
                         "(pt_plan_queues: 128 at 1920x1080 on an empty MI355X, 45 at 3840x2160)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU work budget of the cpu_baseline sample")
    ap.add_argument("--inproc", action="store_true", help="N > 1: one process, the library's own multi-device path")
    ap.add_argument("--devices", default="", help="--inproc: explicit HIP ordinals, e.g. 0,0 = two logical shards on one GPU (rehearsal)")
    ap.add_argument("--rehearse-on-device0", action="store_true",
                    help="N > 1 on a ONE-GPU box: every rank renders on device 0 and the all-reduce goes through gloo (RCCL refuses two ranks "
                         "on one device) — exercises the launcher, sharding and timing code, not xGMI")
    ap.add_argument("--pmc-pass", action="store_true",
                    help="under rocprofv3 --pmc: only full-size batches (no warm-up, no instrumented sample, no CPU leg, no event timing)")
    ap.add_argument("--no-kernel-pass", action="store_true", help="skip the second (event-timed) pass")
    ap.add_argument("--c5-in-memory", action="store_true", help="c5: the procedural snapshot directly instead of the .glb + .exr ingestion path")
    ap.add_argument("--drop-in-loop", action="store_true",
                    help="the reference frontend's call pattern: the timed region calls render() for ONE sample at a time (renderer_pt.cpp:131-153), "
                         "steps x spp-per-step times, instead of one call per step; samples_in_flight stays spp-per-step")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: a FIXED render of --spp samples per pixel split over the N GPUs (BASELINE.json configs[3], C4: "
                         "`--gpus 8 --strong --spp 1024` = 8 x 128, one reduce); --steps is derived = ceil(spp / N / spp-per-step); reports time to image")
    ap.add_argument("--launch-timeout", type=float, default=0.0,
                    help="N > 1 started without a launcher: seconds after which the parent ends the rank processes and reports their last lines "
                         "(default %.0f)" % LAUNCH_TIMEOUT_S)
    ap.add_argument("--rank-timeout", type=float, default=540.0,
                    help="N > 1: seconds after which a rank that has not finished dumps its Python stacks to stderr and exits (no GPU call is made "
                         "by the watchdog); 0 = off")
    ap.add_argument("--test-stall-after-start", action="store_true", help=argparse.SUPPRESS)   # tests/test_multi_gpu_gloo.py only: a rank that never comes back
    ap.add_argument("--spp", type=int, default=0, help="--strong: total samples per pixel of the render (default: the workload's full spp, 1024 for c3 = C4)")
    return ap.parse_args(argv)


LAUNCH_TIMEOUT_S = 540.0   # wall limit of a self-launched N-rank run: below the driver's 600 s, so that a stalled rendezvous is reported by
                           # THIS process (each rank's last lines) instead of the whole job being killed "for writing nothing"


def _rank_logs(log_dir):
    """{(local_rank, 'stdout'|'stderr'): path} of the per-rank files torch.distributed.run --redirects 3 writes under --log-dir."""
    out = {}
    for root, _dirs, files in os.walk(log_dir):
        for f in files:
            if f in ("stdout.log", "stderr.log"):
                try:
                    out[(int(os.path.basename(root)), f[:-4])] = os.path.join(root, f)
                except ValueError:
                    pass
    return out


def _tail(path, n=25):
    try:
        with open(path, "r", errors="replace") as fh:
            return fh.read().splitlines()[-n:]
    except OSError:
        return []


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as fresh child processes.  Nothing in THIS process has
    imported torch or touched the GPU (a process that has initialised the GPU must never exec or fork GPU work), so it is the one
    that may watch the clock: after --launch-timeout seconds it ends the child process GROUP, prints what every rank last wrote and
    exits 124.  Ranks are never restarted (--max-restarts 0): a rank that has touched the GPU is only ever replaced by a fresh run."""
    import signal
    import socket
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    log_dir = tempfile.mkdtemp(prefix="ptamd_bench_ranks_")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "--max-restarts", "0", "--log-dir", log_dir, "--redirects", "3",
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    env.setdefault("NCCL_DEBUG", "WARN")
    limit = args.launch_timeout if args.launch_timeout > 0 else LAUNCH_TIMEOUT_S
    t0 = time.monotonic()
    print("bench.py[launcher pid %d]: starting %d ranks (limit %.0f s, per-rank logs under %s)" % (os.getpid(), args.gpus, limit, log_dir),
          file=sys.stderr, flush=True)
    # the launcher's own stdout / stderr go to FILES (an undrained PIPE blocks torchrun as soon as it has written ~64 KiB while its ranks fail)
    l_out_path, l_err_path = os.path.join(log_dir, "launcher_stdout.log"), os.path.join(log_dir, "launcher_stderr.log")
    with open(l_out_path, "w") as fo, open(l_err_path, "w") as fe:
        child = subprocess.Popen(cmd, env=env, stdout=fo, stderr=fe, stdin=subprocess.DEVNULL, start_new_session=True)
    print("bench.py[launcher]: ranks run in process group %d" % child.pid, file=sys.stderr, flush=True)
    timed_out = False
    last_note = t0
    while True:
        try:
            child.wait(timeout=2.0)
            break
        except subprocess.TimeoutExpired:
            pass
        now = time.monotonic()
        if now - last_note >= 30.0:   # a heartbeat: a long multi-rank run is not mistaken for a hung one
            last_note = now
            logs = _rank_logs(log_dir)
            up = sum(1 for (r, k), pth in logs.items() if k == "stderr" and any("process group up" in ln for ln in _tail(pth, 200)))
            print("bench.py[launcher]: %.0f s, %d/%d ranks have a process group" % (now - t0, up, args.gpus), file=sys.stderr, flush=True)
        if now - t0 > limit:
            timed_out = True
            for sig, grace in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 5.0)):
                try:
                    os.killpg(child.pid, sig)      # the launcher AND its ranks (own session = own process group)
                except ProcessLookupError:
                    break
                try:
                    child.wait(timeout=grace)
                    break
                except subprocess.TimeoutExpired:
                    continue
            break
    l_out, l_err = "\n".join(_tail(l_out_path, 1000)), "\n".join(_tail(l_err_path, 200))
    logs = _rank_logs(log_dir)
    line = None
    for ln in _tail(logs.get((0, "stdout"), ""), 1000) + (l_out or "").splitlines():
        ln = ln.split("]:", 1)[-1].strip() if ln.startswith("[") else ln
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    rc = 124 if timed_out else child.returncode
    ok = line is not None and rc == 0
    if ok:
        print(line)
    # every rank's start-up lines on success; every rank's last lines when anything went wrong
    for r in range(args.gpus):
        err = _tail(logs.get((r, "stderr"), ""), 200)
        if ok:
            err = [ln for ln in err if ln.startswith("bench.py[rank")]
        else:
            err = err[-25:]
        for ln in err:
            print("[rank %d] %s" % (r, ln), file=sys.stderr)
        if not ok and not err:
            print("[rank %d] (wrote nothing to stderr%s)" % (r, "" if (r, "stderr") in logs else ": never started"), file=sys.stderr)
    if not ok:
        for ln in (l_err or "").splitlines()[-15:]:
            print("[launcher] " + ln, file=sys.stderr)
        print("bench.py[launcher]: %s after %.0f s; no result line" % ("TIMEOUT: child process group ended" if timed_out else "ranks exited with code %s" % rc,
              time.monotonic() - t0), file=sys.stderr, flush=True)
        return rc if rc not in (0, None) else 1
    import shutil
    shutil.rmtree(log_dir, ignore_errors=True)
    return 0


def load_pmc_profile(workload):
    """Per-work-item counter figures of the committed rocprofv3 --pmc passes (tools/profile_round.sh + tools/summarize_prof.py)."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_%s.json" % workload)))
    if not cands:
        return None, None
    path = cands[-1]  # the newest round's
    try:
        return json.load(open(path)), os.path.relpath(path, ROOT)
    except Exception:
        return None, None


def library_sha16():
    import hashlib
    from platinum_amd import abi
    try:
        return hashlib.sha256(open(abi.library_path(), "rb").read()).hexdigest()[:16]
    except Exception:
        return None


def kernel_block(name, items, item_name, ms, launches, alg_bytes_per_item, pmc, pmc_src, lib_sha, extra=None):
    """The contract's roofline block for one kernel (bound = HBM, counter bytes) + what else could bind it under `secondary`."""
    sec = ms * 1e-3
    unit_item = item_name[:-1]
    out = {"kernel": name, "launches": int(launches), "avg_launch_ms": round(ms / max(1, launches), 4),
           item_name + "_per_launch": round(items / max(1, launches), 1),
           "g%s_per_s" % item_name: round(items / sec / 1e9, 4) if sec > 0 else 0.0}
    alg = items * alg_bytes_per_item / sec / 1e9 if sec > 0 else 0.0
    k = (pmc or {}).get("kernels", {}).get(name)
    secondary = {}
    achieved, traffic, src = None, None, None
    if k:
        stale = bool(lib_sha and pmc.get("library_sha16") and pmc["library_sha16"] != lib_sha)
        src = {"source": pmc_src, "stale": stale}
        if k.get("hbm_bytes_per_item") is not None:
            achieved = items * k["hbm_bytes_per_item"] / sec / 1e9 if sec > 0 else 0.0
            traffic = k["hbm_bytes_per_item"] * items / max(1, launches)
            out["counter_bytes_per_%s" % unit_item] = round(k["hbm_bytes_per_item"], 1)
            out["fetch_correction"] = k.get("fetch_correction")
            if k.get("fetch_split"):
                out["fetch_split"] = {a: round(b, 1) for a, b in k["fetch_split"].items()}
        if k.get("valu_insts_per_item") is not None:
            tl = items * k["valu_insts_per_item"] * 64.0 / sec / 1e12 if sec > 0 else 0.0
            secondary["valu_issue"] = {"wave_insts_per_%s" % unit_item: round(k["valu_insts_per_item"], 1), "achieved_Tlaneops": round(tl, 2),
                                       "peak_Tlaneops": VALU_PEAK_TLOPS, "frac": round(tl / VALU_PEAK_TLOPS, 4)}
        if k.get("l1_accesses_per_item") is not None:
            tg = items * k["l1_accesses_per_item"] / sec / 1e9 if sec > 0 else 0.0
            secondary["l1_tag_rate"] = {"tag_accesses_per_%s" % unit_item: round(k["l1_accesses_per_item"], 2), "achieved_G_per_s": round(tg, 1),
                                        "peak_G_per_s": L1_TAG_PEAK_G, "frac": round(tg / L1_TAG_PEAK_G, 4)}
        if k.get("l2_read_requests_per_item") is not None:
            rq = items * k["l2_read_requests_per_item"] / sec / 1e9 if sec > 0 else 0.0
            hr = k.get("l2_hit_rate")
            hr = 0.0 if hr is None else hr
            # hits and misses have separate measured ceilings; the fraction of the request path in use is their sum
            f = rq * hr / L2_HIT_REQ_PEAK_G + rq * (1.0 - hr) / L2_MISS_REQ_PEAK_G
            secondary["l2_request_rate"] = {"requests_per_%s" % unit_item: round(k["l2_read_requests_per_item"], 2), "achieved_Greq_per_s": round(rq, 2),
                                            "l2_hit_rate": round(hr, 4), "hit_Greq_per_s": round(rq * hr, 2), "hit_peak_Greq_per_s": L2_HIT_REQ_PEAK_G,
                                            "miss_Greq_per_s": round(rq * (1.0 - hr), 2), "miss_peak_Greq_per_s": L2_MISS_REQ_PEAK_G, "frac": round(f, 4)}
    counter_based = achieved is not None
    # no committed counter pass for this workload: no fraction is claimed (the algorithmic bytes are served by L1 / L2 / Infinity Cache and
    # exceed the HBM peak: a "fraction" of them is not a roofline — VERDICT r2); the algorithmic figure stays, labelled
    out.update({"bound": "hbm", "achieved": round(achieved, 1) if counter_based else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4) if counter_based else None,
                "traffic": traffic, "achieved_is": "counter bytes (FETCH_SIZE + WRITE_SIZE, corrected) / launch time" if counter_based
                else "not measured: no counter pass committed for this workload (tools/profile_round.sh <workload>)",
                "algorithmic_bytes_per_%s" % unit_item: round(alg_bytes_per_item, 1), "algorithmic_GBs": round(alg, 1)})
    if counter_based and achieved > 0:
        out["algorithmic_over_counter"] = round(alg / achieved, 2)
    if src:
        out["counters"] = src
    if secondary:
        out["secondary"] = secondary
        out["binding_secondary"] = max(secondary, key=lambda b: secondary[b]["frac"])
    if extra:
        out.update(extra)
    return out


def main():
    args = parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL / cross-process tensor sharing need on this pool
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.inproc:
        sys.exit(self_launch(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    def say(msg):
        print("bench.py[rank %d/%d pid %d] %s" % (rank, world, os.getpid(), msg), file=sys.stderr, flush=True)

    def watchdog(seconds):
        """(Re-)arm this rank's watchdog for the phase that starts now; 0 / single-rank runs: off."""
        if world > 1 and args.rank_timeout > 0:
            import faulthandler
            faulthandler.cancel_dump_traceback_later()
            if seconds > 0:
                faulthandler.dump_traceback_later(seconds, exit=True, file=sys.stderr)

    if world > 1:
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        say("started: LOCAL_RANK %s, device ordinal %d, MASTER %s:%s" % (os.environ.get("LOCAL_RANK"), 0 if args.rehearse_on_device0 else local_rank,
            os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")))
        # a stalled rendezvous / collective: this rank reports where it stands and exits by itself (the watchdog thread only writes
        # and calls _exit: it never touches the GPU, and the rank is not restarted).  It bounds one PHASE at a time (rendezvous, set-up +
        # warm-up, then the timed / kernel passes with an allowance derived from the measured warm-up step), not the run as a whole:
        # a legitimately long job (--strong, many steps) re-arms it as it goes.
        watchdog(args.rank_timeout)
        if args.test_stall_after_start:   # (hidden test-only flag: what a stalled rendezvous looks like from outside)
            time.sleep(1e6)

    import numpy as np  # noqa: F401
    import torch

    inproc = args.inproc and args.gpus > 1
    if inproc and world != 1:
        raise SystemExit("--inproc drives all devices from one process: do not start it under torch.distributed.run")
    if not inproc and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if args.rehearse_on_device0:
        local_rank = 0
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        t_pg = time.perf_counter()
        if args.rehearse_on_device0:
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=180))
        else:
            if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
                say("no HIP device %d (torch sees %d): this rank cannot run" % (local_rank, torch.cuda.device_count() if torch.cuda.is_available() else 0))
                raise SystemExit(3)
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank),  # nccl == RCCL on ROCm
                                    timeout=datetime.timedelta(seconds=180))
        say("process group up: backend %s, %.1f s" % (dist.get_backend(), time.perf_counter() - t_pg))
        watchdog(args.rank_timeout)   # the next phase: scene, BVH, queues, warm-up
    have_gpu = torch.cuda.is_available()
    if not have_gpu and not (world > 1 and args.rehearse_on_device0):
        if world > 1:
            say("no HIP device: this rank cannot render (rendezvous was fine)")
        raise SystemExit("bench.py needs a HIP device" if world == 1 else 3)

    from platinum_amd import Renderer, abi, scenes
    from platinum_amd.sharding import reduce_accumulator, shard_samples

    factory, W, H, full_spp, B = scenes.CONFIGS[args.workload]
    S, K = args.spp_per_step, args.steps
    if S <= 0:
        import ctypes as _C
        plan = abi.QueuePlan()
        # (a gloo rehearsal on a box WITHOUT a GPU still takes part in the job's one host-side collective before it refuses: tests/test_multi_gpu_gloo.py)
        free_b, _tot = torch.cuda.mem_get_info(local_rank) if have_gpu else (0, 0)
        if args.inproc and args.devices:  # logical shards that share a device share its memory
            _d = [int(x) for x in args.devices.split(",")]
            free_b //= max(_d.count(x) for x in set(_d))
        abi.check(abi.load_library(), abi.load_library().pt_plan_queues(W, H, 1 << 20, 0, int(free_b), 0, 4, _C.byref(plan)))
        S = int(plan.samples_in_flight)
        if dist is not None:
            # every rank must trace the SAME batch size: the sample ranges [g * K * S, (g + 1) * K * S) tile the render only then (a GPU with
            # less free memory would plan a smaller batch on its own)
            t = torch.tensor([S], dtype=torch.int64, device="cpu" if args.rehearse_on_device0 else torch.device("cuda", local_rank))
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if int(t.item()) != S:
                say("planned %d samples in flight, the job runs %d (the smallest plan of all ranks)" % (S, int(t.item())))
            S = int(t.item())
            say("job batch: %d samples in flight on every rank" % S)
    if not have_gpu:
        say("no HIP device: this rank cannot render (rendezvous and the batch-size all-reduce were fine)")
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        watchdog(0)
        raise SystemExit(3)
    ndev_all = args.gpus if (args.inproc and args.gpus > 1) else world
    strong_spp = 0
    if args.strong:
        # C4: the SAME render (1024 spp on the C3 scene) whatever N is; every device gets spp / N samples
        strong_spp = args.spp or (1024 if args.workload == "c3" else full_spp)
        per_dev = -(-strong_spp // ndev_all)
        S = min(S, per_dev)
        K = -(-per_dev // S)
    Wu = 0 if args.pmc_pass else args.warmup
    if args.workload == "c5" and not args.c5_in_memory:
        # the C5-class scene enters as FILES through scene ingestion (pt_scene_import_gltf + pt_scene_load_environment): a .glb with
        # JPEG / PNG textures and a 4096x2048 OpenEXR environment, written once by tools/export_gltf.py (no asset exists offline)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import export_gltf
        scene = export_gltf.atrium_through_ingestion(os.path.join(os.environ.get("TMPDIR", "/tmp"), "ptamd_c5_cache"))
    else:
        scene = factory()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    ndev = args.gpus if inproc else 1       # devices (group members) driven by THIS process
    members_all = args.gpus if inproc else world   # shards of the whole job
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)  # the accumulator lives in a torch tensor so RCCL can reduce it

    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(ndev))
    if inproc and len(devices) != ndev:
        raise SystemExit("--devices must list --gpus ordinals")
    # n_gpus = PHYSICAL devices doing the work: logical shards on one GPU (--devices 0,0 / --rehearse-on-device0) are a rehearsal
    # of the sharding + merge code, not a multi-GPU measurement, and are labelled as such
    n_gpus = len(set(devices)) if inproc else (1 if args.rehearse_on_device0 else world)
    rehearsal = n_gpus != members_all
    r = Renderer(devices=devices) if inproc else Renderer(device=local_rank)
    if world > 1:
        ri = abi.runtime_info()
        say("renderer on device %d; HIP runtime %s (%d mapped), RCCL %s" % (local_rank, ri["hip_runtime_path"], ri["hip_runtimes_mapped"],
            ri["rccl_path"] or "(torch's, bound by torch.distributed)"))
    # samples of this process: K*S per device, timed; the warm-up renders (and discards) W*S of the same range first
    total_spp = K * S * ndev
    if args.strong:
        total_spp = -(-strong_spp // ndev_all) * ndev      # (this process's share of the fixed render)
    first, _ = shard_samples(rank, world, total_spp)
    policy = abi.NONFINITE_ZERO

    def start(spp, first_sample):
        # NONFINITE_ZERO: a NaN/inf sample (the reference's BSDF yields ~1 per 5e8 paths) counts as black instead of
        # poisoning its pixel's running mean; the count is reported in extra.nonfinite_samples.
        r.startRender(scene, (W, H), spp, max_bounces=B, first_sample=first_sample, samples_in_flight=S,
                      external_accumulator=acc.data_ptr(), nonfinite_policy=policy)

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed_pass():
        sync()
        t0 = time.perf_counter()
        if args.drop_in_loop:
            for _ in range(K * S):
                r.render(ndev)                 # one sample (per device) per call, as Frontend::start does once per UI frame
        else:
            for _ in range(K):
                r.render(S * ndev)
        r.wait()                               # (--inproc: includes the library's RCCL all-reduce of the per-device accumulators)
        reduce_accumulator(acc, world, dist)   # one process per GPU: the single RCCL sum-reduce of the accumulation buffer
        sync()
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    # ---- warm-up: W untimed steps ----
    step_s = 0.0
    if Wu > 0:
        start(Wu * S * ndev, first)
        t_w = time.perf_counter()
        for _ in range(Wu):
            r.render(S * ndev)
        r.wait()
        step_s = (time.perf_counter() - t_w) / Wu
    # each remaining pass (instrumented sample, timed, event-timed) gets the base allowance + 4x its expected length
    pass_allowance = args.rank_timeout + 4.0 * step_s * K
    # ---- instrumented sample (outside the timed region): BVH nodes / triangles fetched per ray ----
    nodes_c = tris_c = nodes_s = tris_s = 0.0
    if not args.pmc_pass:
        start(total_spp, first)
        r.measureTraversal(first)
        st0 = r.stats()
        nodes_c, tris_c = st0.nodes_per_closest_ray, st0.tris_per_closest_ray
        nodes_s, tris_s = st0.nodes_per_shadow_ray, st0.tris_per_shadow_ray

    # ---- the timed region: K steps, per-kernel event timing off ----
    watchdog(pass_allowance)
    start(total_spp, first)
    r.setProfiling(False)
    elapsed = timed_pass()
    watchdog(pass_allowance)
    st = r.stats()
    # the batch the library actually traced (it plans again at start, against the memory left after the scene and the accumulator): a step
    # is only "one batch" when that equals S - otherwise say so instead of labelling several batches as one step
    S_lib = int(st.samples_in_flight)
    if S_lib != S and rank == 0:
        print("bench.py: the library runs %d samples in flight, not the %d a step asks for: one step = %d batches" % (S_lib, S, -(-S // max(1, S_lib))), file=sys.stderr)
    spp_all = total_spp * (1 if inproc else world)   # samples per pixel the whole job accumulated in the timed region
    value = W * H * spp_all * B / elapsed / 1e6
    mean_radiance = float(acc[..., :3].mean().item())

    # ---- kernel pass: the same K steps again with HIP events around every launch (not part of `value`) ----
    ks, elapsed_k = st, None
    if not args.pmc_pass and not args.no_kernel_pass:
        start(total_spp, first)
        r.setProfiling(True)
        elapsed_k = timed_pass()
        ks = r.stats()
        r.setProfiling(False)

    # ---- counter calibration (--pmc-pass only): a streaming read of exactly known size through a kernel whose access pattern never changes -
    # torch's vectorised elementwise kernel over a 1 GiB tensor, 16 B per lane - so that tools/summarize_prof.py can check MI355X_MICROARCH.md's "FETCH_SIZE
    # reports half of a wide coalesced streaming read" on THIS box in THIS pass (r4 derived it from k_accumulate, whose pattern r4 changed)
    calib_copy = None
    if args.pmc_pass and rank == 0:
        nbytes, copies = 1 << 30, 3
        src = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        torch.cuda.synchronize(dev)
        for _ in range(copies):
            torch.add(src, 1.0, out=dst)   # (dst.copy_(src) would go through hipMemcpyDtoD's blit kernel)
        torch.cuda.synchronize(dev)
        calib_copy = {"kernel_name_contains": "vectorized_elementwise_kernel", "bytes_per_copy": nbytes, "copies": copies,
                      "note": "torch.add(src, 1.0, out=dst) on 1 GiB float32: reads and writes exactly bytes_per_copy per dispatch (the normal_() fill writes only)"}
        del src, dst

    lib_sha = library_sha16()
    pmc, pmc_src = load_pmc_profile(args.workload)
    if rank == 0 and pmc is not None:
        stale = bool(lib_sha and pmc.get("library_sha16") and pmc["library_sha16"] != lib_sha)
        print("bench.py: counter figures from %s (taken on library %s; this library %s%s)" % (pmc_src, pmc.get("library_sha16"), lib_sha,
              " - STALE: the roofline's counter bytes are NOT of this build, re-run tools/profile_round.sh" if stale else ""), file=sys.stderr)
    blocks = {}
    if ks.ms_closest > 0:
        blocks["k_trace_closest"] = kernel_block(
            "k_trace_closest", ks.closest_rays, "rays", ks.ms_closest, ks.launches_closest, algorithmic_bytes_closest(nodes_c, tris_c), pmc, pmc_src,
            lib_sha, {"nodes_per_ray": round(nodes_c, 2), "tris_per_ray": round(tris_c, 2)})
        blocks["k_shade"] = kernel_block("k_shade", ks.shaded_hits, "hits", ks.ms_shade, ks.launches_closest, ALGORITHMIC_BYTES_SHADE, pmc, pmc_src, lib_sha)
        if ks.ms_shadow > 0:
            blocks["k_trace_shadow"] = kernel_block(
                "k_trace_shadow", ks.shadow_rays, "rays", ks.ms_shadow, ks.launches_shadow, algorithmic_bytes_shadow(nodes_s, tris_s), pmc, pmc_src,
                lib_sha, {"nodes_per_ray": round(nodes_s, 2), "tris_per_ray": round(tris_s, 2)})
        for name in ("k_trace_closest", "k_trace_shadow"):
            b = blocks.get(name)
            if b and b.get("traffic") is not None:
                # north_star: ">= 40 % of HBM peak on the traversal kernel", read on the bytes the HBM counters saw
                b["target_frac"] = HBM_TARGET_FRAC
                b["target_met"] = bool(b["frac"] >= HBM_TARGET_FRAC)
    times = {"k_trace_closest": ks.ms_closest, "k_shade": ks.ms_shade, "k_trace_shadow": ks.ms_shadow}
    dominant = max(times, key=times.get) if blocks else None
    roofline = dict(blocks[dominant]) if dominant else None
    if roofline is not None:
        roofline["share_of_kernel_time"] = round(times[dominant] / max(1e-9, sum(times.values()) + ks.ms_raygen + ks.ms_accumulate), 4)

    out = {
        "metric": "Msamples/s (paths x spp x bounces / s) at %dx%d, %d bounces" % (W, H, B),
        "value": round(value, 2),
        "unit": "Msamples/s",
        "n_gpus": n_gpus,
        "steps": K,
        "warmup": Wu,
        "ms_per_step": round(elapsed / K * 1e3, 3),
        "higher_is_better": True,
        "scaling": "strong" if args.strong else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": WORKLOADS[args.workload],
            "width": W, "height": H, "max_bounces": B, "spp_per_step": S, "samples_in_flight": S_lib, "batches_per_step": -(-S // max(1, S_lib)), "spp_per_gpu": total_spp // ndev, "spp_total": spp_all,
            "integrator": "MIS+NEE", "flags": "MultiscatterGGX", "triangles": int(st.triangles),
            "call_pattern": "render(1) per call, merged by the library (reference frontend loop)" if args.drop_in_loop else "render(spp_per_step) per step",
            "batches": int(st.batches),
            "nonfinite_policy": "zero (a NaN/inf sample counts as black; parity default is propagate)",
            "parallelism": (("sample-sharded x%d, one process, library device group%s" % (members_all, " + RCCL all-reduce" if n_gpus > 1 else "")) if inproc else
                            ("sample-sharded x%d, one process per GPU + RCCL all-reduce" % members_all)) +
                           ((" [REHEARSAL: %d logical shards on %d physical GPU(s); no RCCL call is made]" % (members_all, n_gpus)) if rehearsal else ""),
            "logical_shards": members_all,
        },
        "roofline": roofline,
        "roofline_kernels": blocks,
        "extra": {
            "closest_rays": int(st.closest_rays), "shadow_rays": int(st.shadow_rays), "shaded_hits": int(st.shaded_hits),
            "paths": int(st.paths), "nonfinite_samples": int(st.nonfinite_samples), "mean_path_segments": round(st.closest_rays / max(1, st.paths), 3),
            "bvh_build_ms": round(st.bvh_build_ms, 3), "bvh_build_mtris_per_s": round(st.triangles / max(st.bvh_build_ms, 1e-9) / 1e3, 1),
            "bvh_nodes": int(st.bvh_nodes), "bvh_depth4": int(st.bvh_max_depth),
            "upload_ms": round(st.upload_ms, 3),
            "kernel_ms": {"raygen": round(ks.ms_raygen, 2), "closest": round(ks.ms_closest, 2), "shade": round(ks.ms_shade, 2),
                          "shadow": round(ks.ms_shadow, 2), "accumulate": round(ks.ms_accumulate, 2),
                          "note": "separate pass of the same %d steps with HIP events around every launch" % K,
                          "pass_wall_ms": round(elapsed_k * 1e3, 2) if elapsed_k else None},
            "wall_ms": round(elapsed * 1e3, 2),
            "time_to_image_ms": round(elapsed * 1e3, 2) if args.strong else None,
            "mean_radiance": mean_radiance,
            "library_sha16": lib_sha,
        },
    }
    if calib_copy:
        out["calibration_copy"] = calib_copy

    # ---- CPU baseline (rank 0, N = 1 only), bounded sample of the same workload ----
    # "the same __host__ __device__ kernels built for the host, all host cores" - tests/emu compiles the product's own
    # stage functions (platinum_amd/csrc/pt_*.h: sampler, camera, 6-wide traversal, BSDF, NEE / MIS, accumulate) with g++ -ffp-contract=off and
    # runs them on std::thread workers over 16x16 pixel tiles (kind "same-kernels-host").  The scalar oracle (oracle/pt_oracle.cpp, its own
    # median-split BVH) is timed beside it on a shorter sample (`oracle`, kind "port").  Neither is the target; the roofline is.
    if members_all == 1 and not args.no_cpu_baseline and not args.pmc_pass:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import numpy as _np
        import emu_lib
        import oracle_lib
        from platinum_amd.renderer import make_params
        threads = args.cpu_threads or available_cpus()
        cpu_model = "?"
        try:
            for ln in open("/proc/cpuinfo"):
                if ln.startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
        except OSError:
            pass

        def bounded(render, seconds):
            """1 spp to calibrate, then enough further spp (>= 4) for ~`seconds` of CPU work; only the second run is reported."""
            t0 = time.perf_counter()
            a = render(0, 1, None, 0)
            t1 = time.perf_counter() - t0
            n = int(max(4, min(63, round(seconds / max(t1, 1e-3)))))
            t0 = time.perf_counter()
            a = render(1, n, a, 1)
            dt = time.perf_counter() - t0
            return n, dt, W * H * n * B / dt / 1e6, float(a[..., :3].mean())

        # the product's tree: Morton order, PLOC radius 8, SAH collapse to 6-wide nodes, leaf slots of one or two triangles (unless the
        # library itself runs with $PTAMD_NO_PAIRS)
        os.environ.update(EMU_MORTON="1", EMU_PLOC="8", EMU_WIDE6="1")
        pairs = "PTAMD_NO_PAIRS" not in os.environ
        if pairs:
            os.environ["EMU_PAIRS"] = "1"
        else:
            os.environ.pop("EMU_PAIRS", None)
        t0 = time.perf_counter()
        e = emu_lib.EmuScene(scene, make_params(W, H, 64, B))
        host_build_s = time.perf_counter() - t0
        n_e, dt_e, v_e, mean_e = bounded(lambda f, n, a, n0: e.render(f, n, acc=a, acc_n0=n0, threads=threads), args.cpu_seconds)
        del e
        o = oracle_lib.OracleScene(scene, make_params(W, H, 64, B), use_bvh=True)
        n_o, dt_o, v_o, _ = bounded(lambda f, n, a, n0: o.render(f, n, acc=a, acc_n0=n0, threads=threads), args.cpu_seconds * 0.4)
        out["cpu_baseline"] = {
            "value": round(v_e, 3), "unit": "Msamples/s", "cores": threads, "kind": "port", "port_of": "same-kernels-host", "cpu_model": cpu_model,
            "sample": "%dx%d x %d spp x %d bounces (sample indices 1..%d) of the same scene: the product's stage functions and 6-wide BVH traversal "
                      "(%s leaf slots) compiled for the host (tests/emu), std::thread over 16x16 tiles, %.1f s; host BVH build %.1f s not included"
                      % (W, H, n_e, B, n_e, "pair" if pairs else "one-triangle", dt_e, host_build_s),
            "mean_radiance": mean_e,
            "gpu_over_cpu": round(value / v_e, 1),
            "oracle": {"value": round(v_o, 3), "unit": "Msamples/s", "cores": threads, "kind": "port",
                       "sample": "%d spp of the same workload, scalar oracle with its own BVH, %.1f s" % (n_o, dt_o)},
        }

    if rank == 0:
        print(json.dumps(out), flush=True)
    r.close()
    if dist is not None:
        say("done: %.1f ms timed" % (elapsed * 1e3))
        dist.destroy_process_group()
        if args.rank_timeout > 0:
            import faulthandler
            faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
