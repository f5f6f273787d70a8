#!/usr/bin/env python3
"""bench.py — Msamples/s of the MI355X wavefront path tracer on BASELINE.json's configs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c1] [--spp-per-step S]

One "step" = one pass of the hot path over one batch: S samples per pixel of the whole frame (raygen ->
{closest-hit, shade/BSDF/NEE, shadow} x bounces -> accumulate).  Default: C2 (Cornell + GGX dielectric sphere,
1920x1080, 8 bounces), S = 8, K = 32  => the full 256 spp of BASELINE.json configs[1].
Metric: Msamples/s = W*H*spp*B / t / 1e6, B = configured max bounces.

N > 1 (launched by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`): one process per GPU;
rank g renders the same frame for sample indices [g*K*S, (g+1)*K*S) — independent-sample sharding, no exchange
while rendering — then ONE RCCL all-reduce (sum) of the float accumulator over xGMI inside the timed region.
Weak scaling: per-GPU work is fixed.

Printed by rank 0: ONE JSON line (contract in the task statement) with `roofline` (closest-hit traversal kernel,
HBM-bound accounting) and, at N = 1, `cpu_baseline` (the CPU oracle on the host cores, bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes_closest(nodes_per_ray, tris_per_ray):
    # ray 32 B + hit 20 B + N_node * 64 B + N_tri * 36 B
    return 32.0 + 20.0 + 64.0 * nodes_per_ray + 36.0 * tris_per_ray


def algorithmic_bytes_shadow(nodes_per_ray, tris_per_ray):
    return 32.0 + 4.0 + 64.0 * nodes_per_ray + 36.0 * tris_per_ray


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "c5"])
    ap.add_argument("--spp-per-step", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm

    from platinum_amd import Renderer, abi, scenes
    from platinum_amd.sharding import reduce_accumulator, shard_samples

    factory, W, H, full_spp, B = scenes.CONFIGS[args.workload]
    S, K, Wu = args.spp_per_step, args.steps, args.warmup
    scene = factory()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)  # the accumulator lives in a torch tensor so RCCL can reduce it

    r = Renderer(device=local_rank)
    # samples of this rank: warm-up first (discarded by the restart below), then K*S timed
    total_spp = K * S
    first, _ = shard_samples(rank, world, total_spp)

    def start(spp, first_sample):
        # NONFINITE_ZERO: a NaN/inf sample (the reference's BSDF yields ~1 per 5e8 paths) counts as black instead of
        # poisoning its pixel's running mean; the count is reported in extra.nonfinite_samples.
        r.startRender(scene, (W, H), spp, max_bounces=B, first_sample=first_sample, samples_in_flight=S,
                      external_accumulator=acc.data_ptr(), nonfinite_policy=abi.NONFINITE_ZERO)

    # ---- warm-up: W untimed steps ----
    if Wu > 0:
        start(Wu * S, first)
        for _ in range(Wu):
            r.render(S)
        r.wait()
    # ---- instrumented sample (outside the timed region): BVH nodes / triangles fetched per ray ----
    start(total_spp, first)
    r.measureTraversal(first)
    st0 = r.stats()
    nodes_c, tris_c = st0.nodes_per_closest_ray, st0.tris_per_closest_ray
    nodes_s, tris_s = st0.nodes_per_shadow_ray, st0.tris_per_shadow_ray
    # restart so the instrumented sample is not part of the timed render
    start(total_spp, first)
    r.setProfiling(True)

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    sync()
    t0 = time.perf_counter()
    for _ in range(K):
        r.render(S)
    r.wait()
    reduce_accumulator(acc, world, dist)  # the single RCCL sum-reduce of the accumulation buffer (N > 1)
    sync()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    st = r.stats()
    value = W * H * (K * S) * B * world / elapsed / 1e6

    # ---- roofline of the dominant kernel (closest-hit traversal), HBM-bound accounting ----
    bytes_closest = st.closest_rays * algorithmic_bytes_closest(nodes_c, tris_c)
    bytes_shadow = st.shadow_rays * algorithmic_bytes_shadow(nodes_s, tris_s)
    ms_closest, ms_shadow = st.ms_closest, st.ms_shadow
    achieved = bytes_closest / (ms_closest * 1e-3) / 1e9 if ms_closest > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("closest_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    launches = max(1, st.launches_closest)
    roofline = {
        "kernel": "k_trace_closest",
        "bound": "hbm",
        "achieved": round(achieved, 2),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5),
        "traffic": traffic,
        "bytes_per_ray": round(algorithmic_bytes_closest(nodes_c, tris_c), 1),
        "nodes_per_ray": round(nodes_c, 2), "tris_per_ray": round(tris_c, 2),
        "rays_per_launch": st.closest_rays / launches,
        "avg_launch_ms": ms_closest / launches,
        "launches": int(st.launches_closest),
        "grays_per_s": round(st.closest_rays / (ms_closest * 1e-3) / 1e9, 4) if ms_closest > 0 else 0.0,
        "shadow_kernel": {
            "achieved": round(bytes_shadow / (ms_shadow * 1e-3) / 1e9, 2) if ms_shadow > 0 else 0.0,
            "nodes_per_ray": round(nodes_s, 2), "tris_per_ray": round(tris_s, 2),
            "grays_per_s": round(st.shadow_rays / (ms_shadow * 1e-3) / 1e9, 4) if ms_shadow > 0 else 0.0,
        },
    }

    out = {
        "metric": "Msamples/s (paths x spp x bounces / s) at %dx%d, %d bounces" % (W, H, B),
        "value": round(value, 2),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": K,
        "warmup": Wu,
        "ms_per_step": round(elapsed / K * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": {"c1": "C1 Cornell box 512x512", "c2": "C2 Cornell box + GGX dielectric sphere (6144 tris), 1920x1080",
                         "c3": "C3 1.04M-triangle instanced sphere field, 1920x1080",
                         "c5": "C5 procedural Sponza-class atrium (258k tris, textures, cut-outs, 4096x2048 environment), 3840x2160"}[args.workload],
            "width": W, "height": H, "max_bounces": B, "spp_per_step": S, "spp_per_gpu": K * S, "spp_total": K * S * world,
            "integrator": "MIS+NEE", "flags": "MultiscatterGGX", "triangles": int(st.triangles),
            "parallelism": "sample-sharded x%d" % world,
        },
        "roofline": roofline,
        "extra": {
            "closest_rays": int(st.closest_rays), "shadow_rays": int(st.shadow_rays), "shaded_hits": int(st.shaded_hits),
            "paths": int(st.paths), "nonfinite_samples": int(st.nonfinite_samples), "mean_path_segments": round(st.closest_rays / max(1, st.paths), 3),
            "bvh_build_ms": round(st.bvh_build_ms, 3), "bvh_build_mtris_per_s": round(st.triangles / max(st.bvh_build_ms, 1e-9) / 1e3, 1),
            "bvh_nodes": int(st.bvh_nodes), "bvh_depth4": int(st.bvh_max_depth),
            "upload_ms": round(st.upload_ms, 3),
            "kernel_ms": {"raygen": round(st.ms_raygen, 2), "closest": round(ms_closest, 2), "shade": round(st.ms_shade, 2),
                          "shadow": round(ms_shadow, 2), "accumulate": round(st.ms_accumulate, 2)},
            "wall_ms": round(elapsed * 1e3, 2),
            "mean_radiance": float(acc[..., :3].mean().item()),
        },
    }

    # ---- CPU baseline: the oracle on the host cores, bounded sample of the same workload (rank 0, N = 1 only) ----
    if world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        from platinum_amd.renderer import make_params
        threads = args.cpu_threads or min(os.cpu_count() or 1, 16)
        o = oracle_lib.OracleScene(scene, make_params(W, H, 64, B), use_bvh=True)
        # bounded sample: 1 spp to calibrate, then enough further spp for ~12 s of CPU work; only the second run is reported
        tc0 = time.perf_counter()
        cpu_acc = o.render(0, 1, threads=threads)
        t1spp = time.perf_counter() - tc0
        cpu_spp = int(max(1, min(63, round(12.0 / max(t1spp, 1e-3)))))
        tc0 = time.perf_counter()
        cpu_acc = o.render(1, cpu_spp, acc=cpu_acc, acc_n0=1, threads=threads)
        tc = time.perf_counter() - tc0
        cpu_value = W * H * cpu_spp * B / tc / 1e6
        out["cpu_baseline"] = {
            "value": round(cpu_value, 3), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": "%dx%d x %d spp x %d bounces (sample indices 1..) of the same scene, oracle with its own BVH, %.1f s"
                      % (W, H, cpu_spp, B, tc),
            "gpu_over_cpu": round(value / cpu_value, 1),
        }
        del cpu_acc

    if rank == 0:
        print(json.dumps(out))
    r.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
