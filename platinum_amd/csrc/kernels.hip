// kernels.hip — the gfx950 wavefront path-tracing kernels.
//
// One batch = `nsamples` samples of every pixel traced together.  Per bounce b:
//     k_trace_closest(b)  ->  k_shade(b)  ->  k_trace_shadow(b)
// preceded by k_raygen and followed by k_accumulate.  All kernels are stream-ordered; queue sizes never visit
// the host: they live in BatchCounters and every kernel is a persistent grid whose waves pull 64-path chunks
// with one atomic per wave.  k_shade compacts survivors into the other PathState buffer (ballot + prefix +
// one atomic per wave) and appends NEE shadow rays to the shadow queue the same way.
//
// Compiled with -ffp-contract=off (deterministic fp32 contract, pt_math.h).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "pt_shade.h"

namespace pt {

// ---- wave helpers (wave64) -------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t wave_prefix(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// All 64 lanes must be active. Lane 0 grabs `n` slots from *cursor; everyone gets the base.
__device__ __forceinline__ uint32_t wave_alloc(uint32_t* cursor, uint32_t n, uint32_t lane) {
  uint32_t base = 0;
  if (lane == 0 && n > 0) base = atomicAdd(cursor, n);
  return __builtin_amdgcn_readfirstlane(base);
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- raygen ------------------------------------------------------------------------------------------------------
// Threads cover `nsamples` copies of the image padded to 8x8 tiles; one wave = one tile of one sample, so a wave's
// camera rays are a coherent 8x8 bundle.  Valid lanes are compacted into queue 0.
__global__ void __launch_bounds__(kBlock) k_raygen(DeviceScene S, PathState st, vec4* __restrict__ Lbuf,
                                                    BatchCounters* __restrict__ ctr, uint32_t first_sample,
                                                    uint32_t nsamples, uint32_t tilesX, uint32_t tilesY) {
  const uint32_t lane = wave_lane();
  const uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const uint32_t tiles = tilesX * tilesY;
  const uint64_t wave_id = gid >> 6;
  const uint32_t s = (uint32_t)(wave_id / tiles);
  const uint32_t tile = (uint32_t)(wave_id % tiles);
  const uint32_t x = (tile % tilesX) * 8 + (lane & 7);
  const uint32_t y = (tile / tilesX) * 8 + (lane >> 3);
  const bool valid = s < nsamples && x < S.width && y < S.height;

  RayGenOut rg;
  if (valid) rg = stage_raygen(S, x, y, first_sample + s);

  const unsigned long long m = __ballot(valid);
  const uint32_t base = wave_alloc(&ctr->active[0], (uint32_t)__popcll(m), lane);
  if (valid) {
    const uint32_t j = base + wave_prefix(m);
    const uint32_t pid = s * (S.width * S.height) + y * S.width + x;
    st.rayO[j] = vec4{rg.o.x, rg.o.y, rg.o.z, 0.0f};
    st.rayD[j] = vec4{rg.d.x, rg.d.y, rg.d.z, u2f(rg.dim & kMetaDimMask)};
    st.att[j] = vec4{1.0f, 1.0f, 1.0f, u2f(rg.offset)};
    st.pid[j] = pid;
    Lbuf[pid] = vec4{0.0f, 0.0f, 0.0f, 1.0f};
  }
}

// ---- traversal kernels: persistent waves with per-lane ray replacement ---------------------------------------------
// A wave keeps up to 64 rays in flight.  Whenever fewer than `refill_threshold` lanes still hold a ray (and the queue is
// not exhausted) the wave leaves the traversal loop, idle lanes pull new rays (ballot + one atomic per wave) and
// everybody resumes where they were: a lane's traversal state (TravState + its LDS stack column) survives the refill.
// This removes the "whole wave waits for its slowest ray" tail of a plain persistent loop.

struct WaveQueue {
  uint32_t* cursor;
  uint32_t count;
  bool exhausted;
};
// Idle lanes (need == true) get the index of a fresh queue entry, or kInvalidRef when the queue ran dry.
__device__ __forceinline__ uint32_t wave_refill(WaveQueue& q, bool need, uint32_t lane) {
  uint32_t idx = kInvalidRef;
  if (q.exhausted) return idx;
  const unsigned long long m = __ballot(need);
  const uint32_t n = (uint32_t)__popcll(m);
  if (n == 0) return idx;
  const uint32_t base = wave_alloc(q.cursor, n, lane);
  if (need) {
    const uint32_t i = base + wave_prefix(m);
    if (i < q.count) idx = i;
  }
  if (base + n >= q.count) q.exhausted = true;
  return idx;
}

template <bool COUNT>
__global__ void __launch_bounds__(kBlock) k_trace_closest(DeviceScene S, PathState st, vec4* __restrict__ hit,
                                                           BatchCounters* __restrict__ ctr, uint32_t bounce,
                                                           uint32_t* __restrict__ spill, int32_t* __restrict__ hitlog,
                                                           uint32_t log_stride, uint32_t refill_threshold) {
  __shared__ uint32_t lds_stack[kLdsStack][kBlock];
  const uint32_t lane = wave_lane();
  WaveQueue q{&ctr->work[3 * bounce + 0], ctr->active[bounce], false};
  TraversalStack stack;
  stack.lds = &lds_stack[0][threadIdx.x];
  stack.lds_stride = kBlock;
  stack.spill = spill + ((size_t)blockIdx.x * kBlock + threadIdx.x);
  stack.spill_stride = gridDim.x * kBlock;
  TraversalCount tc;
  TravState ts;
  uint32_t ray = kInvalidRef;  // queue index of the ray this lane is tracing

  auto finish = [&]() {
    const RayHit& h = ts.best;
    hit[ray] = vec4{h.t, h.u, h.v, u2f(h.tri)};
    if (hitlog) {
      const uint32_t pid = st.pid[ray];
      int32_t* hl = &hitlog[((size_t)bounce * log_stride + pid) * 2];
      hl[0] = h.tri != kInvalidRef ? (int32_t)S.tris[h.tri].inst : -1;
      hl[1] = h.tri != kInvalidRef ? (int32_t)S.tris[h.tri].prim : -1;
    }
    ray = kInvalidRef;
  };

  for (;;) {
    const uint32_t fresh = wave_refill(q, ray == kInvalidRef, lane);
    if (fresh != kInvalidRef) {
      ray = fresh;
      const vec4 o4 = st.rayO[ray];
      const vec4 d4 = st.rayD[ray];
      if (trav_init(S, ts, v3(o4.x, o4.y, o4.z), v3(d4.x, d4.y, d4.z), 1e-3f, kInf, stack, false, COUNT ? &tc : nullptr)) finish();
    }
    if (__ballot(ray != kInvalidRef) == 0) {
      if (q.exhausted) break;
      continue;
    }
    while (ray != kInvalidRef) {
      if (trav_step<false, COUNT>(S, ts, &tc)) finish();
      // lanes still in this loop vote; leave for a refill when the wave got too empty
      if (refill_threshold && !q.exhausted && (uint32_t)__popcll(__ballot(ray != kInvalidRef)) < refill_threshold) break;
    }
  }
  if (COUNT) {
    const uint32_t n = wave_sum(tc.nodes), t = wave_sum(tc.tris);
    if (lane == 0) {
      atomicAdd(&ctr->nodes_closest, (unsigned long long)n);
      atomicAdd(&ctr->tris_closest, (unsigned long long)t);
    }
  }
}

// ---- shade -----------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_shade(DeviceScene S, PathState sin, PathState sout,
                                                   const vec4* __restrict__ hit, ShadowQueue sq, vec4* __restrict__ Lbuf,
                                                   BatchCounters* __restrict__ ctr, uint32_t bounce) {
  const uint32_t lane = wave_lane();
  const uint32_t count = ctr->active[bounce];
  uint32_t* cursor = &ctr->work[3 * bounce + 1];
  uint32_t shaded = 0;

  for (;;) {
    const uint32_t base = wave_alloc(cursor, 64, lane);
    if (base >= count) break;
    const uint32_t i = base + lane;
    bool alive = false, shadow = false;
    ShadeOut out;
    uint32_t pid = 0, offset = 0;
    if (i < count) {
      const vec4 h4 = hit[i];
      const uint32_t tri = f2u(h4.w);
      if (tri != kInvalidRef) {  // a miss adds attenuation * backgroundColor (= 0, defs.metal:21) and ends the path
        const vec4 o4 = sin.rayO[i];
        const vec4 d4 = sin.rayD[i];
        const vec4 a4 = sin.att[i];
        pid = sin.pid[i];
        const uint32_t meta = f2u(d4.w);
        offset = f2u(a4.w);
        ShadeIn in;
        in.o = v3(o4.x, o4.y, o4.z);
        in.d = v3(d4.x, d4.y, d4.z);
        in.att = v3(a4.x, a4.y, a4.z);
        in.lastPdf = o4.w;
        in.lastSpecular = (meta & kMetaSpecular) != 0;
        in.offset = offset;
        in.dim = (meta & kMetaDimMask) + 1;  // +1: the alpha-test payload `ir` drawn before intersect (kernel.metal:510)
        in.bounce = bounce;
        in.t = h4.x; in.u = h4.y; in.v = h4.z;
        in.tri = tri;
        out = stage_shade(S, in);
        shaded++;
        alive = out.alive;
        shadow = out.shadow;
        if (out.has_emitted) {
          vec4 L = Lbuf[pid];
          L.x += out.emitted.x; L.y += out.emitted.y; L.z += out.emitted.z;
          Lbuf[pid] = L;
        }
      }
    }
    // survivors -> next bounce's queue
    {
      const unsigned long long m = __ballot(alive);
      const uint32_t b = wave_alloc(&ctr->active[bounce + 1], (uint32_t)__popcll(m), lane);
      if (alive) {
        const uint32_t j = b + wave_prefix(m);
        sout.rayO[j] = vec4{out.next_o.x, out.next_o.y, out.next_o.z, out.next_pdf};
        sout.rayD[j] = vec4{out.next_d.x, out.next_d.y, out.next_d.z,
                            u2f((out.dim & kMetaDimMask) | (out.next_specular ? kMetaSpecular : 0u))};
        sout.att[j] = vec4{out.next_att.x, out.next_att.y, out.next_att.z, u2f(offset)};
        sout.pid[j] = pid;
      }
    }
    // NEE shadow rays
    {
      const unsigned long long m = __ballot(shadow);
      const uint32_t b = wave_alloc(&ctr->shadow[bounce], (uint32_t)__popcll(m), lane);
      if (shadow) {
        const uint32_t j = b + wave_prefix(m);
        sq.o[j] = vec4{out.shadow_o.x, out.shadow_o.y, out.shadow_o.z, out.shadow_tmax};
        sq.d[j] = vec4{out.shadow_d.x, out.shadow_d.y, out.shadow_d.z, u2f(pid)};
        sq.contrib[j] = vec4{out.shadow_contrib.x, out.shadow_contrib.y, out.shadow_contrib.z, 0.0f};
      }
    }
  }
  const uint32_t n = wave_sum(shaded);
  if (lane == 0 && n) atomicAdd(&ctr->shaded, n);
}

// ---- shadow (any hit) --------------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ void __launch_bounds__(kBlock) k_trace_shadow(DeviceScene S, ShadowQueue sq, vec4* __restrict__ Lbuf,
                                                          BatchCounters* __restrict__ ctr, uint32_t bounce,
                                                          uint32_t* __restrict__ spill, uint32_t refill_threshold) {
  __shared__ uint32_t lds_stack[kLdsStack][kBlock];
  const uint32_t lane = wave_lane();
  WaveQueue q{&ctr->work[3 * bounce + 2], ctr->shadow[bounce], false};
  TraversalStack stack;
  stack.lds = &lds_stack[0][threadIdx.x];
  stack.lds_stride = kBlock;
  stack.spill = spill + ((size_t)blockIdx.x * kBlock + threadIdx.x);
  stack.spill_stride = gridDim.x * kBlock;
  TraversalCount tc;
  TravState ts;
  uint32_t ray = kInvalidRef;
  uint32_t pid = 0;

  auto finish = [&]() {
    if (ts.best.tri == kInvalidRef) {  // unoccluded: L += attenuation * Ld (kernel.metal:631-637)
      const vec4 c = sq.contrib[ray];
      vec4 L = Lbuf[pid];
      L.x += c.x; L.y += c.y; L.z += c.z;
      Lbuf[pid] = L;
    }
    ray = kInvalidRef;
  };

  for (;;) {
    const uint32_t fresh = wave_refill(q, ray == kInvalidRef, lane);
    if (fresh != kInvalidRef) {
      ray = fresh;
      const vec4 o4 = sq.o[ray];
      const vec4 d4 = sq.d[ray];
      pid = f2u(d4.w);
      if (trav_init(S, ts, v3(o4.x, o4.y, o4.z), v3(d4.x, d4.y, d4.z), 1e-3f, o4.w, stack, true, COUNT ? &tc : nullptr)) finish();
    }
    if (__ballot(ray != kInvalidRef) == 0) {
      if (q.exhausted) break;
      continue;
    }
    while (ray != kInvalidRef) {
      if (trav_step<true, COUNT>(S, ts, &tc)) finish();
      if (refill_threshold && !q.exhausted && (uint32_t)__popcll(__ballot(ray != kInvalidRef)) < refill_threshold) break;
    }
  }
  if (COUNT) {
    const uint32_t n = wave_sum(tc.nodes), t = wave_sum(tc.tris);
    if (lane == 0) {
      atomicAdd(&ctr->nodes_shadow, (unsigned long long)n);
      atomicAdd(&ctr->tris_shadow, (unsigned long long)t);
    }
  }
}

// ---- accumulate (kernel.metal:672-684): running mean, one sample at a time, in sample order --------------------------
__global__ void __launch_bounds__(kBlock) k_accumulate(vec4* __restrict__ acc, const vec4* __restrict__ Lbuf,
                                                        uint32_t npixels, uint32_t nsamples, uint32_t n0,
                                                        uint32_t nonfinite_policy, BatchCounters* __restrict__ ctr) {
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= npixels) return;
  vec4 a = acc[p];
  for (uint32_t s = 0; s < nsamples; s++) {
    const vec4 L4 = Lbuf[(size_t)s * npixels + p];
    vec3 L = v3(L4.x, L4.y, L4.z);
    if (!(fabsf(L.x) <= 3.0e38f && fabsf(L.y) <= 3.0e38f && fabsf(L.z) <= 3.0e38f)) {  // NaN or inf
      atomicAdd(&ctr->nonfinite, 1u);
      if (nonfinite_policy == PT_NONFINITE_ZERO) L = v3(0.0f);
    }
    const uint32_t localFrameIdx = n0 + s;
    if (localFrameIdx > 0) {
      L = L + v3(a.x, a.y, a.z) * (float)localFrameIdx;
      L = L / (float)(localFrameIdx + 1);
    }
    a = vec4{L.x, L.y, L.z, 1.0f};
  }
  acc[p] = a;
}

// ---- bookkeeping --------------------------------------------------------------------------------------------------------
__global__ void k_fold_counters(const BatchCounters* __restrict__ ctr, Totals* __restrict__ tot, uint32_t max_bounces,
                                uint32_t counted) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned long long closest = 0, shadow = 0;
  for (uint32_t b = 0; b < max_bounces; b++) {
    closest += ctr->active[b];
    shadow += ctr->shadow[b];
  }
  if (!counted) {
    tot->closest_rays += closest;
    tot->shadow_rays += shadow;
    tot->shaded_hits += ctr->shaded;
    tot->paths += ctr->active[0];
    tot->nonfinite += ctr->nonfinite;
  } else {  // instrumented sample (pt_measure_traversal): only feeds the per-ray fetch averages
    tot->nodes_closest += ctr->nodes_closest; tot->tris_closest += ctr->tris_closest;
    tot->nodes_shadow += ctr->nodes_shadow; tot->tris_shadow += ctr->tris_shadow;
    tot->counted_closest += closest; tot->counted_shadow += shadow;
  }
}

// primary-ray records for pt_trace_primary: queue order -> pixel order
__global__ void __launch_bounds__(kBlock) k_hit_records(DeviceScene S, PathState st, const vec4* __restrict__ hit,
                                                         const BatchCounters* __restrict__ ctr,
                                                         pt_hit_record* __restrict__ out) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= ctr->active[0]) return;
  const vec4 h = hit[i];
  const uint32_t tri = f2u(h.w);
  pt_hit_record r;
  if (tri != kInvalidRef) {
    r.t = h.x; r.u = h.y; r.v = h.z;
    r.instance = (int32_t)S.tris[tri].inst;
    r.primitive = (int32_t)S.tris[tri].prim;
  } else {
    r.t = 0.0f; r.u = 0.0f; r.v = 0.0f; r.instance = -1; r.primitive = -1;
  }
  out[st.pid[i]] = r;
}

// ---- launchers (host) ---------------------------------------------------------------------------------------------------
void launch_raygen(hipStream_t s, const DeviceScene& S, PathState st, vec4* Lbuf, BatchCounters* ctr, uint32_t first_sample,
                   uint32_t nsamples) {
  const uint32_t tilesX = (S.width + 7) / 8, tilesY = (S.height + 7) / 8;
  const uint64_t threads = (uint64_t)nsamples * tilesX * tilesY * 64;
  const uint32_t grid = (uint32_t)((threads + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_raygen, dim3(grid), dim3(kBlock), 0, s, S, st, Lbuf, ctr, first_sample, nsamples, tilesX, tilesY);
}
void launch_trace_closest(hipStream_t s, uint32_t grid, const DeviceScene& S, PathState st, vec4* hit, BatchCounters* ctr,
                          uint32_t bounce, uint32_t* spill, int32_t* hitlog, uint32_t log_stride, bool count, uint32_t refill) {
  if (count)
    hipLaunchKernelGGL(k_trace_closest<true>, dim3(grid), dim3(kBlock), 0, s, S, st, hit, ctr, bounce, spill, hitlog, log_stride, refill);
  else
    hipLaunchKernelGGL(k_trace_closest<false>, dim3(grid), dim3(kBlock), 0, s, S, st, hit, ctr, bounce, spill, hitlog, log_stride, refill);
}
void launch_shade(hipStream_t s, uint32_t grid, const DeviceScene& S, PathState sin, PathState sout, const vec4* hit,
                  ShadowQueue sq, vec4* Lbuf, BatchCounters* ctr, uint32_t bounce) {
  hipLaunchKernelGGL(k_shade, dim3(grid), dim3(kBlock), 0, s, S, sin, sout, hit, sq, Lbuf, ctr, bounce);
}
void launch_trace_shadow(hipStream_t s, uint32_t grid, const DeviceScene& S, ShadowQueue sq, vec4* Lbuf, BatchCounters* ctr,
                         uint32_t bounce, uint32_t* spill, bool count, uint32_t refill) {
  if (count)
    hipLaunchKernelGGL(k_trace_shadow<true>, dim3(grid), dim3(kBlock), 0, s, S, sq, Lbuf, ctr, bounce, spill, refill);
  else
    hipLaunchKernelGGL(k_trace_shadow<false>, dim3(grid), dim3(kBlock), 0, s, S, sq, Lbuf, ctr, bounce, spill, refill);
}
void launch_accumulate(hipStream_t s, vec4* acc, const vec4* Lbuf, uint32_t npixels, uint32_t nsamples, uint32_t n0,
                       uint32_t nonfinite_policy, BatchCounters* ctr) {
  hipLaunchKernelGGL(k_accumulate, dim3((npixels + kBlock - 1) / kBlock), dim3(kBlock), 0, s, acc, Lbuf, npixels, nsamples, n0,
                     nonfinite_policy, ctr);
}
void launch_fold_counters(hipStream_t s, const BatchCounters* ctr, Totals* tot, uint32_t max_bounces, bool counted) {
  hipLaunchKernelGGL(k_fold_counters, dim3(1), dim3(64), 0, s, ctr, tot, max_bounces, counted ? 1u : 0u);
}
void launch_hit_records(hipStream_t s, const DeviceScene& S, PathState st, const vec4* hit, const BatchCounters* ctr,
                        pt_hit_record* out, uint32_t npixels) {
  hipLaunchKernelGGL(k_hit_records, dim3((npixels + kBlock - 1) / kBlock), dim3(kBlock), 0, s, S, st, hit, ctr, out);
}

}  // namespace pt
