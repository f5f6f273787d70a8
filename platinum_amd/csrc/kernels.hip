// kernels.hip — the gfx950 wavefront path-tracing kernels.
//
// One batch = `nsamples` samples of every pixel traced together.  Per bounce b:
//     k_trace_closest(b)  ->  k_shade(b)  ->  k_trace_shadow(b)
// preceded by k_raygen and followed by k_accumulate.  All kernels are stream-ordered; queue sizes never visit the host.
//
// Queues are WAVE-PRIVATE SEGMENTS.  Every kernel runs the same grid (Segments::nwaves waves); wave w owns the slots
// {seg_slot(w, r) : r < seg_cap} of every queue array (chunk-interleaved, see seg_slot) and a count per queue.  Raygen deals 8x8 pixel tiles to the waves
// round-robin; k_shade shades a wave's own segment and compacts the survivors (ballot + mbcnt prefix) into its own
// segment of the other state buffer, and its NEE rays into its own shadow segment.  Survivors <= inputs, so a segment
// never overflows and NO atomic sits on the producer side.  (Round-1 measurement: with one global append counter per
// queue, raygen and shade were pinned at the ~88 returning atomics/us one L2 address sustains — MI355X_MICROARCH.md
// "dequeue"; raygen got 6.7x faster without it.)
// The trace kernels consume DENSE CHUNK TABLES: after every producer, k_chunk_tables lists the non-empty 64-entry
// chunks of all segments, wave-major (wave 0's chunks, wave 1's, ...), and the trace waves claim runs of that list
// from one cursor (kClaim chunks per atomic).  A wave owns a few ADJACENT 8x8 pixel tiles under all samples of the
// batch, so the rays in flight at any moment come from a small neighbourhood of the image — the BVH subtrees they
// touch stay in L2 — there are no empty chunks to sweep, and a chunk of survivors still comes from one or two tiles.
// Results are written in place (hit[i], Lbuf[pid]), so it does not matter which wave traces a ray.
// Statistics are kept per wave (no atomics) and reduced by k_fold_counters.
//
// Compiled with -ffp-contract=off (deterministic fp32 contract, pt_math.h).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "pt_post.h"
#include "pt_shade.h"

namespace pt {

// ---- wave helpers (wave64) -------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t wave_prefix(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ uint32_t wave_index() { return blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); }
// Slot of entry r of wave w's segment.  Segments are CHUNK-INTERLEAVED: the k-th 64-entry chunk of every wave is stored
// back to back ((k * nwaves + w) * 64), so the trace kernels, which walk the chunks in exactly that order, stream
// through contiguous memory like a dense queue (a wave-major layout cost the closest-hit kernel 1.4x: every chunk
// then starts a new 44 KB-strided region).
__device__ __forceinline__ uint32_t seg_slot(uint32_t nwaves, uint32_t w, uint32_t r) { return ((r >> 6) * nwaves + w) * 64u + (r & 63u); }

// ---- chunk claims for the trace kernels ------------------------------------------------------------------------------
constexpr uint32_t kClaim = 2;  // 64-ray chunks claimed per cursor atomic

struct ChunkClaims {
  const uint32_t* __restrict__ table;   // [total] (k << 16) | w  for every non-empty chunk, wave-major
  const uint32_t* __restrict__ counts;  // [nwaves] rays per segment
  uint32_t* cursor;
  uint32_t nwaves, total, lane;
  uint32_t next_c, end_c;
  __device__ __forceinline__ void init(const uint32_t* tab, uint32_t total_, const uint32_t* c, uint32_t* cur, const Segments& seg,
                                       uint32_t lane_) {
    table = tab; total = total_; counts = c; cursor = cur; nwaves = seg.nwaves; lane = lane_;
    next_c = end_c = 0;
  }
  // ---- per-lane ray replacement ----
  // The wave keeps a pool = the not yet started rays [pool_next, pool_end) of its current chunk.  Lanes whose ray has
  // finished (need == true) take the next pool entries in lane order; an empty pool is refilled from the claim list.
  // Wave-uniform call. Returns kInvalidRef for lanes that got nothing; `exhausted` = no chunk left anywhere.
  uint32_t pool_next = 0, pool_end = 0;
  bool exhausted = false;
  __device__ __forceinline__ uint32_t take(bool need) {
    uint32_t got = kInvalidRef;
    for (;;) {
      const unsigned long long m = __ballot(need && got == kInvalidRef);
      if (m == 0) return got;
      if (pool_next == pool_end) {
        if (exhausted) return got;
        if (next_c == end_c) {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(cursor, kClaim);
          base = __builtin_amdgcn_readfirstlane(base);
          if (base >= total) { exhausted = true; return got; }
          next_c = base;
          end_c = base + kClaim < total ? base + kClaim : total;
        }
        const uint32_t e = table[next_c++];
        const uint32_t wv = e & 0xffffu, k = e >> 16;
        const uint32_t n = counts[wv];
        const uint32_t left = n - k * 64u;  // > 0: the table lists non-empty chunks only
        pool_next = seg_slot(nwaves, wv, k * 64u);
        pool_end = pool_next + (left < 64u ? left : 64u);
      }
      if (need && got == kInvalidRef) {
        const uint32_t idx = pool_next + wave_prefix(m);
        if (idx < pool_end) got = idx;
      }
      const uint32_t avail = pool_end - pool_next, want = (uint32_t)__popcll(m);
      pool_next += want < avail ? want : avail;
    }
  }

  // Wave-uniform. Returns this lane's queue index, kInvalidRef for an idle lane; `done` when the list is exhausted.
  __device__ __forceinline__ uint32_t next(bool& done) {
    done = false;
    if (next_c == end_c) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(cursor, kClaim);
      base = __builtin_amdgcn_readfirstlane(base);
      if (base >= total) { done = true; return kInvalidRef; }
      next_c = base;
      end_c = base + kClaim < total ? base + kClaim : total;
    }
    const uint32_t e = table[next_c++];
    const uint32_t wv = e & 0xffffu, k = e >> 16;
    const uint32_t n = counts[wv];
    const uint32_t r = k * 64u + lane;
    return r < n ? seg_slot(nwaves, wv, r) : kInvalidRef;
  }
};

// Lists the non-empty chunks of every segment, wave-major.  blockIdx.y = 0: the closest-hit queue (state buffer `cur`);
// 1: the shadow queue.  kTableBlocks blocks per list, each owning a contiguous slice of the producer waves: a block first sums
// the chunk counts of all waves before its slice (its base offset), scans its own slice in LDS, then all of its threads write
// the slice's entries cooperatively (entry -> wave by binary search in the LDS prefix), so the stores are coalesced.  (One block
// per list took 0.18 ms per call at 64 samples in flight — 2.5 % of a C2 step.)
constexpr uint32_t kTableBlocks = 32;
__global__ void __launch_bounds__(1024) k_chunk_tables(Segments seg, uint32_t cur, BatchCounters* __restrict__ ctr,
                                                        uint32_t bounce_closest, uint32_t bounce_shadow, uint32_t do_shadow) {
  const bool sh = blockIdx.y == 1;
  if (sh && !do_shadow) return;
  const uint32_t* __restrict__ counts = sh ? seg.shadow : seg.active[cur];
  uint32_t* __restrict__ table = sh ? seg.table_shadow : seg.table_closest;
  __shared__ uint32_t part[1024];
  __shared__ uint32_t base_sh;
  const uint32_t slice = (seg.nwaves + kTableBlocks - 1) / kTableBlocks;  // waves per block (<= 1024 for nwaves <= 32768)
  const uint32_t w_begin = blockIdx.x * slice;
  const uint32_t w_end = w_begin + slice < seg.nwaves ? w_begin + slice : seg.nwaves;
  // base = chunks of all waves before this slice; the last block also learns the grand total
  uint32_t before = 0, total = 0;
  for (uint32_t w = threadIdx.x; w < seg.nwaves; w += 1024) {
    const uint32_t nc = (counts[w] + 63u) / 64u;
    total += nc;
    if (w < w_begin) before += nc;
  }
  part[threadIdx.x] = before;
  __syncthreads();
  for (uint32_t off = 512; off > 0; off >>= 1) { if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off]; __syncthreads(); }
  if (threadIdx.x == 0) base_sh = part[0];
  __syncthreads();
  if (blockIdx.x == kTableBlocks - 1) {
    part[threadIdx.x] = total;
    __syncthreads();
    for (uint32_t off = 512; off > 0; off >>= 1) { if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0) {
      if (sh) ctr->chunks_shadow[bounce_shadow] = part[0];
      else ctr->chunks_closest[bounce_closest] = part[0];
    }
    __syncthreads();
  }
  // inclusive scan of this slice's per-wave chunk counts (thread t <-> wave w_begin + t)
  const uint32_t mine = (w_begin + threadIdx.x < w_end) ? (counts[w_begin + threadIdx.x] + 63u) / 64u : 0u;
  part[threadIdx.x] = mine;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {  // Hillis-Steele
    const uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  const uint32_t n_entries = part[1023];
  const uint32_t base = base_sh;
  for (uint32_t e = threadIdx.x; e < n_entries; e += 1024) {
    // first t with inclusive prefix part[t] > e
    uint32_t lo = 0, hi = 1023;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (part[mid] > e) hi = mid; else lo = mid + 1; }
    const uint32_t k = e - (lo ? part[lo - 1] : 0u);
    table[base + e] = (k << 16) | (w_begin + lo);
  }
}

// ---- raygen ------------------------------------------------------------------------------------------------------
// One 8x8 pixel tile of one sample per wave iteration (a coherent camera-ray bundle).  Lanes outside the image
// (partial edge tiles) are squeezed out.
__global__ void __launch_bounds__(kBlock) k_raygen(DeviceScene S, PathState st, vec4* __restrict__ Lbuf, Segments seg,
                                                    BatchCounters* __restrict__ ctr, uint32_t first_sample,
                                                    uint32_t nsamples, uint32_t tilesX, uint32_t tilesY) {
  const uint32_t lane = wave_lane();
  const uint32_t w = wave_index();
  const uint32_t tiles = tilesX * tilesY;
  // Wave w owns tiles w, w + nwaves, w + 2 nwaves, ... under all samples of the batch (tile-major, sample-minor): its
  // consecutive chunks are the same 8x8 pixels under successive samples, so a chunk of survivors still comes from one
  // or two image tiles, and because the tiles of one wave are spread over the image every wave carries a statistically
  // similar load through k_shade (which is static per wave).
  const uint32_t per_wave = (tiles + seg.nwaves - 1) / seg.nwaves;
  uint32_t n_out = 0;
  for (uint32_t k = 0; k < per_wave * nsamples; k++) {
    const uint32_t tile = seg.tile_contiguous ? w * per_wave + k / nsamples : (k / nsamples) * seg.nwaves + w;
    const uint32_t s = k % nsamples;
    if (tile >= tiles) break;  // wave-uniform
    const uint32_t ty = tile / tilesX;
    const uint32_t x = (tile - ty * tilesX) * 8 + (lane & 7);
    const uint32_t y = ty * 8 + (lane >> 3);
    const bool valid = x < S.width && y < S.height;
    RayGenOut rg;
    if (valid) rg = stage_raygen(S, x, y, first_sample + s);
    const unsigned long long m = __ballot(valid);
    if (valid) {
      const uint32_t j = seg_slot(seg.nwaves, w, n_out + wave_prefix(m));
      const uint32_t pid = s * (S.width * S.height) + y * S.width + x;
      st.rayO[j] = vec4{rg.o.x, rg.o.y, rg.o.z, 0.0f};
      st.rayD[j] = vec4{rg.d.x, rg.d.y, rg.d.z, u2f(rg.dim & kMetaDimMask)};
      st.att[j] = vec4{1.0f, 1.0f, 1.0f, u2f(rg.offset)};
      st.pid[j] = pid;
      Lbuf[pid] = vec4{0.0f, 0.0f, 0.0f, 1.0f};
    }
    n_out += (uint32_t)__popcll(m);
  }
  if (lane == 0) {
    seg.active[0][w] = n_out;
    WaveStats& ws = seg.stats[w];
    ws.paths += n_out;
    ws.closest += n_out;
  }
}

// ---- wave-cooperative traversal step ---------------------------------------------------------------------------------
// One iteration for the lanes of a wave that still hold a ray: (1) lanes with a node to visit and room in their leaf
// queue visit it (trav_node: slab tests, candidate leaves queued, next node chosen); (2) the wave votes — when at least
// half of these lanes have a queued leaf, or a lane cannot go on without emptying its queue (queue full, or node stack
// exhausted with nobody else able to advance), every lane with a queued leaf runs ONE triangle test.  The triangle code
// therefore executes with most lanes busy instead of whenever a single lane met a leaf.  A ray is finished when its node
// stack is exhausted and its queue is empty (any-hit: on the first accepted hit).
template <bool ANY, bool COUNT>
__device__ __forceinline__ void wave_traverse(const DeviceScene& S, TravState& ts, TraversalCount* tc) {
  if (ts.cur != kInvalidRef && ts.st.npend <= kPendLeaves - 4) trav_node<COUNT>(S, ts, tc);
  const bool pending = ts.st.npend > 0;
  const bool stuck = pending && (ts.cur == kInvalidRef || ts.st.npend > kPendLeaves - 4);
  const bool advancing = ts.cur != kInvalidRef && !stuck;
  const unsigned long long mp = __ballot(pending);
  if (mp == 0) return;
  // triangle round when half of the lanes holding a ray have a queued leaf, a lane's queue is full, or nobody can advance
  const bool go = 2 * __popcll(mp) >= __popcll(__ballot(1)) || __ballot(ts.st.npend > kPendLeaves - 4) != 0 || __ballot(advancing) == 0;
  if (go && pending) {
    if (trav_pending_leaf<ANY, COUNT>(S, ts, tc)) { ts.cur = kInvalidRef; ts.st.npend = 0; ts.st.sp = 0; }
  }
}

// ---- closest hit ---------------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ void __launch_bounds__(kBlock, PT_TRACE_WAVES) k_trace_closest(DeviceScene S, PathState st, vec4* __restrict__ hit, Segments seg,
                                                           uint32_t cur, BatchCounters* __restrict__ ctr, uint32_t bounce,
                                                           uint32_t* __restrict__ spill, int32_t* __restrict__ hitlog,
                                                           uint32_t log_stride) {
  __shared__ uint32_t lds_stack[kLdsStack + 1][kBlock];
  __shared__ uint32_t lds_pend[kPendLeaves + 1][kBlock];
  const uint32_t lane = wave_lane();
  ChunkClaims src;
  src.init(seg.table_closest, ctr->chunks_closest[bounce], seg.active[cur], &ctr->work_closest[bounce], seg, lane);
  TraversalStack stack;
  stack.lds = &lds_stack[0][threadIdx.x];
  stack.pend = &lds_pend[0][threadIdx.x];
  stack.lds_stride = kBlock;
  stack.spill = spill + ((size_t)blockIdx.x * kBlock + threadIdx.x);
  stack.spill_stride = gridDim.x * kBlock;
  TraversalCount tc;

  // (written as an explicit init / step loop: with the whole traversal behind one call the compiler produced a kernel
  //  1.5x slower on secondary rays)
  TravState ts;
  uint32_t ray = kInvalidRef;
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
  const uint32_t refill = seg.refill_threshold;
  for (;;) {
    const uint32_t fresh = src.take(ray == kInvalidRef);
    if (fresh != kInvalidRef) {
      ray = fresh;
      const vec4 o4 = st.rayO[ray];
      const vec4 d4 = st.rayD[ray];
      float ir = 0.0f;  // alpha-test payload: the sample drawn before `intersect` (kernel.metal:510), only needed with cut-outs
      if (S.has_alpha) ir = Halton{S.halton, f2u(st.att[ray].w), f2u(d4.w) & kMetaDimMask}.sample1d();
      if (trav_init(S, ts, v3(o4.x, o4.y, o4.z), v3(d4.x, d4.y, d4.z), 1e-3f, kInf, ir, stack, false, COUNT ? &tc : nullptr)) finish();
    }
    if (__ballot(ray != kInvalidRef) == 0) {
      if (src.exhausted && src.pool_next == src.pool_end) break;
      continue;
    }
    while (ray != kInvalidRef) {
      wave_traverse<false, COUNT>(S, ts, &tc);
      if (ts.cur == kInvalidRef && ts.st.npend == 0) finish();
      // lanes still in this loop vote: leave for a refill once the wave has emptied below the threshold
      if (refill && !src.exhausted && (uint32_t)__popcll(__ballot(ray != kInvalidRef)) < refill) break;
    }
  }
  if (COUNT) {
    const uint32_t nn = wave_sum(tc.nodes), t = wave_sum(tc.tris);
    if (lane == 0) {
      atomicAdd(&ctr->nodes_closest, (unsigned long long)nn);
      atomicAdd(&ctr->tris_closest, (unsigned long long)t);
    }
  }
}

// ---- shade -----------------------------------------------------------------------------------------------------------
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 2
#endif
__global__ void __launch_bounds__(kBlock, PT_SHADE_WAVES) k_shade(const DeviceScene* __restrict__ Sp, PathState sin, PathState sout,
                                                                   const vec4* __restrict__ hit, ShadowQueue sq,
                                                                   vec4* __restrict__ Lbuf, Segments seg, uint32_t cur,
                                                                   BatchCounters* __restrict__ ctr, uint32_t bounce) {
  const DeviceScene& S = *Sp;  // scene table read through the scalar cache: by value it cost 100 spilled SGPRs here
  const uint32_t lane = wave_lane();
  const uint32_t w = wave_index();
  const uint32_t n = seg.active[cur][w];
  uint32_t n_out = 0, n_shadow = 0, shaded = 0;

  for (uint32_t k0 = 0; k0 < n; k0 += 64) {  // wave-uniform trip count: every lane reaches the ballots
    const uint32_t k = k0 + lane;
    const uint32_t i = seg_slot(seg.nwaves, w, k);
    bool alive = false, shadow = false;
    ShadeOut out;
    uint32_t pid = 0, offset = 0;
    if (k < n) {
      const vec4 h4 = hit[i];
      const uint32_t tri = f2u(h4.w);
      if (tri == kInvalidRef) {
        // a miss ends the path; it adds the environment's radiance if there is one (kernel.metal:517-539), then
        // attenuation * backgroundColor (= 0, defs.metal:21)
        if (S.env_texture >= 0) {
          const vec4 o4 = sin.rayO[i];
          const vec4 d4 = sin.rayD[i];
          const vec4 a4 = sin.att[i];
          const vec3 Le = stage_miss(S, v3(d4.x, d4.y, d4.z), v3(a4.x, a4.y, a4.z), bounce, o4.w, (f2u(d4.w) & kMetaSpecular) != 0);
          const uint32_t mpid = sin.pid[i];
          vec4 L = Lbuf[mpid];
          L.x += Le.x; L.y += Le.y; L.z += Le.z;
          Lbuf[mpid] = L;
        }
      } else {
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
    // survivors -> this wave's segment of the next bounce's queue
    {
      const unsigned long long m = __ballot(alive);
      if (alive) {
        const uint32_t j = seg_slot(seg.nwaves, w, n_out + wave_prefix(m));
        sout.rayO[j] = vec4{out.next_o.x, out.next_o.y, out.next_o.z, out.next_pdf};
        sout.rayD[j] = vec4{out.next_d.x, out.next_d.y, out.next_d.z,
                            u2f((out.dim & kMetaDimMask) | (out.next_specular ? kMetaSpecular : 0u))};
        sout.att[j] = vec4{out.next_att.x, out.next_att.y, out.next_att.z, u2f(offset)};
        sout.pid[j] = pid;
      }
      n_out += (uint32_t)__popcll(m);
    }
    // NEE shadow rays -> this wave's shadow segment
    {
      const unsigned long long m = __ballot(shadow);
      if (shadow) {
        const uint32_t j = seg_slot(seg.nwaves, w, n_shadow + wave_prefix(m));
        sq.o[j] = vec4{out.shadow_o.x, out.shadow_o.y, out.shadow_o.z, out.shadow_tmax};
        sq.d[j] = vec4{out.shadow_d.x, out.shadow_d.y, out.shadow_d.z, u2f(pid)};
        sq.contrib[j] = vec4{out.shadow_contrib.x, out.shadow_contrib.y, out.shadow_contrib.z, out.shadow_payload};
      }
      n_shadow += (uint32_t)__popcll(m);
    }
  }
  const uint32_t ns = wave_sum(shaded);
  if (lane == 0) {
    seg.active[cur ^ 1][w] = n_out;
    seg.shadow[w] = n_shadow;
    WaveStats& ws = seg.stats[w];
    ws.closest += n_out;
    ws.shadow += n_shadow;
    ws.shaded += ns;
  }
}

// ---- shadow (any hit) --------------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ void __launch_bounds__(kBlock, PT_TRACE_WAVES) k_trace_shadow(DeviceScene S, ShadowQueue sq, vec4* __restrict__ Lbuf, Segments seg,
                                                          BatchCounters* __restrict__ ctr, uint32_t bounce,
                                                          uint32_t* __restrict__ spill) {
  __shared__ uint32_t lds_stack[kLdsStack + 1][kBlock];
  __shared__ uint32_t lds_pend[kPendLeaves + 1][kBlock];
  const uint32_t lane = wave_lane();
  ChunkClaims src;
  src.init(seg.table_shadow, ctr->chunks_shadow[bounce], seg.shadow, &ctr->work_shadow[bounce], seg, lane);
  TraversalStack stack;
  stack.lds = &lds_stack[0][threadIdx.x];
  stack.pend = &lds_pend[0][threadIdx.x];
  stack.lds_stride = kBlock;
  stack.spill = spill + ((size_t)blockIdx.x * kBlock + threadIdx.x);
  stack.spill_stride = gridDim.x * kBlock;
  TraversalCount tc;

  // (written as an explicit init / step loop: with the whole traversal behind one call the compiler produced a kernel
  //  1.5x slower on secondary rays)
  TravState ts;
  uint32_t ray = kInvalidRef, pid = 0;
  auto finish = [&]() {
    if (ts.best.tri == kInvalidRef) {  // unoccluded: L += attenuation * Ld (kernel.metal:631-637)
      const vec4 c = sq.contrib[ray];
      vec4 L = Lbuf[pid];
      L.x += c.x; L.y += c.y; L.z += c.z;
      Lbuf[pid] = L;
    }
    ray = kInvalidRef;
  };
  const uint32_t refill = seg.refill_threshold;
  for (;;) {
    const uint32_t fresh = src.take(ray == kInvalidRef);
    if (fresh != kInvalidRef) {
      ray = fresh;
      const vec4 o4 = sq.o[ray];
      const vec4 d4 = sq.d[ray];
      pid = f2u(d4.w);
      const float ir = S.has_alpha ? sq.contrib[ray].w : 0.0f;  // kernel.metal:625
      if (trav_init(S, ts, v3(o4.x, o4.y, o4.z), v3(d4.x, d4.y, d4.z), 1e-3f, o4.w, ir, stack, true, COUNT ? &tc : nullptr)) finish();
    }
    if (__ballot(ray != kInvalidRef) == 0) {
      if (src.exhausted && src.pool_next == src.pool_end) break;
      continue;
    }
    while (ray != kInvalidRef) {
      wave_traverse<true, COUNT>(S, ts, &tc);
      if (ts.cur == kInvalidRef && ts.st.npend == 0) finish();
      if (refill && !src.exhausted && (uint32_t)__popcll(__ballot(ray != kInvalidRef)) < refill) break;
    }
  }
  if (COUNT) {
    const uint32_t nn = wave_sum(tc.nodes), t = wave_sum(tc.tris);
    if (lane == 0) {
      atomicAdd(&ctr->nodes_shadow, (unsigned long long)nn);
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

// ---- GMoN ---------------------------------------------------------------------------------------------
// Accumulate into the bucket images exactly as the reference does with RendererFlags_GMoN: sample f goes to bucket
// f / ceil(spp / buckets) (renderer_pt.cpp:124-139) with the running-mean weight n = f / gmonBuckets
// (kernel.metal:675-681 — note: NOT the sample's index inside its bucket; reproduced as is).
__global__ void __launch_bounds__(kBlock) k_accumulate_gmon(vec4* __restrict__ buckets, const vec4* __restrict__ Lbuf,
                                                             uint32_t npixels, uint32_t nsamples, uint32_t n0,
                                                             uint32_t samples_per_bucket, uint32_t gmon_buckets,
                                                             uint32_t nonfinite_policy, BatchCounters* __restrict__ ctr) {
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= npixels) return;
  for (uint32_t s = 0; s < nsamples; s++) {
    const vec4 L4 = Lbuf[(size_t)s * npixels + p];
    vec3 L = v3(L4.x, L4.y, L4.z);
    if (!(fabsf(L.x) <= 3.0e38f && fabsf(L.y) <= 3.0e38f && fabsf(L.z) <= 3.0e38f)) {
      atomicAdd(&ctr->nonfinite, 1u);
      if (nonfinite_policy == PT_NONFINITE_ZERO) L = v3(0.0f);
    }
    const uint32_t f = n0 + s;
    vec4* acc = buckets + (size_t)(f / samples_per_bucket) * npixels;
    const uint32_t localFrameIdx = f / gmon_buckets;
    if (localFrameIdx > 0) {
      const vec4 a = acc[p];
      L = L + v3(a.x, a.y, a.z) * (float)localFrameIdx;
      L = L / (float)(localFrameIdx + 1);
    }
    acc[p] = vec4{L.x, L.y, L.z, 1.0f};
  }
}

// shaders/gmon.metal:14-55: per pixel, sort the bucket means by luma (stable bubble sort), Gini coefficient of the
// sorted lumas, drop c = int(min(G, cap) * (n / 2)) values from both ends, average the rest.
__global__ void __launch_bounds__(kBlock) k_gmon(vec4* __restrict__ acc, const vec4* __restrict__ buckets, uint32_t npixels,
                                                  uint32_t nBuckets, float cap) {
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= npixels) return;
  const vec3 lw = v3(0.2126f, 0.7152f, 0.0722f);  // gmon.metal:10
  vec3 values[32];                                // maxBuckets (gmon.metal:12)
  for (uint32_t i = 0; i < nBuckets; i++) {
    const vec4 b = buckets[(size_t)i * npixels + p];
    values[i] = v3(b.x, b.y, b.z);
  }
  for (uint32_t i = nBuckets; i > 1; i--)
    for (uint32_t j = 1; j < i; j++)
      if (dot(values[j], lw) < dot(values[j - 1], lw)) {
        const vec3 temp = values[j - 1];
        values[j - 1] = values[j];
        values[j] = temp;
      }
  vec3 sum = v3(0.0f), weightedSum = v3(0.0f);
  for (uint32_t i = 0; i < nBuckets; i++) {
    sum = sum + values[i];
    weightedSum = weightedSum + (float)(i + 1) * values[i];
  }
  float G = (2.0f * dot(weightedSum, lw)) / ((float)nBuckets * dot(sum, lw)) - (float)(nBuckets + 1) / (float)nBuckets;
  G = fminf(G, cap);
  // int(G * float(n / 2)); a NaN/negative G (all-black pixel with cap <= 0) keeps every bucket
  const float cf = G * (float)(nBuckets / 2);
  const int c = cf > 0.0f ? (int)cf : 0;
  sum = v3(0.0f);
  for (int i = c; i < (int)nBuckets - c; i++) sum = sum + values[i];
  const vec3 color = sum / (float)((int)nBuckets - 2 * c);
  acc[p] = vec4{color.x, color.y, color.z, 1.0f};
}

// ---- post-process + tonemap -> RGBA8 (pt_post.h) ------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_postprocess(const vec4* __restrict__ acc, uint32_t* __restrict__ rgba8, uint32_t W,
                                                         uint32_t H, PostConstants pc) {
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= W * H) return;
  const uint32_t py = p / W, px = p - py * W;
  rgba8[p] = pp_pack_rgba8(postprocess_pixel(acc, W, H, px, py, pc));
}

// ---- bookkeeping --------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_fold_counters(const BatchCounters* __restrict__ ctr, Totals* __restrict__ tot,
                                                           Segments seg, uint32_t counted) {
  // one block: reduce the per-wave statistics, then clear them for the next batch
  __shared__ unsigned long long red[4][kBlock];
  unsigned long long a[4] = {0, 0, 0, 0};
  for (uint32_t w = threadIdx.x; w < seg.nwaves; w += kBlock) {
    WaveStats& ws = seg.stats[w];
    a[0] += ws.closest; a[1] += ws.shadow; a[2] += ws.shaded; a[3] += ws.paths;
    ws.closest = 0; ws.shadow = 0; ws.shaded = 0; ws.paths = 0;
  }
  for (int k = 0; k < 4; k++) red[k][threadIdx.x] = a[k];
  __syncthreads();
  if (threadIdx.x != 0) return;
  unsigned long long t[4] = {0, 0, 0, 0};
  for (int k = 0; k < 4; k++)
    for (int i = 0; i < kBlock; i++) t[k] += red[k][i];
  if (!counted) {
    tot->closest_rays += t[0];
    tot->shadow_rays += t[1];
    tot->shaded_hits += t[2];
    tot->paths += t[3];
    tot->nonfinite += ctr->nonfinite;
  } else {  // instrumented sample (pt_measure_traversal): only feeds the per-ray fetch averages
    tot->nodes_closest += ctr->nodes_closest; tot->tris_closest += ctr->tris_closest;
    tot->nodes_shadow += ctr->nodes_shadow; tot->tris_shadow += ctr->tris_shadow;
    tot->counted_closest += t[0]; tot->counted_shadow += t[1];
  }
}

// one ShadeRec per flattened triangle, in tris[] order (once per render, after the BVH build)
__global__ void __launch_bounds__(kBlock) k_shade_records(DeviceScene S, ShadeRec* __restrict__ out) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= S.tri_count) return;
  out[t] = make_shade_rec(S, S.tris[t]);
}

// primary-ray records for pt_trace_primary: segment order -> pixel order
__global__ void __launch_bounds__(kBlock) k_hit_records(DeviceScene S, PathState st, const vec4* __restrict__ hit, Segments seg,
                                                         pt_hit_record* __restrict__ out) {
  const uint32_t lane = wave_lane();
  const uint32_t w = wave_index();
  const uint32_t n = seg.active[0][w];
  for (uint32_t k = lane; k < n; k += 64) {
    const uint32_t i = seg_slot(seg.nwaves, w, k);
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
}

// ---- launchers (host) ---------------------------------------------------------------------------------------------------
void launch_raygen(hipStream_t s, uint32_t grid, const DeviceScene& S, PathState st, vec4* Lbuf, Segments seg, BatchCounters* ctr,
                   uint32_t first_sample, uint32_t nsamples) {
  const uint32_t tilesX = (S.width + 7) / 8, tilesY = (S.height + 7) / 8;
  hipLaunchKernelGGL(k_raygen, dim3(grid), dim3(kBlock), 0, s, S, st, Lbuf, seg, ctr, first_sample, nsamples, tilesX, tilesY);
}
void launch_chunk_tables(hipStream_t s, Segments seg, uint32_t cur, BatchCounters* ctr, uint32_t bounce_closest,
                         uint32_t bounce_shadow, bool do_shadow) {
  hipLaunchKernelGGL(k_chunk_tables, dim3(kTableBlocks, 2), dim3(1024), 0, s, seg, cur, ctr, bounce_closest, bounce_shadow, do_shadow ? 1u : 0u);
}
void launch_trace_closest(hipStream_t s, uint32_t grid, const DeviceScene& S, PathState st, vec4* hit, Segments seg, uint32_t cur,
                          BatchCounters* ctr, uint32_t bounce, uint32_t* spill, int32_t* hitlog, uint32_t log_stride, bool count) {
  if (count)
    hipLaunchKernelGGL(k_trace_closest<true>, dim3(grid), dim3(kBlock), 0, s, S, st, hit, seg, cur, ctr, bounce, spill, hitlog, log_stride);
  else
    hipLaunchKernelGGL(k_trace_closest<false>, dim3(grid), dim3(kBlock), 0, s, S, st, hit, seg, cur, ctr, bounce, spill, hitlog, log_stride);
}
void launch_shade(hipStream_t s, uint32_t grid, const DeviceScene* S, PathState sin, PathState sout, const vec4* hit,
                  ShadowQueue sq, vec4* Lbuf, Segments seg, uint32_t cur, BatchCounters* ctr, uint32_t bounce) {
  hipLaunchKernelGGL(k_shade, dim3(grid), dim3(kBlock), 0, s, S, sin, sout, hit, sq, Lbuf, seg, cur, ctr, bounce);
}
void launch_trace_shadow(hipStream_t s, uint32_t grid, const DeviceScene& S, ShadowQueue sq, vec4* Lbuf, Segments seg,
                         BatchCounters* ctr, uint32_t bounce, uint32_t* spill, bool count) {
  if (count)
    hipLaunchKernelGGL(k_trace_shadow<true>, dim3(grid), dim3(kBlock), 0, s, S, sq, Lbuf, seg, ctr, bounce, spill);
  else
    hipLaunchKernelGGL(k_trace_shadow<false>, dim3(grid), dim3(kBlock), 0, s, S, sq, Lbuf, seg, ctr, bounce, spill);
}
void launch_accumulate(hipStream_t s, vec4* acc, const vec4* Lbuf, uint32_t npixels, uint32_t nsamples, uint32_t n0,
                       uint32_t nonfinite_policy, BatchCounters* ctr) {
  hipLaunchKernelGGL(k_accumulate, dim3((npixels + kBlock - 1) / kBlock), dim3(kBlock), 0, s, acc, Lbuf, npixels, nsamples, n0,
                     nonfinite_policy, ctr);
}
void launch_accumulate_gmon(hipStream_t s, vec4* buckets, const vec4* Lbuf, uint32_t npixels, uint32_t nsamples, uint32_t n0,
                            uint32_t samples_per_bucket, uint32_t gmon_buckets, uint32_t nonfinite_policy, BatchCounters* ctr) {
  hipLaunchKernelGGL(k_accumulate_gmon, dim3((npixels + kBlock - 1) / kBlock), dim3(kBlock), 0, s, buckets, Lbuf, npixels, nsamples, n0,
                     samples_per_bucket, gmon_buckets, nonfinite_policy, ctr);
}
void launch_gmon(hipStream_t s, vec4* acc, const vec4* buckets, uint32_t npixels, uint32_t nBuckets, float cap) {
  hipLaunchKernelGGL(k_gmon, dim3((npixels + kBlock - 1) / kBlock), dim3(kBlock), 0, s, acc, buckets, npixels, nBuckets, cap);
}
void launch_postprocess(hipStream_t s, const vec4* acc, uint32_t* rgba8, uint32_t W, uint32_t H, const PostConstants& pc) {
  hipLaunchKernelGGL(k_postprocess, dim3((W * H + kBlock - 1) / kBlock), dim3(kBlock), 0, s, acc, rgba8, W, H, pc);
}
void launch_fold_counters(hipStream_t s, const BatchCounters* ctr, Totals* tot, Segments seg, bool counted) {
  hipLaunchKernelGGL(k_fold_counters, dim3(1), dim3(kBlock), 0, s, ctr, tot, seg, counted ? 1u : 0u);
}
void launch_shade_records(hipStream_t s, const DeviceScene& S, ShadeRec* out) {
  if (S.tri_count) hipLaunchKernelGGL(k_shade_records, dim3((S.tri_count + kBlock - 1) / kBlock), dim3(kBlock), 0, s, S, out);
}
void launch_hit_records(hipStream_t s, uint32_t grid, const DeviceScene& S, PathState st, const vec4* hit, Segments seg,
                        pt_hit_record* out) {
  hipLaunchKernelGGL(k_hit_records, dim3(grid), dim3(kBlock), 0, s, S, st, hit, seg, out);
}

}  // namespace pt
