// kernels.hip — the gfx950 wavefront path-tracing kernels.
//
// One batch = `nsamples` samples of every pixel traced together.  Per bounce b:
//     k_trace_closest(b)  ->  k_shade(b)  ->  k_trace_shadow(b)
// preceded by k_raygen and followed by k_accumulate.  All kernels are stream-ordered; queue sizes never visit the host.
//
// Queues are SEGMENTS.  A segment is the queue share of a few 8x8 pixel tiles (Segments::tiles_per_seg, 1 unless the image
// has more than 32768 tiles) under all samples of the batch: segment s owns the slots {seg_slot(s, r) : r < seg_cap} of
// every queue array (interleaved in groups of 16 chunks, see seg_slot) and one count per queue.  A segment is always processed by ONE wave
// at a time, front to back: k_raygen fills it, k_shade shades it and compacts the survivors (ballot + mbcnt prefix) into
// the same segment of the other state buffer and its NEE rays into the same segment of the shadow queue.  Survivors <=
// inputs, so a segment never overflows and NO atomic sits on the producer side.  (Round-1 measurement: with one global
// append counter per queue, raygen and shade were pinned at the ~88 returning atomics/us one L2 address sustains —
// MI355X_MICROARCH.md "dequeue"; raygen got 6.7x faster without it.)  The producers are persistent grids: there are ~8x
// more segments than resident waves, so the grid can be sized for whatever occupancy a kernel's registers allow and the
// tile count never has to divide it.  k_raygen's waves take segments round-robin; k_shade's waves claim them LONGEST FIRST
// from a list k_chunk_tables sorts after every producer (r6: in plain order the chip idled 10-14 % of every k_shade launch
// waiting for the waves that had drawn a long segment last).
// The trace kernels consume DENSE CHUNK TABLES: after every producer, k_chunk_tables lists the non-empty 64-entry
// chunks of all segments, segment-major (= image tiles in raster order, each under its samples; an entry names its
// segment, chunk and ray count), and the trace waves claim runs of that list from one cursor (8 chunks per atomic, fewer
// towards the end of the list; a run's entries arrive in one load): the rays in flight at any moment come from a
// small neighbourhood of the image, there are no empty chunks to sweep, and a chunk of survivors still comes from one
// tile.  Results are written in place (hit[i], Lbuf[pid]), so it does not matter which wave traces a ray.  Statistics
// are kept per physical wave (no atomics) and reduced by k_fold_counters.
//
// Compiled with -ffp-contract=off (deterministic fp32 contract, pt_math.h).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "pt_post.h"
#include "pt_shade.h"
#include "queue_plan.h"

namespace pt {

// Queue entries are written once and read once.  k_shade moves them with non-temporal loads / stores so that they do not push the
// tables, vertices and LUT texels it gathers out of the caches (k_shade -2.4 % on C3 and C2; the same on the ray loads and hit stores
// of the trace kernels changed nothing on C3 and cost 2 % on C2).
__device__ __forceinline__ vec4 ld_stream(const vec4* p) {
  using f4 = __attribute__((ext_vector_type(4))) float;
  const f4 v = __builtin_nontemporal_load((const f4*)p);
  return vec4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ uint32_t ld_stream(const uint32_t* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_stream(vec4* p, vec4 v) {
  using f4 = __attribute__((ext_vector_type(4))) float;
  __builtin_nontemporal_store(f4{v.x, v.y, v.z, v.w}, (f4*)p);
}
__device__ __forceinline__ void st_stream(uint32_t* p, uint32_t v) { __builtin_nontemporal_store(v, p); }

// ---- wave helpers (wave64) -------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t wave_prefix(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// physical wave id / count of the launch (blockDim.x is a multiple of 64)
__device__ __forceinline__ uint32_t wave_index() { return blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); }
__device__ __forceinline__ uint32_t wave_count() { return gridDim.x * (blockDim.x >> 6); }
// Slot of entry r of segment s.  Segments are interleaved in GROUPS of PT_SEG_GROUP 64-entry chunks: chunks 16g .. 16g + 15 of a
// segment are contiguous (16 KB per array), group g of neighbouring segments follows — a segment's entries are a few long runs
// (the class-binned passes of k_shade gather from them; +1 % over single-chunk interleaving), while concurrently processed
// segments still start in different memory channels (a plain segment-major layout cost the closest-hit kernel 1.4x in round 1).
#ifndef PT_SEG_GROUP
#define PT_SEG_GROUP 16
#endif
__device__ __forceinline__ uint32_t seg_slot(uint32_t nseg, uint32_t s, uint32_t r) {
  const uint32_t k = r >> 6;
  return (((k / PT_SEG_GROUP) * nseg + s) * PT_SEG_GROUP + (k % PT_SEG_GROUP)) * 64u + (r & 63u);
}
uint32_t seg_group_chunks() { return PT_SEG_GROUP; }
static_assert(PT_SEG_GROUP == kSegGroupChunks, "queue_plan.h sizes the queue arrays for this interleaving");

#ifdef PT_TAIL_PROBE
#define PT_TAIL_BEGIN const unsigned long long tail_t0 = wall_clock64(); unsigned long long tail_take = 0, tail_setup = 0, tail_ta = 0;
#define PT_TAIL_T(acc) { const unsigned long long tb_ = wall_clock64(); acc += tb_ - tail_ta; tail_ta = tb_; }
#define PT_TAIL_MARK tail_ta = wall_clock64();
#define PT_TAIL_END(kind)                                                                                         \
  if (wave_lane() == 0 && bounce < 16u) {                                                                           \
    const unsigned long long t1 = wall_clock64();                                                                   \
    atomicMax(&ctr->tail_end_max[kind][bounce], t1); atomicAdd(&ctr->tail_end_sum[kind][bounce], t1);              \
    atomicMax(&ctr->tail_start_inv[kind][bounce], ~tail_t0); atomicAdd(&ctr->tail_waves[kind][bounce], 1ull);     \
    atomicAdd(&ctr->tail_take[kind][bounce], tail_take); atomicAdd(&ctr->tail_setup[kind][bounce], tail_setup);     \
    atomicAdd(&ctr->tail_busy[kind][bounce], t1 - tail_t0);                                                         \
  }
#else
#define PT_TAIL_BEGIN
#define PT_TAIL_END(kind)
#define PT_TAIL_T(acc)
#define PT_TAIL_MARK
#endif

// ---- chunk claims for the trace kernels ------------------------------------------------------------------------------
// 64-ray chunks claimed per cursor atomic.  ONE L2 address sustains ~88 returning atomics per microsecond
// (MI355X_MICROARCH.md "dequeue"); at 2 chunks per claim the C2 closest-hit kernel (10.6 Grays/s = 166 chunks/us) ran AT that
// limit and C3 at two thirds of it — 73 % of the C2 kernel's wave time was s_waitcnt.  Claims are therefore "guided": 8 chunks
// while plenty of work is left, then 4, 2, 1 towards the end of the list so that the last waves finish together
// (a fixed 8 costs C3 3 % in the tail; guided, C2 closest-hit went from 36.2 to 23.0 ms per step, shadow from 26.0 to 15.4).
constexpr uint32_t kClaimMax = 8;

// A chunk-table entry: segment (16 bits: nseg <= 32768), chunk within the segment (10 bits: seg_cap <= 65536), rays in the chunk - 1 (6 bits).
__device__ __forceinline__ uint32_t chunk_entry(uint32_t sg, uint32_t k, uint32_t rays) { return ((rays - 1u) << 26) | (k << 16) | sg; }
static_assert(kMaxSegments <= 65536 && kMaxSegmentSlots / 64 <= 1024, "chunk_entry packs 16 + 10 + 6 bits");

struct ChunkClaims {
  const uint32_t* __restrict__ table;   // [total + 8] chunk_entry(segment, chunk, rays) for every non-empty chunk, segment-major
  uint32_t* cursor;
  uint32_t nseg, total, lane;
  uint32_t next_c, end_c, run_c = 0;
  uint32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0, e6 = 0, e7 = 0;   // the entries of the run being consumed (wave-uniform: scalar registers)
  uint32_t claim_k = kClaimMax, nwaves_grid = 1;
  __device__ __forceinline__ void init(const uint32_t* tab, uint32_t total_, uint32_t* cur, const Segments& seg, uint32_t lane_) {
    table = tab; total = total_; cursor = cur; nseg = seg.nseg; lane = lane_;
    next_c = end_c = 0;
    nwaves_grid = gridDim.x * (blockDim.x >> 6);
    claim_k = guided(total);
  }
  __device__ __forceinline__ uint32_t guided(uint32_t remaining) const {
    return remaining > 32u * nwaves_grid ? kClaimMax : remaining > 8u * nwaves_grid ? 4u : remaining > 2u * nwaves_grid ? 2u : 1u;
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
          if (lane == 0) base = atomicAdd(cursor, claim_k);
          base = __builtin_amdgcn_readfirstlane(base);
          if (base >= total) { exhausted = true; return got; }
          next_c = run_c = base;
          end_c = base + claim_k < total ? base + claim_k : total;
          claim_k = guided(total - end_c);
          // the entries of the whole run in ONE load (lanes 0..7; the table is padded by kClaimMax entries): the wave waits for
          // the claim and for this once per run of up to 8 chunks — with one entry + one segment count fetched per chunk it spent 5 % (C3) to
          // 15 % (C2) of its time here (r6, -DPT_TAIL_PROBE)
          const int ev = (int)table[base + (lane & 7u)];   // (one register; lanes 0..7 hold the run)
          e0 = (uint32_t)__builtin_amdgcn_readlane(ev, 0); e1 = (uint32_t)__builtin_amdgcn_readlane(ev, 1); e2 = (uint32_t)__builtin_amdgcn_readlane(ev, 2);
          e3 = (uint32_t)__builtin_amdgcn_readlane(ev, 3); e4 = (uint32_t)__builtin_amdgcn_readlane(ev, 4); e5 = (uint32_t)__builtin_amdgcn_readlane(ev, 5);
          e6 = (uint32_t)__builtin_amdgcn_readlane(ev, 6); e7 = (uint32_t)__builtin_amdgcn_readlane(ev, 7);
        }
        const uint32_t at = next_c++ - run_c;
        const uint32_t e = at == 0u ? e0 : at == 1u ? e1 : at == 2u ? e2 : at == 3u ? e3 : at == 4u ? e4 : at == 5u ? e5 : at == 6u ? e6 : e7;
        pool_next = seg_slot(nseg, e & 0xffffu, ((e >> 16) & 0x3ffu) * 64u);
        pool_end = pool_next + (e >> 26) + 1u;
      }
      if (need && got == kInvalidRef) {
        const uint32_t idx = pool_next + wave_prefix(m);
        if (idx < pool_end) got = idx;
      }
      const uint32_t avail = pool_end - pool_next, want = (uint32_t)__popcll(m);
      pool_next += want < avail ? want : avail;
    }
  }
};

// Lists the non-empty chunks of every segment, segment-major.  blockIdx.y = 0: the closest-hit queue (state buffer `cur`);
// 1: the shadow queue.  kTableBlocks blocks per list, each owning a contiguous slice of the segments: a block first sums
// the chunk counts of all segments before its slice (its base offset), scans its own slice in LDS, then all of its threads write
// the slice's entries cooperatively (entry -> segment by binary search in the LDS prefix), so the stores are coalesced.  (One block
// per list took 0.18 ms per call at 64 samples in flight — 2.5 % of a C2 step.)
constexpr uint32_t kTableBlocks = 32;
__global__ void __launch_bounds__(1024) k_chunk_tables(Segments seg, uint32_t cur, BatchCounters* __restrict__ ctr,
                                                        uint32_t bounce_closest, uint32_t bounce_shadow, uint32_t do_shadow) {
  if (blockIdx.y == 2) {
    // ---- the order in which k_shade's waves take the segments of the closest-hit queue: LONGEST FIRST.  A segment is one wave's sequential work
    //      there; with the plain segment order the chip waited 10 % (C3) to 14 % (C2) of every k_shade launch — 15-29 % from bounce 3 on, where the
    //      segments differ by an order of magnitude — for the waves that had drawn a long segment last (r6, -DPT_TAIL_PROBE).  Key: the time k_shade
    //      measured for this segment at this bounce in the PREVIOUS batch of the render (same camera, same scene: the same work), else the
    //      segment's size.  A counting sort over 240 logarithmic classes (5 bits of exponent, 3 of mantissa); the order inside a class is whatever
    //      the atomics make it — no result depends on the order in which segments are shaded.
    if (blockIdx.x != 0) return;
    const uint32_t* __restrict__ cnt = seg.active[cur];
    const uint32_t* __restrict__ cost = seg.shade_cost + (size_t)bounce_closest * seg.nseg;
    __shared__ uint32_t hist[256];
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    __syncthreads();
    auto size_class = [&](uint32_t w) {
      const uint32_t c = cost[w], v = c ? c : cnt[w];
      const uint32_t e = 31u - (uint32_t)__builtin_clz(v | 8u);
      return 255u - (v < 8u ? v : (e - 2u) * 8u + ((v >> (e - 3u)) & 7u));
    };
    for (uint32_t w = threadIdx.x; w < seg.nseg; w += 1024) atomicAdd(&hist[size_class(w)], 1u);
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int b = 0; b < 256; b++) { const uint32_t h = hist[b]; hist[b] = run; run += h; } }
    __syncthreads();
    for (uint32_t w = threadIdx.x; w < seg.nseg; w += 1024) seg.shade_order[atomicAdd(&hist[size_class(w)], 1u)] = w;
    return;
  }
  const bool sh = blockIdx.y == 1;
  if (sh && !do_shadow) return;
  const uint32_t* __restrict__ counts = sh ? seg.shadow : seg.active[cur];
  uint32_t* __restrict__ table = sh ? seg.table_shadow : seg.table_closest;
  __shared__ uint32_t part[1024];
  __shared__ uint32_t base_sh;
  const uint32_t slice = (seg.nseg + kTableBlocks - 1) / kTableBlocks;  // segments per block (<= 1024 for nseg <= 32768)
  const uint32_t w_begin = blockIdx.x * slice;
  const uint32_t w_end = w_begin + slice < seg.nseg ? w_begin + slice : seg.nseg;
  // base = chunks of all segments before this slice; the last block also learns the grand total
  uint32_t before = 0, total = 0;
  for (uint32_t w = threadIdx.x; w < seg.nseg; w += 1024) {
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
  // inclusive scan of this slice's per-segment chunk counts (thread t <-> segment w_begin + t)
  __shared__ uint32_t rays_of[1024];   // rays of segment w_begin + t: an entry carries its chunk's ray count, so that the trace waves need no second look-up
  const uint32_t mine_n = (w_begin + threadIdx.x < w_end) ? counts[w_begin + threadIdx.x] : 0u;
  rays_of[threadIdx.x] = mine_n;
  const uint32_t mine = (mine_n + 63u) / 64u;
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
    const uint32_t left = rays_of[lo] - k * 64u;   // >= 1: only non-empty chunks are listed
    table[base + e] = chunk_entry(w_begin + lo, k, left < 64u ? left : 64u);
  }
}

// ---- the per-sample radiance buffer Lbuf -----------------------------------------------------------------------------
// One vec4 per (pixel, sample in flight), laid out TILE-major: the entries of one 8x8 tile under all samples are contiguous (128 KB at 128
// samples in flight) and a segment's rays all belong to its tile(s): the shadow kernel's read-modify-writes of a segment stay inside that
// window (a [sample][pixel] layout spread them over planes 33 MB apart: every access its own line).  `pid` in the path state IS this index.
// Inside a tile the order is [pixel][sample] (r4, PT_PIXEL_MAJOR; r1-r3 had [sample][pixel]): the camera rays of a chunk are 64 SAMPLES OF
// ONE PIXEL (k_raygen), so a chunk's 64 entries are 1 KB contiguous here too, and k_accumulate folds a pixel's samples from consecutive words.
#ifndef PT_PIXEL_MAJOR
#define PT_PIXEL_MAJOR 1
#endif
__device__ __forceinline__ uint32_t lbuf_index(uint32_t tile, uint32_t s, uint32_t nsamples, uint32_t lane) {
#if PT_PIXEL_MAJOR
  return (tile * 64u + lane) * nsamples + s;
#else
  return (tile * nsamples + s) * 64u + lane;
#endif
}
constexpr uint32_t kLbufSampleStride = PT_PIXEL_MAJOR ? 1u : 64u;  // distance between successive samples of one pixel
// (host mirrors for the debug build's $PTAMD_DEBUG_PIXEL read-back: the layout this translation unit was compiled with)
size_t lbuf_index_host(uint32_t tile, uint32_t s, uint32_t nsamples, uint32_t lane) {
  return PT_PIXEL_MAJOR ? ((size_t)tile * 64u + lane) * nsamples + s : ((size_t)tile * nsamples + s) * 64u + lane;
}
size_t lbuf_sample_stride_host() { return kLbufSampleStride; }
__device__ __forceinline__ uint32_t lbuf_index_of_pixel(uint32_t p, uint32_t W, uint32_t s, uint32_t nsamples) {
  const uint32_t y = p / W, x = p - y * W, tilesX = (W + 7u) / 8u;
  return lbuf_index((y >> 3) * tilesX + (x >> 3), s, nsamples, (y & 7u) * 8u + (x & 7u));
}
// A segment's window of Lbuf starts at its first tile; a path's entry is that base + the relative index it carries in rayD.w.
__device__ __forceinline__ uint32_t segment_first_tile(const Segments& seg, uint32_t sg) {
  // consecutive segments cycle over `bands` horizontal bands of the image (the chunk tables list the segments in order, so the
  // rays in flight in a trace kernel come from `bands` neighbourhoods instead of one)
  const uint32_t per_band = seg.nseg / seg.bands;  // nseg is a multiple of bands
  return ((sg % seg.bands) * per_band + sg / seg.bands) * seg.tiles_per_seg;
}
__device__ __forceinline__ uint32_t segment_lbuf_base(const Segments& seg, uint32_t sg) { return segment_first_tile(seg, sg) * seg.nsamples * 64u; }
// the segment a queue slot belongs to (inverse of seg_slot)
__device__ __forceinline__ uint32_t slot_segment(uint32_t nseg, uint32_t slot) { return ((slot >> 6) / PT_SEG_GROUP) % nseg; }
// the pixel (row-major) of a pid of a ONE-sample batch (the debug entry points: pt_trace_primary, pt_debug_sample)
__device__ __forceinline__ uint32_t pixel_of_pid_1spp(uint32_t pid, uint32_t W) {
  const uint32_t lane = pid & 63u, tile = pid >> 6, tilesX = (W + 7u) / 8u;
  const uint32_t ty = tile / tilesX, tx = tile - ty * tilesX;
  return (ty * 8u + (lane >> 3)) * W + tx * 8u + (lane & 7u);
}

// ---- raygen ------------------------------------------------------------------------------------------------------
// Segment s = T consecutive tiles under all samples of the batch.  One wave iteration emits 64 camera rays; lanes outside the image (partial
// edge tiles) are squeezed out.  r4 (PT_PIXEL_MAJOR): the 64 rays of an iteration are consecutive SAMPLES OF ONE PIXEL (pixel-major,
// sample-minor through the tile) where r1-r3 emitted one sample of the tile's 64 pixels.  The samples of a pixel differ by a sub-pixel
// jitter, so the lanes of a bounce-0 chunk walk the same nodes and test the same triangles: their loads fall on the same addresses (one
// L1 look-up instead of 64), they finish together, and their hits shade one triangle of one material.  Nothing else knows the order of a
// segment's entries: a path finds its radiance entry through `pid`, which is computed here from (tile, pixel, sample).
__global__ void __launch_bounds__(kBlock) k_raygen(DeviceScene S, PathState st, vec4* __restrict__ Lbuf, Segments seg,
                                                    BatchCounters* __restrict__ ctr, uint32_t first_sample,
                                                    uint32_t nsamples, uint32_t tilesX, uint32_t tilesY) {
  const uint32_t lane = wave_lane();
  const uint32_t tiles = tilesX * tilesY;
  uint32_t started = 0;
  for (uint32_t sg = wave_index(); sg < seg.nseg; sg += wave_count()) {
    uint32_t n_out = 0;
    const uint32_t first_tile = segment_first_tile(seg, sg);
    for (uint32_t k = 0; k < seg.tiles_per_seg * nsamples; k++) {
      const uint32_t tile = first_tile + k / nsamples;
      if (tile >= tiles) break;  // wave-uniform
#if PT_PIXEL_MAJOR
      const uint32_t r = (k % nsamples) * 64u + lane;   // ray r of the tile's 64 * nsamples, pixel-major
      const uint32_t pl = r / nsamples, s = r - pl * nsamples;
#else
      const uint32_t pl = lane, s = k % nsamples;
#endif
      const uint32_t ty = tile / tilesX;
      const uint32_t x = (tile - ty * tilesX) * 8 + (pl & 7);
      const uint32_t y = ty * 8 + (pl >> 3);
      const bool valid = x < S.width && y < S.height;
      RayGenOut rg;
      if (valid) rg = stage_raygen(S, x, y, first_sample + s);
      const unsigned long long m = __ballot(valid);
      if (valid) {
        const uint32_t j = seg_slot(seg.nseg, sg, n_out + wave_prefix(m));
        const uint32_t pid = lbuf_index(tile, s, nsamples, pl);
        const uint32_t rel = pid - first_tile * nsamples * 64u;  // relative to the segment's window (segment_lbuf_base)
        st.rayO[j] = vec4{rg.o.x, rg.o.y, rg.o.z, 0.0f};
        st.rayD[j] = vec4{rg.d.x, rg.d.y, rg.d.z, u2f((rg.dim & kMetaDimMask) | (rel << kMetaPidShift))};
        st.att[j] = vec4{1.0f, 1.0f, 1.0f, u2f(rg.offset)};
        Lbuf[pid] = vec4{0.0f, 0.0f, 0.0f, 1.0f};
      }
      n_out += (uint32_t)__popcll(m);
    }
    if (lane == 0) { seg.active[0][sg] = n_out; seg.poison[sg] = 0u; }
    started += n_out;
  }
  if (lane == 0) {
    WaveStats& ws = seg.stats[wave_index()];
    ws.paths += started;
    ws.closest += started;
  }
}

// ---- wave-cooperative traversal step ---------------------------------------------------------------------------------
// One iteration for the lanes of a wave that still hold a ray: (1) lanes with a node to visit and room in their leaf
// queue visit it (trav_node: slab tests, candidate leaves queued, next node chosen); (2) the wave votes — when at least
// half of these lanes have a queued leaf, or a lane cannot go on without emptying its queue (queue full, or node stack
// exhausted with nobody else able to advance), every lane with a queued leaf runs ONE triangle test.  The triangle code
// therefore executes with most lanes busy instead of whenever a single lane met a leaf.  A ray is finished when its node
// stack is exhausted and its queue is empty (any-hit: on the first accepted hit).
#ifndef PT_TWO_BLOCK
#define PT_TWO_BLOCK 1024
#define PT_TWO_BLOCKS_PER_CU 1
#define PT_TWO_LDS_NODES 1024
#endif
constexpr uint32_t kTraceBlock2 = PT_TWO_BLOCK;  // two-level kernels: ONE block per CU (4 waves per SIMD) sharing the staged node array
constexpr uint32_t kLdsNodes = PT_TWO_LDS_NODES; // 64 KB of nodes in LDS beside the 88 KB of stacks and leaf queues of 1024 lanes (152 of 160 KB)
#ifndef PT_TRI_VOTE
#define PT_TRI_VOTE 2
#endif
template <bool ANY, bool COUNT, bool TWO, bool W6 = false>
__device__ __forceinline__ void wave_traverse(const DeviceScene& S, const BvhNode* lds_nodes, TravState& ts, TraversalCount* tc) {
  // room a node step needs in the lane's leaf queue: one entry per child it can queue (4-wide), one entry per node (6-wide: a range)
  constexpr int kRoom = W6 ? 1 : 4, kRows = W6 ? kPendLeaves6 : kPendLeaves;
  if (ts.cur != kInvalidRef && ts.st.npend <= kRows - kRoom) {
    if (TWO) {
      // Exit markers are handled at once; ENTERING an instance (two dependent loads, three divides) is voted on like the
      // triangle tests: it runs when half of the lanes that can advance are waiting at an instance, or nobody has a node.
      trav_leave(ts);
      const bool at_inst = ts.cur != kInvalidRef && (ts.cur & kInstBit) != 0;
      const bool at_node = ts.cur != kInvalidRef && !at_inst;
      const unsigned long long mi = __ballot(at_inst), mn = __ballot(at_node);
      if (mi != 0 && (2 * __popcll(mi) >= __popcll(mn) || mn == 0)) {
        if (at_inst) trav_resolve(S, ts);
      } else if (at_node) {
        if (lds_nodes) trav_node<COUNT, true>(lds_nodes, ts, tc); else trav_node<COUNT, true>(S.nodes, ts, tc);
      }
    } else if (W6) {
      trav_node6<COUNT>(S.nodes, ts, tc);
    } else {
      trav_node<COUNT, false>(S.nodes, ts, tc);
    }
  }
  const bool pending = ts.st.npend > 0;
  const bool stuck = pending && (ts.cur == kInvalidRef || ts.st.npend > kRows - kRoom);
  const bool advancing = ts.cur != kInvalidRef && !stuck;
  const unsigned long long mp = __ballot(pending);
  if (mp == 0) return;
  // triangle round when half of the lanes holding a ray have a queued leaf, a lane's queue is full, or nobody can advance
  const bool go = PT_TRI_VOTE * __popcll(mp) >= __popcll(__ballot(1)) || __ballot(ts.st.npend > kRows - kRoom) != 0 || __ballot(advancing) == 0;
  if (go && pending) {
    const bool fin = W6 ? trav_pending_leaf6<ANY, COUNT>(S, ts, tc) : trav_pending_leaf<ANY, COUNT>(S, ts, tc);
    if (fin) { ts.cur = kInvalidRef; ts.st.npend = 0; ts.st.sp = 0; }
  }
}

// the two-level structure's nodes -> LDS (when they fit); returns the pointer the traversal should use (nullptr: HBM)
__device__ __forceinline__ const BvhNode* stage_nodes(const DeviceScene& S, BvhNode* lds_nodes) {
  if (S.node_count > kLdsNodes) return nullptr;
  const uint4* src = reinterpret_cast<const uint4*>(S.nodes);
  uint4* dst = reinterpret_cast<uint4*>(lds_nodes);
  for (uint32_t i = threadIdx.x; i < S.node_count * 4; i += blockDim.x) dst[i] = src[i];
  __syncthreads();
  return lds_nodes;
}

// ---- closest hit ---------------------------------------------------------------------------------------------------
template <bool COUNT, bool TWO, bool W6 = false>
__global__ void __launch_bounds__(TWO ? kTraceBlock2 : kBlock, TWO ? PT_TWO_BLOCKS_PER_CU : PT_CLOSEST_WAVES)
k_trace_closest(DeviceScene S, PathState st, vec4* __restrict__ hit, Segments seg, uint32_t cur, BatchCounters* __restrict__ ctr, uint32_t bounce,
                uint32_t* __restrict__ spill, int32_t* __restrict__ hitlog, uint32_t log_stride) {
  PT_TAIL_BEGIN
  constexpr uint32_t kTB = TWO ? kTraceBlock2 : kBlock;
  __shared__ uint32_t lds_stack[(W6 ? kLdsStack6 : kLdsStack) + 1][kTB];
  __shared__ uint32_t lds_pend[(W6 ? kPendLeaves6 : kPendLeaves) + 1][kTB];
  __shared__ BvhNode lds_nodes_buf[TWO && kLdsNodes ? kLdsNodes : 1];
  const BvhNode* lds_nodes = TWO && kLdsNodes ? stage_nodes(S, lds_nodes_buf) : nullptr;
  const uint32_t lane = wave_lane();
  ChunkClaims src;
  src.init(seg.table_closest, ctr->chunks_closest[bounce], &ctr->work_closest[bounce], seg, lane);
  TraversalStack stack;
  stack.lds = &lds_stack[0][threadIdx.x];
  stack.pend = &lds_pend[0][threadIdx.x];
  stack.lds_stride = kTB;
  stack.spill = spill + ((size_t)blockIdx.x * kTB + threadIdx.x);
  stack.spill_stride = gridDim.x * kTB;
  TraversalCount tc;

  // (written as an explicit init / step loop: with the whole traversal behind one call the compiler produced a kernel
  //  1.5x slower on secondary rays)
  TravState ts;
  uint32_t ray = kInvalidRef;
  auto finish = [&]() {
    const RayHit& h = ts.best;
#ifdef PT_DEBUG_PID
    if (ts.dbg) printf("dbg finish: t %.9g u %.9g v %.9g tri %u gid %u\n", h.t, h.u, h.v, h.tri, h.gid >> 2);
    ts.dbg = false;
#endif
    hit[ray] = vec4{h.t, h.u, h.v, u2f(h.tri == kInvalidRef ? kInvalidRef : (h.tri | ((h.gid & 3u) << 28)))};
    if (hitlog) {
      // (the hit log is only kept for one-sample batches)
      const uint32_t pixel = pixel_of_pid_1spp(segment_lbuf_base(seg, slot_segment(seg.nseg, ray)) + (f2u(st.rayD[ray].w) >> kMetaPidShift), S.width);
      int32_t* hl = &hitlog[((size_t)bounce * log_stride + pixel) * 2];
      hl[0] = hl[1] = -1;
      if (h.tri != kInvalidRef) triangle_ids(S, h.tri, &hl[0], &hl[1]);
    }
    ray = kInvalidRef;
  };
  const uint32_t refill = seg.refill_threshold;
  for (;;) {
    PT_TAIL_MARK
    const uint32_t fresh = src.take(ray == kInvalidRef);
    PT_TAIL_T(tail_take)
    if (fresh != kInvalidRef) {
      ray = fresh;
      const vec4 o4 = st.rayO[ray];
      const vec4 d4 = st.rayD[ray];
      float ir = 0.0f;  // alpha-test payload: the sample drawn before `intersect` (kernel.metal:510), only needed with cut-outs
      if (S.has_alpha) ir = Halton{halton_table(S.halton), f2u(st.att[ray].w), f2u(d4.w) & kMetaDimMask}.sample1d();
#ifdef PT_DEBUG_PID
      ts.dbg = ctr->_pad[1] == bounce + 1u && segment_lbuf_base(seg, slot_segment(seg.nseg, ray)) + (f2u(d4.w) >> kMetaPidShift) == ctr->_pad[0];
      if (ts.dbg) printf("dbg ray slot %u lane %u wave %u o %.9g %.9g %.9g d %.9g %.9g %.9g\n", ray, lane, wave_index(), o4.x, o4.y, o4.z, d4.x, d4.y, d4.z);
#endif
      if (trav_init(S, ts, v3(o4.x, o4.y, o4.z), v3(d4.x, d4.y, d4.z), 1e-3f, kInf, ir, stack, false, COUNT ? &tc : nullptr)) finish();
    }
    PT_TAIL_T(tail_setup)
    if (__ballot(ray != kInvalidRef) == 0) {
      if (src.exhausted && src.pool_next == src.pool_end) break;
      continue;
    }
    while (ray != kInvalidRef) {
      wave_traverse<false, COUNT, TWO, W6>(S, lds_nodes, ts, &tc);
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
  PT_TAIL_END(0)
}

// ---- shade -----------------------------------------------------------------------------------------------------------
// Persistent grid sized for the kernel's occupancy (launch_shade); wave w shades segments w, w + P, ...
// Per block, once: the read-mostly tables every hit touches are staged in LDS — the Halton entries of the dimensions
// this bounce can reach (5 + 7b .. 15 + 12b: raygen consumes 4, every bounce 7 to 12), the area-light table (binary
// search + one record per NEE sample) and the E_avg / E_ms_avg energy tables.
// Per hit: geometry + material + BSDF set-up, then next-event estimation whose shadow record is compacted into the shadow
// queue right away, then the BSDF sample / throughput / roulette and the compaction of the survivor.  NEE before the
// sample (each random number is addressed by its dimension, so the order does not matter) means the shadow record is
// not held in registers across the sampling code.
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 4   // waves per SIMD the kernel is compiled for (128 VGPRs; 14 spill to scratch)
#endif
#ifndef PT_SHADE_BLOCK
#define PT_SHADE_BLOCK 256
#endif
constexpr uint32_t kShadeBlock = PT_SHADE_BLOCK;
constexpr uint32_t kShadeHalton = 128;  // staged Halton window (entries); dimensions beyond it are read from HBM
constexpr uint32_t kShadeLights = 64;   // staged area lights; a bigger table is read from HBM
constexpr int kLutE = 128, kLutEavg = 128, kLutEavgMs = 32;  // table shapes (checked by load_luts)
constexpr uint32_t kBinCap = 128;       // a bin is emptied as soon as it holds 64 entries, and a scan step adds at most 64

__global__ void __launch_bounds__(PT_SHADE_BLOCK, (PT_SHADE_WAVES * 4 * 64) / PT_SHADE_BLOCK)
k_shade(const DeviceScene* __restrict__ Sp, PathState sin, PathState sout, const vec4* __restrict__ hit, ShadowQueue sq,
        vec4* __restrict__ Lbuf, Segments seg, uint32_t cur, BatchCounters* __restrict__ ctr, uint32_t bounce) {
  PT_TAIL_BEGIN
  const DeviceScene& S = *Sp;  // scene table read through the scalar cache: by value it cost 100 spilled SGPRs here
  __shared__ HaltonEntry lds_halton[kShadeHalton];
  __shared__ LightRec lds_lights[kShadeLights];
  __shared__ float lds_Eavg[kLutEavg];
  __shared__ float lds_EavgMs[kLutEavgMs * kLutEavgMs];
  static_assert(kMaxSegmentSlots <= 65536, "slot numbers inside a segment are binned as uint16_t (queue_plan.h bounds seg_cap)");
  __shared__ uint16_t lds_bins[kShadeBlock / 64][5][kBinCap];  // per wave: slot numbers by material class (+ misses), 1.25 KB
  __shared__ uint32_t lds_bin_tri[kShadeBlock / 64][5][kBinCap]; // ... and the triangle hit there: the pass starts its ShadeRec load with the state gather
  const uint32_t halton_base = 5u + 7u * bounce;
  uint32_t halton_count = 11u + 5u * bounce;
  if (halton_base >= (uint32_t)kHaltonDims) halton_count = 0;
  else if (halton_base + halton_count > (uint32_t)kHaltonDims) halton_count = (uint32_t)kHaltonDims - halton_base;
  if (halton_count > kShadeHalton) halton_count = kShadeHalton;
  const bool lights_in_lds = S.lightCount <= kShadeLights;
  {
    const uint4* src = reinterpret_cast<const uint4*>(S.halton + halton_base);
    uint4* dst = reinterpret_cast<uint4*>(lds_halton);
    for (uint32_t i = threadIdx.x; i < halton_count * 2; i += kShadeBlock) dst[i] = ldg(&src[i]);
    if (lights_in_lds) {
      const uint4* ls = reinterpret_cast<const uint4*>(S.light_recs);
      uint4* ld = reinterpret_cast<uint4*>(lds_lights);
      for (uint32_t i = threadIdx.x; i < S.lightCount * (uint32_t)(sizeof(LightRec) / 16); i += kShadeBlock) ld[i] = ldg(&ls[i]);
    }
    for (uint32_t i = threadIdx.x; i < (uint32_t)kLutEavg; i += kShadeBlock) lds_Eavg[i] = ldg(&S.luts.Eavg.d[i]);
    for (uint32_t i = threadIdx.x; i < (uint32_t)(kLutEavgMs * kLutEavgMs); i += kShadeBlock) lds_EavgMs[i] = ldg(&S.luts.EavgMs.d[i]);
  }
  __syncthreads();
  ShadeTables T;
  T.luts.E = Lut{S.luts.E.d, kLutE, kLutE, 1, 0};
  T.luts.Eavg = Lut{lds_Eavg, kLutEavg, 1, 1, 1};
  T.luts.EMs = Lut{S.luts.EMs.d, S.luts.EMs.w, S.luts.EMs.h, S.luts.EMs.depth, 0};
  T.luts.EavgMs = Lut{lds_EavgMs, kLutEavgMs, kLutEavgMs, 1, 1};
  T.luts.ETransIn = Lut{S.luts.ETransIn.d, S.luts.ETransIn.w, S.luts.ETransIn.h, S.luts.ETransIn.depth, 0};
  T.luts.ETransOut = Lut{S.luts.ETransOut.d, S.luts.ETransOut.w, S.luts.ETransOut.h, S.luts.ETransOut.depth, 0};
  T.halton = HaltonTab{S.halton, lds_halton, halton_base, halton_count};
  T.lights = lights_in_lds ? lds_lights : S.light_recs;
  T.light_cdf = S.light_cdf;
  T.lights_lds = lights_in_lds ? 1 : 0;

  const uint32_t lane = wave_lane();
  uint32_t shaded = 0, total_out = 0, total_shadow = 0;
  // first claim = the wave's index; further ones come from a per-launch cursor (segments differ in size by the time the
  // later bounces are reached, so a static deal would leave waves idle at the end of every launch)
  uint32_t claim = wave_index();   // position in seg.shade_order: the segments, longest first (k_chunk_tables)
  uint16_t* bin = &lds_bins[threadIdx.x >> 6][0][0];  // this wave's bins: [class][kBinCap] slot numbers within the segment
  uint32_t* bin_tri = &lds_bin_tri[threadIdx.x >> 6][0][0];
  while (claim < seg.nseg) {
    const uint32_t sg = __builtin_amdgcn_readfirstlane(seg.shade_order[claim]);
    const unsigned long long seg_t0 = wall_clock64();
    const uint32_t lbuf_base = segment_lbuf_base(seg, sg);  // this segment's window of the per-sample radiance buffer
    const uint32_t n = __builtin_amdgcn_readfirstlane(seg.active[cur][sg]);  // (a scalar for the compiler too: the scan loop and the bin counters stay in SGPRs)
    uint32_t n_out = 0, n_shadow = 0;
    // != 0: a path of this segment (paths never leave their segment) carries a throughput with an infinity or a NaN in it
    const uint32_t seg_poisoned = S.env_texture < 0 ? __builtin_amdgcn_readfirstlane(seg.poison[sg]) : 0u;
    unsigned long long poisoned_now = 0;
    // ---- hits are shaded one MATERIAL CLASS per wave pass.  The segment is scanned 64 slots at a time; every slot number goes
    //      to the LDS bin of its hit's class (4 lobe classes + misses); as soon as a bin holds 64 entries they are shaded
    //      together; what is left at the end of the segment is shaded in mixed passes.  (Measured on C3: the same kernel costs
    //      0.092 ns per hit when every sphere is diffuse and 0.126 with the per-instance material mix.)
    uint32_t cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0, cnt4 = 0;  // wave-uniform bin fill levels
    uint32_t k0 = 0;
    for (;;) {
      while (k0 < n && cnt0 < 64 && cnt1 < 64 && cnt2 < 64 && cnt3 < 64 && cnt4 < 64) {
        const uint32_t k = k0 + lane;
        uint32_t cls = 5;  // no entry
        uint32_t w = 0;
        if (k < n) {
          w = f2u(hit[seg_slot(seg.nseg, sg, k)].w);
          cls = w == kInvalidRef ? 4u : (w >> 28) & 3u;
          // a miss without an environment adds attenuation * backgroundColor = attenuation * 0 (kernel.metal:311 / :541, defs.metal:21):
          // nothing — unless the BSDF has driven the throughput to an infinity or a NaN (a zero pdf), when the reference's sum turns NaN
          // (one in ~1e8 paths: the segment's flag says whether the throughputs of its misses have to be looked at at all)
          if (w == kInvalidRef && S.env_texture < 0) {
            if (!seg_poisoned) cls = 5u;
            else {
              const vec4 a4 = sin.att[seg_slot(seg.nseg, sg, k)];
              if (!att_poisoned(v3(a4.x, a4.y, a4.z))) cls = 5u;
            }
          }
        }
        const unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1), m2 = __ballot(cls == 2), m3 = __ballot(cls == 3),
                                 m4 = __ballot(cls == 4);
        if (cls < 5) {
          const uint32_t at = cls == 0 ? cnt0 + wave_prefix(m0) : cls == 1 ? cnt1 + wave_prefix(m1) : cls == 2 ? cnt2 + wave_prefix(m2)
                            : cls == 3 ? cnt3 + wave_prefix(m3) : cnt4 + wave_prefix(m4);
          bin[cls * kBinCap + at] = (uint16_t)k;
          bin_tri[cls * kBinCap + at] = w & kHitTriMask;
        }
        cnt0 += (uint32_t)__popcll(m0); cnt1 += (uint32_t)__popcll(m1); cnt2 += (uint32_t)__popcll(m2);
        cnt3 += (uint32_t)__popcll(m3); cnt4 += (uint32_t)__popcll(m4);
        k0 += 64;
      }
      // one pass: a full bin if there is one, else (segment scanned) whatever is left, class by class
      // (the fill levels are only ever named by constant: they stay in scalar registers)
      uint32_t k = kInvalidRef, tri = 0;
      bool is_miss = false;
      const uint32_t full = cnt0 >= 64 ? 0u : cnt1 >= 64 ? 1u : cnt2 >= 64 ? 2u : cnt3 >= 64 ? 3u : cnt4 >= 64 ? 4u : 5u;
      if (full < 5) {
        uint32_t c = 0;
        if (full == 0) c = cnt0 -= 64; else if (full == 1) c = cnt1 -= 64; else if (full == 2) c = cnt2 -= 64;
        else if (full == 3) c = cnt3 -= 64; else c = cnt4 -= 64;
        k = bin[full * kBinCap + c + lane];
        tri = bin_tri[full * kBinCap + c + lane];
        is_miss = full == 4;
      } else {
        if (cnt0 + cnt1 + cnt2 + cnt3 + cnt4 == 0) break;
        // (k0 >= n here) fill the pass from the bins in class order
        uint32_t taken = 0;
#define PT_TAKE_FROM(c, cnt)                                                                                                   \
        {                                                                                                                      \
          const uint32_t take = cnt < 64 - taken ? cnt : 64 - taken;                                                           \
          if (lane >= taken && lane < taken + take) { k = bin[c * kBinCap + cnt - take + (lane - taken)]; tri = bin_tri[c * kBinCap + cnt - take + (lane - taken)]; is_miss = c == 4; }  \
          cnt -= take;                                                                                                         \
          taken += take;                                                                                                       \
        }
        PT_TAKE_FROM(0, cnt0) PT_TAKE_FROM(1, cnt1) PT_TAKE_FROM(2, cnt2) PT_TAKE_FROM(3, cnt3) PT_TAKE_FROM(4, cnt4)
#undef PT_TAKE_FROM
      }
      const bool has = k != kInvalidRef && !is_miss;
      const uint32_t i = seg_slot(seg.nseg, sg, k != kInvalidRef ? k : 0u);
      if (k != kInvalidRef && is_miss) {
        // a miss ends the path; it adds the environment's radiance (kernel.metal:517-539)
        const vec4 o4 = sin.rayO[i];
        const vec4 d4 = sin.rayD[i];
        const vec4 a4 = sin.att[i];
        const vec3 att = v3(a4.x, a4.y, a4.z);
        const uint32_t mpid = lbuf_base + (f2u(d4.w) >> kMetaPidShift);
        vec4 L = Lbuf[mpid];
        if (S.env_texture >= 0) {
          const vec3 Le = stage_miss(S, v3(d4.x, d4.y, d4.z), att, bounce, o4.w, (f2u(d4.w) & kMetaSpecular) != 0);
          L.x += Le.x; L.y += Le.y; L.z += Le.z;
        }
        // L += attenuation * backgroundColor (kernel.metal:311 / :541; backgroundColor = 0): + 0 for a finite throughput, NaN otherwise
        L.x += att.x * 0.0f; L.y += att.y * 0.0f; L.z += att.z * 0.0f;
        Lbuf[mpid] = L;
      }
      uint32_t c_shadow = 0, c_alive = 0;  // written by the lanes inside the divergent region, made wave-uniform after it
      if (has) {
        const vec4 h4 = ld_stream(&hit[i]);
        const vec4 o4 = ld_stream(&sin.rayO[i]);
        const vec4 d4 = ld_stream(&sin.rayD[i]);
        const vec4 a4 = ld_stream(&sin.att[i]);
        const uint32_t meta = f2u(d4.w);
        const uint32_t pid = lbuf_base + (meta >> kMetaPidShift);
        ShadeIn in;
        in.o = v3(o4.x, o4.y, o4.z);
        in.d = v3(d4.x, d4.y, d4.z);
        in.att = v3(a4.x, a4.y, a4.z);
        in.rayO = &sin.rayO[i];
        in.rayD = &sin.rayD[i];
        in.lastSpecular = (meta & kMetaSpecular) != 0;
        in.offset = f2u(a4.w);
        in.dim = (meta & kMetaDimMask) + 1;  // +1: the alpha-test payload `ir` drawn before intersect (kernel.metal:510)
        in.bounce = bounce;
        in.t = h4.x; in.u = h4.y; in.v = h4.z;
        in.tri = tri;
        ShadeGeom g;
        ShadingContext sc;
        shade_geometry(S, in, g, sc);
        const BSDF bsdf(sc, S.flags, T.luts, g.wo);
        // ---- NEE; its shadow ray goes to this segment of the shadow queue now (ballot over the lanes in here) ----
        uint32_t dim_rr;
        {
          const NeeOut nee = shade_nee(S, T, in, g, sc, bsdf, in.dim + 6);
          dim_rr = nee.dim;
          const unsigned long long m = __ballot(nee.shadow);
          if (nee.shadow) {
            const uint32_t j = seg_slot(seg.nseg, sg, n_shadow + wave_prefix(m));
            st_stream(&sq.o[j], vec4{g.hitPos.x, g.hitPos.y, g.hitPos.z, nee.tmax});
            st_stream(&sq.d[j], vec4{nee.d.x, nee.d.y, nee.d.z, u2f(pid)});
            st_stream(&sq.contrib[j], vec4{nee.contrib.x, nee.contrib.y, nee.contrib.z, nee.payload});
          }
          c_shadow = (uint32_t)__popcll(m);
        }
        // ---- BSDF sample, emission, throughput, roulette; survivors -> this segment of the next bounce's queue ----
        const BounceOut bo = shade_bounce(S, T, in, g, sc, bsdf, dim_rr);
        if (bo.has_emitted) {
          vec4 L = Lbuf[pid];
          L.x += bo.emitted.x; L.y += bo.emitted.y; L.z += bo.emitted.z;
          Lbuf[pid] = L;
        }
        const unsigned long long m = __ballot(bo.alive);
        poisoned_now |= __ballot(bo.alive && att_poisoned(bo.next_att));
        if (bo.alive) {
          const uint32_t j = seg_slot(seg.nseg, sg, n_out + wave_prefix(m));
          st_stream(&sout.rayO[j], vec4{g.hitPos.x, g.hitPos.y, g.hitPos.z, bo.next_pdf});
          st_stream(&sout.rayD[j], vec4{bo.next_d.x, bo.next_d.y, bo.next_d.z, u2f((bo.dim & kMetaDimMask) | (bo.next_specular ? kMetaSpecular : 0u) | (meta & ~(kMetaDimMask | kMetaSpecular)))});
          st_stream(&sout.att[j], vec4{bo.next_att.x, bo.next_att.y, bo.next_att.z, u2f(in.offset)});
        }
        c_alive = (uint32_t)__popcll(m);
      }
      // the counts of the lanes that were in the region, for every lane (the queue cursors are wave-uniform)
      const unsigned long long mh = __ballot(has);
      if (mh) {
        const int src = __builtin_ctzll(mh);
        n_shadow += (uint32_t)__builtin_amdgcn_readlane((int)c_shadow, src);
        n_out += (uint32_t)__builtin_amdgcn_readlane((int)c_alive, src);
        shaded += (uint32_t)__popcll(mh);
      }
    }
    if (lane == 0) {
      seg.active[cur ^ 1][sg] = n_out;
      seg.shadow[sg] = n_shadow;
      if (poisoned_now) seg.poison[sg] = 1u;
      seg.shade_cost[(size_t)bounce * seg.nseg + sg] = (uint32_t)(wall_clock64() - seg_t0) + 1u;   // the next batch's sort key (k_chunk_tables)
    }
    total_out += n_out;
    total_shadow += n_shadow;
    uint32_t next = 0;
    if (lane == 0) next = wave_count() + atomicAdd(&ctr->work_shade[bounce], 1u);
    claim = __builtin_amdgcn_readfirstlane(next);
  }
  if (lane == 0) {
    WaveStats& ws = seg.stats[wave_index()];
    ws.closest += total_out;
    ws.shadow += total_shadow;
    ws.shaded += shaded;
  }
  PT_TAIL_END(1)
}

// ---- shadow (any hit) --------------------------------------------------------------------------------------------------
template <bool COUNT, bool TWO, bool W6 = false>
__global__ void __launch_bounds__(TWO ? kTraceBlock2 : kBlock, TWO ? PT_TWO_BLOCKS_PER_CU : PT_SHADOW_WAVES)
k_trace_shadow(DeviceScene S, ShadowQueue sq, vec4* __restrict__ Lbuf, Segments seg, BatchCounters* __restrict__ ctr, uint32_t bounce,
               uint32_t* __restrict__ spill) {
  PT_TAIL_BEGIN
  constexpr uint32_t kTB = TWO ? kTraceBlock2 : kBlock;
  __shared__ uint32_t lds_stack[(W6 ? kLdsStack6 : kLdsStack) + 1][kTB];
  __shared__ uint32_t lds_pend[(W6 ? kPendLeaves6 : kPendLeaves) + 1][kTB];
  __shared__ BvhNode lds_nodes_buf[TWO && kLdsNodes ? kLdsNodes : 1];
  const BvhNode* lds_nodes = TWO && kLdsNodes ? stage_nodes(S, lds_nodes_buf) : nullptr;
  const uint32_t lane = wave_lane();
  ChunkClaims src;
  src.init(seg.table_shadow, ctr->chunks_shadow[bounce], &ctr->work_shadow[bounce], seg, lane);
  TraversalStack stack;
  stack.lds = &lds_stack[0][threadIdx.x];
  stack.pend = &lds_pend[0][threadIdx.x];
  stack.lds_stride = kTB;
  stack.spill = spill + ((size_t)blockIdx.x * kTB + threadIdx.x);
  stack.spill_stride = gridDim.x * kTB;
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
    PT_TAIL_MARK
    const uint32_t fresh = src.take(ray == kInvalidRef);
    PT_TAIL_T(tail_take)
    if (fresh != kInvalidRef) {
      ray = fresh;
      const vec4 o4 = sq.o[ray];
      const vec4 d4 = sq.d[ray];
      pid = f2u(d4.w);
      const float ir = S.has_alpha ? sq.contrib[ray].w : 0.0f;  // kernel.metal:625
      if (trav_init(S, ts, v3(o4.x, o4.y, o4.z), v3(d4.x, d4.y, d4.z), 1e-3f, o4.w, ir, stack, true, COUNT ? &tc : nullptr)) finish();
    }
    PT_TAIL_T(tail_setup)
    if (__ballot(ray != kInvalidRef) == 0) {
      if (src.exhausted && src.pool_next == src.pool_end) break;
      continue;
    }
    while (ray != kInvalidRef) {
      wave_traverse<true, COUNT, TWO, W6>(S, lds_nodes, ts, &tc);
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
  PT_TAIL_END(2)
}

// ---- accumulate (kernel.metal:672-684): running mean, one sample at a time, in sample order --------------------------
#if PT_PIXEL_MAJOR
// One wave per 8x8 tile.  A tile's entries of Lbuf are [pixel][sample]: eight samples of eight pixels are 8 x 128 contiguous bytes, so the
// wave loads blocks of 64 pixels x 8 samples fully coalesced (eight lanes per 128-byte line), turns them through LDS, and every lane then
// folds the eight samples of ITS pixel in sample order — the running mean is a sequential recurrence per pixel (kernel.metal:672-684).
__global__ void __launch_bounds__(kBlock) k_accumulate(vec4* __restrict__ acc, const vec4* __restrict__ Lbuf,
                                                        uint32_t npixels, uint32_t width, uint32_t nsamples, uint32_t n0,
                                                        uint32_t nonfinite_policy, BatchCounters* __restrict__ ctr) {
  constexpr uint32_t kRow = 9;  // vec4 per pixel row in LDS (8 samples + 1 of padding against bank conflicts)
  __shared__ vec4 stage[kBlock / 64][64 * kRow];
  const uint32_t lane = wave_lane(), w = threadIdx.x >> 6;
  const uint32_t height = npixels / width, tilesX = (width + 7u) / 8u, tiles = tilesX * ((height + 7u) / 8u);
  const uint32_t tile = blockIdx.x * (kBlock / 64) + w;   // (every wave of the block runs the same number of rounds: the barriers are uniform)
  const bool live = tile < tiles;
  const uint32_t ty = live ? tile / tilesX : 0u, x = (tile - ty * tilesX) * 8u + (lane & 7u), y = ty * 8u + (lane >> 3);
  const bool inside = live && x < width && y < height;
  const uint32_t p = y * width + x;
  vec4 a = inside ? acc[p] : vec4{0.0f, 0.0f, 0.0f, 0.0f};
  for (uint32_t s0 = 0; s0 < nsamples; s0 += 8u) {
    const uint32_t nb = nsamples - s0 < 8u ? nsamples - s0 : 8u;
    __syncthreads();
    if (live) {
#pragma unroll
      for (uint32_t i = 0; i < 8u; i++) {
        const uint32_t px = i * 8u + (lane >> 3), j = lane & 7u;
        if (j < nb) stage[w][px * kRow + j] = ld_stream(&Lbuf[(tile * 64u + px) * nsamples + s0 + j]);
      }
    }
    __syncthreads();
    if (inside) {
      for (uint32_t j = 0; j < nb; j++) {
        const vec4 v = stage[w][lane * kRow + j];
        vec3 L = v3(v.x, v.y, v.z);
        if (!(fabsf(L.x) <= 3.0e38f && fabsf(L.y) <= 3.0e38f && fabsf(L.z) <= 3.0e38f)) {  // NaN or inf
          atomicAdd(&ctr->nonfinite, 1u);
          if (nonfinite_policy == PT_NONFINITE_ZERO) L = v3(0.0f);
        }
        const uint32_t localFrameIdx = n0 + s0 + j;
        if (localFrameIdx > 0) {
          L = L + v3(a.x, a.y, a.z) * (float)localFrameIdx;
          L = L / (float)(localFrameIdx + 1);
        }
        a = vec4{L.x, L.y, L.z, 1.0f};
      }
    }
  }
  if (inside) acc[p] = a;
}
#else
__global__ void __launch_bounds__(kBlock) k_accumulate(vec4* __restrict__ acc, const vec4* __restrict__ Lbuf,
                                                        uint32_t npixels, uint32_t width, uint32_t nsamples, uint32_t n0,
                                                        uint32_t nonfinite_policy, BatchCounters* __restrict__ ctr) {
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= npixels) return;
  vec4 a = acc[p];
  const uint32_t l0 = lbuf_index_of_pixel(p, width, 0, nsamples);
  for (uint32_t s = 0; s < nsamples; s++) {
    const vec4 L4 = Lbuf[l0 + s * kLbufSampleStride];
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
#endif

// ---- GMoN ---------------------------------------------------------------------------------------------
// Accumulate into the bucket images exactly as the reference does with RendererFlags_GMoN: sample f goes to bucket
// f / ceil(spp / buckets) (renderer_pt.cpp:124-139) with the running-mean weight n = f / gmonBuckets
// (kernel.metal:675-681 — note: NOT the sample's index inside its bucket; reproduced as is).
__global__ void __launch_bounds__(kBlock) k_accumulate_gmon(vec4* __restrict__ buckets, const vec4* __restrict__ Lbuf,
                                                             uint32_t npixels, uint32_t width, uint32_t nsamples, uint32_t n0,
                                                             uint32_t samples_per_bucket, uint32_t gmon_buckets, uint32_t bucket_base,
                                                             uint32_t nonfinite_policy, BatchCounters* __restrict__ ctr) {
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= npixels) return;
  const uint32_t l0 = lbuf_index_of_pixel(p, width, 0, nsamples);
  constexpr uint32_t kGroup = PT_PIXEL_MAJOR ? 8u : 1u;  // (as k_accumulate: one 128-byte line of samples per round of loads)
  vec4 v[kGroup];
  for (uint32_t s = 0; s < nsamples; s++) {
    if (s % kGroup == 0) {
#pragma unroll
      for (uint32_t j = 0; j < kGroup; j++)
        if (s + j < nsamples) v[j] = ld_stream(&Lbuf[l0 + (s + j) * kLbufSampleStride]);
    }
    vec4 L4 = v[0];
#pragma unroll
    for (uint32_t j = 1; j < kGroup; j++) if (s % kGroup == j) L4 = v[j];
    vec3 L = v3(L4.x, L4.y, L4.z);
    if (!(fabsf(L.x) <= 3.0e38f && fabsf(L.y) <= 3.0e38f && fabsf(L.z) <= 3.0e38f)) {
      atomicAdd(&ctr->nonfinite, 1u);
      if (nonfinite_policy == PT_NONFINITE_ZERO) L = v3(0.0f);
    }
    const uint32_t f = n0 + s;
    vec4* acc = buckets + (size_t)(f / samples_per_bucket - bucket_base) * npixels;  // (bucket_base: the buckets[] of a device-group member starts there)
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

// ---- device-group merge (multi_device.hip): weighted sum of the members' running means ------------------------------------
__global__ void __launch_bounds__(kBlock) k_weighted_add(vec4* __restrict__ out, const vec4* __restrict__ in, float w, uint32_t npixels,
                                                          uint32_t first, uint32_t last) {
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= npixels) return;
  const vec4 a = in[p];
  vec4 o = first ? vec4{0.0f, 0.0f, 0.0f, 0.0f} : out[p];
  o.x += w * a.x; o.y += w * a.y; o.z += w * a.z;
  o.w = last ? 1.0f : 0.0f;
  out[p] = o;
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
  for (uint32_t w = threadIdx.x; w < seg.nstats; w += kBlock) {
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

// one LightRec per area light (once per render)
__global__ void __launch_bounds__(kBlock) k_light_records(DeviceScene S, LightRec* __restrict__ out, float* __restrict__ cdf) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < S.lightCount) { out[i] = make_light_rec(S, S.lights[i]); cdf[i] = S.lights[i].cumulativePower; }
}

// one ShadeRec per triangle of every leaf slot: entry 2 * slot + half (once per render, after the BVH build)
__global__ void __launch_bounds__(kBlock) k_shade_records(DeviceScene S, ShadeRec* __restrict__ out) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= 2u * S.slot_count) return;
  const TriRec& tr = S.tris[t >> 1];
  const uint32_t gid = (t & 1u) ? tr.gid_b : tr.gid_a;
  if (gid == kInvalidRef) { out[t] = ShadeRec{}; return; }  // the B entry of a one-triangle slot: never referred to
  const uint32_t inst = tr.inst_code & kSlotInstMask;
  out[t] = make_shade_rec(S, inst, (gid >> 2) - S.instances[inst].tri_global_base);
}

// primary-ray records for pt_trace_primary: segment order -> pixel order
__global__ void __launch_bounds__(kBlock) k_hit_records(DeviceScene S, PathState st, const vec4* __restrict__ hit, Segments seg,
                                                         pt_hit_record* __restrict__ out) {
  const uint32_t lane = wave_lane();
  for (uint32_t sg = wave_index(); sg < seg.nseg; sg += wave_count()) {
    const uint32_t n = seg.active[0][sg];
    for (uint32_t k = lane; k < n; k += 64) {
      const uint32_t i = seg_slot(seg.nseg, sg, k);
      const vec4 h = hit[i];
      const uint32_t tri = f2u(h.w) == kInvalidRef ? kInvalidRef : (f2u(h.w) & kHitTriMask);
      pt_hit_record r;
      if (tri != kInvalidRef) {
        r.t = h.x; r.u = h.y; r.v = h.z;
        triangle_ids(S, tri, &r.instance, &r.primitive);
      } else {
        r.t = 0.0f; r.u = 0.0f; r.v = 0.0f; r.instance = -1; r.primitive = -1;
      }
      out[pixel_of_pid_1spp(segment_lbuf_base(seg, sg) + (f2u(st.rayD[i].w) >> kMetaPidShift), S.width)] = r;
    }
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
  hipLaunchKernelGGL(k_chunk_tables, dim3(kTableBlocks, 3), dim3(1024), 0, s, seg, cur, ctr, bounce_closest, bounce_shadow, do_shadow ? 1u : 0u);
}
// One-BVH scenes: `grid` blocks of 256 threads (7 per CU).  Two-level scenes (S.two_level): one 1024-thread block per CU.
uint32_t trace_block_threads(bool two_level) { return two_level ? kTraceBlock2 : (uint32_t)kBlock; }
uint32_t trace_blocks_per_cu_two_level() { return PT_TWO_BLOCKS_PER_CU; }
void launch_trace_closest(hipStream_t s, uint32_t grid, const DeviceScene& S, PathState st, vec4* hit, Segments seg, uint32_t cur,
                          BatchCounters* ctr, uint32_t bounce, uint32_t* spill, int32_t* hitlog, uint32_t log_stride, bool count) {
  if (S.two_level) {
    if (count)
      hipLaunchKernelGGL((k_trace_closest<true, true>), dim3(grid), dim3(kTraceBlock2), 0, s, S, st, hit, seg, cur, ctr, bounce, spill, hitlog, log_stride);
    else
      hipLaunchKernelGGL((k_trace_closest<false, true>), dim3(grid), dim3(kTraceBlock2), 0, s, S, st, hit, seg, cur, ctr, bounce, spill, hitlog, log_stride);
    return;
  }
  if (S.wide6) {
    if (count)
      hipLaunchKernelGGL((k_trace_closest<true, false, true>), dim3(grid), dim3(kBlock), 0, s, S, st, hit, seg, cur, ctr, bounce, spill, hitlog, log_stride);
    else
      hipLaunchKernelGGL((k_trace_closest<false, false, true>), dim3(grid), dim3(kBlock), 0, s, S, st, hit, seg, cur, ctr, bounce, spill, hitlog, log_stride);
    return;
  }
  if (count)
    hipLaunchKernelGGL((k_trace_closest<true, false>), dim3(grid), dim3(kBlock), 0, s, S, st, hit, seg, cur, ctr, bounce, spill, hitlog, log_stride);
  else
    hipLaunchKernelGGL((k_trace_closest<false, false>), dim3(grid), dim3(kBlock), 0, s, S, st, hit, seg, cur, ctr, bounce, spill, hitlog, log_stride);
}
uint32_t shade_block_threads() { return kShadeBlock; }
uint32_t shade_blocks_per_cu() { return (PT_SHADE_WAVES * 4 * 64) / PT_SHADE_BLOCK; }
void launch_shade(hipStream_t s, uint32_t grid, const DeviceScene* S, PathState sin, PathState sout, const vec4* hit,
                  ShadowQueue sq, vec4* Lbuf, Segments seg, uint32_t cur, BatchCounters* ctr, uint32_t bounce) {
  hipLaunchKernelGGL(k_shade, dim3(grid), dim3(kShadeBlock), 0, s, S, sin, sout, hit, sq, Lbuf, seg, cur, ctr, bounce);
}
void launch_trace_shadow(hipStream_t s, uint32_t grid, const DeviceScene& S, ShadowQueue sq, vec4* Lbuf, Segments seg,
                         BatchCounters* ctr, uint32_t bounce, uint32_t* spill, bool count) {
  if (S.two_level) {
    if (count) hipLaunchKernelGGL((k_trace_shadow<true, true>), dim3(grid), dim3(kTraceBlock2), 0, s, S, sq, Lbuf, seg, ctr, bounce, spill);
    else hipLaunchKernelGGL((k_trace_shadow<false, true>), dim3(grid), dim3(kTraceBlock2), 0, s, S, sq, Lbuf, seg, ctr, bounce, spill);
    return;
  }
  if (S.wide6) {
    if (count) hipLaunchKernelGGL((k_trace_shadow<true, false, true>), dim3(grid), dim3(kBlock), 0, s, S, sq, Lbuf, seg, ctr, bounce, spill);
    else hipLaunchKernelGGL((k_trace_shadow<false, false, true>), dim3(grid), dim3(kBlock), 0, s, S, sq, Lbuf, seg, ctr, bounce, spill);
    return;
  }
  if (count) hipLaunchKernelGGL((k_trace_shadow<true, false>), dim3(grid), dim3(kBlock), 0, s, S, sq, Lbuf, seg, ctr, bounce, spill);
  else hipLaunchKernelGGL((k_trace_shadow<false, false>), dim3(grid), dim3(kBlock), 0, s, S, sq, Lbuf, seg, ctr, bounce, spill);
}
void launch_accumulate(hipStream_t s, vec4* acc, const vec4* Lbuf, uint32_t npixels, uint32_t width, uint32_t nsamples, uint32_t n0,
                       uint32_t nonfinite_policy, BatchCounters* ctr) {
#if PT_PIXEL_MAJOR
  const uint32_t tiles = ((width + 7u) / 8u) * ((npixels / width + 7u) / 8u);   // one wave per 8x8 tile
  hipLaunchKernelGGL(k_accumulate, dim3((tiles + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, s, acc, Lbuf, npixels, width, nsamples, n0,
                     nonfinite_policy, ctr);
#else
  hipLaunchKernelGGL(k_accumulate, dim3((npixels + kBlock - 1) / kBlock), dim3(kBlock), 0, s, acc, Lbuf, npixels, width, nsamples, n0,
                     nonfinite_policy, ctr);
#endif
}
void launch_accumulate_gmon(hipStream_t s, vec4* buckets, const vec4* Lbuf, uint32_t npixels, uint32_t width, uint32_t nsamples, uint32_t n0,
                            uint32_t samples_per_bucket, uint32_t gmon_buckets, uint32_t bucket_base, uint32_t nonfinite_policy, BatchCounters* ctr) {
  hipLaunchKernelGGL(k_accumulate_gmon, dim3((npixels + kBlock - 1) / kBlock), dim3(kBlock), 0, s, buckets, Lbuf, npixels, width, nsamples, n0,
                     samples_per_bucket, gmon_buckets, bucket_base, nonfinite_policy, ctr);
}
void launch_weighted_add(hipStream_t s, vec4* out, const vec4* in, float w, uint32_t npixels, bool first, bool last) {
  hipLaunchKernelGGL(k_weighted_add, dim3((npixels + kBlock - 1) / kBlock), dim3(kBlock), 0, s, out, in, w, npixels, first ? 1u : 0u, last ? 1u : 0u);
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
  if (S.slot_count) hipLaunchKernelGGL(k_shade_records, dim3((2u * S.slot_count + kBlock - 1) / kBlock), dim3(kBlock), 0, s, S, out);
}
void launch_light_records(hipStream_t s, const DeviceScene& S, LightRec* out, float* cdf) {
  if (S.lightCount) hipLaunchKernelGGL(k_light_records, dim3((S.lightCount + kBlock - 1) / kBlock), dim3(kBlock), 0, s, S, out, cdf);
}
void launch_hit_records(hipStream_t s, uint32_t grid, const DeviceScene& S, PathState st, const vec4* hit, Segments seg,
                        pt_hit_record* out) {
  hipLaunchKernelGGL(k_hit_records, dim3(grid), dim3(kBlock), 0, s, S, st, hit, seg, out);
}

}  // namespace pt
