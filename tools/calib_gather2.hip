// tools/calib_gather2.hip — what limits random record gathers on MI355X: the L2's request rate, or the per-CU vector-L1 (TA/TCP)
// address rate?  ("validate the 80.9 G/s ceiling at 8 and 16 records in flight per lane and at 128-B records".)
//
// Every wave fetches pseudo-random aligned records of a table in one of these shapes:
//   mode 0  lane-private 64-B record:  each lane issues 4 x global_load_dwordx4 to its OWN record (the shape of the r2 trace kernels)
//   mode 1  quad-cooperative 64-B:     lanes 4q..4q+3 read the four 16-B parts of ONE record in one instruction (16 records / instr)
//   mode 2  oct-cooperative 128-B:     lanes 8q..8q+7 read the eight parts of ONE 128-B record in one instruction (8 records / instr)
//   mode 3  lane-private 128-B record: 8 x dwordx4 per lane
//   mode 4  lane-private 16 B:         one dwordx4 per lane at a random 64-B-aligned place (per-lane address rate)
//   mode 5  quad-cooperative 64-B through LDS-DMA (global_load_lds_dwordx4, per-lane source address) + each lane reads "its" record
//           back with 4 x ds_read_b128 (what a traversal kernel would do to hand every lane a whole node)
//   mode 6  lane-private 4 B:          one dword per lane
//   mode 7  pair-cooperative 32-B:     lanes 2q, 2q+1 read the halves of one 32-B record
//   mode 8  column-cooperative 64-B:   lanes l, l+16, l+32, l+48 (one lane per 16-lane row) read the four parts of ONE record in one
//                                       instruction: would the L1 merge them too?  (a transposition across ROWS is 4x cheaper: v_permlane*_swap)
// `inflight` independent records (modes 0,3,4,6: per lane; modes 1,2,5,7: instructions) are issued before any result is used.
//   ./calib_gather2 <mode> <table_MiB> <records_per_lane_or_instr_groups> <inflight 1|2|4|8|16>
// prints one JSON line: records/s, bytes/s, "lane-addresses"/s (active lanes x load instructions).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef unsigned long long u64;

__device__ __forceinline__ u64 mix(u64& x) {
  x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; x *= 0x94D049BB133111EBull; x ^= x >> 32;
  return x;
}
// uniform in [0, n) without a 64-bit division (n < 2^32)
__device__ __forceinline__ unsigned pick(u64& x, unsigned n) { return (unsigned)(((mix(x) >> 32) * (u64)n) >> 32); }

template <int MODE, int INFL>
__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ table, unsigned nrec, int iters, unsigned* __restrict__ sink) {
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
  u64 xl = (u64)(blockIdx.x * 256u + threadIdx.x) * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;  // per-lane stream
  u64 xq = (u64)(wave * 64u + (lane >> 2)) * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;          // per-quad stream
  u64 xo = (u64)(wave * 64u + (lane >> 3)) * 0x9E3779B97F4A7C15ull + 0x2545F4914F6CDD1Dull;          // per-oct stream
  u64 xp = (u64)(wave * 64u + (lane >> 1)) * 0x9E3779B97F4A7C15ull + 0x14057B7EF767814Full;          // per-pair stream
  u64 xc = (u64)(wave * 64u + (lane & 15u)) * 0x9E3779B97F4A7C15ull + 0x3C6EF372FE94F82Bull;         // per-column stream
  uint4 acc = {0, 0, 0, 0};
  constexpr int G = INFL > 2 ? 2 : INFL;                     // mode 5: groups of 64 records in flight per wave
  __shared__ uint4 stage[MODE == 5 ? 4 * 256 * G : 1];        //         4 waves x G x 4 KiB
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
      const uint4* p[INFL];
      for (int k = 0; k < INFL; k++) p[k] = table + (size_t)pick(xl, nrec) * 4;
      uint4 v[INFL][4];
      for (int k = 0; k < INFL; k++) for (int j = 0; j < 4; j++) v[k][j] = p[k][j];
      for (int k = 0; k < INFL; k++) { acc.x ^= v[k][0].x ^ v[k][1].y ^ v[k][2].z ^ v[k][3].w; acc.y += v[k][0].y + v[k][3].x; }
    } else if (MODE == 1) {
      uint4 v[INFL];
      for (int k = 0; k < INFL; k++) v[k] = table[(size_t)pick(xq, nrec) * 4 + (lane & 3u)];
      for (int k = 0; k < INFL; k++) { acc.x ^= v[k].x ^ v[k].w; acc.y += v[k].y + v[k].z; }
    } else if (MODE == 2) {
      uint4 v[INFL];
      for (int k = 0; k < INFL; k++) v[k] = table[(size_t)pick(xo, nrec) * 8 + (lane & 7u)];
      for (int k = 0; k < INFL; k++) { acc.x ^= v[k].x ^ v[k].w; acc.y += v[k].y + v[k].z; }
    } else if (MODE == 3) {
      const uint4* p[INFL];
      for (int k = 0; k < INFL; k++) p[k] = table + (size_t)pick(xl, nrec) * 8;
      uint4 v[INFL][8];
      for (int k = 0; k < INFL; k++) for (int j = 0; j < 8; j++) v[k][j] = p[k][j];
      for (int k = 0; k < INFL; k++) for (int j = 0; j < 8; j++) { acc.x ^= v[k][j].x; acc.y += v[k][j].w; }
    } else if (MODE == 4) {
      uint4 v[INFL];
      for (int k = 0; k < INFL; k++) v[k] = table[(size_t)pick(xl, nrec) * 4];
      for (int k = 0; k < INFL; k++) { acc.x ^= v[k].x ^ v[k].w; acc.y += v[k].y + v[k].z; }
    } else if (MODE == 6) {
      unsigned v[INFL];
      for (int k = 0; k < INFL; k++) v[k] = reinterpret_cast<const unsigned*>(table)[(size_t)pick(xl, nrec) * 16];
      for (int k = 0; k < INFL; k++) acc.x ^= v[k];
    } else if (MODE == 7) {
      uint4 v[INFL];
      for (int k = 0; k < INFL; k++) v[k] = table[(size_t)pick(xp, nrec) * 2 + (lane & 1u)];
      for (int k = 0; k < INFL; k++) { acc.x ^= v[k].x ^ v[k].w; acc.y += v[k].y + v[k].z; }
    } else if (MODE == 8) {
      uint4 v[INFL];
      for (int k = 0; k < INFL; k++) v[k] = table[(size_t)pick(xc, nrec) * 4 + (lane >> 4)];
      for (int k = 0; k < INFL; k++) { acc.x ^= v[k].x ^ v[k].w; acc.y += v[k].y + v[k].z; }
    } else if (MODE == 5) {
      // INFL groups of 4 LDS-DMA instructions: group g brings 64 records (one per lane of the wave) into 4 KiB of LDS
      uint4* my = stage + (size_t)(threadIdx.x >> 6) * (G * 256);
      for (int g = 0; g < G; g++)
        for (int j = 0; j < 4; j++) {
          const uint4* src = table + (size_t)pick(xq, nrec) * 4 + (lane & 3u);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(my + g * 256 + j * 64), 16, 0, 0);
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int g = 0; g < G; g++) {
        // lane L owns record L of the group: slot (L / 16) * 64 + (L % 16) * 4 .. + 3; parts read in a rotated order so that the 16
        // lanes of a ds_read_b128 group that share (slot mod 4) hit different banks
        const unsigned base = g * 256 + lane * 4, rot = (lane >> 2) & 3u;
        for (int t = 0; t < 4; t++) { const uint4 r = my[base + ((t + rot) & 3u)]; acc.x ^= r.x ^ r.w; acc.y += r.y + r.z; }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  if ((acc.x ^ acc.y) == 0x12345678u) sink[0] = acc.x;  // keep the loads alive
}

__global__ void k_fill(uint4* t, u64 n) {
  for (u64 i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) t[i] = uint4{(unsigned)i, (unsigned)(i >> 7), 3u, 4u};
}

template <int MODE>
static void launch(int infl, int blocks, const uint4* table, unsigned nrec, int iters, unsigned* sink) {
  switch (infl) {
    case 1: hipLaunchKernelGGL((k_gather<MODE, 1>), dim3(blocks), dim3(256), 0, 0, table, nrec, iters, sink); break;
    case 2: hipLaunchKernelGGL((k_gather<MODE, 2>), dim3(blocks), dim3(256), 0, 0, table, nrec, iters, sink); break;
    case 4: hipLaunchKernelGGL((k_gather<MODE, 4>), dim3(blocks), dim3(256), 0, 0, table, nrec, iters, sink); break;
    case 8: hipLaunchKernelGGL((k_gather<MODE, 8>), dim3(blocks), dim3(256), 0, 0, table, nrec, iters, sink); break;
    default: hipLaunchKernelGGL((k_gather<MODE, 16>), dim3(blocks), dim3(256), 0, 0, table, nrec, iters, sink); break;
  }
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const double mib = argc > 2 ? atof(argv[2]) : 15.0;
  const int per = argc > 3 ? atoi(argv[3]) : 64;
  int infl = argc > 4 ? atoi(argv[4]) : 4;
  if (infl != 1 && infl != 2 && infl != 4 && infl != 8) infl = 16;
  if (mode == 0 && infl > 8) infl = 8;   // 4 x 16 B x inflight registers
  if (mode == 3 && infl > 4) infl = 4;   // 8 x 16 B x inflight registers
  if (mode == 5 && infl > 2) infl = 2;   // LDS: 4 waves x inflight x 4 KiB
  const unsigned rec_bytes = (mode == 2 || mode == 3) ? 128u : (mode == 7 ? 32u : 64u);
  const u64 bytes = (u64)(mib * (1ull << 20));
  const unsigned nrec = (unsigned)(bytes / rec_bytes);
  uint4* table; unsigned* sink;
  if (hipMalloc(&table, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, table, bytes / 16);
  const int blocks = 256 * 8 * 4;  // 8 blocks per CU resident x 4 rounds
  const int iters = per / infl > 0 ? per / infl : 1;
  auto go = [&](int it) {
    switch (mode) {
      case 0: launch<0>(infl, blocks, table, nrec, it, sink); break;
      case 1: launch<1>(infl, blocks, table, nrec, it, sink); break;
      case 2: launch<2>(infl, blocks, table, nrec, it, sink); break;
      case 3: launch<3>(infl, blocks, table, nrec, it, sink); break;
      case 4: launch<4>(infl, blocks, table, nrec, it, sink); break;
      case 5: launch<5>(infl, blocks, table, nrec, it, sink); break;
      case 6: launch<6>(infl, blocks, table, nrec, it, sink); break;
      case 7: launch<7>(infl, blocks, table, nrec, it, sink); break;
      default: launch<8>(infl, blocks, table, nrec, it, sink); break;
    }
  };
  go(1);  // warm-up (also brings a small table into the L2s)
  go(iters);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  double best = 1e30;
  for (int rep = 0; rep < 3; rep++) {
    const auto t0 = std::chrono::steady_clock::now();
    go(iters);
    (void)hipDeviceSynchronize();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (s < best) best = s;
  }
  const double lanes = (double)blocks * 256, waves = lanes / 64;
  double recs, lane_addr;
  const double n_it = (double)iters * infl;
  switch (mode) {
    case 0: recs = lanes * n_it; lane_addr = recs * 4; break;
    case 1: case 8: recs = waves * 16 * n_it; lane_addr = waves * 64 * n_it; break;
    case 2: recs = waves * 8 * n_it; lane_addr = waves * 64 * n_it; break;
    case 3: recs = lanes * n_it; lane_addr = recs * 8; break;
    case 4: recs = lanes * n_it; lane_addr = recs; break;
    case 5: recs = waves * 64 * n_it; lane_addr = waves * 64 * 4 * n_it; break;
    case 6: recs = lanes * n_it; lane_addr = recs; break;
    default: recs = waves * 32 * n_it; lane_addr = waves * 64 * n_it; break;
  }
  const double fetched = mode == 4 ? recs * 16 : mode == 6 ? recs * 4 : recs * rec_bytes;
  printf("{\"mode\": %d, \"table_MiB\": %.1f, \"inflight\": %d, \"record_bytes\": %u, \"records\": %.0f, \"seconds\": %.6f, \"Grecords_per_s\": %.2f, "
         "\"GBs_fetched\": %.1f, \"Glane_addresses_per_s\": %.1f, \"lane_addresses_per_clk_per_CU_at_2.1GHz\": %.3f}\n",
         mode, mib, infl, rec_bytes, recs, best, recs / best / 1e9, fetched / best / 1e9, lane_addr / best / 1e9, lane_addr / best / 256 / 2.1e9);
  return 0;
}
