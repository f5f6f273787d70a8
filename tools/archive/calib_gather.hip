// tools/calib_gather.hip — calibrates rocprofv3's FETCH_SIZE for the access pattern of the traversal kernels: every lane
// reads one 64-byte-aligned 64-byte record (4 x global_load_dwordx4, a BvhNode) at a pseudo-random place of a table far
// larger than the 256 MB Infinity Cache, so every record comes from HBM exactly once per read.
//   ./calib_gather [table_GiB=4] [reads_per_lane=64]     prints records read, wall time, algorithmic GB/s (64 B per record)
// Run it plain for the time, and under `rocprofv3 --pmc FETCH_SIZE` for the counter: FETCH_SIZE_bytes / records = what the
// counter reports per 64-byte gather; time tells whether HBM moved 64 or 128 bytes for it (MI355X_MICROARCH.md §HBM asks
// for exactly this calibration before an absolute is trusted for an access width other than wide coalesced streams).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>

__global__ void __launch_bounds__(256) k_gather64(const uint4* __restrict__ table, unsigned long long nrec, int reads, unsigned* __restrict__ sink) {
  unsigned long long x = (unsigned long long)(blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
  uint4 acc = {0, 0, 0, 0};
  for (int i = 0; i < reads; i += 4) {  // four independent records in flight per lane
    const uint4* p[4];
    for (int k = 0; k < 4; k++) {
      x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
      p[k] = table + (x % nrec) * 4;
    }
    uint4 v[4][4];
    for (int k = 0; k < 4; k++) for (int j = 0; j < 4; j++) v[k][j] = p[k][j];
    for (int k = 0; k < 4; k++) { acc.x ^= v[k][0].x ^ v[k][1].y ^ v[k][2].z ^ v[k][3].w; acc.y += v[k][0].y + v[k][1].z + v[k][2].w + v[k][3].x; }
  }
  if ((acc.x ^ acc.y) == 0x12345678u) sink[0] = acc.x;  // keep the loads alive
}

__global__ void k_fill(uint4* t, unsigned long long n) {
  for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) t[i] = uint4{(unsigned)i, (unsigned)(i >> 7), 3u, 4u};
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 4.0;
  const int reads = argc > 2 ? atoi(argv[2]) : 64;
  const unsigned long long nrec = (unsigned long long)(gib * (1ull << 30)) / 64;
  uint4* table; unsigned* sink;
  if (hipMalloc(&table, nrec * 64) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, table, nrec * 4);
  const int blocks = 256 * 8 * 4;  // 8 blocks per CU resident x 4 rounds
  hipLaunchKernelGGL(k_gather64, dim3(blocks), dim3(256), 0, 0, table, nrec, 4, sink);  // warm-up
  hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(k_gather64, dim3(blocks), dim3(256), 0, 0, table, nrec, reads, sink);
  hipDeviceSynchronize();
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const double recs = (double)blocks * 256 * reads;
  printf("{\"table_GiB\": %.1f, \"records\": %.0f, \"seconds\": %.6f, \"algorithmic_GBs_at_64B\": %.1f, \"Grecords_per_s\": %.3f}\n", gib, recs, s,
         recs * 64 / s / 1e9, recs / s / 1e9);
  return 0;
}
