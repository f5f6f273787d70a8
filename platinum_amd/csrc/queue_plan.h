// queue_plan.h — how pt_start_render sizes the wavefront queues (pure host arithmetic; exported as pt_plan_queues for tests).
//
// A SEGMENT is the queue share of `tiles_per_seg` 8x8 pixel tiles under all `samples_in_flight` samples of a batch
// (kernels.hip).  What bounds the numbers:
//   * slot numbers inside a segment travel through k_shade's per-wave class bins as 16-bit values  -> seg_cap <= 65536
//   * a path's radiance-buffer entry relative to its segment rides in 21 bits of rayD.w            -> seg_cap <= 2^21 (implied)
//   * the chunk tables pack (chunk << 16) | segment                                                 -> nseg <= 65536, chunks < 65536
//   * queue slots and the per-sample radiance buffer are indexed with 32 bits                       -> tiles * 64 * samples < 2^31
// When a limit would be exceeded the samples in flight are halved (fewer samples per batch, same image); only an image that
// does not fit with ONE sample in flight is refused.
#pragma once
#include <algorithm>
#include <cstdint>

#include "../../include/ptamd.h"

namespace pt {

constexpr uint32_t kMaxSegments = 32768;      // chunk tables: 16 bits of segment id; k_chunk_tables' per-block slice <= 1024 segments
constexpr uint32_t kMaxSegmentSlots = 65536;  // k_shade: uint16_t slot numbers in the class bins
constexpr uint32_t kSegGroupChunks = 16;      // kernels.hip PT_SEG_GROUP

// returns PT_OK, or PT_ERR_INVALID_ARGUMENT / PT_ERR_UNSUPPORTED with *why set
inline int plan_queues(uint32_t width, uint32_t height, uint32_t spp, uint32_t samples_in_flight, uint64_t free_hbm_bytes,
                       uint32_t tiles_per_seg_override, uint32_t seg_bands, pt_queue_plan* out, const char** why) {
  *why = "";
  if (width == 0 || height == 0 || spp == 0) { *why = "empty size or spp"; return PT_ERR_INVALID_ARGUMENT; }
  if ((uint64_t)width * height > (1ull << 28)) { *why = "image too large"; return PT_ERR_INVALID_ARGUMENT; }
  if (seg_bands == 0) seg_bands = 1;
  const uint64_t npix = (uint64_t)width * height;
  const uint64_t tiles = (uint64_t)((width + 7) / 8) * ((height + 7) / 8);
  uint32_t sif = samples_in_flight;
  if (sif == 0) {
    // As many samples of the frame in flight as a quarter of the free HBM holds (~200 B of queue state per path), up to 128:
    // the deep bounces of a batch carry few rays, and only a big batch keeps those launches wide (128 x 1080p = 53 GB of the
    // 288; measured on MI355X, Msamples/s at 64 / 128 / 256 in flight: C3 8175 / 8323 / 8376, C2 15948 / 16606 / 16753).
    if (free_hbm_bytes == 0) free_hbm_bytes = 8ull << 30;
    sif = (uint32_t)std::min<uint64_t>(128, std::max<uint64_t>(1, (free_hbm_bytes / 4) / (npix * 200ull)));
  }
  sif = std::min<uint32_t>(std::min<uint32_t>(sif, spp), 256);
  while (sif > 1 && tiles * 64 * sif >= (1ull << 31)) sif /= 2;
  if (tiles * 64 * sif >= (1ull << 31)) { *why = "the image is too large for the 32-bit queue indices"; return PT_ERR_UNSUPPORTED; }
  uint32_t tps = (uint32_t)((tiles + (kMaxSegments - 128) - 1) / (kMaxSegments - 128));  // nseg <= 32768 after rounding up to the band count
  if (tiles_per_seg_override) tps = std::max(tps, tiles_per_seg_override);
  while (sif > 1 && (uint64_t)tps * sif * 64 > kMaxSegmentSlots) sif /= 2;
  if ((uint64_t)tps * sif * 64 > kMaxSegmentSlots) { *why = "tiles per segment too large for the 16-bit segment slots"; return PT_ERR_UNSUPPORTED; }
  uint32_t nseg = (uint32_t)((tiles + tps - 1) / tps);
  nseg = (nseg + seg_bands - 1) / seg_bands * seg_bands;  // (segments past the last tile stay empty)
  if (nseg > kMaxSegments) { *why = "too many segments for the chunk tables"; return PT_ERR_UNSUPPORTED; }
  out->samples_in_flight = sif;
  out->tiles_per_seg = tps;
  out->nseg = nseg;
  out->seg_cap = tps * sif * 64;
  const uint64_t K = out->seg_cap / 64;
  out->capacity = (uint64_t)nseg * ((K + kSegGroupChunks - 1) / kSegGroupChunks * kSegGroupChunks) * 64;  // >= npix * sif
  out->lbuf_entries = tiles * 64 * sif;
  return PT_OK;
}

}  // namespace pt
