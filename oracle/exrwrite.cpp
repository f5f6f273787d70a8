// TEST INFRASTRUCTURE (oracle/): a driver around the reference's own vendored tinyexr (/root/reference/deps/tinyexr) that WRITES
// OpenEXR files — used to mint fixtures in compressions only tinyexr's encoder can produce here (PIZ) for the product's EXR
// reader (platinum_amd/csrc/scene_image.cpp).  Built only by `make -C oracle ref` into oracle/_ref/.  No reference source is copied.
//   exrwrite in.f32 W H C {half|float} {none|rle|zips|zip|piz} out.exr [TX TY]  (in.f32: H*W*C floats, C = 1 (Y), 3 (RGB) or 4 (RGBA);
//                                                                                TX TY: write a one-level TILED file with TX x TY tiles)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#include "tinyexr.h"

int main(int argc, char** argv) {
  if (argc != 8 && argc != 10) { fprintf(stderr, "usage: exrwrite in.f32 W H C half|float none|rle|zips|zip|piz out.exr [TX TY]\n"); return 2; }
  const int TX = argc == 10 ? atoi(argv[8]) : 0, TY = argc == 10 ? atoi(argv[9]) : 0;
  const int W = atoi(argv[2]), H = atoi(argv[3]), C = atoi(argv[4]);
  const bool half = !strcmp(argv[5], "half");
  const char* comps[] = {"none", "rle", "zips", "zip", "piz"};
  int comp = -1;
  for (int i = 0; i < 5; i++) if (!strcmp(argv[6], comps[i])) comp = i;
  if (W <= 0 || H <= 0 || (C != 1 && C != 3 && C != 4) || comp < 0) return 2;
  std::vector<float> in((size_t)W * H * C);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(in.data(), sizeof(float), in.size(), f) != in.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  fclose(f);
  // planar channel images in the order tinyexr wants them: alphabetical (A, B, G, R) / (B, G, R) / (Y)
  const char* names4[] = {"A", "B", "G", "R"}; const int src4[] = {3, 2, 1, 0};
  const char* names3[] = {"B", "G", "R"};      const int src3[] = {2, 1, 0};
  const char* names1[] = {"Y"};                const int src1[] = {0};
  const char** names = C == 4 ? names4 : C == 3 ? names3 : names1;
  const int* src = C == 4 ? src4 : C == 3 ? src3 : src1;
  std::vector<std::vector<float>> planes(C, std::vector<float>((size_t)W * H));
  for (int c = 0; c < C; c++) for (size_t i = 0; i < (size_t)W * H; i++) planes[c][i] = in[i * C + src[c]];
  std::vector<unsigned char*> ptrs(C);
  for (int c = 0; c < C; c++) ptrs[c] = (unsigned char*)planes[c].data();
  EXRHeader header; InitEXRHeader(&header);
  EXRImage image; InitEXRImage(&image);
  image.num_channels = C; image.images = ptrs.data(); image.width = W; image.height = H;
  header.num_channels = C;
  std::vector<EXRChannelInfo> ch(C);
  std::vector<int> pt(C, TINYEXR_PIXELTYPE_FLOAT), rpt(C, half ? TINYEXR_PIXELTYPE_HALF : TINYEXR_PIXELTYPE_FLOAT);
  for (int c = 0; c < C; c++) { memset(&ch[c], 0, sizeof(EXRChannelInfo)); strncpy(ch[c].name, names[c], 255); }
  header.channels = ch.data(); header.pixel_types = pt.data(); header.requested_pixel_types = rpt.data();
  header.compression_type = comp;
  // tiled: the picture cut into TX x TY tiles (tile images are TX x TY planes whatever the tile's actual size, row stride TX)
  std::vector<EXRTile> tiles;
  std::vector<std::vector<std::vector<float>>> tile_planes;
  std::vector<std::vector<unsigned char*>> tile_ptrs;
  if (TX > 0 && TY > 0) {
    const int nx = (W + TX - 1) / TX, ny = (H + TY - 1) / TY;
    tiles.resize((size_t)nx * ny); tile_planes.resize(tiles.size()); tile_ptrs.resize(tiles.size());
    for (int ty = 0; ty < ny; ty++)
      for (int tx = 0; tx < nx; tx++) {
        const size_t t = (size_t)ty * nx + tx;
        EXRTile& T = tiles[t];
        T.offset_x = tx; T.offset_y = ty; T.level_x = 0; T.level_y = 0;
        T.width = std::min(TX, W - tx * TX); T.height = std::min(TY, H - ty * TY);
        tile_planes[t].assign(C, std::vector<float>((size_t)TX * TY, 0.0f));
        tile_ptrs[t].resize(C);
        for (int c = 0; c < C; c++) {
          for (int y = 0; y < T.height; y++)
            for (int x = 0; x < T.width; x++) tile_planes[t][c][(size_t)y * TX + x] = planes[c][(size_t)(ty * TY + y) * W + tx * TX + x];
          tile_ptrs[t][c] = (unsigned char*)tile_planes[t][c].data();
        }
        T.images = tile_ptrs[t].data();
      }
    image.images = nullptr; image.tiles = tiles.data(); image.num_tiles = (int)tiles.size();
    header.tiled = 1; header.tile_size_x = TX; header.tile_size_y = TY;
    header.data_window.min_x = 0; header.data_window.min_y = 0; header.data_window.max_x = W - 1; header.data_window.max_y = H - 1;
    header.tile_level_mode = TINYEXR_TILE_ONE_LEVEL; header.tile_rounding_mode = TINYEXR_TILE_ROUND_DOWN;
  }
  const char* err = nullptr;
  if (SaveEXRImageToFile(&image, &header, argv[7], &err) != TINYEXR_SUCCESS) { fprintf(stderr, "SaveEXRImageToFile: %s\n", err ? err : "?"); return 1; }
  return 0;
}
