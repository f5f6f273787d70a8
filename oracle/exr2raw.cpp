// TEST INFRASTRUCTURE (oracle/): a 30-line driver around the reference's own vendored tinyexr
// (/root/reference/deps/tinyexr, miniz) used once, in the authoring container, to decode the
// reference's LUT data files exactly the way the reference does
// (renderer_pt.cpp:385-446: LoadEXR -> RGBA floats, channel read at [4*i+3]).
// Built only by `make -C oracle ref` into oracle/_ref/ (git-ignored). No reference source is copied.
#include <cstdio>
#include <cstdlib>
#include "tinyexr.h"

int main(int argc, char** argv) {
  if (argc != 3 && argc != 4) { fprintf(stderr, "usage: exr2raw in.exr out.f32 [rgba]\n"); return 2; }
  float* rgba = nullptr; int w = 0, h = 0; const char* err = nullptr;
  int r = LoadEXR(&rgba, &w, &h, argv[1], &err);
  if (r < 0) { fprintf(stderr, "LoadEXR failed: %s\n", err ? err : "?"); return 1; }
  FILE* f = fopen(argv[2], "wb");
  if (!f) return 1;
  int hdr[2] = {w, h};
  fwrite(hdr, sizeof(int), 2, f);
  if (argc == 4) fwrite(rgba, sizeof(float), (size_t)w * h * 4, f);  // all four channels: the environment-map path (loaders/texture.cpp:93)
  else for (int i = 0; i < w * h; i++) fwrite(&rgba[4 * i + 3], sizeof(float), 1, f);  // renderer_pt.cpp:405-407
  fclose(f);
  free(rgba);
  return 0;
}
