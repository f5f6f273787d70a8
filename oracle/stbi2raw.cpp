// TEST INFRASTRUCTURE — decodes an image with the REFERENCE's own vendored stb_image (deps/stb_image, compiled from where it
// lies by `make -C oracle ref`) exactly as loaders/texture.cpp:111-119 calls it (4 components) and writes
// "W H\n" + raw RGBA8 to stdout.  The oracle for platinum_amd/csrc/scene_jpeg.cpp (and the PNG decoder).
#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h"
#include <cstdio>
int main(int argc, char** argv) {
  if (argc < 2) return 2;
  int w = 0, h = 0;
  unsigned char* px = stbi_load(argv[1], &w, &h, nullptr, 4);
  if (!px) { fprintf(stderr, "stbi_load failed: %s\n", stbi_failure_reason()); return 1; }
  printf("%d %d\n", w, h);
  fwrite(px, 1, (size_t)w * h * 4, stdout);
  return 0;
}
