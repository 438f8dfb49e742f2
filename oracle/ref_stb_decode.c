/* Test infrastructure: a driver around the reference's own image decoder, src/stb_image.h (vendored stb_image
 * v2.15, the file behind stbi_load("earthmap.jpg", ...) in src/ptMain.cu:399).  The header is compiled from where
 * it lies under the reference tree (the Makefile passes its directory with -I; nothing of it is copied here) into
 * oracle/_ref/stb_decode, which tests/test_jpeg.py runs to hold pt_image_decode_jpeg to the reference's bytes.
 *     stb_decode in.jpg out.rgb      writes nx*ny*channels bytes, prints "nx ny channels" */
#define STB_IMAGE_IMPLEMENTATION
#define STBI_ONLY_JPEG
#include "stb_image.h"
#include <stdio.h>

int main(int argc, char** argv)
{
    int nx = 0, ny = 0, nn = 0;
    unsigned char* texels;
    FILE* out;
    if (argc != 3) { fprintf(stderr, "usage: stb_decode in.jpg out.raw\n"); return 2; }
    texels = stbi_load(argv[1], &nx, &ny, &nn, 0);          /* the reference's call: no forced channel count */
    if (!texels) { fprintf(stderr, "stbi_load: %s\n", stbi_failure_reason()); return 1; }
    out = fopen(argv[2], "wb");
    if (!out || fwrite(texels, 1, (size_t)nx * ny * nn, out) != (size_t)nx * ny * nn) { fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
    fclose(out);
    printf("%d %d %d\n", nx, ny, nn);
    return 0;
}
