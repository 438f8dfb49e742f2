// Image writers chosen by file extension, as the reference's writeImage does
// (src/ptMain.cu:432-487): .ppm (ASCII P3), .hdr (Radiance RGBE), .png, .tga, .bmp.
#ifndef PT_IMAGE_IO_H
#define PT_IMAGE_IO_H

#include <string>

namespace pt {

// rgb: nx*ny*3 floats, row 0 = top.  Returns false (with *err set) on an unknown extension or
// an I/O failure.  8-bit formats quantise as the reference does: Clamp(int(255.99 * c), 0, 255).
bool write_image(const std::string& path, const float* rgb, int nx, int ny, std::string* err);

}  // namespace pt

#endif
