// Self-contained image writers (the reference delegates to stb_image_write, which is not part of
// this build).  The PNG writer emits stored (uncompressed) deflate blocks: valid, just not small.
#include "pt_image_io.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

namespace {

inline int quant(float c)              // src/ptMain.cu:450-452: int(255.99 * col), clamped
{
    int v = (int)(255.99 * (double)c);
    return v > 255 ? 255 : (v < 0 ? 0 : v);
}

std::vector<uint8_t> to_rgb8(const float* rgb, int nx, int ny)
{
    std::vector<uint8_t> out((size_t)nx * ny * 3);
    for (size_t i = 0; i < out.size(); i++) out[i] = (uint8_t)quant(rgb[i]);
    return out;
}

bool write_all(const std::string& path, const std::vector<uint8_t>& bytes, std::string* err)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { *err = "cannot open " + path; return false; }
    const bool ok = std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    std::fclose(f);
    if (!ok) *err = "short write to " + path;
    return ok;
}

void put32be(std::vector<uint8_t>& v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s)); }
void put16le(std::vector<uint8_t>& v, uint32_t x) { v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); }
void put32le(std::vector<uint8_t>& v, uint32_t x) { for (int s = 0; s < 32; s += 8) v.push_back((uint8_t)(x >> s)); }

uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc = 0)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 255] ^ (crc >> 8);
    return ~crc;
}

void png_chunk(std::vector<uint8_t>& out, const char* tag, const std::vector<uint8_t>& data)
{
    put32be(out, (uint32_t)data.size());
    std::vector<uint8_t> body(tag, tag + 4);
    body.insert(body.end(), data.begin(), data.end());
    out.insert(out.end(), body.begin(), body.end());
    put32be(out, crc32(body.data(), body.size()));
}

bool write_png(const std::string& path, const std::vector<uint8_t>& rgb, int nx, int ny, std::string* err)
{
    std::vector<uint8_t> raw;                       // filter byte 0 + row
    raw.reserve((size_t)ny * (3 * nx + 1));
    for (int y = 0; y < ny; y++) { raw.push_back(0); raw.insert(raw.end(), rgb.begin() + (size_t)y * nx * 3, rgb.begin() + (size_t)(y + 1) * nx * 3); }
    std::vector<uint8_t> z = { 0x78, 0x01 };        // zlib header, then stored blocks of <= 65535 bytes
    size_t pos = 0;
    do {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        put16le(z, (uint32_t)n); put16le(z, (uint32_t)(~n & 0xffff));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
    } while (pos < raw.size());
    uint32_t a = 1, b = 0;                          // adler32
    for (uint8_t c : raw) { a = (a + c) % 65521; b = (b + a) % 65521; }
    put32be(z, (b << 16) | a);
    std::vector<uint8_t> out = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    std::vector<uint8_t> ihdr;
    put32be(ihdr, (uint32_t)nx); put32be(ihdr, (uint32_t)ny);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    png_chunk(out, "IHDR", ihdr);
    png_chunk(out, "IDAT", z);
    png_chunk(out, "IEND", {});
    return write_all(path, out, err);
}

bool write_tga(const std::string& path, const std::vector<uint8_t>& rgb, int nx, int ny, std::string* err)
{
    std::vector<uint8_t> out(18, 0);
    out[2] = 2;                                     // uncompressed true colour
    out[12] = (uint8_t)nx; out[13] = (uint8_t)(nx >> 8); out[14] = (uint8_t)ny; out[15] = (uint8_t)(ny >> 8);
    out[16] = 24; out[17] = 0x20;                   // top-left origin
    for (size_t i = 0; i < (size_t)nx * ny; i++) { out.push_back(rgb[3 * i + 2]); out.push_back(rgb[3 * i + 1]); out.push_back(rgb[3 * i]); }
    return write_all(path, out, err);
}

bool write_bmp(const std::string& path, const std::vector<uint8_t>& rgb, int nx, int ny, std::string* err)
{
    const int pad = (4 - (nx * 3) % 4) % 4;
    const uint32_t img = (uint32_t)(nx * 3 + pad) * ny;
    std::vector<uint8_t> out = { 'B', 'M' };
    put32le(out, 54 + img); put32le(out, 0); put32le(out, 54);
    put32le(out, 40); put32le(out, (uint32_t)nx); put32le(out, (uint32_t)ny); put16le(out, 1); put16le(out, 24);
    put32le(out, 0); put32le(out, img); put32le(out, 2835); put32le(out, 2835); put32le(out, 0); put32le(out, 0);
    for (int y = ny - 1; y >= 0; y--) {             // bottom-up rows, BGR
        for (int x = 0; x < nx; x++) { const uint8_t* p = &rgb[((size_t)y * nx + x) * 3]; out.push_back(p[2]); out.push_back(p[1]); out.push_back(p[0]); }
        for (int k = 0; k < pad; k++) out.push_back(0);
    }
    return write_all(path, out, err);
}

bool write_hdr(const std::string& path, const float* rgb, int nx, int ny, std::string* err)
{
    std::string head = "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y " + std::to_string(ny) + " +X " + std::to_string(nx) + "\n";
    std::vector<uint8_t> out(head.begin(), head.end());
    for (size_t i = 0; i < (size_t)nx * ny; i++) {   // flat (non run-length) RGBE pixels
        const float r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
        const float m = std::fmax(r, std::fmax(g, b));
        if (!(m > 1e-32f)) { out.insert(out.end(), { 0, 0, 0, 0 }); continue; }
        int e;
        const float scale = std::frexp(m, &e) * 256.0f / m;
        out.push_back((uint8_t)(r * scale)); out.push_back((uint8_t)(g * scale)); out.push_back((uint8_t)(b * scale)); out.push_back((uint8_t)(e + 128));
    }
    return write_all(path, out, err);
}

}  // namespace

namespace pt {

bool write_image(const std::string& path, const float* rgb, int nx, int ny, std::string* err)
{
    const size_t dot = path.rfind('.');
    const std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
    if (ext == "ppm") {
        FILE* f = std::fopen(path.c_str(), "w");
        if (!f) { *err = "cannot open " + path; return false; }
        std::fprintf(f, "P3\n%d %d\n255\n", nx, ny);
        for (size_t i = 0; i < (size_t)nx * ny; i++)
            std::fprintf(f, "%d %d %d\n", quant(rgb[3 * i]), quant(rgb[3 * i + 1]), quant(rgb[3 * i + 2]));
        std::fclose(f);
        return true;
    }
    if (ext == "hdr") return write_hdr(path, rgb, nx, ny, err);
    const std::vector<uint8_t> rgb8 = to_rgb8(rgb, nx, ny);
    if (ext == "png") return write_png(path, rgb8, nx, ny, err);
    if (ext == "tga") return write_tga(path, rgb8, nx, ny, err);
    if (ext == "bmp") return write_bmp(path, rgb8, nx, ny, err);
    *err = "unknown image extension '" + ext + "' (ppm, hdr, png, tga, bmp)";
    return false;
}

}  // namespace pt
