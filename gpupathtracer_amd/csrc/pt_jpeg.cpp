// Baseline JPEG decoder for image textures: the reference loads its one texture, earthmap.jpg, with
// stbi_load("earthmap.jpg", &nx, &ny, &nz, 0) (src/ptMain.cu:396-402, vendored src/stb_image.h v2.15), and the
// texels decide what ImageTexture::value returns, so this decoder is held to that decoder's output byte for byte
// (tests/test_jpeg.py compares the two on generated files and on the reference's own file where it is present).
// The arithmetic that fixes the bytes, restated from the published algorithms:
//   * coefficients are de-quantised as 16-bit values in natural order (src/stb_image.h:1905-1958);
//   * the inverse DCT is the Independent JPEG Group's jidctint "islow" (Loeffler-Ligtenberg-Moschytz, 11
//     multiplies per 1-D pass) with 12-bit constants rounded as (int)(c * 4096 + 0.5), two extra bits kept
//     between the column and the row pass, +128 folded into the last rounding (src/stb_image.h:2114-2212);
//   * chroma is enlarged by the triangle filters 3:1 (one direction) and 9:3:3:1 (both), nearest beyond a
//     factor of two, row pairs chosen as in src/stb_image.h:3110-3186,3554-3596;
//   * YCbCr -> RGB in 20-bit fixed point with the constants' low 8 bits dropped and the Cb term of green
//     truncated to 16 bits (src/stb_image.h:3316-3342).
// Sequential Huffman frames (SOF0 / SOF1, 8-bit samples), 1 or 3 components, any sampling factors up to 4,
// restart intervals, interleaved or one-component scans.  Progressive, arithmetic-coded and 12-bit files are
// refused with a message (the reference's decoder reads progressive files; its one asset is not one).
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ptcore.h"
#include "pt_host.h"

namespace {

constexpr uint8_t ZIGZAG[64] = {                       // coefficient k of the stream -> place in the 8x8 block (ITU T.81 figure A.6)
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

struct Huffman {                                       // ITU T.81 annex C / F.2.2.3: canonical codes by length
    bool present = false;
    uint8_t symbols[256] = {};
    int first_code[18] = {}, first_index[18] = {}, count[18] = {};
    bool build(const uint8_t counts[16], const uint8_t* syms, int n_syms)
    {
        int code = 0, index = 0;
        for (int len = 1; len <= 16; len++) {
            count[len] = counts[len - 1];
            first_code[len] = code;
            first_index[len] = index;
            code += count[len];
            index += count[len];
            if (code > (1 << len)) return false;       // more codes of this length than there are
            code <<= 1;
        }
        if (index != n_syms || index > 256) return false;
        std::memcpy(symbols, syms, (size_t)n_syms);
        present = true;
        return true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int w = 0, hgt = 0;                                // samples the image needs of this component
    int stride = 0, rows = 0;                          // the decoded plane, whole blocks of whole MCUs
    int dc_pred = 0;
    std::vector<uint8_t> plane;
};

struct Bits {                                          // entropy-coded segment: a stuffed 0xFF00 is a data byte 0xFF, any other marker ends the data
    const uint8_t* p; const uint8_t* end;
    uint32_t acc = 0; int n = 0;
    int marker = 0;                                    // the marker that ended the data, once met; zeros are fed from then on
    void fill()
    {
        while (n <= 24) {
            int byte = 0;
            if (!marker && p < end) {
                byte = *p++;
                if (byte == 0xFF) {
                    int next = p < end ? *p++ : 0xD9;
                    while (next == 0xFF && p < end) next = *p++;     // fill bytes before a marker
                    if (next != 0) { marker = next; byte = 0; }
                }
            }
            acc |= (uint32_t)byte << (24 - n);
            n += 8;
        }
    }
    int get(int k)                                     // k <= 16
    {
        if (k == 0) return 0;
        if (n < k) fill();
        const int v = (int)(acc >> (32 - k));
        acc <<= k; n -= k;
        return v;
    }
    void reset() { acc = 0; n = 0; }
};

int decode_symbol(Bits& b, const Huffman& h)
{
    int code = 0;
    for (int len = 1; len <= 16; len++) {
        code = (code << 1) | b.get(1);
        if (code - h.first_code[len] < h.count[len] && code >= h.first_code[len]) return h.symbols[h.first_index[len] + code - h.first_code[len]];
    }
    return -1;
}
// T.81 F.2.2.1: the value of `s` received bits
int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

constexpr int fix12(float c) { return (int)(c * 4096 + 0.5); }
struct Idct8 { int even_a, even_b, even_c, even_d, odd_a, odd_b, odd_c, odd_d; };
// One 1-D pass of the IJG islow inverse DCT, constants scaled by 2^12: returns the four even-part sums (rows 0..3
// take even + odd, rows 7..4 even - odd) and the four odd-part sums.
inline Idct8 idct_1d(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7)
{
    Idct8 r;
    // even part: a rotation of (s2, s6), a butterfly of (s0, s4)
    const int z = (s2 + s6) * fix12(0.5411961f);
    const int rot6 = z + s6 * fix12(-1.847759065f);
    const int rot2 = z + s2 * fix12(0.765366865f);
    const int sum04 = (s0 + s4) << 12, dif04 = (s0 - s4) << 12;
    r.even_a = sum04 + rot2; r.even_d = sum04 - rot2;
    r.even_b = dif04 + rot6; r.even_c = dif04 - rot6;
    // odd part: s7, s5, s3, s1
    const int z1 = s7 + s1, z2 = s5 + s3, z3 = s7 + s3, z4 = s5 + s1;
    const int z5 = (z3 + z4) * fix12(1.175875602f);
    const int m7 = s7 * fix12(0.298631336f), m5 = s5 * fix12(2.053119869f), m3 = s3 * fix12(3.072711026f), m1 = s1 * fix12(1.501321110f);
    const int y1 = z5 + z1 * fix12(-0.899976223f), y2 = z5 + z2 * fix12(-2.562915447f);
    const int y3 = z3 * fix12(-1.961570560f), y4 = z4 * fix12(-0.390180644f);
    r.odd_a = m1 + (y1 + y4);                          // pairs with even_a (rows 0 and 7)
    r.odd_b = m3 + (y2 + y3);                          // rows 1 and 6
    r.odd_c = m5 + (y2 + y4);                          // rows 2 and 5
    r.odd_d = m7 + (y1 + y3);                          // rows 3 and 4
    return r;
}
inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
void idct_block(const int16_t* c, uint8_t* out, int stride)
{
    int mid[64];
    for (int x = 0; x < 8; x++) {                      // columns: 2^12 from the constants, two bits kept
        const Idct8 r = idct_1d(c[x], c[8 + x], c[16 + x], c[24 + x], c[32 + x], c[40 + x], c[48 + x], c[56 + x]);
        const int a = r.even_a + 512, b = r.even_b + 512, cc = r.even_c + 512, d = r.even_d + 512;
        mid[x] = (a + r.odd_a) >> 10;       mid[56 + x] = (a - r.odd_a) >> 10;
        mid[8 + x] = (b + r.odd_b) >> 10;   mid[48 + x] = (b - r.odd_b) >> 10;
        mid[16 + x] = (cc + r.odd_c) >> 10; mid[40 + x] = (cc - r.odd_c) >> 10;
        mid[24 + x] = (d + r.odd_d) >> 10;  mid[32 + x] = (d - r.odd_d) >> 10;
    }
    for (int y = 0; y < 8; y++) {                      // rows: 2^12, the two kept bits and 2^3 of the transform's own scale; +128 before the shift
        const int* m = mid + 8 * y;
        const Idct8 r = idct_1d(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
        const int bias = 65536 + (128 << 17);
        const int a = r.even_a + bias, b = r.even_b + bias, cc = r.even_c + bias, d = r.even_d + bias;
        uint8_t* o = out + (size_t)stride * y;
        o[0] = clamp8((a + r.odd_a) >> 17);  o[7] = clamp8((a - r.odd_a) >> 17);
        o[1] = clamp8((b + r.odd_b) >> 17);  o[6] = clamp8((b - r.odd_b) >> 17);
        o[2] = clamp8((cc + r.odd_c) >> 17); o[5] = clamp8((cc - r.odd_c) >> 17);
        o[3] = clamp8((d + r.odd_d) >> 17);  o[4] = clamp8((d - r.odd_d) >> 17);
    }
}

struct Decoder {
    const uint8_t* data; size_t size; size_t pos = 0;
    std::string err;
    uint16_t quant[4][64] = {};
    bool have_quant[4] = {};
    Huffman dc[4], ac[4];
    int nx = 0, ny = 0, n_comp = 0, h_max = 1, v_max = 1, mcus_x = 0, mcus_y = 0, restart_interval = 0;
    bool have_frame = false, jfif = false;
    int adobe_transform = -1;
    Component comp[3];

    bool fail(const char* what) { if (err.empty()) err = what; return false; }
    int u8() { return pos < size ? data[pos++] : 0; }
    int u16() { const int a = u8(); return (a << 8) | u8(); }

    bool read_dqt(int len)
    {
        while (len > 0) {
            const int pq_tq = u8(), wide = pq_tq >> 4, t = pq_tq & 15;
            if (wide > 1 || t > 3) return fail("JPEG: bad quantisation table header");
            for (int k = 0; k < 64; k++) quant[t][ZIGZAG[k]] = (uint16_t)(wide ? u16() : u8());
            have_quant[t] = true;
            len -= 1 + (wide ? 128 : 64);
        }
        return len == 0 || fail("JPEG: quantisation table segment of odd length");
    }
    bool read_dht(int len)
    {
        while (len > 0) {
            const int tc_th = u8(), cls = tc_th >> 4, t = tc_th & 15;
            if (cls > 1 || t > 3) return fail("JPEG: bad Huffman table header");
            uint8_t counts[16]; int total = 0;
            for (int k = 0; k < 16; k++) { counts[k] = (uint8_t)u8(); total += counts[k]; }
            if (total > 256 || pos + (size_t)total > size) return fail("JPEG: Huffman table too long");
            if (!(cls ? ac[t] : dc[t]).build(counts, data + pos, total)) return fail("JPEG: Huffman table is not a prefix code");
            pos += (size_t)total;
            len -= 17 + total;
        }
        return len == 0 || fail("JPEG: Huffman table segment of odd length");
    }
    bool read_sof(int len)
    {
        if (have_frame) return fail("JPEG: more than one frame");
        if (u8() != 8) return fail("JPEG: only 8-bit samples are read");
        ny = u16(); nx = u16(); n_comp = u8();
        if (nx <= 0 || ny <= 0) return fail("JPEG: empty image");
        if ((int64_t)nx * ny > (int64_t)1 << 28) return fail("JPEG: image too large");
        if (n_comp != 1 && n_comp != 3) return fail("JPEG: 1 or 3 components are read (no CMYK)");
        if (len != 8 + 3 * n_comp - 2) return fail("JPEG: frame header of odd length");
        for (int k = 0; k < n_comp; k++) {
            Component& c = comp[k];
            c.id = u8();
            const int hv = u8(); c.h = hv >> 4; c.v = hv & 15; c.tq = u8();
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail("JPEG: bad sampling factors");
            h_max = c.h > h_max ? c.h : h_max; v_max = c.v > v_max ? c.v : v_max;
        }
        for (int k = 0; k < n_comp; k++) {
            if (h_max % comp[k].h || v_max % comp[k].v) return fail("JPEG: sampling factors that do not divide the largest are not read");
        }
        mcus_x = (nx + 8 * h_max - 1) / (8 * h_max); mcus_y = (ny + 8 * v_max - 1) / (8 * v_max);
        for (int k = 0; k < n_comp; k++) {
            Component& c = comp[k];
            c.w = (nx * c.h + h_max - 1) / h_max; c.hgt = (ny * c.v + v_max - 1) / v_max;
            c.stride = mcus_x * c.h * 8; c.rows = mcus_y * c.v * 8;
            c.plane.assign((size_t)c.stride * c.rows, 0);
        }
        have_frame = true;
        return true;
    }

    bool decode_block(Bits& b, Component& c, int16_t* coef)
    {
        std::memset(coef, 0, 64 * sizeof(int16_t));
        const uint16_t* q = quant[c.tq];
        const int t = decode_symbol(b, dc[c.td]);
        if (t < 0 || t > 15) return fail("JPEG: bad DC code");
        const int diff = t ? extend(b.get(t), t) : 0;
        c.dc_pred += diff;
        coef[0] = (int16_t)(c.dc_pred * q[0]);
        for (int k = 1; k < 64;) {
            const int rs = decode_symbol(b, ac[c.ta]);
            if (rs < 0) return fail("JPEG: bad AC code");
            const int run = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (rs != 0xF0) break;                 // end of block
                k += 16;
            } else {
                k += run;
                if (k > 63) return fail("JPEG: coefficient run past the block");
                const int z = ZIGZAG[k++];
                coef[z] = (int16_t)(extend(b.get(s), s) * q[z]);
            }
        }
        return true;
    }

    bool read_scan(int len)
    {
        if (!have_frame) return fail("JPEG: scan before frame");
        const int ns = u8();
        if (ns < 1 || ns > n_comp || len != 4 + 2 * ns) return fail("JPEG: bad scan header");
        int order[3];
        for (int k = 0; k < ns; k++) {
            const int id = u8(), tt = u8();
            int which = -1;
            for (int j = 0; j < n_comp; j++) if (comp[j].id == id) which = j;
            if (which < 0) return fail("JPEG: scan names an unknown component");
            comp[which].td = tt >> 4; comp[which].ta = tt & 15;
            if (comp[which].td > 3 || comp[which].ta > 3 || !dc[comp[which].td].present || !ac[comp[which].ta].present) return fail("JPEG: scan names a missing Huffman table");
            if (!have_quant[comp[which].tq]) return fail("JPEG: scan names a missing quantisation table");
            order[k] = which;
        }
        const int ss = u8(), se = u8(), ahal = u8();
        if (ss != 0 || se != 63 || ahal != 0) return fail("JPEG: not a sequential scan");
        Bits b; b.p = data + pos; b.end = data + size;
        for (int k = 0; k < n_comp; k++) comp[k].dc_pred = 0;
        int16_t coef[64];
        int todo = restart_interval ? restart_interval : 0x7fffffff;
        // false: a restart marker was due and is not there -- the scan ends here, as it does in the reference's decoder
        auto after_mcu = [&]() -> bool {
            if (--todo > 0) return true;
            if (b.n < 24) b.fill();                    // bits left in the buffer are padding
            if (b.marker < 0xD0 || b.marker > 0xD7) return false;
            b.marker = 0; b.reset();
            for (int k = 0; k < n_comp; k++) comp[k].dc_pred = 0;
            todo = restart_interval;
            return true;
        };
        bool more = true;
        if (ns == 1) {                                 // one component: its own blocks in raster order, only those the image needs
            Component& c = comp[order[0]];
            const int bw = (c.w + 7) >> 3, bh = (c.hgt + 7) >> 3;
            for (int by = 0; by < bh && more; by++)
                for (int bx = 0; bx < bw && more; bx++) {
                    if (!decode_block(b, c, coef)) return false;
                    idct_block(coef, c.plane.data() + (size_t)c.stride * by * 8 + bx * 8, c.stride);
                    more = after_mcu();
                }
        } else {
            for (int my = 0; my < mcus_y && more; my++)
                for (int mx = 0; mx < mcus_x && more; mx++) {
                    for (int k = 0; k < ns; k++) {
                        Component& c = comp[order[k]];
                        for (int y = 0; y < c.v; y++)
                            for (int x = 0; x < c.h; x++) {
                                if (!decode_block(b, c, coef)) return false;
                                idct_block(coef, c.plane.data() + (size_t)c.stride * (my * c.v + y) * 8 + (mx * c.h + x) * 8, c.stride);
                            }
                    }
                    more = after_mcu();
                }
        }
        // carry on behind the entropy-coded data: at the marker that ended it, or behind the bytes consumed
        pos = (size_t)(b.p - data);
        if (b.marker) pos -= 2;
        return true;
    }

    bool parse()
    {
        if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail("JPEG: no start-of-image marker");
        pos = 2;
        bool scanned = false;
        while (pos + 4 <= size) {
            if (data[pos] != 0xFF) { pos++; continue; }     // between segments: skip to the next marker
            const int m = data[pos + 1];
            if (m == 0xFF) { pos++; continue; }
            pos += 2;
            if (m == 0xD9) break;
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7) || m == 0x00) continue;
            const int len = u16() - 2;
            if (len < 0 || pos + (size_t)len > size) return fail("JPEG: segment runs past the end of the file");
            const size_t next = pos + (size_t)len;
            bool ok = true;
            switch (m) {
            case 0xDB: ok = read_dqt(len); break;
            case 0xC4: ok = read_dht(len); break;
            case 0xC0: case 0xC1: ok = read_sof(len); break;
            case 0xC2: return fail("JPEG: progressive files are not read; re-save the texture as a baseline JPEG or a binary PPM");
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                return fail("JPEG: lossless, hierarchical and arithmetic-coded files are not read");
            case 0xDD: if (len != 2) return fail("JPEG: bad restart interval segment"); restart_interval = u16(); break;
            case 0xE0: jfif = len >= 5 && std::memcmp(data + pos, "JFIF\0", 5) == 0; break;
            case 0xEE: if (len >= 12 && std::memcmp(data + pos, "Adobe", 5) == 0) adobe_transform = data[pos + 11]; break;
            case 0xDA:
                if (!read_scan(len)) return false;
                scanned = true;
                continue;                              // read_scan left pos behind the entropy-coded data
            default: break;                            // comments, application data, anything else with a length
            }
            if (!ok) return false;
            pos = next;
        }
        if (!have_frame || !scanned) return fail("JPEG: no image data");
        return true;
    }

    // Enlarging one component to the image's width, one output row at a time.
    static void widen(const Component& c, int hs, int vs, const uint8_t* near, const uint8_t* far, int w_lo, uint8_t* out)
    {
        if (hs == 1 && vs == 1) { std::memcpy(out, near, (size_t)w_lo); return; }
        if (hs == 1 && vs == 2) { for (int i = 0; i < w_lo; i++) out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2); return; }
        if (hs == 2 && vs == 1) {
            if (w_lo == 1) { out[0] = out[1] = near[0]; return; }
            out[0] = near[0];
            out[1] = (uint8_t)((near[0] * 3 + near[1] + 2) >> 2);
            for (int i = 1; i < w_lo - 1; i++) {
                const int n = 3 * near[i] + 2;
                out[2 * i] = (uint8_t)((n + near[i - 1]) >> 2);
                out[2 * i + 1] = (uint8_t)((n + near[i + 1]) >> 2);
            }
            out[2 * (w_lo - 1)] = (uint8_t)((near[w_lo - 2] * 3 + near[w_lo - 1] + 2) >> 2);
            out[2 * (w_lo - 1) + 1] = near[w_lo - 1];
            return;
        }
        if (hs == 2 && vs == 2) {
            if (w_lo == 1) { out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2); return; }
            int cur = 3 * near[0] + far[0];
            out[0] = (uint8_t)((cur + 2) >> 2);
            for (int i = 1; i < w_lo; i++) {
                const int before = cur;
                cur = 3 * near[i] + far[i];
                out[2 * i - 1] = (uint8_t)((3 * before + cur + 8) >> 4);
                out[2 * i] = (uint8_t)((3 * cur + before + 8) >> 4);
            }
            out[2 * w_lo - 1] = (uint8_t)((cur + 2) >> 2);
            return;
        }
        for (int i = 0; i < w_lo; i++) for (int j = 0; j < hs; j++) out[i * hs + j] = near[i];       // other factors: nearest
        (void)c;
    }

    bool to_rgb(std::vector<uint8_t>* rgb)
    {
        rgb->assign((size_t)3 * nx * ny, 0);
        // three components that are not YCbCr: Adobe says "no transform", or no JFIF header and component ids R, G, B
        const bool plain_rgb = n_comp == 3 && ((comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') || (adobe_transform == 0 && !jfif));
        struct Row { int hs, vs, step, w_lo, y; const uint8_t* line0; const uint8_t* line1; std::vector<uint8_t> buf; } rows[3];
        for (int k = 0; k < n_comp; k++) {
            Row& r = rows[k];
            r.hs = h_max / comp[k].h; r.vs = v_max / comp[k].v;
            r.step = r.vs >> 1; r.w_lo = (nx + r.hs - 1) / r.hs; r.y = 0;
            r.line0 = r.line1 = comp[k].plane.data();
            r.buf.assign((size_t)nx + 8, 0);
        }
        constexpr int c_cr_r = ((int)(1.40200f * 4096.0f + 0.5f)) << 8, c_cr_g = ((int)(0.71414f * 4096.0f + 0.5f)) << 8,
                      c_cb_g = ((int)(0.34414f * 4096.0f + 0.5f)) << 8, c_cb_b = ((int)(1.77200f * 4096.0f + 0.5f)) << 8;
        for (int j = 0; j < ny; j++) {
            const uint8_t* line[3] = { nullptr, nullptr, nullptr };
            for (int k = 0; k < n_comp; k++) {
                Row& r = rows[k];
                const bool lower = r.step >= (r.vs >> 1);      // which of the two source rows is the nearer one
                widen(comp[k], r.hs, r.vs, lower ? r.line1 : r.line0, lower ? r.line0 : r.line1, r.w_lo, r.buf.data());
                line[k] = r.buf.data();
                if (++r.step >= r.vs) {
                    r.step = 0;
                    r.line0 = r.line1;
                    if (++r.y < comp[k].hgt) r.line1 += comp[k].stride;
                }
            }
            uint8_t* out = rgb->data() + (size_t)3 * nx * j;
            if (n_comp == 1) {
                for (int i = 0; i < nx; i++) out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = line[0][i];
            } else if (plain_rgb) {
                for (int i = 0; i < nx; i++) { out[3 * i] = line[0][i]; out[3 * i + 1] = line[1][i]; out[3 * i + 2] = line[2][i]; }
            } else {
                for (int i = 0; i < nx; i++) {
                    const int32_t y_fixed = ((int32_t)line[0][i] << 20) + (1 << 19);
                    const int32_t cb = (int32_t)line[1][i] - 128, cr = (int32_t)line[2][i] - 128;
                    int32_t r = y_fixed + cr * c_cr_r;
                    int32_t g = (int32_t)((uint32_t)y_fixed + (uint32_t)(cr * -c_cr_g) + ((uint32_t)(cb * -c_cb_g) & 0xffff0000u));
                    int32_t b = y_fixed + cb * c_cb_b;
                    r >>= 20; g >>= 20; b >>= 20;
                    out[3 * i] = clamp8(r); out[3 * i + 1] = clamp8(g); out[3 * i + 2] = clamp8(b);
                }
            }
        }
        return true;
    }
};

}  // namespace

extern "C" int pt_image_decode_jpeg(const void* data, size_t bytes, int32_t* nx, int32_t* ny, uint8_t* rgb, size_t capacity)
{
    if (!data || !nx || !ny) { pt::set_error("pt_image_decode_jpeg: null argument"); return PT_ERR_INVALID; }
    Decoder d;
    d.data = static_cast<const uint8_t*>(data); d.size = bytes;
    std::vector<uint8_t> out;
    if (!d.parse() || !d.to_rgb(&out)) { pt::set_error(("pt_image_decode_jpeg: " + (d.err.empty() ? std::string("not a JPEG file") : d.err)).c_str()); return PT_ERR_INVALID; }
    *nx = d.nx; *ny = d.ny;
    if (rgb) {
        if (capacity < out.size()) { pt::set_error("pt_image_decode_jpeg: buffer too small"); return PT_ERR_INVALID; }
        std::memcpy(rgb, out.data(), out.size());
    }
    return PT_OK;
}
