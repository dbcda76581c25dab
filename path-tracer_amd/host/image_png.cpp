// image_png.cpp -- PNG -> RGBA8 (top row first), the part of stbi_load(path, ..., 4)
// (source/model_loader.cpp:25) this build needs: 8-bit and 16-bit samples, grey / grey+alpha / RGB /
// RGBA / palette (1, 2, 4, 8 bits), tRNS, no interlace.  Inflate comes from zlib.
#include "model_loader.hpp"
#include <cstdio>
#include <cstring>
#include <zlib.h>

static uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

bool LoadPngRGBA8(const std::string& path, std::vector<uint8_t>& rgba, uint32_t& width, uint32_t& height)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<uint8_t> file;
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
    std::fclose(f);
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (file.size() < 33 || std::memcmp(file.data(), sig, 8) != 0) return false;
    uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    size_t p = 8;
    while (p + 12 <= file.size()) {
        uint32_t len = be32(&file[p]);
        const char* type = reinterpret_cast<const char*>(&file[p + 4]);
        const uint8_t* data = &file[p + 8];
        if (p + 12 + len > file.size()) return false;
        if (!std::memcmp(type, "IHDR", 4)) { w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12]; }
        else if (!std::memcmp(type, "PLTE", 4)) palette.assign(data, data + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        p += 12 + len;
    }
    if (!w || !h || interlace != 0) return false;
    const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!channels || (depth != 8 && depth != 16 && !(depth < 8 && (ctype == 0 || ctype == 3)))) return false;
    const size_t bpp_bits = size_t(channels) * depth;
    const size_t stride = (w * bpp_bits + 7) / 8;
    const size_t bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf out_len = static_cast<uLongf>(raw.size());
    if (uncompress(raw.data(), &out_len, idat.data(), static_cast<uLong>(idat.size())) != Z_OK || out_len != raw.size()) return false;
    // undo the per-row filters in place
    std::vector<uint8_t> prev(stride, 0);
    for (uint32_t y = 0; y < h; ++y) {
        uint8_t* row = &raw[y * (stride + 1)];
        const int ft = row[0];
        uint8_t* cur = row + 1;
        for (size_t i = 0; i < stride; ++i) {
            int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0, v = cur[i];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: { int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
            default: return false;
            }
            cur[i] = static_cast<uint8_t>(v);
        }
        std::memcpy(prev.data(), cur, stride);
    }
    rgba.assign(size_t(w) * h * 4, 255);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* cur = &raw[y * (stride + 1) + 1];
        for (uint32_t x = 0; x < w; ++x) {
            uint8_t* o = &rgba[(size_t(y) * w + x) * 4];
            auto sample = [&](int ch) -> int {
                if (depth == 8) return cur[size_t(x) * channels + ch];
                if (depth == 16) return cur[(size_t(x) * channels + ch) * 2];   // high byte, like stb's 16->8 conversion
                const int per = 8 / depth, idx = x / per, shift = (per - 1 - x % per) * depth;
                return (cur[idx] >> shift) & ((1 << depth) - 1);
            };
            if (ctype == 3) {
                int i = sample(0);
                if (size_t(i) * 3 + 2 < palette.size()) { o[0] = palette[i * 3]; o[1] = palette[i * 3 + 1]; o[2] = palette[i * 3 + 2]; }
                o[3] = size_t(i) < trns.size() ? trns[i] : 255;
            } else if (ctype == 0 || ctype == 4) {
                int g = sample(0);
                if (depth < 8) g = g * 255 / ((1 << depth) - 1);
                o[0] = o[1] = o[2] = static_cast<uint8_t>(g);
                if (ctype == 4) o[3] = static_cast<uint8_t>(sample(1));
            } else {
                o[0] = static_cast<uint8_t>(sample(0)); o[1] = static_cast<uint8_t>(sample(1)); o[2] = static_cast<uint8_t>(sample(2));
                if (ctype == 6) o[3] = static_cast<uint8_t>(sample(3));
            }
        }
    }
    width = w; height = h;
    return true;
}
