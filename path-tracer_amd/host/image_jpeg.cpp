// image_jpeg.cpp -- JPEG (ITU T.81) decoder to RGBA8 for the model loader: baseline and extended
// sequential (SOF0/SOF1) and progressive (SOF2) Huffman streams, 8-bit samples, one or three
// components, any sampling factors, restart intervals.  Stands in for stbi_load(path, ..., 4) of
// source/model_loader.cpp:25 on .jpg textures (assets/dragon/checkerboard.jpg is baseline,
// Dragon_ThicknessMap.jpg progressive).  The inverse DCT is evaluated in floating point and rounded
// once; chroma is upsampled with the triangle filter libjpeg calls "fancy" for 2:1 ratios, by
// replication otherwise.  Decoders differ by a level or two in exactly these steps, so the result is
// compared with another decoder under a tolerance, not bit for bit.
#include "model_loader.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace {

struct Huffman {
    bool present = false;
    uint8_t bits[17] = {};
    uint8_t vals[256] = {};
    int mincode[17] = {}, maxcode[18] = {}, valptr[17] = {};
    void Prepare()
    {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k; mincode[l] = code;
            code += bits[l]; k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int blocksW = 0, blocksH = 0;          // allocated blocks (padded to whole MCUs)
    int unitsW = 0, unitsH = 0;            // blocks that cover the component's own samples (non-interleaved scans)
    std::vector<int16_t> coef;             // blocksW * blocksH * 64, natural order
    std::vector<uint8_t> plane;            // blocksW*8 x blocksH*8 samples
    int dcTable = 0, acTable = 0, dcPred = 0;
};

const uint8_t kZigzag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                              35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

struct Decoder {
    const uint8_t* data = nullptr; size_t size = 0, pos = 0;
    uint32_t bitBuf = 0; int bitCnt = 0; bool hitMarker = false;
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1, mcusX = 0, mcusY = 0;
    bool progressive = false, adobe = false; int adobeTransform = -1;
    uint16_t qt[4][64] = {};
    Huffman dc[4], ac[4];
    Component comp[3];
    int restartInterval = 0;
    int eobrun = 0;
    bool failed = false;

    int U8() { return pos < size ? data[pos++] : (failed = true, 0); }
    int U16() { int a = U8(); return (a << 8) | U8(); }

    // ---- entropy-coded bits (T.81 F.2.2.5): 0xFF00 is a stuffed 0xFF, any other marker ends the segment
    int Bit()
    {
        if (bitCnt == 0) {
            int b = 0;
            if (!hitMarker && pos < size) {
                b = data[pos++];
                if (b == 0xFF) {
                    int b2 = pos < size ? data[pos] : 0xD9;
                    if (b2 == 0) ++pos;
                    else { hitMarker = true; --pos; b = 0; }
                }
            }
            bitBuf = (uint32_t)b; bitCnt = 8;
        }
        --bitCnt;
        return (bitBuf >> bitCnt) & 1;
    }
    int Bits(int n) { int v = 0; while (n-- > 0) v = (v << 1) | Bit(); return v; }
    static int Extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }
    int Decode(const Huffman& h)
    {
        int code = 0;
        for (int l = 1; l <= 16; ++l) {
            code = (code << 1) | Bit();
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
        }
        failed = true;
        return 0;
    }
    void ResetEntropy() { bitCnt = 0; hitMarker = false; eobrun = 0; for (int c = 0; c < ncomp; ++c) comp[c].dcPred = 0; }
    bool Restart()
    {
        bitCnt = 0; hitMarker = false;
        // skip to the RSTn marker
        while (pos + 1 < size && !(data[pos] == 0xFF && data[pos + 1] >= 0xD0 && data[pos + 1] <= 0xD7)) ++pos;
        if (pos + 1 >= size) return false;
        pos += 2;
        eobrun = 0;
        for (int c = 0; c < ncomp; ++c) comp[c].dcPred = 0;
        return true;
    }

    // ---- one block of one scan
    void BlockSequential(Component& c, int16_t* blk)
    {
        const int t = Decode(dc[c.dcTable]);
        c.dcPred += Extend(Bits(t), t);
        blk[0] = (int16_t)c.dcPred;
        for (int k = 1; k < 64;) {
            const int rs = Decode(ac[c.acTable]), r = rs >> 4, s = rs & 15;
            if (s == 0) { if (r == 15) { k += 16; continue; } break; }
            k += r;
            if (k > 63) { failed = true; return; }
            blk[kZigzag[k]] = (int16_t)Extend(Bits(s), s);
            ++k;
        }
    }
    void BlockDcFirst(Component& c, int16_t* blk, int al)
    {
        const int t = Decode(dc[c.dcTable]);
        c.dcPred += Extend(Bits(t), t);
        blk[0] = (int16_t)(c.dcPred * (1 << al));
    }
    void BlockDcRefine(int16_t* blk, int al) { if (Bit()) blk[0] |= (int16_t)(1 << al); }
    void BlockAcFirst(Component& c, int16_t* blk, int ss, int se, int al)
    {
        if (eobrun > 0) { --eobrun; return; }
        for (int k = ss; k <= se;) {
            const int rs = Decode(ac[c.acTable]), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += Bits(r); break; }
                k += 16; continue;
            }
            k += r;
            if (k > 63) { failed = true; return; }
            blk[kZigzag[k]] = (int16_t)(Extend(Bits(s), s) * (1 << al));
            ++k;
        }
    }
    void BlockAcRefine(Component& c, int16_t* blk, int ss, int se, int al)
    {
        const int p1 = 1 << al, m1 = -(1 << al);
        int k = ss;
        if (eobrun <= 0) {
            for (; k <= se;) {
                const int rs = Decode(ac[c.acTable]); int r = rs >> 4; const int s = rs & 15;
                int value = 0;
                if (s == 0) {
                    if (r < 15) { eobrun = (1 << r); if (r) eobrun += Bits(r); break; }
                } else {
                    if (s != 1) { failed = true; return; }
                    value = Bit() ? p1 : m1;
                }
                // skip r zero-history coefficients, refining the non-zero ones on the way
                while (k <= se) {
                    int16_t& co = blk[kZigzag[k]];
                    if (co != 0) {
                        if (Bit() && (co & p1) == 0) co = (int16_t)(co >= 0 ? co + p1 : co + m1);
                    } else {
                        if (r == 0) { if (value) co = (int16_t)value; ++k; break; }
                        --r;
                    }
                    ++k;
                }
                if (failed) return;
            }
        }
        if (eobrun > 0) {
            for (; k <= se; ++k) {
                int16_t& co = blk[kZigzag[k]];
                if (co != 0 && Bit() && (co & p1) == 0) co = (int16_t)(co >= 0 ? co + p1 : co + m1);
            }
            --eobrun;
        }
    }

    bool Scan()
    {
        const int len = U16(); const size_t end = pos + len - 2;
        const int ns = U8();
        if (ns < 1 || ns > ncomp) return false;
        int idx[3];
        for (int i = 0; i < ns; ++i) {
            const int id = U8(), tb = U8();
            idx[i] = -1;
            for (int c = 0; c < ncomp; ++c) if (comp[c].id == id) idx[i] = c;
            if (idx[i] < 0) return false;
            comp[idx[i]].dcTable = (tb >> 4) & 3; comp[idx[i]].acTable = tb & 3;
        }
        const int ss = U8(), se = U8(), a = U8(), ah = a >> 4, al = a & 15;
        if (failed || pos != end) return false;
        if (!progressive && (ss != 0 || se != 63 || ah != 0 || al != 0)) return false;
        if (progressive && (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1))) return false;
        for (int i = 0; i < ns; ++i) {
            const Component& c = comp[idx[i]];
            if ((ss == 0 && !(progressive && ah) && !dc[c.dcTable].present) || (se > 0 && !ac[c.acTable].present)) return false;
        }
        ResetEntropy();
        auto one = [&](Component& c, int bx, int by) {
            int16_t* blk = &c.coef[((size_t)by * c.blocksW + bx) * 64];
            if (!progressive) BlockSequential(c, blk);
            else if (ss == 0) { if (ah == 0) BlockDcFirst(c, blk, al); else BlockDcRefine(blk, al); }
            else if (ah == 0) BlockAcFirst(c, blk, ss, se, al);
            else BlockAcRefine(c, blk, ss, se, al);
        };
        int count = 0;
        if (ns == 1) {      // non-interleaved: the component's own blocks in raster order
            Component& c = comp[idx[0]];
            for (int by = 0; by < c.unitsH; ++by) for (int bx = 0; bx < c.unitsW; ++bx) {
                if (restartInterval && count == restartInterval) { if (!Restart()) return false; count = 0; }
                one(c, bx, by); ++count;
                if (failed) return false;
            }
        } else {
            for (int my = 0; my < mcusY; ++my) for (int mx = 0; mx < mcusX; ++mx) {
                if (restartInterval && count == restartInterval) { if (!Restart()) return false; count = 0; }
                for (int i = 0; i < ns; ++i) {
                    Component& c = comp[idx[i]];
                    for (int v = 0; v < c.v; ++v) for (int h = 0; h < c.h; ++h) one(c, mx * c.h + h, my * c.v + v);
                }
                ++count;
                if (failed) return false;
            }
        }
        // continue after the entropy-coded segment: at the marker that ended it
        bitCnt = 0;
        if (!hitMarker) while (pos + 1 < size && !(data[pos] == 0xFF && data[pos + 1] != 0 && !(data[pos + 1] >= 0xD0 && data[pos + 1] <= 0xD7))) ++pos;
        hitMarker = false;
        return true;
    }

    bool Frame(int marker)
    {
        progressive = marker == 0xC2;
        const int len = U16(); const size_t end = pos + len - 2;
        if (U8() != 8) return false;                      // sample precision
        height = U16(); width = U16(); ncomp = U8();
        if (width <= 0 || height <= 0 || (ncomp != 1 && ncomp != 3)) return false;
        for (int c = 0; c < ncomp; ++c) {
            comp[c].id = U8(); const int hv = U8(); comp[c].h = hv >> 4; comp[c].v = hv & 15; comp[c].tq = U8() & 3;
            if (comp[c].h < 1 || comp[c].h > 4 || comp[c].v < 1 || comp[c].v > 4) return false;
            hmax = comp[c].h > hmax ? comp[c].h : hmax; vmax = comp[c].v > vmax ? comp[c].v : vmax;
        }
        if (failed || pos != end) return false;
        mcusX = (width + 8 * hmax - 1) / (8 * hmax); mcusY = (height + 8 * vmax - 1) / (8 * vmax);
        for (int c = 0; c < ncomp; ++c) {
            Component& k = comp[c];
            k.blocksW = mcusX * k.h; k.blocksH = mcusY * k.v;
            const int cw = (width * k.h + hmax - 1) / hmax, ch = (height * k.v + vmax - 1) / vmax;
            k.unitsW = (cw + 7) / 8; k.unitsH = (ch + 7) / 8;
            k.coef.assign((size_t)k.blocksW * k.blocksH * 64, 0);
        }
        return true;
    }

    bool Tables(int marker)
    {
        const int len = U16(); const size_t end = pos + len - 2;
        if (end > size) return false;
        if (marker == 0xDB) {
            while (pos < end) {
                const int pq = U8(), t = pq & 15, prec = pq >> 4;
                if (t > 3) return false;
                for (int k = 0; k < 64; ++k) qt[t][kZigzag[k]] = (uint16_t)(prec ? U16() : U8());
            }
        } else if (marker == 0xC4) {
            while (pos < end) {
                const int tc = U8(); Huffman& h = (tc >> 4) ? ac[tc & 3] : dc[tc & 3];
                int total = 0;
                h.bits[0] = 0;
                for (int l = 1; l <= 16; ++l) { h.bits[l] = (uint8_t)U8(); total += h.bits[l]; }
                if (total > 256) return false;
                for (int k = 0; k < total; ++k) h.vals[k] = (uint8_t)U8();
                h.Prepare();
            }
        } else if (marker == 0xDD) restartInterval = U16();
        else if (marker == 0xEE && len >= 14 && std::memcmp(data + pos, "Adobe", 5) == 0) { adobe = true; adobeTransform = data[pos + 11]; pos = end; }
        else pos = end;
        return !failed && pos == end;
    }

    // ---- reconstruction
    void InverseDct()
    {
        static double cs[8][8]; static bool init = false;
        if (!init) {
            for (int x = 0; x < 8; ++x) for (int u = 0; u < 8; ++u) cs[x][u] = (u == 0 ? std::sqrt(0.5) : 1.0) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0) * 0.5;
            init = true;
        }
        for (int c = 0; c < ncomp; ++c) {
            Component& k = comp[c];
            const int stride = k.blocksW * 8;
            k.plane.assign((size_t)stride * k.blocksH * 8, 0);
            const uint16_t* q = qt[k.tq];
            for (int by = 0; by < k.blocksH; ++by) for (int bx = 0; bx < k.blocksW; ++bx) {
                const int16_t* blk = &k.coef[((size_t)by * k.blocksW + bx) * 64];
                double f[64], tmp[64];
                for (int i = 0; i < 64; ++i) f[i] = (double)blk[i] * q[i];
                for (int v = 0; v < 8; ++v) for (int x = 0; x < 8; ++x) { double s = 0; for (int u = 0; u < 8; ++u) s += cs[x][u] * f[v * 8 + u]; tmp[v * 8 + x] = s; }
                for (int x = 0; x < 8; ++x) for (int y = 0; y < 8; ++y) {
                    double s = 0; for (int v = 0; v < 8; ++v) s += cs[y][v] * tmp[v * 8 + x];
                    const int val = (int)std::floor(s + 128.5);
                    k.plane[(size_t)(by * 8 + y) * stride + bx * 8 + x] = (uint8_t)(val < 0 ? 0 : (val > 255 ? 255 : val));
                }
            }
        }
    }

    // full-resolution plane of one component
    std::vector<uint8_t> Upsample(const Component& k) const
    {
        const int cw = (width * k.h + hmax - 1) / hmax, ch = (height * k.v + vmax - 1) / vmax, stride = k.blocksW * 8;
        std::vector<uint8_t> out((size_t)width * height);
        auto at = [&](int x, int y) { x = x < 0 ? 0 : (x >= cw ? cw - 1 : x); y = y < 0 ? 0 : (y >= ch ? ch - 1 : y); return (int)k.plane[(size_t)y * stride + x]; };
        const int fx = hmax / k.h, fy = vmax / k.v;
        const bool exact = hmax % k.h == 0 && vmax % k.v == 0;
        if (exact && fx == 1 && fy == 1) {
            for (int y = 0; y < height; ++y) for (int x = 0; x < width; ++x) out[(size_t)y * width + x] = (uint8_t)at(x, y);
        } else if (exact && fx == 2 && fy == 1) {            // h2v1 triangle filter: 3/4 nearer, 1/4 further sample
            for (int y = 0; y < height; ++y) for (int x = 0; x < width; ++x) {
                const int i = x >> 1, n = (x & 1) ? i + 1 : i - 1;
                out[(size_t)y * width + x] = (uint8_t)((3 * at(i, y) + at(n, y) + ((x & 1) ? 2 : 1)) >> 2);
            }
        } else if (exact && fx == 2 && fy == 2) {            // h2v2: the same filter in both directions, one rounding
            for (int y = 0; y < height; ++y) {
                const int j = y >> 1, m = (y & 1) ? j + 1 : j - 1;
                for (int x = 0; x < width; ++x) {
                    const int i = x >> 1, n = (x & 1) ? i + 1 : i - 1;
                    const int near = 3 * at(i, j) + at(i, m), far = 3 * at(n, j) + at(n, m);
                    out[(size_t)y * width + x] = (uint8_t)((3 * near + far + ((x & 1) ? 7 : 8)) >> 4);
                }
            }
        } else {                                              // any other ratio: nearest sample
            for (int y = 0; y < height; ++y) for (int x = 0; x < width; ++x) out[(size_t)y * width + x] = (uint8_t)at(x * k.h / hmax, y * k.v / vmax);
        }
        return out;
    }

    bool Run(std::vector<uint8_t>& rgba)
    {
        if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return false;
        pos = 2;
        bool haveFrame = false, done = false;
        while (!done && pos + 1 < size) {
            if (data[pos] != 0xFF) { ++pos; continue; }
            const int m = data[pos + 1];
            if (m == 0xFF) { ++pos; continue; }
            pos += 2;
            if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01 || m == 0x00) continue;
            if (m == 0xD9) { done = true; break; }
            if (m == 0xC0 || m == 0xC1 || m == 0xC2) { if (haveFrame || !Frame(m)) return false; haveFrame = true; }
            else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC4 && m != 0xCC) ) return false;     // lossless, differential, arithmetic
            else if (m == 0xDA) { if (!haveFrame || !Scan()) return false; }
            else if (!Tables(m)) return false;
        }
        if (!haveFrame || failed) return false;
        InverseDct();
        rgba.assign((size_t)width * height * 4, 255);
        if (ncomp == 1) {
            const std::vector<uint8_t> y = Upsample(comp[0]);
            for (size_t i = 0; i < y.size(); ++i) { rgba[4 * i] = rgba[4 * i + 1] = rgba[4 * i + 2] = y[i]; }
            return true;
        }
        const std::vector<uint8_t> p0 = Upsample(comp[0]), p1 = Upsample(comp[1]), p2 = Upsample(comp[2]);
        const bool isRgb = (adobe && adobeTransform == 0) || (!adobe && comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B');
        for (size_t i = 0; i < p0.size(); ++i) {
            if (isRgb) { rgba[4 * i] = p0[i]; rgba[4 * i + 1] = p1[i]; rgba[4 * i + 2] = p2[i]; continue; }
            const double Y = p0[i], cb = p1[i] - 128.0, cr = p2[i] - 128.0;
            const double c[3] = { Y + 1.402 * cr, Y - 0.344136286 * cb - 0.714136286 * cr, Y + 1.772 * cb };
            for (int k = 0; k < 3; ++k) { const int v = (int)std::floor(c[k] + 0.5); rgba[4 * i + k] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
        }
        return true;
    }
};

bool ReadFile(const std::string& path, std::vector<uint8_t>& bytes)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    bytes.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return !bytes.empty();
}

} // namespace

bool LoadJpegRGBA8(const std::string& path, std::vector<uint8_t>& rgba, uint32_t& width, uint32_t& height)
{
    std::vector<uint8_t> bytes;
    if (!ReadFile(path, bytes)) return false;
    Decoder d;
    d.data = bytes.data(); d.size = bytes.size();
    if (!d.Run(rgba)) return false;
    width = (uint32_t)d.width; height = (uint32_t)d.height;
    return true;
}

// stbi_load dispatches on the file's signature, not on its name
bool LoadImageRGBA8(const std::string& path, std::vector<uint8_t>& rgba, uint32_t& width, uint32_t& height)
{
    unsigned char sig[4] = {};
    { std::ifstream f(path, std::ios::binary); if (!f) return false; f.read(reinterpret_cast<char*>(sig), 4); }
    if (sig[0] == 0xFF && sig[1] == 0xD8) return LoadJpegRGBA8(path, rgba, width, height);
    return LoadPngRGBA8(path, rgba, width, height);
}

// C entry point for tests and scripts: decodes into `out` (capacity in bytes); returns 0 on success.
extern "C" int pt_host_decode_image(const char* path, uint32_t* width, uint32_t* height, uint8_t* out, size_t capacity)
{
    std::vector<uint8_t> rgba; uint32_t w = 0, h = 0;
    if (!LoadImageRGBA8(path, rgba, w, h)) return 1;
    if (width) *width = w;
    if (height) *height = h;
    if (out) { if (capacity < rgba.size()) return 2; std::memcpy(out, rgba.data(), rgba.size()); }
    return 0;
}
