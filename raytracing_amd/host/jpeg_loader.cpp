// jpeg_loader.cpp -- JPEG -> packed RGBA8 texels for Scene::LoadTexture.
//
// The reference decodes textures with stb_image v2.27 through LoadSTB
// (src/loaders/image_loader.cpp:30-63, 3rdparty/stb/stb_image.h): stbi_load(..., req_comp = 0)
// returns 3 channels for colour files and 1 for greyscale, and the texel is
// r | g<<8 | b<<16 (alpha 0; greyscale: r only).  JPEG decoding is only defined up to the
// accuracy of the inverse DCT, so "the same texels" means reproducing stb_image's integer
// pipeline stage by stage:
//   * entropy decoding (baseline / extended sequential and progressive Huffman, restart
//     intervals) follows ITU T.81 -- any conforming decoder yields the same coefficients;
//     coefficients live in 16-bit words and dequantisation wraps like stb's `short` math
//     (stb_image.h:2182-2384, 3039-3063);
//   * inverse DCT: the jidctint "islow" factorisation in 12-bit fixed point, column pass kept
//     with 2 extra bits (+512 >> 10, a DC-only column is dc*4), row pass +65536 + (128<<17)
//     >> 17 with clamping (stb_image.h:2392-2493);
//   * chroma upsampling: 1x, the 3:1 linear filters for 2x horizontal / vertical, the 9:3:3:1
//     filter for 2x2, nearest for every other factor, with stb's row stepping
//     (stb_image.h:3411-3474, 3592-3603, 3840-3880);
//   * YCbCr -> RGB in 20-bit fixed point with stb's truncated constants and the
//     `& 0xffff0000` on the Cb term of green (stb_image.h:3606-3630); files whose component
//     ids are 'R','G','B', or Adobe files with transform 0 and no JFIF header, are taken as RGB.
// Pinned texel for texel against the reference's own stb build (tests/test_host_layer.py).
// Not supported (rejected): 4-component (CMYK / YCCK) files, 12-bit and arithmetic coding.
#include <cstdio>
#include <cstring>
#include <vector>
#include "scene.hpp"

namespace rt
{
namespace
{
typedef unsigned char u8;

const u8 kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7,
    14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
    54, 47, 55, 62, 63};

struct Huffman
{
    bool defined = false;
    int mincode[17], maxcode[18], valptr[17];
    u8 values[256];

    bool Build(const u8 counts[16], const u8* symbols, int n)           // T.81 annex C / F.2.2.3
    {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len)
        {
            valptr[len] = k;
            mincode[len] = code;
            code += counts[len - 1];
            k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            if (code > (1 << len)) return false;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        std::memcpy(values, symbols, (size_t)n);
        defined = true;
        return true;
    }
};

// MSB-first bit reader over the entropy-coded segment: 0xFF00 -> 0xFF, any other 0xFFxx is a
// marker: it is remembered and the stream continues with zero bits (stb_image.h:2052-2075)
struct BitReader
{
    const u8* p = nullptr;
    const u8* end = nullptr;
    unsigned acc = 0;
    int bits = 0;
    int marker = -1;

    void Reset() { acc = 0; bits = 0; marker = -1; }
    void Fill()
    {
        while (bits <= 24)
        {
            unsigned b = 0;
            if (marker < 0 && p < end)
            {
                b = *p++;
                if (b == 0xFF)
                {
                    unsigned c = p < end ? *p++ : 0;
                    while (c == 0xFF) c = p < end ? *p++ : 0;            // fill bytes
                    if (c != 0) { marker = (int)c; b = 0; }
                }
            }
            acc |= b << (24 - bits);
            bits += 8;
        }
    }
    int Bit()
    {
        if (bits < 1) Fill();
        int b = (int)(acc >> 31);
        acc <<= 1; --bits;
        return b;
    }
    int Bits(int n)
    {
        if (n == 0) return 0;
        if (bits < n) Fill();
        int v = (int)(acc >> (32 - n));
        acc <<= n; bits -= n;
        return v;
    }
    int Receive(int n)                                                  // T.81 F.2.2.1 EXTEND(RECEIVE(n), n)
    {
        int v = Bits(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
    int Decode(const Huffman& h)                                        // -1: no such code
    {
        int code = 0;
        for (int len = 1; len <= 16; ++len)
        {
            code = (code << 1) | Bit();
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len])
                return h.values[h.valptr[len] + code - h.mincode[len]];
        }
        return -1;
    }
};

struct Component
{
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0, dc_pred = 0;
    int x = 0, y = 0, w2 = 0, h2 = 0;          // sample dimensions, MCU-padded dimensions
    int blocks_w = 0;                           // coefficient blocks per row (progressive)
    std::vector<u8> data;                       // w2 x h2 samples
    std::vector<short> coeff;                   // progressive: blocks_w x (h2/8) x 64
};

inline u8 Clamp255(int x) { return (u8)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// one 8-point pass of the "islow" inverse DCT in 12-bit fixed point
struct Idct1D
{
    int x0, x1, x2, x3, t0, t1, t2, t3;
    static int F(double v) { return (int)(v * 4096 + 0.5); }
    Idct1D(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7)
    {
        int p2 = s2, p3 = s6;
        int p1 = (p2 + p3) * F(0.5411961f);
        int e2 = p1 + p3 * F(-1.847759065f);
        int e3 = p1 + p2 * F(0.765366865f);
        p2 = s0; p3 = s4;
        int e0 = (p2 + p3) * 4096, e1 = (p2 - p3) * 4096;
        x0 = e0 + e3; x3 = e0 - e3; x1 = e1 + e2; x2 = e1 - e2;
        t0 = s7; t1 = s5; t2 = s3; t3 = s1;
        p3 = t0 + t2;
        int p4 = t1 + t3;
        p1 = t0 + t3; p2 = t1 + t2;
        int p5 = (p3 + p4) * F(1.175875602f);
        t0 = t0 * F(0.298631336f);
        t1 = t1 * F(2.053119869f);
        t2 = t2 * F(3.072711026f);
        t3 = t3 * F(1.501321110f);
        p1 = p5 + p1 * F(-0.899976223f);
        p2 = p5 + p2 * F(-2.562915447f);
        p3 = p3 * F(-1.961570560f);
        p4 = p4 * F(-0.390180644f);
        t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
    }
};

void InverseDct(u8* out, int stride, const short d[64])
{
    int val[64];
    for (int c = 0; c < 8; ++c)
    {
        const short* s = d + c;
        int* v = val + c;
        if (s[8] == 0 && s[16] == 0 && s[24] == 0 && s[32] == 0 && s[40] == 0 && s[48] == 0 && s[56] == 0)
        {
            int dc = s[0] * 4;
            for (int r = 0; r < 8; ++r) v[r * 8] = dc;
            continue;
        }
        Idct1D k(s[0], s[8], s[16], s[24], s[32], s[40], s[48], s[56]);
        int x0 = k.x0 + 512, x1 = k.x1 + 512, x2 = k.x2 + 512, x3 = k.x3 + 512;
        v[0] = (x0 + k.t3) >> 10; v[56] = (x0 - k.t3) >> 10;
        v[8] = (x1 + k.t2) >> 10; v[48] = (x1 - k.t2) >> 10;
        v[16] = (x2 + k.t1) >> 10; v[40] = (x2 - k.t1) >> 10;
        v[24] = (x3 + k.t0) >> 10; v[32] = (x3 - k.t0) >> 10;
    }
    for (int r = 0; r < 8; ++r)
    {
        const int* v = val + r * 8;
        u8* o = out + (size_t)r * stride;
        Idct1D k(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        const int bias = 65536 + (128 << 17);
        int x0 = k.x0 + bias, x1 = k.x1 + bias, x2 = k.x2 + bias, x3 = k.x3 + bias;
        o[0] = Clamp255((x0 + k.t3) >> 17); o[7] = Clamp255((x0 - k.t3) >> 17);
        o[1] = Clamp255((x1 + k.t2) >> 17); o[6] = Clamp255((x1 - k.t2) >> 17);
        o[2] = Clamp255((x2 + k.t1) >> 17); o[5] = Clamp255((x2 - k.t1) >> 17);
        o[3] = Clamp255((x3 + k.t0) >> 17); o[4] = Clamp255((x3 - k.t0) >> 17);
    }
}

// ---- upsampling of one output row; `near_row` is the closer source row -------------------
const u8* Upsample(u8* out, const u8* near_row, const u8* far_row, int w, int hs, int vs)
{
    if (hs == 1 && vs == 1) return near_row;
    if (hs == 1 && vs == 2)
    {
        for (int i = 0; i < w; ++i) out[i] = (u8)((3 * near_row[i] + far_row[i] + 2) >> 2);
        return out;
    }
    if (hs == 2 && vs == 1)
    {
        const u8* in = near_row;
        if (w == 1) { out[0] = out[1] = in[0]; return out; }
        out[0] = in[0];
        out[1] = (u8)((in[0] * 3 + in[1] + 2) >> 2);
        int i;
        for (i = 1; i < w - 1; ++i)
        {
            int n = 3 * in[i] + 2;
            out[i * 2 + 0] = (u8)((n + in[i - 1]) >> 2);
            out[i * 2 + 1] = (u8)((n + in[i + 1]) >> 2);
        }
        out[i * 2 + 0] = (u8)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
        out[i * 2 + 1] = in[w - 1];
        return out;
    }
    if (hs == 2 && vs == 2)
    {
        if (w == 1) { out[0] = out[1] = (u8)((3 * near_row[0] + far_row[0] + 2) >> 2); return out; }
        int t1 = 3 * near_row[0] + far_row[0];
        out[0] = (u8)((t1 + 2) >> 2);
        for (int i = 1; i < w; ++i)
        {
            int t0 = t1;
            t1 = 3 * near_row[i] + far_row[i];
            out[i * 2 - 1] = (u8)((3 * t0 + t1 + 8) >> 4);
            out[i * 2] = (u8)((3 * t1 + t0 + 8) >> 4);
        }
        out[w * 2 - 1] = (u8)((t1 + 2) >> 2);
        return out;
    }
    for (int i = 0; i < w; ++i)
        for (int j = 0; j < hs; ++j) out[i * hs + j] = near_row[i];
    return out;
}

struct Decoder
{
    std::vector<u8> file;
    size_t pos = 0;
    int width = 0, height = 0, ncomp = 0;
    bool progressive = false, jfif = false;
    int adobe_transform = -1, rgb_ids = 0, restart_interval = 0;
    int h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0;
    unsigned short dequant[4][64];
    Huffman dc[4], ac[4];
    Component comp[4];
    // scan state
    int scan_n = 0, order[4] = {0, 0, 0, 0};
    int spec_start = 0, spec_end = 0, succ_high = 0, succ_low = 0, eob_run = 0, todo = 0;
    BitReader br;

    bool Eof() const { return pos >= file.size(); }
    int Get8() { return pos < file.size() ? file[pos++] : 0; }
    int Get16() { int a = Get8(); return (a << 8) | Get8(); }

    int NextMarker()                                                   // 0xFF (fills) xx; -1: none here
    {
        if (br.marker >= 0) { int m = br.marker; br.marker = -1; return m; }
        int x = Get8();
        if (x != 0xFF) return -1;
        while (x == 0xFF) x = Get8();
        return x;
    }

    bool ProcessMarker(int m)
    {
        switch (m)
        {
        case 0xDD:                                                     // DRI
            if (Get16() != 4) return false;
            restart_interval = Get16();
            return true;
        case 0xDB:                                                     // DQT
        {
            int len = Get16() - 2;
            while (len > 0)
            {
                int q = Get8(), p = q >> 4, t = q & 15;
                if ((p != 0 && p != 1) || t > 3) return false;
                for (int i = 0; i < 64; ++i) dequant[t][kZigzag[i]] = (unsigned short)(p ? Get16() : Get8());
                len -= p ? 129 : 65;
            }
            return len == 0;
        }
        case 0xC4:                                                     // DHT
        {
            int len = Get16() - 2;
            while (len > 0)
            {
                int q = Get8(), tc = q >> 4, th = q & 15;
                if (tc > 1 || th > 3) return false;
                u8 counts[16], symbols[256];
                int n = 0;
                for (int i = 0; i < 16; ++i) { counts[i] = (u8)Get8(); n += counts[i]; }
                if (n > 256) return false;
                for (int i = 0; i < n; ++i) symbols[i] = (u8)Get8();
                if (!(tc ? ac[th] : dc[th]).Build(counts, symbols, n)) return false;
                len -= 17 + n;
            }
            return len == 0;
        }
        default:
            break;
        }
        if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE)                     // APPn, COM
        {
            int len = Get16();
            if (len < 2) return false;
            len -= 2;
            if (m == 0xE0 && len >= 5)
            {
                static const char tag[5] = {'J', 'F', 'I', 'F', 0};
                bool ok = true;
                for (int i = 0; i < 5; ++i) if (Get8() != tag[i]) ok = false;
                len -= 5;
                if (ok) jfif = true;
            }
            else if (m == 0xEE && len >= 12)
            {
                static const char tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
                bool ok = true;
                for (int i = 0; i < 6; ++i) if (Get8() != tag[i]) ok = false;
                len -= 6;
                if (ok)
                {
                    Get8(); Get16(); Get16();                          // version, flags0, flags1
                    adobe_transform = Get8();
                    len -= 6;
                }
            }
            pos += (size_t)len;
            return pos <= file.size();
        }
        return false;                                                  // unknown / unsupported marker
    }

    bool FrameHeader(int m)
    {
        progressive = m == 0xC2;
        int len = Get16();
        if (len < 11 || Get8() != 8) return false;                     // 8-bit samples only
        height = Get16();
        width = Get16();
        ncomp = Get8();
        if (height == 0 || width == 0 || (ncomp != 1 && ncomp != 3) || len != 8 + 3 * ncomp) return false;
        static const u8 rgb[3] = {'R', 'G', 'B'};
        for (int i = 0; i < ncomp; ++i)
        {
            comp[i].id = Get8();
            if (ncomp == 3 && comp[i].id == rgb[i]) ++rgb_ids;
            int q = Get8();
            comp[i].h = q >> 4; comp[i].v = q & 15;
            comp[i].tq = Get8();
            if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3) return false;
            if (comp[i].h > h_max) h_max = comp[i].h;
            if (comp[i].v > v_max) v_max = comp[i].v;
        }
        for (int i = 0; i < ncomp; ++i)
            if (h_max % comp[i].h != 0 || v_max % comp[i].v != 0) return false;
        mcu_x = (width + h_max * 8 - 1) / (h_max * 8);
        mcu_y = (height + v_max * 8 - 1) / (v_max * 8);
        for (int i = 0; i < ncomp; ++i)
        {
            Component& c = comp[i];
            c.x = (width * c.h + h_max - 1) / h_max;
            c.y = (height * c.v + v_max - 1) / v_max;
            c.w2 = mcu_x * c.h * 8;
            c.h2 = mcu_y * c.v * 8;
            c.data.assign((size_t)c.w2 * c.h2, 0);
            if (progressive)
            {
                c.blocks_w = c.w2 / 8;
                c.coeff.assign((size_t)c.w2 * c.h2, 0);
            }
        }
        return true;
    }

    bool ScanHeader()
    {
        int len = Get16();
        scan_n = Get8();
        if (scan_n < 1 || scan_n > ncomp || len != 6 + 2 * scan_n) return false;
        for (int i = 0; i < scan_n; ++i)
        {
            int id = Get8(), q = Get8(), which = 0;
            for (; which < ncomp; ++which) if (comp[which].id == id) break;
            if (which == ncomp) return false;
            comp[which].hd = q >> 4; comp[which].ha = q & 15;
            if (comp[which].hd > 3 || comp[which].ha > 3) return false;
            order[i] = which;
        }
        spec_start = Get8();
        spec_end = Get8();
        int a = Get8();
        succ_high = a >> 4; succ_low = a & 15;
        if (progressive)
        {
            if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13) return false;
        }
        else
        {
            if (spec_start != 0 || succ_high != 0 || succ_low != 0) return false;
            spec_end = 63;
        }
        return true;
    }

    void ResetEntropy()
    {
        br.Reset();
        for (int i = 0; i < 4; ++i) comp[i].dc_pred = 0;
        eob_run = 0;
        todo = restart_interval ? restart_interval : 0x7FFFFFFF;
    }

    // ---- block decoders ---------------------------------------------------------------
    bool BlockSequential(short d[64], Component& c)
    {
        const unsigned short* q = dequant[c.tq];
        if (!dc[c.hd].defined || !ac[c.ha].defined) return false;
        int t = br.Decode(dc[c.hd]);
        if (t < 0 || t > 15) return false;
        std::memset(d, 0, 64 * sizeof(short));
        int diff = t ? br.Receive(t) : 0;
        c.dc_pred += diff;
        d[0] = (short)(c.dc_pred * q[0]);
        int k = 1;
        do
        {
            int rs = br.Decode(ac[c.ha]);
            if (rs < 0) return false;
            int s = rs & 15, r = rs >> 4;
            if (s == 0)
            {
                if (rs != 0xF0) break;
                k += 16;
            }
            else
            {
                k += r;
                if (k > 63) return false;
                int z = kZigzag[k++];
                d[z] = (short)(br.Receive(s) * q[z]);
            }
        } while (k < 64);
        return true;
    }

    bool BlockProgressiveDc(short d[64], Component& c)
    {
        if (spec_end != 0) return false;
        if (succ_high == 0)
        {
            if (!dc[c.hd].defined) return false;
            std::memset(d, 0, 64 * sizeof(short));
            int t = br.Decode(dc[c.hd]);
            if (t < 0 || t > 15) return false;
            int diff = t ? br.Receive(t) : 0;
            c.dc_pred += diff;
            d[0] = (short)(c.dc_pred * (1 << succ_low));
        }
        else if (br.Bit())
            d[0] = (short)(d[0] + (short)(1 << succ_low));
        return true;
    }

    static void Refine(short& p, short bit)                            // correction bit of a non-zero coefficient
    {
        if ((p & bit) == 0) p = (short)(p > 0 ? p + bit : p - bit);
    }

    bool BlockProgressiveAc(short d[64], Component& c)
    {
        if (spec_start == 0 || !ac[c.ha].defined) return false;
        if (succ_high == 0)
        {
            if (eob_run) { --eob_run; return true; }
            int k = spec_start;
            do
            {
                int rs = br.Decode(ac[c.ha]);
                if (rs < 0) return false;
                int s = rs & 15, r = rs >> 4;
                if (s == 0)
                {
                    if (r < 15)
                    {
                        eob_run = (1 << r);
                        if (r) eob_run += br.Bits(r);
                        --eob_run;
                        break;
                    }
                    k += 16;
                }
                else
                {
                    k += r;
                    if (k > 63) return false;
                    int z = kZigzag[k++];
                    d[z] = (short)(br.Receive(s) * (1 << succ_low));
                }
            } while (k <= spec_end);
            return true;
        }
        const short bit = (short)(1 << succ_low);
        if (eob_run)
        {
            --eob_run;
            for (int k = spec_start; k <= spec_end; ++k)
            {
                short& p = d[kZigzag[k]];
                if (p != 0 && br.Bit()) Refine(p, bit);
            }
            return true;
        }
        int k = spec_start;
        do
        {
            int rs = br.Decode(ac[c.ha]);
            if (rs < 0) return false;
            int s = rs & 15, r = rs >> 4;
            if (s == 0)
            {
                if (r < 15)
                {
                    eob_run = (1 << r) - 1;
                    if (r) eob_run += br.Bits(r);
                    r = 64;                                            // finish the block: only corrections remain
                }
            }
            else
            {
                if (s != 1) return false;
                s = br.Bit() ? bit : -bit;
            }
            while (k <= spec_end)
            {
                short& p = d[kZigzag[k++]];
                if (p != 0)
                {
                    if (br.Bit()) Refine(p, bit);
                }
                else
                {
                    if (r == 0) { p = (short)s; break; }
                    --r;
                }
            }
        } while (k <= spec_end);
        return true;
    }

    // true: continue with the next restart interval; false: stop decoding this scan
    bool RestartPoint()
    {
        if (--todo > 0) return true;
        br.Fill();                                                     // runs into the RSTn marker
        if (br.marker < 0xD0 || br.marker > 0xD7) return false;
        int hd[4], ha[4];
        for (int i = 0; i < 4; ++i) { hd[i] = comp[i].hd; ha[i] = comp[i].ha; }
        ResetEntropy();
        for (int i = 0; i < 4; ++i) { comp[i].hd = hd[i]; comp[i].ha = ha[i]; }
        return true;
    }

    bool DecodeScan()
    {
        br.p = file.data() + pos;
        br.end = file.data() + file.size();
        ResetEntropy();
        short block[64];
        bool ok = true, go = true;
        if (scan_n == 1)
        {
            Component& c = comp[order[0]];
            int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h && go; ++j)
                for (int i = 0; i < w && go; ++i)
                {
                    if (progressive)
                    {
                        short* d = &c.coeff[64 * ((size_t)i + (size_t)j * c.blocks_w)];
                        ok = spec_start == 0 ? BlockProgressiveDc(d, c) : BlockProgressiveAc(d, c);
                    }
                    else
                    {
                        ok = BlockSequential(block, c);
                        if (ok) InverseDct(&c.data[(size_t)c.w2 * j * 8 + (size_t)i * 8], c.w2, block);
                    }
                    if (!ok) go = false;
                    else if (!RestartPoint()) go = false;
                }
        }
        else
        {
            for (int j = 0; j < mcu_y && go; ++j)
                for (int i = 0; i < mcu_x && go; ++i)
                {
                    for (int k = 0; k < scan_n && ok; ++k)
                    {
                        Component& c = comp[order[k]];
                        for (int y = 0; y < c.v && ok; ++y)
                            for (int x = 0; x < c.h && ok; ++x)
                            {
                                int bx = i * c.h + x, by = j * c.v + y;
                                if (progressive)
                                    ok = BlockProgressiveDc(&c.coeff[64 * ((size_t)bx + (size_t)by * c.blocks_w)], c);
                                else
                                {
                                    ok = BlockSequential(block, c);
                                    if (ok) InverseDct(&c.data[(size_t)c.w2 * by * 8 + (size_t)bx * 8], c.w2, block);
                                }
                            }
                    }
                    if (!ok) go = false;
                    else if (!RestartPoint()) go = false;
                }
        }
        pos = (size_t)(br.p - file.data());
        return ok;
    }

    void FinishProgressive()
    {
        for (int n = 0; n < ncomp; ++n)
        {
            Component& c = comp[n];
            int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; ++j)
                for (int i = 0; i < w; ++i)
                {
                    short* d = &c.coeff[64 * ((size_t)i + (size_t)j * c.blocks_w)];
                    const unsigned short* q = dequant[c.tq];
                    for (int k = 0; k < 64; ++k) d[k] = (short)(d[k] * q[k]);
                    InverseDct(&c.data[(size_t)c.w2 * j * 8 + (size_t)i * 8], c.w2, d);
                }
        }
    }

    bool Decode()
    {
        std::memset(dequant, 0, sizeof(dequant));
        if (Get8() != 0xFF || Get8() != 0xD8) return false;            // SOI
        int m = NextMarker();
        while (m != 0xC0 && m != 0xC1 && m != 0xC2)
        {
            if (m < 0 || !ProcessMarker(m)) return false;
            m = NextMarker();
            while (m < 0) { if (Eof()) return false; m = NextMarker(); }
        }
        if (!FrameHeader(m)) return false;
        m = NextMarker();
        while (m != 0xD9)                                              // EOI
        {
            if (m == 0xDA)
            {
                if (!ScanHeader() || !DecodeScan()) return false;
                if (br.marker < 0)                                     // trailing bytes before the next marker
                    while (!Eof())
                        if (Get8() == 0xFF) { br.marker = Get8(); break; }
            }
            else if (m == 0xDC)                                        // DNL must repeat the frame height
            {
                int len = Get16(), lines = Get16();
                if (len != 4 || lines != height) return false;
            }
            else if (m < 0 || !ProcessMarker(m)) return false;
            m = NextMarker();
            if (m < 0 && Eof()) break;
        }
        if (progressive) FinishProgressive();
        return true;
    }
};
} // namespace

bool LoadJPEG(const char* filename, Image& res)
{
    FILE* f = std::fopen(filename, "rb");
    if (!f) return false;
    Decoder d;
    unsigned char buf[65536];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) d.file.insert(d.file.end(), buf, buf + got);
    std::fclose(f);
    if (!d.Decode()) return false;

    const bool is_rgb = d.ncomp == 3 && (d.rgb_ids == 3 || (d.adobe_transform == 0 && !d.jfif));
    const int W = d.width, H = d.height;
    res.width = (std::uint32_t)W;
    res.height = (std::uint32_t)H;
    res.data.assign((size_t)W * H, 0);

    struct Resample { int hs, vs, ystep, w_lores, ypos; const u8 *line0, *line1; std::vector<u8> buf; };
    Resample rs[3];
    for (int k = 0; k < d.ncomp; ++k)
    {
        Resample& r = rs[k];
        r.hs = d.h_max / d.comp[k].h;
        r.vs = d.v_max / d.comp[k].v;
        r.ystep = r.vs >> 1;
        r.w_lores = (W + r.hs - 1) / r.hs;
        r.ypos = 0;
        r.line0 = r.line1 = d.comp[k].data.data();
        r.buf.assign((size_t)W + 3 + 8, 0);
    }
    for (int j = 0; j < H; ++j)
    {
        const u8* row[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < d.ncomp; ++k)
        {
            Resample& r = rs[k];
            bool bottom = r.ystep >= (r.vs >> 1);
            row[k] = Upsample(r.buf.data(), bottom ? r.line1 : r.line0, bottom ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
            if (++r.ystep >= r.vs)
            {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < d.comp[k].y) r.line1 += d.comp[k].w2;
            }
        }
        std::uint32_t* out = &res.data[(size_t)j * W];
        if (d.ncomp == 1)
            for (int i = 0; i < W; ++i) out[i] = row[0][i];                          // LoadSTB: r only
        else if (is_rgb)
            for (int i = 0; i < W; ++i) out[i] = (std::uint32_t)row[0][i] | (std::uint32_t)row[1][i] << 8 | (std::uint32_t)row[2][i] << 16;
        else
            for (int i = 0; i < W; ++i)
            {
                // 20-bit fixed point, constants truncated to 12 bits then shifted (stb_image.h:3606-3630)
                const int kCrR = ((int)(1.40200f * 4096.0f + 0.5f)) << 8, kCrG = ((int)(0.71414f * 4096.0f + 0.5f)) << 8;
                const int kCbG = ((int)(0.34414f * 4096.0f + 0.5f)) << 8, kCbB = ((int)(1.77200f * 4096.0f + 0.5f)) << 8;
                int yf = (row[0][i] << 20) + (1 << 19);
                int cr = row[2][i] - 128, cb = row[1][i] - 128;
                int r = yf + cr * kCrR;
                int g = yf + cr * -kCrG + (int)((unsigned)(cb * -kCbG) & 0xffff0000u);
                int b = yf + cb * kCbB;
                out[i] = (std::uint32_t)Clamp255(r >> 20) | (std::uint32_t)Clamp255(g >> 20) << 8 | (std::uint32_t)Clamp255(b >> 20) << 16;
            }
    }
    return true;
}
} // namespace rt
