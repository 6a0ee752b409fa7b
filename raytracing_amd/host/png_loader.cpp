// png_loader.cpp -- PNG -> packed RGBA8 texels for Scene::LoadTexture.
//
// The reference decodes textures with stb_image through LoadSTB
// (src/loaders/image_loader.cpp:30-63): stbi_load(..., req_comp = 0) keeps the
// file's channel count n, and texel = r | g<<8 | b<<16 | a<<24 with the channels
// beyond n left 0 (so a grey+alpha image puts its alpha in "g").  PNG is lossless,
// so any conforming decoder yields the same samples; this one follows the PNG
// specification (chunks, zlib stream, the five scanline filters) with zlib's
// inflate, and stb_image's conventions for what it hands back:
//   * 16-bit samples are reduced to their high byte,
//   * palette images expand to RGB (RGBA when a tRNS chunk is present),
//   * a tRNS colour key on grey / RGB images adds an alpha channel (0 for the key).
// Adam7-interlaced files (PNG spec 8.2: seven reduced images, each filtered on its own) are decoded pass by pass.
#include <cstdio>
#include <cstring>
#include <vector>
#include <zlib.h>
#include "scene.hpp"

namespace rt
{
namespace
{
std::uint32_t be32(const unsigned char* p) { return (std::uint32_t)p[0] << 24 | (std::uint32_t)p[1] << 16 | (std::uint32_t)p[2] << 8 | p[3]; }

int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}
} // namespace

bool LoadPNG(const char* filename, Image& res)
{
    FILE* f = fopen(filename, "rb");
    if (!f) return false;
    std::vector<unsigned char> file;
    unsigned char buf[65536];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + got);
    fclose(f);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 + 25 || memcmp(file.data(), sig, 8)) return false;

    std::uint32_t w = 0, h = 0;
    int depth = 0, color = 0, interlace = 0;
    std::vector<unsigned char> idat, plte, trns;
    size_t pos = 8;
    bool have_ihdr = false, done = false;
    while (!done && pos + 12 <= file.size())
    {
        std::uint32_t len = be32(&file[pos]);
        const unsigned char* type = &file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) return false;
        const unsigned char* data = &file[pos + 8];
        if (!memcmp(type, "IHDR", 4))
        {
            if (len != 13) return false;
            w = be32(data); h = be32(data + 4);
            depth = data[8]; color = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0) return false;
            have_ihdr = true;
        }
        else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) done = true;
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || w == 0 || h == 0 || interlace > 1) return false;
    int samples;   // samples per pixel in the file
    switch (color)
    {
    case 0: samples = 1; break;
    case 2: samples = 3; break;
    case 3: samples = 1; break;
    case 4: samples = 2; break;
    case 6: samples = 4; break;
    default: return false;
    }
    if (!(depth == 8 || depth == 16 || ((color == 0 || color == 3) && (depth == 1 || depth == 2 || depth == 4)))) return false;
    if (color == 3 && (depth == 16 || plte.empty())) return false;
    const size_t bpp_bits = (size_t)samples * depth;
    const size_t fb = bpp_bits >= 8 ? bpp_bits / 8 : 1;   // filter byte distance
    // the reduced images of the file: one (the whole image), or the seven Adam7 passes (PNG spec 8.2)
    struct Pass { std::uint32_t x0, y0, dx, dy, pw, ph; size_t stride; };
    std::vector<Pass> passes;
    if (!interlace) passes.push_back({0, 0, 1, 1, w, h, 0});
    else
    {
        static const std::uint32_t X0[7] = {0, 4, 0, 2, 0, 1, 0}, Y0[7] = {0, 0, 4, 0, 2, 0, 1};
        static const std::uint32_t DX[7] = {8, 8, 4, 4, 2, 2, 1}, DY[7] = {8, 8, 8, 4, 4, 2, 2};
        for (int i = 0; i < 7; ++i)
        {
            const std::uint32_t pw = w > X0[i] ? (w - X0[i] + DX[i] - 1) / DX[i] : 0, ph = h > Y0[i] ? (h - Y0[i] + DY[i] - 1) / DY[i] : 0;
            if (pw && ph) passes.push_back({X0[i], Y0[i], DX[i], DY[i], pw, ph, 0});
        }
    }
    size_t raw_size = 0;
    for (Pass& ps : passes) { ps.stride = ((size_t)ps.pw * bpp_bits + 7) / 8; raw_size += (ps.stride + 1) * ps.ph; }
    std::vector<unsigned char> raw(raw_size);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) return false;

    // unfilter every reduced image on its own (PNG spec 9.2) and scatter its samples (file bit depth) to their pixels
    std::vector<std::uint16_t> smp((size_t)w * h * samples);
    size_t in_pos = 0;
    for (const Pass& ps : passes)
    {
        std::vector<unsigned char> img(ps.stride * ps.ph);
        for (std::uint32_t y = 0; y < ps.ph; ++y)
        {
            const unsigned char* in = &raw[in_pos + (ps.stride + 1) * y];
            unsigned char* out = &img[ps.stride * y];
            const unsigned char* up = y ? &img[ps.stride * (y - 1)] : nullptr;
            int ft = in[0];
            for (size_t x = 0; x < ps.stride; ++x)
            {
                int a = x >= fb ? out[x - fb] : 0;
                int b = up ? up[x] : 0;
                int c = (up && x >= fb) ? up[x - fb] : 0;
                int v = in[1 + x];
                switch (ft)
                {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: return false;
                }
                out[x] = (unsigned char)v;
            }
            for (std::uint32_t x = 0; x < ps.pw; ++x)
                for (int k = 0; k < samples; ++k)
                {
                    const size_t i = (size_t)x * samples + k;
                    unsigned v;
                    if (depth == 8) v = out[i];
                    else if (depth == 16) v = (unsigned)out[2 * i] << 8 | out[2 * i + 1];
                    else
                    {
                        const size_t bit = i * depth;
                        v = (out[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
                    }
                    smp[((size_t)(ps.y0 + y * ps.dy) * w + (ps.x0 + x * ps.dx)) * samples + k] = (std::uint16_t)v;
                }
        }
        in_pos += (ps.stride + 1) * ps.ph;
    }

    // expand to 8-bit samples per pixel, n output channels (stb conventions)
    int n = samples;
    if (color == 3) n = trns.empty() ? 3 : 4;
    else if (!trns.empty() && (color == 0 || color == 2)) n = samples + 1;
    std::vector<unsigned char> px((size_t)w * h * n);
    for (std::uint32_t y = 0; y < h; ++y)
    {
        for (std::uint32_t x = 0; x < w; ++x)
        {
            unsigned s8[4] = {0, 0, 0, 0};
            unsigned s16[4] = {0, 0, 0, 0};
            for (int k = 0; k < samples; ++k)
            {
                const unsigned v = smp[((size_t)y * w + x) * samples + k];
                s16[k] = v;
                if (depth == 8) s8[k] = v;
                else if (depth == 16) s8[k] = v >> 8;
                // stb scales 1/2/4-bit grey to 0..255; palette indices stay as they are
                else s8[k] = color == 3 ? v : v * (depth == 1 ? 255u : depth == 2 ? 85u : 17u);
            }
            unsigned char* o = &px[((size_t)y * w + x) * n];
            if (color == 3)
            {
                unsigned idx = s8[0];
                if ((size_t)idx * 3 + 2 >= plte.size()) return false;
                o[0] = plte[idx * 3]; o[1] = plte[idx * 3 + 1]; o[2] = plte[idx * 3 + 2];
                if (n == 4) o[3] = idx < trns.size() ? trns[idx] : 255;
            }
            else
            {
                for (int k = 0; k < samples; ++k) o[k] = (unsigned char)s8[k];
                if (n == samples + 1)   // colour-key transparency
                {
                    bool key = true;
                    for (int k = 0; k < samples && key; ++k)
                    {
                        if (trns.size() < (size_t)(2 * k + 2)) { key = false; break; }
                        unsigned tv = (unsigned)trns[2 * k] << 8 | trns[2 * k + 1];
                        unsigned cmp = depth == 16 ? s16[k] : (depth == 8 ? s8[k] : s16[k]);
                        key = cmp == tv;
                    }
                    o[samples] = key ? 0 : 255;
                }
            }
        }
    }

    res.width = w;
    res.height = h;
    res.data.resize((size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; ++i)   // LoadSTB channel mapping, image_loader.cpp:47-59
    {
        const unsigned char* p = &px[i * n];
        std::uint32_t r = p[0], g = n > 1 ? p[1] : 0, b = n > 2 ? p[2] : 0, a = n > 3 ? p[3] : 0;
        res.data[i] = r | (g << 8) | (b << 16) | (a << 24);
    }
    return true;
}
} // namespace rt
