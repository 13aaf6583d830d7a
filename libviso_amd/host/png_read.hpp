// png_read.hpp — minimal PNG decoder for the KITTI odometry images the
// reference reads with cv::imread(name, CV_LOAD_IMAGE_GRAYSCALE)
// (src/viso.h:92-93): 8-bit, non-interlaced; grayscale (colour type 0) as is,
// RGB / RGBA / gray+alpha reduced to gray with OpenCV's integer BGR2GRAY
// weights.  Own RFC 1950/1951 inflate (stored, fixed and dynamic Huffman
// blocks); CRCs are not verified.  Host-side file I/O only.  The file is NOT trusted: IHDR must be the first chunk
// and 13 bytes long, images above PNG_MAX_PIXELS are refused before anything is reserved, inflate stops as soon as
// the output exceeds the size the header implies, stored blocks check LEN against NLEN.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace viso {
namespace png_detail {

constexpr size_t PNG_MAX_PIXELS = (size_t)64 << 20;   // 64 Mpx: far above any KITTI frame (1241 x 376)

struct BitReader {
    const uint8_t* p; size_t n, pos = 0; uint32_t bitbuf = 0; int bitcnt = 0; bool fail = false;
    BitReader(const uint8_t* d, size_t len) : p(d), n(len) {}
    int bit() {
        if (!bitcnt) { if (pos >= n) { fail = true; return 0; } bitbuf = p[pos++]; bitcnt = 8; }
        const int b = bitbuf & 1; bitbuf >>= 1; --bitcnt; return b;
    }
    uint32_t bits(int k) { uint32_t v = 0; for (int i = 0; i < k; ++i) v |= (uint32_t)bit() << i; return v; }
    void align() { bitcnt = 0; }
};

struct Huff {   // canonical Huffman decoding table (counts per length + symbols in order)
    uint16_t count[16], symbol[288];
    void build(const uint8_t* len, int n) {
        for (int i = 0; i < 16; ++i) count[i] = 0;
        for (int i = 0; i < n; ++i) count[len[i]]++;
        count[0] = 0;
        uint16_t offs[16]; offs[1] = 0;
        for (int i = 1; i < 15; ++i) offs[i + 1] = offs[i] + count[i];
        for (int i = 0; i < n; ++i) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
    }
    int decode(BitReader& br) const {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; ++l) {
            code |= br.bit();
            const int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
            if (br.fail) return -1;
        }
        return -1;
    }
};

// max_out: the decoder fails once it would produce more than that many bytes (a small IDAT cannot blow up memory)
inline bool inflate(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t max_out) {
    static const uint16_t lbase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
    static const uint16_t lext[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
    static const uint16_t dbase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
    static const uint16_t dext[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
    if (n < 2) return false;
    BitReader br(src + 2, n - 2);   // skip the zlib header (CMF, FLG)
    for (;;) {
        const int last = br.bit();
        const uint32_t type = br.bits(2);
        if (br.fail) return false;
        if (type == 0) {
            br.align();
            if (br.pos + 4 > br.n) return false;
            const uint32_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8);
            const uint32_t nlen = br.p[br.pos + 2] | (br.p[br.pos + 3] << 8);
            if ((len ^ nlen) != 0xffffu) return false;
            br.pos += 4;
            if (br.pos + len > br.n || out.size() + len > max_out) return false;
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huff lit, dist;
            uint8_t lens[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; ++i) lens[i] = 8;
                for (; i < 256; ++i) lens[i] = 9;
                for (; i < 280; ++i) lens[i] = 7;
                for (; i < 288; ++i) lens[i] = 8;
                lit.build(lens, 288);
                for (i = 0; i < 30; ++i) lens[i] = 5;
                dist.build(lens, 30);
            } else {
                static const uint8_t order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
                const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
                if (nlen > 286 || ndist > 30) return false;
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)br.bits(3);
                Huff clh; clh.build(cl, 19);
                int idx = 0;
                while (idx < nlen + ndist) {
                    const int sym = clh.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) lens[idx++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (!idx) return false; val = lens[idx - 1]; rep = 3 + (int)br.bits(2); }
                        else if (sym == 17) rep = 3 + (int)br.bits(3);
                        else rep = 11 + (int)br.bits(7);
                        if (idx + rep > nlen + ndist) return false;
                        while (rep--) lens[idx++] = (uint8_t)val;
                    }
                }
                lit.build(lens, nlen);
                dist.build(lens + nlen, ndist);
            }
            for (;;) {
                const int sym = lit.decode(br);
                if (sym < 0 || br.fail) return false;
                if (sym < 256) { if (out.size() >= max_out) return false; out.push_back((uint8_t)sym); }
                else if (sym == 256) break;
                else {
                    const int s = sym - 257;
                    if (s >= 29) return false;
                    const int len = lbase[s] + (int)br.bits(lext[s]);
                    const int ds = dist.decode(br);
                    if (ds < 0 || ds >= 30) return false;
                    const size_t d = dbase[ds] + br.bits(dext[ds]);
                    if (d > out.size() || out.size() + (size_t)len > max_out) return false;
                    const size_t from = out.size() - d;
                    for (int i = 0; i < len; ++i) out.push_back(out[from + i]);
                }
            }
        } else return false;
        if (last) break;
    }
    return !br.fail;
}

inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace png_detail

// rows/cols/data of an 8-bit grayscale rendition; empty data on any failure.
inline bool read_png_gray(const std::string& file_name, int& rows, int& cols, std::vector<uint8_t>& gray) {
    using namespace png_detail;
    gray.clear(); rows = cols = 0;
    FILE* fp = std::fopen(file_name.c_str(), "rb");
    if (!fp) return false;
    std::vector<uint8_t> f;
    uint8_t buf[65536];
    size_t k;
    while ((k = std::fread(buf, 1, sizeof buf, fp)) > 0) f.insert(f.end(), buf, buf + k);
    std::fclose(fp);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (f.size() < 8 + 25) return false;
    for (int i = 0; i < 8; ++i) if (f[(size_t)i] != sig[i]) return false;
    auto be32 = [&](size_t o) { return ((uint32_t)f[o] << 24) | ((uint32_t)f[o + 1] << 16) | ((uint32_t)f[o + 2] << 8) | f[o + 3]; };
    uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> z;
    size_t o = 8;
    bool have_ihdr = false;
    while (o + 12 <= f.size()) {
        const uint32_t len = be32(o);
        const std::string type((const char*)&f[o + 4], 4);
        if ((size_t)len > f.size() || o + 12 + (size_t)len > f.size()) return false;
        if (!have_ihdr) {   // the first chunk must be a 13-byte IHDR
            if (type != "IHDR" || len != 13) return false;
            w = be32(o + 8); h = be32(o + 12); depth = f[o + 16]; ctype = f[o + 17]; interlace = f[o + 20];
            have_ihdr = true;
        }
        else if (type == "IHDR") return false;
        else if (type == "IDAT") z.insert(z.end(), f.begin() + (long)(o + 8), f.begin() + (long)(o + 8 + len));
        else if (type == "IEND") break;
        o += 12 + len;
    }
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!w || !h || depth != 8 || !ch || interlace || w > 65535 || h > 65535 || (size_t)w * h > PNG_MAX_PIXELS) return false;
    const size_t stride = (size_t)w * ch, raw_size = (size_t)h * (stride + 1);
    std::vector<uint8_t> raw;
    raw.reserve(raw_size);
    if (!inflate(z.data(), z.size(), raw, raw_size)) return false;
    if (raw.size() < raw_size) return false;
    std::vector<uint8_t> img((size_t)h * stride), zero(stride, 0);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* s = &raw[(size_t)y * (stride + 1)];
        const int ft = s[0];
        uint8_t* d = &img[(size_t)y * stride];
        const uint8_t* up = y ? &img[(size_t)(y - 1) * stride] : zero.data();
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= (size_t)ch ? d[x - ch] : 0, b = up[x], c = x >= (size_t)ch ? up[x - ch] : 0;
            int v = s[1 + x];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: return false;
            }
            d[x] = (uint8_t)v;
        }
    }
    gray.resize((size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        const uint8_t* p = &img[i * ch];
        if (ch <= 2) gray[i] = p[0];
        else gray[i] = (uint8_t)((p[0] * 4899 + p[1] * 9617 + p[2] * 1868 + 8192) >> 14);   // OpenCV RGB2GRAY fixed point
    }
    rows = (int)h; cols = (int)w;
    return true;
}

}  // namespace viso
