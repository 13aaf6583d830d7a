// png_read.hpp — PNG decoder for the KITTI odometry images the reference reads with
// cv::imread(name, CV_LOAD_IMAGE_GRAYSCALE) (src/viso.h:92-93): 8-bit, non-interlaced; grayscale (colour
// type 0) as is, RGB / RGBA / gray+alpha reduced to gray with OpenCV's integer BGR2GRAY weights.
//
// The sequence runner is bound by this file, not by the GPU (a rank's kernels need ~8 us per 1241 x 376 frame,
// inflating its two PNGs takes milliseconds), so the inflate is a table-driven one: a 64-bit bit buffer refilled
// eight bytes at a time, an 11-bit first-level table for literal/length codes and an 8-bit one for distances with
// second-level tables behind the long codes, entries that carry base value + extra-bit count, word-wide match
// copies.  Own code (RFC 1950/1951); CRCs and the Adler checksum are not verified.
//
// The file is NOT trusted: IHDR must be the first chunk and 13 bytes long, images above PNG_MAX_PIXELS are refused
// before anything is reserved, the decoder never writes past the size the header implies and never reads past the
// IDAT bytes, over-subscribed Huffman codes are refused, unassigned codes fail when met, stored blocks check LEN
// against NLEN.  Host-side file I/O only; thread safe (no static state besides constant tables).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace viso {
namespace png_detail {

constexpr size_t PNG_MAX_PIXELS = (size_t)64 << 20;   // 64 Mpx: far above any KITTI frame (1241 x 376)

// ---- inflate -------------------------------------------------------------------------------------------------
// Table entry (uint32): bits 0-7 code bits to consume | bits 8-12 extra bits (or sub-table index bits) |
// bits 13-15 kind | bits 16-31 value (literal, base length / distance, or sub-table offset).
enum : uint32_t { K_INVALID = 0, K_LIT = 1, K_BASE = 2, K_EOB = 3, K_SUB = 4 };
inline uint32_t entry(uint32_t kind, uint32_t value, uint32_t extra, uint32_t len) { return len | (extra << 8) | (kind << 13) | (value << 16); }

struct Table {
    std::vector<uint32_t> e;
    int root = 0;
};

// lens[0..n): code lengths (0 = unused).  kind_of(sym) fills the entry of a symbol (without its code length).
// Returns false for an over-subscribed code.  An incomplete code leaves K_INVALID entries.
template <class F>
inline bool build_table(Table& t, const uint8_t* lens, int n, int root, F sym_entry) {
    int count[16] = {0};
    for (int i = 0; i < n; ++i) count[lens[i]]++;
    count[0] = 0;
    int maxlen = 0;
    long left = 1;
    for (int l = 1; l < 16; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return false;                   // over-subscribed
        if (count[l]) maxlen = l;
    }
    t.root = root;
    const uint32_t rsize = 1u << root;
    t.e.assign(rsize, 0);
    if (!maxlen) return true;
    uint32_t next[16];
    uint32_t code = 0;
    for (int l = 1; l < 16; ++l) { next[l] = code; code = (code + (uint32_t)count[l]) << 1; }
    auto rev = [](uint32_t c, int l) { uint32_t r = 0; for (int i = 0; i < l; ++i) { r = (r << 1) | (c & 1); c >>= 1; } return r; };
    // pass 1: short codes straight into the root table; per root prefix the longest code behind it
    std::vector<uint8_t> sub_bits;
    if (maxlen > root) sub_bits.assign(rsize, 0);
    uint32_t nx[16];
    std::memcpy(nx, next, sizeof nx);
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t r = rev(nx[l]++, l);
        if (l <= root) {
            const uint32_t en = sym_entry(s) | (uint32_t)l;
            for (uint32_t i = r; i < rsize; i += 1u << l) t.e[i] = en;
        } else {
            uint8_t& sb = sub_bits[r & (rsize - 1)];
            if (l - root > sb) sb = (uint8_t)(l - root);
        }
    }
    if (maxlen <= root) return true;
    // pass 2: allocate the second-level tables, fill the long codes
    for (uint32_t p = 0; p < rsize; ++p)
        if (sub_bits[p]) {
            const uint32_t off = (uint32_t)t.e.size();
            if (off > 0xffff) return false;
            t.e[p] = entry(K_SUB, off, sub_bits[p], (uint32_t)root);
            t.e.resize(off + (1u << sub_bits[p]), 0);
        }
    std::memcpy(nx, next, sizeof nx);
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t r = rev(nx[l]++, l);
        if (l <= root) continue;
        const uint32_t p = r & (rsize - 1);
        const uint32_t off = t.e[p] >> 16, sb = sub_bits[p];
        const uint32_t en = sym_entry(s) | (uint32_t)(l - root);
        for (uint32_t i = r >> root; i < (1u << sb); i += 1u << (l - root)) t.e[off + i] = en;
    }
    return true;
}

inline uint32_t litlen_entry(int s) {
    static const uint16_t lbase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
    static const uint8_t lext[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
    if (s < 256) return entry(K_LIT, (uint32_t)s, 0, 0);
    if (s == 256) return entry(K_EOB, 0, 0, 0);
    if (s - 257 >= 29) return entry(K_INVALID, 0, 0, 0);
    return entry(K_BASE, lbase[s - 257], lext[s - 257], 0);
}
inline uint32_t dist_entry(int s) {
    static const uint16_t dbase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
    static const uint8_t dext[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
    if (s >= 30) return entry(K_INVALID, 0, 0, 0);
    return entry(K_BASE, dbase[s], dext[s], 0);
}

struct Bits {
    const uint8_t* in; const uint8_t* end;
    uint64_t bb = 0; int bc = 0;          // bc may go negative: the stream was read past its end
    Bits(const uint8_t* p, size_t n) : in(p), end(p + n) {}
    inline void refill() {                // afterwards bc >= 56 unless the input is exhausted
        if (end - in >= 8) {
            uint64_t w;
            std::memcpy(&w, in, 8);       // little-endian host (x86-64)
            bb |= w << bc;
            in += (63 - bc) >> 3;
            bc |= 56;
        } else {
            while (bc <= 56 && in < end) { bb |= (uint64_t)*in++ << bc; bc += 8; }
        }
    }
    inline uint32_t peek(int n) const { return (uint32_t)(bb & (((uint64_t)1 << n) - 1)); }
    inline void drop(int n) { bb >>= n; bc -= n; }
    inline uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
};

// One symbol of `t`: the entry with the bits of its code dropped (K_INVALID on an unassigned code).
inline uint32_t decode(Bits& b, const Table& t) {
    uint32_t e = t.e[b.peek(t.root)];
    if ((e >> 13 & 7) == K_SUB) {
        b.drop((int)(e & 0xff));
        e = t.e[(e >> 16) + b.peek((int)(e >> 8 & 31))];
    }
    b.drop((int)(e & 0xff));
    return e;
}

// zlib stream -> out[0..out_size); false unless exactly representable within out_size bytes (shorter output is
// reported through *produced; longer output fails).  `out` needs 8 bytes of slack behind out_size.
inline bool inflate(const uint8_t* src, size_t n, uint8_t* out, size_t out_size, size_t* produced) {
    if (n < 2) return false;
    Bits b(src + 2, n - 2);               // skip the zlib header (CMF, FLG)
    uint8_t* op = out;
    uint8_t* const oend = out + out_size;
    Table lit, dist;
    Table fixed_lit, fixed_dist;
    for (;;) {
        b.refill();
        const uint32_t last = b.take(1), type = b.take(2);
        if (b.bc < 0) return false;
        if (type == 0) {
            b.drop(b.bc & 7);                                   // to the byte boundary
            b.in -= b.bc >> 3; b.bb = 0; b.bc = 0;              // give the whole bytes of the bit buffer back
            if (b.end - b.in < 4) return false;
            const uint32_t len = b.in[0] | (b.in[1] << 8), nlen = b.in[2] | (b.in[3] << 8);
            if ((len ^ nlen) != 0xffffu) return false;
            b.in += 4;
            if ((size_t)(b.end - b.in) < len || (size_t)(oend - op) < len) return false;
            std::memcpy(op, b.in, len);
            op += len; b.in += len;
        } else if (type == 1 || type == 2) {
            const Table *tl, *td;
            if (type == 1) {
                if (fixed_lit.e.empty()) {
                    uint8_t lens[288];
                    int i = 0;
                    for (; i < 144; ++i) lens[i] = 8;
                    for (; i < 256; ++i) lens[i] = 9;
                    for (; i < 280; ++i) lens[i] = 7;
                    for (; i < 288; ++i) lens[i] = 8;
                    build_table(fixed_lit, lens, 288, 11, litlen_entry);
                    for (i = 0; i < 30; ++i) lens[i] = 5;
                    build_table(fixed_dist, lens, 30, 8, dist_entry);
                }
                tl = &fixed_lit; td = &fixed_dist;
            } else {
                static const uint8_t order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
                const int nlen = (int)b.take(5) + 257, ndist = (int)b.take(5) + 1, ncode = (int)b.take(4) + 4;
                if (nlen > 286 || ndist > 30) return false;
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; ++i) { if (i % 12 == 0) b.refill(); cl[order[i]] = (uint8_t)b.take(3); }
                if (b.bc < 0) return false;
                Table clt;
                if (!build_table(clt, cl, 19, 7, [](int s) { return entry(K_LIT, (uint32_t)s, 0, 0); })) return false;
                uint8_t lens[320];
                int idx = 0;
                while (idx < nlen + ndist) {
                    b.refill();
                    const uint32_t e = decode(b, clt);
                    if ((e >> 13 & 7) != K_LIT) return false;
                    const int sym = (int)(e >> 16);
                    if (sym < 16) lens[idx++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (!idx) return false; val = lens[idx - 1]; rep = 3 + (int)b.take(2); }
                        else if (sym == 17) rep = 3 + (int)b.take(3);
                        else rep = 11 + (int)b.take(7);
                        if (idx + rep > nlen + ndist) return false;
                        while (rep--) lens[idx++] = (uint8_t)val;
                    }
                    if (b.bc < 0) return false;
                }
                if (!lens[256]) return false;                   // no end-of-block code
                if (!build_table(lit, lens, nlen, 11, litlen_entry) || !build_table(dist, lens + nlen, ndist, 8, dist_entry)) return false;
                tl = &lit; td = &dist;
            }
            for (;;) {
                b.refill();                                     // >= 56 bits: a whole length/distance pair needs <= 48
                uint32_t e = decode(b, *tl);
                uint32_t kind = e >> 13 & 7;
                if (kind == K_LIT) {
                    if (op >= oend) return false;
                    *op++ = (uint8_t)(e >> 16);
                    // up to two more literals on the same refill (15 bits each at most)
                    e = decode(b, *tl); kind = e >> 13 & 7;
                    if (kind == K_LIT) {
                        if (op >= oend) return false;
                        *op++ = (uint8_t)(e >> 16);
                        e = decode(b, *tl); kind = e >> 13 & 7;
                        if (kind == K_LIT) {
                            if (op >= oend) return false;
                            *op++ = (uint8_t)(e >> 16);
                            if (b.bc < 0) return false;
                            continue;
                        }
                    }
                    b.refill();                                 // up to 45 bits went into the literals
                }
                if (kind == K_EOB) { if (b.bc < 0) return false; break; }
                if (kind != K_BASE) return false;
                const size_t len = (e >> 16) + b.take((int)(e >> 8 & 31));
                const uint32_t de = decode(b, *td);
                if ((de >> 13 & 7) != K_BASE) return false;
                const size_t d = (de >> 16) + b.take((int)(de >> 8 & 31));
                if (b.bc < 0) return false;
                if (d > (size_t)(op - out) || len > (size_t)(oend - op)) return false;
                const uint8_t* from = op - d;
                if (d >= 8) {                                   // word-wide copy (the slack behind out_size takes the overrun)
                    uint8_t* q = op;
                    uint8_t* const qe = op + len;
                    do { uint64_t w; std::memcpy(&w, from, 8); std::memcpy(q, &w, 8); from += 8; q += 8; } while (q < qe);
                } else if (d == 1) {
                    std::memset(op, *from, len);
                } else {
                    for (size_t i = 0; i < len; ++i) op[i] = from[i];
                }
                op += len;
            }
        } else return false;
        if (last) break;
    }
    *produced = (size_t)(op - out);
    return true;
}

inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// One row of the five PNG filters (RFC 2083 6): s = filtered bytes, up = reconstructed previous row (zeros for
// the first), d = out.  bpp = bytes per pixel.
inline bool unfilter_row(int ft, const uint8_t* s, const uint8_t* up, uint8_t* d, size_t stride, size_t bpp) {
    switch (ft) {
    case 0: std::memcpy(d, s, stride); return true;
    case 1:
        for (size_t x = 0; x < bpp && x < stride; ++x) d[x] = s[x];
        for (size_t x = bpp; x < stride; ++x) d[x] = (uint8_t)(s[x] + d[x - bpp]);
        return true;
    case 2:
        for (size_t x = 0; x < stride; ++x) d[x] = (uint8_t)(s[x] + up[x]);
        return true;
    case 3:
        for (size_t x = 0; x < bpp && x < stride; ++x) d[x] = (uint8_t)(s[x] + (up[x] >> 1));
        for (size_t x = bpp; x < stride; ++x) d[x] = (uint8_t)(s[x] + ((d[x - bpp] + up[x]) >> 1));
        return true;
    case 4:
        for (size_t x = 0; x < bpp && x < stride; ++x) d[x] = (uint8_t)(s[x] + up[x]);   // paeth(0, b, 0) = b
        if (bpp == 1) {                                         // the grayscale case: a, c carried in registers
            // branch free: the predictor choice is data dependent noise to a branch predictor, and `a` is a serial chain
            int a = stride ? d[0] : 0, c = stride ? up[0] : 0;
            for (size_t x = 1; x < stride; ++x) {
                const int bb = up[x];
                const int p = bb - c, q = a - c;                // pa = |p|, pb = |q|, pc = |p + q|
                const int pa = p < 0 ? -p : p, pb = q < 0 ? -q : q, pc = p + q < 0 ? -(p + q) : p + q;
                const int not_a = -(int)((pa > pb) | (pa > pc));
                const int take_c = -(int)(pb > pc);
                const int bc_ = bb ^ ((bb ^ c) & take_c);
                const int pred = a ^ ((a ^ bc_) & not_a);
                a = (uint8_t)(s[x] + pred);
                d[x] = (uint8_t)a;
                c = bb;
            }
        } else {
            for (size_t x = bpp; x < stride; ++x) d[x] = (uint8_t)(s[x] + paeth(d[x - bpp], up[x], up[x - bpp]));
        }
        return true;
    default: return false;
    }
}

inline bool read_file(const std::string& file_name, std::vector<uint8_t>& f) {
    FILE* fp = std::fopen(file_name.c_str(), "rb");
    if (!fp) return false;
    bool ok = std::fseek(fp, 0, SEEK_END) == 0;
    const long size = ok ? std::ftell(fp) : -1;
    ok = ok && size >= 0 && std::fseek(fp, 0, SEEK_SET) == 0;
    if (ok) {
        f.resize((size_t)size);
        ok = size == 0 || std::fread(f.data(), 1, (size_t)size, fp) == (size_t)size;
    }
    std::fclose(fp);
    return ok;
}

}  // namespace png_detail

// Decodes `file_name` to 8-bit gray.  provide(rows, cols) returns where the rows x cols bytes go, or nullptr to
// refuse the geometry (then the call fails).  false on any failure.
template <class Provide>
inline bool read_png_gray_with(const std::string& file_name, int& rows, int& cols, Provide provide) {
    using namespace png_detail;
    rows = cols = 0;
    std::vector<uint8_t> f;
    if (!read_file(file_name, f)) return false;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (f.size() < 8 + 25 || std::memcmp(f.data(), sig, 8) != 0) return false;
    auto be32 = [&](size_t o) { return ((uint32_t)f[o] << 24) | ((uint32_t)f[o + 1] << 16) | ((uint32_t)f[o + 2] << 8) | f[o + 3]; };
    uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    // the IDAT payloads are moved together in place (towards the front of f): no second buffer
    size_t zbeg = 0, zlen = 0;
    size_t o = 8;
    bool have_ihdr = false;
    while (o + 12 <= f.size()) {
        const uint32_t len = be32(o);
        if ((size_t)len > f.size() || o + 12 + (size_t)len > f.size()) return false;
        const bool is = std::memcmp(&f[o + 4], "IDAT", 4) == 0;
        if (!have_ihdr) {   // the first chunk must be a 13-byte IHDR
            if (std::memcmp(&f[o + 4], "IHDR", 4) != 0 || len != 13) return false;
            w = be32(o + 8); h = be32(o + 12); depth = f[o + 16]; ctype = f[o + 17]; interlace = f[o + 20];
            have_ihdr = true;
        }
        else if (std::memcmp(&f[o + 4], "IHDR", 4) == 0) return false;
        else if (is) {
            if (!zlen) zbeg = o + 8;
            else std::memmove(&f[zbeg + zlen], &f[o + 8], len);
            zlen += len;
        }
        else if (std::memcmp(&f[o + 4], "IEND", 4) == 0) break;
        o += 12 + (size_t)len;
    }
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!w || !h || depth != 8 || !ch || interlace || w > 65535 || h > 65535 || (size_t)w * h > PNG_MAX_PIXELS) return false;
    // the destination is asked for BEFORE anything is sized from the (untrusted) header: a caller that knows the geometry
    // (read_png_gray_to) refuses another one here, before up to 256 MB are allocated and inflated for nothing
    uint8_t* dst = provide((int)h, (int)w);
    if (!dst) return false;
    const size_t stride = (size_t)w * ch, raw_size = (size_t)h * (stride + 1);
    std::vector<uint8_t> raw(raw_size + 8);
    size_t produced = 0;
    if (!inflate(f.data() + zbeg, zlen, raw.data(), raw_size, &produced) || produced < raw_size) return false;
    std::vector<uint8_t> zero(stride, 0), rowbuf;
    if (ch == 1) {   // straight into the destination
        for (uint32_t y = 0; y < h; ++y) {
            const uint8_t* s = &raw[(size_t)y * (stride + 1)];
            if (!unfilter_row(s[0], s + 1, y ? dst + (size_t)(y - 1) * stride : zero.data(), dst + (size_t)y * stride, stride, 1)) return false;
        }
    } else {
        std::vector<uint8_t> img((size_t)h * stride);
        for (uint32_t y = 0; y < h; ++y) {
            const uint8_t* s = &raw[(size_t)y * (stride + 1)];
            if (!unfilter_row(s[0], s + 1, y ? &img[(size_t)(y - 1) * stride] : zero.data(), &img[(size_t)y * stride], stride, (size_t)ch)) return false;
        }
        for (size_t i = 0; i < (size_t)w * h; ++i) {
            const uint8_t* p = &img[i * ch];
            if (ch == 2) dst[i] = p[0];
            else dst[i] = (uint8_t)((p[0] * 4899 + p[1] * 9617 + p[2] * 1868 + 8192) >> 14);   // OpenCV RGB2GRAY fixed point
        }
    }
    rows = (int)h; cols = (int)w;
    return true;
}

// rows/cols/data of an 8-bit grayscale rendition; empty data on any failure.
inline bool read_png_gray(const std::string& file_name, int& rows, int& cols, std::vector<uint8_t>& gray) {
    gray.clear();
    const bool ok = read_png_gray_with(file_name, rows, cols, [&](int r, int c) { gray.resize((size_t)r * c); return gray.data(); });
    if (!ok) { gray.clear(); rows = cols = 0; }
    return ok;
}

// The same into caller memory of a known geometry (a pinned upload buffer): fails when the file's size differs.
// *other_geometry (may be null) = the file is a readable PNG header of ANOTHER size: a different failure from "cannot be
// opened / decoded" (the end of a sequence, src/viso.h:94-96) for callers that must tell the two apart.
inline bool read_png_gray_to(const std::string& file_name, int rows, int cols, uint8_t* dst, bool* other_geometry = nullptr) {
    int r = 0, c = 0;
    if (other_geometry) *other_geometry = false;
    return read_png_gray_with(file_name, r, c, [&](int fr, int fc) {
        if (fr == rows && fc == cols) return dst;
        if (other_geometry) *other_geometry = true;
        return static_cast<uint8_t*>(nullptr);
    });
}

}  // namespace viso
