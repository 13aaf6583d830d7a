// harris.hip — binned Harris corner detector on device: the reference's
// HarrisBinnedFeatureDetector::detectImpl (src/viso.cpp:926-975) with
// cv::cornerHarris(image, R, blockSize 3, ksize 5, k, BORDER_DEFAULT) restated
// (SURVEY.md 8(f) row 2).  The reference never initialises its k
// (src/viso.cpp:915-919,978) and leaves the order inside a bin to
// std::nth_element (:963); here k is an explicit argument and a bin's corners
// are ordered by (|response| desc, push order asc).  Arithmetic contract (shared
// with the oracle, oracle/viso_oracle.c): cv::cornerHarris in OpenCV's evaluation
// order — Sobel with the scale folded into the float smoothing taps, row pass then
// column pass (symmetric grouping), cov products in float, box filter as row sums then
// column sums with BORDER_REFLECT_101 on the cov image, R = (float)((double)(a*c - b*b)
// - k*(a+c)*(a+c)).  OpenCV's running column sums are the one thing an algorithm-level
// restatement cannot pin (see the oracle's header): parity with a real OpenCV is unpinned.
#include "common.h"
#include <stdlib.h>

// ---- the response of a tile, one pass, registers only ---------------------------------------------------------
// Arithmetic contract = the oracle's restatement of cv::cornerHarris in OpenCV's evaluation order (oracle/viso_oracle.c,
// harris_cov / oracle_harris_response): scale folded into the float smoothing taps, row pass then column pass with
// the symmetric / anti-symmetric grouping of SymmColumnFilter, cov products in float, box filter as row sums then
// column sums, BORDER_REFLECT_101 on the source for the Sobel passes and on the cov image for the box.
//
// Work decomposition.  A tile is tw <= 62 columns wide and th rows tall.  Its uint8 pixels (+ 3-pixel halo) go to LDS
// once.  Then wave b walks DOWN band b of the tile's rows with lane = column (lane lx <-> image column tx0 - 1 + lx, so
// the tile plus its one-column cov ring fits one wave), keeping everything a column needs in registers: a five-row
// window of the two row-pass results (H: derivative taps, exact; G: scaled smoothing taps, the float chain), from which
// each step yields Dx, Dy and the three cov products of one more row; the row sums of the box filter come from the
// neighbouring lanes (two wave shifts per channel); a three-row window of those gives the box sums and the response.
// No intermediate image (Dx, Dy, cov, row sums) ever exists in memory; per output pixel the walk costs ~75 vector
// instructions plus 6 / rows_per_band of warm-up.
#define HW_MAXW 62                  // tile width limit: tw + 2 ring columns = one wave
#define HW_DIRECT_MAXW 58         // widest bin the LDS-free walk takes: tw + 6 pixel columns on 64 lanes
#define HW_PITCH 72                 // LDS row pitch of the uint8 tile (tw + 6 <= 68 used)
#define HW_THREADS 256

__device__ __forceinline__ int h_reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// uint8 tile rows ty0-3 .. ty0+th+2, columns tx0-3 .. tx0+tw+2 -> s_img (BORDER_REFLECT_101 outside the image)
// NW waves share the rows of one tile (NW = 1: a wave loads its own tile)
template <int NW>
__device__ __forceinline__ void harris_load_tile(const uint8_t* __restrict__ im, int rows, int cols, int tx0, int ty0,
                                                 int tw, int th, unsigned char* s_img) {
    const int wave = NW == 1 ? 0 : (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool inside = tx0 >= 3 && ty0 >= 3 && tx0 + tw + 3 <= cols && ty0 + th + 3 <= rows;
    const int ndw = (tw + 6 + 3) / 4;                             // dwords per LDS row
    if (tx0 >= 3 && ndw <= 16 && tx0 - 3 + 4 * ndw <= cols) {
        // interior in x (rows may still reflect): dword loads straight from the tile's first byte (KITTI's 1241-pixel
        // rows put that at any alignment: the hardware takes unaligned dword loads), 16 lanes per row, four rows per
        // wave instruction, eight instructions in flight
        const int sub = lane >> 4, dw = lane & 15;
        for (int base = wave * 4 + sub; base < th + 6; base += 8 * 4 * NW) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ly = base + u * 4 * NW;
                v[u] = 0;
                if (ly < th + 6 && dw < ndw) {
                    int gy = ty0 - 3 + ly;
                    if (!inside) gy = h_reflect101(gy, rows);
                    uint32_t w;
                    __builtin_memcpy(&w, im + (size_t)gy * cols + (tx0 - 3) + 4 * dw, 4);
                    v[u] = w;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ly = base + u * 4 * NW;
                if (ly < th + 6 && dw < ndw) reinterpret_cast<uint32_t*>(s_img + ly * HW_PITCH)[dw] = v[u];
            }
        }
        return;
    }
    // wave per row, lane per byte (tw + 6 <= 68: lanes 0..3 take a second byte); eight rows of loads in flight before
    // the first LDS store (one row per round trip made the tile load the longest stage of the kernel)
    int gx0 = tx0 - 3 + lane, gx1 = tx0 - 3 + 64 + lane;
    if (!inside) { gx0 = h_reflect101(gx0, cols); gx1 = h_reflect101(gx1, cols); }
    const bool second = lane + 64 < tw + 6;
    const bool first = lane < tw + 6;
    for (int base = wave; base < th + 6; base += 8 * NW) {
        unsigned char v0[8], v1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ly = base + u * NW;
            v0[u] = 0; v1[u] = 0;
            if (ly < th + 6) {
                int gy = ty0 - 3 + ly;
                if (!inside) gy = h_reflect101(gy, rows);
                const uint8_t* row = im + (size_t)gy * cols;
                if (first) v0[u] = row[gx0];
                if (second) v1[u] = row[gx1];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ly = base + u * NW;
            if (ly < th + 6) {
                if (first) s_img[ly * HW_PITCH + lane] = v0[u];
                if (second) s_img[ly * HW_PITCH + 64 + lane] = v1[u];
            }
        }
    }
}

// lane i <- lane i - 1 / lane i + 1 of the whole wave (gfx9 DPP wave_shr:1 / wave_shl:1); the end lanes get 0
__device__ __forceinline__ float wave_shr1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_shl1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

// Wave `band` of the workgroup walks output rows [y0, y1) of the tile (relative to ty0); sink(y, lx - 1, R, valid) is
// called by EVERY lane for every row, in row order; valid = the lane owns an output column (0 <= lx - 1 < tw).
//
// The walk keeps three sliding windows per lane: the last five rows of the two row passes (H, G) and the last three
// box row sums (ra, rb, rc).  Sliding them by moves cost 22 of the ~85 vector instructions of a row (round 4).  They
// are RINGS now: the steady-state rows run in blocks of 15 (= lcm(5, 3)) fully unrolled rows in which every window slot
// is a compile-time register, the new row overwrites the oldest slot, and after 15 rows every ring is back in its
// canonical order (slot 0 = oldest) -- the order the rolled code of the warm-up rows and of the remainder expects.  A
// KITTI bin is 75 rows: five blocks, no remainder.  The five pixel bytes of a row come as one dword + one byte and are
// converted by v_cvt_f32_ubyte0..3 straight from the dword (no shifts).
struct HarrisRow {   // one lane's state of the walk
    float H[5], G[5];                                  // rows q-4 .. q of the two row passes (canonical order between blocks)
    float ra[3], rb[3], rc[3];                         // box row sums of cov rows r-2 .. r
    uint32_t nw; uint32_t nb4;                         // LDS source: the NEXT row's five bytes (p0..p3 packed, p4);
                                                       // direct source: this lane's pixel of rows q and q + 1
};
// Where a row's pixels come from.  LDS source (s_img != null): the tile was loaded into LDS first, lane lx reads the five
// bytes of its cov column.  DIRECT source (round 5, tiles of up to 58 columns): no tile in LDS at all -- lane l owns PIXEL
// column tx0 - 3 + l, loads its one byte per row straight from the image (one 64-byte request per wave and row, two rows
// ahead) and gets its four neighbours by wave shifts of the converted value: no tile-load phase, no LDS per wave beyond the
// candidate list (8 instead of 5 waves per SIMD), one conversion per row instead of five.  Same arithmetic on the same
// values: the cov column of lane l is pixel column tx0 - 3 + l, i.e. logical lane lx = l - 2 of the LDS layout.
struct HarrisSrc {
    const unsigned char* s_img;   // LDS tile, or null
    __amdgpu_buffer_rsrc_t rsrc; int pcol;   // direct: the image as a raw buffer (rows * cols bytes), this lane's (reflected) pixel column
};

// HS / RS: ring slot the new row-pass row / the new row sums go to (the oldest of the window); in the rolled code both
// are the last slot after an explicit shift (HS = 4, RS = 2, shift = true)
#ifndef HW_BLOCK
#define HW_BLOCK 15                 // unrolled rows per block: 15 (both rings) or 5 (H / G rings, the row sums shift: 386 against 379 us)
#endif
// (Measured and dropped, round 5: a second instantiation for tiles that touch no image edge -- no ring column, no cov row
// to reflect, no row below the image: three uniform branches less per row -- 378 us either way.)
template <int HS, int RS, bool SHIFT, bool FULL, class Sink, bool SHIFT_R = SHIFT, bool DIRECT = false>
__device__ __forceinline__ bool harris_row(HarrisRow& w, const HarrisSrc& src, int q, int y0, int y1, int rows, int ty0,
                                           int lxc, int lx, int tw, int gx, int cols, bool ring_tile, int ring_src, double k,
                                           float t0, float t1, float t2, unsigned long long vmask, Sink& sink) {
    float p0, p1, p2, p3, p4;
    if (DIRECT) {
        p2 = (float)(w.nw & 255u);
        w.nw = w.nb4;
        {   // the pixel of row q + 2 is asked for before this row's arithmetic
            int gy = ty0 + min(q + 2, y1 + 2);         // at most three rows outside the image
            if (rows >= 4) { gy = abs(gy); gy = min(gy, 2 * (rows - 1) - gy); }   // uniform: one reflection is enough, three scalar instructions
            else { asm volatile("" ::: "memory"); gy = h_reflect101(gy, rows); }   // (kept a branch of its own: the empty asm)
            w.nb4 = __builtin_amdgcn_raw_buffer_load_b8(src.rsrc, src.pcol, gy * cols, 0);   // the row's offset rides in an SGPR: no address arithmetic
            __builtin_amdgcn_sched_barrier(0);
        }
        p1 = wave_shr1(p2); p0 = wave_shr1(p1);            // lane l <- lane l - 1: the pixel one / two columns to the left
        p3 = wave_shl1(p2); p4 = wave_shl1(p3);
    } else {
        // (float)((dword >> 8 n) & 255) is what the backend selects v_cvt_f32_ubyte<n> for
        p0 = (float)(w.nw & 255u); p1 = (float)((w.nw >> 8) & 255u); p2 = (float)((w.nw >> 16) & 255u);
        p3 = (float)(w.nw >> 24); p4 = (float)(w.nb4 & 255u);
        // the five bytes of the NEXT row are asked for before this row's arithmetic (the LDS round trip at the top of every
        // row was exposed: the walk is one dependent chain per row)
        const unsigned char* p = src.s_img + (min(q + 1, y1 + 2) + 3) * HW_PITCH + lxc;
        uint32_t v; __builtin_memcpy(&v, p, 4);
        w.nw = v; w.nb4 = p[4];
        __builtin_amdgcn_sched_barrier(0);
    }
    if (SHIFT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { w.H[i] = w.H[i + 1]; w.G[i] = w.G[i + 1]; }
    }
    // RowFilter: k0*S0, += k1*S1, ... left to right.  Derivative taps -1,-2,0,2,1: small integers, exact in any order
    w.H[HS] = (p4 - p0) + 2.f * (p3 - p1);
    float g = t0 * p0;
    g += t1 * p1; g += t2 * p2; g += t1 * p3; g += t0 * p4;
    w.G[HS] = g;
#define HROW(i) ((HS + 1 + (i)) % 5)                    /* slot of row q - 4 + i */
    const int r = q - 2;                               // cov row now complete (needs rows r-2 .. r+2 = q-4 .. q)
    if (!FULL && r < y0 - 1) return true;              // uniform: still warming up
    // SymmColumnFilter, symmetric: f0*S0 + delta, += f1*(S1 + S-1), += f2*(S2 + S-2)
    float dx = t2 * w.H[HROW(2)] + 0.f;
    dx += t1 * (w.H[HROW(3)] + w.H[HROW(1)]);
    dx += t0 * (w.H[HROW(4)] + w.H[HROW(0)]);
    // SymmColumnFilter, anti-symmetric (taps 0, 2, 1): delta, += 2*(S1 - S-1), += 1*(S2 - S-2)
    float dy = 0.f;
    dy += 2.f * (w.G[HROW(3)] - w.G[HROW(1)]);
    dy += 1.f * (w.G[HROW(4)] - w.G[HROW(0)]);
#undef HROW
    float ca = dx * dx, cb = dx * dy, cc = dy * dy;
    if (ring_tile) {                                   // uniform: this tile touches the left or right image edge
        // a lane whose cov column lies outside the image takes the cov of the reflected column (BORDER_REFLECT_101 of the
        // cov IMAGE, not of the source)
        const float ua = __shfl(ca, ring_src), ub = __shfl(cb, ring_src), uc = __shfl(cc, ring_src);   // ring_src: a PHYSICAL lane
        if (ring_src != (int)(threadIdx.x & 63)) { ca = ua; cb = ub; cc = uc; }
    }
    // box filter, row sums: (S[x-1] + S[x]) + S[x+1]; the neighbours by DPP wave shifts
    float sa, sb, sc;
    {   // six v_add_f32 with a DPP operand (left to the compiler: six DPP moves and three packed adds).  One asm block
        // behind an s_nop 1: a register written by the previous two vector instructions must not be a DPP source, and
        // the compiler's hazard recognizer does not look inside inline asm; inside the block every DPP source is older
        float ta, tb, tc;
        asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
            "v_add_f32_dpp %1, %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
            "v_add_f32_dpp %2, %8, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
            "v_add_f32_dpp %3, %6, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
            "v_add_f32_dpp %4, %7, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
            "v_add_f32_dpp %5, %8, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
            : "=&v"(ta), "=&v"(tb), "=&v"(tc), "=&v"(sa), "=&v"(sb), "=&v"(sc) : "v"(ca), "v"(cb), "v"(cc));
    }
    if (SHIFT_R) {
        w.ra[0] = w.ra[1]; w.ra[1] = w.ra[2];
        w.rb[0] = w.rb[1]; w.rb[1] = w.rb[2];
        w.rc[0] = w.rc[1]; w.rc[1] = w.rc[2];
    }
    w.ra[RS] = sa; w.rb[RS] = sb; w.rc[RS] = sc;
#define RROW(i) ((RS + 1 + (i)) % 3)                    /* slot of row sums r - 2 + i */
    const int y = r - 1;                               // output row now complete (needs row sums y-1 .. y+1 = r-2 .. r)
    if (!FULL && y < y0) return true;                  // uniform
    const int gy = ty0 + y;
    if (gy >= rows) return false;                      // uniform: rows below the image are nobody's
    const float a0 = w.ra[RROW(0)], a1 = w.ra[RROW(1)], a2 = w.ra[RROW(2)];
    const float b0 = w.rb[RROW(0)], b1 = w.rb[RROW(1)], b2 = w.rb[RROW(2)];
    const float c0 = w.rc[RROW(0)], c1 = w.rc[RROW(1)], c2 = w.rc[RROW(2)];
    float a, b, c;                                     // column sums: (rs[y-1] + rs[y]) + rs[y+1]
    // BORDER_REFLECT_101 of the cov image in y: row -1 is row 1, row `rows` is row rows - 2 (the image's first and last
    // row only; as real branches -- the empty asm keeps the compiler from turning them into selects on every row of
    // every tile.  The sums of rows outside the image are never stored: only this row's operands change)
    // (only the sums differ on those rows, computed in a branch of their own: as operand swaps in front of common adds
    // they cost the other rows three register moves each)
    if ((unsigned)(gy - 1) >= (unsigned)(rows - 2)) {  // uniform: gy == 0 || gy == rows - 1 (gy < rows here), one compare
        asm volatile("" ::: "memory");
        if (rows > 1) {
            // the first row's row -1 is row 1; the last row's row `rows` is row rows - 2 (never both: rows > 1)
            const float ta = gy == 0 ? a2 : a0, tb = gy == 0 ? b2 : b0, tc = gy == 0 ? c2 : c0;
            const float ua = gy == rows - 1 ? ta : a2, ub = gy == rows - 1 ? tb : b2, uc = gy == rows - 1 ? tc : c2;
            a = (ta + a1) + ua; b = (tb + b1) + ub; c = (tc + c1) + uc;
        } else { a = (a1 + a1) + a1; b = (b1 + b1) + b1; c = (c1 + c1) + c1; }
    } else { a = (a0 + a1) + a2; b = (b0 + b1) + b2; c = (c0 + c1) + c2; }
#undef RROW
    const float m1 = a * c, m2 = b * b;
    const float m3 = m1 - m2;
    const float tr = a + c;
    const float R = (float)((double)m3 - k * (double)tr * (double)tr);
    sink(y, lx - 1, R, lx >= 1 && lx <= tw && gx < cols, vmask);   // every lane calls (wave-wide operations inside are fine)
    return true;
}

template <int U, bool DIRECT, class Sink>
__device__ __forceinline__ bool harris_block15(HarrisRow& w, const HarrisSrc& s_img, int q, int y0, int y1, int rows, int ty0,
                                               int lxc, int lx, int tw, int gx, int cols, bool ring_tile, int ring_src, double k,
                                               float t0, float t1, float t2, unsigned long long vmask, Sink& sink) {
    if constexpr (U < HW_BLOCK) {
        if (!harris_row<U % 5, HW_BLOCK == 15 ? U % 3 : 2, false, true, Sink, HW_BLOCK != 15, DIRECT>(w, s_img, q + U, y0, y1, rows, ty0, lxc, lx, tw, gx, cols, ring_tile, ring_src, k, t0, t1, t2, vmask, sink)) return false;
        return harris_block15<U + 1, DIRECT>(w, s_img, q, y0, y1, rows, ty0, lxc, lx, tw, gx, cols, ring_tile, ring_src, k, t0, t1, t2, vmask, sink);
    } else {
        return true;
    }
}

template <bool DIRECT, class Sink>
__device__ __forceinline__ void harris_walk_band_impl(const unsigned char* s_img, const uint8_t* im, int rows, int cols, int tx0, int ty0,
                                                      int tw, int y0, int y1, double k, Sink sink) {
    const int lane = threadIdx.x & 63;
    const int lx = DIRECT ? lane - 2 : lane;           // logical lane: cov column tx0 - 1 + lx
    const int gx = tx0 - 1 + lx;                       // this lane's cov column
    const bool col_used = lx >= 0 && lx < tw + 2;
    const int lxc = col_used ? lx : 0;                 // idle lanes read a valid address
    const float scale = (float)(1.0 / (16.0 * 3.0 * 255.0));
    const float t0 = 1.f * scale, t1 = 4.f * scale, t2 = 6.f * scale;   // tap_i = fl32(s_i * fl32(scale)), symmetric
    // gx = -1 <- gx = 1, gx = cols <- gx = cols - 2 (lanes two up / two down; one for a one-column image)
    const bool ring_tile = tx0 == 0 || tx0 + tw == cols;
    const int ring_src = (gx == -1 || gx == cols) ? lane + (h_reflect101(gx, cols) - gx) : lane;   // a physical lane
    const unsigned long long vmask = __builtin_amdgcn_ballot_w64(lx >= 1 && lx <= tw && gx < cols);   // the lanes that own an output column
    HarrisSrc src;
    src.s_img = s_img;
    if (DIRECT) src.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)im, 0, rows * cols, 0x00020000);
    src.pcol = DIRECT ? h_reflect101(min(tx0 - 3 + lane, tx0 + tw + 2), cols) : 0;   // lanes past the tile's halo repeat its last column
    HarrisRow w;
#pragma unroll
    for (int i = 0; i < 5; ++i) { w.H[i] = 0.f; w.G[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { w.ra[i] = 0.f; w.rb[i] = 0.f; w.rc[i] = 0.f; }
    if (DIRECT) {
        w.nw = __builtin_amdgcn_raw_buffer_load_b8(src.rsrc, src.pcol, h_reflect101(ty0 + y0 - 3, rows) * cols, 0);
        w.nb4 = __builtin_amdgcn_raw_buffer_load_b8(src.rsrc, src.pcol, h_reflect101(ty0 + min(y0 - 2, y1 + 2), rows) * cols, 0);
    } else {
        const unsigned char* p = s_img + (y0 - 3 + 3) * HW_PITCH + lxc;
        uint32_t v; __builtin_memcpy(&v, p, 4);
        w.nw = v; w.nb4 = p[4];
    }
    int q = y0 - 3;                                    // row-pass row q (relative to ty0) = LDS row q + 3
    // warm-up: four rows that only feed the row passes, two that also produce row sums (rolled code, shifting windows)
    for (; q < y0 + 3 && q < y1 + 3; ++q)
        if (!harris_row<4, 2, true, false, Sink, true, DIRECT>(w, src, q, y0, y1, rows, ty0, lxc, lx, tw, gx, cols, ring_tile, ring_src, k, t0, t1, t2, vmask, sink)) return;
    // steady state: blocks of 15 unrolled rows on rings
    for (; q + HW_BLOCK <= y1 + 3; q += HW_BLOCK)
        if (!harris_block15<0, DIRECT>(w, src, q, y0, y1, rows, ty0, lxc, lx, tw, gx, cols, ring_tile, ring_src, k, t0, t1, t2, vmask, sink)) return;
    // remainder (tiles whose height is not a multiple of 15)
    for (; q < y1 + 3; ++q)
        if (!harris_row<4, 2, true, true, Sink, true, DIRECT>(w, src, q, y0, y1, rows, ty0, lxc, lx, tw, gx, cols, ring_tile, ring_src, k, t0, t1, t2, vmask, sink)) return;
}
// ---- cv::cornerHarris as an image (plain API, and bins too large for the fused detector) -----------------------
// One wave per (58-column strip, band of `band` rows), walking the image itself (HarrisSrc direct): no LDS, no barrier.
// The band height is the launcher's choice: tall bands pay the walk's six warm-up rows less often, short ones put more
// waves on a single image.
#define HR_TW HW_DIRECT_MAXW
__global__ __launch_bounds__(HW_THREADS) void harris_response_kernel(const uint8_t* __restrict__ images, int rows,
                                                                     int cols, double k, float* __restrict__ resp, int band) {
    const int img = blockIdx.z;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave uniform: scalar loop control
    const int tx0 = blockIdx.x * HR_TW, ty0 = (blockIdx.y * (HW_THREADS / 64) + wave) * band;
    if (ty0 >= rows) return;
    const int tw = min(HR_TW, cols - tx0), th = min(band, rows - ty0);
    const uint8_t* im = images + (size_t)img * rows * cols;
    float* out = resp + (size_t)img * rows * cols;
    harris_walk_band_impl<true>(nullptr, im, rows, cols, tx0, ty0, tw, 0, th, k,
                                [&](int y, int x, float R, bool valid, unsigned long long) { if (valid) out[(size_t)(ty0 + y) * cols + tx0 + x] = R; });
}

struct BinArgs {
    const float* resp;      // [n_img][rows][cols]
    int rows, cols, n_img;
    int nbinx, nbiny, stridex, stridey, per;
    float2* tmp_kp;         // [n_img][nbins][per]
    float* tmp_resp;        // [n_img][nbins][per]
    int* cnt;               // [n_img][nbins]
};

// One wave per (image, bin).  key = (|response| bits, ~push position): larger is
// better and ties go to the earlier push.  Fast path: ONE pass over the bin in
// which every lane keeps its three largest keys, then `per` rounds of wave-max
// over the lanes' heads.  That is exact unless some lane's third-best key is
// still above the last pick (a fourth could hide behind it); such bins (a few
// percent) are redone by the exact multi-pass loop.
// max over the wave of a u32, to every lane as a scalar: DPP row shifts inside the 16-lane rows, two row broadcasts, the
// total lands in lane 63 (gfx9 reduction idiom; six v_max_u32_dpp + a readlane instead of six ds_bpermute round trips)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#define HMAX_DPP(ctrl, rmask) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xf, false))
    HMAX_DPP(0x111, 0xf);   // row_shr:1
    HMAX_DPP(0x112, 0xf);   // row_shr:2
    HMAX_DPP(0x114, 0xf);   // row_shr:4
    HMAX_DPP(0x118, 0xf);   // row_shr:8   -> lane 15 of every row = the row's max
    HMAX_DPP(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    HMAX_DPP(0x143, 0xc);   // row_bcast:31 into rows 2 and 3 -> lane 63 = the wave's max
#undef HMAX_DPP
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// max over the wave of a u64 key (|response| bits : ~position): the high words first, then the low words of the lanes
// that hold the maximal high word (ties of |response| are rare, the second reduction is still cheap)
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    const uint32_t mh = wave_max_u32(hi);
    const uint32_t ml = wave_max_u32(hi == mh ? lo : 0u);
    return ((unsigned long long)mh << 32) | ml;
}

#define HD_MAXPER 32  // corners per bin the fused detector keeps (the reference has 10)
#define HB_LIST 256   // keys >= the last optimistic pick collected by the one-walk exact path (per wave)

// The `per` largest keys, descending, of the bin pixels idx = first + lane + step * i < P (idx in MEMORY order: x
// fastest), by ONE wave.  key = (|response| bits << 32) | ~push position, push position = xo * stridey + yo (the
// reference pushes x outer, y inner, :953-955): larger is better, ties of |response| go to the earlier push, keys are
// unique.  fetch(xo, yo) = |response| (0 where the bin reaches past the image; zeros are skipped like the reference's
// isEqual(response, 0) test).  emit(n, key) is called with wave-uniform arguments for the n-th best key.  Returns the
// count.  `list`: LIST keys of LDS scratch of this wave.
// Fast path: ONE pass in which every lane keeps its three largest keys, then `per` rounds of wave-max over the lanes'
// heads.  That is exact unless some lane's third-best key is still above the last pick (a fourth could hide behind
// it); such subsets (a few percent) collect all keys >= the last optimistic pick in one more walk and select among
// them, with the pick-by-pick multi-pass loop as the last resort.
__device__ __forceinline__ unsigned long long harris_key(float v, int pos) {
    if (fabsf(v - 0.f) <= 1e-6f * fabsf(v)) return 0ull;   // isEqual(response, .0f), src/misc.cpp:10-14
    return ((unsigned long long)__float_as_uint(v) << 32) | (uint32_t)(0xffffffffu - (uint32_t)pos);
}

template <int LIST, class Fetch, class Emit>
__device__ __forceinline__ int harris_top_keys(int stridex, int stridey, int P, int per, int first, int step,
                                               Fetch fetch, Emit emit, unsigned long long* list) {
    const int lane = threadIdx.x & 63;
    const int sq = step / stridex, sr = step % stridex;    // idx += step  <=>  (xo, yo) += (sr, sq) with one carry
    const int f0 = first + lane;
    auto walk = [&](auto visit) {                          // visit(key) for every pixel of the subset, four loads in flight
        int yo = f0 / stridex, xo = f0 % stridex;
        for (int idx = f0; idx < P; idx += 4 * step) {
            float v[4];
            int pos[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                pos[u] = xo * stridey + yo;
                v[u] = idx + step * u < P ? fetch(xo, yo) : 0.f;
                xo += sr; yo += sq;
                if (xo >= stridex) { xo -= stridex; ++yo; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) visit(harris_key(v[u], pos[u]));
        }
    };
    // ---- pass 1: per-lane top 3
    unsigned long long k0 = 0, k1 = 0, k2 = 0;             // k0 >= k1 >= k2
    walk([&](unsigned long long key) {
        if (key > k2) {
            if (key > k1) { k2 = k1; if (key > k0) { k1 = k0; k0 = key; } else k1 = key; }
            else k2 = key;
        }
    });
    // ---- merge: `per` rounds over the lanes' heads
    const unsigned long long third = k2;                   // what this lane might be hiding behind
    unsigned long long last = 0;
    unsigned long long picked = 0;                         // lane n keeps the n-th pick until the verdict (per <= 64)
    int n = 0;
    for (int round = 0; round < per; ++round) {
        const unsigned long long best = wave_max_u64(k0);
        if (best == 0) break;
        if (k0 == best) { k0 = k1; k1 = k2; k2 = 0; }      // keys are unique: exactly one lane pops
        last = best;
        if (per <= 64) { if (lane == n) picked = best; } else emit(n, best);
        ++n;
    }
    // exact iff no lane used up all three of its keys while more picks could lie below them
    const bool suspicious = (n == per) ? (third > last) : (third != 0 && k0 == 0 && n < per);
    if (!__any(suspicious)) {
        if (per <= 64)
            for (int i = 0; i < n; ++i) emit(i, __shfl(picked, i));
        return n;
    }
    // The optimistic picks are `per` real keys of the subset, so every key of the true top `per` is >= the last
    // optimistic pick: ONE more walk collects all keys >= last (a handful more than `per`) into LDS and the exact top
    // `per` is selected among them.  Subsets where that does not apply (fewer than `per` picks, or an implausibly long
    // list) take the multi-pass path below.
    if (n == per) {
        int cnt = 0;
        walk([&](unsigned long long key) {
            const bool take = key >= last;                 // last != 0 here
            const unsigned long long m = __ballot(take);
            const int at = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (take && at < LIST) list[at] = key;
            cnt += __popcll(m);
        });
        // the walk's trip count falls with the lane (P need not be a multiple of its stride): the lanes that left early
        // missed the last appends; lane 0 runs every iteration and has the whole count
        cnt = __builtin_amdgcn_readfirstlane(cnt);
        __builtin_amdgcn_wave_barrier();
        if (cnt <= LIST) {
            unsigned long long mine[LIST / 64];
#pragma unroll
            for (int u = 0; u < LIST / 64; ++u) mine[u] = (lane + 64 * u < cnt) ? list[lane + 64 * u] : 0ull;
            n = 0;
            for (int round = 0; round < per; ++round) {
                unsigned long long loc = mine[0];
#pragma unroll
                for (int u = 1; u < LIST / 64; ++u) loc = mine[u] > loc ? mine[u] : loc;
                const unsigned long long best = wave_max_u64(loc);
                if (best == 0) break;
#pragma unroll
                for (int u = 0; u < LIST / 64; ++u) if (mine[u] == best) mine[u] = 0;   // keys are unique
                emit(n, best);
                ++n;
            }
            return n;
        }
    }
    // ---- exact path: `per` rounds of "largest key below the previous pick"
    unsigned long long prev = ~0ull;
    n = 0;
    for (int round = 0; round < per; ++round) {
        unsigned long long best = 0;
        walk([&](unsigned long long key) { if (key < prev && key > best) best = key; });
        best = wave_max_u64(best);
        if (best == 0) break;
        prev = best;
        emit(n, best);
        ++n;
    }
    return n;
}

// emit of the final corners of (img, bin): key -> keypoint + |response| in the bin's slots
struct BinEmit {
    const BinArgs& a; size_t obase; int x0, y0;
    __device__ __forceinline__ void operator()(int n, unsigned long long best) const {
        if ((threadIdx.x & 63) == 0) {
            const int pos = (int)(0xffffffffu - (uint32_t)best);
            a.tmp_kp[obase + n] = make_float2((float)(x0 + pos / a.stridey), (float)(y0 + pos % a.stridey));
            a.tmp_resp[obase + n] = __uint_as_float((uint32_t)(best >> 32));
        }
    }
};

// Bins from a response image in memory (bins too large for the fused detector below): one wave per (image, bin).
__global__ __launch_bounds__(256) void harris_bins_kernel(BinArgs a) {
    __shared__ unsigned long long s_list[4][HB_LIST];
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nbins = a.nbinx * a.nbiny;
    if (wave >= (long long)a.n_img * nbins) return;
    const int img = (int)(wave / nbins), bin = (int)(wave % nbins);
    const int x0 = (bin / a.nbiny) * a.stridex, y0 = (bin % a.nbiny) * a.stridey;
    const float* r = a.resp + (size_t)img * a.rows * a.cols;
    const BinEmit emit{a, ((size_t)img * nbins + bin) * a.per, x0, y0};
    const int n = harris_top_keys<HB_LIST>(a.stridex, a.stridey, a.stridex * a.stridey, a.per, 0, 64, [&](int xo, int yo) {
        const int x = x0 + xo, y = y0 + yo;
        return (x < a.cols && y < a.rows) ? fabsf(r[(size_t)y * a.cols + x]) : 0.f;
    }, emit, s_list[threadIdx.x >> 6]);
    if ((threadIdx.x & 63) == 0) a.cnt[(size_t)img * nbins + bin] = n;
}

// The fused detector: one WAVE per (image, bin), four bins per workgroup, no workgroup barrier.  The bin IS the tile: the
// wave loads its pixels (+ halo) into its slice of LDS, walks all of its rows (one warm-up of six rows per bin; with
// the rows split over four waves the warm-up was a third of the work) and selects the corners ON THE WAY: a key that
// beats the running threshold tau (the `per`-th best key so far, 0 until then) is appended to the wave's candidate list
// in LDS; when the list could overflow on the next row its `per` best are kept and tau rises.  Every key ever dropped
// was <= tau at that time <= the final `per`-th best, so the true top `per` are in the list at the end: exact, no second
// pass, and neither the response image nor the bin's responses ever exist in memory.  Per image the HBM traffic is the
// uint8 pixels in (+ halo re-reads from L2) and the corners out.
#define HD_KEEP 48    // keys a mid-walk harris_keep_best may leave in the list (>= HD_MAXPER; the list refills from there)
#define HD_CAND 256   // candidate keys per wave; a row appends at most HW_MAXW
// The `per` largest of list[0..n) stay in list[0..kept) (descending if `sorted`), tau = the smallest kept (0 if fewer than `per`).
// Called when the wave's candidate list is nearly full (a few times per bin) and once at the end of the bin.
//
// Round 4 picked the maximum `per` times (wave-wide 64-bit max, then removing it: ~55 vector instructions per pick, ~600
// per call: a fifth of the kernel's instructions).  Now the per-th largest HIGH word (|response| bits) is found bit by
// bit from the top -- 31 steps of four compares whose ballots are counted on the scalar unit -- and everything at or
// above it is compacted: ~190 vector instructions.  Keys that share the threshold's high word are decided by their low
// words (push order): when more of them exist than places are left, the old pick-by-pick loop runs (rare: equal
// |response| bits).  The walk only tests the high word of tau, so the set kept here is the exact top `per`.
struct HarrisKept { unsigned long long tau; int n; };
template <int CAND>
__device__ __forceinline__ HarrisKept harris_keep_best_picks(unsigned long long (&mine)[CAND / 64], unsigned long long* list, int per) {
    const int lane = threadIdx.x & 63;
    int kept = 0;
    unsigned long long last = 0;
    for (int round = 0; round < per; ++round) {
        unsigned long long loc = mine[0];
#pragma unroll
        for (int u = 1; u < CAND / 64; ++u) loc = mine[u] > loc ? mine[u] : loc;
        const unsigned long long best = wave_max_u64(loc);
        if (best == 0) break;
#pragma unroll
        for (int u = 0; u < CAND / 64; ++u) if (mine[u] == best) mine[u] = 0;   // keys are unique
        if (lane == 0) list[kept] = best;
        last = best;
        ++kept;
    }
    __builtin_amdgcn_wave_barrier();
    HarrisKept out;
    out.tau = kept == per ? last : 0ull;
    out.n = kept;
    return out;
}

template <int CAND = HD_CAND>
__device__ __forceinline__ HarrisKept harris_keep_best(unsigned long long* list, int n, int per, bool sorted) {
    const int lane = threadIdx.x & 63;
    unsigned long long mine[CAND / 64];
#pragma unroll
    for (int u = 0; u < CAND / 64; ++u) mine[u] = (lane + 64 * u < n) ? list[lane + 64 * u] : 0ull;
    __builtin_amdgcn_wave_barrier();
    HarrisKept out;
    if (n <= per && !sorted) {                              // uniform: nothing to drop, nothing to order
        out.n = n;
        out.tau = 0ull;
        if (n == per) {                                     // the smallest key is the threshold
            unsigned long long mn = mine[0] ? mine[0] : ~0ull;
            mn = ~wave_max_u64(~mn);
            out.tau = mn;
        }
        return out;
    }
    uint32_t hi[CAND / 64];
#pragma unroll
    for (int u = 0; u < CAND / 64; ++u) hi[u] = (uint32_t)(mine[u] >> 32);
    // T = a high word with #(hi >= T) >= per, built from the top bit down (the greatest such T = the per-th largest high
    // word when every bit is decided; 0 when fewer than `per` keys exist).  Every prefix is such a T already, so the
    // descent stops as soon as few enough keys are at or above it: exactly `per` for the final, ordered list (they ARE the
    // best `per` then); HD_KEEP for a call in the middle of the walk, which only has to make room and hand the walk a
    // threshold that no key of the final answer falls below -- a third of the steps on the bench's images
    uint32_t T = 0;
    int kept = n;                                           // #(hi >= T): every key's high word is >= 1
    const int stop = sorted ? per : max(per, HD_KEEP);
    int bit = 30;                                           // |response| bits: the sign bit is clear
    for (; bit >= 0 && kept > stop; --bit) {
        const uint32_t cand = T | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < CAND / 64; ++u) cnt += __popcll(__ballot(hi[u] >= cand));
        if (cnt >= per) { T = cand; kept = cnt; }           // uniform
    }
    // every bit decided and still more than `per` at or above T: keys tie T's high word, the low words decide among them
    if (bit < 0 && T != 0 && kept > per) return harris_keep_best_picks<CAND>(mine, list, per);   // uniform
    // compaction of everything at or above T (T == 0: fewer than `per` keys, all stay)
    int base = 0;
#pragma unroll
    for (int u = 0; u < CAND / 64; ++u) {
        const bool take = mine[u] != 0ull && hi[u] >= T;
        const unsigned long long m = __ballot(take);
        const int at = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (take) list[at] = mine[u];
        base += __popcll(m);
    }
    __builtin_amdgcn_wave_barrier();
    out.n = base;
    out.tau = base >= per ? ((unsigned long long)T << 32) : 0ull;   // the walk tests the high word only (T == 0: no threshold yet)
    if (sorted && base > 1) {                               // descending: every key to the place its rank says (base <= per <= HD_MAXPER <= 64)
        const unsigned long long key = lane < base ? list[lane] : 0ull;
        int rank = 0;
        for (int j = 0; j < base; ++j) {
            const unsigned long long kj = __shfl(key, j);
            rank += kj > key ? 1 : 0;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < base) list[rank] = key;
        __builtin_amdgcn_wave_barrier();
    }
    return out;
}

// DIRECT: bins of up to HW_DIRECT_MAXW columns walk the image itself (HarrisSrc), the LDS holds candidate lists only
template <bool DIRECT>
__global__ __launch_bounds__(HW_THREADS) void harris_detect_kernel(BinArgs a, const uint8_t* __restrict__ images, double k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char h_smem[];
    const int nbins = a.nbinx * a.nbiny;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long long unit = (long long)blockIdx.x * (HW_THREADS / 64) + wave;
    if (unit >= (long long)a.n_img * nbins) return;
    const int img = (int)(unit / nbins), bin = (int)(unit % nbins);
    const int tx0 = (bin / a.nbiny) * a.stridex, ty0 = (bin % a.nbiny) * a.stridey;
    const int tw = a.stridex, th = a.stridey;              // bins never reach past the image: stride * nbin <= size
    const int per = a.per;
    const size_t tile_bytes = DIRECT ? 0 : ((size_t)(th + 6) * HW_PITCH + 15) & ~(size_t)15;
    unsigned char* s_img = h_smem + (size_t)wave * (tile_bytes + HD_CAND * sizeof(unsigned long long));
    unsigned long long* list = reinterpret_cast<unsigned long long*>(s_img + tile_bytes);
    const uint8_t* im = images + (size_t)img * a.rows * a.cols;
    if (!DIRECT) {
        harris_load_tile<1>(im, a.rows, a.cols, tx0, ty0, tw, th, s_img);
        __builtin_amdgcn_wave_barrier();
    }
    unsigned long long tau = 0;
    uint32_t thi = 1;                                      // max(tau's high word, 1): the row test
    int n = 0;
    harris_walk_band_impl<DIRECT>(s_img, im, a.rows, a.cols, tx0, ty0, tw, 0, th, k, [&](int y, int x, float R, bool, unsigned long long vmask) {
        // a candidate is everything above tau; the test is on the key's HIGH word (|response| bits) alone: the few pixels
        // that tie tau's high word come along and lose in harris_keep_best, and the 64-bit key (push position of the
        // reference's scan: x outer, y inner, :953-955) is only built for the lanes that store one.
        // isEqual(response, .0f) (src/misc.cpp:10-14: |v - 0| <= 1e-6 |v|) holds for v == 0 alone -- denormals are kept in
        // this build, 1e-6 |v| < |v| for every other finite v, a NaN compares false there and is a candidate here -- so
        // "not equal to zero" is hi >= 1 and rides in the same compare.  The row's whole test is that one compare, its
        // mask and the mask of the lanes that own a column (scalar); the lane's own bit comes back without a vector
        // instruction (inverse ballot)
        const uint32_t hi = __float_as_uint(R) & 0x7fffffffu;
        const unsigned long long m = vmask & __builtin_amdgcn_ballot_w64(hi >= thi);
        if (m) {                                           // uniform
            const bool take = __builtin_amdgcn_inverse_ballot_w64(m);
            const int at = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (take) list[at] = ((unsigned long long)hi << 32) | (uint32_t)(0xffffffffu - (uint32_t)(x * th + y));
            n += __popcll(m);
            if (n > HD_CAND - 64) {                        // uniform: the next row might not fit
                __builtin_amdgcn_wave_barrier();
                const HarrisKept kb = harris_keep_best(list, n, per, false);
                n = kb.n; tau = kb.tau;
                thi = max((uint32_t)(tau >> 32), 1u);
            }
        }
    });
    __builtin_amdgcn_wave_barrier();
    { const HarrisKept kb = harris_keep_best(list, n, per, true); n = kb.n; tau = kb.tau; }
    const BinEmit emit{a, ((size_t)img * nbins + bin) * a.per, tx0, ty0};
    for (int i = 0; i < n; ++i) emit(i, list[i]);
    if (lane == 0) a.cnt[(size_t)img * nbins + bin] = n;
}


#ifdef VISO_DEBUG_VARIANTS
// ---- strips that ignore the bin boundaries (round 6): MEASURED AND LOST, built in -DVISO_DEBUG_VARIANTS libraries only -------
// (0.268 ms against 0.241 ms per 258 images alone, alternating on one box, tools/experiments/harris_strips_ab.sh: one wave
// in twelve less, but every PART of a bin prunes its own candidate list -- 2.1 parts per wave, each with its own mid-walk and
// final harris_keep_best -- and a narrow part takes ~19 rows instead of 3 to fill its list and get a threshold, during which
// every row takes the append path.  HISTORY.md, round 6.)
// A bin of the reference's geometry is 51 columns wide (1241 / 24): the walk above, a wave per bin with lane = pixel
// column, leaves 13 of 64 lanes without an output column.  Here a wave takes a STRIP of HW_DIRECT_MAXW = 58 output columns
// of one bin ROW instead -- 22 strips cover what 24 bins cover, one wave in twelve less for the same pixels -- and keeps
// one candidate list per bin COLUMN its lanes fall into (a lane's bin column never changes during the walk: at most
// HS_NG = 3 groups of lanes, each with its own list, count and threshold; the row's test stays ONE compare against a
// per-lane threshold and one ballot, the appends only run for groups that have a candidate).  At the end of the walk
// every group leaves the best `per` keys of its part of the bin, in order; a bin's corners are the best `per` of the
// (at most two) parts that cover it: the true top `per` of a union is contained in the union of the parts' top `per`, and
// harris_merge_kernel selects it.  Same arithmetic per pixel (the walk is the same function), same keys, same order:
// bit-identical to the wave-per-bin kernel (tests/test_gpu_harris.py runs both).
#define HS_NG 3        // bin columns a strip can touch: needs stridex >= (HW_DIRECT_MAXW - 1) / 2 = 29 (launcher)
#define HS_CAND 192    // candidate keys per group: 3 x 1.5 KB per wave keeps 8 waves per SIMD inside the LDS
// the mid-walk pruning as a real call: inlined into 15 unrolled rows x 3 groups it made the kernel 80 KB of code (the
// wave-per-bin kernel: 40 KB; a call there measured the same time as the inlined body, HISTORY.md round 5)
__device__ __noinline__ HarrisKept harris_keep_best_strip(unsigned long long* list, int n, int per) {
    return harris_keep_best<HS_CAND>(list, n, per, false);
}
struct StripArgs {
    int nstrips;                 // strips per bin row: ceil(nbinx * stridex / HW_DIRECT_MAXW)
    unsigned long long* part;    // [n_img][nbins][2][per] keys of the parts, descending
    int* part_cnt;               // [n_img][nbins][2]
};
__global__ __launch_bounds__(HW_THREADS) void harris_strip_kernel(BinArgs a, StripArgs sa, const uint8_t* __restrict__ images, double k) {
    __shared__ unsigned long long s_list[HW_THREADS / 64][HS_NG][HS_CAND];
    const int nbins = a.nbinx * a.nbiny;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long long unit = (long long)blockIdx.x * (HW_THREADS / 64) + wave;
    const int per_img = sa.nstrips * a.nbiny;
    if (unit >= (long long)a.n_img * per_img) return;
    const int img = (int)(unit / per_img), rem = (int)(unit % per_img);
    const int by = rem / sa.nstrips, strip = rem % sa.nstrips;
    const int xend = a.nbinx * a.stridex;                  // columns past the last bin are nobody's (:934-936)
    const int tx0 = strip * HW_DIRECT_MAXW, ty0 = by * a.stridey;
    const int tw = min(HW_DIRECT_MAXW, xend - tx0), th = a.stridey;
    const int per = a.per;
    const uint8_t* im = images + (size_t)img * a.rows * a.cols;
    // this lane's output column (DIRECT layout: lane l <-> pixel column tx0 - 3 + l, output column x = l - 3), its bin
    // column, its group and its column inside the bin
    const int x = lane - 3;
    const bool owns = x >= 0 && x < tw;
    const int gxo = tx0 + (owns ? x : 0);
    const int bx0 = tx0 / a.stridex;
    const int bx = gxo / a.stridex;
    const int grp = bx - bx0;                              // 0 .. HS_NG - 1
    const int xin = gxo - bx * a.stridex;
    unsigned long long gmask[HS_NG];
    int n[HS_NG];
#pragma unroll
    for (int g = 0; g < HS_NG; ++g) { gmask[g] = __builtin_amdgcn_ballot_w64(owns && grp == g); n[g] = 0; }
    uint32_t thi = 1;                                      // per LANE: max(its group's tau high word, 1)
    const uint32_t posx = (uint32_t)(xin * th);
    harris_walk_band_impl<true>(nullptr, im, a.rows, a.cols, tx0, ty0, tw, 0, th, k, [&](int y, int, float R, bool, unsigned long long vmask) {
        const uint32_t hi = __float_as_uint(R) & 0x7fffffffu;
        const unsigned long long m = vmask & __builtin_amdgcn_ballot_w64(hi >= thi);
        if (m) {                                           // uniform
            const unsigned long long key = ((unsigned long long)hi << 32) | (uint32_t)(0xffffffffu - (posx + (uint32_t)y));
#pragma unroll
            for (int g = 0; g < HS_NG; ++g) {
                const unsigned long long mg = m & gmask[g];
                if (mg) {                                  // uniform
                    unsigned long long* list = s_list[wave][g];
                    const bool take = __builtin_amdgcn_inverse_ballot_w64(mg);
                    const int at = n[g] + __builtin_amdgcn_mbcnt_hi((uint32_t)(mg >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mg, 0u));
                    if (take) list[at] = key;
                    n[g] += __popcll(mg);
                    if (n[g] > HS_CAND - HW_DIRECT_MAXW) { // uniform: the next row might not fit
                        __builtin_amdgcn_wave_barrier();
                        const HarrisKept kb = harris_keep_best_strip(list, n[g], per);
                        n[g] = kb.n;
                        const uint32_t t = max((uint32_t)(kb.tau >> 32), 1u);
                        if (__builtin_amdgcn_inverse_ballot_w64(gmask[g])) thi = t;
                    }
                }
            }
        }
    });
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g = 0; g < HS_NG; ++g) {
        if (!gmask[g]) continue;                           // uniform: the strip does not reach that bin column
        unsigned long long* list = s_list[wave][g];
        const HarrisKept kb = harris_keep_best<HS_CAND>(list, n[g], per, true);
        const int bxg = bx0 + g;
        const int slot = strip - (bxg * a.stridex) / HW_DIRECT_MAXW;   // 0: the first strip that reaches the bin, 1: the second
        const size_t o = (((size_t)img * nbins + (size_t)bxg * a.nbiny + by) * 2 + (size_t)slot);
        if (lane < kb.n) sa.part[o * per + lane] = list[lane];         // per <= HD_MAXPER <= 64
        if (lane == 0) sa.part_cnt[o] = kb.n;
        __builtin_amdgcn_wave_barrier();
    }
}

// A bin's corners from its parts: thread per (image, bin), merge of at most two descending key lists.
__global__ __launch_bounds__(256) void harris_merge_kernel(BinArgs a, StripArgs sa) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int nbins = a.nbinx * a.nbiny;
    if (t >= (long long)a.n_img * nbins) return;
    const int img = (int)(t / nbins), bin = (int)(t % nbins);
    const int bxg = bin / a.nbiny, by = bin % a.nbiny;
    const int s0 = (bxg * a.stridex) / HW_DIRECT_MAXW, s1 = (bxg * a.stridex + a.stridex - 1) / HW_DIRECT_MAXW;
    const size_t o = ((size_t)img * nbins + bin) * 2;
    const unsigned long long* A = sa.part + o * a.per;
    const unsigned long long* B = A + a.per;
    const int na = sa.part_cnt[o], nb = s1 > s0 ? sa.part_cnt[o + 1] : 0;
    const int x0 = bxg * a.stridex, y0 = by * a.stridey;
    const size_t obase = ((size_t)img * nbins + bin) * a.per;
    int i = 0, j = 0, n = 0;
    while (n < a.per && (i < na || j < nb)) {
        const unsigned long long ka = i < na ? A[i] : 0ull, kb = j < nb ? B[j] : 0ull;   // keys are unique and non-zero
        const unsigned long long best = ka > kb ? ka : kb;
        if (ka > kb) ++i; else ++j;
        const int pos = (int)(0xffffffffu - (uint32_t)best);
        a.tmp_kp[obase + n] = make_float2((float)(x0 + pos / a.stridey), (float)(y0 + pos % a.stridey));
        a.tmp_resp[obase + n] = __uint_as_float((uint32_t)(best >> 32));
        ++n;
    }
    a.cnt[(size_t)img * nbins + bin] = n;
}

#endif   // VISO_DEBUG_VARIANTS

// One workgroup per image: concatenate the bins' corners in bin order.
__global__ __launch_bounds__(256) void harris_compact_kernel(BinArgs a, float2* kp_out, float* resp_out, int* n_out,
                                                             int cap, size_t kp_stride) {
    extern __shared__ int s_off[];
    const int img = blockIdx.x;
    const int nbins = a.nbinx * a.nbiny;
    if (threadIdx.x == 0) {
        int o = 0;
        for (int b = 0; b < nbins; ++b) { s_off[b] = o; o += a.cnt[(size_t)img * nbins + b]; }
        s_off[nbins] = o;
        n_out[img] = o < cap ? o : cap;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nbins * a.per; e += 256) {
        const int b = e / a.per, r = e % a.per;
        if (r >= a.cnt[(size_t)img * nbins + b]) continue;
        const int o = s_off[b] + r;
        if (o >= cap) continue;
        const size_t src = ((size_t)img * nbins + b) * a.per + r;
        kp_out[(size_t)img * kp_stride + o] = a.tmp_kp[src];
        if (resp_out) resp_out[(size_t)img * kp_stride + o] = a.tmp_resp[src];
    }
}

int launch_harris_response(hipStream_t s, const uint8_t* images, int n_img, int rows, int cols, double k, float* resp) {
    if (n_img <= 0) return VISO_OK;
    // bands of 76 rows once the strips alone fill the GPU's wave slots a few times over, of 19 rows for a few images
    const int strips = (cols + HR_TW - 1) / HR_TW;
    int band = (long long)n_img * strips * ((rows + 75) / 76) >= 16384 ? 76 : 19;
    if (const char* e = getenv("VISO_HARRIS_BAND")) { const int v = atoi(e); if (v >= 1 && v <= 4096) band = v; }   // tuning / tests: results do not depend on it
    dim3 grid(strips, (rows + band * (HW_THREADS / 64) - 1) / (band * (HW_THREADS / 64)), n_img);
    hipLaunchKernelGGL(harris_response_kernel, grid, dim3(HW_THREADS), 0, s, images, rows, cols, k, resp, band);
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}

// kp_out: [n_img][kp_stride] float2, n_out: [n_img]; tmp_*: scratch sized n_img*nbins*per, cnt n_img*nbins.
// bytes of dynamic LDS the fused detector needs for this bin geometry, or 0 when it does not apply (bins wider than a
// wave's columns, or beyond 64 KB of responses + pixels)
size_t harris_fused_lds(int rows, int cols, int nbinx, int nbiny, int per) {
    const int sx = cols / nbinx, sy = rows / nbiny;
    if (sx <= 0 || sy <= 0 || sx > HW_MAXW || per > HD_MAXPER) return 0;
    const size_t tile = sx <= HW_DIRECT_MAXW ? 0 : ((size_t)(sy + 6) * HW_PITCH + 15) & ~(size_t)15;
    const size_t b = (HW_THREADS / 64) * (tile + HD_CAND * sizeof(unsigned long long));   // per wave: its bin's pixels + candidate keys
    return b <= 64 * 1024 ? b : 0;
}

// Scratch of the strip kernel for this geometry in bytes (0: the geometry, or a batch too small to fill the GPU with strips,
// keeps the wave-per-bin kernel; always 0 in the product build).  $VISO_HARRIS_STRIPS=0 / 1 forces either (tests run both).
size_t harris_strip_bytes(int n_img, int rows, int cols, int nbinx, int nbiny, int per) {
#ifndef VISO_DEBUG_VARIANTS
    (void)n_img; (void)rows; (void)cols; (void)nbinx; (void)nbiny; (void)per;
    return 0;   // the strip kernel lost (see above): not in the product build
#else
    const int sx = cols / nbinx, sy = rows / nbiny;
    if (sx <= 0 || sy <= 0 || per <= 0 || per > HD_MAXPER) return 0;
    if (sx > HW_DIRECT_MAXW || sx < (HW_DIRECT_MAXW - 1) / 2 + 1) return 0;            // a strip must not reach more than HS_NG bin columns
    const int nstrips = (nbinx * sx + HW_DIRECT_MAXW - 1) / HW_DIRECT_MAXW;
    int mode = -1;
    if (const char* e = getenv("VISO_HARRIS_STRIPS")) mode = atoi(e) != 0;
    if (mode == 0) return 0;
    if (mode < 0 && (nstrips >= nbinx || (long long)n_img * nstrips * nbiny < 16384)) return 0;   // nothing to gain / too few waves
    const size_t nb = (size_t)n_img * nbinx * nbiny;
    return nb * 2 * per * sizeof(unsigned long long) + nb * 2 * sizeof(int);
#endif
}

// images -> corners without a response image (callers check harris_fused_lds first)
// part: harris_strip_bytes(...) bytes of scratch, or null: the wave-per-bin kernel
int launch_harris_detect(hipStream_t s, const uint8_t* images, int n_img, int rows, int cols, int n_features, int nbinx,
                         int nbiny, double k, float2* tmp_kp, float* tmp_resp, int* cnt, float2* kp_out, float* resp_out,
                         int* n_out, int cap, size_t kp_stride, void* part) {
    if (n_img <= 0) return VISO_OK;
    BinArgs a;
    a.resp = nullptr; a.rows = rows; a.cols = cols; a.n_img = n_img;
    a.nbinx = nbinx; a.nbiny = nbiny; a.stridex = cols / nbinx; a.stridey = rows / nbiny;
    a.per = n_features / (nbinx * nbiny);
    a.tmp_kp = tmp_kp; a.tmp_resp = tmp_resp; a.cnt = cnt;
    const int nbins = nbinx * nbiny;
    const size_t lds = harris_fused_lds(rows, cols, nbinx, nbiny, a.per);
    if (!lds) { viso_set_error("harris: bin geometry does not fit the fused detector"); return VISO_ERR_UNSUPPORTED; }
#ifdef VISO_DEBUG_VARIANTS
    if (part && harris_strip_bytes(n_img, rows, cols, nbinx, nbiny, a.per)) {
        StripArgs sa;
        sa.nstrips = (nbinx * a.stridex + HW_DIRECT_MAXW - 1) / HW_DIRECT_MAXW;
        sa.part = reinterpret_cast<unsigned long long*>(part);
        sa.part_cnt = reinterpret_cast<int*>(sa.part + (size_t)n_img * nbins * 2 * a.per);
        const long long units = (long long)n_img * sa.nstrips * nbiny;
        hipLaunchKernelGGL(harris_strip_kernel, dim3((unsigned)((units + HW_THREADS / 64 - 1) / (HW_THREADS / 64))), dim3(HW_THREADS), 0, s, a, sa, images, k);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(harris_merge_kernel, dim3((unsigned)(((long long)n_img * nbins + 255) / 256)), dim3(256), 0, s, a, sa);
        HIP_TRY(hipGetLastError());
    } else
#endif
    {
        (void)part;
        const dim3 grid((unsigned)(((long long)n_img * nbins + HW_THREADS / 64 - 1) / (HW_THREADS / 64)));
        if (a.stridex <= HW_DIRECT_MAXW) {
            hipLaunchKernelGGL(harris_detect_kernel<true>, grid, dim3(HW_THREADS), lds, s, a, images, k);
        } else {
            if (lds > 32 * 1024)
                HIP_TRY(hipFuncSetAttribute((const void*)harris_detect_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(harris_detect_kernel<false>, grid, dim3(HW_THREADS), lds, s, a, images, k);
        }
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(harris_compact_kernel, dim3(n_img), dim3(256), sizeof(int) * (size_t)(nbins + 1), s, a, kp_out,
                       resp_out, n_out, cap, kp_stride);
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}

int launch_harris_bins(hipStream_t s, const float* resp, int n_img, int rows, int cols, int n_features, int nbinx,
                       int nbiny, float2* tmp_kp, float* tmp_resp, int* cnt, float2* kp_out, float* resp_out,
                       int* n_out, int cap, size_t kp_stride) {
    if (n_img <= 0) return VISO_OK;
    BinArgs a;
    a.resp = resp; a.rows = rows; a.cols = cols; a.n_img = n_img;
    a.nbinx = nbinx; a.nbiny = nbiny; a.stridex = cols / nbinx; a.stridey = rows / nbiny;
    a.per = n_features / (nbinx * nbiny);
    a.tmp_kp = tmp_kp; a.tmp_resp = tmp_resp; a.cnt = cnt;
    const int nbins = nbinx * nbiny;
    const long long waves = (long long)n_img * nbins;
    hipLaunchKernelGGL(harris_bins_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(harris_compact_kernel, dim3(n_img), dim3(256), sizeof(int) * (size_t)(nbins + 1), s, a, kp_out,
                       resp_out, n_out, cap, kp_stride);
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}

static int check_detect_args(int rows, int cols, int n_features, int nbinx, int nbiny) {
    if (rows <= 0 || cols <= 0 || n_features < 0 || nbinx <= 0 || nbiny <= 0) return VISO_ERR_ARG;   // assert(nbinx>0 && nbiny>0), :920
    if (cols / nbinx <= 0 || rows / nbiny <= 0) return VISO_ERR_ARG;                                   // assert(stridex>0 && stridey>0), :934
    if ((long long)nbinx * nbiny > 16384) return VISO_ERR_UNSUPPORTED;
    return VISO_OK;
}

extern "C" int viso_harris_response(const uint8_t* img, int rows, int cols, double k, float* resp) {
    if (!img || !resp || rows <= 0 || cols <= 0) { viso_set_error("viso_harris_response: bad argument"); return VISO_ERR_ARG; }
    PlainLock lk;
    viso_ctx* c = viso_default_ctx();
    if (!c) return VISO_ERR_HIP;
    uint8_t* dimg; float* dr;
    int r;
    const size_t px = (size_t)rows * cols;
    if ((r = ctx_scratch(c, 0, px, (void**)&dimg)) < 0) return r;
    if ((r = ctx_scratch(c, 1, sizeof(float) * px, (void**)&dr)) < 0) return r;
    HIP_TRY(hipMemcpyAsync(dimg, img, px, hipMemcpyHostToDevice, c->stream));
    if ((r = launch_harris_response(c->stream, dimg, 1, rows, cols, k, dr)) < 0) return r;
    HIP_TRY(hipMemcpyAsync(resp, dr, sizeof(float) * px, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VISO_OK;
}

extern "C" int viso_detect_harris_binned(const uint8_t* img, int rows, int cols, int n_features, int nbinx, int nbiny,
                                         double k, float* kp, float* resp_out, int* n_out) {
    if (!img || !n_out || (n_features > 0 && !kp)) { viso_set_error("viso_detect_harris_binned: bad argument"); return VISO_ERR_ARG; }
    int r = check_detect_args(rows, cols, n_features, nbinx, nbiny);
    if (r < 0) { viso_set_error("viso_detect_harris_binned: bad bin geometry"); return r; }
    *n_out = 0;
    const int nbins = nbinx * nbiny, per = n_features / nbins;
    if (per == 0) return VISO_OK;
    PlainLock lk;
    viso_ctx* c = viso_default_ctx();
    if (!c) return VISO_ERR_HIP;
    uint8_t* dimg; float *dr, *dtr, *dro; float2 *dtk, *dko; int* dcnt;
    const size_t px = (size_t)rows * cols, slots = (size_t)nbins * per;
    if ((r = ctx_scratch(c, 0, px, (void**)&dimg)) < 0) return r;
    if ((r = ctx_scratch(c, 1, sizeof(float) * px, (void**)&dr)) < 0) return r;
    if ((r = ctx_scratch(c, 2, sizeof(float2) * slots, (void**)&dtk)) < 0) return r;
    if ((r = ctx_scratch(c, 3, sizeof(float) * slots, (void**)&dtr)) < 0) return r;
    if ((r = ctx_scratch(c, 4, sizeof(int) * (size_t)(nbins + 4), (void**)&dcnt)) < 0) return r;
    if ((r = ctx_scratch(c, 5, sizeof(float2) * slots, (void**)&dko)) < 0) return r;
    if ((r = ctx_scratch(c, 6, sizeof(float) * slots, (void**)&dro)) < 0) return r;
    HIP_TRY(hipMemcpyAsync(dimg, img, px, hipMemcpyHostToDevice, c->stream));
    if (harris_fused_lds(rows, cols, nbinx, nbiny, per)) {
        void* part = nullptr;   // one image never fills the GPU with strips: only when $VISO_HARRIS_STRIPS=1 forces them (tests)
        if (const size_t pb = harris_strip_bytes(1, rows, cols, nbinx, nbiny, per))
            if ((r = ctx_scratch(c, 7, pb, &part)) < 0) return r;
        if ((r = launch_harris_detect(c->stream, dimg, 1, rows, cols, n_features, nbinx, nbiny, k, dtk, dtr, dcnt, dko, dro,
                                      dcnt + nbins, (int)slots, slots, part)) < 0) return r;
    } else {
        if ((r = launch_harris_response(c->stream, dimg, 1, rows, cols, k, dr)) < 0) return r;
        if ((r = launch_harris_bins(c->stream, dr, 1, rows, cols, n_features, nbinx, nbiny, dtk, dtr, dcnt, dko, dro,
                                    dcnt + nbins, (int)slots, slots)) < 0) return r;
    }
    int n = 0;
    HIP_TRY(hipMemcpyAsync(&n, dcnt + nbins, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n > 0) {
        HIP_TRY(hipMemcpy(kp, dko, sizeof(float2) * (size_t)n, hipMemcpyDeviceToHost));
        if (resp_out) HIP_TRY(hipMemcpy(resp_out, dro, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
    }
    *n_out = n;
    return VISO_OK;
}
