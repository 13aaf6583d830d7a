// match_dev.h — device helpers shared by the matcher kernels (match.hip,
// match_tile.hip).  Arithmetic here is parity critical: same operation order
// and float roundings as the reference (cited per function).
#pragma once
#include "common.h"

#include <math.h>

__device__ __forceinline__ int mbcnt(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// v = this lane's share (elements 2l, 2l+1) of a row: the four 32-element block sums (16 lanes each), clamped to int16,
// biased, packed as 4 x u16.  Valid in every lane.
__device__ __forceinline__ uint2 pack_block_sums(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);    // lane ^ 1
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);    // lane ^ 2
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);   // 7 - lane within 8
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);   // 15 - lane within 16
    v = min(max(v, -32768), 32767) + VISO_BIAS;
    const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane(v, 0), s1 = (uint32_t)__builtin_amdgcn_readlane(v, 16);
    const uint32_t s2 = (uint32_t)__builtin_amdgcn_readlane(v, 32), s3 = (uint32_t)__builtin_amdgcn_readlane(v, 48);
    return make_uint2(s0 | (s1 << 16), s2 | (s3 << 16));
}

// 8-bit plane of a packed row (ImageView::rows8) at shift s: h_s is monotone and |h(a) - h(b)| <= (|a - b| + 2^s - 1) >> s.
// The clamp comes FIRST and through inline asm: written as clamp((v + 1024) >> 3, 0, 255) on two values that are then
// packed, hipcc 7.2 selects gfx950's v_ashr_pk_u8_i32 and treats bits 31:16 of its result as zero, while the hardware
// leaves the destination's upper half as it was (0xffff where v + 1024 was negative: the neighbour's bytes of the plane
// dword came out as 255; found by tests/test_gpu_union8.py, values below -1024).
__device__ __forceinline__ uint32_t row8_of(int v, int s) {
    int x = v + (128 << s);
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(x) : "v"(x), "v"((256 << s) - 1));
    return (uint32_t)x >> s;
}

// lane l holds elements 2l, 2l+1 of row `row` (pad elements as 0): the even lanes write the plane's dwords
__device__ __forceinline__ void store_row8(uint8_t* rows8, size_t row, int lane, int va, int vb, int s) {
    const uint32_t h2 = row8_of(va, s) | (row8_of(vb, s) << 8);
    const uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h2, 0x101, 0xf, 0xf, true);   // row_shl:1: lane + 1's pair
    if ((lane & 1) == 0)
        ((__attribute__((address_space(1))) uint32_t*)reinterpret_cast<uint32_t*>(rows8 + row * VISO_ROW8))[lane >> 1] = h2 | (nb << 16);
}

// Magnitude statistics of the rows a pack wave converts (what the NEXT run's shift is chosen from): call once per row with
// the lane's two elements, then r8_flush once per wave.  Counts are element PAIRS (lanes) whose larger magnitude reaches
// 128 / 256 / 512; all wave uniform (ballots).  Only about VISO_R8_WAVES waves of a launch take part (r8_mask: a power of two
// minus one, from the launcher), and a wave sends its four counts as TWO 64-bit atomics: the counters share a cache
// line, and with every wave adding to it the pack kernels were ten times slower, with one wave in 64 still 6 % (the line's
// memory channel is on every wave's path); a few thousand rows are more than a 1-in-256 threshold needs
#define VISO_R8_WAVES 256
static inline unsigned r8_mask(long long waves) {   // host: sample waves whose index & mask == 0
    unsigned m = 1;
    while ((long long)m * VISO_R8_WAVES * 2 <= waves) m <<= 1;
    return m - 1;
}
struct R8Count { int c128, c256, c512, rows; };
__device__ __forceinline__ void r8_count(R8Count& c, int va, int vb) {
    const int m = max(abs(va), abs(vb));
    c.c128 += __popcll(__ballot(m >= 128));
    c.c256 += __popcll(__ballot(m >= 256));
    c.c512 += __popcll(__ballot(m >= 512));
    c.rows += 1;
}
__device__ __forceinline__ void r8_flush(const R8Count& c, int* row8, int lane) {
    if (lane == 0 && c.rows) {   // four consecutive ints, 8-byte aligned
        atomicAdd(reinterpret_cast<unsigned long long*>(row8 + VISO_R8_C128), (unsigned long long)(unsigned)c.c128 | ((unsigned long long)(unsigned)c.c256 << 32));
        atomicAdd(reinterpret_cast<unsigned long long*>(row8 + VISO_R8_C128 + 2), (unsigned long long)(unsigned)c.c512 | ((unsigned long long)(unsigned)c.rows << 32));
    }
}

// cvflann::L1<float> over 2 elements: result = 0; result += |a0-b0|; result += |a1-b1|
__device__ __forceinline__ float l1_kp(float qx, float qy, float2 t) {
    float r = fabsf(qx - t.x);
    r += fabsf(qy - t.y);
    return r;
}

// sampsonDistance + algebricDistance, src/viso.cpp:655-666, 390-407 — same
// operation order and the same float roundings (Q4).
__device__ __forceinline__ double sampson_dev(const double* F, float p1x, float p1y, float p2x,
                                              float p2y) {
    double Fx0 = F[0] * p1x + F[1] * p1y + F[2];
    double Fx1 = F[3] * p1x + F[4] * p1y + F[5];
    double Ftx0 = F[0] * p2x + F[3] * p2y + F[6];
    double Ftx1 = F[1] * p2x + F[4] * p2y + F[7];
    float a0 = p1x, a1 = p1y, a2 = 1.f, b0 = p2x, b1 = p2y, b2 = 1.f;
    double adv = b0 * F[0] * a0 + b0 * F[1] * a1 + b0 * F[2] * a2 + b1 * F[3] * a0 +
                 b1 * F[4] * a1 + b1 * F[5] * a2 + b2 * F[6] * a0 + b2 * F[7] * a1 +
                 b2 * F[8] * a2;
    float ad = (float)adv;
    float ad2 = ad * ad;
    return ad2 / (Fx0 * Fx0 + Fx1 * Fx1 + Ftx0 * Ftx0 + Ftx1 * Ftx1);
}

// Epipolar band (stereo call): a bound `band` such that every target p2 within L1 distance r of a query p1 of the
// box [xa,xb] x [ya,yb] with |p2.y - p1.y| > band is REJECTED by the gate of src/viso.cpp:695-701 (its computed
// Sampson value exceeds thresh, or is not finite).  The gate itself stays sampson_dev, bit for bit; the band only
// spares candidates that cannot pass from being evaluated.  Derivation (a = (x1,y1,1), b = a + (dx,dy,0)):
//   e = b' F a = a' F a + dx (F a)_0 + dy (F a)_1,  |e| >= |dy| m1 - r M0 - G
//   den = (F a)_0^2 + (F a)_1^2 + (F' b)_0^2 + (F' b)_1^2 <= D
// with m1 = min |(F a)_1|, M0 = max |(F a)_0|, M1 = max |(F a)_1| and G >= |a' F a| over the query box, N0, N1 =
// max |(F' b)_{0,1}| over the box grown by r; the computed value is >= (|e| (1 - 1e-6) - 1e-9 T)^2 (1 - 1e-6) / D
// (T bounds the sum of the absolute terms of e: double summation, one rounding to float, one float product, one
// double division — the slack is orders of magnitude above their error), so it exceeds thresh whenever
// |dy| > band.  Returns +inf (no restriction) whenever F, the box or thresh admit no finite band.
// For a rectified pair (F ~ [0 0 0; 0 0 -c; 0 c 0]) band = sqrt(2 thresh) (1 + 4e-6): |dy| <= 1 for thresh = 1.
__device__ __forceinline__ float epipolar_band(const double* F, double thresh, float xa, float xb, float ya, float yb, float r) {
    const float inf = __builtin_huge_valf();
    if (!(thresh >= 0.0) || !(r >= 0.f) || !(xa <= xb) || !(ya <= yb)) return inf;
    const double x[2] = {(double)xa, (double)xb}, y[2] = {(double)ya, (double)yb};
    const double X = fmax(fabs(x[0]), fabs(x[1])), Y = fmax(fabs(y[0]), fabs(y[1]));
    double m1 = 1.7976931348623157e308, M0 = 0, M1 = 0;
    bool pos = false, neg = false;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {   // linear functions: extrema at the corners
            const double f0 = F[0] * x[i] + F[1] * y[j] + F[2];
            const double f1 = F[3] * x[i] + F[4] * y[j] + F[5];
            M0 = fmax(M0, fabs(f0)); M1 = fmax(M1, fabs(f1));
            m1 = fmin(m1, fabs(f1));
            pos = pos || f1 > 0; neg = neg || f1 < 0;
        }
    if (pos && neg) m1 = 0;
    // corner values carry a rounding error of a few ulp of the absolute sums: shrink / grow by a relative 1e-9 of those
    const double S0 = fabs(F[0]) * X + fabs(F[1]) * Y + fabs(F[2]), S1 = fabs(F[3]) * X + fabs(F[4]) * Y + fabs(F[5]);
    m1 -= 1e-9 * S1; M0 += 1e-9 * S0; M1 += 1e-9 * S1;
    if (!(m1 > 0)) return inf;
    const double Xt = X + (double)r, Yt = Y + (double)r;
    const double N0 = fabs(F[0]) * Xt + fabs(F[3]) * Yt + fabs(F[6]), N1 = fabs(F[1]) * Xt + fabs(F[4]) * Yt + fabs(F[7]);
    const double G = fabs(F[0]) * X * X + fabs(F[1] + F[3]) * X * Y + fabs(F[4]) * Y * Y + fabs(F[2] + F[6]) * X +
                     fabs(F[5] + F[7]) * Y + fabs(F[8]) +
                     1e-9 * ((fabs(F[1]) + fabs(F[3])) * X * Y + (fabs(F[2]) + fabs(F[6])) * X + (fabs(F[5]) + fabs(F[7])) * Y);
    const double T = Xt * (fabs(F[0]) * X + fabs(F[1]) * Y + fabs(F[2])) + Yt * (fabs(F[3]) * X + fabs(F[4]) * Y + fabs(F[5])) +
                     fabs(F[6]) * X + fabs(F[7]) * Y + fabs(F[8]);
    const double D = M0 * M0 + M1 * M1 + N0 * N0 + N1 * N1;
    // The bound rests on the computed ad = (float)e and ad * ad being NORMAL floats.  Where the smallest |e| it relies
    // on, sqrt(thresh D), is below ~1e-18 its square is no longer one (a tiny-scaled F, or thresh = 0): ad * ad may
    // flush to 0, the reference's gate then ACCEPTS the candidate, and no band may cull it.
    if (!(sqrt(thresh * D) >= 1e-18)) return inf;
    const double Emin = (sqrt(thresh * D) * (1.0 + 2e-6) + 1e-9 * T) * (1.0 + 2e-6);
    const double band = ((Emin + (double)r * M0 + G) / m1) * (1.0 + 1e-6) + 1e-6;
    if (!(band == band) || band > 3.0e38) return inf;
    // round up to float: a float >= band
    float bf = (float)band;
    if ((double)bf < band) bf = __uint_as_float(__float_as_uint(bf) + 1u);
    return bf;
}

__device__ __forceinline__ bool key_less(uint32_t ad, uint32_t ai, uint32_t bd, uint32_t bi) {
    return ad < bd || (ad == bd && ai < bi);
}

// Column bucket of x.  NaN x shares the last bucket with the largest columns; a NaN product (x0 = +-inf with scale 0)
// goes to bucket 0 explicitly — never (int)NaN.
__device__ __forceinline__ int bucket_of(float x, float x0, float scale) {
    if (x != x) return VISO_NB - 1;
    const float f = floorf((x - x0) * scale);
    if (f != f) return 0;
    return f <= 0.f ? 0 : (f >= (float)(VISO_NB - 1) ? VISO_NB - 1 : (int)f);
}

// Ascending bitonic sort of `npad` (power of two, >= 64) 64-bit keys in LDS by a workgroup of THREADS threads.
// Every wave owns a contiguous chunk of comparators (CW per stage), hence a contiguous chunk of 2*CW keys: the
// stages whose partner distance j fits the chunk are wave local and need no workgroup barrier (LDS operations of one
// wave complete in order) — for 2048 keys and 8 waves that is 60 of the 66 stages.  Index arithmetic by shifts.
// Call with the keys written and a __syncthreads() done; returns after a final __syncthreads().
template <int THREADS>
__device__ __forceinline__ void bitonic_sort_lds(unsigned long long* keys, int npad) {
    constexpr int NW = THREADS / 64;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int half = npad >> 1;
    const int cw = half >= THREADS ? half / NW : 64;   // comparators per wave and stage
    const bool active = wave * cw < half;
    for (int k = 2, lk = 1; k <= npad; k <<= 1, ++lk) {
        for (int j = k >> 1, lj = lk - 1; j > 0; j >>= 1, --lj) {
            const bool cross = j > cw;   // partner in another wave's chunk
            if (cross) __syncthreads(); else __builtin_amdgcn_wave_barrier();
            if (active) {
                for (int m = lane; m < cw; m += 64) {
                    const int t = wave * cw + m;
                    const int i = ((t >> lj) << (lj + 1)) + (t & (j - 1));
                    const int l = i + j;
                    const bool up = (i & k) == 0;
                    const unsigned long long a = keys[i], b = keys[l];
                    if ((a > b) == up) { keys[i] = b; keys[l] = a; }
                }
            }
            if (cross) __syncthreads();
        }
    }
    __syncthreads();
}

