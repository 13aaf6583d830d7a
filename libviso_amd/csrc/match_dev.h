// match_dev.h — device helpers shared by the matcher kernels (match.hip,
// match_tile.hip).  Arithmetic here is parity critical: same operation order
// and float roundings as the reference (cited per function).
#pragma once
#include "common.h"

#include <math.h>

__device__ __forceinline__ int mbcnt(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
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

__device__ __forceinline__ bool key_less(uint32_t ad, uint32_t ai, uint32_t bd, uint32_t bi) {
    return ad < bd || (ad == bd && ai < bi);
}

__device__ __forceinline__ int bucket_of(float x, float x0, float scale) {
    if (x != x) return VISO_NB - 1;
    const float f = floorf((x - x0) * scale);
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

