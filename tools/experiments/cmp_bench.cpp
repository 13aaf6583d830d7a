// tools/experiments/cmp_bench.cpp: how fast can one core PROVE two 968 KB blocks equal?  memcmp against hand-written loops (AVX2 /
// AVX-512 xor-or accumulation over 256 / 512 bytes per test), on blocks that were written a few hundred microseconds ago (the shadow)
// and read once (the caller's array) -- the situation of the plain family's temporal calls.  g++ -O2 -march=native.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <immintrin.h>
#include <vector>
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static bool eq_avx2(const char* a, const char* b, size_t n) {
    size_t i = 0;
    for (; i + 256 <= n; i += 256) {
        __m256i acc = _mm256_setzero_si256();
#pragma GCC unroll 8
        for (int k = 0; k < 8; ++k)
            acc = _mm256_or_si256(acc, _mm256_xor_si256(_mm256_loadu_si256((const __m256i*)(a + i + 32 * k)), _mm256_loadu_si256((const __m256i*)(b + i + 32 * k))));
        if (!_mm256_testz_si256(acc, acc)) return false;
    }
    return memcmp(a + i, b + i, n - i) == 0;
}
#ifdef __AVX512F__
static bool eq_avx512(const char* a, const char* b, size_t n) {
    size_t i = 0;
    for (; i + 512 <= n; i += 512) {
        __m512i acc = _mm512_setzero_si512();
#pragma GCC unroll 8
        for (int k = 0; k < 8; ++k)
            acc = _mm512_or_si512(acc, _mm512_xor_si512(_mm512_loadu_si512(a + i + 64 * k), _mm512_loadu_si512(b + i + 64 * k)));
        if (_mm512_test_epi64_mask(acc, acc)) return false;
    }
    return memcmp(a + i, b + i, n - i) == 0;
}
#endif
static bool eq_prefetch(const char* a, const char* b, size_t n) {
    size_t i = 0;
    for (; i + 256 <= n; i += 256) {
        _mm_prefetch(a + i + 2048, _MM_HINT_T0); _mm_prefetch(b + i + 2048, _MM_HINT_T0);
        _mm_prefetch(a + i + 2048 + 64, _MM_HINT_T0); _mm_prefetch(b + i + 2048 + 64, _MM_HINT_T0);
        _mm_prefetch(a + i + 2048 + 128, _MM_HINT_T0); _mm_prefetch(b + i + 2048 + 128, _MM_HINT_T0);
        _mm_prefetch(a + i + 2048 + 192, _MM_HINT_T0); _mm_prefetch(b + i + 2048 + 192, _MM_HINT_T0);
        __m256i acc = _mm256_setzero_si256();
#pragma GCC unroll 8
        for (int k = 0; k < 8; ++k)
            acc = _mm256_or_si256(acc, _mm256_xor_si256(_mm256_loadu_si256((const __m256i*)(a + i + 32 * k)), _mm256_loadu_si256((const __m256i*)(b + i + 32 * k))));
        if (!_mm256_testz_si256(acc, acc)) return false;
    }
    return memcmp(a + i, b + i, n - i) == 0;
}
int main() {
    const size_t n = 968000, NBUF = 64;   // 64 pairs: 124 MB, far more than a core's L2; the L3 holds some of it
    std::vector<char*> A(NBUF), B(NBUF);
    for (size_t k = 0; k < NBUF; ++k) {
        A[k] = (char*)aligned_alloc(64, n + 64); B[k] = (char*)aligned_alloc(64, n + 64);
        for (size_t i = 0; i < n; ++i) A[k][i] = (char)(i * 131 + k);
        memcpy(B[k], A[k], n);
    }
    struct { const char* name; bool (*f)(const char*, const char*, size_t); } fn[] = {
        {"memcmp", [](const char* a, const char* b, size_t m) { return memcmp(a, b, m) == 0; }},
        {"avx2 xor-or / 256 B", eq_avx2},
#ifdef __AVX512F__
        {"avx512 xor-or / 512 B", eq_avx512},
#endif
        {"avx2 + prefetch 2 KB ahead", eq_prefetch},
    };
    for (int rep = 0; rep < 2; ++rep)
        for (auto& f : fn) {
            // (a) cold-ish: walk all pairs once (each pair was last touched 63 pairs ago)
            double t0 = now_us(); int ok = 0;
            for (size_t k = 0; k < NBUF; ++k) ok += f.f(A[k], B[k], n);
            const double cold = (now_us() - t0) / NBUF;
            // (b) warm: the same pair again and again (both blocks in L2 / L3)
            t0 = now_us();
            for (int r = 0; r < 64; ++r) ok += f.f(A[0], B[0], n);
            const double warm = (now_us() - t0) / 64;
            // (c) the loop's situation: copy a block (the shadow is written), do ~2 MB of other traffic, then compare
            double tc = 0;
            for (size_t k = 0; k < NBUF; ++k) {
                memcpy(B[k], A[k], n);
                memcpy(B[(k + 7) % NBUF], A[(k + 7) % NBUF], n);
                const double t1 = now_us();
                ok += f.f(A[k], B[k], n);
                tc += now_us() - t1;
            }
            printf("%-28s walk %.1f us  warm %.1f us  after copy + 1 MB of other traffic %.1f us  (%d)\n", f.name, cold, warm, tc / NBUF, ok);
        }
    return 0;
}
