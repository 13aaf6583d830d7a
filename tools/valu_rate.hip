// valu_rate.hip — what one SIMD of gfx950 sustains for the instructions the matcher's scoring loop is made of.
// Every wave runs ITERS iterations of UNROLL independent chains of one instruction kind; 8 waves per SIMD; the
// figure printed is cycles per wave-instruction per SIMD (wall time x clock x SIMDs / instructions).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o tools/valu_rate && tools/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITERS 4096
#define CHAINS 8

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t* out, uint32_t seed) {
    uint32_t a[CHAINS], b = seed ^ threadIdx.x, c = seed * 3u + blockIdx.x;
    double d[CHAINS], db = 1.0 + 1e-9 * threadIdx.x, dc = 1e-12 * seed;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) d[i] = 1.0 + i * 0.001;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) a[i] = seed + i * 77u + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (KIND == 0) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (KIND == 1) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (KIND == 3) asm volatile("v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
            if (KIND == 4) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (KIND == 5) asm volatile("v_min_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (KIND == 6) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (KIND == 7) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (KIND == 8) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (KIND == 9) asm volatile("v_lshl_or_b32 %0, %0, 9, %1" : "+v"(a[i]) : "v"(b));
            if (KIND == 10) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(db), "v"(dc));
            if (KIND == 11) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[i]) : "v"(db));
            if (KIND == 12) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(d[i]) : "v"(db));
            if (KIND == 13) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += a[i] + (uint32_t)d[i];
    if (s == 0x12345678u) out[0] = s;   // never true in practice; keeps the chains alive
}

template <int KIND>
static void run(const char* name, uint32_t* d, double clk_ghz) {
    const int blocks = 256 * 8;   // 8 blocks of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d, 2u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 * ITERS * CHAINS;   // wave-instructions
    const double cyc = ms * 1e-3 * clk_ghz * 1e9 * 1024 / insts;
    printf("%-28s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, ms, cyc, clk_ghz);
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 64);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const double ghz = clk_khz * 1e-6;
    run<7>("v_fma_f32", d, ghz);
    run<1>("v_add_u32", d, ghz);
    run<0>("v_sad_u16", d, ghz);
    run<6>("v_sad_u32", d, ghz);
    run<8>("v_sad_u8", d, ghz);
    run<3>("v_add_u32 dpp quad_perm", d, ghz);
    run<4>("v_med3_u32", d, ghz);
    run<5>("v_min_u32", d, ghz);
    run<9>("v_lshl_or_b32", d, ghz);
    run<10>("v_fma_f64", d, ghz);
    run<11>("v_add_f64", d, ghz);
    run<12>("v_mul_f64", d, ghz);
    run<13>("v_rcp_f64", d, ghz);
    return 0;
}
