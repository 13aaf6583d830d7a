// valu_rate.hip — what one SIMD of gfx950 sustains for the instructions the matcher's scoring loop is made of.
// Every wave runs ITERS iterations of CHAINS independent chains of one instruction kind; 8 waves per SIMD.  Two
// figures per kind: cycles per wave-instruction per SIMD from the wall time at the nominal clock, and from the
// shader-cycle counter read inside the kernel (s_memtime ticks = shader cycles: no clock assumption, no launch ramp).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o tools/valu_rate && tools/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITERS 4096
#define CHAINS 8

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t* out, uint32_t seed, unsigned long long* cyc) {
    uint32_t a[CHAINS], b = seed ^ threadIdx.x, c = seed * 3u + blockIdx.x;
    double d[CHAINS], db = 1.0 + 1e-9 * threadIdx.x, dc = 1e-12 * seed;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) d[i] = 1.0 + i * 0.001;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) a[i] = seed + i * 77u + threadIdx.x;
    const unsigned long long sel = seed * 0x9E3779B97F4A7C15ULL;   // a lane mask for v_cndmask
    unsigned long long sm[2] = {0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
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
            if (KIND == 14) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(sel));
            if (KIND == 15) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (KIND == 16) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
            if (KIND == 17) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (KIND == 18) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
            if (KIND == 19) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b));
            if (KIND == 20) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (KIND == 21) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(a[i]));
            if (KIND == 22) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (KIND == 23) asm volatile("v_max_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (KIND == 24) asm volatile("v_or_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (KIND == 25) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (KIND == 26) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (KIND == 27) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (KIND == 28) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            if (KIND == 29) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
            if (KIND == 30) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (KIND == 31) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (KIND == 32) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (KIND == 33) asm volatile("v_add_u32_dpp %0, %1, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
            if (KIND == 34) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a[i]) : "v"(b));
            if (KIND == 35) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (KIND == 36) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (KIND == 37) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b));
            if (KIND == 38) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b));
            if (KIND == 39) asm volatile("v_add_u32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5" : "+v"(a[i]) : "v"(b));
            if (KIND == 40) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (KIND == 41) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 42) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(db), "v"(dc));
            if (KIND == 43) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "s"(sel), "v"(dc));
            if (KIND == 44) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(d[i]) : "v"(db));
            if (KIND == 45) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(d[i]) : "v"(db));
            if (KIND == 46) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 47) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(sm[i & 1]) : "v"(a[i]), "v"(b));
            if (KIND == 48) asm volatile("v_add_f32_e64 %0, |%1|, |%0|" : "+v"(a[i]) : "v"(b));
            if (KIND == 49) asm volatile("v_fma_f32 %0, |%1|, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (KIND == 50) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(seed), "v"(c));
            if (KIND == 51) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(d[i]) : "s"(sel), "v"(dc));
            if (KIND == 52) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (KIND == 53) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;   // wave 0 of the workgroup
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += a[i] + (uint32_t)d[i];
    s += (uint32_t)(sm[0] ^ sm[1]);
    if (s == 0x12345678u) out[0] = s;   // never true in practice; keeps the chains alive
}

template <int KIND>
static void run(const char* name, uint32_t* d, double clk_ghz, int per_iter = 1) {
    const int blocks = 256 * 8;   // 8 blocks of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    static unsigned long long* dc = nullptr;
    if (!dc) hipMalloc(&dc, sizeof(unsigned long long) * blocks);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1u, dc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d, 2u, dc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 * ITERS * CHAINS * per_iter;   // wave-instructions
    const double cyc = ms * 1e-3 * clk_ghz * 1e9 * 1024 / insts;
    static unsigned long long hc[256 * 8];
    hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < blocks; ++i) sum += (double)hc[i];
    // a wave's loop lasts `ticks` shader cycles while its SIMD issues the streams of 8 waves
    const double true_cyc = sum / blocks / (8.0 * ITERS * CHAINS * per_iter);
    printf("%-28s %8.3f ms  %6.2f cycles per wave-instruction per SIMD at the nominal %.2f GHz, %6.2f by the shader-cycle counter\n",
           name, ms, cyc, clk_ghz, true_cyc);
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
    run<14>("v_cndmask_b32 (sgpr mask)", d, ghz);
    run<22>("v_cmp_lt_u32 + v_cndmask", d, ghz, 2);
    run<15>("v_and_b32", d, ghz);
    run<24>("v_or_b32", d, ghz);
    run<16>("v_lshlrev_b32", d, ghz);
    run<17>("v_sub_u32", d, ghz);
    run<23>("v_max_u32", d, ghz);
    run<18>("v_mov_b32", d, ghz);
    run<19>("v_lshl_add_u32", d, ghz);
    run<20>("v_add3_u32", d, ghz);
    run<21>("v_bfe_u32", d, ghz);
    run<25>("v_add_f32", d, ghz);
    run<26>("v_mul_lo_u32", d, ghz);
    run<27>("v_cndmask_b32_e32 (vcc)", d, ghz);
    run<28>("v_cmp_lt_u32_e32", d, ghz);
    run<29>("v_mov_b32 dpp quad_perm", d, ghz);
    run<33>("v_add_u32 dpp half_mirror", d, ghz);
    run<30>("v_pk_sub_u16", d, ghz);
    run<31>("v_pk_max_u16", d, ghz);
    run<32>("v_xor_b32", d, ghz);
    run<34>("v_alignbit_b32", d, ghz);
    run<35>("v_perm_b32", d, ghz);
    run<36>("v_bfi_b32", d, ghz);
    run<37>("v_permlane32_swap_b32", d, ghz);
    run<38>("v_permlane16_swap_b32", d, ghz);
    run<39>("v_add_u32 dpp row_shl:4 bank_mask", d, ghz);
    run<40>("v_bcnt_u32_b32", d, ghz);
    run<41>("v_cvt_i32_f32", d, ghz);
    run<42>("v_pk_fma_f32", d, ghz);
    run<43>("v_pk_fma_f32 (sgpr pair operand)", d, ghz);
    run<51>("v_pk_fma_f32 (sgpr, op_sel_hi splat)", d, ghz);
    run<44>("v_pk_mul_f32", d, ghz);
    run<45>("v_pk_add_f32", d, ghz);
    run<46>("v_rcp_f32", d, ghz);
    run<47>("v_cmp_lt_f32_e64 (sgpr dst)", d, ghz);
    run<53>("v_cmp_lt_f32_e32 (vcc)", d, ghz);
    run<48>("v_add_f32 |a|, |b|", d, ghz);
    run<49>("v_fma_f32 |a|, b, c", d, ghz);
    run<50>("v_fma_f32 (sgpr operand)", d, ghz);
    run<52>("v_mul_f32", d, ghz);
    return 0;
}
