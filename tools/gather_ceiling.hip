// gather_ceiling.hip — microbenchmark behind DESIGN.md's "gather ceiling": how fast can one MI355X serve the
// matcher's access pattern (8 lanes read one 256-B descriptor row as 2 x 16 B per lane, rows picked at random
// inside a ~330-row window of a 512-KB image that sits in L2), with nothing else in the kernel?
// Also reports the shader clock the chip actually ran at (s_memtime vs the 100 MHz s_memrealtime).
//   hipcc -O3 --offload-arch=gfx950 tools/gather_ceiling.hip -o gpurun_out/gather_ceiling && gpurun_out/gather_ceiling
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NP>
__global__ __launch_bounds__(256) void gather_kernel(const u32x4* __restrict__ rows, int rows_per_image, int n_images,
                                                     int window, int passes, uint32_t* out, unsigned long long* clk) {
    const int lane = threadIdx.x & 63, sub = lane & 7, g8 = lane >> 3;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int image = (blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) % (n_images / 8));   // blocks b, b+8 share an XCD
    const int tile = (blockIdx.x >> 3) / (n_images / 8);
    const int lo = (tile * 32) % (rows_per_image - window);
    const u32x4* base = rows + (size_t)image * rows_per_image * 16;
    uint32_t rng = 0x9e3779b9u * (uint32_t)(wave_global * 8 + g8 + 1);
    const unsigned long long c0 = __builtin_readcyclecounter(), r0c = wall_clock64();
    u32x4 a[NP], b[NP];
    uint32_t acc = 0;
    auto issue = [&](int s) {
        rng = rng * 1664525u + 1013904223u;
        const int row = lo + (int)((rng >> 8) % (uint32_t)window);
        const u32x4* p = base + (size_t)row * 16 + sub;
        a[s] = p[0];
        b[s] = p[8];
    };
#pragma unroll
    for (int s = 0; s < NP; ++s) issue(s);
    for (int t = 0; t < passes; t += NP) {
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            uint32_t v = __builtin_amdgcn_sad_u16(a[s].x, 0x12345678u, 0u);
            v = __builtin_amdgcn_sad_u16(a[s].y, 0x12345678u, v);
            v = __builtin_amdgcn_sad_u16(a[s].z, 0x12345678u, v);
            v = __builtin_amdgcn_sad_u16(a[s].w, 0x12345678u, v);
            v = __builtin_amdgcn_sad_u16(b[s].x, 0x12345678u, v);
            v = __builtin_amdgcn_sad_u16(b[s].y, 0x12345678u, v);
            v = __builtin_amdgcn_sad_u16(b[s].z, 0x12345678u, v);
            v = __builtin_amdgcn_sad_u16(b[s].w, 0x12345678u, v);
            acc += v;
            issue(s);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1c = wall_clock64();
    if (acc == 0x7fffffffu) out[wave_global] = acc;
    if (threadIdx.x == 0) {
        atomicAdd(&clk[0], c1 - c0);
        atomicAdd(&clk[1], r1c - r0c);
    }
}

template <int NP>
static int run(const u32x4* rows, int rpi, int n_images, int window, int passes, int blocks, uint32_t* out,
               unsigned long long* clk) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(clk, 0, 16));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather_kernel<NP>, dim3(blocks), dim3(256), 0, 0, rows, rpi, n_images, window, passes, out, clk);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
    }
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2];
    CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double pairs = (double)blocks * 4 * 8 * (passes + NP);
    const double bytes = pairs * 256.0;
    printf("{\"np\": %d, \"blocks\": %d, \"passes\": %d, \"window\": %d, \"ms\": %.4f, \"pairs_per_s\": %.4g, \"gathered_TBps\": %.3f, "
           "\"shader_clock_MHz\": %.1f, \"bytes_per_clk_per_cu\": %.2f}\n",
           NP, blocks, passes, window, ms, pairs / (ms * 1e-3), bytes / (ms * 1e-3) / 1e12,
           (double)h[0] / (double)h[1] * 100.0, bytes / (ms * 1e-3) / 256.0 / ((double)h[0] / (double)h[1] * 1e8));
    return 0;
}

int main() {
    const int rpi = 2000, n_images = 512, window = 330;
    const size_t n = (size_t)n_images * rpi * 16;
    u32x4* rows; uint32_t* out; unsigned long long* clk;
    CHECK(hipMalloc(&rows, n * sizeof(u32x4)));
    CHECK(hipMemset(rows, 0x5a, n * sizeof(u32x4)));
    CHECK(hipMalloc(&out, 1 << 22));
    CHECK(hipMalloc(&clk, 16));
    const int blocks = 512 * 63;   // as the temporal matcher launch: 512 problems x 63 tiles of 32 queries
    const int passes = 52;         // 2 rounds x 26 passes per wave, as the matcher
    if (run<2>(rows, rpi, n_images, window, passes, blocks, out, clk)) return 1;
    if (run<4>(rows, rpi, n_images, window, passes, blocks, out, clk)) return 1;
    if (run<8>(rows, rpi, n_images, window, passes, blocks, out, clk)) return 1;
    for (int f = 2; f <= 32; f *= 2)   // fewer, longer blocks: the same work in blocks that live f times longer
        if (run<4>(rows, rpi, n_images, window, passes * f, blocks / f, out, clk)) return 1;
    return 0;
}
