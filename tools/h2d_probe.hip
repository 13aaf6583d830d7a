// h2d_probe — what a synchronous host->device->host round trip costs on this box: the floor of one plain-family call
// (tools/README.md).  pageable hipMemcpyAsync vs staging through pinned memory with 1..T copy threads, an empty kernel
// + 4-byte read-back, and the single-thread memcpy rate.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void empty_kernel(int* p) { if (threadIdx.x == 0) p[0] += 1; }

__global__ void blit_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}
__global__ void blit_stride_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

struct Pool {   // spinning helpers: thread i copies slice i when `gen` advances
    std::vector<std::thread> th; std::atomic<int> gen{0}, done{0}; std::atomic<bool> stop{false};
    const char* src = nullptr; char* dst = nullptr; size_t bytes = 0; int n = 0;
    explicit Pool(int n_) : n(n_) {
        for (int i = 0; i < n; ++i) th.emplace_back([this, i] {
            int seen = 0;
            while (!stop.load(std::memory_order_relaxed)) {
                if (gen.load(std::memory_order_acquire) == seen) { __builtin_ia32_pause(); continue; }
                ++seen;
                const size_t lo = bytes * i / (n + 1), hi = bytes * (i + 1) / (n + 1);
                memcpy(dst + lo, src + lo, hi - lo);
                done.fetch_add(1, std::memory_order_release);
            }
        });
    }
    void copy(char* d, const char* s, size_t b) {   // caller takes the last slice
        src = s; dst = d; bytes = b; done.store(0); gen.fetch_add(1, std::memory_order_release);
        const size_t lo = b * n / (n + 1);
        memcpy(d + lo, s + lo, b - lo);
        while (done.load(std::memory_order_acquire) < n) __builtin_ia32_pause();
    }
    ~Pool() { stop = true; gen.fetch_add(1); for (auto& t : th) t.join(); }
};

__global__ void flag_kernel(int* flag, int seq) {
    if (threadIdx.x == 0) { __threadfence_system(); __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}

int main() {
    const size_t B = 2000 * 121 * 4;   // one image's descriptors
    char *pin, *dev; int* dflag; int* hflag;
    CK(hipHostMalloc((void**)&pin, 4 * B)); CK(hipMalloc((void**)&dev, 4 * B)); CK(hipMalloc((void**)&dflag, 64)); CK(hipHostMalloc((void**)&hflag, 64));
    CK(hipMemset(dflag, 0, 64));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<std::vector<char>> src(16, std::vector<char>(2 * B, 1));   // rotate: not always cache hot
    const int R = 200;
    // (0) empty kernel + 4-byte pinned read-back + synchronize
    for (int w = 0; w < 2; ++w) {
        double t0 = now();
        for (int i = 0; i < R; ++i) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, dflag); CK(hipMemcpyAsync(hflag, dflag, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }
        if (w) printf("empty kernel + 4 B D2H (pinned) + sync: %.1f us\n", (now() - t0) / R);
    }
    for (int w = 0; w < 2; ++w) {
        double t0 = now();
        for (int i = 0; i < R; ++i) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, dflag); CK(hipStreamSynchronize(s)); }
        if (w) printf("empty kernel + sync: %.1f us\n", (now() - t0) / R);
    }
    // (round 6) the same round trip with the host SPINNING on a word of pinned memory that the kernel writes (system-scope
    // release) instead of hipStreamSynchronize: what the completion signal's path costs
    for (int w = 0; w < 2; ++w) {
        double t0 = now();
        volatile int* f = hflag;
        for (int i = 0; i < R; ++i) {
            const int seq = w * R + i + 1;
            hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, hflag, seq);
            while (*f != seq) __builtin_ia32_pause();
        }
        if (w) printf("kernel that writes a pinned flag + host spin on it (no synchronize): %.1f us\n", (now() - t0) / R);
        CK(hipStreamSynchronize(s));
    }
    { int x; double t0 = now();
      for (int i = 0; i < R; ++i) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, dflag); CK(hipMemcpyAsync(&x, dflag, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }
      printf("empty kernel + 4 B D2H (pageable) + sync: %.1f us\n", (now() - t0) / R); }
    { double t0 = now();
      for (int i = 0; i < R; ++i) { for (int k = 0; k < 6; ++k) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, dflag); CK(hipStreamSynchronize(s)); }
      printf("6 empty kernels + sync: %.1f us\n", (now() - t0) / R); }
    // (1) pageable hipMemcpyAsync H2D, then sync
    for (size_t bytes : {B, 2 * B}) {
        double t0 = now();
        for (int i = 0; i < R; ++i) { CK(hipMemcpyAsync(dev, src[i % 16].data(), bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }
        double dt = (now() - t0) / R;
        printf("pageable H2D %zu B + sync: %.1f us = %.1f GB/s\n", bytes, dt, bytes / dt * 1e-3);
    }
    // (2) single-thread memcpy to pinned + pinned H2D
    for (size_t bytes : {B, 2 * B}) {
        double t0 = now(), tc = 0;
        for (int i = 0; i < R; ++i) { double a = now(); memcpy(pin, src[i % 16].data(), bytes); tc += now() - a; CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }
        double dt = (now() - t0) / R;
        printf("memcpy->pinned (%.1f us = %.1f GB/s) + pinned H2D %zu B + sync: %.1f us\n", tc / R, bytes / (tc / R) * 1e-3, bytes, dt);
    }
    // (3) chunked: memcpy chunk k while chunk k-1 is on the wire
    for (int chunks : {2, 4, 8}) {
        const size_t bytes = 2 * B, cb = bytes / chunks;
        double t0 = now();
        for (int i = 0; i < R; ++i) {
            for (int k = 0; k < chunks; ++k) { memcpy(pin + k * cb, src[i % 16].data() + k * cb, cb); CK(hipMemcpyAsync(dev + k * cb, pin + k * cb, cb, hipMemcpyHostToDevice, s)); }
            CK(hipStreamSynchronize(s));
        }
        double dt = (now() - t0) / R;
        printf("chunked x%d memcpy->pinned + H2D %zu B + sync: %.1f us = %.1f GB/s\n", chunks, bytes, dt, bytes / dt * 1e-3);
    }
    // (4) T helper threads copy slices, then one H2D
    for (int T : {1, 2, 3, 5, 7}) {
        Pool pool(T);
        const size_t bytes = 2 * B;
        double t0 = now(), tc = 0;
        for (int i = 0; i < R; ++i) { double a = now(); pool.copy(pin, src[i % 16].data(), bytes); tc += now() - a; CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }
        double dt = (now() - t0) / R;
        printf("%d+1 threads memcpy->pinned (%.1f us = %.1f GB/s) + H2D %zu B + sync: %.1f us = %.1f GB/s\n", T, tc / R, bytes / (tc / R) * 1e-3, bytes, dt, bytes / dt * 1e-3);
    }
    // (6) blit kernels instead of the copy engine: the GPU reads pinned host memory / writes it itself
    {
        const size_t small = 34 * 1024, res = 24 * 1024;
        for (int w = 0; w < 2; ++w) {
            double t0 = now();
            for (int i = 0; i < R; ++i) {
                CK(hipMemcpyAsync(dev, pin, small, hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, dflag);
                CK(hipMemcpyAsync(pin + B, dev, res, hipMemcpyDeviceToHost, s));
                CK(hipStreamSynchronize(s));
            }
            if (w) printf("small call, copy engine: H2D 34 KB + kernel + D2H 24 KB + sync: %.1f us\n", (now() - t0) / R);
        }
        for (int w = 0; w < 2; ++w) {
            double t0 = now();
            for (int i = 0; i < R; ++i) {
                hipLaunchKernelGGL(blit_kernel, dim3((small / 16 + 255) / 256), dim3(256), 0, s, (const uint4*)pin, (uint4*)dev, small / 16);
                hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, dflag);
                hipLaunchKernelGGL(blit_kernel, dim3((res / 16 + 255) / 256), dim3(256), 0, s, (const uint4*)dev, (uint4*)(pin + B), res / 16);
                CK(hipStreamSynchronize(s));
            }
            if (w) printf("small call, blit kernels: pinned->dev 34 KB + kernel + dev->pinned 24 KB + sync: %.1f us\n", (now() - t0) / R);
        }
        for (int w = 0; w < 2; ++w) {
            double t0 = now();
            for (int i = 0; i < R; ++i) {
                hipLaunchKernelGGL(blit_kernel, dim3((small / 16 + 255) / 256), dim3(256), 0, s, (const uint4*)pin, (uint4*)(pin + B), res / 16);
                CK(hipStreamSynchronize(s));
            }
            if (w) printf("small call, ONE kernel reading and writing pinned memory + sync: %.1f us\n", (now() - t0) / R);
        }
        for (size_t bytes : {B, 2 * B}) {
            for (int blocks : {64, 256, 1024}) {
                double t0 = now();
                for (int i = 0; i < R; ++i) {
                    hipLaunchKernelGGL(blit_stride_kernel, dim3(blocks), dim3(256), 0, s, (const uint4*)pin, (uint4*)dev, bytes / 16);
                    CK(hipStreamSynchronize(s));
                }
                double dt = (now() - t0) / R;
                printf("blit pinned->dev %zu B, %d blocks + sync: %.1f us = %.1f GB/s\n", bytes, blocks, dt, bytes / dt * 1e-3);
            }
            double t0 = now();
            for (int i = 0; i < R; ++i) { CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }
            double dt = (now() - t0) / R;
            printf("copy engine pinned->dev %zu B + sync: %.1f us = %.1f GB/s\n", bytes, dt, bytes / dt * 1e-3);
        }
    }
    // (5) D2H of 24 KB: pinned vs pageable
    { std::vector<char> out(24000); double t0 = now();
      for (int i = 0; i < R; ++i) { CK(hipMemcpyAsync(pin, dev, 24000, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); memcpy(out.data(), pin, 24000); }
      printf("D2H 24 KB pinned + sync + memcpy: %.1f us\n", (now() - t0) / R);
      t0 = now();
      for (int i = 0; i < R; ++i) { CK(hipMemcpyAsync(out.data(), dev, 24000, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }
      printf("D2H 24 KB pageable + sync: %.1f us\n", (now() - t0) / R);
      t0 = now();
      for (int i = 0; i < R; ++i) { CK(hipMemcpy(out.data(), dev, 24000, hipMemcpyDeviceToHost)); }
      printf("D2H 24 KB blocking hipMemcpy pageable: %.1f us\n", (now() - t0) / R); }
    return 0;
}
