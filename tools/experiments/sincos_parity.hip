// device sincos (ocml) against the host's libm sin / cos, bit for bit, over several argument ranges:
// hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o tools/experiments/sincos_parity tools/experiments/sincos_parity.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
__global__ void k(const double* x, double* s, double* c, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { double sv, cv; sincos(x[i], &sv, &cv); s[i] = sv; c[i] = cv; }
}
int main() {
    const int n = 1 << 20;
    const double ranges[] = {0.01, 0.5, 3.2, 100.0, 1e4, 1e9, 1e15};
    std::vector<double> x(n), s(n), c(n);
    double *dx, *ds, *dc;
    hipMalloc(&dx, n * 8); hipMalloc(&ds, n * 8); hipMalloc(&dc, n * 8);
    srand48(7);
    for (double r : ranges) {
        for (int i = 0; i < n; ++i) x[i] = (2 * drand48() - 1) * r;
        hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
        k<<<n / 256, 256>>>(dx, ds, dc, n);
        hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), dc, n * 8, hipMemcpyDeviceToHost);
        long bad_s = 0, bad_c = 0; double worst = 0;
        for (int i = 0; i < n; ++i) {
            const double hs = sin(x[i]), hc = cos(x[i]);
            if (memcmp(&hs, &s[i], 8)) { ++bad_s; worst = fmax(worst, fabs(hs - s[i]) / fmax(fabs(hs), 1e-300)); }
            if (memcmp(&hc, &c[i], 8)) { ++bad_c; worst = fmax(worst, fabs(hc - c[i]) / fmax(fabs(hc), 1e-300)); }
        }
        printf("|x| < %-8g: sin differs in %ld, cos in %ld of %d (largest relative difference %.3g)\n", r, bad_s, bad_c, n, worst);
    }
    return 0;
}
