// harris.hip — binned Harris corner detector on device: the reference's
// HarrisBinnedFeatureDetector::detectImpl (src/viso.cpp:926-975) with
// cv::cornerHarris(image, R, blockSize 3, ksize 5, k, BORDER_DEFAULT) restated
// (SURVEY.md 8(f) row 2).  The reference never initialises its k
// (src/viso.cpp:915-919,978) and leaves the order inside a bin to
// std::nth_element (:963); here k is an explicit argument and a bin's corners
// are ordered by (|response| desc, push order asc).  Arithmetic contract (shared
// with the oracle): exact integer 5x5 Sobel sums, dx = (float)Dx * (float)scale,
// cov products in float, 3x3 box sums added row-major in float with
// BORDER_REFLECT_101 on the cov image, R = (float)((double)(a*c - b*b) - k*(a+c)*(a+c)).
#include "common.h"

#define HT_X 64
#define HT_Y 16
#define HT_THREADS 256

__device__ __forceinline__ int h_reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// One workgroup = one 64x16 tile of the response map.  LDS stages: reflected
// uint8 halo tile -> horizontal 5-tap sums (derivative and smoothing) -> vertical
// 5-tap sums = Dx, Dy -> cov (3 floats) on the tile + 1 ring -> 3x3 box + response.
__global__ __launch_bounds__(HT_THREADS) void harris_response_kernel(const uint8_t* __restrict__ images, int rows,
                                                                     int cols, double k, float* __restrict__ resp) {
    __shared__ unsigned char s_img[HT_Y + 6][HT_X + 8];
    __shared__ int s_hd[HT_Y + 6][HT_X + 2];
    __shared__ int s_hs[HT_Y + 6][HT_X + 2];
    __shared__ float s_cov[3][HT_Y + 2][HT_X + 2];
    const int img = blockIdx.z;
    const int tx0 = blockIdx.x * HT_X, ty0 = blockIdx.y * HT_Y;
    const uint8_t* im = images + (size_t)img * rows * cols;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < (HT_Y + 6) * (HT_X + 6); idx += HT_THREADS) {
        const int ly = idx / (HT_X + 6), lx = idx % (HT_X + 6);
        const int gy = h_reflect101(ty0 - 3 + ly, rows), gx = h_reflect101(tx0 - 3 + lx, cols);
        s_img[ly][lx] = im[(size_t)gy * cols + gx];
    }
    __syncthreads();
    for (int idx = tid; idx < (HT_Y + 6) * (HT_X + 2); idx += HT_THREADS) {
        const int ly = idx / (HT_X + 2), lx = idx % (HT_X + 2);   // column of global x = tx0 - 1 + lx
        const int p0 = s_img[ly][lx], p1 = s_img[ly][lx + 1], p2 = s_img[ly][lx + 2], p3 = s_img[ly][lx + 3],
                  p4 = s_img[ly][lx + 4];
        s_hd[ly][lx] = -p0 - 2 * p1 + 2 * p3 + p4;
        s_hs[ly][lx] = p0 + 4 * p1 + 6 * p2 + 4 * p3 + p4;
    }
    __syncthreads();
    const float scale = (float)(1.0 / (16.0 * 3.0 * 255.0));
    for (int idx = tid; idx < (HT_Y + 2) * (HT_X + 2); idx += HT_THREADS) {
        const int ly = idx / (HT_X + 2), lx = idx % (HT_X + 2);   // row of global y = ty0 - 1 + ly
        const int Dx = s_hd[ly][lx] + 4 * s_hd[ly + 1][lx] + 6 * s_hd[ly + 2][lx] + 4 * s_hd[ly + 3][lx] + s_hd[ly + 4][lx];
        const int Dy = -s_hs[ly][lx] - 2 * s_hs[ly + 1][lx] + 2 * s_hs[ly + 3][lx] + s_hs[ly + 4][lx];
        const float dx = (float)Dx * scale, dy = (float)Dy * scale;
        s_cov[0][ly][lx] = dx * dx;
        s_cov[1][ly][lx] = dx * dy;
        s_cov[2][ly][lx] = dy * dy;
    }
    __syncthreads();
    for (int idx = tid; idx < HT_Y * HT_X; idx += HT_THREADS) {
        const int oy = idx / HT_X, ox = idx % HT_X;
        const int gy = ty0 + oy, gx = tx0 + ox;
        if (gy >= rows || gx >= cols) continue;
        float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
        for (int i = -1; i <= 1; ++i) {
            const int ly = h_reflect101(gy + i, rows) - (ty0 - 1);
#pragma unroll
            for (int j = -1; j <= 1; ++j) {
                const int lx = h_reflect101(gx + j, cols) - (tx0 - 1);
                a += s_cov[0][ly][lx];
                b += s_cov[1][ly][lx];
                c += s_cov[2][ly][lx];
            }
        }
        const float t1 = a * c, t2 = b * b;
        const float t3 = t1 - t2;
        const float tr = a + c;
        resp[((size_t)img * rows + gy) * cols + gx] = (float)((double)t3 - k * (double)tr * (double)tr);
    }
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
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const unsigned long long o = __shfl_xor(v, m);
        v = o > v ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(256) void harris_bins_kernel(BinArgs a) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nbins = a.nbinx * a.nbiny;
    if (wave >= (long long)a.n_img * nbins) return;
    const int img = (int)(wave / nbins), bin = (int)(wave % nbins);
    const int bx = bin / a.nbiny, by = bin % a.nbiny;      // bins in (binx outer, biny inner) order, :949-951
    const int x0 = bx * a.stridex, y0 = by * a.stridey;
    const int P = a.stridex * a.stridey;
    const float* r = a.resp + (size_t)img * a.rows * a.cols;
    const size_t obase = ((size_t)img * nbins + bin) * a.per;
    // ---- pass 1: per-lane top 3
    unsigned long long k0 = 0, k1 = 0, k2 = 0;             // k0 >= k1 >= k2
    {
        // lanes walk the bin in MEMORY order (x fastest: coalesced rows); the key carries the
        // reference's push position pos = xo * stridey + yo (x outer, y inner, :953-955)
        int yo = lane / a.stridex, xo = lane % a.stridex;
        for (int idx = lane; idx < P; idx += 64) {
            const int x = x0 + xo, y = y0 + yo;
            const int pos = xo * a.stridey + yo;
            if (x < a.cols && y < a.rows) {
                const float v = fabsf(r[(size_t)y * a.cols + x]);
                if (!(fabsf(v - 0.f) <= 1e-6f * fabsf(v))) {   // isEqual(response, .0f), src/misc.cpp:10-14
                    const unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | (uint32_t)(0xffffffffu - (uint32_t)pos);
                    if (key > k2) {
                        if (key > k1) { k2 = k1; if (key > k0) { k1 = k0; k0 = key; } else k1 = key; }
                        else k2 = key;
                    }
                }
            }
            xo += 64;
            while (xo >= a.stridex) { xo -= a.stridex; ++yo; }
        }
    }
    // ---- merge: `per` rounds over the lanes' heads
    const unsigned long long third = k2;                   // what this lane might be hiding behind
    unsigned long long last = 0;
    int n = 0;
    for (int round = 0; round < a.per; ++round) {
        const unsigned long long best = wave_max_u64(k0);
        if (best == 0) break;
        if (k0 == best) { k0 = k1; k1 = k2; k2 = 0; }      // keys are unique: exactly one lane pops
        last = best;
        if (lane == 0) {
            const int pos = (int)(0xffffffffu - (uint32_t)best);
            a.tmp_kp[obase + n] = make_float2((float)(x0 + pos / a.stridey), (float)(y0 + pos % a.stridey));
            a.tmp_resp[obase + n] = __uint_as_float((uint32_t)(best >> 32));
        }
        ++n;
    }
    // exact iff no lane used up all three of its keys while more picks could lie below them
    const bool suspicious = (n == a.per) ? (third > last) : (third != 0 && k0 == 0 && n < a.per);
    if (__any(suspicious)) {
        // ---- exact path: `per` rounds of "largest key below the previous pick"
        unsigned long long prev = ~0ull;
        n = 0;
        for (int round = 0; round < a.per; ++round) {
            unsigned long long best = 0;
            int yo = lane / a.stridex, xo = lane % a.stridex;
            for (int idx = lane; idx < P; idx += 64) {
                const int x = x0 + xo, y = y0 + yo;
                const int pos = xo * a.stridey + yo;
                if (x < a.cols && y < a.rows) {
                    const float v = fabsf(r[(size_t)y * a.cols + x]);
                    if (!(fabsf(v - 0.f) <= 1e-6f * fabsf(v))) {
                        const unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | (uint32_t)(0xffffffffu - (uint32_t)pos);
                        if (key < prev && key > best) best = key;
                    }
                }
                xo += 64;
                while (xo >= a.stridex) { xo -= a.stridex; ++yo; }
            }
            best = wave_max_u64(best);
            if (best == 0) break;
            prev = best;
            if (lane == 0) {
                const int pos = (int)(0xffffffffu - (uint32_t)best);
                a.tmp_kp[obase + n] = make_float2((float)(x0 + pos / a.stridey), (float)(y0 + pos % a.stridey));
                a.tmp_resp[obase + n] = __uint_as_float((uint32_t)(best >> 32));
            }
            ++n;
        }
    }
    if (lane == 0) a.cnt[(size_t)img * nbins + bin] = n;
}

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
    dim3 grid((cols + HT_X - 1) / HT_X, (rows + HT_Y - 1) / HT_Y, n_img);
    hipLaunchKernelGGL(harris_response_kernel, grid, dim3(HT_THREADS), 0, s, images, rows, cols, k, resp);
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}

// kp_out: [n_img][kp_stride] float2, n_out: [n_img]; tmp_*: scratch sized n_img*nbins*per, cnt n_img*nbins.
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
    if ((r = launch_harris_response(c->stream, dimg, 1, rows, cols, k, dr)) < 0) return r;
    if ((r = launch_harris_bins(c->stream, dr, 1, rows, cols, n_features, nbinx, nbiny, dtk, dtr, dcnt, dko, dro,
                                dcnt + nbins, (int)slots, slots)) < 0) return r;
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
