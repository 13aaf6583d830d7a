// extract.hip — descriptor extraction on device: MyFeatureExtractor::computeImpl
// (reference src/viso.cpp:1004-1024).  cv::Sobel(image, CV_32F, dx=1, dy=0,
// ksize=3, BORDER_REFLECT_101) sampled on a (2r+1)^2 window around each
// keypoint; no Sobel image is materialised — every descriptor element
// recomputes its 3x3 stencil from the uint8 image (L2-resident, 467 KB at
// 1241x376).
#include "common.h"
#include "match_dev.h"

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

__global__ __launch_bounds__(256) void extract_desc_kernel(const uint8_t* __restrict__ img, int rows,
                                                           int cols, const float2* __restrict__ kp,
                                                           int n, int radius, float* __restrict__ desc) {
    const int side = 2 * radius + 1, dlen = side * side;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)n * dlen) return;
    const int k = (int)(gid / dlen), col = (int)(gid % dlen);
    const float2 p = kp[k];
    // Point2i p = kp.pt (:1013): saturate_cast<int>(float) rounds to nearest even
    const int px = (int)rintf(p.x), py = (int)rintf(p.y);
    const int y = py + col / side - radius, x = px + col % side - radius;
    float val = 0.f;
    if (y > 0 && y < rows && x > 0 && x < cols) {   // strict > 0 (:1018)
        const int ym = reflect101(y - 1, rows), yp = reflect101(y + 1, rows);
        const int xm = reflect101(x - 1, cols), xp = reflect101(x + 1, cols);
        const uint8_t* r0 = img + (size_t)ym * cols;
        const uint8_t* r1 = img + (size_t)y * cols;
        const uint8_t* r2 = img + (size_t)yp * cols;
        const int v = ((int)r0[xp] - (int)r0[xm]) + 2 * ((int)r1[xp] - (int)r1[xm]) + ((int)r2[xp] - (int)r2[xm]);
        val = (float)v;
    }
    desc[gid] = val;
}

int launch_extract(hipStream_t s, const uint8_t* img, int rows, int cols, const float2* kp, int n,
                   int radius, float* desc) {
    const int side = 2 * radius + 1;
    const long long total = (long long)n * side * side;
    if (total <= 0) return VISO_OK;
    hipLaunchKernelGGL(extract_desc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, img,
                       rows, cols, kp, n, radius, desc);
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}

extern "C" int viso_extract_descriptors(const uint8_t* img, int rows, int cols, const float* kp, int n,
                                        int radius, float* desc) {
    if (!img || rows <= 0 || cols <= 0 || n < 0 || radius < 0 || (n && (!kp || !desc))) {
        viso_set_error("viso_extract_descriptors: bad argument");
        return VISO_ERR_ARG;
    }
    if (n == 0) return VISO_OK;
    PlainLock lk;
    viso_ctx* c = viso_default_ctx();
    if (!c) return VISO_ERR_HIP;
    const int side = 2 * radius + 1;
    uint8_t* dimg; float2* dkp; float* dd;
    int r;
    if ((r = ctx_scratch(c, 0, (size_t)rows * cols, (void**)&dimg)) < 0) return r;
    if ((r = ctx_scratch(c, 1, sizeof(float2) * (size_t)n, (void**)&dkp)) < 0) return r;
    if ((r = ctx_scratch(c, 2, sizeof(float) * (size_t)n * side * side, (void**)&dd)) < 0) return r;
    HIP_TRY(hipMemcpyAsync(dimg, img, (size_t)rows * cols, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(dkp, kp, sizeof(float2) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    if ((r = launch_extract(c->stream, dimg, rows, cols, dkp, n, radius, dd)) < 0) return r;
    HIP_TRY(hipMemcpyAsync(desc, dd, sizeof(float) * (size_t)n * side * side, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VISO_OK;
}

// ---------------------------------------------------------------------------
// Fused extract + pack for the batch pipeline: the
// 11x11 Sobel-x window of every keypoint goes straight from the uint8 image
// into the matcher's biased-u16 row (column-bucket order); the N x 121 float
// descriptor matrix of the reference (src/viso.cpp:1008) is never materialised.
// Sobel of uint8 is an integer in [-1020, 1020], so the u16 path is always exact.
// One wave = one keypoint at a time: lane l produces elements 2l and 2l+1.
#define VISO_EXT_KPW 4   // keypoints per wave
#define VISO_EXT_WIN 13  // 11x11 descriptor window + the 3x3 Sobel's ring

// One wave = VISO_EXT_KPW keypoints: their 13x13 uint8 windows are fetched first (3 bytes per lane and keypoint, all
// in flight together, BORDER_REFLECT_101 applied to the coordinates), staged in LDS, then lane l produces elements
// 2l and 2l+1 of every descriptor from LDS.
__global__ __launch_bounds__(256) void extract_pack_kernel(const ImageView* __restrict__ imgs, int n_img, int cap,
                                                           const uint8_t* __restrict__ images, int rows, int cols, int extras, int r8s, int* __restrict__ r8cnt, unsigned r8m) {
    __shared__ unsigned char s_win[4][VISO_EXT_KPW][VISO_EXT_WIN * VISO_EXT_WIN + 7];
    typedef const __attribute__((address_space(1))) uint8_t* gbyte_t;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long wave = (long long)blockIdx.x * 4 + wv;
    const long long k0 = wave * VISO_EXT_KPW;
    if (k0 >= (long long)n_img * cap) return;
    const int img = (int)(k0 / cap);           // cap is a multiple of VISO_EXT_KPW
    const ImageView I = imgs[img];
    const int n = *I.n;
    const int j0 = (int)(k0 % cap);
    if (j0 >= n) return;
    const int nk = min(VISO_EXT_KPW, n - j0);
    const gbyte_t im = (gbyte_t)(images + (size_t)img * rows * cols);
    // keypoint of lane k (k < nk), broadcast below
    float2 pl = make_float2(0.f, 0.f);
    if (lane < nk) pl = I.skp[j0 + lane];
    int px[VISO_EXT_KPW], py[VISO_EXT_KPW];
#pragma unroll
    for (int k = 0; k < VISO_EXT_KPW; ++k) {
        px[k] = (int)rintf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(pl.x), k)));   // Point2i p = kp.pt, src/viso.cpp:1013
        py[k] = (int)rintf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(pl.y), k)));
    }
    // window pixel t of keypoint k: image pixel (py - 6 + t / 13, px - 6 + t % 13), coordinates reflected
    unsigned char w[VISO_EXT_KPW][3];
#pragma unroll
    for (int k = 0; k < VISO_EXT_KPW; ++k) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int t = lane + 64 * u;
            w[k][u] = 0;
            if (k < nk && t < VISO_EXT_WIN * VISO_EXT_WIN) {
                const int wy = t / VISO_EXT_WIN, wx = t - wy * VISO_EXT_WIN;
                const int ry = py[k] - 6 + wy, rx = px[k] - 6 + wx;
                if (px[k] >= 6 && py[k] >= 6 && px[k] + 6 < cols && py[k] + 6 < rows) {   // wave uniform: the whole window inside the image
                    w[k][u] = im[(uint32_t)(ry * cols + rx)];
                } else if (ry >= 0 && ry <= rows && rx >= 0 && rx <= cols) {
                    // only neighbours of in-image centres (0 < y < rows, 0 < x < cols) are ever read: rows 0..rows,
                    // columns 0..cols, of which only `rows` / `cols` themselves need the reflection
                    w[k][u] = im[(size_t)reflect101(ry, rows) * cols + reflect101(rx, cols)];
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < VISO_EXT_KPW; ++k)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int t = lane + 64 * u;
            if (t < VISO_EXT_WIN * VISO_EXT_WIN) s_win[wv][k][t] = w[k][u];
        }
    __builtin_amdgcn_wave_barrier();
    const bool r8on = r8cnt && ((unsigned)wave & r8m) == 0;   // uniform: ~256 waves of the launch
    R8Count r8c = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < VISO_EXT_KPW; ++k) {
        if (k >= nk) break;                       // wave uniform
        const unsigned char* win = s_win[wv][k];
        uint32_t packed = 0;
        int vsum = 0, v2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 2 * lane + h;
            int v = 0;
            if (c < 121) {
                const int ey = c / 11, ex = c - ey * 11;
                const int y = py[k] + ey - 5, x = px[k] + ex - 5;
                if (y > 0 && y < rows && x > 0 && x < cols) {   // strict > 0, src/viso.cpp:1018
                    // Sobel centre = window (ey + 1, ex + 1); rows y-1, y, y+1 = window rows ey, ey+1, ey+2
                    const unsigned char* r0 = win + ey * VISO_EXT_WIN + ex;
                    const unsigned char* r1 = r0 + VISO_EXT_WIN;
                    const unsigned char* r2 = r1 + VISO_EXT_WIN;
                    v = ((int)r0[2] - (int)r0[0]) + 2 * ((int)r1[2] - (int)r1[0]) + ((int)r2[2] - (int)r2[0]);
                }
            }
            packed |= ((uint32_t)(v + VISO_BIAS) & 0xffffu) << (16 * h);
            vsum += v;
            v2[h] = v;
        }
        reinterpret_cast<uint32_t*>(I.rows + (size_t)(j0 + k) * VISO_ROW)[lane] = packed;
        if (extras & VISO_PACK_SUMS) {   // uniform; ImageView::sums (match_prune_kernel)
            const uint2 bs = pack_block_sums(vsum);
            if (lane == 0) I.sums[j0 + k] = bs;
        }
        if (extras & VISO_PACK_ROWS8) { store_row8(I.rows8, (size_t)(j0 + k), lane, v2[0], v2[1], r8s); if (r8on) r8_count(r8c, v2[0], v2[1]); }   // uniform; match_union8_kernel
    }
    if (r8on) r8_flush(r8c, r8cnt, lane);
}

int launch_extract_pack(hipStream_t s, const ImageView* imgs_dev, int n_img, int cap, const uint8_t* images,
                        int rows, int cols, int extras, int r8s, int* r8cnt) {
    const int capp = (cap + VISO_EXT_KPW - 1) / VISO_EXT_KPW * VISO_EXT_KPW;
    const long long waves = (long long)n_img * capp / VISO_EXT_KPW;
    if (waves == 0) return VISO_OK;
    hipLaunchKernelGGL(extract_pack_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, imgs_dev, n_img, capp,
                       images, rows, cols, extras, r8s, (extras & VISO_PACK_ROWS8) ? r8cnt : nullptr, r8_mask(waves));
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}
