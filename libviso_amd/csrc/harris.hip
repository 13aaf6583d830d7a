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
#define HT_Y 32
#define HT_THREADS 256
#define HT_IW (HT_X + 8)   // LDS row pitch of the uint8 tile (HT_X + 6 used)
#define HT_CW (HT_X + 2)   // cov tile width (tile + 1 ring)

__device__ __forceinline__ int h_reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// One workgroup = one 64x32 tile of the response map.
//   stage 1  uint8 tile + 3-pixel halo -> LDS (BORDER tiles: BORDER_REFLECT_101 per pixel)
//   stage 2  thread = one column of the cov tile and a band of its rows: horizontal 5-tap sums (derivative and
//            smoothing) of each image row from 5 LDS bytes, vertical 5-tap sums by a sliding register window
//            -> Dx, Dy -> the three cov products -> LDS
//   stage 3  thread = one output column and a strip of 8 rows: 3x3 box sums (row-major float additions, the
//            contract shared with the oracle) over a sliding window of cov rows -> response
// BORDER tiles (touching the image edge) take BORDER_REFLECT_101 of the cov image tap by tap.
template <bool BORDER>
__device__ __forceinline__ void harris_tile(const uint8_t* __restrict__ im, int rows, int cols, int tx0, int ty0,
                                            double k, float* __restrict__ out, unsigned char (*s_img)[HT_IW],
                                            float (*s_cov)[HT_Y + 2][HT_CW]) {
    const int tid = threadIdx.x;
    for (int idx = tid; idx < (HT_Y + 6) * (HT_X + 6); idx += HT_THREADS) {
        const int ly = idx / (HT_X + 6), lx = idx - ly * (HT_X + 6);
        int gy = ty0 - 3 + ly, gx = tx0 - 3 + lx;
        if (BORDER) { gy = h_reflect101(gy, rows); gx = h_reflect101(gx, cols); }
        s_img[ly][lx] = im[(size_t)gy * cols + gx];
    }
    __syncthreads();
    {
        // column lx of the cov tile = global x tx0 - 1 + lx; rows ly = global y ty0 - 1 + ly, in 3 bands
        const int band = tid / HT_CW, lx = tid - band * HT_CW;
        if (band < 3) {
            const int r0 = band * 12, r1 = band == 2 ? HT_Y + 2 : r0 + 12;
            const float scale = (float)(1.0 / (16.0 * 3.0 * 255.0));
            int hd[5], hs[5];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const unsigned char* p = &s_img[r0 + t][lx];
                hd[t + 1] = -(int)p[0] - 2 * (int)p[1] + 2 * (int)p[3] + (int)p[4];
                hs[t + 1] = (int)p[0] + 4 * (int)p[1] + 6 * (int)p[2] + 4 * (int)p[3] + (int)p[4];
            }
            for (int ly = r0; ly < r1; ++ly) {
#pragma unroll
                for (int t = 0; t < 4; ++t) { hd[t] = hd[t + 1]; hs[t] = hs[t + 1]; }
                const unsigned char* p = &s_img[ly + 4][lx];
                hd[4] = -(int)p[0] - 2 * (int)p[1] + 2 * (int)p[3] + (int)p[4];
                hs[4] = (int)p[0] + 4 * (int)p[1] + 6 * (int)p[2] + 4 * (int)p[3] + (int)p[4];
                const int Dx = hd[0] + 4 * hd[1] + 6 * hd[2] + 4 * hd[3] + hd[4];
                const int Dy = -hs[0] - 2 * hs[1] + 2 * hs[3] + hs[4];
                const float dx = (float)Dx * scale, dy = (float)Dy * scale;
                s_cov[0][ly][lx] = dx * dx;
                s_cov[1][ly][lx] = dx * dy;
                s_cov[2][ly][lx] = dy * dy;
            }
        }
    }
    __syncthreads();
    const int ox = tid & (HT_X - 1), strip = tid >> 6;   // 4 strips of 8 rows
    const int gx = tx0 + ox;
    if (gx >= cols) return;
    if (!BORDER) {
        // cov rows oy-1, oy, oy+1 of output row oy are local rows oy, oy+1, oy+2; columns ox, ox+1, ox+2
        float ra[3][3], rb[3][3], rc[3][3];   // [row in window][column]
        const int oy0 = strip * 8;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                ra[r + 1][j] = s_cov[0][oy0 + r][ox + j];
                rb[r + 1][j] = s_cov[1][oy0 + r][ox + j];
                rc[r + 1][j] = s_cov[2][oy0 + r][ox + j];
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int oy = oy0 + i, gy = ty0 + oy;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                ra[0][j] = ra[1][j]; ra[1][j] = ra[2][j]; ra[2][j] = s_cov[0][oy + 2][ox + j];
                rb[0][j] = rb[1][j]; rb[1][j] = rb[2][j]; rb[2][j] = s_cov[1][oy + 2][ox + j];
                rc[0][j] = rc[1][j]; rc[1][j] = rc[2][j]; rc[2][j] = s_cov[2][oy + 2][ox + j];
            }
            float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int j = 0; j < 3; ++j) { a += ra[r][j]; b += rb[r][j]; c += rc[r][j]; }
            const float t1 = a * c, t2 = b * b;
            const float t3 = t1 - t2;
            const float tr = a + c;
            if (gy < rows) out[(size_t)gy * cols + gx] = (float)((double)t3 - k * (double)tr * (double)tr);
        }
    } else {
        for (int i = 0; i < 8; ++i) {
            const int oy = strip * 8 + i, gy = ty0 + oy;
            if (gy >= rows) break;
            float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
            for (int di = -1; di <= 1; ++di) {
                const int ly = h_reflect101(gy + di, rows) - (ty0 - 1);
#pragma unroll
                for (int dj = -1; dj <= 1; ++dj) {
                    const int lx = h_reflect101(gx + dj, cols) - (tx0 - 1);
                    a += s_cov[0][ly][lx];
                    b += s_cov[1][ly][lx];
                    c += s_cov[2][ly][lx];
                }
            }
            const float t1 = a * c, t2 = b * b;
            const float t3 = t1 - t2;
            const float tr = a + c;
            out[(size_t)gy * cols + gx] = (float)((double)t3 - k * (double)tr * (double)tr);
        }
    }
}

__global__ __launch_bounds__(HT_THREADS) void harris_response_kernel(const uint8_t* __restrict__ images, int rows,
                                                                     int cols, double k, float* __restrict__ resp) {
    __shared__ __attribute__((aligned(16))) unsigned char s_img[HT_Y + 6][HT_IW];
    __shared__ float s_cov[3][HT_Y + 2][HT_CW];
    const int img = blockIdx.z;
    const int tx0 = blockIdx.x * HT_X, ty0 = blockIdx.y * HT_Y;
    const uint8_t* im = images + (size_t)img * rows * cols;
    float* out = resp + (size_t)img * rows * cols;
    // interior: the tile's 3-pixel halo (image) and 1-pixel ring (cov) lie inside the image
    const bool interior = tx0 >= 3 && ty0 >= 3 && tx0 + HT_X + 3 <= cols && ty0 + HT_Y + 3 <= rows;
    if (interior) harris_tile<false>(im, rows, cols, tx0, ty0, k, out, s_img, s_cov);
    else harris_tile<true>(im, rows, cols, tx0, ty0, k, out, s_img, s_cov);
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

#define HB_LIST 256   // keys >= the last optimistic pick collected by the one-walk exact path (per wave)

__global__ __launch_bounds__(256) void harris_bins_kernel(BinArgs a) {
    __shared__ unsigned long long s_list[4][HB_LIST];
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
        // reference's push position pos = xo * stridey + yo (x outer, y inner, :953-955).
        // Four independent loads per step: the walk is latency bound (60 steps per bin otherwise).
        int yo = lane / a.stridex, xo = lane % a.stridex;
        for (int idx = lane; idx < P; idx += 4 * 64) {
            float v[4];
            int pos[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int x = x0 + xo, y = y0 + yo;
                pos[u] = xo * a.stridey + yo;
                v[u] = 0.f;   // 0 is skipped by the isEqual test below
                if (idx + 64 * u < P && x < a.cols && y < a.rows) v[u] = fabsf(r[(size_t)y * a.cols + x]);
                xo += 64;
                while (xo >= a.stridex) { xo -= a.stridex; ++yo; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!(fabsf(v[u] - 0.f) <= 1e-6f * fabsf(v[u]))) {   // isEqual(response, .0f), src/misc.cpp:10-14
                    const unsigned long long key = ((unsigned long long)__float_as_uint(v[u]) << 32) | (uint32_t)(0xffffffffu - (uint32_t)pos[u]);
                    if (key > k2) {
                        if (key > k1) { k2 = k1; if (key > k0) { k1 = k0; k0 = key; } else k1 = key; }
                        else k2 = key;
                    }
                }
            }
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
        // The optimistic picks are `per` real keys of the bin, so every key of the true top `per` is >= the last
        // optimistic pick: ONE more walk collects all keys >= last (a handful more than `per`) into LDS and the
        // exact top `per` is selected among them.  Bins where that does not apply (fewer than `per` picks, or an
        // implausibly long list) take the multi-pass path below.
        bool done = false;
        if (n == a.per) {
            unsigned long long* list = s_list[threadIdx.x >> 6];
            int cnt = 0;
            int yo = lane / a.stridex, xo = lane % a.stridex;
            for (int idx = lane; idx < P; idx += 64) {
                const int x = x0 + xo, y = y0 + yo;
                const int pos = xo * a.stridey + yo;
                unsigned long long key = 0;
                if (x < a.cols && y < a.rows) {
                    const float v = fabsf(r[(size_t)y * a.cols + x]);
                    if (!(fabsf(v - 0.f) <= 1e-6f * fabsf(v)))
                        key = ((unsigned long long)__float_as_uint(v) << 32) | (uint32_t)(0xffffffffu - (uint32_t)pos);
                }
                const bool take = key >= last;   // last != 0 here
                const unsigned long long m = __ballot(take);
                const int at = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (take && at < HB_LIST) list[at] = key;
                cnt += __popcll(m);
                xo += 64;
                while (xo >= a.stridex) { xo -= a.stridex; ++yo; }
            }
            __builtin_amdgcn_wave_barrier();
            if (cnt <= HB_LIST) {
                unsigned long long mine[HB_LIST / 64];
#pragma unroll
                for (int u = 0; u < HB_LIST / 64; ++u) mine[u] = (lane + 64 * u < cnt) ? list[lane + 64 * u] : 0ull;
                n = 0;
                for (int round = 0; round < a.per; ++round) {
                    unsigned long long loc = mine[0];
#pragma unroll
                    for (int u = 1; u < HB_LIST / 64; ++u) loc = mine[u] > loc ? mine[u] : loc;
                    const unsigned long long best = wave_max_u64(loc);
                    if (best == 0) break;
#pragma unroll
                    for (int u = 0; u < HB_LIST / 64; ++u) if (mine[u] == best) mine[u] = 0;   // keys are unique
                    if (lane == 0) {
                        const int pos = (int)(0xffffffffu - (uint32_t)best);
                        a.tmp_kp[obase + n] = make_float2((float)(x0 + pos / a.stridey), (float)(y0 + pos % a.stridey));
                        a.tmp_resp[obase + n] = __uint_as_float((uint32_t)(best >> 32));
                    }
                    ++n;
                }
                done = true;
            }
        }
        if (!done) {
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
