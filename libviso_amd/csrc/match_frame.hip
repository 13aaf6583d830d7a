// match_frame.hip -- ONE launch for the matcher problems of a single frame (the plain family's stereo call with its two temporal
// problems riding along, csrc/plain.hip): workgroups [0, n_stereo) run match_stereo_kernel's body on the stereo problem,
// the rest match_union8_kernel's body on the temporal problems.  In a batch the two kernels are a millisecond of work each
// and run one after the other; for ONE frame each is a single wave of workgroups on a tenth of the GPU (22 and 20 us:
// a workgroup's own latency), and running them side by side takes the longer of the two instead of the sum off the chain
// the caller waits for.  The bodies ARE the kernels' (this file includes their sources with the signature macros set):
// same code, same results; only the block numbering comes from an argument.
#include "common.h"
#include "match_dev.h"

#ifdef VISO_DEBUG_VARIANTS   // (tools/experiments/frame_phases.py) time stamps of the phases of one interior tile of the first temporal problem, wave 0
__device__ unsigned long long viso_dbg_frame_uclk[8];
#define MU_CLK(I) do { if (mu_vblock == 1 + 8 * 12 && threadIdx.x == 0) viso_dbg_frame_uclk[I] = wall_clock64(); } while (0)
#endif
#define MU_KERNEL_SIG static __device__ __forceinline__ void match_union8_part(const BatchMatchArgs& a, const int mu_vblock)
#define MU_BLOCK mu_vblock
#define MU_NO_LAUNCHER
#include "match_union8.hip"

#ifdef VISO_DEBUG_VARIANTS   // timing aid (tools/experiments/frame_phases.py): 100 MHz time stamps of the phases of the stereo part's first tile
__device__ unsigned long long viso_dbg_frame_clk[16];
__device__ unsigned long long viso_dbg_frame_blk[2][256];   // start / end of every workgroup of the last launch
extern "C" int viso_debug_frame_uclocks(unsigned long long* out8) {
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(viso_dbg_frame_uclk), sizeof(unsigned long long) * 8, 0, hipMemcpyDeviceToHost) == hipSuccess ? VISO_OK : VISO_ERR_HIP;
}
extern "C" int viso_debug_frame_blocks(unsigned long long* out512) {
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out512, HIP_SYMBOL(viso_dbg_frame_blk), sizeof(unsigned long long) * 512, 0, hipMemcpyDeviceToHost) == hipSuccess ? VISO_OK : VISO_ERR_HIP;
}
extern "C" int viso_debug_frame_clocks(unsigned long long* out16) {
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(viso_dbg_frame_clk), sizeof(unsigned long long) * 16, 0, hipMemcpyDeviceToHost) == hipSuccess ? VISO_OK : VISO_ERR_HIP;
}
#define ST_CLK(I) do { if (st_vblock == 0 && threadIdx.x == 0) viso_dbg_frame_clk[I] = wall_clock64(); } while (0)
#endif
#define ST_KERNEL_SIG static __device__ __forceinline__ void match_stereo_part(const BatchMatchArgs& a, const int st_vblock)
#define ST_BLOCK st_vblock
#define ST_NO_LAUNCHER
#include "match_stereo.hip"

// at FIRST: match_union8's body reads its planes' shift from the kernel-argument segment at offsetof(BatchMatchArgs, r8s)
__global__ __launch_bounds__(256) void match_frame_kernel(BatchMatchArgs at, BatchMatchArgs as, int n_stereo) {
#ifdef VISO_DEBUG_VARIANTS
    if (threadIdx.x == 0 && blockIdx.x < 256) { viso_dbg_frame_blk[0][blockIdx.x] = wall_clock64(); viso_dbg_frame_blk[1][blockIdx.x] = 0; }
#endif
    if ((int)blockIdx.x < n_stereo) match_stereo_part(as, (int)blockIdx.x);
    else match_union8_part(at, (int)blockIdx.x - n_stereo);
#ifdef VISO_DEBUG_VARIANTS
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) atomicMax(&viso_dbg_frame_blk[1][blockIdx.x], (unsigned long long)wall_clock64());
    if (blockIdx.x == 0 && threadIdx.x == 0) viso_dbg_frame_clk[8] = wall_clock64();
    if ((int)blockIdx.x == n_stereo && threadIdx.x == 0) viso_dbg_frame_clk[9] = wall_clock64();   // the first union8 tile's end
#endif
}

// at / as: the temporal and the stereo launch's arguments as launch_match_batch prepares them (as.bpp is set here, like
// launch_match_stereo does); blocks_t: the temporal grid
int launch_match_frame(hipStream_t s, const BatchMatchArgs& at, const BatchMatchArgs& as64, long long blocks_t, int cap_max) {
    BatchMatchArgs as = as64;
    const int tiles = (cap_max + ST_QPW - 1) / ST_QPW;
    as.bpp = (tiles + ST_WAVES - 1) / ST_WAVES;
    const int groups = (as.n_probs + 7) / 8;
    long long blocks_s = (long long)groups * 8 * as.bpp;
    if (as.gs == 3) blocks_s = (long long)((groups + 2) / 3) * as.gc * 8 * as.bpp;
    if (blocks_s + blocks_t > 0x7fffffffLL) { viso_set_error("matcher grid too large"); return VISO_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(match_frame_kernel, dim3((unsigned)(blocks_s + blocks_t)), dim3(256), 0, s, at, as, (int)blocks_s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { viso_set_error("match_frame_kernel launch: %s", hipGetErrorString(e)); return VISO_ERR_HIP; }
    return VISO_OK;
}
