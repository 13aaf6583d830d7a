// solver.hip — RANSAC + Gauss-Newton stereo reprojection pose solver kernels
// (reference src/viso.cpp:1543-1580 ransac_minimize_reproj, :1583-1623
// minimize_reproj, :1509-1537 get_inliers).  Latency-bound fp64 work: all hypotheses of all
// frames run concurrently (one lane per 3-point hypothesis, which draws its own
// sample triple in O(1) first; the few that
// need more than VISO_GN_SPLIT iterations continue one wave each), support sets
// are counted per frame x 10 hypotheses, and the all-inlier refit builds
// J^T J / J^T r per iteration as a workgroup reduction with the 6x6 LU solve on
// one lane — the whole loop stays on the device.  No kernel here uses scratch
// memory (check with -Rpass-analysis=kernel-resource-usage after any change).
#include "solver_dev.h"

#include <string.h>
#include <vector>

struct SolverArgs {
    const SolverItem* items;
    int n_items;
    int iters;
    unsigned long long seed;
    SolverParamsDev sp;
    int split;    // iterations done by ransac_hyp_kernel (VISO_GN_SPLIT)
    int* queue;   // [0] = number of undecided hypotheses, ZERO when the chain starts (zeroed at allocation, and again by
                  //       ransac_rot_kernel -- the kernel behind the list's only reader -- after saving it to [1]);
                  // [1] = the last chain's count (diagnostics); [2..] = item * iters + h of each undecided hypothesis (any order)
    RefitMirror mir;   // plain family, n_items == 1: the refit kernel leaves the results in pinned host memory and signals (sig.flag == null: no)
    const OutArgs* ride; int ride_blocks;   // plain family: a copy-out of the kernel BEFORE the chain rides in ransac_coop_kernel's launch (common.h, OutArgs):
    int coop_blocks;                         // ... as the workgroups behind the kernel's own coop_blocks
};

// ---- stage 0: the sample triples -- no kernel of their own any more --------------------------------------------------
// Rounds 1-5 drew every triple with the reference's algorithm S (src/viso.cpp:88-107: one uniform draw per candidate
// index) in ransac_sample_kernel, a wave per hypothesis testing 64 candidates per step: 16.9 M vector instructions per
// 25 600 triples, the second largest kernel of the chain, 6 us of every per-call frame.  The stream is this build's
// definition (the reference's is random_device, Q9), so the triple is now defined in O(1) -- three splitmix64 draws
// through Floyd's subset sampling, viso_sample3 in solver_dev.h: the same distribution, uniform 3-subsets in ascending
// order -- and every lane of ransac_hyp_kernel draws its own in its prologue.

// ---- stage 1: one lane per (frame, hypothesis): 3-point GN from zero -------
// Almost every hypothesis converges (or turns singular) within a few iterations; the ~1-2 % that do not run all 100
// (src/viso.cpp:1622) and would keep their whole wave (and the kernel: a 0.7 ms serial fp64 chain) waiting.  Stage 1
// therefore stops after VISO_GN_SPLIT iterations and marks the unfinished ones (ok_h = 2, tr_h = state so far);
// ransac_coop_kernel continues each of them with a whole wave per hypothesis.
#define VISO_GN_SPLIT 10

__global__ __launch_bounds__(256) void ransac_hyp_kernel(SolverArgs a) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= a.n_items * a.iters) return;
    // a few hundred waves on a serial fp64 chain, usually beside another batch's matcher kernels: win the
    // instruction-issue arbitration on the SIMD (the chain's latency is what the batch waits for)
    __builtin_amdgcn_s_setprio(3);
    const int item = gid / a.iters, h = gid % a.iters;
    const SolverItem S = a.items[item];
    const int m = *S.m_ptr;
    int ok = 0;
    double tr[6] = {0, 0, 0, 0, 0, 0};        // "start search from 0", :1557
    int sample[3] = {0, 0, 0};
    if (S.samples) {   // explicit triples: copied, so that the later stages read one place
#pragma unroll
        for (int k = 0; k < 3; ++k) sample[k] = S.samples[3 * h + k];
    } else {
        viso_sample3(a.seed, S.frame, h, m, sample);   // randomsample(3, m, .), :1558 (zeros for m < 3)
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) S.samp_h[3 * h + k] = sample[k];   // ransac_coop_kernel, viso_batch_get_hypotheses read them here
    if (m >= 3) {
        bool valid = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) valid = valid && sample[k] >= 0 && sample[k] < m;
        if (valid) ok = gn_serial<3>(S.X, S.obs, S.ld, sample, tr, a.sp, 0, a.split);
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) S.tr_h[6 * h + j] = tr[j];
    S.ok_h[h] = ok;
    if (ok == 2) a.queue[2 + atomicAdd(&a.queue[0], 1)] = gid;   // ransac_coop_kernel continues it
}

// ---- stage 1b: one WAVE per unfinished hypothesis: iterations VISO_GN_SPLIT..99, same arithmetic ------------
// A single wave on a serial fp64 chain issues one instruction every 4+ cycles, so its time per iteration is its
// instruction count: the work is spread over the lanes instead of being repeated in every lane.
//   Jacobian   lane (p, j), p < 3 points, j < 6 parameters (lanes 0..17): column j of point p's rows and, for
//              j == 0, the residuals, exactly as accumulate_point computes them -> LDS JE[p][row 0..3][col 0..6]
//              (row 3 repeats row 1, src/viso.cpp:1479,1481; col 6 = weighted residual)
//   J^T J, J^T r   matrix lane (r, c) = lane r * 8 + c, r < 6, c < 7: ONE entry of the augmented system [A | b],
//              summed over the 12 rows in the reference's order (A[r][c] and A[c][r] are the same products in the
//              same order: symmetric bit for bit, what symmetrize() copies in gn_serial)
//   6x6 LU     cv::solve(DECOMP_LU) = lu_solve6 of solver_dev.h with one entry per lane: per pivot column the
//              candidates come to every lane as scalars (v_readlane: first strict maximum of |.|, singular iff
//              < DBL_EPSILON), rows i and k change places through one cross-lane move, every lane below / right of
//              the pivot does its own  a += (a_ri * d) * a_ic  (d = -1 / pivot; a_ri by ds_swizzle inside the
//              lane's 8-group, a_ic by ds_bpermute), lane (i, i) keeps -d
//   back substitution   the 21 + 6 entries it needs come back as scalars; s -= A[i][c] * x[c] in the reference's
//              order c = i+1..5, x[i] = s * (1 / pivot): computed uniformly by every lane (the step, the
//              convergence test and tr stay wave uniform)
// Values are bit-identical to gn_serial's (tests/test_gpu_solver_edges.py runs every hypothesis both ways): only
// who computes them changes.  ~60 VGPRs instead of 205, about a third of the instructions per iteration.
__device__ __forceinline__ double rdlane(double v, int src_lane) {   // src_lane: compile-time constant
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}
template <int I>
__device__ __forceinline__ double swz_bcast8(double v) {   // element I of the lane's aligned group of 8 lanes
    constexpr int pat = (I << 5) | 0x18;                   // bitmask mode: lane' = (lane & 0x18) | I  (per 32 lanes)
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), pat);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), pat);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bperm(double v, int src_lane) {
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// one pivot step of the lane-distributed LU; returns false when the system is singular (uniform)
template <int I>
__device__ __forceinline__ bool lu_lane_step(double& a, int lane, int row, int col) {
    double cv[6];
#pragma unroll
    for (int j = I; j < 6; ++j) cv[j] = rdlane(a, j * 8 + I);      // column I, rows I..5: scalars
    int k = I;
    double best = fabs(cv[I]), piv = cv[I];
#pragma unroll
    for (int j = I + 1; j < 6; ++j) {
        const double v = fabs(cv[j]);
        if (v > best) { best = v; k = j; piv = cv[j]; }            // first strict maximum
    }
    if (best < DBL_EPSILON) return false;
    k = __builtin_amdgcn_readfirstlane(k);
    if (k != I) {                                                  // uniform: rows I and k change places
        const int d8 = (k - I) * 8;
        const int src = lane + (row == I ? d8 : row == k ? -d8 : 0);
        a = bperm(a, src);
    }
    const double d = -1 / piv;
    const double a_ri = swz_bcast8<I>(a);                          // A[row][I]
    const double a_ic = bperm(a, I * 8 + col);                     // A[I][col]
    const double alpha = a_ri * d;
    const double upd = a + alpha * a_ic;
    if (row > I && row < 6 && col > I && col < 7) a = upd;
    if (lane == I * 8 + I) a = -d;
    return true;
}

__global__ __launch_bounds__(256) void ransac_coop_kernel(SolverArgs a) {
    __shared__ double s_JE[4][3][4][8];      // [wave][point][row][col 0..5 = J, 6 = residual, 7 = pad]
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __builtin_amdgcn_s_setprio(3);                   // see ransac_hyp_kernel
    // a SMALL grid walks the list of undecided hypotheses (a launch with one wave per hypothesis would push
    // thousands of workgroups, nearly all of which leave at once, through a GPU that is busy with another batch's
    // matcher)
    // riders (plain family): the copy-out of the join kernel's results.  Here rather than in ransac_hyp_kernel's launch: that
    // kernel is 9 us of one wave's work, the copy-out 11 -- it set the kernel's duration, in the chain the caller waits for
    if (a.ride && (int)blockIdx.x >= a.coop_blocks) { plain_out_blocks(*a.ride, blockIdx.x - (unsigned)a.coop_blocks); return; }
    const int n_undecided = a.queue[0];
    const int row = lane >> 3, col = lane & 7;       // this lane's entry of [A | b] (row < 6, col < 7)
    const int mrow = min(row, 5), mcol = min(col, 6);
    for (int qi = (int)blockIdx.x * 4 + wv; qi < n_undecided; qi += a.coop_blocks * 4) {
    const int gid = a.queue[2 + qi];
    const int item = gid / a.iters, h = gid % a.iters;
    const SolverItem S = a.items[item];
    const int sample[3] = {S.samp_h[3 * h], S.samp_h[3 * h + 1], S.samp_h[3 * h + 2]};   // ransac_hyp_kernel left them there
    const SolverParamsDev& sp = a.sp;
    const double* X = S.X;
    const double* obs = S.obs;
    const int ld = S.ld;
    double tr[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) tr[j] = S.tr_h[6 * h + j];
    // this lane's Jacobian job: point p (position in the sample), parameter j
    const int p = min(lane / 6, 2), j = lane % 6;
    const int ai = p == 0 ? sample[0] : p == 1 ? sample[1] : sample[2];
    const double X1p = X[0 * ld + ai], Y1p = X[1 * ld + ai], Z1p = X[2 * ld + ai];
    const double o0 = obs[0 * ld + ai], o1 = obs[1 * ld + ai], o2 = obs[2 * ld + ai], o3 = obs[3 * ld + ai];
    const double weight = 1.0 / (fabs(obs[0 * ld + p] - sp.cu) / fabs(sp.cu) + 0.05);   // Q6: position, not index
    int ok = 0;
    for (int it = a.split; it < 100; ++it) {
        // rotation: one sincos per lane (lanes 0..2 matter), the six values come back as scalars
        double sv, cv;
        const int l3 = lane % 3;
        sincos(l3 == 0 ? tr[0] : l3 == 1 ? tr[1] : tr[2], &sv, &cv);
        const double sx = rdlane(sv, 0), cx = rdlane(cv, 0), sy = rdlane(sv, 1), cy = rdlane(cv, 1), sz = rdlane(sv, 2), cz = rdlane(cv, 2);
        RotDev R;
        rot_from_sincos(sx, cx, sy, cy, sz, cz, tr, R);
        // column j of point p (the switch of accumulate_point, evaluated for all three rotation parameters and
        // selected: the selected value is the one the switch would have computed)
        double pred[4], X1c, Y1c, Z1c, X2c;
        predict_point(R, sp, X1p, Y1p, Z1p, pred, X1c, Y1c, Z1c, X2c);
        const double wf = weight * sp.f, zz = Z1c * Z1c;   // divisions as the reference has them, src/viso.cpp:1478-1481
        const double y0 = R.rdrx10 * X1p + R.rdrx11 * Y1p + R.rdrx12 * Z1p, z0 = R.rdrx20 * X1p + R.rdrx21 * Y1p + R.rdrx22 * Z1p;
        const double x1 = R.rdry00 * X1p + R.rdry01 * Y1p + R.rdry02 * Z1p, y1 = R.rdry10 * X1p + R.rdry11 * Y1p + R.rdry12 * Z1p,
                     z1 = R.rdry20 * X1p + R.rdry21 * Y1p + R.rdry22 * Z1p;
        const double x2 = R.rdrz00 * X1p + R.rdrz01 * Y1p, y2 = R.rdrz10 * X1p + R.rdrz11 * Y1p, z2 = R.rdrz20 * X1p + R.rdrz21 * Y1p;
        const double X1cd = j == 0 ? 0.0 : j == 1 ? x1 : j == 2 ? x2 : j == 3 ? 1.0 : 0.0;
        const double Y1cd = j == 0 ? y0 : j == 1 ? y1 : j == 2 ? y2 : j == 4 ? 1.0 : 0.0;
        const double Z1cd = j == 0 ? z0 : j == 1 ? z1 : j == 2 ? z2 : j == 5 ? 1.0 : 0.0;
        if (lane < 18) {
            const double j0 = wf * (X1cd * Z1c - X1c * Z1cd) / zz;
            const double j1 = wf * (Y1cd * Z1c - Y1c * Z1cd) / zz;
            const double j2 = wf * (X1cd * Z1c - X2c * Z1cd) / zz;
            s_JE[wv][p][0][j] = j0;
            s_JE[wv][p][1][j] = j1;
            s_JE[wv][p][2][j] = j2;
            s_JE[wv][p][3][j] = j1;
            if (j == 0) {
                s_JE[wv][p][0][6] = weight * (o0 - pred[0]);
                s_JE[wv][p][1][6] = weight * (o1 - pred[1]);
                s_JE[wv][p][2][6] = weight * (o2 - pred[2]);
                s_JE[wv][p][3][6] = weight * (o3 - pred[3]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // this lane's entry of [J^T J | J^T r], summed over points and rows in the reference's order
        double acc = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc += s_JE[wv][i][r][mrow] * s_JE[wv][i][r][mcol];
        __builtin_amdgcn_wave_barrier();   // s_JE is rewritten next iteration
        // cv::solve(DECOMP_LU), one entry per lane (src/viso.cpp:1602-1606)
        bool regular = lu_lane_step<0>(acc, lane, row, col);
        regular = regular && lu_lane_step<1>(acc, lane, row, col);
        regular = regular && lu_lane_step<2>(acc, lane, row, col);
        regular = regular && lu_lane_step<3>(acc, lane, row, col);
        regular = regular && lu_lane_step<4>(acc, lane, row, col);
        regular = regular && lu_lane_step<5>(acc, lane, row, col);
        if (!regular) { ok = 0; break; }
        double B[6];
#pragma unroll
        for (int i = 5; i >= 0; --i) {
            double sacc = rdlane(acc, i * 8 + 6);
#pragma unroll
            for (int c = i + 1; c < 6; ++c) sacc -= rdlane(acc, i * 8 + c) * B[c];
            B[i] = sacc * rdlane(acc, i * 8 + i);
        }
        bool converged = true;
#pragma unroll
        for (int jj = 0; jj < 6; ++jj)
            if (B[jj] > sp.thresh) converged = false;  // Q7
        if (converged) { ok = 1; break; }
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) tr[jj] = tr[jj] + B[jj];
    }
    if (lane == 0) {
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) S.tr_h[6 * h + jj] = tr[jj];
        S.ok_h[h] = ok;
    }
    __builtin_amdgcn_wave_barrier();   // the wave's LDS slice is reused by its next hypothesis
    }
}

// ---- stage 2: support set sizes (get_inliers per hypothesis, src/viso.cpp:1561-1562) -------------------------
// The one RANSAC kernel with real arithmetic in it (25 600 hypotheses x ~1 350 points per 512 frame pairs), on a GPU
// whose vector ALUs the matcher keeps busy.  Two kernels:
//
// ransac_rot_kernel: thread per hypothesis: make_rot once, the rotation stored as fp64 (tiers 2 / 3) and as fp32, the
// fp32 copies of hypotheses 2p and 2p + 1 INTERLEAVED ({r_j of 2p, r_j of 2p + 1}: what a packed instruction takes as
// one 64-bit operand) with the pair's share of the tier-1 bound behind them; zeroes the hypothesis' count.  A failed
// hypothesis gets the identity: the counting loop has no case for it, its count is dropped at the end.
//
// inlier_count_kernel: one-wave workgroup per (frame, 64 points), a point per lane in registers, the wave walks ALL the
// frame's hypotheses two at a time.  The pair's 28 floats are wave uniform: read with SCALAR loads (constant address
// space: nothing writes them while this kernel runs) and used as scalar operands of the packed fp32 instructions
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) — no vector register, no LDS traffic, no barrier and no per-workgroup
// prologue goes into the rotations (round 3's workgroup per (frame, 10 hypotheses) built them on ten lanes while 246
// waited, and re-derived every point's constants for each group of ten).  ballot + popcount per wave (the verdicts
// stay lane masks: the logic on them is scalar), counts packed per pair in a lane of one register, one global atomic
// per (wave, hypothesis) at the end; what tier 1 leaves undecided is queued as a bit per lane, pair and half and
// decided behind the loop, every lane with something queued taking one entry per turn.
//
// What that bought, measured (tools/alone_pmc.sh, tools/alone_ab.sh, tools/e2e_ab.sh): 41.0 M -> 19.7 M vector
// instructions per 512 pairs — and the SAME 78-80 us alone, +0.6 % end to end.  Instruction counts mislead on this
// chip: a plain v_fma_f32 / v_add_f32 / v_mul_f32 on vector registers issues every ~2.6 cycles per SIMD, a packed one
// every ~4.7 (two FMAs: 2.3 each — no faster than two plain ones), one with a scalar operand 4.2, a compare 4.4,
// v_rcp_f32 8 (tools/valu_rate.hip).  Round 3's loop was already made of the cheap kind (~170 issue cycles per 64
// points and hypothesis); this one is ~130, on 25 % fewer (wave, hypothesis) steps (64-point instead of 256-point
// granularity), and the tail of fp64 decisions (0.22 % of the tests, 19 % of the instructions) costs it 30 of the
// 80 us alone.  Kept because it is what the end-to-end step sees: less vector-ALU time under the matcher.
//
// The verdict "err2 < inlier_threshold^2" is taken in three tiers, the same verdicts bit for bit:
//   1. fp32 with a rigorous bound of its own error (below): decides every point whose err2 is further from the
//      threshold than that bound (~0.5 % of the threshold for KITTI-like numbers);
//   2. fp64 with one refined reciprocal of Z and its error band (is_inlier_pt);
//   3. the reference's expression with its three divisions.
// Tier-1 error bound, u = 2^-24.  Inputs rounded to float: relative error u each.  |R_ij| <= 1, so every camera
// coordinate (three nested FMAs over products of rounded inputs) is off by at most D = 8u (|X|+|Y|+|Z| + T), T = the
// larger max|t_i| of the PAIR's two hypotheses (a wild hypothesis loosens the bound of its pair only; round 3 took the
// largest of ten); and |Xc|, |Yc|, |Xc - base| <= N = |X|+|Y|+|Z| + T + base + D.  With
// z = 1 / |Zc| and D z <= 1/64, g = f / Zc (v_rcp_f32: 1 ulp) has relative error <= 1.1 D z + 4u, so a projection
// g * n + c is off by at most |g| (D + 1.2 N (1.1 D z + 4u)) + u (|c| + |p|) = |g| (c1 + c2 z) + u (|c| + |p|) with
// c1 = D + 4.8 u N, c2 = 1.32 D N, and a residual e = o - p by that + u (|o| + |e|).  Summed over the four residuals
// (|p_k| <= |o_k| + |e_k|): every residual is off by at most de = |g| (c1 + c2 z) + u (Cm + 2 sum|e|), Cm = |cu| + |cv| +
// 2 sum|o|, and err2 = sum e^2 by at most de (2 sum|e| + 4 de) + 4u err2.  Added to it: what the rounding of the
// REFERENCE's own fp64 evaluation can move its err2 (tier 2's band, generously: 2e-12 Cm^2 + 8e-12 err2).  The bound is
// itself evaluated in float: inflated by 1/16 and an absolute 1e-30; anything not finite is "undecided".  D and N are
// sums of a per-point and a per-pair part (two packed adds per point and pair give D, K D, N, 1.32 K D; c1 and c2 one
// FMA and one multiply more), Cm is per point.  The residuals are computed as e = fma(-n, g, o - c) (o - c rounded
// once per point): one rounding less than o - (g n + c), which is what the bound charges.
//
// What has no packed form stays per hypothesis: v_rcp_f32, the three adds of sum|e| and the two FMAs of the bound that
// take |1/Zc| and |g| (absolute values are source modifiers of the unpacked encodings only), the compares: ~57 vector
// instructions per point and PAIR, against ~46 per point and hypothesis in round 3.
typedef float inl_f2 __attribute__((ext_vector_type(2)));
#define INL_FMA(A, B, C) __builtin_elementwise_fma((A), (B), (C))
#define INL_K 2.2578125f   // 1.0625 (the bound's inflation) x 2.125 (the loop works with 2 de, inflated once more)

// layout of a frame's rotation store (SolverItem::rot, viso_rot_bytes(iters) bytes): np = ceil(iters / 2) pair blocks of
// INL_PAIR_F floats (12 x float2 rotation / translation, then the pair's four bound constants), then iters x 12 doubles
#define INL_PAIR_F 28
__host__ __device__ inline size_t rot_off_d(int iters) { return (size_t)((iters + 1) / 2) * (INL_PAIR_F * 4); }
size_t viso_rot_bytes(int iters) { return (rot_off_d(iters) + (size_t)iters * 96 + 127) & ~(size_t)127; }

__global__ __launch_bounds__(256) void ransac_rot_kernel(SolverArgs a) {
    const int np2 = ((a.iters + 1) / 2) * 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    // the list of undecided hypotheses has had its only reader (ransac_coop_kernel, the kernel in front of this one): empty
    // it for the next chain -- no launch in front of ransac_hyp_kernel just to zero a word -- and keep its length for
    // viso_batch_get_hypotheses (viso_support_sizes comes here without a list)
    if (idx == 0 && a.queue) { a.queue[1] = a.queue[0]; a.queue[0] = 0; }
    if (idx >= (long long)a.n_items * np2) return;
    const int item = (int)(idx / np2), t = (int)(idx % np2);
    const SolverItem S = a.items[item];
    float* F = reinterpret_cast<float*>(S.rot) + (size_t)(t >> 1) * INL_PAIR_F + (t & 1);
    const bool ok = t < a.iters && S.ok_h[t] != 0;
    if (t < a.iters) S.cnt_h[t] = 0;
    float tm = 0.f;
    if (!ok) {   // a failed hypothesis (or the padding of an odd count): the identity — counted like any other, dropped at the end
#pragma unroll
        for (int j = 0; j < 12; ++j) F[2 * j] = (j == 0 || j == 4 || j == 8) ? 1.f : 0.f;
    } else {
        double tr[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) tr[j] = S.tr_h[6 * t + j];
        RotDev R;
        make_rot(tr, R);
        double* o = reinterpret_cast<double*>(S.rot + rot_off_d(a.iters)) + (size_t)t * 12;
        o[0] = R.r00; o[1] = R.r01; o[2] = R.r02; o[3] = R.r10; o[4] = R.r11; o[5] = R.r12;
        o[6] = R.r20; o[7] = R.r21; o[8] = R.r22; o[9] = R.tx; o[10] = R.ty; o[11] = R.tz;
        const double v[12] = {R.r00, R.r01, R.r02, R.r10, R.r11, R.r12, R.r20, R.r21, R.r22, R.tx, R.ty, R.tz};
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const float f = (float)v[j];
            F[2 * j] = f;
            if (j >= 9) tm = fmaxf(tm, fabsf(f));
        }
    }
    // the pair's share of the tier-1 bound (inlier_count_kernel): T = the larger max|t_i| of the two (threads t, t ^ 1:
    // the same wave, the pair count per frame is even), rounded up; NaN / inf stay what they are (nothing is decided then)
    const float U = 5.9604645e-8f;
    const float T = fmaxf(tm, __shfl_xor(tm, 1)) * (1.f + 4.f * U);
    if (!(t & 1)) {
        float* P = F + 24;
        const float dT = 8.f * U * T * (1.f + 4.f * U);
        P[0] = dT;                                   // D  = dP + dT
        P[1] = dT * INL_K * (1.f + 4.f * U);         // DK = K D
        P[2] = T * (1.f + 4.f * U);                  // N  = nP + T
        P[3] = dT * (1.32f * INL_K) * (1.f + 4.f * U);   // 1.32 K D
    }
}

__device__ __forceinline__ float inl_abs_add_abs(float a, float b) { float d; asm("v_add_f32_e64 %0, |%1|, |%2|" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float inl_add_abs(float acc, float a, float b) {   // acc + |a| + |b|
    float d;
    asm("v_add_f32_e64 %0, %1, |%2|\n\tv_add_f32_e64 %0, %0, |%3|" : "=&v"(d) : "v"(acc), "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float inl_fma_abs0(float a, float b, float c) { float d; asm("v_fma_f32 %0, |%1|, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ float inl_fma_abs1(float a, float b, float c) { float d; asm("v_fma_f32 %0, %1, |%2|, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

__global__ __launch_bounds__(64) void inlier_count_kernel(SolverArgs a, int chunks, int pg) {
    // ONE wave per workgroup: nothing is shared between waves, and a finished wave's slot is refilled at once — with four
    // waves per workgroup the waves of a CU started, reached their fp64 tail and ended together, and the latencies at both
    // ends (the chain of dependent loads at the start, the trip to the fp64 rotations at the end) were paid by idle SIMDs
    typedef unsigned long long u64;
    // pg > 1 (a launch of a frame or two: launch_inlier_count): the hypothesis pairs of a 64-point chunk go to pg waves instead
    // of one -- a lone wave walks its pairs at the latency of their scalar loads, nothing else on its SIMD to hide it
    const int pgi = blockIdx.x % pg, chunk = (blockIdx.x / pg) % chunks, item = blockIdx.x / (pg * chunks);
    const SolverItem S = a.items[item];
    const int lane = threadIdx.x;
    const int i = chunk * 64 + lane;
    // the point's loads do not wait for the point count (rows up to ld exist; what lies past the count is never counted)
    const int ii = max(min(i, S.ld - 1), 0);
    const int m = *S.m_ptr;
    const float U = 5.9604645e-8f;                       // 2^-24
    const float ff = (float)a.sp.f, basef = (float)a.sp.base;
    const float thr2f = (float)(a.sp.inlier_threshold * a.sp.inlier_threshold);
    const float thr_lo = thr2f * (1.f - 4.f * U), thr_hi = thr2f * (1.f + 4.f * U);   // float brackets of the double threshold
    const float cabs = fabsf((float)a.sp.cu) + fabsf((float)a.sp.cv);
    const int iters = a.iters, npair = (iters + 1) >> 1;
    // the point: tier-1 operands and the per-point constants of its error bound
    const double X0d = S.X[0 * S.ld + ii], X1d = S.X[1 * S.ld + ii], X2d = S.X[2 * S.ld + ii];
    const double o0d = S.obs[0 * S.ld + ii], o1d = S.obs[1 * S.ld + ii], o2d = S.obs[2 * S.ld + ii], o3d = S.obs[3 * S.ld + ii];
    const float x0 = (float)X0d, x1 = (float)X1d, x2 = (float)X2d;
    if (i - lane >= m) return;   // wave uniform: none of the wave's points exists
    const bool have = i < m;
    const float qc0 = (float)(o0d - a.sp.cu), qc1 = (float)(o1d - a.sp.cv), qc2 = (float)(o2d - a.sp.cu), qc3 = (float)(o3d - a.sp.cv);
    // D = dP + dT, N = nP + T, the point's share here, the pair's (the larger max|t_i| of its two hypotheses) from the
    // rotation store; PD = {dP, K dP}, PN = {nP, 1.32 K dP}: two packed adds give D, K D, N and 1.32 K D
    inl_f2 PD, PN;
    float k1j, kmag;
    {
        const float q0 = (float)o0d, q1 = (float)o1d, q2 = (float)o2d, q3 = (float)o3d;
        const float pP = (fabsf(x0) + fabsf(x1) + fabsf(x2)) * (1.f + 8.f * U);
        const float dP = fmaf(8.f * U, pP, 1e-30f);
        PD.x = dP; PD.y = dP * INL_K * (1.f + 4.f * U);
        PN.x = fmaf(pP + basef, 1.f + 32.f * U, 1e-30f);   // N >= |X|+|Y|+|Z| + T + base + D: D <= 2^-20 (|X|+|Y|+|Z| + T)
        PN.y = dP * (1.32f * INL_K) * (1.f + 4.f * U);
        const float Cm = (cabs + 2.f * (fabsf(q0) + fabsf(q1) + fabsf(q2) + fabsf(q3))) * (1.f + 8.f * U);
        const float k1 = U * Cm * 1.0625f;
        kmag = fmaf(2e-12f * Cm, Cm, 1e-30f);
        // the loop works with de2 = 2.125 de (2 de, inflated): err2 is off by de (2 sum|e| + 4 de) = 2 de (sum|e| + 2 de)
        k1j = k1 * 2.125f;
    }
    const inl_f2 X0 = {x0, x0}, X1 = {x1, x1}, X2 = {x2, x2};
    const inl_f2 QC0 = {qc0, qc0}, QC1 = {qc1, qc1}, QC2 = {qc2, qc2}, QC3 = {qc3, qc3};
    const inl_f2 K1 = {k1j, k1j}, KM = {kmag, kmag};
    const inl_f2 FF = {ff, ff}, BASE = {basef, basef};
    const inl_f2 ESC = {2.125f * U * 2.125f, 2.125f * U * 2.125f}, S2C = {5.3125f * U, 5.3125f * U};
    const u64 act = __ballot(true), mhave = __ballot(have);
    typedef const __attribute__((address_space(4))) inl_f2* crot_t;
    const crot_t Fall = (crot_t)(const void*)S.rot;
    const double* rotd = reinterpret_cast<const double*>(S.rot + rot_off_d(iters));
    for (int kb = 0; kb < npair; kb += 64) {   // blocks of 64 pairs: lane p of vcnt holds the counts of pair kb + p, 16 bits each
    int vcnt = 0;
    u64 qa = 0, qb = 0;   // per lane: pairs of the block whose first / second hypothesis tier 1 left undecided for the lane's point
    const int ke_all = min(npair, kb + 64);
    const int kb0 = kb + (int)((long long)(ke_all - kb) * pgi / pg), ke = kb + (int)((long long)(ke_all - kb) * (pgi + 1) / pg);   // this wave's pairs of the block
    // did the two hypotheses of pair kb + lane converge?  Asked for here, used when the lane flushes the pair's counts: a
    // failed hypothesis (and the padding of an odd count) has the identity in the rotation store, is counted like any
    // other and dropped at the end — the loop has no case for it (the scalar unit issues one instruction per cycle for the
    // whole CU, as the four SIMDs do together: every scalar instruction of the loop counts like a vector one)
    const bool ok0 = kb + lane < ke && S.ok_h[2 * (kb + lane)] != 0;
    const bool ok1 = kb + lane < ke && 2 * (kb + lane) + 1 < iters && S.ok_h[2 * (kb + lane) + 1] != 0;
    for (int kp = kb0; kp < ke; ++kp) {
        const crot_t F = Fall + (size_t)kp * (INL_PAIR_F / 2);
        const inl_f2 r0 = F[0], r1 = F[1], r2 = F[2], r3 = F[3], r4 = F[4], r5 = F[5], r6 = F[6], r7 = F[7], r8 = F[8],
                     tx = F[9], ty = F[10], tz = F[11];
        // the bound's constants for this point and pair: c1 = D + 4.8 u N, c2 = 1.32 D N, times K
        const inl_f2 VD = PD + F[12], VN = PN + F[13];              // {D, K D}, {N, 1.32 K D}
        const float c1j = fmaf(4.8f * U * INL_K * (1.f + 4.f * U), VN.x, VD.y), c2j = VN.y * VN.x;
        const inl_f2 xc = INL_FMA(r0, X0, INL_FMA(r1, X1, INL_FMA(r2, X2, tx)));
        const inl_f2 yc = INL_FMA(r3, X0, INL_FMA(r4, X1, INL_FMA(r5, X2, ty)));
        const inl_f2 zc = INL_FMA(r6, X0, INL_FMA(r7, X1, INL_FMA(r8, X2, tz)));
        inl_f2 rz;
        rz.x = __builtin_amdgcn_rcpf(zc.x);
        rz.y = __builtin_amdgcn_rcpf(zc.y);
        const inl_f2 g = rz * FF;
        const inl_f2 xb = xc - BASE;
        const inl_f2 e0 = INL_FMA(-xc, g, QC0), e1 = INL_FMA(-yc, g, QC1), e2 = INL_FMA(-xb, g, QC2), e3 = INL_FMA(-yc, g, QC3);
        const inl_f2 s2 = INL_FMA(e0, e0, INL_FMA(e1, e1, INL_FMA(e2, e2, e3 * e3)));
        // sum|e| and the two FMAs on |1/Zc| and |g|: absolute values are source modifiers of the unpacked encodings (as C
        // the compiler builds them from twelve v_and_b32 and packed adds)
        inl_f2 es, de2;   // sum|e|; 2.125 de
        es.x = inl_add_abs(inl_abs_add_abs(e0.x, e1.x), e2.x, e3.x);
        es.y = inl_add_abs(inl_abs_add_abs(e0.y, e1.y), e2.y, e3.y);
        const inl_f2 tail = INL_FMA(ESC, es, K1);     // 2u sum|e| + u Cm, inflated
        de2.x = inl_fma_abs0(g.x, inl_fma_abs1(c2j, rz.x, c1j), tail.x);
        de2.y = inl_fma_abs0(g.y, inl_fma_abs1(c2j, rz.y, c1j), tail.y);
        const inl_f2 ds = INL_FMA(de2, de2 + es, INL_FMA(S2C, s2, KM));
        const inl_f2 hi = s2 + ds, lo = s2 - ds;
        // the verdicts as lane masks (the compares' own results; the logic on them is scalar): "in" = bound holds and
        // err2 + ds below the threshold; undecided = neither that nor (bound holds and err2 - ds above).  Anything not
        // finite fails every compare: undecided
        const inl_f2 DD = {VD.x, VD.x};
        const inl_f2 dz = rz * DD;                                   // D z <= 1/64
        const u64 oka = __ballot(fabsf(dz.x) <= 0.015625f), okb = __ballot(fabsf(dz.y) <= 0.015625f);
        const u64 lta = __ballot(hi.x < thr_lo), ltb = __ballot(hi.y < thr_lo);
        const u64 gta = __ballot(lo.x > thr_hi), gtb = __ballot(lo.y > thr_hi);
        const u64 ina = oka & lta, inb = okb & ltb;
        const u64 una = act & ~(oka & (lta | gta)), unb = act & ~(okb & (ltb | gtb));
        if (una | unb) {   // wave uniform, ~0.2 % of the points: queued for tiers 2 and 3 behind the block (a bit per pair and half)
            const u64 bit = 1ull << (kp - kb);
            if ((una >> lane) & 1) qa |= bit;
            if ((unb >> lane) & 1) qb |= bit;
        }
        const int add = __popcll(ina & mhave) + (__popcll(inb & mhave) << 16);
        vcnt += lane == (kp & 63) ? add : 0;
    }
    // one point per lane: <= 64 per field
    if (ok0 && (vcnt & 0xffff)) atomicAdd(&S.cnt_h[2 * (kb + lane)], vcnt & 0xffff);
    if (ok1 && ((unsigned)vcnt >> 16)) atomicAdd(&S.cnt_h[2 * (kb + lane) + 1], (int)((unsigned)vcnt >> 16));
    // tiers 2 and 3 for what tier 1 left undecided (also everything that is not finite), all of the block's at once:
    // every lane with something queued takes its first entry per turn — one trip to the fp64 rotations per turn instead
    // of one per pair with an undecided point (taken on the spot, with one or two lanes active, those trips were 40 % of
    // the kernel's time).  An inlier adds itself to its hypothesis' count
    while (qa | qb) {   // per lane
        int h;
        if (qa) { h = 2 * (kb + __builtin_ctzll(qa)); qa &= qa - 1; }
        else { h = 2 * (kb + __builtin_ctzll(qb)) + 1; qb &= qb - 1; }
        if (h >= iters || S.ok_h[h] == 0) continue;   // a failed hypothesis: not counted
        RotDev R;               // only the entries predict_point reads
        const double* r = rotd + (size_t)h * 12;
        R.r00 = r[0]; R.r01 = r[1]; R.r02 = r[2]; R.r10 = r[3]; R.r11 = r[4]; R.r12 = r[5];
        R.r20 = r[6]; R.r21 = r[7]; R.r22 = r[8]; R.tx = r[9]; R.ty = r[10]; R.tz = r[11];
        if (is_inlier_pt(R, a.sp, X0d, X1d, X2d, o0d, o1d, o2d, o3d, nullptr) && have) atomicAdd(&S.cnt_h[h], 1);
    }
    }
}

// ransac_rot_kernel + inlier_count_kernel over n_items frames of at most max_points points
static int launch_inlier_count(hipStream_t s, const SolverArgs& a, int max_points) {
    const long long nt = (long long)a.n_items * (((a.iters + 1) / 2) * 2);
    if (nt <= 0) return VISO_OK;
    hipLaunchKernelGGL(ransac_rot_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, a);
    HIP_TRY(hipGetLastError());
    const int chunks = (max_points + 63) / 64;
    if (chunks <= 0) return VISO_OK;
    if ((long long)a.n_items * chunks > 0x7fffffffLL) { viso_set_error("ransac: too many points in one launch"); return VISO_ERR_UNSUPPORTED; }
    const int npair = (a.iters + 1) / 2;
    int pg = 1;   // a launch that cannot fill the chip: more waves, fewer pairs each (the same verdicts, the same atomic adds)
    if ((long long)a.n_items * chunks <= 512) pg = npair >= 20 ? 5 : npair >= 8 ? 2 : 1;
    hipLaunchKernelGGL(inlier_count_kernel, dim3((unsigned)(a.n_items * chunks * pg)), dim3(64), 0, s, a, chunks, pg);
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}

// ---- workgroup helpers ------------------------------------------------------
#ifndef REFIT_THREADS
#define REFIT_THREADS 256
#endif
#define REFIT_WAVES (REFIT_THREADS / 64)

// ordered (ascending index) compaction of the inliers of `tr` into out[];
// returns the count to every thread.  scratch: >= REFIT_WAVES ints of LDS.
__device__ int block_inliers(const double* tr, const SolverParamsDev& sp, const double* X,
                             const double* obs, int ld, int m, int* out, int* scratch,
                             double* last_err2) {
    RotDev R;
    make_rot(tr, R);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int running = 0;
    for (int base = 0; base < m; base += REFIT_THREADS) {
        const int i = base + threadIdx.x;
        double e2 = 0;
        const bool want_e2 = last_err2 && i == m - 1;    // Q8: the rms is the LAST point's error (:1535): that one exactly
        const bool in = (i < m) && is_inlier(R, sp, X, obs, ld, i, want_e2 ? &e2 : nullptr);
        if (want_e2) *last_err2 = e2;
        const unsigned long long msk = __ballot(in);
        if (lane == 0) scratch[wave] = __popcll(msk);
        __syncthreads();
        int off = running;
        for (int w = 0; w < wave; ++w) off += scratch[w];
        int total = 0;
#pragma unroll
        for (int w = 0; w < REFIT_WAVES; ++w) total += scratch[w];
        if (in) out[off + __builtin_amdgcn_mbcnt_hi((uint32_t)(msk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)msk, 0u))] = i;
        running += total;
        __syncthreads();
    }
    return running;
}

// A wave-uniform double into scalar registers (the compiler cannot know a value that came back from LDS or from a
// cross-lane read is uniform): v_readfirstlane of both halves.
__device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// Sum of a double over the wave, valid in lane 63: row_shr 1, 2, 4, 8 inside the rows of 16 lanes (lanes without a source
// add 0), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3 (gfx9 reduction idiom; the halves of the
// double move as two 32-bit DPP moves).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
    v += dpp_f64<0x111, 0xf>(v);   // row_shr:1
    v += dpp_f64<0x112, 0xf>(v);   // row_shr:2
    v += dpp_f64<0x114, 0xf>(v);   // row_shr:4
    v += dpp_f64<0x118, 0xf>(v);   // row_shr:8  -> lane 15 of every row = the row's sum
    v += dpp_f64<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v += dpp_f64<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 = the wave's sum
    return v;
}

#ifdef VISO_DEBUG_VARIANTS   // timing aid (tools/experiments/refit_phases.py): 100 MHz time stamps of the refit's phases, item 0
__device__ unsigned long long viso_dbg_clk[16];
extern "C" int viso_debug_refit_clocks(unsigned long long* out16) {
    (void)hipDeviceSynchronize();
    const hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(viso_dbg_clk), sizeof(unsigned long long) * 16, 0, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { fprintf(stderr, "viso_debug_refit_clocks: %s\n", hipGetErrorString(e)); return VISO_ERR_HIP; }
    return VISO_OK;
}
#define DBG_CLK(I) do { if (threadIdx.x == 0 && item == 0) viso_dbg_clk[I] = wall_clock64(); } while (0)
#define DBG_SET(I, V) do { if (threadIdx.x == 0 && item == 0) viso_dbg_clk[I] = (V); } while (0)
#else
#define DBG_CLK(I) do {} while (0)
#define DBG_SET(I, V) do {} while (0)
#endif
// minimize_reproj over an arbitrary active list, one workgroup.  Every thread
// accumulates its points' contribution to the 21+6 sums, the workgroup reduces
// them (wave shuffles, then a fixed-order sum over the waves in LDS), the first
// wave solves the 6x6 system with ONE ENTRY PER LANE (lu_lane_step: bit-identical
// to lu_solve6, see ransac_coop_kernel) and publishes the step.  tr_s: 6 doubles
// in LDS (in/out).  red: >= REFIT_WAVES*27+8 doubles of LDS.  Returns 1/0 to every thread.
// The rotation table is wave uniform: lanes 0..2 take the three sincos, the 36
// entries live in SCALAR registers while the points are accumulated (round 3
// kept them, the Jacobian rows, the 27 sums and lane 0's 6x6 matrix in 254
// vector registers: two waves per SIMD, and a workgroup that only starts where
// half a SIMD's registers are free on all four SIMDs of a CU).
__device__ int gn_block(const double* X, const double* obs, int ld, const int* active, int n,
                        double* tr_s, const SolverParamsDev& sp, double* red) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = lane >> 3, col = lane & 7;       // wave 0: this lane's entry of [A | b] (row < 6, col < 7)
    if (n <= 0) return 0;
    for (int it = 0; it < 100; ++it) {
#ifdef VISO_DEBUG_VARIANTS
        if (threadIdx.x == 0 && blockIdx.x == 0) viso_dbg_clk[7] = (unsigned long long)(it + 1);
#endif
        double tr[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) tr[j] = uni(tr_s[j]);
        RotDev R;
        {
            double sv, cv;
            const int l3 = lane % 3;
            sincos(l3 == 0 ? tr[0] : l3 == 1 ? tr[1] : tr[2], &sv, &cv);
            const double sx = rdlane(sv, 0), cx = rdlane(cv, 0), sy = rdlane(sv, 1), cy = rdlane(cv, 1), sz = rdlane(sv, 2), cz = rdlane(cv, 2);
            rot_from_sincos(sx, cx, sy, cy, sz, cz, tr, R);   // scalar operands in, uniform values out
            R.r00 = uni(R.r00); R.r01 = uni(R.r01); R.r02 = uni(R.r02); R.r10 = uni(R.r10); R.r11 = uni(R.r11); R.r12 = uni(R.r12);
            R.r20 = uni(R.r20); R.r21 = uni(R.r21); R.r22 = uni(R.r22);
            R.rdrx10 = uni(R.rdrx10); R.rdrx11 = uni(R.rdrx11); R.rdrx12 = uni(R.rdrx12);
            R.rdrx20 = uni(R.rdrx20); R.rdrx21 = uni(R.rdrx21); R.rdrx22 = uni(R.rdrx22);
            R.rdry00 = uni(R.rdry00); R.rdry01 = uni(R.rdry01); R.rdry02 = uni(R.rdry02);
            R.rdry10 = uni(R.rdry10); R.rdry11 = uni(R.rdry11); R.rdry12 = uni(R.rdry12);
            R.rdry20 = uni(R.rdry20); R.rdry21 = uni(R.rdry21); R.rdry22 = uni(R.rdry22);
            R.rdrz00 = uni(R.rdrz00); R.rdrz01 = uni(R.rdrz01); R.rdrz10 = uni(R.rdrz10); R.rdrz11 = uni(R.rdrz11);
            R.rdrz20 = uni(R.rdrz20); R.rdrz21 = uni(R.rdrz21);
        }
        double S[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) S[k] = 0;
        for (int i = threadIdx.x; i < n; i += REFIT_THREADS)
            accumulate_point_rows(R, sp, X, obs, ld, active[i], i, S);
        // wave reduction of the 27 sums: DPP row shifts and row broadcasts, the total in lane 63 (six LDS-crossbar
        // butterflies per sum were 5.8 us of an iteration: two ds_bpermute and their wait per step)
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const double v = wave_sum_to_lane63(S[k]);
            if (lane == 63) red[wave * 27 + k] = v;
            if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four chains in flight, not 27: registers
        }
        __syncthreads();
        if (wave == 0) {
            // entry (row, col) of the augmented system [J^T J | J^T r]: the sum of upper-triangle slot (min, max), or of
            // slot 21 + row for the right-hand side; lanes outside the 6 x 7 grid carry a copy of a valid entry
            const int r6 = min(row, 5), c7 = min(col, 6);
            const int lo = min(r6, c7), hi = max(r6, c7);
            const int k = c7 == 6 ? 21 + r6 : lo * 6 - lo * (lo - 1) / 2 + (hi - lo);
            double acc = red[k];
#pragma unroll
            for (int w = 1; w < REFIT_WAVES; ++w) acc += red[27 * w + k];   // fixed order over the waves
            bool regular = lu_lane_step<0>(acc, lane, row, col);              // cv::solve(DECOMP_LU), :1602-1606
            regular = regular && lu_lane_step<1>(acc, lane, row, col);
            regular = regular && lu_lane_step<2>(acc, lane, row, col);
            regular = regular && lu_lane_step<3>(acc, lane, row, col);
            regular = regular && lu_lane_step<4>(acc, lane, row, col);
            regular = regular && lu_lane_step<5>(acc, lane, row, col);
            int status = 2;   // 0 = continue, 1 = converged, 2 = singular
            if (regular) {
                double Bs[6];
#pragma unroll
                for (int i = 5; i >= 0; --i) {
                    double sacc = rdlane(acc, i * 8 + 6);
#pragma unroll
                    for (int cc = i + 1; cc < 6; ++cc) sacc -= rdlane(acc, i * 8 + cc) * Bs[cc];
                    Bs[i] = sacc * rdlane(acc, i * 8 + i);
                }
                bool converged = true;
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    if (Bs[j] > sp.thresh) converged = false;                 // Q7, :1610
                status = converged ? 1 : 0;
                if (!converged && lane == 0) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) tr_s[j] = tr[j] + Bs[j];
                }
            }
            if (lane == 0) reinterpret_cast<int*>(red + REFIT_WAVES * 27)[0] = status;
        }
        __syncthreads();
        const int status = reinterpret_cast<int*>(red + REFIT_WAVES * 27)[0];
        __syncthreads();
        if (status == 1) return 1;
        if (status == 2) return 0;
    }
    return 0;
}

// ---- stage 3: best hypothesis -> support set -> refit -> final support ------
__device__ void refit_item(const SolverArgs& a, int item, double* tr_s, double* red, int* scratch) {
    DBG_CLK(0);
    const SolverItem S = a.items[item];
    const int m = *S.m_ptr;
    if (m < 3) {   // sequence_odometry's guard (:1283); randomsample(3,m) would not return
        if (threadIdx.x == 0) { *S.ok = 0; *S.n_inl = 0; if (S.kept) *S.kept = 1; }
        if (!S.kept && threadIdx.x < 6) S.tr[threadIdx.x] = 0.0;   // vector<double> tr(6,0), :1312 (no memset in front of the stage)
        return;
    }
    // The hypothesis with the largest support, the FIRST of them among equals (strict >, :1564), none without support:
    // the maximum of (count << 32 | ~index) over the hypotheses, all threads at once (one thread walking the 200
    // flags and counts in memory was 7 of this kernel's 54 us on a single frame)
    {
        unsigned long long key = 0;
        for (int h = threadIdx.x; h < a.iters; h += REFIT_THREADS) {
            const int c = S.ok_h[h] ? S.cnt_h[h] : 0;
            const unsigned long long kh = c > 0 ? ((unsigned long long)(unsigned)c << 32) | (unsigned)~h : 0ull;
            key = kh > key ? kh : key;
        }
        key = viso_wave_max63(key);
        unsigned long long* wkey = reinterpret_cast<unsigned long long*>(red);   // red is free until gn_block
        if ((threadIdx.x & 63) == 63) wkey[threadIdx.x >> 6] = key;
        __syncthreads();
        key = wkey[0];
#pragma unroll
        for (int w = 1; w < REFIT_WAVES; ++w) key = wkey[w] > key ? wkey[w] : key;
        const int best = key ? (int)~(unsigned)key : -1;
        if (threadIdx.x < 6) tr_s[threadIdx.x] = best >= 0 ? S.tr_h[6 * best + threadIdx.x] : 0.0;
        if (threadIdx.x == 0) scratch[REFIT_WAVES] = best;
    }
    __syncthreads();
    const int best = scratch[REFIT_WAVES];
    __syncthreads();
    if (best < 0) {   // no hypothesis found any support: best_inliers stays empty, :1571
        if (threadIdx.x == 0) { *S.ok = 0; *S.n_inl = 0; if (S.kept) *S.kept = 1; }   // best_tr is assigned on improvement only (:1564-1568):
        if (!S.kept && threadIdx.x < 6) S.tr[threadIdx.x] = 0.0;                       // the caller's value stays -- zeros in sequence_odometry, :1312
        return;
    }
    double tr[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) tr[j] = tr_s[j];
    DBG_CLK(1);
    int n = block_inliers(tr, a.sp, S.X, S.obs, S.ld, m, S.inl, scratch, nullptr);
    __syncthreads();
    DBG_CLK(2);
    DBG_SET(6, (unsigned long long)n);
    int ok = 0;
    if (n >= 6) {
        __threadfence_block();
        ok = gn_block(S.X, S.obs, S.ld, S.inl, n, tr_s, a.sp, red);   // :1572
        __syncthreads();
        DBG_CLK(3);
        if (ok) {
#pragma unroll
            for (int j = 0; j < 6; ++j) tr[j] = tr_s[j];
            n = block_inliers(tr, a.sp, S.X, S.obs, S.ld, m, S.inl, scratch, nullptr);   // :1575-1576
        }
    }
    DBG_CLK(4);
    if (threadIdx.x == 0) {
        for (int j = 0; j < 6; ++j) S.tr[j] = tr_s[j];
        *S.ok = ok;
        *S.n_inl = n;
        if (S.kept) *S.kept = 0;
    }
}

// Grid-stride over the frames (the launch gives every frame its own workgroup; thinner grids that last longer were
// measured and lose: the chain's latency then limits the batches in flight).
__global__ __launch_bounds__(REFIT_THREADS) void ransac_refit_kernel(SolverArgs a) {
    __shared__ double tr_s[6];
    __shared__ double red[REFIT_WAVES * 27 + 8];
    __shared__ int scratch[REFIT_WAVES + 8];
    __builtin_amdgcn_s_setprio(3);                   // see ransac_hyp_kernel
    for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
        refit_item(a, item, tr_s, red, scratch);
        __syncthreads();
    }
    if (a.mir.sig.flag) {   // uniform (one item, one workgroup): what this workgroup has just written, into the call's mirror
        __threadfence_block();
        __syncthreads();
        const RefitMirror& M = a.mir;
        for (int i = threadIdx.x; i < M.res_words; i += REFIT_THREADS) M.res_dst[i] = M.res_src[i];
        int n = *M.n_inl;
        n = n < 0 ? 0 : n > M.max_inl ? M.max_inl : n;
        for (int i = threadIdx.x; i < n; i += REFIT_THREADS) M.inl_dst[i] = M.inl_src[i];
        plain_signal_done(M.sig, gridDim.x);
    }
#ifdef VISO_DEBUG_VARIANTS
    if (threadIdx.x == 0 && blockIdx.x == 0) viso_dbg_clk[5] = wall_clock64();
#endif
}


int launch_ransac(hipStream_t s, const SolverItem* items_dev, int n_items, int iters,
                  unsigned long long seed, const SolverParamsDev& sp, int* queue, int split, int max_points, const RefitMirror* mir,
                  const OutArgs* ride, int ride_blocks) {
    if (n_items <= 0) return VISO_OK;
    if (mir && n_items != 1) { viso_set_error("ransac: a mirror is for one item"); return VISO_ERR_ARG; }
    SolverArgs a;
    a.items = items_dev; a.n_items = n_items; a.iters = iters; a.seed = seed; a.sp = sp; a.queue = queue;
    a.mir = RefitMirror{};
    if (mir) a.mir = *mir;
    a.ride = nullptr; a.ride_blocks = 0; a.coop_blocks = 0;
    if (ride && ride_blocks > 0 && (long long)n_items * iters > 0) { a.ride = ride; a.ride_blocks = ride_blocks; }
    else if (ride) { viso_set_error("ransac: nothing for the copy-out to ride in"); return VISO_ERR_ARG; }
    a.split = split >= 1 && split <= 100 ? split : VISO_GN_SPLIT;
    const long long nh = (long long)n_items * iters;
    if (nh > 0x7fffffffLL) { viso_set_error("ransac: too many hypotheses in one launch"); return VISO_ERR_UNSUPPORTED; }
    if (nh > 0) {
        // ransac_coop_kernel's grid: one wave per undecided hypothesis.  After 10 iterations 1-2 % of them are undecided (room
        // for 2.5 %); a shorter first stage hands on more (the waves are light: 117 VGPRs), so the grid grows with what can
        // be expected; waves without an entry leave at once, entries beyond the grid are taken in further turns of the same
        // waves (the kernel strides by coop_blocks: riders may sit behind them in the launch)
        const int div = a.split >= 10 ? 40 : a.split >= 6 ? 16 : a.split >= 4 ? 6 : 3;
        long long cb = (nh / div + 3) / 4;
        if (cb < 128) cb = 128;
        if (cb > (nh + 3) / 4) cb = (nh + 3) / 4;
        a.coop_blocks = (int)cb;
        // four waves per workgroup: a 256-register wave halves what its SIMD can hold of another batch's matcher, so the
        // 200 waves go to 50 CUs instead of one to each of 200
        hipLaunchKernelGGL(ransac_hyp_kernel, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, s, a);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(ransac_coop_kernel, dim3((unsigned)cb + (unsigned)a.ride_blocks), dim3(256), 0, s, a);
        int ri = hipGetLastError() == hipSuccess ? VISO_OK : VISO_ERR_HIP;
        if (ri >= 0) ri = launch_inlier_count(s, a, max_points);
        if (ri < 0) {   // the chain broke behind ransac_hyp_kernel: the list would stay filled for the next one
            (void)hipMemsetAsync(queue, 0, sizeof(int), s);
            viso_set_error("ransac: a launch of the chain failed");
            return ri;
        }
    }
    hipLaunchKernelGGL(ransac_refit_kernel, dim3(n_items), dim3(REFIT_THREADS), 0, s, a);
    HIP_TRY(hipGetLastError());
    return VISO_OK;
}

// ---- single minimize_reproj / get_inliers (plain family) --------------------
struct GnArgs {
    const double* X; const double* obs; int m, ld;
    const int* active; int n;
    const double* tr_in;   // the caller's tr (input block)
    double* tr; int* ok; int* inl; int* n_inl; double* rms;   // result block
    SolverParamsDev sp;
};

__global__ __launch_bounds__(REFIT_THREADS) void minimize_reproj_kernel(GnArgs a) {
    __shared__ double tr_s[6];
    __shared__ double red[REFIT_WAVES * 27 + 8];
    if (threadIdx.x < 6) tr_s[threadIdx.x] = a.tr_in[threadIdx.x];
    __syncthreads();
    const int ok = gn_block(a.X, a.obs, a.ld, a.active, a.n, tr_s, a.sp, red);
    __syncthreads();
    if (threadIdx.x < 6) a.tr[threadIdx.x] = tr_s[threadIdx.x];
    if (threadIdx.x == 0) *a.ok = ok;
}

__global__ __launch_bounds__(REFIT_THREADS) void get_inliers_kernel(GnArgs a) {
    __shared__ int scratch[REFIT_WAVES + 8];
    __shared__ double last;
    double tr[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) tr[j] = a.tr_in[j];
    if (threadIdx.x == 0) last = 0;
    __syncthreads();
    const int n = block_inliers(tr, a.sp, a.X, a.obs, a.ld, a.m, a.inl, scratch, &last);
    __syncthreads();
    if (threadIdx.x == 0) {
        *a.n_inl = n;
        *a.rms = sqrt(last / a.m);   // Q8: error of the last point only, :1535
    }
}

// ------------------------------------------------------------ host entry points
void fill_solver_params(SolverParamsDev* d, const viso_param* h) {
    d->base = h->base; d->f = h->f; d->cu = h->cu; d->cv = h->cv;
    d->inlier_threshold = h->inlier_threshold; d->thresh = h->thresh;
    d->ransac_iter = h->ransac_iter; d->_pad = 0;
}

extern "C" int viso_minimize_reproj(const double* X, const double* obs, int m, double tr[6],
                                    const viso_param* p, const int32_t* active, int n_active) {
    if (!X || !obs || !tr || !p || m < 0 || n_active < 0 || (n_active > 0 && !active)) {
        viso_set_error("viso_minimize_reproj: bad argument");
        return VISO_ERR_ARG;
    }
    for (int i = 0; i < n_active; ++i)
        if (active[i] < 0 || active[i] >= m || i >= m) { viso_set_error("viso_minimize_reproj: active index out of range"); return VISO_ERR_ARG; }
    if (n_active == 0) return 0;
    PlainLock lk;
    viso_ctx* c = viso_default_ctx();
    if (!c) return VISO_ERR_HIP;
    HIP_TRY(hipSetDevice(c->device));
    PlainProf pp(VISO_PLAIN_MINIMIZE, c->stream);
    int r;
    PlainStage in;   // ONE upload: X | obs | active; ONE read-back: tr[6], ok
    if ((r = in.begin(c, PlainStage::need(sizeof(double) * 3 * (size_t)m) + PlainStage::need(sizeof(double) * 4 * (size_t)m) +
                         PlainStage::need(sizeof(int) * (size_t)n_active) + PlainStage::need(64))) < 0) return r;
    char *dout, *hout;
    if ((r = ctx_scratch(c, PLAIN_SLOT_OUT, 128, (void**)&dout)) < 0) return r;
    if ((r = ctx_pinned(c, 1, 128, &hout)) < 0) return r;
    GnArgs a{};
    a.X = in.put(X, 3 * (size_t)m); a.obs = in.put(obs, 4 * (size_t)m); a.m = m; a.ld = m;
    a.active = in.put(active, (size_t)n_active); a.n = n_active;
    a.tr_in = in.put(tr, 6);
    a.tr = reinterpret_cast<double*>(dout); a.ok = reinterpret_cast<int*>(dout + 64);
    fill_solver_params(&a.sp, p);
    if ((r = in.flush(c->stream)) < 0) return r;
    pp.mark(1);
    hipLaunchKernelGGL(minimize_reproj_kernel, dim3(1), dim3(REFIT_THREADS), 0, c->stream, a);
    HIP_TRY(hipGetLastError());
    pp.mark(2);
    PlainSignal sig_;
    if ((r = plain_signal_next(c, &sig_)) < 0) return r;
    if ((r = plain_blit(c->stream, dout, hout, 32, nullptr, 0, 0, &sig_)) < 0) return r;
    pp.wait_begin();
    if ((r = plain_signal_wait(c, c->stream, sig_.seq)) < 0) return r;
    pp.wait_end();
    memcpy(tr, hout, sizeof(double) * 6);
    const int ok = *reinterpret_cast<const int*>(hout + 64);
    pp.mark(3);
    return ok ? 1 : 0;
}

extern "C" int viso_get_inliers(const double* X, const double* obs, int m, const double tr[6],
                                const viso_param* p, int32_t* inliers, int* n_inliers, double* rms) {
    if (!X || !obs || !tr || !p || m < 0 || !inliers || !n_inliers) {
        viso_set_error("viso_get_inliers: bad argument");
        return VISO_ERR_ARG;
    }
    *n_inliers = 0;
    if (m == 0) { if (rms) *rms = NAN; return VISO_OK; }
    PlainLock lk;
    viso_ctx* c = viso_default_ctx();
    if (!c) return VISO_ERR_HIP;
    HIP_TRY(hipSetDevice(c->device));
    PlainProf pp(VISO_PLAIN_GET_INLIERS, c->stream);
    int r;
    PlainStage in;   // ONE upload: X | obs | tr; ONE read-back: {n, rms} | inliers[m]
    if ((r = in.begin(c, PlainStage::need(sizeof(double) * 3 * (size_t)m) + PlainStage::need(sizeof(double) * 4 * (size_t)m) + PlainStage::need(64))) < 0) return r;
    char *dout, *hout;
    const size_t out_bytes = 64 + sizeof(int) * (size_t)m;
    if ((r = ctx_scratch(c, PLAIN_SLOT_OUT, out_bytes, (void**)&dout)) < 0) return r;
    if ((r = ctx_pinned(c, 1, out_bytes, &hout)) < 0) return r;
    GnArgs a{};
    a.X = in.put(X, 3 * (size_t)m); a.obs = in.put(obs, 4 * (size_t)m); a.m = m; a.ld = m;
    a.tr_in = in.put(tr, 6);
    a.n_inl = reinterpret_cast<int*>(dout); a.rms = reinterpret_cast<double*>(dout + 8); a.inl = reinterpret_cast<int*>(dout + 64);
    fill_solver_params(&a.sp, p);
    if ((r = in.flush(c->stream)) < 0) return r;
    pp.mark(1);
    hipLaunchKernelGGL(get_inliers_kernel, dim3(1), dim3(REFIT_THREADS), 0, c->stream, a);
    HIP_TRY(hipGetLastError());
    pp.mark(2);
    PlainSignal sig_;
    if ((r = plain_signal_next(c, &sig_)) < 0) return r;
    if ((r = plain_blit(c->stream, dout, hout, 16, a.n_inl, 1, m, &sig_)) < 0) return r;
    pp.wait_begin();
    if ((r = plain_signal_wait(c, c->stream, sig_.seq)) < 0) return r;
    pp.wait_end();
    const int n = *reinterpret_cast<const int*>(hout);
    if (n < 0 || n > m) { viso_set_error("viso_get_inliers: device returned %d inliers of %d points", n, m); return VISO_ERR_HIP; }
    if (n > 0) memcpy(inliers, hout + 64, sizeof(int) * (size_t)n);
    *n_inliers = n;
    if (rms) memcpy(rms, hout + 8, sizeof(double));
    pp.mark(3);
    return VISO_OK;
}

// Support sizes of given motions (diagnostics / tests): inlier_count_kernel, the RANSAC stage's counting kernel, on n_h
// motions tr_h[n_h][6] over one point set -- cnt[h] must equal the length of get_inliers(X, obs, tr_h[h]) (:1509-1537).
extern "C" int viso_support_sizes(const double* X, const double* obs, int m, const double* tr_h, int n_h,
                                  const viso_param* p, int32_t* cnt) {
    if (!X || !obs || !tr_h || !p || !cnt || m < 0 || n_h < 0) { viso_set_error("viso_support_sizes: bad argument"); return VISO_ERR_ARG; }
    if (n_h == 0) return VISO_OK;
    PlainLock lk;
    viso_ctx* c = viso_default_ctx();
    if (!c) return VISO_ERR_HIP;
    double *dX, *dobs, *dtrh; int *dmisc; SolverItem* ditem;
    int r;
    if ((r = ctx_scratch(c, 0, sizeof(double) * 3 * (size_t)(m + 1), (void**)&dX)) < 0) return r;
    if ((r = ctx_scratch(c, 1, sizeof(double) * 4 * (size_t)(m + 1), (void**)&dobs)) < 0) return r;
    if ((r = ctx_scratch(c, 4, sizeof(int) * (4 + 2 * (size_t)n_h), (void**)&dmisc)) < 0) return r;
    if ((r = ctx_scratch(c, 5, sizeof(double) * 6 * (size_t)n_h, (void**)&dtrh)) < 0) return r;
    if ((r = ctx_scratch(c, 6, sizeof(SolverItem), (void**)&ditem)) < 0) return r;
    std::vector<int> hm(4 + 2 * (size_t)n_h, 1);   // m, -, -, -, ok_h = 1 ..., cnt_h
    hm[0] = m;
    if (m > 0) {
        HIP_TRY(hipMemcpyAsync(dX, X, sizeof(double) * 3 * m, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(dobs, obs, sizeof(double) * 4 * m, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(dtrh, tr_h, sizeof(double) * 6 * n_h, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(dmisc, hm.data(), sizeof(int) * hm.size(), hipMemcpyHostToDevice, c->stream));
    SolverItem it{};
    it.X = dX; it.obs = dobs; it.m_ptr = dmisc; it.ld = m; it.tr_h = dtrh; it.ok_h = dmisc + 4; it.cnt_h = dmisc + 4 + n_h;
    if ((r = ctx_scratch(c, 9, viso_rot_bytes(n_h), (void**)&it.rot)) < 0) return r;
    HIP_TRY(hipMemcpyAsync(ditem, &it, sizeof(it), hipMemcpyHostToDevice, c->stream));
    SolverArgs a{};
    a.items = ditem; a.n_items = 1; a.iters = n_h;
    fill_solver_params(&a.sp, p);
    if ((r = launch_inlier_count(c->stream, a, m)) < 0) return r;
    HIP_TRY(hipMemcpyAsync(cnt, dmisc + 4 + n_h, sizeof(int) * n_h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VISO_OK;
}

extern "C" int viso_ransac_minimize_reproj(const double* X, const double* obs, int m,
                                           double best_tr[6], int32_t* best_inl, int* n_inl,
                                           const viso_param* p, const int32_t* samples,
                                           uint64_t seed, uint64_t frame) {
    if (!X || !obs || !best_tr || !p || m < 0 || !best_inl || !n_inl || p->ransac_iter < 0) {
        viso_set_error("viso_ransac_minimize_reproj: bad argument");
        return VISO_ERR_ARG;
    }
    *n_inl = 0;
    if (m < 3) return 0;
    const int iters = p->ransac_iter;
    if (samples)
        for (int i = 0; i < 3 * iters; ++i)
            if (samples[i] < 0 || samples[i] >= m) { viso_set_error("viso_ransac_minimize_reproj: sample index out of range"); return VISO_ERR_ARG; }
    PlainLock lk;
    viso_ctx* c = viso_default_ctx();
    if (!c) return VISO_ERR_HIP;
    HIP_TRY(hipSetDevice(c->device));
    PlainProf pp(VISO_PLAIN_RANSAC, c->stream);
    {   // the frame's stereo call may have solved exactly this problem already (plain.hip)
        int ret = 0;
        if (plain_try_ransac(c, X, obs, m, best_tr, best_inl, n_inl, p, samples, seed, frame, &ret)) return ret;
    }
    int r;
    // ONE upload: X | obs | samples | {m} | the item; ONE read-back: {kept, ok, n_inl} | tr[6] | inliers[m]
    PlainStage in;
    if ((r = in.begin(c, PlainStage::need(sizeof(double) * 3 * (size_t)m) + PlainStage::need(sizeof(double) * 4 * (size_t)m) +
                         PlainStage::need(sizeof(int) * 3 * (size_t)(iters + 1)) + PlainStage::need(16) + PlainStage::need(sizeof(SolverItem)))) < 0) return r;
    char *dout, *hout;
    const size_t out_bytes = 128 + sizeof(int) * (size_t)m;
    if ((r = ctx_scratch(c, PLAIN_SLOT_OUT, out_bytes, (void**)&dout)) < 0) return r;
    if ((r = ctx_pinned(c, 1, out_bytes, &hout)) < 0) return r;
    double* dtrh; int *dhyp, *dqueue;
    if ((r = ctx_scratch(c, 4, sizeof(int) * (4 + 2 * (size_t)iters), (void**)&dhyp)) < 0) return r;
    if ((r = ctx_scratch(c, 5, sizeof(double) * 6 * (size_t)(iters + 1), (void**)&dtrh)) < 0) return r;
    // undecided-hypothesis list, then the triples in use
    if ((r = ctx_scratch(c, 8, sizeof(int) * (2 + 4 * (size_t)iters + 3), (void**)&dqueue, true)) < 0) return r;
    SolverItem it{};
    it.X = in.put(X, 3 * (size_t)m); it.obs = in.put(obs, 4 * (size_t)m); it.ld = m; it.frame = frame;
    it.samples = samples ? in.put(samples, 3 * (size_t)iters) : nullptr;
    const int hm[4] = {m, 0, 0, 0};
    it.m_ptr = in.put(hm, 4);
    it.samp_h = dqueue + 2 + iters;
    it.tr_h = dtrh; it.ok_h = dhyp; it.cnt_h = dhyp + iters;
    if ((r = ctx_scratch(c, 9, viso_rot_bytes(iters), (void**)&it.rot)) < 0) return r;
    // the refit writes kept, ok and n_inl whatever happens (refit_item): nothing of the result block needs a value in advance.
    // best_tr is an input of the reference's function only as the value that STAYS when no hypothesis finds support
    // (src/viso.cpp:1564-1568): the stage says so in `kept` and the caller's array is then left alone, below.
    it.kept = reinterpret_cast<int*>(dout);
    it.ok = reinterpret_cast<int*>(dout) + 1; it.n_inl = reinterpret_cast<int*>(dout) + 2;
    it.tr = reinterpret_cast<double*>(dout + 64); it.inl = reinterpret_cast<int*>(dout + 128);
    const SolverItem* ditem = in.put(&it, 1);
    if ((r = in.flush(c->stream)) < 0) return r;
    pp.mark(1);
    SolverParamsDev sp;
    fill_solver_params(&sp, p);
    // one frame's 50 hypotheses are ONE wave of the lane-per-hypothesis kernel (5 us per iteration): hand over to the
    // wave-per-hypothesis kernel after the first iteration (2.7 us each, all hypotheses side by side) unless a split was asked
    // for (viso_ctx_set_gn_split): 203 -> 183 us per call (tools/dropin_probe.py, GN_SPLIT sweep); same hypotheses bit for bit
    PlainSignal sig_;
    if ((r = plain_signal_next(c, &sig_)) < 0) return r;
    RefitMirror mir{};   // the result block and the inliers go into pinned memory from the refit kernel itself, which signals
    mir.res_src = reinterpret_cast<const uint32_t*>(dout); mir.res_dst = reinterpret_cast<uint32_t*>(hout); mir.res_words = 32;
    mir.n_inl = it.n_inl; mir.inl_src = reinterpret_cast<const uint32_t*>(dout + 128); mir.inl_dst = reinterpret_cast<uint32_t*>(hout + 128); mir.max_inl = m;
    mir.sig = sig_;
    if ((r = launch_ransac(c->stream, ditem, 1, iters, seed, sp, dqueue, c->gn_split ? c->gn_split : 1, m, &mir)) < 0) return r;
    pp.mark(2);
    pp.wait_begin();
    if ((r = plain_signal_wait(c, c->stream, sig_.seq)) < 0) return r;
    pp.wait_end();
    const int* res = reinterpret_cast<const int*>(hout);
    if (res[2] < 0 || res[2] > m) { viso_set_error("viso_ransac_minimize_reproj: device returned %d inliers of %d points", res[2], m); return VISO_ERR_HIP; }
    if (!res[0]) memcpy(best_tr, hout + 64, sizeof(double) * 6);   // kept: in/out like the reference's (:1564-1568)
    *n_inl = res[2];
    if (res[2] > 0) memcpy(best_inl, hout + 128, sizeof(int) * (size_t)res[2]);
    pp.mark(3);
    return res[1] ? 1 : 0;
}
