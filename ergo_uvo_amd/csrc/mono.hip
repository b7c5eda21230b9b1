// mono.hip -- the mono node's relative-pose stage on gfx950 (replaces estimate_relative_pose and what it
// calls, VO_utility.cpp:134-180: cv::findEssentialMat + cv::recoverPose, cv::findHomography +
// recover_pose_homography (cv::decomposeHomographyMat, cv::triangulatePoints), VO_utility.cpp:581-624).
//
// Same scheme as the PnP stage: the host replays cv::RNG to produce OpenCV's exact sample subsets
// (including HomographyEstimatorCallback::checkSubset), ALL hypotheses are solved and scored on the
// device in parallel, and the host replays the sequential RANSAC ("better than best => shrink niters")
// or LMedS ("smaller median wins") scan to pick the identical winner.  Small sequential fp64 pieces
// (homography refit + Levenberg-Marquardt polish, decomposeHomographyMat) stay on the host as
// SURVEY.md 2.3 (K5, K6) plans.
//   k_fivepoint_hyp : four 5-point hypotheses per wave (Nister/Stewenius as in OpenCV's five-point.cpp)
//   k_e_score       : Sampson error of every (hypothesis, model) over all matches -> inlier count or median
//   k_h_hyp         : normalised-DLT homographies, sixteen lanes per hypothesis (9x9 Jacobi eigen: pivot search, rotations and index rescans across the lanes)
//   k_h_score       : reprojection error -> inlier count or median
//   k_model_mask    : mask of the winning model
//   k_recover_pose  : cheirality test of the 4 (R,t) candidates
#include "uvo_ctx.h"
#include "uvo_mono.h"
#include "uvo_epnp.h"
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>

namespace uvo {

// ------------------------------------------------------------------------------------------
// five-point solver (device)
// ------------------------------------------------------------------------------------------
// monomial order of the 20 columns: x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy | xz^2 xz x yz^2 yz y z^3 z^2 z 1
__device__ const signed char kMono[20][3] = {
    {3,0,0},{0,3,0},{2,1,0},{1,2,0},{2,0,1},{2,0,0},{0,2,1},{0,2,0},{1,1,1},{1,1,0},
    {1,0,2},{1,0,1},{1,0,0},{0,1,2},{0,1,1},{0,1,0},{0,0,3},{0,0,2},{0,0,1},{0,0,0} };
// index of x^a y^b z^c in kMono (a + b + c <= 3), as a lookup table
__device__ const signed char kMonoIdx[64] = {
    /* a=0 */ 19, 18, 17, 16,  15, 14, 13, -1,   7,  6, -1, -1,   1, -1, -1, -1,
    /* a=1 */ 12, 11, 10, -1,   9,  8, -1, -1,   3, -1, -1, -1,  -1, -1, -1, -1,
    /* a=2 */  5,  4, -1, -1,   2, -1, -1, -1,  -1, -1, -1, -1,  -1, -1, -1, -1,
    /* a=3 */  0, -1, -1, -1,  -1, -1, -1, -1,  -1, -1, -1, -1,  -1, -1, -1, -1 };
__device__ __forceinline__ int mono_index(int a, int b, int c) { return kMonoIdx[a * 16 + b * 4 + c]; }
__device__ void p_zero(double* p) { for (int i = 0; i < 20; i++) p[i] = 0; }
// out = a * b (out must not alias a or b); products of total degree > 3 never occur
__device__ void p_mul(const double* a, const double* b, double* out)
{
    p_zero(out);
    for (int i = 0; i < 20; i++) {
        double ai = a[i];
        if (ai == 0) continue;
        for (int j = 0; j < 20; j++) {
            double bj = b[j];
            if (bj == 0) continue;
            int e0 = kMono[i][0] + kMono[j][0], e1 = kMono[i][1] + kMono[j][1], e2 = kMono[i][2] + kMono[j][2];
            if (e0 + e1 + e2 > 3) continue;
            out[mono_index(e0, e1, e2)] += ai * bj;
        }
    }
}
__device__ void p_axpy(double* y, double a, const double* x) { for (int i = 0; i < 20; i++) y[i] += a * x[i]; }

struct cplx { double re, im; };
__device__ __forceinline__ cplx c_mul(cplx a, cplx b) { return cplx{ a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re }; }
__device__ __forceinline__ cplx c_sub(cplx a, cplx b) { return cplx{ a.re - b.re, a.im - b.im }; }
__device__ __forceinline__ cplx c_add(cplx a, cplx b) { return cplx{ a.re + b.re, a.im + b.im }; }
__device__ __forceinline__ cplx c_div(cplx a, cplx b)
{
    double t = 1./(b.re*b.re + b.im*b.im);
    return cplx{ (a.re*b.re + a.im*b.im)*t, (-a.re*b.im + a.im*b.re)*t };
}
// cv::solvePoly, real coefficients c[0..10] (increasing powers) in LDS at `cw`, 300 Durand-Kerner sweeps,
// run by the whole wave: lane i owns root i in registers.
//   * the Horner numerators of a sweep depend only on each root's own old value -> all lanes at once;
//   * the denominators are Gauss-Seidel (root i uses the NEW roots j < i) and OpenCV multiplies the
//     factors in the order j = 0..n-1, so after root k is final every lane i > k takes the factor
//     (p_i - z_k) in one SIMD step, and lane k alone finishes its chain over the old roots j > k.
// Every operation and its order are those of the scalar loop; only independent work moves to other lanes.
// Exact shortcut: once a full sweep leaves every root bitwise unchanged, all later sweeps recompute the
// same corrections from the same roots, so stopping there gives the result of the full 300 iterations.
// Four hypotheses share a wave, one per DPP row of 16 lanes.  row_bcast<K>(v) = v of lane K of the caller's own
// row: one full-rate VALU move per 32 bits (row_newbcast), no LDS round trip on the critical path.
template <int K>
__device__ __forceinline__ double row_bcast(double v)
{
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], 0x150 + K, 0xf, 0xf, false);
    u.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], 0x150 + K, 0xf, 0xf, false);
    return u.d;
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
// denom *= df unless df == 0 (OpenCV skips coincident roots), branch-free
__device__ __forceinline__ cplx dk_factor(cplx denom, cplx df)
{
    cplx pr = c_mul(denom, df);
    bool nz = df.re != 0 || df.im != 0;
    return cplx{ nz ? pr.re : denom.re, nz ? pr.im : denom.im };
}
// One Durand-Kerner sweep of cv::solvePoly for the hypothesis of this row; lane l (0..9) of the row owns root l
// in registers.
//   * the Horner numerators of a sweep depend only on each root's own old value -> all lanes at once;
//   * the denominators are Gauss-Seidel (root i uses the NEW roots j < i) and OpenCV multiplies the factors in
//     the order j = 0..n-1, so after root k is final every lane i > k takes the factor (p_i - z_k) in one SIMD
//     step, and lane k alone finishes its chain over the old roots j > k.
// Every operation and its order are those of the scalar loop; only independent work moves to other lanes.
// NC = 10: the degree is 10 at compile time (the general case); NC = 0: runtime degree n < 10 (may differ per
// row).  CHECKED = false multiplies every factor without OpenCV's "skip a zero difference" test and reports in
// `sawzero` whether any factor that counts was zero; the caller then redoes the sweep CHECKED.
template <int NC, bool CHECKED>
__device__ __forceinline__ cplx dk_sweep(const cplx p, const double* cr, double cn, int n, int l, cplx& q, bool& sawzero)
{
    constexpr int n0 = 10;
#define DK_ON(j) (NC != 0 || (j) < n)
    cplx z = p;
    cplx num{cn, 0};
#pragma unroll
    for (int j = 0; j < n0; j++) if (DK_ON(j)) num = c_add(c_mul(num, p), cplx{cr[j], 0});
    cplx dfo[n0];                                // p - (old root j): the factors lane k multiplies after its own turn
    bool zero = false;
    static_for<1, n0>([&](auto J) {
        constexpr int j = decltype(J)::value;
        dfo[j] = c_sub(p, cplx{ row_bcast<j>(p.re), row_bcast<j>(p.im) });
        if (!CHECKED) zero = zero || (l < j && DK_ON(j) && dfo[j].re == 0 && dfo[j].im == 0);
    });
    cplx denom{cn, 0};
    q = cplx{0, 0};
    static_for<0, n0>([&](auto K) {
        constexpr int k = decltype(K)::value;
        if (l == k && DK_ON(k)) {
#pragma unroll
            for (int j = k + 1; j < n0; j++) if (DK_ON(j)) denom = CHECKED ? dk_factor(denom, dfo[j]) : c_mul(denom, dfo[j]);
            q = c_div(num, denom);
            z = c_sub(p, q);
        }
        if constexpr (k + 1 < n0) {              // lanes > k take the factor of the new root k (other lanes: dead value)
            cplx df = c_sub(p, cplx{ row_bcast<k>(z.re), row_bcast<k>(z.im) });
            if (!CHECKED) zero = zero || (l > k && l < n && df.re == 0 && df.im == 0);
            denom = CHECKED ? dk_factor(denom, df) : c_mul(denom, df);
        }
    });
#undef DK_ON
    sawzero = zero;
    return z;
}
// 300 sweeps, or fewer when exact: OpenCV stops when the largest correction is <= 0; and once a full sweep leaves
// every root bitwise unchanged all later sweeps recompute the same corrections, so stopping there gives the
// result of the full 300 iterations.  The four rows of the wave stop independently (`live`).
// (Round 5 looked for more exits of that kind -- a sweep is a function of the ten roots alone, so a state that recurs bit for bit
// makes the sequence periodic and the state after sweep 300 computable: with Brent's one-saved-state scheme two hypotheses in
// three do recur, with periods from 2 to 126, but a third wander on without recurring inside 300 sweeps, and the launch lasts as long
// as its slowest hypothesis: 912 us with the test against 879 without.  DESIGN.md section 9.)
template <int NC>
__device__ __forceinline__ cplx dk_sweeps(cplx z, const double* cr, double cn, int n, int l, int row)
{
    bool live = true;
#pragma unroll 1
    for (int iter = 0; iter < 300; iter++) {
        cplx q; bool sawzero;
        cplx zn = dk_sweep<NC, false>(z, cr, cn, n, l, q, sawzero);
        if (__any(sawzero)) zn = dk_sweep<NC, true>(z, cr, cn, n, l, q, sawzero);
        bool moved = zn.re != z.re || zn.im != z.im;
        bool nonzero = q.re*q.re + q.im*q.im > 0;                   // sqrt(s) > 0  <=>  s > 0
        unsigned long long bn = __ballot(nonzero && l < n), bm = __ballot(moved && l < n);
        if (live) z = zn;
        bool go = ((bn >> (16 * row)) & 0xffffull) != 0 && ((bm >> (16 * row)) & 0xffffull) != 0;
        live = live && go;
        if (!__any(live)) break;
    }
    return z;
}
// cv::solvePoly, real coefficients c[0..10] (increasing powers) in LDS at `cw`; roots to rre/rim.
__device__ void solve_poly10(const double* cw, double* rre, double* rim, int l, int row)
{
    const int n0 = 10;
    int n = n0;
    for (; n > 1; n--) if (fabs(cw[n]) + fabs(0.0) > DBL_EPSILON) break;
    double cr[n0];                               // cr[j] = c[n-1-j]: the coefficient Horner step j adds
#pragma unroll
    for (int j = 0; j < n0; j++) cr[j] = j < n ? cw[n - 1 - j] : 0.0;
    const double cn = cw[n];
    cplx z{0, 0};
    {
        cplx p{1, 0}; const cplx r{1, 1};
#pragma unroll 1
        for (int i = 0; i < n0; i++) { if (i == l && i < n) z = p; p = c_mul(p, r); }
    }
    z = __all(n == n0) ? dk_sweeps<10>(z, cr, cn, n, l, row) : dk_sweeps<0>(z, cr, cn, n, l, row);
    if (l < n && fabs(z.im) < 1e-100) z.im = 0;
    int src = (l < n ? l : n - 1) + 16 * row;
    double ore = __shfl(z.re, src), oim = __shfl(z.im, src);
    if (l < n0) { rre[l] = ore; rim[l] = oim; }
}
__device__ void pz_mul(const double* a, int da, const double* b, int db, double* out)
{
    for (int i = 0; i <= da + db; i++) out[i] = 0;
    for (int i = 0; i <= da; i++) for (int j = 0; j <= db; j++) out[i + j] += a[i] * b[j];
}

// LDS layout of one hypothesis (doubles)
// (everything from FP_AL on reuses the per-lane polynomial scratch, dead once A is assembled: 15.2 KB per
// workgroup, so the 2000 hypotheses of a RANSAC run are resident in one round on 256 CUs)
enum { FP_AT = 0, FP_W = 81, FP_WT = 90, FP_V5 = 99, FP_E = 124, FP_EET = 304, FP_TR = 484, FP_A = 504, FP_LANE = 704,
       FP_AL = FP_LANE, FP_AINV = FP_AL + 100, FP_AR = FP_AINV + 100, FP_AP = FP_AR + 100, FP_B = FP_AP + 100, FP_C = FP_B + 39,
       FP_ROOTS = FP_C + 11, FP_RT = FP_ROOTS + 20, FP_CAND = FP_RT + 330, FP_FLAG = FP_CAND + 90, FP_POLY = FP_FLAG + 14,
       FP_TOTAL = FP_LANE + 1200 };
static_assert(FP_POLY + 44 <= FP_TOTAL, "five-point LDS overlay");

// diagnostic: 100 MHz wall-clock stamps of hypothesis 0's phases (printed by the host when UVO_DBG_PHASE is set)
__device__ long long g_fp_clk[8];
#define FP_STAMP(i) do { if (hyp == 0 && l == 0) g_fp_clk[i] = wall_clock64(); } while (0)

// Four hypotheses per 64-thread workgroup (one per DPP row of 16 lanes, `l` = lane within the row): the
// Durand-Kerner stage is VALU-issue bound, so sharing every instruction between four hypotheses quarters its
// cost, and the 500 single-wave workgroups of a 2000-iteration RANSAC run all start at once (2 per CU by LDS).
constexpr int kFpPerWg = 4;
__global__ __launch_bounds__(64) void k_fivepoint_hyp(const double* q1, const double* q2, const int* subsets, int nhyp,
                                                      double* models /* nhyp x 10 x 9 */, int* nmodels)
{
    __shared__ double S2[kFpPerWg * FP_TOTAL];
    __shared__ int s_piv[kFpPerWg], s_sing[kFpPerWg];
    const int row = threadIdx.x >> 4, l = threadIdx.x & 15;       // DPP row 0..3 = hypothesis slot
    const int hyp_raw = blockIdx.x * kFpPerWg + row;
    const bool real = hyp_raw < nhyp;                              // a ragged tail recomputes the last hypothesis, writes nothing
    const int hyp = real ? hyp_raw : nhyp - 1;
    double* S = S2 + row * FP_TOTAL;
    using A1 = SArr<1>;
    double* At = S + FP_AT;
    FP_STAMP(0);
    // ---- Q (5 x 9) in the first five rows of the 9 x 9 buffer, rest zero ----
    for (int i = l; i < 81; i += 16) At[i] = 0;
    __syncthreads();
    if (l < 5) {
        int id = subsets[hyp * 5 + l];
        double x1 = q1[2*id], y1 = q1[2*id+1], x2 = q2[2*id], y2 = q2[2*id+1];
        double* r = At + l * 9;
        r[0] = x2*x1; r[1] = x2*y1; r[2] = x2; r[3] = y2*x1; r[4] = y2*y1; r[5] = y2; r[6] = x1; r[7] = y1; r[8] = 1.0;
    }
    __syncthreads();
    // SVD::compute(Q, FULL_UV): rows 5..8 of Vt = JacobiSVD's completion = null-space basis
    if (l == 0) jacobi_svd_rt(A1{At}, A1{S + FP_W}, A1{S + FP_V5}, A1{S + FP_WT}, 9, 5, 9);
    __syncthreads();
    FP_STAMP(1);
    const double* EE = At + 45;                                   // 4 x 9
    double* Ep = S + FP_E;                                         // E[9][20]
    double* EEt = S + FP_EET;                                      // [9][20]
    double* tr = S + FP_TR;
    double* Amat = S + FP_A;                                       // 10 x 20
    double* L = S + FP_LANE + (l < 10 ? l : 0) * 120;              // private: 6 polys
    if (l < 9) {                                                   // p_lin
        double* p = Ep + l * 20;
        p_zero(p);
        p[12] = EE[0*9 + l]; p[15] = EE[1*9 + l]; p[18] = EE[2*9 + l]; p[19] = EE[3*9 + l];
    }
    __syncthreads();
#define EPOLY(i, j) (Ep + ((i)*3 + (j)) * 20)
    if (l < 9) {                                                   // (E E^T)(i,j)
        int i = l / 3, j = l - 3*i;
        double* acc = EEt + l * 20; double* t = L;
        p_zero(acc);
        for (int k = 0; k < 3; k++) { p_mul(EPOLY(i, k), EPOLY(j, k), t); p_axpy(acc, 1.0, t); }
    } else if (l == 9) {                                           // det(E) -> row 0
        double *m0 = L, *m1 = L + 20, *m2 = L + 40, *t = L + 60, *d = L + 80;
        p_mul(EPOLY(1,1), EPOLY(2,2), m0); p_mul(EPOLY(1,2), EPOLY(2,1), t); p_axpy(m0, -1.0, t);
        p_mul(EPOLY(1,0), EPOLY(2,2), m1); p_mul(EPOLY(1,2), EPOLY(2,0), t); p_axpy(m1, -1.0, t);
        p_mul(EPOLY(1,0), EPOLY(2,1), m2); p_mul(EPOLY(1,1), EPOLY(2,0), t); p_axpy(m2, -1.0, t);
        p_mul(EPOLY(0,0), m0, d);
        p_mul(EPOLY(0,1), m1, t); p_axpy(d, -1.0, t);
        p_mul(EPOLY(0,2), m2, t); p_axpy(d, 1.0, t);
        for (int k = 0; k < 20; k++) Amat[k] = d[k];
    }
    __syncthreads();
    if (l == 0) { p_zero(tr); for (int i = 0; i < 3; i++) p_axpy(tr, 1.0, EEt + (i*3 + i) * 20); }
    __syncthreads();
    if (l < 9) {                                                   // 2 E E^T E - trace(E E^T) E
        int i = l / 3, j = l - 3*i;
        double* acc = L + 20; double* t = L;
        p_zero(acc);
        for (int k = 0; k < 3; k++) { p_mul(EEt + (i*3 + k) * 20, EPOLY(k, j), t); p_axpy(acc, 2.0, t); }
        p_mul(tr, EPOLY(i, j), t); p_axpy(acc, -1.0, t);
        for (int k = 0; k < 20; k++) Amat[(1 + l) * 20 + k] = acc[k];
    }
#undef EPOLY
    __syncthreads();
    FP_STAMP(2);
    // ---- A = inv(A[:, :10]) * A[:, 10:]  (LU with partial pivoting, eps = 100*DBL_EPSILON, then gemm) ----
    double *Al = S + FP_AL, *Ainv = S + FP_AINV, *Ar = S + FP_AR, *Ap = S + FP_AP;
    for (int e = l; e < 100; e += 16) { int i = e / 10, j = e - 10*i; Al[e] = Amat[i*20 + j]; Ar[e] = Amat[i*20 + 10 + j]; Ainv[e] = i == j; }
    if (l == 0) s_sing[row] = 0;
    __syncthreads();
    for (int i = 0; i < 10; i++) {                                 // (a singular hypothesis idles through the remaining steps)
        if (l == 0 && !s_sing[row]) {
            int k = i;
            for (int j = i+1; j < 10; j++) if (fabs(Al[j*10 + i]) > fabs(Al[k*10 + i])) k = j;
            s_piv[row] = k;
            if (fabs(Al[k*10 + i]) < DBL_EPSILON * 100) s_sing[row] = 1;
        }
        __syncthreads();
        const bool on = !s_sing[row];
        int k = s_piv[row];
        if (on && k != i && l < 10) {
            if (l >= i) { double t = Al[i*10 + l]; Al[i*10 + l] = Al[k*10 + l]; Al[k*10 + l] = t; }
            double t = Ainv[i*10 + l]; Ainv[i*10 + l] = Ainv[k*10 + l]; Ainv[k*10 + l] = t;
        }
        __syncthreads();
        int j = i + 1 + l;
        if (on && j < 10) {
            double d = -1/Al[i*10 + i];
            double alpha = Al[j*10 + i]*d;
            for (int kk = i+1; kk < 10; kk++) Al[j*10 + kk] += alpha*Al[i*10 + kk];
            for (int kk = 0; kk < 10; kk++) Ainv[j*10 + kk] += alpha*Ainv[i*10 + kk];
        }
        __syncthreads();
    }
    const bool sing = s_sing[row] != 0;
    for (int i = 9; i >= 0; i--) {
        if (!sing && l < 10) {
            double s = Ainv[i*10 + l];
            for (int k = i+1; k < 10; k++) s -= Al[i*10 + k]*Ainv[k*10 + l];
            Ainv[i*10 + l] = s/Al[i*10 + i];
        }
        __syncthreads();
    }
    if (sing) for (int e = l; e < 100; e += 16) Ainv[e] = 0;        // invert() of a singular matrix yields zeros
    __syncthreads();
    for (int e = l; e < 100; e += 16) {
        int i = e / 10, j = e - 10*i;
        double s = 0;
        for (int k = 0; k < 10; k++) s += Ainv[i*10 + k] * Ar[k*10 + j];
        Ap[e] = s;
    }
    __syncthreads();
    FP_STAMP(3);
    double* b = S + FP_B;                                           // 3 x 13
    if (l < 3) {
        const double* a1 = Ap + (l*2 + 4) * 10; const double* a2 = Ap + (l*2 + 5) * 10;
        double row1[13], row2[13];
        for (int k = 0; k < 13; k++) row1[k] = row2[k] = 0;
        for (int k = 0; k < 3; k++) { row1[1 + k] = a1[k]; row1[5 + k] = a1[3 + k]; }
        for (int k = 0; k < 4; k++) row1[9 + k] = a1[6 + k];
        for (int k = 0; k < 3; k++) { row2[k] = a2[k]; row2[4 + k] = a2[3 + k]; }
        for (int k = 0; k < 4; k++) row2[8 + k] = a2[6 + k];
        for (int k = 0; k < 13; k++) b[l*13 + k] = row1[k] - row2[k];
    }
    __syncthreads();
    double* rre = S + FP_ROOTS; double* rim = rre + 10;
    if (l == 0) {                                                   // det B(z)
        double e[3][3][5], t1[12], t2[12], m[12];
        double* c = S + FP_POLY;
        for (int r = 0; r < 3; r++) {
            for (int k = 0; k < 4; k++) { e[r][0][k] = b[r*13 + 3 - k]; e[r][1][k] = b[r*13 + 7 - k]; }
            e[r][0][4] = e[r][1][4] = 0;
            for (int k = 0; k < 5; k++) e[r][2][k] = b[r*13 + 12 - k];
        }
        for (int k = 0; k < 11; k++) c[k] = 0;
        pz_mul(e[1][1], 3, e[2][2], 4, t1); pz_mul(e[1][2], 4, e[2][1], 3, t2);
        for (int k = 0; k <= 7; k++) m[k] = t1[k] - t2[k];
        pz_mul(e[0][0], 3, m, 7, t1); for (int k = 0; k <= 10; k++) c[k] += t1[k];
        pz_mul(e[1][0], 3, e[2][2], 4, t1); pz_mul(e[1][2], 4, e[2][0], 3, t2);
        for (int k = 0; k <= 7; k++) m[k] = t1[k] - t2[k];
        pz_mul(e[0][1], 3, m, 7, t1); for (int k = 0; k <= 10; k++) c[k] -= t1[k];
        pz_mul(e[1][0], 3, e[2][1], 3, t1); pz_mul(e[1][1], 3, e[2][0], 3, t2);
        for (int k = 0; k <= 6; k++) m[k] = t1[k] - t2[k];
        pz_mul(e[0][2], 4, m, 6, t1); for (int k = 0; k <= 10; k++) c[k] += t1[k];
    }
    __syncthreads();
    FP_STAMP(4);
    solve_poly10(S + FP_POLY, rre, rim, l, row);                  // its roots
    __syncthreads();
    FP_STAMP(5);
    double* cand = S + FP_CAND; double* flag = S + FP_FLAG;
    if (l < 10) {                                                   // one real root per lane
        flag[l] = 0;
        if (!(fabs(rim[l]) > 1e-10)) {
            double z1 = rre[l], z2 = z1*z1, z3 = z2*z1, z4 = z3*z1;
            double* R = S + FP_RT + l * 33;                         // bz 9 | at 9 | w 3 | vt 9 | wt 3
            for (int j = 0; j < 3; j++) {
                const double* br = b + j*13;
                R[j*3 + 0] = br[0]*z3 + br[1]*z2 + br[2]*z1 + br[3];
                R[j*3 + 1] = br[4]*z3 + br[5]*z2 + br[6]*z1 + br[7];
                R[j*3 + 2] = br[8]*z4 + br[9]*z3 + br[10]*z2 + br[11]*z1 + br[12];
            }
            svd_square<3>(A1{R}, A1{R + 9}, A1{R + 18}, A1{R + 21}, A1{R + 30});
            const double* vt = R + 21;
            if (!(fabs(vt[8]) < 1e-10)) {
                double xs = vt[6] / vt[8], ys = vt[7] / vt[8];
                double Ev[9], nrm = 0;
                for (int k = 0; k < 9; k++) Ev[k] = EE[0*9 + k]*xs + EE[1*9 + k]*ys + EE[2*9 + k]*z1 + EE[3*9 + k];
                for (int k = 0; k < 9; k++) nrm += Ev[k]*Ev[k];
                nrm = sqrt(nrm);
                for (int k = 0; k < 9; k++) cand[l*9 + k] = Ev[k] / nrm;
                flag[l] = 1;
            }
        }
    }
    __syncthreads();
    if (l == 0 && real) {
        int count = 0;
        double* out = models + (size_t)hyp * 90;
        for (int i = 0; i < 10; i++) if (flag[i] != 0) { for (int k = 0; k < 9; k++) out[count*9 + k] = cand[i*9 + k]; count++; }
        nmodels[hyp] = count;
    }
    FP_STAMP(6);
}

// ------------------------------------------------------------------------------------------
// scoring: one workgroup per (hypothesis, model); mode 0 = inlier count (RANSAC), 1 = median (LMedS)
// ------------------------------------------------------------------------------------------
__device__ void block_sort_median(float* s_err, int n, int npow2, double* out)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int k = 2; k <= npow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npow2; i += nt) {
                int ixj = i ^ j;
                if (ixj > i) {
                    float a = s_err[i], b = s_err[ixj];
                    bool up = (i & k) == 0;
                    if ((a > b) == up) { s_err[i] = b; s_err[ixj] = a; }
                }
            }
            __syncthreads();
        }
    if (tid == 0) *out = n % 2 != 0 ? (double)s_err[n/2] : (s_err[n/2 - 1] + s_err[n/2]) * 0.5;
}

__global__ __launch_bounds__(256) void k_e_score(const double* q1, const double* q2, int n, const double* models, const int* nmodels,
                                                 int mode, float thr2, int npow2, int* counts, double* medians, int* nmodels_host = nullptr)
{
    extern __shared__ float s_err[];
    __shared__ int s_cnt;
    const int m = blockIdx.x, hyp = blockIdx.y, tid = threadIdx.x;
    if (m == 0 && tid == 0 && nmodels_host) nmodels_host[hyp] = nmodels[hyp];       // (counts / medians may be pinned host memory too: the host's scan reads them without a copy)
    if (m >= nmodels[hyp]) return;
    double E[9];
    for (int k = 0; k < 9; k++) E[k] = models[(size_t)hyp * 90 + m * 9 + k];
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    int cnt = 0;
    for (int i = tid; i < (mode ? npow2 : n); i += 256) {
        float e = INFINITY;
        if (i < n) e = sampson_error1(E, q1[2*i], q1[2*i+1], q2[2*i], q2[2*i+1]);
        if (mode) s_err[i] = e; else cnt += (i < n && e <= thr2) ? 1 : 0;
    }
    if (mode == 0) {
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
        if ((tid & 63) == 0) atomicAdd(&s_cnt, cnt);
        __syncthreads();
        if (tid == 0) counts[hyp * 10 + m] = s_cnt;
    } else {
        __syncthreads();
        block_sort_median(s_err, n, npow2, &medians[hyp * 10 + m]);
    }
}

// HomographyEstimatorCallback::runKernel for one 4-point subset, SIXTEEN LANES PER HYPOTHESIS (one DPP row; four hypotheses per wave).
// Until round 5 a thread solved a hypothesis alone: the 9 x 9 Jacobi eigen-solver (JacobiImpl_: ~150 rotations, each a pivot search over
// 16 candidates, two hypots, ~25 two-element rotations and four index rescans, every operand an LDS round trip) kept a lane busy for
// 0.70 ms.  Nothing in it is sequential except the rotations themselves:
//   * the pivot search "first largest |a| among A[i][indR[i]] (i = 0..7), then A[indC[i]][i] (i = 1..8)" is an arg-max over 16 candidates
//     with ties to the earlier one -- one candidate per lane, four rotate-and-compare steps inside the row;
//   * the rotation touches element pairs that are independent of each other: lane i < 9 takes the pair of A that carries index i and
//     the pair (V[k][i], V[l][i]);
//   * indR / indC are rescanned for the two pivot rows only (as OpenCV does -- the stale entries of other rows are part of the
//     algorithm): four arg-maxes over <= 8 lanes, ties to the lower index.
// Every floating-point operation and comparison is the scalar loop's (uvo_mono.h: jacobi_eigen, homography_kernel -- what the host
// refit runs); only independent work moved to other lanes.  The rows of a wave stop independently (`live`).
static const int kHPerWg = 4;
template <int D>
__device__ __forceinline__ double row_ror_d(double v)
{
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], 0x120 + D, 0xf, 0xf, false);
    u.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], 0x120 + D, 0xf, 0xf, false);
    return u.d;
}
template <int D>
__device__ __forceinline__ int row_ror_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x120 + D, 0xf, 0xf, false); }
// all 16 lanes of the row end with the largest v and, among equal v, the smallest c
__device__ __forceinline__ void row_argmax16(double& v, int& c)
{
#define UVO_AM_STEP(D) { const double ov = row_ror_d<D>(v); const int oc = row_ror_i<D>(c); if (ov > v || (ov == v && oc < c)) { v = ov; c = oc; } }
    UVO_AM_STEP(1) UVO_AM_STEP(2) UVO_AM_STEP(4) UVO_AM_STEP(8)
#undef UVO_AM_STEP
}
__global__ __launch_bounds__(64) void k_h_hyp(const float* src, const float* dst, const int* subsets, int nhyp,
                                              double* models /* nhyp x 9 */, int* nmodels)
{
    constexpr int n = 9;
    __shared__ double s_A[kHPerWg][81], s_V[kHPerWg][81], s_W[kHPerWg][9];
    __shared__ int s_indR[kHPerWg][9], s_indC[kHPerWg][9];
    const int row = threadIdx.x >> 4, l = threadIdx.x & 15;
    const int hyp_raw = blockIdx.x * kHPerWg + row;
    const bool real = hyp_raw < nhyp;                               // a ragged tail recomputes the last hypothesis, writes nothing
    const int hyp = real ? hyp_raw : nhyp - 1;
    double* A = s_A[row]; double* V = s_V[row]; double* W = s_W[row];
    int* indR = s_indR[row]; int* indC = s_indC[row];
    float M[8], m[8];
    for (int i = 0; i < 4; i++) {
        const int id = subsets[hyp * 4 + i];
        M[2*i] = src[2*id]; M[2*i+1] = src[2*id+1]; m[2*i] = dst[2*id]; m[2*i+1] = dst[2*id+1];
    }
    // ---- normalisation (every lane the same scalars) ----
    const int count = 4;
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < count; i++) { cmx += m[2*i]; cmy += m[2*i+1]; cMx += M[2*i]; cMy += M[2*i+1]; }
    cmx /= count; cmy /= count; cMx /= count; cMy /= count;
    for (int i = 0; i < count; i++) {
        smx += fabs(m[2*i] - cmx); smy += fabs(m[2*i+1] - cmy);
        sMx += fabs(M[2*i] - cMx); sMy += fabs(M[2*i+1] - cMy);
    }
    const bool degenerate = fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON;
    smx = count/smx; smy = count/smy; sMx = count/sMx; sMy = count/sMy;
    // ---- L^T L: the rows Lx, Ly of the four points into V (scratch until the solver initialises it), then entry (j, k >= j) = the sum over
    // the points in their order, 45 entries over the lanes ----
    if (l < count) {
        const int i = l;
        const double x = (m[2*i] - cmx)*smx, y = (m[2*i+1] - cmy)*smy;
        const double X = (M[2*i] - cMx)*sMx, Y = (M[2*i+1] - cMy)*sMy;
        const double Lx[9] = { X, Y, 1, 0, 0, 0, -x*X, -x*Y, -x };
        const double Ly[9] = { 0, 0, 0, X, Y, 1, -y*X, -y*Y, -y };
        for (int j = 0; j < 9; j++) { V[i*18 + j] = Lx[j]; V[i*18 + 9 + j] = Ly[j]; }
    }
    __syncthreads();
    for (int e = l; e < 81; e += 16) {
        const int j = e / 9, k = e - 9 * j;
        if (k < j) continue;
        double acc = 0;
        for (int i = 0; i < count; i++) acc += V[i*18 + j]*V[i*18 + k] + V[i*18 + 9 + j]*V[i*18 + 9 + k];
        A[j*9 + k] = acc; A[k*9 + j] = acc;
    }
    __syncthreads();
    // ---- JacobiImpl_<double>(A, 9, W, V) ----
    const double eps = DBL_EPSILON;
    for (int e = l; e < 81; e += 16) V[e] = (e % 10) == 0 ? 1.0 : 0.0;
    if (l < n) {
        const int k = l;
        W[k] = A[(n + 1)*k];
        if (k < n - 1) { int mi = k + 1; double mv = fabs(A[n*k + mi]); for (int i = k + 2; i < n; i++) { const double val = fabs(A[n*k + i]); if (mv < val) mv = val, mi = i; } indR[k] = mi; }
        if (k > 0) { int mi = 0; double mv = fabs(A[k]); for (int i = 1; i < k; i++) { const double val = fabs(A[n*i + k]); if (mv < val) mv = val, mi = i; } indC[k] = mi; }
    }
    __syncthreads();
    bool live = !degenerate;                                        // (a degenerate subset returns before the solver in the reference: no model)
#pragma unroll 1
    for (int iters = 0; iters < n*n*30; iters++) {
        // the pivot: candidate l < 8 is A[l][indR[l]], candidate l >= 8 is A[indC[l - 7]][l - 7]
        int ck, cl;
        if (l < 8) { ck = l; cl = indR[l]; } else { cl = l - 7; ck = indC[cl]; }
        double val = fabs(A[n*ck + cl]);
        int c = l;
        row_argmax16(val, c);
        int k, pl;
        if (c < 8) { k = c; pl = indR[c]; } else { pl = c - 7; k = indC[pl]; }
        const double p = A[n*k + pl];
        live = live && !(fabs(p) <= eps);
        if (!__any(live)) break;
        const double Wk = W[k], Wl = W[pl];
        const double y = (Wl - Wk)*0.5;
        double t = fabs(y) + det_hypot(p, y);
        double s = det_hypot(p, t);
        const double cc = t/s;
        s = p/s; t = (p/t)*p;
        if (y < 0) s = -s, t = -t;
        // the element pairs lane l rotates
        int a_i0 = -1, a_i1 = -1;
        if (l < n && l != k && l != pl) {
            if (l < k) { a_i0 = n*l + k; a_i1 = n*l + pl; }
            else if (l < pl) { a_i0 = n*k + l; a_i1 = n*l + pl; }
            else { a_i0 = n*k + l; a_i1 = n*pl + l; }
        }
        double a0 = 0, b0 = 0, v0 = 0, v1 = 0;
        if (a_i0 >= 0) { a0 = A[a_i0]; b0 = A[a_i1]; }
        if (l < n) { v0 = V[n*k + l]; v1 = V[n*pl + l]; }
        __syncthreads();                                            // every lane holds what it reads of the old matrix
        if (live) {
            if (l == 0) { A[n*k + pl] = 0; W[k] = Wk - t; W[pl] = Wl + t; }
            if (a_i0 >= 0) { A[a_i0] = a0*cc - b0*s; A[a_i1] = a0*s + b0*cc; }
            if (l < n) { V[n*k + l] = v0*cc - v1*s; V[n*pl + l] = v0*s + v1*cc; }
        }
        __syncthreads();
        // indR / indC of the two pivot rows from the rotated matrix
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int idx = j == 0 ? k : pl;
            double vr = -1.0, vc = -1.0;
            int ir = l, ic = l;
            if (l < n && l > idx) vr = fabs(A[n*idx + l]);
            if (l < idx) vc = fabs(A[n*l + idx]);
            row_argmax16(vr, ir);
            row_argmax16(vc, ic);
            if (live && l == 0) { if (idx < n - 1) indR[idx] = ir; if (idx > 0) indC[idx] = ic; }
        }
        __syncthreads();
    }
    __syncthreads();
    // ---- the eigenvalue sort (selection sort, first largest wins; rows of V swapped with it): which row ends up last; then de-normalise ----
    if (l == 0 && real) {
        int perm[9];
        double w[9];
        for (int i = 0; i < n; i++) { perm[i] = i; w[i] = W[i]; }
        for (int k = 0; k < n - 1; k++) {
            int mi = k;
            for (int i = k + 1; i < n; i++) if (w[mi] < w[i]) mi = i;
            if (k != mi) { const double t = w[mi]; w[mi] = w[k]; w[k] = t; const int q = perm[mi]; perm[mi] = perm[k]; perm[k] = q; }
        }
        const double* H0 = V + 9 * perm[8];
        const double invHnorm[9] = { 1./smx, 0, cmx, 0, 1./smy, cmy, 0, 0, 1 };
        const double Hnorm2[9] = { sMx, 0, -cMx*sMx, 0, sMy, -cMy*sMy, 0, 0, 1 };
        double Ht[9], H1[9];
        for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++)
            Ht[r*3 + q] = invHnorm[r*3]*H0[q] + invHnorm[r*3+1]*H0[3 + q] + invHnorm[r*3+2]*H0[6 + q];
        for (int r = 0; r < 3; r++) for (int q = 0; q < 3; q++)
            H1[r*3 + q] = Ht[r*3]*Hnorm2[q] + Ht[r*3+1]*Hnorm2[3 + q] + Ht[r*3+2]*Hnorm2[6 + q];
        const double sc = 1./H1[8];
        nmodels[hyp] = degenerate ? 0 : 1;
        if (!degenerate) for (int q = 0; q < 9; q++) models[(size_t)hyp * 9 + q] = H1[q]*sc;
    }
}

__global__ __launch_bounds__(256) void k_h_score(const float* src, const float* dst, int n, const double* models, const int* nmodels,
                                                 int mode, float thr2, int npow2, int* counts, double* medians)
{
    extern __shared__ float s_err[];
    __shared__ int s_cnt;
    const int hyp = blockIdx.x, tid = threadIdx.x;
    if (nmodels[hyp] <= 0) return;
    float Hf[8];
    for (int k = 0; k < 8; k++) Hf[k] = (float)models[(size_t)hyp * 9 + k];
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    int cnt = 0;
    for (int i = tid; i < (mode ? npow2 : n); i += 256) {
        float e = INFINITY;
        if (i < n) e = homography_error1(Hf, src[2*i], src[2*i+1], dst[2*i], dst[2*i+1]);
        if (mode) s_err[i] = e; else cnt += (i < n && e <= thr2) ? 1 : 0;
    }
    if (mode == 0) {
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
        if ((tid & 63) == 0) atomicAdd(&s_cnt, cnt);
        __syncthreads();
        if (tid == 0) counts[hyp] = s_cnt;
    } else {
        __syncthreads();
        block_sort_median(s_err, n, npow2, &medians[hyp]);
    }
}

// findInliers for one model: kind 0 = essential (q1,q2 doubles), 1 = homography (floats)
__global__ void k_model_mask(int kind, const double* q1, const double* q2, const float* src, const float* dst, int n,
                             const double* model, float thr2, uint8_t* mask)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float e;
    if (kind == 0) e = sampson_error1(model, q1[2*i], q1[2*i+1], q2[2*i], q2[2*i+1]);
    else { float Hf[8]; for (int k = 0; k < 8; k++) Hf[k] = (float)model[k]; e = homography_error1(Hf, src[2*i], src[2*i+1], dst[2*i], dst[2*i+1]); }
    mask[i] = (uint8_t)(e <= thr2);
}

// recoverPose: candidate c = blockIdx.y, triangulatePoints (double points) + the cheirality / distance tests
struct PoseCands { double P[4][12]; };
static const int kRpThreads = 64;
__global__ __launch_bounds__(kRpThreads) void k_recover_pose(const double* q1, const double* q2, int n, PoseCands pc, const uint8_t* mask_in,
                                                             uint8_t* masks /* 4 x n */, int* good /* 4 */)
{
    __shared__ double lds[(16 + 16 + 16 + 4 + 4) * kRpThreads];
    const int c = blockIdx.y, i = blockIdx.x * kRpThreads + threadIdx.x;
    if (i >= n) return;
    using A = SArr<kRpThreads>;
    A Am{lds + threadIdx.x}, At = Am + 16, Vt = Am + 32, W = Am + 48, Wt = Am + 52;
    const double* P = pc.P[c];
    const double xa = q1[2*i], ya = q1[2*i+1], xb = q2[2*i], yb = q2[2*i+1];
    // P0 = [I | 0]
    Am[0] = xa*0. - 1.; Am[1] = xa*0. - 0.; Am[2] = xa*1. - 0.; Am[3] = xa*0. - 0.;
    Am[4] = ya*0. - 0.; Am[5] = ya*0. - 1.; Am[6] = ya*1. - 0.; Am[7] = ya*0. - 0.;
    for (int k = 0; k < 4; k++) { Am[8 + k] = xb * P[8 + k] - P[k]; Am[12 + k] = yb * P[8 + k] - P[4 + k]; }
    svd_square<4>(Am, At, W, Vt, Wt);
    double X = Vt[12], Y = Vt[13], Z = Vt[14], Wv = Vt[15];
    const double distanceThresh = 50;
    bool m = Z * Wv > 0;
    X /= Wv; Y /= Wv; Z /= Wv; Wv /= Wv;
    m = (Z < distanceThresh) && m;
    double z2 = P[8]*X + P[9]*Y + P[10]*Z + P[11]*Wv;
    m = (z2 > 0) && m;
    m = (z2 < distanceThresh) && m;
    m = m && mask_in[i];
    masks[(size_t)c * n + i] = m ? 1 : 0;
    if (m) atomicAdd(&good[c], 1);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// Host staging in pinned memory: a copy to or from pageable memory is staged by the runtime and blocks the caller (15-25 us each, and
// the pose stage makes about twenty of them per frame); pinned copies are queued like kernels.  Sized once, used like a vector.
template <class T>
struct Pinned {
    T* p = nullptr; size_t n = 0;
    bool resize(size_t count) { if (count <= n) return true; if (p) (void)hipHostFree(p); p = nullptr; n = 0; if (hipHostMalloc(reinterpret_cast<void**>(&p), sizeof(T) * count) != hipSuccess) return false; n = count; return true; }
    T* data() { return p; } const T* data() const { return p; }
    T& operator[](size_t i) { return p[i]; } const T& operator[](size_t i) const { return p[i]; }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; n = 0; }
};
struct MonoWs {                       // device workspace of the mono stage, owned by the context
    double *q1 = nullptr, *q2 = nullptr;           // normalised points (cap x 2)
    float *src = nullptr, *dst = nullptr;          // pixel points (cap x 2)
    int* subsets = nullptr;                        // kMaxHyp x 5
    double* models = nullptr;                      // kMaxHyp x 10 x 9
    int* nmodels = nullptr;                        // kMaxHyp
    int* counts = nullptr;  double* medians = nullptr;   // kMaxHyp x 10
    uint8_t* masks = nullptr;                      // 5 x cap
    int* good = nullptr;                           // 4
    double* best = nullptr;                        // 9
    bool h_one_round = false;                      // findHomography's RANSAC: the last scan went on past the first round's hypotheses, so the next
                                                   // call solves them all in one launch (a launch takes 0.35 ms whether it carries 128 or 2000)
    Pinned<int> h_subsets, h_nmodels, h_counts;
    Pinned<double> h_medians, h_models;
    Pinned<double> h_q1, h_q2;                     // normalised points on their way to q1 / q2
    Pinned<float> h_src, h_dst;                    // pixel points on their way to src / dst
    Pinned<uint8_t> h_mask;                        // masks in both directions
    Pinned<int> h_good;                            // recoverPose's four counts
    // the loop's pose stage with its points resident on the device (mono_essential_resident)
    uvo_point2f *in1 = nullptr, *in2 = nullptr;    // extract_inliers' output (cap each)
    int* pick = nullptr;                           // [0] best candidate of recoverPose, [1] n_in
    Pinned<uvo_point2f> h_x1, h_x2;                // the frame's matched points, mirrored by k_mono_prep
    Pinned<int> h_prep;                            // k_mono_prep: [0] use_essential of select_estimation_method, [1] M it saw
    Pinned<int> h_pick;                            // k_mono_pick: best, good[4], n_in, valid inliers; k_mono_scale: G, n_front
    Pinned<double> h_zs;                           // z of the points convert_3Dpoints_camera keeps (cap)
};

void mono_ws_free(Ctx* c);
static MonoWs* mono_ws(Ctx* c)
{
    if (c->mono_ws) return static_cast<MonoWs*>(c->mono_ws);
    MonoWs* w = new MonoWs();
    size_t cap = (size_t)c->cap;
    bool ok = hipMalloc((void**)&w->q1, sizeof(double) * 2 * cap) == hipSuccess && hipMalloc((void**)&w->q2, sizeof(double) * 2 * cap) == hipSuccess &&
              hipMalloc((void**)&w->src, sizeof(float) * 2 * cap) == hipSuccess && hipMalloc((void**)&w->dst, sizeof(float) * 2 * cap) == hipSuccess &&
              hipMalloc((void**)&w->subsets, sizeof(int) * kMaxHyp * 5) == hipSuccess &&
              hipMalloc((void**)&w->models, sizeof(double) * kMaxHyp * 90) == hipSuccess &&
              hipMalloc((void**)&w->nmodels, sizeof(int) * kMaxHyp) == hipSuccess &&
              hipMalloc((void**)&w->counts, sizeof(int) * kMaxHyp * 10) == hipSuccess &&
              hipMalloc((void**)&w->medians, sizeof(double) * kMaxHyp * 10) == hipSuccess &&
              hipMalloc((void**)&w->masks, 5 * cap) == hipSuccess && hipMalloc((void**)&w->good, sizeof(int) * 4) == hipSuccess &&
              hipMalloc((void**)&w->best, sizeof(double) * 9) == hipSuccess &&
              hipMalloc((void**)&w->in1, sizeof(uvo_point2f) * cap) == hipSuccess && hipMalloc((void**)&w->in2, sizeof(uvo_point2f) * cap) == hipSuccess &&
              hipMalloc((void**)&w->pick, sizeof(int) * 4) == hipSuccess;
    if (!ok) { delete w; return nullptr; }
    ok = w->h_subsets.resize(kMaxHyp * 5) && w->h_nmodels.resize(kMaxHyp) && w->h_counts.resize(kMaxHyp * 10) &&
         w->h_medians.resize(kMaxHyp * 10) && w->h_models.resize((size_t)kMaxHyp * 90) &&
         w->h_q1.resize(2 * cap + 2) && w->h_q2.resize(2 * cap + 2) && w->h_src.resize(2 * cap) && w->h_dst.resize(2 * cap) &&
         w->h_mask.resize(cap) && w->h_good.resize(4) && w->h_x1.resize(cap) && w->h_x2.resize(cap) && w->h_prep.resize(4) && w->h_pick.resize(16) &&
         w->h_zs.resize(cap);
    if (!ok) { c->mono_ws = w; mono_ws_free(c); return nullptr; }
    c->mono_ws = w;
    return w;
}
void mono_ws_free(Ctx* c)
{
    MonoWs* w = static_cast<MonoWs*>(c->mono_ws);
    if (!w) return;
    void* ptrs[] = { w->q1, w->q2, w->src, w->dst, w->subsets, w->models, w->nmodels, w->counts, w->medians, w->masks, w->good, w->best, w->in1, w->in2, w->pick };
    for (void* p : ptrs) (void)hipFree(p);
    w->h_subsets.release(); w->h_nmodels.release(); w->h_counts.release(); w->h_medians.release(); w->h_models.release();
    w->h_q1.release(); w->h_q2.release(); w->h_src.release(); w->h_dst.release(); w->h_mask.release(); w->h_good.release();
    w->h_x1.release(); w->h_x2.release(); w->h_prep.release(); w->h_pick.release(); w->h_zs.release();
    delete w;
    c->mono_ws = nullptr;
}

static int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

// HomographyEstimatorCallback::checkSubset (host, on the sampled points)
static bool have_collinear_points(const float* m, int count)
{
    int j, k, i = count - 1;
    for (j = 0; j < i; j++) {
        double dx1 = m[2*j] - m[2*i], dy1 = m[2*j+1] - m[2*i+1];
        for (k = 0; k < j; k++) {
            double dx2 = m[2*k] - m[2*i], dy2 = m[2*k+1] - m[2*i+1];
            if (fabs(dx2*dy1 - dy2*dx1) <= FLT_EPSILON*(fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return true;
        }
    }
    return false;
}
static double det3d(double a0, double a1, double a2, double a3, double a4, double a5, double a6, double a7, double a8)
{
    return a0*(a4*a8 - a5*a7) - a1*(a3*a8 - a5*a6) + a2*(a3*a7 - a4*a6);
}
static bool h_check_subset(const float* ms1, const float* ms2, int count)
{
    if (have_collinear_points(ms1, count) || have_collinear_points(ms2, count)) return false;
    if (count == 4) {
        static const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
        int negative = 0;
        for (int i = 0; i < 4; i++) {
            const int* t = tt[i];
            double dA = det3d(ms1[2*t[0]], ms1[2*t[0]+1], 1., ms1[2*t[1]], ms1[2*t[1]+1], 1., ms1[2*t[2]], ms1[2*t[2]+1], 1.);
            double dB = det3d(ms2[2*t[0]], ms2[2*t[0]+1], 1., ms2[2*t[1]], ms2[2*t[1]+1], 1., ms2[2*t[2]], ms2[2*t[2]+1], 1.);
            negative += dA*dB < 0;
        }
        if (negative != 0 && negative != 4) return false;
    }
    return true;
}

// getSubset replay for `niters` iterations.  Returns the number of subsets produced (< niters only when an
// attempt budget ran out; *failed_first tells whether that happened on the first iteration).
static int make_subsets(int* out, int niters, int model_points, int count, int maxAttempts,
                        const float* src, const float* dst, bool* failed_first)
{
    uint64_t state = (uint64_t)-1;
    *failed_first = false;
    for (int it = 0; it < niters; it++) {
        int* idx = out + it * model_points;
        bool found = false;
        for (int iters = 0; iters < maxAttempts && !found; ++iters) {
            float ms1[8], ms2[8];
            for (int i = 0; i < model_points; ++i) {
                int idx_i;
                for (;;) {
                    idx_i = (int)(rng_next(state) % (uint32_t)count);
                    bool dup = false;
                    for (int q = 0; q < i; q++) dup = dup || idx[q] == idx_i;
                    if (!dup) break;
                }
                idx[i] = idx_i;
                if (src) { ms1[2*i] = src[2*idx_i]; ms1[2*i+1] = src[2*idx_i+1]; ms2[2*i] = dst[2*idx_i]; ms2[2*i+1] = dst[2*idx_i+1]; }
            }
            found = !src || h_check_subset(ms1, ms2, model_points);
        }
        if (!found) { if (it == 0) *failed_first = true; return it; }
    }
    return niters;
}

// sequential replays over the device scores
struct Winner { int hyp = -1, model = 0; double min_median = DBL_MAX; int max_good = 0; };
static Winner replay_ransac(const int* nmodels, const int* counts, int stride, int nsub, int niters0, int count, int modelPoints, double confidence,
                            int* niters_out = nullptr)
{
    Winner w; int niters = niters0;
    for (int iter = 0; iter < niters && iter < nsub; iter++) {
        for (int i = 0; i < nmodels[iter]; i++) {
            int goodCount = counts[iter * stride + i];
            if (goodCount > (w.max_good > modelPoints - 1 ? w.max_good : modelPoints - 1)) {
                w.hyp = iter; w.model = i; w.max_good = goodCount;
                niters = ransac_update_num_iters(confidence, (double)(count - goodCount) / count, modelPoints, niters);
            }
        }
    }
    if (niters_out) *niters_out = niters;          // the iteration count the adaptive rule has settled on so far
    return w;
}
static Winner replay_lmeds(const int* nmodels, const double* medians, int stride, int nsub)
{
    Winner w;
    for (int iter = 0; iter < nsub; iter++)
        for (int i = 0; i < nmodels[iter]; i++) {
            double median = medians[iter * stride + i];
            if (median < w.min_median) { w.min_median = median; w.hyp = iter; w.model = i; }
        }
    return w;
}

// cv::findEssentialMat on host points (VOU:147)
uvo_status mono_find_essential(Ctx* c, const uvo_point2f* p1, const uvo_point2f* p2, int n, const double* K, int method,
                               double prob, double threshold, int maxIters, double* E, uint8_t* mask, int* ok)
{
    *ok = 0;
    memset(mask, 0, n);
    if (n > c->cap) { c->err = "point count exceeds the context's max_kpts"; return UVO_CAPACITY; }
    MonoWs* w = mono_ws(c);
    if (!w) { c->err = "mono workspace allocation failed"; return UVO_HIP_ERROR; }
    const int modelPoints = 5;
    if (n < modelPoints) return UVO_OK;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const double ax = 1. / fx, bx = -cx * ax, ay = 1. / fy, by = -cy * ay;
    Pinned<double>& q1 = w->h_q1; Pinned<double>& q2 = w->h_q2;
    for (int i = 0; i < n; i++) {
        q1[2*i] = p1[i].x * ax + bx; q1[2*i+1] = p1[i].y * ay + by;
        q2[2*i] = p2[i].x * ax + bx; q2[2*i+1] = p2[i].y * ay + by;
    }
    threshold /= (fx + fy) / 2;
    hipStream_t st = c->stream;
    UVO_HIP_TRY(c, hipMemcpyAsync(w->q1, q1.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice, st));
    UVO_HIP_TRY(c, hipMemcpyAsync(w->q2, q2.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice, st));
    const bool lmeds = method != 8;
    int niters;
    if (n == modelPoints) niters = 1;
    else if (lmeds) { niters = ransac_update_num_iters(prob, 0.45, modelPoints, maxIters); niters = niters > 3 ? niters : 3; }
    else niters = maxIters > 1 ? maxIters : 1;
    if (niters > kMaxHyp) { c->err = "max_iters exceeds the compiled hypothesis capacity (2048)"; return UVO_CAPACITY; }
    bool failed_first = false;
    int nsub;
    if (n == modelPoints) { for (int i = 0; i < 5; i++) w->h_subsets[i] = i; nsub = 1; }
    else nsub = make_subsets(w->h_subsets.data(), niters, modelPoints, n, lmeds ? 1000 : 10000, nullptr, nullptr, &failed_first);
    UVO_HIP_TRY(c, hipMemcpyAsync(w->subsets, w->h_subsets.data(), sizeof(int) * 5 * nsub, hipMemcpyHostToDevice, st));
    // RANSAC hypotheses are solved and scored in rounds: the first kFirstRound subsets, then -- only if the adaptive iteration count
    // still reaches past them after the replayed scan -- the rest.  The scan visits the same counts in the same order either
    // way.  (A five-point solve keeps a SIMD busy for 0.85 ms whether 32 or 500 waves run, so the first round's latency is the
    // same, but it leaves the chip to the other frames of a pipeline and scores 16x fewer models.)  LMedS needs all of them.
    const int kFirstRound = 128;
    const int first = (!lmeds && n != modelPoints && nsub > kFirstRound) ? kFirstRound : nsub;
    hipLaunchKernelGGL(k_fivepoint_hyp, dim3((first + kFpPerWg - 1) / kFpPerWg), dim3(64), 0, st, w->q1, w->q2, w->subsets, first, w->models, w->nmodels);
    UVO_HIP_TRY(c, hipGetLastError());
    if (getenv("UVO_DBG_PHASE")) {
        long long clk[8];
        UVO_HIP_TRY(c, host_sync(c, st));
        UVO_HIP_TRY(c, hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_fp_clk), sizeof(clk)));
        fprintf(stderr, "[uvo] five-point phases (us): svd %.1f polys %.1f lu %.1f detB %.1f roots %.1f solveZ %.1f | nsub %d\n",
                (clk[1]-clk[0])*0.01, (clk[2]-clk[1])*0.01, (clk[3]-clk[2])*0.01, (clk[4]-clk[3])*0.01, (clk[5]-clk[4])*0.01,
                (clk[6]-clk[5])*0.01, nsub);
    }
    if (n == modelPoints) {
        UVO_HIP_TRY(c, hipMemcpyAsync(w->h_nmodels.data(), w->nmodels, sizeof(int), hipMemcpyDeviceToHost, st));
        UVO_HIP_TRY(c, hipMemcpyAsync(w->h_models.data(), w->models, sizeof(double) * 9, hipMemcpyDeviceToHost, st));
        UVO_HIP_TRY(c, host_sync(c, st));
        memcpy(E, w->h_models.data(), sizeof(double) * 9);
        if (w->h_nmodels[0] <= 0) return UVO_OK;
        memset(mask, 1, n); *ok = 1;
        return UVO_OK;
    }
    const int npow2 = next_pow2(n);
    const float thr2 = (float)(threshold * threshold);
    hipLaunchKernelGGL(k_e_score, dim3(10, first), dim3(256), lmeds ? sizeof(float) * npow2 : 0, st, w->q1, w->q2, n, w->models, w->nmodels,
                       lmeds ? 1 : 0, thr2, npow2, w->counts, w->medians);
    UVO_HIP_TRY(c, hipGetLastError());
    UVO_HIP_TRY(c, hipMemcpyAsync(w->h_nmodels.data(), w->nmodels, sizeof(int) * first, hipMemcpyDeviceToHost, st));
    if (lmeds) UVO_HIP_TRY(c, hipMemcpyAsync(w->h_medians.data(), w->medians, sizeof(double) * 10 * first, hipMemcpyDeviceToHost, st));
    else UVO_HIP_TRY(c, hipMemcpyAsync(w->h_counts.data(), w->counts, sizeof(int) * 10 * first, hipMemcpyDeviceToHost, st));
    UVO_HIP_TRY(c, host_sync(c, st));
    Winner win;
    if (lmeds) win = replay_lmeds(w->h_nmodels.data(), w->h_medians.data(), 10, nsub);
    else {
        int settled = niters;
        win = replay_ransac(w->h_nmodels.data(), w->h_counts.data(), 10, first, niters, n, modelPoints, prob, &settled);
        if (first < nsub && settled > first) {                 // the scan goes on past the first round: solve and score the rest, rescan
            const int rest = nsub - first;
            hipLaunchKernelGGL(k_fivepoint_hyp, dim3((rest + kFpPerWg - 1) / kFpPerWg), dim3(64), 0, st, w->q1, w->q2, w->subsets + (size_t)5 * first, rest,
                               w->models + (size_t)90 * first, w->nmodels + first);
            hipLaunchKernelGGL(k_e_score, dim3(10, rest), dim3(256), 0, st, w->q1, w->q2, n, w->models + (size_t)90 * first, w->nmodels + first,
                               0, thr2, npow2, w->counts + (size_t)10 * first, w->medians + (size_t)10 * first);
            UVO_HIP_TRY(c, hipGetLastError());
            UVO_HIP_TRY(c, hipMemcpyAsync(w->h_nmodels.data() + first, w->nmodels + first, sizeof(int) * rest, hipMemcpyDeviceToHost, st));
            UVO_HIP_TRY(c, hipMemcpyAsync(w->h_counts.data() + (size_t)10 * first, w->counts + (size_t)10 * first, sizeof(int) * 10 * rest, hipMemcpyDeviceToHost, st));
            UVO_HIP_TRY(c, host_sync(c, st));
            win = replay_ransac(w->h_nmodels.data(), w->h_counts.data(), 10, nsub, niters, n, modelPoints, prob);
        }
    }
    if (win.hyp < 0) return UVO_OK;
    double final_thr = threshold;
    if (lmeds) {
        double sigma = 2.5 * 1.4826 * (1 + 5. / (n - modelPoints)) * sqrt(win.min_median);
        final_thr = sigma > 0.001 ? sigma : 0.001;
    }
    const double* best = w->models + (size_t)win.hyp * 90 + win.model * 9;
    hipLaunchKernelGGL(k_model_mask, dim3((n + 255) / 256), dim3(256), 0, st, 0, w->q1, w->q2, nullptr, nullptr, n, best,
                       (float)(final_thr * final_thr), w->masks);
    UVO_HIP_TRY(c, hipGetLastError());
    UVO_HIP_TRY(c, hipMemcpyAsync(w->h_mask.data(), w->masks, n, hipMemcpyDeviceToHost, st));
    UVO_HIP_TRY(c, hipMemcpyAsync(w->h_models.data(), best, sizeof(double) * 9, hipMemcpyDeviceToHost, st));
    UVO_HIP_TRY(c, host_sync(c, st));
    memcpy(mask, w->h_mask.data(), (size_t)n); memcpy(E, w->h_models.data(), sizeof(double) * 9);
    if (lmeds) { int good = 0; for (int i = 0; i < n; i++) good += mask[i]; *ok = good >= modelPoints; }
    else *ok = 1;
    return UVO_OK;
}

// decomposeEssentialMat (host; 3x3 Jacobi SVD)
static void decompose_essential(const double* E, double* R1, double* R2, double* t)
{
    double Ein[9], at[9], w[3], vt[9], wt[3], U[9], Vt[9];
    memcpy(Ein, E, sizeof(Ein));
    svd_square<3>(SArr<1>{Ein}, SArr<1>{at}, SArr<1>{w}, SArr<1>{vt}, SArr<1>{wt});
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { U[i*3 + j] = at[j*3 + i]; Vt[i*3 + j] = vt[i*3 + j]; }
    if (det3(U) < 0) for (int i = 0; i < 9; i++) U[i] *= -1.;
    if (det3(Vt) < 0) for (int i = 0; i < 9; i++) Vt[i] *= -1.;
    const double W[9] = { 0, 1, 0, -1, 0, 0, 0, 0, 1 }, Wt[9] = { 0, -1, 0, 1, 0, 0, 0, 0, 1 };
    double UW[9];
    mat3_mul(U, W, UW); mat3_mul(UW, Vt, R1);
    mat3_mul(U, Wt, UW); mat3_mul(UW, Vt, R2);
    t[0] = U[2]; t[1] = U[5]; t[2] = U[8];
}

// cv::recoverPose(E, p1, p2, K, R, t, mask), distanceThresh = 50 (VOU:149)
uvo_status mono_recover_pose(Ctx* c, const double* E, const uvo_point2f* p1, const uvo_point2f* p2, int n, const double* K,
                             double* R, double* t, uint8_t* mask, int* good_out)
{
    *good_out = 0;
    if (n > c->cap) { c->err = "point count exceeds the context's max_kpts"; return UVO_CAPACITY; }
    MonoWs* w = mono_ws(c);
    if (!w) { c->err = "mono workspace allocation failed"; return UVO_HIP_ERROR; }
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const double ax = 1. / fx, bx = -cx * ax, ay = 1. / fy, by = -cy * ay;
    Pinned<double>& q1 = w->h_q1; Pinned<double>& q2 = w->h_q2;
    for (int i = 0; i < n; i++) {
        q1[2*i] = p1[i].x * ax + bx; q1[2*i+1] = p1[i].y * ay + by;
        q2[2*i] = p2[i].x * ax + bx; q2[2*i+1] = p2[i].y * ay + by;
    }
    double R1[9], R2[9], tt[3];
    decompose_essential(E, R1, R2, tt);
    const double* Rc[4] = { R1, R2, R1, R2 };
    const double sg[4] = { 1, 1, -1, -1 };
    PoseCands pc;
    for (int cnd = 0; cnd < 4; cnd++)
        for (int i = 0; i < 3; i++) { pc.P[cnd][i*4] = Rc[cnd][i*3]; pc.P[cnd][i*4+1] = Rc[cnd][i*3+1]; pc.P[cnd][i*4+2] = Rc[cnd][i*3+2]; pc.P[cnd][i*4+3] = sg[cnd] * tt[i]; }
    int good[4] = {0, 0, 0, 0};
    if (n > 0) {
        hipStream_t st = c->stream;
        UVO_HIP_TRY(c, hipMemcpyAsync(w->q1, q1.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice, st));
        UVO_HIP_TRY(c, hipMemcpyAsync(w->q2, q2.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice, st));
        memcpy(w->h_mask.data(), mask, (size_t)n);
        UVO_HIP_TRY(c, hipMemcpyAsync(w->masks + 4 * (size_t)c->cap, w->h_mask.data(), n, hipMemcpyHostToDevice, st));
        UVO_HIP_TRY(c, hipMemsetAsync(w->good, 0, sizeof(int) * 4, st));
        hipLaunchKernelGGL(k_recover_pose, dim3((n + kRpThreads - 1) / kRpThreads, 4), dim3(kRpThreads), 0, st, w->q1, w->q2, n, pc,
                           w->masks + 4 * (size_t)c->cap, w->masks, w->good);
        UVO_HIP_TRY(c, hipGetLastError());
        UVO_HIP_TRY(c, hipMemcpyAsync(w->h_good.data(), w->good, sizeof(good), hipMemcpyDeviceToHost, st));
        UVO_HIP_TRY(c, host_sync(c, st));
        memcpy(good, w->h_good.data(), sizeof(good));
    }
    int best;
    if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3]) best = 0;
    else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3]) best = 1;
    else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3]) best = 2;
    else best = 3;
    memcpy(R, Rc[best], sizeof(double) * 9);
    for (int i = 0; i < 3; i++) t[i] = sg[best] * tt[i];
    if (n > 0) {
        UVO_HIP_TRY(c, hipMemcpyAsync(w->h_mask.data(), w->masks + (size_t)best * n, n, hipMemcpyDeviceToHost, c->stream));
        UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
        memcpy(mask, w->h_mask.data(), (size_t)n);
    }
    *good_out = good[best];
    return UVO_OK;
}

// ---- homography: refit + LM polish (host, fundam.cpp / levmarq.cpp order) ----
static void refine_compute(const float* M, const float* m, int count, const double* h, double* err, double* J)
{
    for (int i = 0; i < count; i++) {
        double Mx = M[2*i], My = M[2*i+1];
        double ww = h[6]*Mx + h[7]*My + 1.;
        ww = fabs(ww) > DBL_EPSILON ? 1./ww : 0;
        double xi = (h[0]*Mx + h[1]*My + h[2])*ww;
        double yi = (h[3]*Mx + h[4]*My + h[5])*ww;
        err[i*2] = xi - m[2*i];
        err[i*2+1] = yi - m[2*i+1];
        if (J) {
            double* Jp = J + (size_t)i*16;
            Jp[0] = Mx*ww; Jp[1] = My*ww; Jp[2] = ww;
            Jp[3] = Jp[4] = Jp[5] = 0.;
            Jp[6] = -Mx*ww*xi; Jp[7] = -My*ww*xi;
            Jp[8] = Jp[9] = Jp[10] = 0.;
            Jp[11] = Mx*ww; Jp[12] = My*ww; Jp[13] = ww;
            Jp[14] = -Mx*ww*yi; Jp[15] = -My*ww*yi;
        }
    }
}
static double norm_l2sqr(const double* a, int n)
{
    double s = 0; int i = 0;
    for (; i <= n - 4; i += 4) { double v0 = a[i], v1 = a[i+1], v2 = a[i+2], v3 = a[i+3]; s += v0*v0 + v1*v1 + v2*v2 + v3*v3; }
    for (; i < n; i++) { double v = a[i]; s += v*v; }
    return s;
}
static double dot_n(const double* a, const double* b, int n)
{
    double r = 0; int i = 0;
    for (; i <= n - 4; i += 4) r += a[i]*b[i] + a[i+1]*b[i+1] + a[i+2]*b[i+2] + a[i+3]*b[i+3];
    for (; i < n; i++) r += a[i]*b[i];
    return r;
}
static double norm_inf(const double* a, int n) { double s = 0; for (int i = 0; i < n; i++) { double v = fabs(a[i]); if (s < v) s = v; } return s; }
static void jtj_jtr(const double* J, const double* r, int rows, double* A, double* v)
{
    for (int i = 0; i < 8; i++) {
        for (int j = i; j < 8; j++) { double s = 0; for (int k = 0; k < rows; k++) s += J[k*8 + i]*J[k*8 + j]; A[i*8 + j] = s; }
        double s = 0; for (int k = 0; k < rows; k++) s += J[k*8 + i]*r[k];
        v[i] = s * 1.0;
    }
    for (int i = 0; i < 8; i++) for (int j = 0; j < i; j++) A[i*8 + j] = A[j*8 + i];
}
static void eig8(const double* A, double* w, double* v)
{
    double a[64]; int ind[16];
    memcpy(a, A, sizeof(a));
    jacobi_eigen(SArr<1>{a}, 8, SArr<1>{w}, SArr<1>{v}, ind, ind + 8);
}
static void solve_eig8(const double* A, const double* b, double* x)
{
    double w[8], v[64];
    eig8(A, w, v);
    double threshold = 0;
    for (int i = 0; i < 8; i++) { x[i] = 0; threshold += w[i]; }
    threshold *= DBL_EPSILON * 2;
    for (int i = 0; i < 8; i++) {
        double wi = w[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1/wi;
        double s = 0;
        for (int j = 0; j < 8; j++) s += v[i*8 + j]*b[j];
        s *= wi;
        for (int j = 0; j < 8; j++) x[j] = x[j] + s*v[i*8 + j];
    }
}
static void invert_eig8(const double* A, double* Ainv)
{
    double w[8], v[64];
    eig8(A, w, v);
    double threshold = 0;
    for (int i = 0; i < 64; i++) Ainv[i] = 0;
    for (int i = 0; i < 8; i++) threshold += w[i];
    threshold *= DBL_EPSILON * 2;
    for (int i = 0; i < 8; i++) {
        double wi = w[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1/wi;
        double buffer[8];
        for (int j = 0; j < 8; j++) buffer[j] = v[i*8 + j]*wi;
        for (int k = 0; k < 8; k++) { double sv = v[i*8 + k]; for (int j = 0; j < 8; j++) Ainv[k*8 + j] = Ainv[k*8 + j] + sv*buffer[j]; }
    }
}
static void lm_refine_homography(const float* M, const float* m, int count, double* h)
{
    const int lx = 8, maxIters = 10; const double epsx = FLT_EPSILON, epsf = FLT_EPSILON;
    int rows = 2*count;
    std::vector<double> r(rows), rd(rows), J((size_t)rows*8);
    double x[8], xd[8], d[8], v[8], A[64], Ap[64], D[8], temp_d[8];
    memcpy(x, h, sizeof(x));
    refine_compute(M, m, count, x, r.data(), J.data());
    double S = norm_l2sqr(r.data(), rows);
    jtj_jtr(J.data(), r.data(), rows, A, v);
    for (int i = 0; i < lx; i++) D[i] = A[i*8 + i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    int iter = 0;
    for (;;) {
        memcpy(Ap, A, sizeof(Ap));
        for (int i = 0; i < lx; i++) Ap[i*8 + i] += lambda*D[i];
        solve_eig8(Ap, v, d);
        for (int i = 0; i < lx; i++) xd[i] = x[i] - d[i];
        refine_compute(M, m, count, xd, rd.data(), nullptr);
        double Sd = norm_l2sqr(rd.data(), rows);
        for (int i = 0; i < lx; i++) { double s0 = 0; for (int k = 0; k < lx; k++) s0 += A[i*8 + k]*d[k]; temp_d[i] = s0*-1 + v[i]*2; }
        double dS = dot_n(d, temp_d, lx);
        double R = (S - Sd)/(fabs(dS) > DBL_EPSILON ? dS : 1);
        if (R > Rhi) { lambda *= 0.5; if (lambda < lc) lambda = 0; }
        else if (R < Rlo) {
            double t = dot_n(d, v, lx);
            double nu = (Sd - S)/(fabs(t) > DBL_EPSILON ? t : 1) + 2;
            nu = nu > 2. ? nu : 2.; nu = nu < 10. ? nu : 10.;
            if (lambda == 0) {
                invert_eig8(A, Ap);
                double maxval = DBL_EPSILON;
                for (int i = 0; i < lx; i++) { double a = fabs(Ap[i*8 + i]); if (maxval < a) maxval = a; }
                lambda = lc = 1./maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            memcpy(x, xd, sizeof(x));
            refine_compute(M, m, count, x, r.data(), J.data());
            jtj_jtr(J.data(), r.data(), rows, A, v);
        }
        iter++;
        bool proceed = iter < maxIters && norm_inf(d, lx) >= epsx && norm_inf(r.data(), rows) >= epsf;
        if (!proceed) break;
    }
    memcpy(h, x, sizeof(x));
}

// cv::findHomography (VOU:152)
uvo_status mono_find_homography(Ctx* c, const uvo_point2f* p1, const uvo_point2f* p2, int n, int method, double thr, int maxIters,
                                double confidence, double* H, uint8_t* mask, int* ok)
{
    *ok = 0;
    memset(mask, 0, n);
    if (n > c->cap) { c->err = "point count exceeds the context's max_kpts"; return UVO_CAPACITY; }
    MonoWs* w = mono_ws(c);
    if (!w) { c->err = "mono workspace allocation failed"; return UVO_HIP_ERROR; }
    if (thr <= 0) thr = 3;
    const int modelPoints = 4;
    if (n < modelPoints) return UVO_OK;
    Pinned<float>& src = w->h_src; Pinned<float>& dst = w->h_dst;
    for (int i = 0; i < n; i++) { src[2*i] = p1[i].x; src[2*i+1] = p1[i].y; dst[2*i] = p2[i].x; dst[2*i+1] = p2[i].y; }
    double scratch[171]; int iscratch[18];
    int result = 0;
    if (n == 4) {
        memset(mask, 1, n);
        result = homography_kernel(src.data(), dst.data(), n, H, SArr<1>{scratch}, SArr<1>{scratch + 81}, SArr<1>{scratch + 90}, iscratch) > 0;
    } else {
        hipStream_t st = c->stream;
        const bool lmeds = method != 8;
        int niters;
        if (lmeds) { niters = ransac_update_num_iters(confidence, 0.45, modelPoints, maxIters); niters = niters > 3 ? niters : 3; }
        else niters = maxIters > 1 ? maxIters : 1;
        if (niters > kMaxHyp) { c->err = "max_iters exceeds the compiled hypothesis capacity (2048)"; return UVO_CAPACITY; }
        bool failed_first = false;
        int nsub = make_subsets(w->h_subsets.data(), niters, modelPoints, n, lmeds ? 1000 : 10000, src.data(), dst.data(), &failed_first);
        if (nsub > 0) {
            UVO_HIP_TRY(c, hipMemcpyAsync(w->src, src.data(), sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
            UVO_HIP_TRY(c, hipMemcpyAsync(w->dst, dst.data(), sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
            UVO_HIP_TRY(c, hipMemcpyAsync(w->subsets, w->h_subsets.data(), sizeof(int) * 4 * nsub, hipMemcpyHostToDevice, st));
            // RANSAC in rounds, as mono_find_essential: the first kFirstRound subsets, the rest only if the adaptive count reaches past them
            // (at the shipped 0.1-px threshold the scan of a low-inlier frame never settles inside the first round, and the second launch
            //  costs the frame another 0.35 ms and a synchronisation: the workspace remembers what the last scan needed)
            const int kFirstRound = 128;
            const int first = (!lmeds && nsub > kFirstRound && !w->h_one_round) ? kFirstRound : nsub;
            hipLaunchKernelGGL(k_h_hyp, dim3((first + kHPerWg - 1) / kHPerWg), dim3(64), 0, st, w->src, w->dst, w->subsets, first,
                               w->models, w->nmodels);
            const int npow2 = next_pow2(n);
            const float thr2 = (float)(thr * thr);
            hipLaunchKernelGGL(k_h_score, dim3(first), dim3(256), lmeds ? sizeof(float) * npow2 : 0, st, w->src, w->dst, n, w->models, w->nmodels,
                               lmeds ? 1 : 0, thr2, npow2, w->counts, w->medians);
            UVO_HIP_TRY(c, hipGetLastError());
            UVO_HIP_TRY(c, hipMemcpyAsync(w->h_nmodels.data(), w->nmodels, sizeof(int) * first, hipMemcpyDeviceToHost, st));
            if (lmeds) UVO_HIP_TRY(c, hipMemcpyAsync(w->h_medians.data(), w->medians, sizeof(double) * first, hipMemcpyDeviceToHost, st));
            else UVO_HIP_TRY(c, hipMemcpyAsync(w->h_counts.data(), w->counts, sizeof(int) * first, hipMemcpyDeviceToHost, st));
            UVO_HIP_TRY(c, host_sync(c, st));
            Winner win;
            if (lmeds) win = replay_lmeds(w->h_nmodels.data(), w->h_medians.data(), 1, nsub);
            else {
                int settled = niters;
                win = replay_ransac(w->h_nmodels.data(), w->h_counts.data(), 1, first, niters, n, modelPoints, confidence, &settled);
                if (nsub > kFirstRound) w->h_one_round = settled > kFirstRound;          // the scan's result is the same either way
                if (first < nsub && settled > first) {
                    const int rest = nsub - first;
                    hipLaunchKernelGGL(k_h_hyp, dim3((rest + kHPerWg - 1) / kHPerWg), dim3(64), 0, st, w->src, w->dst,
                                       w->subsets + (size_t)4 * first, rest, w->models + (size_t)9 * first, w->nmodels + first);
                    hipLaunchKernelGGL(k_h_score, dim3(rest), dim3(256), 0, st, w->src, w->dst, n, w->models + (size_t)9 * first, w->nmodels + first,
                                       0, thr2, npow2, w->counts + first, w->medians + first);
                    UVO_HIP_TRY(c, hipGetLastError());
                    UVO_HIP_TRY(c, hipMemcpyAsync(w->h_nmodels.data() + first, w->nmodels + first, sizeof(int) * rest, hipMemcpyDeviceToHost, st));
                    UVO_HIP_TRY(c, hipMemcpyAsync(w->h_counts.data() + first, w->counts + first, sizeof(int) * rest, hipMemcpyDeviceToHost, st));
                    UVO_HIP_TRY(c, host_sync(c, st));
                    win = replay_ransac(w->h_nmodels.data(), w->h_counts.data(), 1, nsub, niters, n, modelPoints, confidence);
                }
            }
            if (win.hyp >= 0) {
                double final_thr = thr;
                if (lmeds) {
                    double sigma = 2.5 * 1.4826 * (1 + 5. / (n - modelPoints)) * sqrt(win.min_median);
                    final_thr = sigma > 0.001 ? sigma : 0.001;
                }
                const double* best = w->models + (size_t)win.hyp * 9;
                hipLaunchKernelGGL(k_model_mask, dim3((n + 255) / 256), dim3(256), 0, st, 1, nullptr, nullptr, w->src, w->dst, n, best,
                                   (float)(final_thr * final_thr), w->masks);
                UVO_HIP_TRY(c, hipGetLastError());
                UVO_HIP_TRY(c, hipMemcpyAsync(w->h_mask.data(), w->masks, n, hipMemcpyDeviceToHost, st));
                UVO_HIP_TRY(c, hipMemcpyAsync(w->h_models.data(), best, sizeof(double) * 9, hipMemcpyDeviceToHost, st));
                UVO_HIP_TRY(c, host_sync(c, st));
                memcpy(mask, w->h_mask.data(), (size_t)n); memcpy(H, w->h_models.data(), sizeof(double) * 9);
                if (lmeds) { int good = 0; for (int i = 0; i < n; i++) good += mask[i]; result = good >= modelPoints; }
                else result = 1;
            }
        }
    }
    if (result && n > 4) {
        int k = 0;                                                       // compressElems
        for (int i = 0; i < n; i++) if (mask[i]) { src[2*k] = src[2*i]; src[2*k+1] = src[2*i+1]; dst[2*k] = dst[2*i]; dst[2*k+1] = dst[2*i+1]; k++; }
        if (k > 0) {
            homography_kernel(src.data(), dst.data(), k, H, SArr<1>{scratch}, SArr<1>{scratch + 81}, SArr<1>{scratch + 90}, iscratch);
            lm_refine_homography(src.data(), dst.data(), k, H);
        }
    }
    if (!result) memset(mask, 0, n);
    *ok = result;
    return UVO_OK;
}

// ---- cv::decomposeHomographyMat (HomographyDecompInria), host ----
static void m3inv(const double* a, double* b)
{
    double d = det3(a);
    if (d == 0) { memset(b, 0, sizeof(double)*9); return; }
    d = 1./d;
    b[0] = (a[4]*a[8] - a[5]*a[7])*d; b[1] = (a[2]*a[7] - a[1]*a[8])*d; b[2] = (a[1]*a[5] - a[2]*a[4])*d;
    b[3] = (a[5]*a[6] - a[3]*a[8])*d; b[4] = (a[0]*a[8] - a[2]*a[6])*d; b[5] = (a[2]*a[3] - a[0]*a[5])*d;
    b[6] = (a[3]*a[7] - a[4]*a[6])*d; b[7] = (a[1]*a[6] - a[0]*a[7])*d; b[8] = (a[0]*a[4] - a[1]*a[3])*d;
}
static double opposite_of_minor(const double* M, int row, int col)
{
    int x1 = col == 0 ? 1 : 0, x2 = col == 2 ? 1 : 2, y1 = row == 0 ? 1 : 0, y2 = row == 2 ? 1 : 2;
    return M[y1*3 + x2]*M[y2*3 + x1] - M[y1*3 + x1]*M[y2*3 + x2];
}
static int signd(double x) { return x >= 0 ? 1 : -1; }
static void find_rmat(const double* Hn, const double* tstar, const double* n, double v, double* R)
{
    double T[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T[i*3 + j] = (i == j ? 1.0 : 0.0) - (2/v)*tstar[i]*n[j];
    mat3_mul(Hn, T, R);
    if (det3(R) < 0) for (int i = 0; i < 9; i++) R[i] *= -1;
}
int decompose_homography_mat(const double* H, const double* K, double* Rs, double* ts, double* ns)
{
    double Kinv[9], Hn[9], tmp[9];
    m3inv(K, Kinv);
    mat3_mul(Kinv, H, tmp); mat3_mul(tmp, K, Hn);
    { double Hc[9], at[9], w[3], vt[9], wt[3]; memcpy(Hc, Hn, sizeof(Hc));
      svd_square<3>(SArr<1>{Hc}, SArr<1>{at}, SArr<1>{w}, SArr<1>{vt}, SArr<1>{wt});
      double s = 1.0/w[1]; for (int i = 0; i < 9; i++) Hn[i] = Hn[i]*s; }
    const double epsilon = 0.001;
    double S[9], Ht[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Ht[i*3 + j] = Hn[j*3 + i];
    mat3_mul(Ht, Hn, S);
    S[0] -= 1.0; S[4] -= 1.0; S[8] -= 1.0;
    double ninf = 0;
    for (int i = 0; i < 9; i++) { double a = fabs(S[i]); if (ninf < a) ninf = a; }
    if (ninf < epsilon) {
        memcpy(Rs, Hn, sizeof(double)*9);
        for (int i = 0; i < 3; i++) { ts[i] = 0; ns[i] = 0; }
        return 1;
    }
    double npa[3], npb[3];
    double M00 = opposite_of_minor(S, 0, 0), M11 = opposite_of_minor(S, 1, 1), M22 = opposite_of_minor(S, 2, 2);
    double rtM00 = sqrt(M00), rtM11 = sqrt(M11), rtM22 = sqrt(M22);
    double M01 = opposite_of_minor(S, 0, 1), M12 = opposite_of_minor(S, 1, 2), M02 = opposite_of_minor(S, 0, 2);
    int e12 = signd(M12), e02 = signd(M02), e01 = signd(M01);
    double nS00 = fabs(S[0]), nS11 = fabs(S[4]), nS22 = fabs(S[8]);
    int indx = 0;
    if (nS00 < nS11) { indx = 1; if (nS11 < nS22) indx = 2; }
    else { if (nS00 < nS22) indx = 2; }
    switch (indx) {
    case 0:
        npa[0] = S[0];               npb[0] = S[0];
        npa[1] = S[1] + rtM22;       npb[1] = S[1] - rtM22;
        npa[2] = S[2] + e12*rtM11;   npb[2] = S[2] - e12*rtM11;
        break;
    case 1:
        npa[0] = S[1] + rtM22;       npb[0] = S[1] - rtM22;
        npa[1] = S[4];               npb[1] = S[4];
        npa[2] = S[5] - e02*rtM00;   npb[2] = S[5] + e02*rtM00;
        break;
    default:
        npa[0] = S[2] + e01*rtM11;   npb[0] = S[2] - e01*rtM11;
        npa[1] = S[5] + rtM00;       npb[1] = S[5] - rtM00;
        npa[2] = S[8];               npb[2] = S[8];
        break;
    }
    double traceS = S[0] + S[4] + S[8];
    double v = 2.0 * sqrtf((float)(1 + traceS - M00 - M11 - M22));
    double ESii = signd(S[indx*3 + indx]);
    double r_2 = 2 + traceS + v, nt_2 = 2 + traceS - v;
    double r = sqrt(r_2), n_t = sqrt(nt_2);
    double na[3], nb[3];
    { double nn = sqrt(npa[0]*npa[0] + npa[1]*npa[1] + npa[2]*npa[2]); for (int i = 0; i < 3; i++) na[i] = npa[i] / nn; }
    { double nn = sqrt(npb[0]*npb[0] + npb[1]*npb[1] + npb[2]*npb[2]); for (int i = 0; i < 3; i++) nb[i] = npb[i] / nn; }
    double half_nt = 0.5*n_t, esii_t_r = ESii*r;
    double ta_star[3], tb_star[3];
    for (int i = 0; i < 3; i++) { ta_star[i] = half_nt*(esii_t_r*nb[i] - n_t*na[i]); tb_star[i] = half_nt*(esii_t_r*na[i] - n_t*nb[i]); }
    double Ra[9], Rb[9], ta[3], tb[3];
    find_rmat(Hn, ta_star, na, v, Ra);
    find_rmat(Hn, tb_star, nb, v, Rb);
    for (int i = 0; i < 3; i++) {
        ta[i] = Ra[i*3]*ta_star[0] + Ra[i*3+1]*ta_star[1] + Ra[i*3+2]*ta_star[2];
        tb[i] = Rb[i*3]*tb_star[0] + Rb[i*3+1]*tb_star[1] + Rb[i*3+2]*tb_star[2];
    }
    memcpy(Rs, Ra, 72); memcpy(Rs + 9, Ra, 72); memcpy(Rs + 18, Rb, 72); memcpy(Rs + 27, Rb, 72);
    for (int i = 0; i < 3; i++) {
        ts[i] = ta[i]; ns[i] = na[i];
        ts[3 + i] = -ta[i]; ns[3 + i] = -na[i];
        ts[6 + i] = tb[i]; ns[6 + i] = nb[i];
        ts[9 + i] = -tb[i]; ns[9 + i] = -nb[i];
    }
    return 4;
}

void projection_matrix(const double* R, const double* t, const double* K, double* P)      // VOU:9-15
{
    double Rt[12];
    for (int i = 0; i < 3; i++) { Rt[i*4] = R[i*3]; Rt[i*4+1] = R[i*3+1]; Rt[i*4+2] = R[i*3+2]; Rt[i*4+3] = t[i]; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) P[i*4 + j] = K[i*3]*Rt[j] + K[i*3+1]*Rt[4 + j] + K[i*3+2]*Rt[8 + j];
}

// recover_pose_homography (VOU:581-624), including the reference's reinterpretation of two neighbouring
// floats as one double at VOU:601-602 (the last column, read past the buffer there, is not counted).
uvo_status mono_recover_pose_homography(Ctx* c, const double* H, const uvo_point2f* p1, const uvo_point2f* p2, int n, const double* K,
                                        double HOMOGRAPHY_DISTANCE, double* R, double* t, int* max_good_out)
{
    *max_good_out = 0;
    if (n > c->cap) { c->err = "point count exceeds the context's max_kpts"; return UVO_CAPACITY; }
    double Rs[36], ts[12], ns[12];
    int solutions = decompose_homography_mat(H, K, Rs, ts, ns);
    const double I[9] = {1,0,0,0,1,0,0,0,1}, z[3] = {0,0,0};
    double proj_std[12];
    projection_matrix(I, z, K, proj_std);
    int best = -1, max_good = 0;
    std::vector<float4> p4(n > 0 ? n : 1);
    std::vector<float> zrow(n + 1);
    hipStream_t st = c->stream;
    if (n > 0) {
        UVO_HIP_TRY(c, hipMemcpyAsync(c->d_x1, p1, sizeof(uvo_point2f) * n, hipMemcpyHostToDevice, st));
        UVO_HIP_TRY(c, hipMemcpyAsync(c->d_x2, p2, sizeof(uvo_point2f) * n, hipMemcpyHostToDevice, st));
    }
    for (int i = 0; i < solutions; i++) {
        double P[12];
        projection_matrix(Rs + 9*i, ts + 3*i, K, P);
        if (n > 0) {
            UVO_TRY(pose_triangulate(c, proj_std, P, nullptr, n));
            UVO_HIP_TRY(c, hipMemcpyAsync(p4.data(), c->d_pts4, sizeof(float4) * n, hipMemcpyDeviceToHost, st));
            UVO_HIP_TRY(c, host_sync(c, st));
        }
        for (int j = 0; j < n; j++) {                       // convert_from_homogeneous_coords (VOU:71-83): col / w in float
            float inv = (float)(1.0 / (double)p4[j].w);
            zrow[j] = p4[j].z * inv + 0.f;
        }
        int good = 0;
        for (int j = 0; j + 1 < n; j++) {
            uint32_t lo, hi; memcpy(&lo, &zrow[j], 4); memcpy(&hi, &zrow[j + 1], 4);
            uint64_t bits = ((uint64_t)hi << 32) | lo;
            double v; memcpy(&v, &bits, 8);
            if (v > 0 && v < HOMOGRAPHY_DISTANCE) good++;
        }
        if (good > max_good) { best = i; max_good = good; }
    }
    if (best != -1) {
        const double* tb = ts + 3*best;
        double nrm = sqrt(tb[0]*tb[0] + tb[1]*tb[1] + tb[2]*tb[2]);
        double inv = 1.0 / nrm;
        memcpy(R, Rs + 9*best, sizeof(double)*9);
        for (int k = 0; k < 3; k++) t[k] = tb[k] * inv;
    }
    *max_good_out = max_good;
    return UVO_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// The mono loop's pose stage with its points RESIDENT ON THE DEVICE (round 5).  Until round 4 the lane's worker copied the M matches
// and both point lists to the host, called the host-pointer operator (which uploaded them again, twice: findEssentialMat and
// recoverPose each normalised on the host), downloaded the mask, compacted the inliers on the host and uploaded them three more times
// for the triangulation -- eight host syncs per frame.  Here:
//   k_mono_prep     (tail of the frame's stage A) normalised points q1, q2 (findEssentialMat: (p - c) / f in double) from the matched
//                   points the gather kernel left in d_x1 / d_x2; select_estimation_method's decision "median displacement < DISTANCE"
//                   from one count and the two values next to the threshold (VO_utility.cpp:725-748, math_utility.cpp:65-86: exactly
//                   the comparison the sorted median would give); the points mirrored into pinned memory for the host-side pieces
//                   that want them (checkSubset of the homography sampler, its refit)
//   essential first (the frame's method is E): subsets by the host's cv::RNG replay (no points needed) read from pinned memory,
//                   k_fivepoint_hyp + k_e_score with counts / medians written straight into pinned memory -- ONE host sync -- the
//                   host's scan, then without another sync: k_model_mask, recoverPose's four candidates (decomposeEssentialMat on
//                   the host from the mirrored model), k_mono_pick (the winning candidate by recoverPose's >= chain, extract_inliers
//                   as an ordered compaction, the valid-inlier count), triangulatePoints + extract_3Dpoints on the compacted pairs
//                   with the candidate chosen ON THE DEVICE, convert_3Dpoints_camera's z list -- the SECOND and last host sync.
// Two syncs instead of eight; nothing is uploaded.  Frames whose first method is the homography, and second attempts after a failed
// acceptance test (VO_utility.cpp:165-178), take the host-pointer operators as before, on the pinned mirrors.  Same kernels, same
// operands, same order: the results are those of the host-pointer path bit for bit (tests/test_gpu_parity.py, test_gpu_configs.py).
// ------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_mono_prep(const uvo_point2f* __restrict__ x1, const uvo_point2f* __restrict__ x2, const int* __restrict__ count, int cap,
                                                    double ax, double bx, double ay, double by, int distance,
                                                    double* __restrict__ q1, double* __restrict__ q2, uvo_point2f* __restrict__ h_x1, uvo_point2f* __restrict__ h_x2,
                                                    int* __restrict__ h_prep)
{
    const int tid = threadIdx.x, M = min(*count, cap);
    const double D = (double)distance;
    int lt = 0; double max_lt = -1.0, min_ge = DBL_MAX;
    for (int i = tid; i < M; i += 1024) {
        const uvo_point2f a = x1[i], b = x2[i];
        q1[2*i] = a.x * ax + bx; q1[2*i + 1] = a.y * ay + by;
        q2[2*i] = b.x * ax + bx; q2[2*i + 1] = b.y * ay + by;
        h_x1[i] = a; h_x2[i] = b;
        const double dx = a.x - b.x, dy = a.y - b.y;                 // float differences, as VO_utility.cpp:733-734
        const double d = sqrt(dx * dx + dy * dy);
        if (d < D) { lt++; max_lt = d > max_lt ? d : max_lt; } else min_ge = d < min_ge ? d : min_ge;
    }
    __shared__ int s_lt[16]; __shared__ double s_max[16], s_min[16];
    for (int o = 32; o > 0; o >>= 1) {
        lt += __shfl_down(lt, o);
        const double m1 = __shfl_down(max_lt, o), m2 = __shfl_down(min_ge, o);
        max_lt = m1 > max_lt ? m1 : max_lt; min_ge = m2 < min_ge ? m2 : min_ge;
    }
    if ((tid & 63) == 0) { s_lt[tid >> 6] = lt; s_max[tid >> 6] = max_lt; s_min[tid >> 6] = min_ge; }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < 16; k++) { lt += s_lt[k]; max_lt = s_max[k] > max_lt ? s_max[k] : max_lt; min_ge = s_min[k] < min_ge ? s_min[k] : min_ge; }
        // compute_median (math_utility.cpp:65-86) of the M distances against DISTANCE, without sorting them: the sorted vector has its
        // first `lt` entries below D.  Odd M: v[M/2] < D iff lt > M/2.  Even M: (v[mid-1] + v[mid]) / 2.0 with mid = M/2 -- both below
        // D (lt > mid), both at or above it (lt < mid), or exactly the two values either side of D (lt == mid).
        bool below;
        const int mid = M / 2;
        if (M == 0) below = 0.0 < D;                                  // compute_median of an empty vector returns 0.0
        else if (M & 1) below = lt > mid;
        else below = lt > mid ? true : (lt < mid ? false : (max_lt + min_ge) / 2.0 < D);
        h_prep[1] = M;
        __threadfence_system();
        h_prep[0] = below ? 0 : 1;                                    // select_estimation_method: true = essential
    }
}
uvo_status mono_prep_launch(Ctx* c, hipStream_t st, const double* K, int distance)
{
    MonoWs* w = mono_ws(c);
    if (!w) { c->err = "mono workspace allocation failed"; return UVO_HIP_ERROR; }
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const double ax = 1. / fx, bx = -cx * ax, ay = 1. / fy, by = -cy * ay;
    w->h_prep[0] = -1;
    hipLaunchKernelGGL(k_mono_prep, dim3(1), dim3(1024), 0, st, c->d_x1, c->d_x2, c->d_counts + CN_M, c->cap, ax, bx, ay, by, distance,
                       w->q1, w->q2, w->h_x1.data(), w->h_x2.data(), w->h_prep.data());
    UVO_HIP_TRY(c, hipGetLastError());
    return UVO_OK;
}
int mono_prep_method(Ctx* c, int M, const uvo_point2f** k1, const uvo_point2f** k2)
{
    MonoWs* w = static_cast<MonoWs*>(c->mono_ws);
    *k1 = w->h_x1.data(); *k2 = w->h_x2.data();
    return (w->h_prep[1] == M) ? w->h_prep[0] : -1;
}

// device -> pinned mirror of a few KB (a copy command costs the stage ~10 us; this is a kernel among kernels)
__global__ __launch_bounds__(256) void k_mirror_words(const int* __restrict__ src, int* __restrict__ dst, int n)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dst[i] = src[i];
}

struct PoseRt { double R[4][9], t[4][3]; };
// recoverPose's choice among its four candidates (the >= chain of five-point.cpp: the first of the best), extract_inliers
// (VO_utility.cpp:306-329) on findEssentialMat's mask as an ordered compaction, and the count of the winning candidate's mask
__global__ __launch_bounds__(1024) void k_mono_pick(const uvo_point2f* __restrict__ x1, const uvo_point2f* __restrict__ x2, int n, const uint8_t* __restrict__ emask,
                                                    const uint8_t* __restrict__ masks, const int* __restrict__ good, uvo_point2f* __restrict__ in1, uvo_point2f* __restrict__ in2,
                                                    int* __restrict__ pick, int* __restrict__ h_pick, uint8_t* __restrict__ h_mask)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ int wtot[16], s_base, s_valid;
    int best;
    const int g0 = good[0], g1 = good[1], g2 = good[2], g3 = good[3];
    if (g0 >= g1 && g0 >= g2 && g0 >= g3) best = 0;
    else if (g1 >= g0 && g1 >= g2 && g1 >= g3) best = 1;
    else if (g2 >= g0 && g2 >= g1 && g2 >= g3) best = 2;
    else best = 3;
    if (tid == 0) { s_base = 0; s_valid = 0; }
    __syncthreads();
    int valid = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const bool keep = i < n && emask[i] != 0;
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wtot[wv] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int k = 0; k < wv; k++) off += wtot[k];
        if (keep) { in1[off + before] = x1[i]; in2[off + before] = x2[i]; }
        __syncthreads();
        if (tid == 0) { int tot = 0; for (int k = 0; k < 16; k++) tot += wtot[k]; s_base += tot; }
        if (i < n) { const uint8_t m = masks[(size_t)best * n + i]; h_mask[i] = m; valid += m != 0; }
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) valid += __shfl_down(valid, o);
    if (lane == 0) atomicAdd(&s_valid, valid);
    __syncthreads();
    if (tid == 0) {
        pick[0] = best; pick[1] = s_base;
        h_pick[0] = best; h_pick[1] = g0; h_pick[2] = g1; h_pick[3] = g2; h_pick[4] = g3; h_pick[5] = s_base; h_pick[6] = s_valid;
    }
}
// convert_3Dpoints_camera (VO_utility.cpp:46-63) for compute_scale_factor: the z of every good point whose z under the recovered pose
// is positive, into pinned memory (the host takes the median: (float)range / median, VO_utility.cpp:23-38)
__global__ __launch_bounds__(1024) void k_mono_scale(PoseRt rt, const int* __restrict__ pick, const double* __restrict__ good_pts, const int* __restrict__ G_p,
                                                     double* __restrict__ h_zs, int* __restrict__ h_pick)
{
    const int tid = threadIdx.x, best = pick[0], G = *G_p;
    const double* R = rt.R[best]; const double* t = rt.t[best];
    __shared__ int s_n;
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (int i = tid; i < G; i += 1024) {
        const double* q = good_pts + 3 * (size_t)i;
        const double zt = (R[6]*q[0] + R[7]*q[1] + R[8]*q[2]) * 1.0 + t[2] * 1.0;
        if (zt > 0) h_zs[atomicAdd(&s_n, 1)] = q[2];                  // (the order does not reach the median)
    }
    __syncthreads();
    if (tid == 0) { h_pick[7] = G; h_pick[8] = s_n; __threadfence_system(); h_pick[9] = 1; }
}

// findEssentialMat + recoverPose + triangulatePoints + extract_3Dpoints + convert_3Dpoints_camera on the n matched points k_mono_prep
// left on the device; two host syncs.  out->ok = findEssentialMat produced a model (LMedS: with >= 5 inliers).
uvo_status mono_essential_resident(Ctx* c, int n, const double* K, int method, double prob, double threshold, int maxIters, MonoResident* out)
{
    memset(out, 0, sizeof(*out));
    MonoWs* w = mono_ws(c);
    if (!w) { c->err = "mono workspace allocation failed"; return UVO_HIP_ERROR; }
    if (n > c->cap) { c->err = "point count exceeds the context's max_kpts"; return UVO_CAPACITY; }
    const int modelPoints = 5;
    out->mask = w->h_mask.data(); out->zs = w->h_zs.data();
    if (n < modelPoints) { memset(w->h_mask.data(), 0, (size_t)(n > 0 ? n : 0)); return UVO_OK; }
    const double fx = K[0], fy = K[4];
    threshold /= (fx + fy) / 2;
    hipStream_t st = c->stream;
    const bool lmeds = method != 8;
    int niters;
    if (n == modelPoints) niters = 1;
    else if (lmeds) { niters = ransac_update_num_iters(prob, 0.45, modelPoints, maxIters); niters = niters > 3 ? niters : 3; }
    else niters = maxIters > 1 ? maxIters : 1;
    if (niters > kMaxHyp) { c->err = "max_iters exceeds the compiled hypothesis capacity (2048)"; return UVO_CAPACITY; }
    bool failed_first = false;
    int nsub;
    if (n == modelPoints) { for (int i = 0; i < 5; i++) w->h_subsets[i] = i; nsub = 1; }
    else nsub = make_subsets(w->h_subsets.data(), niters, modelPoints, n, lmeds ? 1000 : 10000, nullptr, nullptr, &failed_first);
    const int kFirstRound = 128;
    const int first = (!lmeds && n != modelPoints && nsub > kFirstRound) ? kFirstRound : nsub;
    const int npow2 = next_pow2(n);
    const float thr2 = (float)(threshold * threshold);
    // round(s) of hypotheses: subsets read from pinned memory, counts / medians / model counts written to it, the models mirrored
    auto round = [&](int from, int cnt) -> uvo_status {
        hipLaunchKernelGGL(k_fivepoint_hyp, dim3((cnt + kFpPerWg - 1) / kFpPerWg), dim3(64), 0, st, w->q1, w->q2, w->h_subsets.data() + (size_t)5 * from, cnt,
                           w->models + (size_t)90 * from, w->nmodels + from);
        if (n != modelPoints)
            hipLaunchKernelGGL(k_e_score, dim3(10, cnt), dim3(256), lmeds ? sizeof(float) * npow2 : 0, st, w->q1, w->q2, n, w->models + (size_t)90 * from, w->nmodels + from,
                               lmeds ? 1 : 0, thr2, npow2, w->h_counts.data() + (size_t)10 * from, w->h_medians.data() + (size_t)10 * from, w->h_nmodels.data() + from);
        else hipLaunchKernelGGL(k_mirror_words, dim3(1), dim3(256), 0, st, w->nmodels, w->h_nmodels.data(), 1);
        hipLaunchKernelGGL(k_mirror_words, dim3(std::min(64, (cnt * 180 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const int*>(w->models + (size_t)90 * from),
                           reinterpret_cast<int*>(w->h_models.data() + (size_t)90 * from), cnt * 180);
        UVO_HIP_TRY(c, hipGetLastError());
        UVO_HIP_TRY(c, host_sync(c, st));
        return UVO_OK;
    };
    UVO_TRY(round(0, first));
    Winner win;
    double final_thr = threshold;
    if (n == modelPoints) {
        if (w->h_nmodels[0] <= 0) { memset(w->h_mask.data(), 0, (size_t)n); return UVO_OK; }
        win.hyp = 0; win.model = 0;
        final_thr = DBL_MAX;                                         // every point an inlier (RANSACPointSetRegistrator::run, count == modelPoints)
    } else if (lmeds) win = replay_lmeds(w->h_nmodels.data(), w->h_medians.data(), 10, nsub);
    else {
        int settled = niters;
        win = replay_ransac(w->h_nmodels.data(), w->h_counts.data(), 10, first, niters, n, modelPoints, prob, &settled);
        if (first < nsub && settled > first) {
            UVO_TRY(round(first, nsub - first));
            win = replay_ransac(w->h_nmodels.data(), w->h_counts.data(), 10, nsub, niters, n, modelPoints, prob);
        }
    }
    if (win.hyp < 0) { memset(w->h_mask.data(), 0, (size_t)n); return UVO_OK; }
    if (lmeds && n != modelPoints) {
        const double sigma = 2.5 * 1.4826 * (1 + 5. / (n - modelPoints)) * sqrt(win.min_median);
        final_thr = sigma > 0.001 ? sigma : 0.001;
    }
    const double* E = w->h_models.data() + (size_t)win.hyp * 90 + win.model * 9;
    memcpy(out->E, E, sizeof(out->E));
    uint8_t* emask = w->masks + 4 * (size_t)c->cap;
    if (n == modelPoints) UVO_HIP_TRY(c, hipMemsetAsync(emask, 1, n, st));
    else hipLaunchKernelGGL(k_model_mask, dim3((n + 255) / 256), dim3(256), 0, st, 0, w->q1, w->q2, nullptr, nullptr, n,
                            w->models + (size_t)win.hyp * 90 + win.model * 9, (float)(final_thr * final_thr), emask);
    // recoverPose: the four candidates of decomposeEssentialMat, cheirality counts on the device
    double R1[9], R2[9], tt[3];
    decompose_essential(E, R1, R2, tt);
    const double* Rc[4] = { R1, R2, R1, R2 };
    const double sg[4] = { 1, 1, -1, -1 };
    PoseCands pc; PoseRt rt;
    for (int cnd = 0; cnd < 4; cnd++) {
        for (int i = 0; i < 3; i++) { pc.P[cnd][i*4] = Rc[cnd][i*3]; pc.P[cnd][i*4+1] = Rc[cnd][i*3+1]; pc.P[cnd][i*4+2] = Rc[cnd][i*3+2]; pc.P[cnd][i*4+3] = sg[cnd] * tt[i]; }
        memcpy(rt.R[cnd], Rc[cnd], sizeof(double) * 9);
        for (int i = 0; i < 3; i++) rt.t[cnd][i] = sg[cnd] * tt[i];
    }
    UVO_HIP_TRY(c, hipMemsetAsync(w->good, 0, sizeof(int) * 4, st));
    hipLaunchKernelGGL(k_recover_pose, dim3((n + kRpThreads - 1) / kRpThreads, 4), dim3(kRpThreads), 0, st, w->q1, w->q2, n, pc, emask, w->masks, w->good);
    w->h_pick[9] = 0;
    hipLaunchKernelGGL(k_mono_pick, dim3(1), dim3(1024), 0, st, c->d_x1, c->d_x2, n, emask, w->masks, w->good, w->in1, w->in2, w->pick, w->h_pick.data(), w->h_mask.data());
    UVO_HIP_TRY(c, hipGetLastError());
    // triangulatePoints + extract_3Dpoints (visual_odometry.h:355-356) on the inlier pairs, under the candidate k_mono_pick chose
    const double I[9] = {1,0,0,0,1,0,0,0,1}, z[3] = {0,0,0};
    double P_prev[12], P_cand[4][12];
    projection_matrix(I, z, K, P_prev);
    for (int cnd = 0; cnd < 4; cnd++) projection_matrix(rt.R[cnd], rt.t[cnd], K, P_cand[cnd]);
    UVO_TRY(pose_triangulate_extract3d_pick(c, P_prev, &P_cand[0][0], &rt.R[0][0], &rt.t[0][0], K, w->pick, w->in1, w->in2, w->pick + 1, n));
    hipLaunchKernelGGL(k_mono_scale, dim3(1), dim3(1024), 0, st, rt, w->pick, c->d_good_pts[0], c->d_counts + CN_G, w->h_zs.data(), w->h_pick.data());
    UVO_HIP_TRY(c, hipGetLastError());
    UVO_HIP_TRY(c, host_sync(c, st));
    if (w->h_pick[9] != 1) { c->err = "mono pose stage: the device chain did not report"; return UVO_HIP_ERROR; }
    const int best = w->h_pick[0];
    out->n_in = w->h_pick[5]; out->valid_inliers = w->h_pick[6]; out->good = w->h_pick[1 + best];
    out->G = w->h_pick[7]; out->n_front = w->h_pick[8];
    out->ok = lmeds && n != modelPoints ? (out->n_in >= modelPoints) : 1;
    memcpy(out->R, rt.R[best], sizeof(out->R)); memcpy(out->t, rt.t[best], sizeof(out->t));
    return UVO_OK;
}

}  // namespace uvo
