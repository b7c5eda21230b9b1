// uvo_epnp.h -- EPnP pose solver + Rodrigues for the PnP-RANSAC kernels (device side), in the
// operation order of OpenCV 4.5 calib3d (epnp.cpp compute_pose and helpers, calibration.cpp
// cvRodrigues2), as reached by cv::solvePnPRansac(flags = SOLVEPNP_EPNP) at
// visual_odometry.h:647-648.  One source, two execution policies:
//   GroupPolicy<G>: G 5-point hypotheses per wave, 8 lanes each, arrays in LDS interleaved by group.
//   BlockPolicy   : one n-point refit per workgroup.
// Parallelism never changes a result: independent accumulators / matrix entries / the three beta
// approximations go to different lanes, every floating-point SUM keeps the reference's sequential
// order inside one lane, and the 12x12 Jacobi sweep runs its 66 rotations in 21 dependency levels
// (rotations on disjoint row pairs commute exactly).
#pragma once
#include "uvo_linalg.h"

namespace uvo {

// GroupPolicy<G>: G problems per workgroup, 8 consecutive lanes cooperate on each (blockDim = 8*G,
// which must be ONE wave so that sync() is cheap).  Arrays live in LDS interleaved by group.
// The interleave stride is G + 1, not G: the lanes of a group walk rows of 12 doubles, and 12 rows x 8 groups x 2 banks is a multiple
// of the 64 LDS banks, so with stride G all six active lanes of a group hit one bank; with G + 1 they spread.
template <int G>
struct GroupPolicy {
    static constexpr int kStride = G + 1;
    using Arr = SArr<kStride>;
    static constexpr bool kStaged = false;
    __device__ static __forceinline__ int tid() { return threadIdx.x & 7; }
    __device__ static __forceinline__ int nth() { return 8; }
    __device__ static __forceinline__ void sync() { __syncthreads(); }
};
// One wave of a workgroup working alone on stride-1 arrays (lanes exchange data through LDS in program order, so a
// compiler-level barrier is all a "sync" needs): the 12 x 12 Jacobi sweeps of the block-cooperative refit.
struct WavePolicy {
    using Arr = SArr<1>;
    static constexpr bool kStaged = false;
    __device__ static __forceinline__ int tid() { return threadIdx.x & 63; }
    __device__ static __forceinline__ int nth() { return 64; }
    __device__ static __forceinline__ void sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
};
// BlockPolicy: one problem per workgroup, plain (stride-1) arrays.
struct BlockPolicy {
    using Arr = SArr<1>;
    static constexpr bool kStaged = true;     // long sums are staged through LDS (Epnp::stage)
    __device__ static __forceinline__ int tid() { return threadIdx.x; }
    __device__ static __forceinline__ int nth() { return blockDim.x; }
    __device__ static __forceinline__ void sync() { __syncthreads(); }
};

// Sequential floating-point sum acc = (((0 + f(0)) + f(1)) + ...) with the terms fetched eight at a
// time, so the (independent) loads and products of a batch overlap while the additions keep the
// reference's order exactly.
template <class F>
__device__ __forceinline__ double seq_sum(int n, F f)
{
    double acc = 0;
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        double t[8];
#pragma unroll
        for (int q = 0; q < 8; q++) t[q] = f(i + q);
#pragma unroll
        for (int q = 0; q < 8; q++) acc += t[q];
    }
    for (; i < n; i++) acc += f(i);
    return acc;
}

// acc = ((acc + f(0)) + f(1)) + ... + f(n-1) with the operands fetched sixteen at a time, one batch ahead of the
// additions, so that the LDS latency of batch k+1 hides behind the (dependent) additions of batch k.
template <class F>
__device__ __forceinline__ double seq_sum_pipelined(double acc, int n, F f)
{
    double cur[16], nxt[16];
    int i = 0;
    if (n >= 16) {
#pragma unroll
        for (int q = 0; q < 16; q++) cur[q] = f(q);
        for (; i + 32 <= n; i += 16) {
#pragma unroll
            for (int q = 0; q < 16; q++) nxt[q] = f(i + 16 + q);
#pragma unroll
            for (int q = 0; q < 16; q++) acc += cur[q];
#pragma unroll
            for (int q = 0; q < 16; q++) cur[q] = nxt[q];
        }
#pragma unroll
        for (int q = 0; q < 16; q++) acc += cur[q];
        i += 16;
    }
    for (; i < n; i++) acc += f(i);
    return acc;
}

// ---- cvRodrigues2 (calibration.cpp) ----
// vector -> matrix; R is 9 plain doubles
__host__ __device__ inline void rodrigues_vec2mat(const double* rv, double* R)
{
    double rx = rv[0], ry = rv[1], rz = rv[2];
    double theta = sqrt(rx*rx + ry*ry + rz*rz);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = 0;
        R[0] = R[4] = R[8] = 1;
        return;
    }
    double s, c; det_sincos(theta, &s, &c);
    double c1 = 1. - c;
    double itheta = theta ? 1./theta : 0.;
    rx *= itheta; ry *= itheta; rz *= itheta;
    const double rrt[9] = { rx*rx, rx*ry, rx*rz, rx*ry, ry*ry, ry*rz, rx*rz, ry*rz, rz*rz };
    const double r_x[9] = { 0, -rz, ry, rz, 0, -rx, -ry, rx, 0 };
    const double eye[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (int k = 0; k < 9; k++) R[k] = c*eye[k] + c1*rrt[k] + s*r_x[k];
}

// matrix -> vector.  Rin: 9 values (accessor), sc: 33 doubles of scratch (a9 v9 w3 wt3 R9)
template <class A>
__host__ __device__ void rodrigues_mat2vec(A Rin, A sc, double* rv)
{
    for (int i = 0; i < 9; i++) {
        double v = Rin[i];
        if (!(v >= -100 && v < 100)) { rv[0] = rv[1] = rv[2] = 0; return; }
    }
    A a = sc, v = sc + 9, w = sc + 18, wt = sc + 21, R = sc + 24;
    svd_square<3>(Rin, a, w, v, wt);
    // R = U * Vt ; U(i,k) = a[k*3+i]
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += a[k*3 + i]*v[k*3 + j];
        R[i*3 + j] = s;
    }
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx*rx + ry*ry + rz*rz)*0.25);
    double c = (R[0] + R[4] + R[8] - 1)*0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = det_acos(c);
    if (s < 1e-5) {
        double t;
        if (c > 0) rx = ry = rz = 0;
        else {
            t = (R[0] + 1)*0.5; rx = sqrt(t > 0. ? t : 0.);
            t = (R[4] + 1)*0.5; ry = sqrt(t > 0. ? t : 0.)*(R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1)*0.5; rz = sqrt(t > 0. ? t : 0.)*(R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry*rz > 0)) rz = -rz;
            theta /= sqrt(rx*rx + ry*ry + rz*rz);
            rx *= theta; ry *= theta; rz *= theta;
        }
    } else {
        double vth = 1/(2*s);
        vth *= theta;
        rx *= vth; ry *= vth; rz *= vth;
    }
    rv[0] = rx; rv[1] = ry; rv[2] = rz;
}

// cvProjectPoints2 with zero distortion: u = ((R X + t).x * (1/z)) * fx + cx
__host__ __device__ __forceinline__ void project_point(double X, double Y, double Z, const double* R, const double* t,
                                                       double fx, double fy, double cx, double cy, double* u, double* v)
{
    double x = R[0]*X + R[1]*Y + R[2]*Z + t[0];
    double y = R[3]*X + R[4]*Y + R[5]*Z + t[1];
    double z = R[6]*X + R[7]*Y + R[8]*Z + t[2];
    z = z ? 1./z : 1;
    x *= z; y *= z;
    *u = x*fx + cx;
    *v = y*fy + cy;
}

// ------------------------------------------------------------------------------------------
// 12x12 (generally MxN) Jacobi with U only, rotations scheduled by dependency level.
// In OpenCV's cyclic order (0,1),(0,2)..(0,N-1),(1,2).. rotation (i,j) depends only on the previous
// rotation that touched row i and the previous one that touched row j, which are (i,j-1)/(i-1,i) and
// (i-1,j): level(i,j) = i + j.  All rotations of one level touch disjoint rows, so they run on
// different lanes; levels run in order, which reproduces the sequential result bit for bit.
// `flag` is one double of shared scratch.  Needs P::nth() >= N/2 lanes.
// ------------------------------------------------------------------------------------------
template <class P, int M, int N>
__device__ void jacobi_svd_u_levels(typename P::Arr At, typename P::Arr W_out, typename P::Arr W, typename P::Arr flag)
{
    using A = typename P::Arr;
    const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
    const int lane = P::tid();
    const int max_iter = M > 30 ? M : 30;
    for (int i = lane; i < N; i += P::nth()) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; k++) { double t = At[i*M + k]; sd += t*t; }
        W[i] = sd;
    }
    P::sync();
#pragma unroll 1
    for (int iter = 0; iter < max_iter; iter++) {
        if (lane == 0) flag[0] = 0;
        P::sync();
#pragma unroll 1
        for (int t = 1; t <= 2*N - 3; t++) {
            const int i_lo = t - (N - 1) > 0 ? t - (N - 1) : 0, i_hi = (t - 1) / 2;
            const int i = i_lo + lane, j = t - i;
            if (i <= i_hi) {
                A Ai = At + i*M, Aj = At + j*M;
                double ai[M], aj[M];
#pragma unroll
                for (int k = 0; k < M; k++) { ai[k] = Ai[k]; aj[k] = Aj[k]; }
                double a = W[i], p = 0, b = W[j];
#pragma unroll
                for (int k = 0; k < M; k++) p += ai[k]*aj[k];
                if (!(fabs(p) <= eps*sqrt(a*b))) {
                    double c, s;
                    p *= 2;
                    double beta = a - b, gamma = det_hypot(p, beta);
                    if (beta < 0) {
                        double delta = (gamma - beta)*0.5;
                        s = sqrt(delta/gamma);
                        c = p/(gamma*s*2);
                    } else {
                        c = sqrt((gamma + beta)/(gamma*2));
                        s = p/(gamma*c*2);
                    }
                    a = b = 0;
#pragma unroll
                    for (int k = 0; k < M; k++) {
                        double t0 = c*ai[k] + s*aj[k];
                        double t1 = -s*ai[k] + c*aj[k];
                        Ai[k] = t0; Aj[k] = t1;
                        a += t0*t0; b += t1*t1;
                    }
                    W[i] = a; W[j] = b;
                    flag[0] = 1;
                }
            }
            P::sync();
        }
        const bool changed = flag[0] != 0;
        P::sync();
        if (!changed) break;
    }
    for (int i = lane; i < N; i += P::nth()) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; k++) { double t = At[i*M + k]; sd += t*t; }
        W[i] = sqrt(sd);
    }
    P::sync();
    if (lane == 0) {
        for (int i = 0; i < N-1; i++) {
            int j = i;
            for (int k = i+1; k < N; k++) if (W[j] < W[k]) j = k;
            if (i != j) {
                double t = W[i]; W[i] = W[j]; W[j] = t;
                for (int k = 0; k < M; k++) { t = At[i*M+k]; At[i*M+k] = At[j*M+k]; At[j*M+k] = t; }
            }
        }
        for (int i = 0; i < N; i++) W_out[i] = W[i];
        uint64_t rng = 0x12345678ULL;
        for (int i = 0; i < N; i++) {
            double sd = W[i];
            for (int ii = 0; ii < 100 && sd <= minval; ii++) {
                const double val0 = 1./M;
                for (int k = 0; k < M; k++) { double val = (rng_next(rng) & 256) != 0 ? val0 : -val0; At[i*M + k] = val; }
                for (int it = 0; it < 2; it++)
                    for (int j = 0; j < i; j++) {
                        sd = 0;
                        for (int k = 0; k < M; k++) sd += At[i*M + k]*At[j*M + k];
                        double asum = 0;
                        for (int k = 0; k < M; k++) { double t = At[i*M + k] - sd*At[j*M + k]; At[i*M + k] = t; asum += fabs(t); }
                        asum = asum > eps*100 ? 1/asum : 0;
                        for (int k = 0; k < M; k++) At[i*M + k] *= asum;
                    }
                sd = 0;
                for (int k = 0; k < M; k++) { double t = At[i*M + k]; sd += t*t; }
                sd = sqrt(sd);
            }
            double s = sd > minval ? 1/sd : 0.;
            for (int k = 0; k < M; k++) At[i*M + k] *= s;
        }
    }
    P::sync();
}

// ------------------------------------------------------------------------------------------
// EPnP.  Fixed-size state lives in `s` (EPNP_SMALL doubles): a common part and one block per beta
// approximation N = 1..3 (the three run on lanes 0..2 concurrently).  Per-point arrays: pws(3n)
// us(2n) alphas(4n) pcs(3 x 3n) tmp(3 x n).
// ------------------------------------------------------------------------------------------
enum {
    EP_CWS = 0, EP_MTM = 12, EP_D = 156, EP_WT = 168, EP_FLAG = 180, EP_L = 181, EP_RHO = 241,
    EP_PW0 = 247, EP_SC = 250,      // common scratch: 72 doubles
    EP_BR = 322,                    // per-branch blocks start here
    EPB_CCS = 0, EPB_BETAS = 12, EPB_REP = 16, EPB_RS = 17, EPB_TS = 26, EPB_PC0 = 29, EPB_ABT = 32, EPB_SC = 41,  // scratch: 100
    EPB_SIZE = 141,
    EPNP_SMALL = EP_BR + 3 * EPB_SIZE
};

template <class P>
struct Epnp {
    using Arr = typename P::Arr;
    double uc, vc, fu, fv;
    int n;
    Arr pws, us, alphas, pcs, tmp, s;         // pcs: 3 branches x 3n, tmp: 3 branches x n
    double* stage = nullptr;                  // BlockPolicy: LDS staging buffer of kStageDoubles
    long long* clk = nullptr;                 // diagnostic phase stamps (100 MHz); null = off
#define EP_STAMP(i) do { if (clk && P::tid() == 0) clk[i] = wall_clock64(); } while (0)
    static constexpr int kStageDoubles = 16384;          // 128 KB: few, large chunks (each costs a barrier and a memory round trip)

    // E independent sequential sums out[e] = (((0 + term(e,0)) + term(e,1)) + ...) over i < n.
    // Group policy: chain e runs on lane e (mod 8).  Block policy: all threads evaluate the terms of a
    // chunk in parallel into LDS (coalesced loads), then lane e adds its chunk in index order; the
    // running sums stay in the lanes' registers across chunks, so the order of additions is the
    // reference's.
    template <class TermF, class StoreF>
    __device__ __forceinline__ void multi_sum(int E, int n_terms, TermF term, StoreF store)
    {
        if (!P::kStaged) {
            for (int ch = P::tid(); ch < E; ch += P::nth()) store(ch, seq_sum(n_terms, [&](int i) { return term(ch, i); }));
            P::sync();
        } else {
            // two half-size staging buffers: while the lanes of wave 0 add chunk c in order, the other waves evaluate the
            // terms of chunk c + 1 (one barrier per chunk)
            int CH = (kStageDoubles / 2 / E - 1) & ~7;
            if (CH > 512) CH = 512;
            const int CHS = CH + 1;
            const int tid = P::tid(), nth = P::nth();
            const int ft = tid - 64, fnth = nth - 64;          // fill threads: waves 1.. (all threads for the first chunk)
            double acc = 0;
            auto fill = [&](double* buf, int base, int t0, int tn) {
                const int cnt = n_terms - base < CH ? n_terms - base : CH;
                const int total = E * cnt;
                const float inv_cnt = 1.0f / cnt;                  // idx / cnt by multiplication: exact for idx < 2^20
                for (int idx0 = t0; idx0 < total; idx0 += 16 * tn) {   // sixteen independent terms in flight per thread
                    double t[16]; int off[16];
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        const int idx = idx0 + q * tn;
                        int ch = (int)((idx + 0.5f) * inv_cnt);
                        if (ch > E - 1) ch = E - 1;
                        const int i = idx - ch * cnt;
                        off[q] = idx < total ? ch * CHS + i : -1;
                        t[q] = idx < total ? term(ch, base + i) : 0.0;
                    }
#pragma unroll
                    for (int q = 0; q < 16; q++) if (off[q] >= 0) buf[off[q]] = t[q];
                }
            };
            double* buf0 = stage; double* buf1 = stage + kStageDoubles / 2;
            fill(buf0, 0, tid, nth);
            P::sync();
            int par = 0;
            for (int base = 0; base < n_terms; base += CH, par ^= 1) {
                double* cur = par ? buf1 : buf0; double* nxt = par ? buf0 : buf1;
                const int cnt = n_terms - base < CH ? n_terms - base : CH;
                if (tid >= 64) { if (base + CH < n_terms) fill(nxt, base + CH, ft, fnth); }
                else if (tid < E) {
                    const double* row = cur + tid * CHS;
                    acc = seq_sum_pipelined(acc, cnt, [&](int i) { return row[i]; });
                }
                P::sync();
            }
            if (tid < E) store(tid, acc);
            P::sync();
        }
    }

    __device__ __forceinline__ Arr br(int b) const { return s + (EP_BR + b * EPB_SIZE); }
    __device__ __forceinline__ double dot3(Arr a, Arr b) const { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
    __device__ __forceinline__ double dist2(Arr p1, Arr p2) const
    {
        return (p1[0]-p2[0])*(p1[0]-p2[0]) + (p1[1]-p2[1])*(p1[1]-p2[1]) + (p1[2]-p2[2])*(p1[2]-p2[2]);
    }
    // element (k, c) of the 2n x 12 matrix M of fill_M
    __device__ __forceinline__ double Mval(int k, int c) const
    {
        int p = k >> 1, r = k & 1, a = c / 3, q = c - 3*a;
        double as = alphas[4*p + a];
        if (r == 0) return q == 0 ? as * fu : q == 1 ? 0.0 : as * (uc - us[2*p]);
        return q == 0 ? 0.0 : q == 1 ? as * fv : as * (vc - us[2*p + 1]);
    }

    __device__ void choose_control_points()
    {
        Arr cws = s + EP_CWS, sc = s + EP_SC;
        multi_sum(3, n, [&](int j, int i) { return pws[3*i + j]; }, [&](int j, double acc) { cws[j] = acc / n; });
        // PW0^T PW0 (upper triangle, sequential over points), then mirrored
        Arr ptp = sc;                         // 9
        multi_sum(6, n,
                  [&](int e, int k) {
                      int a = e < 3 ? 0 : e < 5 ? 1 : 2, b = e < 3 ? e : e < 5 ? e - 2 : 2;
                      return (pws[3*k + a] - cws[a]) * (pws[3*k + b] - cws[b]);
                  },
                  [&](int e, double s0) {
                      int a = e < 3 ? 0 : e < 5 ? 1 : 2, b = e < 3 ? e : e < 5 ? e - 2 : 2;
                      ptp[a*3 + b] = s0; ptp[b*3 + a] = s0;
                  });
        if (P::tid() == 0) {
            Arr at = sc + 9, dc = sc + 18, vt = sc + 21, wt = sc + 30;
            svd_square<3>(ptp, at, dc, vt, wt);       // rows of `at` = U^T = uct
            for (int i = 1; i < 4; i++) {
                double k = sqrt(dc[i-1] / n);
                for (int j = 0; j < 3; j++) cws[3*i + j] = cws[j] + k * at[3*(i-1) + j];
            }
        }
        P::sync();
    }

    __device__ void compute_barycentric_coordinates()
    {
        Arr cws = s + EP_CWS, sc = s + EP_SC;
        Arr cc = sc, ci = sc + 9;
        if (P::tid() == 0) {
            for (int i = 0; i < 3; i++) for (int j = 1; j < 4; j++) cc[3*i + j - 1] = cws[3*j + i] - cws[i];
            invert3_svd(cc, ci, sc + 18, sc + 27, sc + 36, sc + 39);
        }
        P::sync();
        for (int i = P::tid(); i < n; i += P::nth()) {
            Arr pi = pws + 3*i, a = alphas + 4*i;
            for (int j = 0; j < 3; j++)
                a[1 + j] = ci[3*j] * (pi[0] - cws[0]) + ci[3*j + 1] * (pi[1] - cws[1]) + ci[3*j + 2] * (pi[2] - cws[2]);
            a[0] = 1.0f - a[1] - a[2] - a[3];
        }
        P::sync();
    }

    __device__ void build_mtm()
    {
        Arr mtm = s + EP_MTM;
        auto entry = [](int e, int* pi, int* pj) {
            int i = 0, rem = e;
            while (rem >= 12 - i) { rem -= 12 - i; i++; }
            *pi = i; *pj = i + rem;
        };
        // mulTransposed (MulTransposedR): upper triangle, each entry a sequential sum over the 2n rows of M
        if (!P::kStaged) {
            // M (2n x 12) once into the still unused per-branch blocks, then lane l owns entries l, l + nth, ...
            Arr M = s + EP_BR;
            const int rows = 2*n;                                 // <= 3*EPB_SIZE/12 = 35 rows: the 5-point hypotheses
            for (int idx = P::tid(); idx < rows*12; idx += P::nth()) { int k = idx / 12; M[idx] = Mval(k, idx - 12*k); }
            P::sync();
            for (int e = P::tid(); e < 78; e += P::nth()) {
                int i, j; entry(e, &i, &j);
                const double s0 = seq_sum(rows, [&](int k) { return M[k*12 + i] * M[k*12 + j]; });
                mtm[i*12 + j] = s0; mtm[j*12 + i] = s0;
            }
            P::sync();
        } else {
            // rows of M (fill_M) are produced chunk-wise into LDS by all threads; lane e < 78 owns entry (i,j)
            // and walks the rows in order
            constexpr int PTS = kStageDoubles / 24;            // points per chunk (2 rows of 12 each)
            const int tid = P::tid();
            int ei = 0, ej = 0;
            if (tid < 78) entry(tid, &ei, &ej);
            double acc = 0;
            for (int base = 0; base < n; base += PTS) {
                const int cnt = n - base < PTS ? n - base : PTS;
                for (int p = tid; p < cnt; p += P::nth()) {
                    double* M1 = stage + (2*p)*12; double* M2 = M1 + 12;
                    Arr as = alphas + 4*(base + p);
                    double u = us[2*(base + p)], v = us[2*(base + p) + 1];
                    for (int a = 0; a < 4; a++) {
                        double al = as[a];
                        M1[3*a] = al * fu; M1[3*a + 1] = 0.0; M1[3*a + 2] = al * (uc - u);
                        M2[3*a] = 0.0; M2[3*a + 1] = al * fv; M2[3*a + 2] = al * (vc - v);
                    }
                }
                P::sync();
                if (tid < 78) {
                    const double* ri = stage + ei; const double* rj = stage + ej;
                    const int rows = 2*cnt;
                    acc = seq_sum_pipelined(acc, rows, [&](int k) { return ri[k*12] * rj[k*12]; });
                }
                P::sync();
            }
            if (tid < 78) { mtm[ei*12 + ej] = acc; mtm[ej*12 + ei] = acc; }
            P::sync();
        }
    }

    // compute_L_6x10: 24 difference vectors, then 60 independent dot products -- spread over the policy's lanes
    // (every element is computed by one lane with the reference's expression; nothing is summed across lanes)
    __device__ void compute_L_6x10(Arr ut, Arr l)
    {
        Arr dv = s + EP_SC;                   // dv[4][6][3] = 72
        for (int e = P::tid(); e < 24; e += P::nth()) {
            const int i = e / 6, j = e - 6*i;
            // pairs (a, b) in the order (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
            const int a = j < 3 ? 0 : (j < 5 ? 1 : 2), b = j < 3 ? j + 1 : (j < 5 ? j - 1 : 3);
            Arr v = ut + 12*(11 - i);
            dv[e*3 + 0] = v[3*a] - v[3*b];
            dv[e*3 + 1] = v[3*a + 1] - v[3*b + 1];
            dv[e*3 + 2] = v[3*a + 2] - v[3*b + 2];
        }
        P::sync();
#define UVO_DV(i, j) (dv + ((i)*6 + (j))*3)
        for (int e = P::tid(); e < 60; e += P::nth()) {
            const int i = e / 10, c = e - 10*i;
            // column c <-> (p, q) with p <= q: 0:(0,0) 1:(0,1) 2:(1,1) 3:(0,2) 4:(1,2) 5:(2,2) 6:(0,3) 7:(1,3) 8:(2,3) 9:(3,3)
            const int q = c < 1 ? 0 : (c < 3 ? 1 : (c < 6 ? 2 : 3));
            const int p0 = c - (q*(q + 1))/2;
            const double d = dot3(UVO_DV(p0, i), UVO_DV(q, i));
            l[10*i + c] = p0 == q ? d : 2.0f * d;
        }
#undef UVO_DV
        P::sync();
    }

    // find_betas_approx_{1,2,3}: cvSolve(L_6xK, Rho, B, CV_SVD) with K = 4, 3, 5 columns picked from L_6x10.
    // Written shape-generic so that lanes 0..2 run the three approximations in one instruction stream.
    // scratch: Ls(30) a(30) v(25) w(5) wt(5) b(5)
    __device__ void find_betas(int which, Arr L, Arr rho, Arr betas, Arr sc)
    {
        Arr Ls = sc, a = sc + 30, v = sc + 60, w = sc + 85, wt = sc + 90, b = sc + 95;
        const int K = which == 1 ? 4 : which == 2 ? 3 : 5;
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < K; j++) {
                int col = which == 1 ? (j == 0 ? 0 : j == 1 ? 1 : j == 2 ? 3 : 6) : j;
                Ls[K*i + j] = L[10*i + col];
            }
        // cv::solve(DECOMP_SVD): a = Ls^T, JacobiSVD, back-substitution
        for (int i = 0; i < 6; i++) for (int j = 0; j < K; j++) a[j*6 + i] = Ls[i*K + j];
        jacobi_svd_rt6(a, w, v, wt, K);
        svbksb_vec(6, K, w, a, 6, v, K, rho, b);
        if (which == 1) {
            if (b[0] < 0) { betas[0] = sqrt(-b[0]); betas[1] = -b[1] / betas[0]; betas[2] = -b[2] / betas[0]; betas[3] = -b[3] / betas[0]; }
            else          { betas[0] = sqrt(b[0]);  betas[1] = b[1] / betas[0];  betas[2] = b[2] / betas[0];  betas[3] = b[3] / betas[0]; }
        } else {
            if (b[0] < 0) { betas[0] = sqrt(-b[0]); betas[1] = (b[2] < 0) ? sqrt(-b[2]) : 0.0; }
            else          { betas[0] = sqrt(b[0]);  betas[1] = (b[2] > 0) ? sqrt(b[2]) : 0.0; }
            if (b[1] < 0) betas[0] = -betas[0];
            betas[2] = which == 2 ? 0.0 : b[3] / betas[0];
            betas[3] = 0.0;
        }
    }

    // epnp.cpp qr_solve, nr = 6, nc = 4 (pA 24, pb 6, pX 4, A1 4, A2 4)
    __device__ void qr_solve(Arr pA, Arr pb, Arr pX, Arr A1, Arr A2)
    {
        const int nr = 6, nc = 4;
        int kk = 0;                                // index of A[k][k]
        for (int k = 0; k < nc; k++) {
            int p1 = kk; double eta = fabs(pA[p1]);
            for (int i = k + 1; i < nr; i++) { double elt = fabs(pA[p1]); if (eta < elt) eta = elt; p1 += nc; }
            if (eta == 0) { A1[k] = A2[k] = 0.0; return; }
            else {
                int p2 = kk; double sum2 = 0.0, inv_eta = 1. / eta;
                for (int i = k; i < nr; i++) { double t = pA[p2] * inv_eta; pA[p2] = t; sum2 += t * t; p2 += nc; }
                double sigma = sqrt(sum2);
                if (pA[kk] < 0) sigma = -sigma;
                pA[kk] += sigma;
                A1[k] = sigma * pA[kk];
                A2[k] = -eta * sigma;
                for (int j = k + 1; j < nc; j++) {
                    int p = kk; double sum = 0;
                    for (int i = k; i < nr; i++) { sum += pA[p] * pA[p + j - k]; p += nc; }
                    double tau = sum / A1[k];
                    p = kk;
                    for (int i = k; i < nr; i++) { pA[p + j - k] -= tau * pA[p]; p += nc; }
                }
            }
            kk += nc + 1;
        }
        int jj = 0;
        for (int j = 0; j < nc; j++) {
            int p = jj; double tau = 0;
            for (int i = j; i < nr; i++) { tau += pA[p] * pb[i]; p += nc; }
            tau /= A1[j];
            p = jj;
            for (int i = j; i < nr; i++) { pb[i] -= tau * pA[p]; p += nc; }
            jj += nc + 1;
        }
        pX[nc - 1] = pb[nc - 1] / A2[nc - 1];
        for (int i = nc - 2; i >= 0; i--) {
            int p = i*nc + (i + 1); double sum = 0;
            for (int j = i + 1; j < nc; j++) { sum += pA[p] * pX[j]; p++; }
            pX[i] = (pb[i] - sum) / A2[i];
        }
    }

    // epnp.cpp qr_solve (nr = 6, nc = 4) on register arrays: every index is a compile-time constant after unrolling, so
    // the Householder steps run without a memory round trip per element.  Same operations, same order; X is left
    // untouched when a column vanishes (the reference returns before writing it).
    __device__ __forceinline__ void qr_solve_reg(double (&pA)[24], double (&pb)[6], double (&pX)[4])
    {
        constexpr int nr = 6, nc = 4;
        double A1[nc], A2[nc];
        bool singular = false;
#pragma unroll
        for (int k = 0; k < nc; k++) {
            if (singular) break;
            const int kk = k * (nc + 1);
            double eta = fabs(pA[kk]);
#pragma unroll
            for (int i = k + 1; i < nr; i++) { double elt = fabs(pA[kk + (i - k - 1) * nc]); if (eta < elt) eta = elt; }   // epnp.cpp reads before it advances: rows k .. nr-2
            if (eta == 0) { singular = true; break; }
            double sum2 = 0.0; const double inv_eta = 1. / eta;
#pragma unroll
            for (int i = k; i < nr; i++) { double t = pA[kk + (i - k) * nc] * inv_eta; pA[kk + (i - k) * nc] = t; sum2 += t * t; }
            double sigma = sqrt(sum2);
            if (pA[kk] < 0) sigma = -sigma;
            pA[kk] += sigma;
            A1[k] = sigma * pA[kk];
            A2[k] = -eta * sigma;
#pragma unroll
            for (int j = k + 1; j < nc; j++) {
                double sum = 0;
#pragma unroll
                for (int i = k; i < nr; i++) sum += pA[kk + (i - k) * nc] * pA[kk + (i - k) * nc + j - k];
                const double tau = sum / A1[k];
#pragma unroll
                for (int i = k; i < nr; i++) pA[kk + (i - k) * nc + j - k] -= tau * pA[kk + (i - k) * nc];
            }
        }
        if (singular) return;
#pragma unroll
        for (int j = 0; j < nc; j++) {
            const int jj = j * (nc + 1);
            double tau = 0;
#pragma unroll
            for (int i = j; i < nr; i++) tau += pA[jj + (i - j) * nc] * pb[i];
            tau /= A1[j];
#pragma unroll
            for (int i = j; i < nr; i++) pb[i] -= tau * pA[jj + (i - j) * nc];
        }
        pX[nc - 1] = pb[nc - 1] / A2[nc - 1];
#pragma unroll
        for (int i = nc - 2; i >= 0; i--) {
            double sum = 0;
#pragma unroll
            for (int j = i + 1; j < nc; j++) sum += pA[i*nc + j] * pX[j];
            pX[i] = (pb[i] - sum) / A2[i];
        }
    }

    __device__ void gauss_newton(Arr L, Arr rho, Arr betas, Arr sc)
    {
        (void)sc;
        double Lr[60], rh[6], be[4], x[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 60; i++) Lr[i] = L[i];
#pragma unroll
        for (int i = 0; i < 6; i++) rh[i] = rho[i];
#pragma unroll
        for (int i = 0; i < 4; i++) be[i] = betas[i];
#pragma unroll 1
        for (int it = 0; it < 5; it++) {
            const double b0 = be[0], b1 = be[1], b2 = be[2], b3 = be[3];
            double A[24], b[6];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const double* rowL = Lr + i*10;
                A[i*4 + 0] = 2*rowL[0]*b0 +   rowL[1]*b1 +   rowL[3]*b2 +   rowL[6]*b3;
                A[i*4 + 1] =   rowL[1]*b0 + 2*rowL[2]*b1 +   rowL[4]*b2 +   rowL[7]*b3;
                A[i*4 + 2] =   rowL[3]*b0 +   rowL[4]*b1 + 2*rowL[5]*b2 +   rowL[8]*b3;
                A[i*4 + 3] =   rowL[6]*b0 +   rowL[7]*b1 +   rowL[8]*b2 + 2*rowL[9]*b3;
                b[i] = rh[i] -
                    (rowL[0]*b0*b0 + rowL[1]*b0*b1 + rowL[2]*b1*b1 +
                     rowL[3]*b0*b2 + rowL[4]*b1*b2 + rowL[5]*b2*b2 +
                     rowL[6]*b0*b3 + rowL[7]*b1*b3 + rowL[8]*b2*b3 +
                     rowL[9]*b3*b3);
            }
            qr_solve_reg(A, b, x);
#pragma unroll
            for (int i = 0; i < 4; i++) be[i] += x[i];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) betas[i] = be[i];
    }

    // epnp::compute_pose followed by Rodrigues(R, rvec).  Outputs valid on tid 0.
    __device__ void compute_pose(double* rvec, double* tvec)
    {
        EP_STAMP(0);
        choose_control_points();
        EP_STAMP(1);
        compute_barycentric_coordinates();
        EP_STAMP(2);
        build_mtm();
        EP_STAMP(3);
        Arr mtm = s + EP_MTM, L = s + EP_L, rho = s + EP_RHO, cws = s + EP_CWS, pw0 = s + EP_PW0;
        // cvSVD(MtM, D, Ut, 0, MODIFY_A | U_T): MtM is symmetric so At = MtM^T is MtM itself; rows -> U^T
        if constexpr (P::kStaged) {                   // block policy: 21 levels x ~8 sweeps of workgroup barriers cost more than they buy
            if (P::tid() < 64) jacobi_svd_u_levels<WavePolicy, 12, 12>(mtm, s + EP_D, s + EP_WT, s + EP_FLAG);
            P::sync();
        } else jacobi_svd_u_levels<P, 12, 12>(mtm, s + EP_D, s + EP_WT, s + EP_FLAG);
        const int tid = P::tid();
        EP_STAMP(4);
        compute_L_6x10(mtm, L);
        EP_STAMP(10);
        if (tid == 0) {
            rho[0] = dist2(cws, cws + 3); rho[1] = dist2(cws, cws + 6); rho[2] = dist2(cws, cws + 9);
            rho[3] = dist2(cws + 3, cws + 6); rho[4] = dist2(cws + 3, cws + 9); rho[5] = dist2(cws + 6, cws + 9);
        }
        P::sync();
        // ---- the three beta approximations, one per lane: find_betas + gauss_newton + compute_ccs ----
        if (tid < 3) {
            Arr B = br(tid), betas = B + EPB_BETAS, ccs = B + EPB_CCS;
            find_betas(tid + 1, L, rho, betas, B + EPB_SC);
            EP_STAMP(11);
            gauss_newton(L, rho, betas, B + EPB_SC);
            EP_STAMP(12);
            for (int i = 0; i < 12; i++) ccs[i] = 0.0f;
            for (int i = 0; i < 4; i++) {
                Arr v = mtm + 12*(11 - i);
                double bi = betas[i];
                for (int j = 0; j < 4; j++) for (int k = 0; k < 3; k++) ccs[3*j + k] += bi * v[3*j + k];
            }
        }
        P::sync();
        EP_STAMP(5);
        // compute_pcs for the three branches
        for (int it = tid; it < 3*n; it += P::nth()) {
            int b = it / n, i = it - b*n;
            Arr ccs = br(b) + EPB_CCS, a = alphas + 4*i, pc = pcs + 3*(b*n + i);
            for (int j = 0; j < 3; j++) pc[j] = a[0]*ccs[j] + a[1]*ccs[3 + j] + a[2]*ccs[6 + j] + a[3]*ccs[9 + j];
        }
        P::sync();
        // solve_for_sign
        bool flip[3];
        for (int b = 0; b < 3; b++) flip[b] = pcs[3*(b*n) + 2] < 0.0;
        P::sync();
        if (flip[0] || flip[1] || flip[2]) {
            if (tid < 3 && flip[tid]) { Arr ccs = br(tid) + EPB_CCS; for (int i = 0; i < 12; i++) ccs[i] = -ccs[i]; }
            for (int it = tid; it < 3*n; it += P::nth()) {
                int b = it / n;
                if (flip[b]) { Arr pc = pcs + 3*it; pc[0] = -pc[0]; pc[1] = -pc[1]; pc[2] = -pc[2]; }
            }
            P::sync();
        }
        EP_STAMP(6);
        // estimate_R_and_t: centroids (pw0 does not depend on the branch), 3x3 covariances
        multi_sum(12, n,
                  [&](int e, int i) {
                      if (e < 3) return pws[3*i + e];
                      int b = (e - 3) / 3, j = (e - 3) - 3*b;
                      return pcs[3*(b*n + i) + j];
                  },
                  [&](int e, double acc) {
                      if (e < 3) pw0[e] = acc / n;
                      else { int b = (e - 3) / 3, j = (e - 3) - 3*b; (br(b) + EPB_PC0)[j] = acc / n; }
                  });
        multi_sum(27, n,
                  [&](int e, int i) {
                      int b = e / 9, r = e - 9*b, j = r / 3, q = r - 3*j;
                      return (pcs[3*(b*n + i) + j] - (br(b) + EPB_PC0)[j]) * (pws[3*i + q] - pw0[q]);
                  },
                  [&](int e, double acc) { int b = e / 9, r = e - 9*b; (br(b) + EPB_ABT)[r] = acc; });
        EP_STAMP(7);
        if (tid < 3) {
            Arr B = br(tid), abt = B + EPB_ABT, R = B + EPB_RS, t = B + EPB_TS, pc0 = B + EPB_PC0, sc = B + EPB_SC;
            Arr at = sc, w = sc + 9, vt = sc + 12, wt = sc + 21;
            svd_square<3>(abt, at, w, vt, wt);
            // R[i][j] = dot(U row i, V row j) = sum_k U(i,k) V(j,k) = sum_k at[k][i] * vt[k][j]
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
                R[3*i + j] = at[0*3 + i]*vt[0*3 + j] + at[1*3 + i]*vt[1*3 + j] + at[2*3 + i]*vt[2*3 + j];
            const double det =
                R[0]*R[4]*R[8] + R[1]*R[5]*R[6] + R[2]*R[3]*R[7] -
                R[2]*R[4]*R[6] - R[1]*R[3]*R[8] - R[0]*R[5]*R[7];
            if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
            t[0] = pc0[0] - dot3(R, pw0);
            t[1] = pc0[1] - dot3(R + 3, pw0);
            t[2] = pc0[2] - dot3(R + 6, pw0);
        }
        P::sync();
        EP_STAMP(8);
        // reprojection_error: per-point terms in parallel, summed in order by one lane per branch
        for (int it = tid; it < 3*n; it += P::nth()) {
            int b = it / n, i = it - b*n;
            Arr R = br(b) + EPB_RS, t = br(b) + EPB_TS, pw = pws + 3*i;
            double Xc = dot3(R, pw) + t[0];
            double Yc = dot3(R + 3, pw) + t[1];
            double inv_Zc = 1.0 / (dot3(R + 6, pw) + t[2]);
            double ue = uc + fu * Xc * inv_Zc;
            double ve = vc + fv * Yc * inv_Zc;
            double u = us[2*i], v = us[2*i + 1];
            tmp[it] = sqrt((u - ue)*(u - ue) + (v - ve)*(v - ve));
        }
        P::sync();
        multi_sum(3, n, [&](int b, int i) { return tmp[b*n + i]; }, [&](int b, double sum2) { (br(b) + EPB_REP)[0] = sum2 / n; });
        if (tid == 0) {
            double rep1 = (br(0) + EPB_REP)[0], rep2 = (br(1) + EPB_REP)[0], rep3 = (br(2) + EPB_REP)[0];
            int N = 1; double repN = rep1;
            if (rep2 < rep1) { N = 2; repN = rep2; }
            if (rep3 < repN) N = 3;
            Arr B = br(N - 1), ts = B + EPB_TS;
            tvec[0] = ts[0]; tvec[1] = ts[1]; tvec[2] = ts[2];
            rodrigues_mat2vec(B + EPB_RS, B + EPB_SC, rvec);
        }
        P::sync();
        EP_STAMP(9);
    }
#undef EP_STAMP
};

}  // namespace uvo
