// uvo_epnp_fast.h -- the inlier REFIT of solvePnPRansac (visual_odometry.h:647-648 -> solvepnp.cpp: EPnP on all inliers
// after the RANSAC loop) as a workgroup-parallel kernel.
//
// The RANSAC hypotheses (k_pnp_hyp) are bit-exact restatements of OpenCV's epnp.cpp because the inlier MASK depends on
// them.  The refit is different: it runs after the mask is fixed, its only output is the pose, and the pose's contract is
// 1e-4 relative (BASELINE.json north_star) -- so its long floating-point sums need not keep OpenCV's sequential order.
// Same algorithm (epnp::compute_pose: PCA control points, barycentric coordinates, M^T M, its four smallest eigenvectors,
// the three beta approximations + Gauss-Newton, absolute orientation, pick the smallest reprojection error), but:
//   * every sum over the points is a tree reduction over 1024 threads (wave DPP reduction, then 16 partials in LDS);
//   * M^T M (2n x 12 -> 12 x 12) runs on the fp64 matrix pipe, v_mfma_f64_16x16x4_f64, A = B = a 4-row slab of M;
//   * the 12 x 12 Jacobi runs 6 disjoint rotations at a time (round-robin ordering, 11 rounds per sweep, 8 lanes per
//     rotation) with reciprocal / rsqrt seeds + Newton steps instead of IEEE divisions and square roots -- a rotation only
//     has to be orthonormal to working precision, not correctly rounded;
//   * the 6 x K least-squares systems of the beta approximations are solved by Householder QR in registers (SVD fall-back
//     when a diagonal entry of R collapses), the 3 x 3 SVDs by the same fast Jacobi.
// Results agree with the sequential refit (k_pnp_refit, kept for fewer than kFastRefitMin inliers, where M^T M is (nearly)
// rank-deficient and the null-space basis is ordering-dependent) to ~1e-11 relative on the bench scene; tests hold 1e-4.
#pragma once
#include "uvo_epnp.h"

namespace uvo {

static const int kFastRefitMin = 24;        // fewer inliers: the sequential, OpenCV-ordered refit

typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// sum over the 8 lanes of an aligned octet, result on all 8: quad xor 1, quad xor 2, mirror inside the octet
__device__ __forceinline__ double octet_sum(double v)
{
    v += dpp_mov_f64<0xB1>(v);              // quad_perm [1,0,3,2]
    v += dpp_mov_f64<0x4E>(v);              // quad_perm [2,3,0,1]
    v += dpp_mov_f64<0x141>(v);             // row_half_mirror
    return v;
}
// sum over the wave, result on all lanes
__device__ __forceinline__ double wave_sum(double v)
{
    v = octet_sum(v);
    v += dpp_mov_f64<0x140>(v);             // row_mirror: the other octet of the row of 16
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// 1/d and 1/sqrt(d) from the hardware seeds (~2^-23 relative) and Newton steps; STEPS = 1: ~2^-45, 2: working precision
template <int STEPS>
__device__ __forceinline__ double fast_rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int i = 0; i < STEPS; i++) r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    return r;
}
template <int STEPS>
__device__ __forceinline__ double fast_rsq(double d)
{
    double y = __builtin_amdgcn_rsq(d);
#pragma unroll
    for (int i = 0; i < STEPS; i++) { const double h = 0.5 * d * y; y = __builtin_fma(y, __builtin_fma(-h, y, 0.5), y); }
    return y;
}
// The Jacobi rotation of OpenCV's JacobiSVDImpl_ for two rows with squared norms a, b and inner product p:
// x' = c x + s y, y' = -s x + c y, with ITS choice of angle (for a < b the rotation also moves the longer row first), so
// that the decompositions end up with the vectors in the order and with the signs the sequential solver gets -- EPnP's
// result depends on the sign of the PCA axes that place the control points.  Fast arithmetic: rsqrt seeds + two Newton
// steps instead of hypot / divisions / square roots (c^2 + s^2 = 1 to working precision).
__device__ __forceinline__ void fast_rotation(double a, double b, double p, double& c, double& s)
{
    const double p2 = 2.0 * p, beta = a - b;
    const double g2 = __builtin_fma(p2, p2, beta * beta);
    const double rg = fast_rsq<2>(g2);                // 1 / gamma
    const double gamma = g2 * rg;
    if (beta < 0) {
        const double q = (gamma - beta) * 0.5 * rg;   // delta / gamma, in [0.5, 1]
        const double rs = fast_rsq<2>(q);
        s = q * rs;                                   // sqrt(delta / gamma)
        c = p * rg * rs;                              // (2p) / (gamma s 2)
    } else {
        const double q = (gamma + beta) * 0.5 * rg;
        const double rc = fast_rsq<2>(q);
        c = q * rc;
        s = p * rg * rc;
    }
}

// One-sided Jacobi SVD of a 3 x 3 matrix A (row-major) on one lane, in registers: u[k] / v[k] = k-th left / right singular
// vector, w descending.  A vanishing singular value leaves u[k] = 0.
__device__ inline void fast_svd3(const double* A, double (&w)[3], double (&u)[3][3], double (&v)[3][3])
{
    const double eps2 = (DBL_EPSILON * 10) * (DBL_EPSILON * 10);
    double at[3][3];                         // row k = column k of A
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 3; i++) { at[k][i] = A[i*3 + k]; v[k][i] = i == k ? 1.0 : 0.0; }
#pragma unroll 1
    for (int sweep = 0; sweep < 30; sweep++) {
        bool changed = false;
#pragma unroll
        for (int pr = 0; pr < 3; pr++) {
            const int i = pr == 2 ? 1 : 0, j = pr == 0 ? 1 : 2;
            const double a = at[i][0]*at[i][0] + at[i][1]*at[i][1] + at[i][2]*at[i][2];
            const double b = at[j][0]*at[j][0] + at[j][1]*at[j][1] + at[j][2]*at[j][2];
            const double p = at[i][0]*at[j][0] + at[i][1]*at[j][1] + at[i][2]*at[j][2];
            if (p * p > eps2 * a * b) {
                double c, s;
                fast_rotation(a, b, p, c, s);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const double x = at[i][k], y = at[j][k];
                    at[i][k] = c*x + s*y; at[j][k] = c*y - s*x;
                    const double vx = v[i][k], vy = v[j][k];
                    v[i][k] = c*vx + s*vy; v[j][k] = c*vy - s*vx;
                }
                changed = true;
            }
        }
        if (!changed) break;
    }
    double nrm[3];
#pragma unroll
    for (int k = 0; k < 3; k++) nrm[k] = sqrt(at[k][0]*at[k][0] + at[k][1]*at[k][1] + at[k][2]*at[k][2]);
    // descending order (three compare-exchanges on (nrm, at, v))
#pragma unroll
    for (int pass = 0; pass < 3; pass++) {
        const int i = pass == 1 ? 1 : 0, j = pass == 1 ? 2 : 1;          // (0,1) (1,2) (0,1)
        if (nrm[i] < nrm[j]) {
            double t = nrm[i]; nrm[i] = nrm[j]; nrm[j] = t;
#pragma unroll
            for (int k = 0; k < 3; k++) { t = at[i][k]; at[i][k] = at[j][k]; at[j][k] = t; t = v[i][k]; v[i][k] = v[j][k]; v[j][k] = t; }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        w[k] = nrm[k];
        const double inv = nrm[k] > DBL_MIN ? 1.0 / nrm[k] : 0.0;
#pragma unroll
        for (int i = 0; i < 3; i++) u[k][i] = at[k][i] * inv;
    }
}

// Jacobi eigen-decomposition of the symmetric 12 x 12 matrix G (LDS, row-major) by ONE wave: one-sided rotations of its rows,
// six disjoint pairs at a time.  Afterwards row i of G = i-th eigenvector (unit norm), W[i] = |eigenvalue|, descending --
// what cvSVD(MtM, D, Ut, 0, CV_SVD_MODIFY_A | CV_SVD_U_T) leaves in epnp.cpp.  `tmp` = 12 x 12 + 12 doubles of LDS.
__device__ inline void fast_jacobi12(double* G, double* W, double* tmp, int lane)
{
    const double eps2 = (DBL_EPSILON * 10) * (DBL_EPSILON * 10);
    const int g = lane >> 3, sub = lane & 7;
    const bool act = g < 6;
    const bool two = sub < 4;                   // this lane also owns element sub + 8
#pragma unroll 1
    for (int sweep = 0; sweep < 30; sweep++) {
        bool any_rot = false;
#pragma unroll 1
        for (int r = 0; r < 11; r++) {
            // round-robin tournament: player 0 stays, the others rotate; pair g = (seat g, seat 11 - g)
            int i = g == 0 ? 0 : 1 + (g - 1 + r) % 11;
            int j = 1 + (10 - g + r) % 11;
            if (!act) { i = 0; j = 1; }
            if (i > j) { const int t = i; i = j; j = t; }          // the rotation moves the longer row to the lower index, as the cyclic order does
            double* ri = G + i * 12; double* rj = G + j * 12;
            const double x0 = ri[sub], y0 = rj[sub];
            const double x1 = two ? ri[sub + 8] : 0.0, y1 = two ? rj[sub + 8] : 0.0;
            const double p = octet_sum(x0*y0 + x1*y1), a = octet_sum(x0*x0 + x1*x1), b = octet_sum(y0*y0 + y1*y1);
            const bool rot = act && p * p > eps2 * a * b;
            __builtin_amdgcn_wave_barrier();
            if (rot) {
                double c, s;
                fast_rotation(a, b, p, c, s);
                ri[sub] = c*x0 + s*y0; rj[sub] = c*y0 - s*x0;
                if (two) { ri[sub + 8] = c*x1 + s*y1; rj[sub + 8] = c*y1 - s*x1; }
            }
            any_rot = any_rot || rot;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (!__any(any_rot)) break;
    }
    // singular values (row norms), rank in descending order, normalised rows written in that order
    if (lane < 12) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) { const double t = G[lane*12 + k]; sd += t*t; }
        tmp[144 + lane] = sqrt(sd);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < 12) {
        const double me = tmp[144 + lane];
        int rank = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) { const double o = tmp[144 + k]; rank += (o > me || (o == me && k < lane)) ? 1 : 0; }
        const double inv = me > DBL_MIN ? 1.0 / me : 0.0;
#pragma unroll
        for (int k = 0; k < 12; k++) tmp[rank*12 + k] = G[lane*12 + k] * inv;
        W[rank] = me;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < 144; e += 64) G[e] = tmp[e];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// least squares min |L x - rho| for the 6 x K system (K = 3, 4, 5 at run time) by Householder QR in registers; columns
// K..4 of `a` must be zero.  Returns false when R is (numerically) rank-deficient: the caller falls back to the SVD.
__device__ inline bool qr_lstsq6(double (&a)[6][5], double (&b)[6], int K, double (&x)[5])
{
    double diag[5], dmax = 0, dmin = DBL_MAX;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        diag[k] = 0;
        if (k < K) {
            double nrm2 = 0;
#pragma unroll
            for (int i = k; i < 6; i++) nrm2 += a[i][k] * a[i][k];
            const double nrm = sqrt(nrm2);
            const double alpha = a[k][k] > 0 ? -nrm : nrm;
            diag[k] = alpha;
            dmax = fmax(dmax, fabs(alpha)); dmin = fmin(dmin, fabs(alpha));
            // v = a[k:,k] - alpha e_k, H = I - 2 v v^T / (v^T v)
            const double v0 = a[k][k] - alpha;
            const double vtv = nrm2 - a[k][k]*a[k][k] + v0*v0;
            if (vtv > 0) {
                const double beta = 2.0 / vtv;
#pragma unroll
                for (int j = k + 1; j < 5; j++) {
                    double d = v0 * a[k][j];
#pragma unroll
                    for (int i = k + 1; i < 6; i++) d += a[i][k] * a[i][j];
                    d *= beta;
                    a[k][j] -= d * v0;
#pragma unroll
                    for (int i = k + 1; i < 6; i++) a[i][j] -= d * a[i][k];
                }
                double d = v0 * b[k];
#pragma unroll
                for (int i = k + 1; i < 6; i++) d += a[i][k] * b[i];
                d *= beta;
                b[k] -= d * v0;
#pragma unroll
                for (int i = k + 1; i < 6; i++) b[i] -= d * a[i][k];
            }
        }
    }
    if (!(dmin > 1e-10 * dmax)) return false;
#pragma unroll
    for (int k = 4; k >= 0; k--) {
        x[k] = 0;
        if (k < K) {
            double sacc = b[k];
#pragma unroll
            for (int j = k + 1; j < 5; j++) if (j < K) sacc -= a[k][j] * x[j];
            x[k] = sacc / diag[k];
        }
    }
    return true;
}

// cvRodrigues2, matrix -> vector, for a matrix that is orthonormal to working precision already (the re-orthogonalisation
// through an SVD, which cvRodrigues2 starts with, changes it by rounding errors only)
__device__ inline void rodrigues_mat2vec_orthonormal(const double* R, double* rv)
{
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx*rx + ry*ry + rz*rz)*0.25);
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

// The refit.  One workgroup of kFastThreads threads; lds = kFastLdsDoubles doubles of LDS.
// ws (global): pws 3c | us 2c | alphas 4c | pcs 9c (c = cap), as the sequential refit.
static const int kFastThreads = 256, kFastWaves = kFastThreads / 64;       // 4 waves, one per SIMD: a workgroup that fits beside the detection kernels
static const int kFastTile = 64 * 6;                                       // per-wave tile: 64 points x (a0..a3, uc - u, vc - v)
static const int kFastUnion = kFastWaves * kFastTile > kFastWaves * 256 ? kFastWaves * kFastTile : kFastWaves * 256;
static const int kFastLdsDoubles = EPNP_SMALL + 160 + kFastWaves * 40 + 40 + kFastUnion;

struct EpnpFast {
    double uc, vc, fu, fv;
    int n, cap;
    double* ws;
    double* lds;
    long long* clk;

    __device__ __forceinline__ void stamp(int i) const { if (clk && threadIdx.x == 0) clk[i] = wall_clock64(); }

    // K sums over the workgroup: v[k] summed over all threads -> out[k] (LDS), valid after the call on every thread
    template <int K>
    __device__ __forceinline__ void block_sum(double (&v)[K], double* red, double* out) const
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < K; k++) { const double t = wave_sum(v[k]); if (lane == 0) red[k * kFastWaves + wave] = t; }
        __syncthreads();
        if (threadIdx.x < K) {
            double acc = 0;
#pragma unroll
            for (int w2 = 0; w2 < kFastWaves; w2++) acc += red[threadIdx.x * kFastWaves + w2];
            out[threadIdx.x] = acc;
        }
        __syncthreads();
    }

    __device__ void compute_pose(double* rvec, double* tvec)
    {
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        double* small = lds;                               // EPNP_SMALL: the fixed-size state, laid out as Epnp<>::s
        double* jtmp = lds + EPNP_SMALL;                   // 156: fast_jacobi12 scratch
        double* red = jtmp + 160;                          // kFastWaves * 40
        double* sums = red + kFastWaves * 40;              // 40
        double* uni = sums + 40;                           // per-wave tiles, then the waves' 16 x 16 partial products
        double* pws = ws; double* us = ws + 3 * (size_t)cap; double* alphas = ws + 5 * (size_t)cap; double* pcs = ws + 9 * (size_t)cap;
        using WA = SArr<1>;
        Epnp<WavePolicy> ew;                               // the fixed-size stages shared with the exact solver
        ew.uc = uc; ew.vc = vc; ew.fu = fu; ew.fv = fv; ew.n = n; ew.s = WA{small};
        double* cws = small + EP_CWS; double* mtm = small + EP_MTM; double* L = small + EP_L; double* rho = small + EP_RHO;
        double* pw0 = small + EP_PW0; double* sc = small + EP_SC;
        const double inv_n = 1.0 / n;
        stamp(0);
        // ---- choose_control_points: centroid, covariance, PCA ----
        {
            double v[3] = {0, 0, 0};
            for (int i = tid; i < n; i += kFastThreads) { v[0] += pws[3*i]; v[1] += pws[3*i + 1]; v[2] += pws[3*i + 2]; }
            block_sum<3>(v, red, sums);
            const double c0 = sums[0] * inv_n, c1 = sums[1] * inv_n, c2 = sums[2] * inv_n;
            __syncthreads();                               // sums is reused
            double q[6] = {0, 0, 0, 0, 0, 0};
            for (int i = tid; i < n; i += kFastThreads) {
                const double dx = pws[3*i] - c0, dy = pws[3*i + 1] - c1, dz = pws[3*i + 2] - c2;
                q[0] += dx*dx; q[1] += dx*dy; q[2] += dx*dz; q[3] += dy*dy; q[4] += dy*dz; q[5] += dz*dz;
            }
            block_sum<6>(q, red, sums);
            if (tid == 0) {
                const double ptp[9] = { sums[0], sums[1], sums[2], sums[1], sums[3], sums[4], sums[2], sums[4], sums[5] };
                double w[3], u[3][3], vv[3][3];
                fast_svd3(ptp, w, u, vv);
                cws[0] = c0; cws[1] = c1; cws[2] = c2;
                // CC = [k_1 u_1 | k_2 u_2 | k_3 u_3] with orthonormal u: its inverse is diag(1/k) U^T (sc[9..17])
                for (int i = 1; i < 4; i++) {
                    const double k = sqrt(w[i-1] * inv_n);
                    const double ik = k > 0 ? 1.0 / k : 0.0;
                    for (int j = 0; j < 3; j++) { cws[3*i + j] = cws[j] + k * u[i-1][j]; sc[9 + 3*(i-1) + j] = u[i-1][j] * ik; }
                }
            }
            __syncthreads();
        }
        stamp(1); stamp(2);
        // ---- barycentric coordinates + M^T M on the fp64 matrix pipe ----
        // wave w takes points [64 (w + 16 it), +64): each lane computes its point's alphas (kept for compute_pcs), the wave
        // shares them through its LDS tile, then every 4 rows of M (2 points) are one v_mfma_f64_16x16x4_f64 with A = B:
        // lane l supplies M[row 4c + (l >> 4)][col l & 15].
        {
            const double c0 = cws[0], c1 = cws[1], c2 = cws[2];
            double ci[9];
#pragma unroll
            for (int k = 0; k < 9; k++) ci[k] = sc[9 + k];
            double* tile = uni + wave * kFastTile;
            f64x4 acc = {0, 0, 0, 0};
            const int col = lane & 15, kr = lane >> 4;      // column of M, row inside the slab
            const int pa = col / 3, pq = col - 3 * pa;      // control point and coordinate of the column
            // element (row 2p + r, col 3a + q) of M = alpha_a(p) * f, f = (fu, 0, uc - u_p) for r = 0, (0, fv, vc - v_p) for r = 1:
            // per lane a constant factor cf, or (wt = 1) the point's tile entry 4 + r -- no branches inside the MFMA loop
            const int r = kr & 1;
            const bool live = col < 12;
            const double cf = !live ? 0.0 : (r == 0 ? (pq == 0 ? fu : 0.0) : (pq == 1 ? fv : 0.0));
            const double wt = live && pq == 2 ? 1.0 : 0.0;
            const int off_as = (kr >> 1) * 6 + (live ? pa : 0), off_t = (kr >> 1) * 6 + 4 + r;
            for (int base = wave * 64; base < n; base += kFastWaves * 64) {
                const int i = base + lane;
                double* t = tile + lane * 6;
                if (i < n) {
                    const double dx = pws[3*i] - c0, dy = pws[3*i + 1] - c1, dz = pws[3*i + 2] - c2;
                    const double a1 = ci[0]*dx + ci[1]*dy + ci[2]*dz, a2 = ci[3]*dx + ci[4]*dy + ci[5]*dz, a3 = ci[6]*dx + ci[7]*dy + ci[8]*dz;
                    const double a0 = 1.0 - a1 - a2 - a3;
                    double* al = alphas + 4 * (size_t)i;
                    al[0] = a0; al[1] = a1; al[2] = a2; al[3] = a3;
                    t[0] = a0; t[1] = a1; t[2] = a2; t[3] = a3; t[4] = uc - us[2*i]; t[5] = vc - us[2*i + 1];
                } else {
                    t[0] = t[1] = t[2] = t[3] = t[4] = t[5] = 0.0;     // rows of zeros add nothing
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll 8
                for (int c = 0; c < 32; c++) {
                    const double* tp = tile + c * 12;                  // points 2c, 2c + 1
                    const double m = tp[off_as] * __builtin_fma(wt, tp[off_t], cf);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m, m, acc, 0, 0, 0);
                }
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();                               // every wave is done with its tile: the region now takes the partial products
            // D[row][col]: col = lane & 15, row = (lane >> 4) + 4 reg
#pragma unroll
            for (int reg = 0; reg < 4; reg++) uni[wave * 256 + ((lane >> 4) + 4 * reg) * 16 + (lane & 15)] = acc[reg];
            __syncthreads();
            if (tid < 144) {
                const int i = tid / 12, j = tid - 12 * i;
                const int lo = i < j ? i : j, hi = i < j ? j : i;   // the upper triangle, mirrored: exactly symmetric
                double a2 = 0;
#pragma unroll
                for (int w2 = 0; w2 < kFastWaves; w2++) a2 += uni[w2 * 256 + lo * 16 + hi];
                mtm[tid] = a2;
            }
            __syncthreads();
        }
        stamp(3);
        // ---- eigenvectors of M^T M, L_6x10, rho, the three beta approximations (wave 0) ----
        if (wave == 0) {
            fast_jacobi12(mtm, small + EP_D, jtmp, lane);
            if (clk && tid == 0) clk[4] = wall_clock64();
            ew.compute_L_6x10(WA{mtm}, WA{L});
            if (lane == 0) {
                WA c{cws};
                rho[0] = ew.dist2(c, c + 3); rho[1] = ew.dist2(c, c + 6); rho[2] = ew.dist2(c, c + 9);
                rho[3] = ew.dist2(c + 3, c + 6); rho[4] = ew.dist2(c + 3, c + 9); rho[5] = ew.dist2(c + 6, c + 9);
            }
            WavePolicy::sync();
            if (clk && tid == 0) clk[10] = wall_clock64();
            if (lane < 3) {
                WA B = ew.br(lane), betas = B + EPB_BETAS, ccs = B + EPB_CCS;
                const int which = lane + 1, K = which == 1 ? 4 : which == 2 ? 3 : 5;
                double a[6][5], bb[6], x[5];
#pragma unroll
                for (int i = 0; i < 6; i++) {
#pragma unroll
                    for (int j = 0; j < 5; j++) {
                        const int colL = which == 1 ? (j == 0 ? 0 : j == 1 ? 1 : j == 2 ? 3 : 6) : j;
                        a[i][j] = j < K ? L[10*i + colL] : 0.0;
                    }
                    bb[i] = rho[i];
                }
                if (qr_lstsq6(a, bb, K, x)) {
                    // the sign / ratio conventions of find_betas_approx_{1,2,3}
                    if (which == 1) {
                        if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = -x[1] / betas[0]; betas[2] = -x[2] / betas[0]; betas[3] = -x[3] / betas[0]; }
                        else          { betas[0] = sqrt(x[0]);  betas[1] = x[1] / betas[0];  betas[2] = x[2] / betas[0];  betas[3] = x[3] / betas[0]; }
                    } else {
                        if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = (x[2] < 0) ? sqrt(-x[2]) : 0.0; }
                        else          { betas[0] = sqrt(x[0]);  betas[1] = (x[2] > 0) ? sqrt(x[2]) : 0.0; }
                        if (x[1] < 0) betas[0] = -betas[0];
                        betas[2] = which == 2 ? 0.0 : x[3] / betas[0];
                        betas[3] = 0.0;
                    }
                } else ew.find_betas(which, WA{L}, WA{rho}, betas, B + EPB_SC);
                if (clk && tid == 0) clk[11] = wall_clock64();
                ew.gauss_newton(WA{L}, WA{rho}, betas, B + EPB_SC);
                if (clk && tid == 0) clk[12] = wall_clock64();
                for (int i = 0; i < 12; i++) ccs[i] = 0.0;
                for (int i = 0; i < 4; i++) {
                    const double* v = mtm + 12*(11 - i);
                    const double bi = betas[i];
                    for (int j = 0; j < 12; j++) ccs[j] += bi * v[j];
                }
            }
        }
        __syncthreads();
        stamp(5);
        // ---- compute_pcs + solve_for_sign for the three branches, centroids ----
        double ccs[3][12];
#pragma unroll
        for (int b = 0; b < 3; b++) {
            const double* cb = small + EP_BR + b * EPB_SIZE + EPB_CCS;
            // sign of the first point's depth (solve_for_sign)
            const double z0 = alphas[0]*cb[2] + alphas[1]*cb[5] + alphas[2]*cb[8] + alphas[3]*cb[11];
            const double sg = z0 < 0.0 ? -1.0 : 1.0;
#pragma unroll
            for (int k = 0; k < 12; k++) ccs[b][k] = sg * cb[k];
        }
        {
            double v[12];
#pragma unroll
            for (int k = 0; k < 12; k++) v[k] = 0;
            for (int i = tid; i < n; i += kFastThreads) {
                const double a0 = alphas[4*i], a1 = alphas[4*i + 1], a2 = alphas[4*i + 2], a3 = alphas[4*i + 3];
                v[0] += pws[3*i]; v[1] += pws[3*i + 1]; v[2] += pws[3*i + 2];
#pragma unroll
                for (int b = 0; b < 3; b++)
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        const double pc = a0*ccs[b][j] + a1*ccs[b][3 + j] + a2*ccs[b][6 + j] + a3*ccs[b][9 + j];
                        pcs[3*((size_t)b*n + i) + j] = pc;
                        v[3 + 3*b + j] += pc;
                    }
            }
            block_sum<12>(v, red, sums);
        }
        stamp(6);
        // ---- estimate_R_and_t: 3 x 3 covariances, absolute orientation ----
        {
            double m0[12];
#pragma unroll
            for (int k = 0; k < 12; k++) m0[k] = sums[k] * inv_n;       // pw0 | pc0 of the three branches
            __syncthreads();
            double v[27];
#pragma unroll
            for (int k = 0; k < 27; k++) v[k] = 0;
            for (int i = tid; i < n; i += kFastThreads) {
                const double wx = pws[3*i] - m0[0], wy = pws[3*i + 1] - m0[1], wz = pws[3*i + 2] - m0[2];
#pragma unroll
                for (int b = 0; b < 3; b++)
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        const double d = pcs[3*((size_t)b*n + i) + j] - m0[3 + 3*b + j];
                        v[9*b + 3*j] += d * wx; v[9*b + 3*j + 1] += d * wy; v[9*b + 3*j + 2] += d * wz;
                    }
            }
            block_sum<27>(v, red, sums);
            stamp(7);
            if (tid < 3) {
                double* B = small + EP_BR + tid * EPB_SIZE;
                double* R = B + EPB_RS; double* t = B + EPB_TS;
                double w[3], u[3][3], vv[3][3];
                fast_svd3(sums + 9 * tid, w, u, vv);
                for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3*i + j] = u[0][i]*vv[0][j] + u[1][i]*vv[1][j] + u[2][i]*vv[2][j];
                const double det = R[0]*R[4]*R[8] + R[1]*R[5]*R[6] + R[2]*R[3]*R[7] - R[2]*R[4]*R[6] - R[1]*R[3]*R[8] - R[0]*R[5]*R[7];
                if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
                for (int i = 0; i < 3; i++) t[i] = m0[3 + 3*tid + i] - (R[3*i]*m0[0] + R[3*i + 1]*m0[1] + R[3*i + 2]*m0[2]);
            }
            __syncthreads();
        }
        stamp(8);
        // ---- reprojection error of the three branches, the best one's pose ----
        {
            double Rt[3][12];
#pragma unroll
            for (int b = 0; b < 3; b++) {
                const double* B = small + EP_BR + b * EPB_SIZE;
#pragma unroll
                for (int k = 0; k < 9; k++) Rt[b][k] = B[EPB_RS + k];
#pragma unroll
                for (int k = 0; k < 3; k++) Rt[b][9 + k] = B[EPB_TS + k];
            }
            double v[3] = {0, 0, 0};
            for (int i = tid; i < n; i += kFastThreads) {
                const double X = pws[3*i], Y = pws[3*i + 1], Z = pws[3*i + 2], u = us[2*i], vv = us[2*i + 1];
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    const double Xc = Rt[b][0]*X + Rt[b][1]*Y + Rt[b][2]*Z + Rt[b][9];
                    const double Yc = Rt[b][3]*X + Rt[b][4]*Y + Rt[b][5]*Z + Rt[b][10];
                    const double iz = 1.0 / (Rt[b][6]*X + Rt[b][7]*Y + Rt[b][8]*Z + Rt[b][11]);
                    const double du = u - (uc + fu * Xc * iz), dv = vv - (vc + fv * Yc * iz);
                    v[b] += sqrt(du*du + dv*dv);
                }
            }
            block_sum<3>(v, red, sums);
            if (tid == 0) {
                for (int b = 0; b < 3; b++) small[EP_BR + b * EPB_SIZE + EPB_REP] = sums[b] * inv_n;
                int N = 0; double best = sums[0];
                if (sums[1] < best) { N = 1; best = sums[1]; }
                if (sums[2] < best) N = 2;
                double Rs[12];
#pragma unroll
                for (int k = 0; k < 12; k++) Rs[k] = N == 0 ? Rt[0][k] : (N == 1 ? Rt[1][k] : Rt[2][k]);
                tvec[0] = Rs[9]; tvec[1] = Rs[10]; tvec[2] = Rs[11];
                rodrigues_mat2vec_orthonormal(Rs, rvec);
            }
        }
        stamp(9);
    }
};

}  // namespace uvo
