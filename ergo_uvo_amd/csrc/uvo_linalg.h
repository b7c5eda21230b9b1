// uvo_linalg.h -- small dense fp64 linear algebra for the pose kernels (device side).
// One-sided Jacobi SVD, SVD back-substitution, least-squares solve and 3x3 inverse in the
// operation order of OpenCV 4.5's lapack.cpp (JacobiSVDImpl_, SVBkSbImpl_, solve/invert with
// DECOMP_SVD), which is what cv::triangulatePoints, cv::Rodrigues and the EPnP solver reach.
// Arrays go through a strided accessor so one source serves both layouts:
//   SArr<S>: element i lives at p[i*S].  S = threads-per-block for the "one problem per thread"
//   kernels (LDS, [element][lane] interleave -> conflict-free), S = 1 for block-cooperative code.
#pragma once
#include "uvo_math.h"

namespace uvo {

template <int S>
struct SArr {
    double* p;
    __host__ __device__ __forceinline__ double& operator[](int i) const { return p[i * S]; }
    __host__ __device__ __forceinline__ SArr operator+(int o) const { return SArr{p + o * S}; }
};

// JacobiSVDImpl_<double>(At, astep, W, Vt, vstep, m, n, n1, DBL_MIN, DBL_EPSILON*10).
// At: n rows of m (row i at At[i*astep..]); Vt (n x n) optional; Wt: n doubles of scratch.
template <class A>
__host__ __device__ void jacobi_svd(A At, int astep, A W_out, A Vt, int vstep, bool hasV, int m, int n, int n1, A W)
{
    const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
    int i, j, k, iter, max_iter = m > 30 ? m : 30;
    double c, s, sd;
    if (!hasV) n1 = 0;

    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) { double t = At[i*astep + k]; sd += t*t; }
        W[i] = sd;
        if (hasV) { for (k = 0; k < n; k++) Vt[i*vstep + k] = 0; Vt[i*vstep + i] = 1; }
    }
#pragma unroll 1
    for (iter = 0; iter < max_iter; iter++) {
        bool changed = false;
#pragma unroll 1
        for (i = 0; i < n-1; i++)
#pragma unroll 1
            for (j = i+1; j < n; j++) {
                A Ai = At + i*astep, Aj = At + j*astep;
                double a = W[i], p = 0, b = W[j];
                for (k = 0; k < m; k++) p += Ai[k]*Aj[k];
                if (fabs(p) <= eps*sqrt(a*b)) continue;
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
                for (k = 0; k < m; k++) {
                    double x = Ai[k], y = Aj[k];
                    double t0 = c*x + s*y;
                    double t1 = -s*x + c*y;
                    Ai[k] = t0; Aj[k] = t1;
                    a += t0*t0; b += t1*t1;
                }
                W[i] = a; W[j] = b;
                changed = true;
                if (hasV) {
                    A Vi = Vt + i*vstep, Vj = Vt + j*vstep;
                    for (k = 0; k < n; k++) {
                        double x = Vi[k], y = Vj[k];
                        double t0 = c*x + s*y;
                        double t1 = -s*x + c*y;
                        Vi[k] = t0; Vj[k] = t1;
                    }
                }
            }
        if (!changed) break;
    }
    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) { double t = At[i*astep + k]; sd += t*t; }
        W[i] = sqrt(sd);
    }
    for (i = 0; i < n-1; i++) {
        j = i;
        for (k = i+1; k < n; k++) if (W[j] < W[k]) j = k;
        if (i != j) {
            double t = W[i]; W[i] = W[j]; W[j] = t;
            if (hasV) {
                for (k = 0; k < m; k++) { t = At[i*astep+k]; At[i*astep+k] = At[j*astep+k]; At[j*astep+k] = t; }
                for (k = 0; k < n; k++) { t = Vt[i*vstep+k]; Vt[i*vstep+k] = Vt[j*vstep+k]; Vt[j*vstep+k] = t; }
            }
        }
    }
    for (i = 0; i < n; i++) W_out[i] = W[i];
    if (!hasV) return;

    uint64_t rng = 0x12345678ULL;
    for (i = 0; i < n1; i++) {
        sd = i < n ? W[i] : 0;
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const double val0 = 1./m;
            for (k = 0; k < m; k++) {
                double val = (rng_next(rng) & 256) != 0 ? val0 : -val0;
                At[i*astep + k] = val;
            }
            for (iter = 0; iter < 2; iter++) {
                for (j = 0; j < i; j++) {
                    sd = 0;
                    for (k = 0; k < m; k++) sd += At[i*astep + k]*At[j*astep + k];
                    double asum = 0;
                    for (k = 0; k < m; k++) {
                        double t = At[i*astep + k] - sd*At[j*astep + k];
                        At[i*astep + k] = t;
                        asum += fabs(t);
                    }
                    asum = asum > eps*100 ? 1/asum : 0;
                    for (k = 0; k < m; k++) At[i*astep + k] *= asum;
                }
            }
            sd = 0;
            for (k = 0; k < m; k++) { double t = At[i*astep + k]; sd += t*t; }
            sd = sqrt(sd);
        }
        s = sd > minval ? 1/sd : 0.;
        for (k = 0; k < m; k++) At[i*astep + k] *= s;
    }
}

// cv::SVD::compute of a square n x n matrix A (row-major, tight, element (i,j) at A[i*n+j]):
// writes At = A^T in place of `At`, runs Jacobi; afterwards row i of At is the i-th LEFT
// singular vector (U^T), row i of Vt the i-th right singular vector.  W, Wt: n doubles each.
template <class A>
__host__ __device__ void svd_square(A Ain, A At, A W, A Vt, A Wt, int n)
{
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) At[j*n + i] = Ain[i*n + j];
    jacobi_svd(At, n, W, Vt, n, true, n, n, n, Wt);
}

// SVBkSbImpl_<double>, nb == 1 with right-hand side b: x = V diag(1/w) U^T b.
// ut: row i = i-th left vector (length m), vt: row i = i-th right vector (length n).
template <class A>
__host__ __device__ void svbksb_vec(int m, int n, A w, A ut, int ldu, A vt, int ldv, A b, A x)
{
    double threshold = 0; int i, j, nm = m < n ? m : n;
    for (i = 0; i < n; i++) x[i] = 0;
    for (i = 0; i < nm; i++) threshold += w[i];
    threshold *= DBL_EPSILON * 2;
    for (i = 0; i < nm; i++) {
        double wi = w[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1/wi;
        double s = 0;
        for (j = 0; j < m; j++) s += ut[i*ldu + j]*b[j];
        s *= wi;
        for (j = 0; j < n; j++) x[j] = x[j] + s*vt[i*ldv + j];
    }
}

// cv::solve(A[m x n], b, DECOMP_SVD).  scratch: a (n*m), v (n*n), w (n), wt (n)
template <class A>
__host__ __device__ void solve_svd(A Amat, int m, int n, A b, A x, A a, A v, A w, A wt)
{
    for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) a[j*m + i] = Amat[i*n + j];
    jacobi_svd(a, m, w, v, n, true, m, n, n, wt);
    svbksb_vec(m, n, w, a, m, v, n, b, x);
}

// cv::invert(A[3x3], DECOMP_SVD) = SVD::compute + SVD::backSubst(rhs = empty).
// scratch: a(9) v(9) w(3) wt(3)
template <class A>
__host__ __device__ void invert3_svd(A Amat, A Ainv, A a, A v, A w, A wt)
{
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a[j*3 + i] = Amat[i*3 + j];
    jacobi_svd(a, 3, w, v, 3, true, 3, 3, 3, wt);
    double threshold = 0;
    for (int i = 0; i < 9; i++) Ainv[i] = 0;
    for (int i = 0; i < 3; i++) threshold += w[i];
    threshold *= DBL_EPSILON * 2;
    for (int i = 0; i < 3; i++) {
        double wi = w[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1/wi;
        double b0 = a[i*3 + 0]*wi, b1 = a[i*3 + 1]*wi, b2 = a[i*3 + 2]*wi;
        for (int k = 0; k < 3; k++) {
            double sv = v[i*3 + k];
            Ainv[k*3 + 0] = Ainv[k*3 + 0] + sv*b0;
            Ainv[k*3 + 1] = Ainv[k*3 + 1] + sv*b1;
            Ainv[k*3 + 2] = Ainv[k*3 + 2] + sv*b2;
        }
    }
}

}  // namespace uvo
