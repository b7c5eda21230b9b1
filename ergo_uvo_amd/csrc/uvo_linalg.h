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

// JacobiSVDImpl_<double>(At, astep = M, W, Vt, vstep = N, m = M, n = N, n1 = N, DBL_MIN, DBL_EPSILON*10)
// with Vt present (OpenCV always passes it on the paths used here).  At: N rows of M (row i =
// column i of A); afterwards row i of At = i-th left singular vector, row i of Vt = i-th right
// singular vector.  M, N are compile-time so the two rows of a rotation live in registers and the
// k-loops unroll (same operations, same order as the reference loop).  ACCUM_V = false skips the
// V rotations (they never feed back into At or W) for callers that only consume U and W.
template <int M, int N, bool ACCUM_V, class A>
__host__ __device__ void jacobi_svd(A At, A W_out, A Vt, A W)
{
    const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
    int i, j, k, iter;
    const int max_iter = M > 30 ? M : 30;
    double c, s, sd;

#pragma unroll 1
    for (i = 0; i < N; i++) {
        sd = 0;
#pragma unroll
        for (k = 0; k < M; k++) { double t = At[i*M + k]; sd += t*t; }
        W[i] = sd;
        if (ACCUM_V) {
#pragma unroll
            for (k = 0; k < N; k++) Vt[i*N + k] = 0;
            Vt[i*N + i] = 1;
        }
    }
#pragma unroll 1
    for (iter = 0; iter < max_iter; iter++) {
        bool changed = false;
#pragma unroll 1
        for (i = 0; i < N-1; i++)
#pragma unroll 1
            for (j = i+1; j < N; j++) {
                A Ai = At + i*M, Aj = At + j*M;
                double ai[M], aj[M];
#pragma unroll
                for (k = 0; k < M; k++) { ai[k] = Ai[k]; aj[k] = Aj[k]; }
                double a = W[i], p = 0, b = W[j];
#pragma unroll
                for (k = 0; k < M; k++) p += ai[k]*aj[k];
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
#pragma unroll
                for (k = 0; k < M; k++) {
                    double t0 = c*ai[k] + s*aj[k];
                    double t1 = -s*ai[k] + c*aj[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += t0*t0; b += t1*t1;
                }
                W[i] = a; W[j] = b;
                changed = true;
                if (ACCUM_V) {
                    A Vi = Vt + i*N, Vj = Vt + j*N;
#pragma unroll
                    for (k = 0; k < N; k++) {
                        double x = Vi[k], y = Vj[k];
                        double t0 = c*x + s*y;
                        double t1 = -s*x + c*y;
                        Vi[k] = t0; Vj[k] = t1;
                    }
                }
            }
        if (!changed) break;
    }
#pragma unroll 1
    for (i = 0; i < N; i++) {
        sd = 0;
#pragma unroll
        for (k = 0; k < M; k++) { double t = At[i*M + k]; sd += t*t; }
        W[i] = sqrt(sd);
    }
#pragma unroll 1
    for (i = 0; i < N-1; i++) {
        j = i;
        for (k = i+1; k < N; k++) if (W[j] < W[k]) j = k;
        if (i != j) {
            double t = W[i]; W[i] = W[j]; W[j] = t;
            for (k = 0; k < M; k++) { t = At[i*M+k]; At[i*M+k] = At[j*M+k]; At[j*M+k] = t; }
            if (ACCUM_V) for (k = 0; k < N; k++) { t = Vt[i*N+k]; Vt[i*N+k] = Vt[j*N+k]; Vt[j*N+k] = t; }
        }
    }
    for (i = 0; i < N; i++) W_out[i] = W[i];

    uint64_t rng = 0x12345678ULL;
#pragma unroll 1
    for (i = 0; i < N; i++) {
        sd = W[i];
#pragma unroll 1
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            // zero singular value: random +-1/m vector orthogonalised against the previous rows
            const double val0 = 1./M;
            for (k = 0; k < M; k++) {
                double val = (rng_next(rng) & 256) != 0 ? val0 : -val0;
                At[i*M + k] = val;
            }
            for (iter = 0; iter < 2; iter++) {
                for (j = 0; j < i; j++) {
                    sd = 0;
                    for (k = 0; k < M; k++) sd += At[i*M + k]*At[j*M + k];
                    double asum = 0;
                    for (k = 0; k < M; k++) {
                        double t = At[i*M + k] - sd*At[j*M + k];
                        At[i*M + k] = t;
                        asum += fabs(t);
                    }
                    asum = asum > eps*100 ? 1/asum : 0;
                    for (k = 0; k < M; k++) At[i*M + k] *= asum;
                }
            }
            sd = 0;
            for (k = 0; k < M; k++) { double t = At[i*M + k]; sd += t*t; }
            sd = sqrt(sd);
        }
        s = sd > minval ? 1/sd : 0.;
#pragma unroll
        for (k = 0; k < M; k++) At[i*M + k] *= s;
    }
}

// cv::SVD::compute of a square N x N matrix A (row-major, tight): writes At = A^T into `At`, runs
// Jacobi; afterwards row i of At is the i-th LEFT singular vector (U^T), row i of Vt the i-th right
// singular vector.  W, Wt: N doubles each.
template <int N, class A>
__host__ __device__ void svd_square(A Ain, A At, A W, A Vt, A Wt)
{
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) At[j*N + i] = Ain[i*N + j];
    jacobi_svd<N, N, true>(At, W, Vt, Wt);
}

// ------------------------------------------------------------------------------------------
// Runtime-sized one-sided Jacobi (same operations as jacobi_svd<M,N,true>); used where several
// lanes run differently-shaped solves at once (the three beta approximations) and for the 5x9 system of
// the five-point solver, whose FULL_UV decomposition completes n1 > n rows (the null-space basis).
// ------------------------------------------------------------------------------------------
template <class A>
__host__ __device__ void jacobi_svd_rt(A At, A W_out, A Vt, A W, int m, int n, int n1 = -1)
{
    if (n1 < 0) n1 = n;
    const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
    int i, j, k, iter, max_iter = m > 30 ? m : 30;
    double c, s, sd;
    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) { double t = At[i*m + k]; sd += t*t; }
        W[i] = sd;
        for (k = 0; k < n; k++) Vt[i*n + k] = 0;
        Vt[i*n + i] = 1;
    }
#pragma unroll 1
    for (iter = 0; iter < max_iter; iter++) {
        bool changed = false;
#pragma unroll 1
        for (i = 0; i < n-1; i++)
#pragma unroll 1
            for (j = i+1; j < n; j++) {
                A Ai = At + i*m, Aj = At + j*m;
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
                A Vi = Vt + i*n, Vj = Vt + j*n;
                for (k = 0; k < n; k++) {
                    double x = Vi[k], y = Vj[k];
                    double t0 = c*x + s*y;
                    double t1 = -s*x + c*y;
                    Vi[k] = t0; Vj[k] = t1;
                }
            }
        if (!changed) break;
    }
    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) { double t = At[i*m + k]; sd += t*t; }
        W[i] = sqrt(sd);
    }
    for (i = 0; i < n-1; i++) {
        j = i;
        for (k = i+1; k < n; k++) if (W[j] < W[k]) j = k;
        if (i != j) {
            double t = W[i]; W[i] = W[j]; W[j] = t;
            for (k = 0; k < m; k++) { t = At[i*m+k]; At[i*m+k] = At[j*m+k]; At[j*m+k] = t; }
            for (k = 0; k < n; k++) { t = Vt[i*n+k]; Vt[i*n+k] = Vt[j*n+k]; Vt[j*n+k] = t; }
        }
    }
    for (i = 0; i < n; i++) W_out[i] = W[i];
    uint64_t rng = 0x12345678ULL;
    for (i = 0; i < n1; i++) {
        sd = i < n ? W[i] : 0;
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const double val0 = 1./m;
            for (k = 0; k < m; k++) { double val = (rng_next(rng) & 256) != 0 ? val0 : -val0; At[i*m + k] = val; }
            for (iter = 0; iter < 2; iter++)
                for (j = 0; j < i; j++) {
                    sd = 0;
                    for (k = 0; k < m; k++) sd += At[i*m + k]*At[j*m + k];
                    double asum = 0;
                    for (k = 0; k < m; k++) { double t = At[i*m + k] - sd*At[j*m + k]; At[i*m + k] = t; asum += fabs(t); }
                    asum = asum > eps*100 ? 1/asum : 0;
                    for (k = 0; k < m; k++) At[i*m + k] *= asum;
                }
            sd = 0;
            for (k = 0; k < m; k++) { double t = At[i*m + k]; sd += t*t; }
            sd = sqrt(sd);
        }
        s = sd > minval ? 1/sd : 0.;
        for (k = 0; k < m; k++) At[i*m + k] *= s;
    }
}

// jacobi_svd_rt for m == 6 rows and n <= 5 columns decided at run time (the three beta approximations of EPnP solve
// 6 x 4, 6 x 3 and 6 x 5 systems side by side on three lanes): the same operations in the same order, but the loops
// over m are unrolled and the two rows of a rotation are fetched in one batch, so a rotation waits for memory once
// instead of once per element.
template <class A>
__host__ __device__ void jacobi_svd_rt6(A At, A W_out, A Vt, A W, int n)
{
    constexpr int m = 6, NMAX = 5;
    const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
    int i, j, k, iter;
    const int max_iter = 30;
    double c, s, sd;
    for (i = 0; i < n; i++) {
        double r[m];
#pragma unroll
        for (k = 0; k < m; k++) r[k] = At[i*m + k];
        sd = 0;
#pragma unroll
        for (k = 0; k < m; k++) sd += r[k]*r[k];
        W[i] = sd;
#pragma unroll
        for (k = 0; k < NMAX; k++) if (k < n) Vt[i*n + k] = 0;
        Vt[i*n + i] = 1;
    }
#pragma unroll 1
    for (iter = 0; iter < max_iter; iter++) {
        bool changed = false;
#pragma unroll 1
        for (i = 0; i < n-1; i++)
#pragma unroll 1
            for (j = i+1; j < n; j++) {
                A Ai = At + i*m, Aj = At + j*m;
                double x[m], y[m];
#pragma unroll
                for (k = 0; k < m; k++) { x[k] = Ai[k]; y[k] = Aj[k]; }
                double a = W[i], p = 0, b = W[j];
#pragma unroll
                for (k = 0; k < m; k++) p += x[k]*y[k];
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
#pragma unroll
                for (k = 0; k < m; k++) {
                    double t0 = c*x[k] + s*y[k];
                    double t1 = -s*x[k] + c*y[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += t0*t0; b += t1*t1;
                }
                W[i] = a; W[j] = b;
                changed = true;
                A Vi = Vt + i*n, Vj = Vt + j*n;
                double vx[NMAX], vy[NMAX];
#pragma unroll
                for (k = 0; k < NMAX; k++) if (k < n) { vx[k] = Vi[k]; vy[k] = Vj[k]; }
#pragma unroll
                for (k = 0; k < NMAX; k++) if (k < n) {
                    double t0 = c*vx[k] + s*vy[k];
                    double t1 = -s*vx[k] + c*vy[k];
                    Vi[k] = t0; Vj[k] = t1;
                }
            }
        if (!changed) break;
    }
    for (i = 0; i < n; i++) {
        double r[m];
#pragma unroll
        for (k = 0; k < m; k++) r[k] = At[i*m + k];
        sd = 0;
#pragma unroll
        for (k = 0; k < m; k++) sd += r[k]*r[k];
        W[i] = sqrt(sd);
    }
    for (i = 0; i < n-1; i++) {
        j = i;
        for (k = i+1; k < n; k++) if (W[j] < W[k]) j = k;
        if (i != j) {
            double t = W[i]; W[i] = W[j]; W[j] = t;
            for (k = 0; k < m; k++) { t = At[i*m+k]; At[i*m+k] = At[j*m+k]; At[j*m+k] = t; }
            for (k = 0; k < n; k++) { t = Vt[i*n+k]; Vt[i*n+k] = Vt[j*n+k]; Vt[j*n+k] = t; }
        }
    }
    for (i = 0; i < n; i++) W_out[i] = W[i];
    uint64_t rng = 0x12345678ULL;
    for (i = 0; i < n; i++) {                  // (rows with a vanishing singular value are rebuilt exactly as in jacobi_svd_rt)
        sd = W[i];
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const double val0 = 1./m;
            for (k = 0; k < m; k++) { double val = (rng_next(rng) & 256) != 0 ? val0 : -val0; At[i*m + k] = val; }
            for (iter = 0; iter < 2; iter++)
                for (j = 0; j < i; j++) {
                    sd = 0;
                    for (k = 0; k < m; k++) sd += At[i*m + k]*At[j*m + k];
                    double asum = 0;
                    for (k = 0; k < m; k++) { double t = At[i*m + k] - sd*At[j*m + k]; At[i*m + k] = t; asum += fabs(t); }
                    asum = asum > eps*100 ? 1/asum : 0;
                    for (k = 0; k < m; k++) At[i*m + k] *= asum;
                }
            sd = 0;
            for (k = 0; k < m; k++) { double t = At[i*m + k]; sd += t*t; }
            sd = sqrt(sd);
        }
        s = sd > minval ? 1/sd : 0.;
#pragma unroll
        for (k = 0; k < m; k++) At[i*m + k] *= s;
    }
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
template <int M, int N, class A>
__host__ __device__ void solve_svd(A Amat, A b, A x, A a, A v, A w, A wt)
{
    for (int i = 0; i < M; i++) for (int j = 0; j < N; j++) a[j*M + i] = Amat[i*N + j];
    jacobi_svd<M, N, true>(a, w, v, wt);
    svbksb_vec(M, N, w, a, M, v, N, b, x);
}

// cv::invert(A[3x3], DECOMP_SVD) = SVD::compute + SVD::backSubst(rhs = empty).
// scratch: a(9) v(9) w(3) wt(3)
template <class A>
__host__ __device__ void invert3_svd(A Amat, A Ainv, A a, A v, A w, A wt)
{
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a[j*3 + i] = Amat[i*3 + j];
    jacobi_svd<3, 3, true>(a, w, v, wt);
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
