/*
 * o_core.c -- CPU ORACLE (test infrastructure): scalar helpers, RNG, dense
 * linear algebra.  Restates [UPSTREAM] OpenCV 4.5.x core routines that the
 * reference reaches through calib3d.
 * PARITY UNPINNED vs OpenCV (see uvo_oracle.h).
 */
#include "uvo_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

/* ---- [UPSTREAM] core/fast_math.hpp: cvRound = lrint (round half to even) ---- */
int orc_cvRound(double v)  { return (int)lrint(v); }
int orc_cvRoundf(float v)  { return (int)lrintf(v); }
int orc_cvFloor(double v)  { int i = (int)v; return i - (i > v); }
int orc_cvCeil(double v)   { int i = (int)v; return i + (i < v); }

/* ---- [UPSTREAM] cv::RNG (core/operations.hpp): multiply-with-carry ---- */
void orc_rng_init(orc_rng* r, uint64_t seed) { r->state = seed ? seed : 0xffffffffULL; }
uint32_t orc_rng_next(orc_rng* r)
{
    r->state = (uint64_t)(uint32_t)r->state * 4164903690U + (uint32_t)(r->state >> 32);
    return (uint32_t)r->state;
}
int orc_rng_uniform(orc_rng* r, int a, int b)
{
    return a == b ? a : (int)(orc_rng_next(r) % (uint32_t)(b - a) + a);
}

/* ---- deterministic elementary functions (shared op-for-op with the HIP path) ---- */
double orc_hypot(double a, double b)
{
    a = fabs(a); b = fabs(b);
    if (a < b) { double t = a; a = b; b = t; }
    if (a == 0.0) return 0.0;
    double t = b / a;
    return a * sqrt(1.0 + t * t);
}

/* sin/cos: Cody-Waite reduction by pi/2 + Taylor polynomials on [-pi/4, pi/4]. */
void orc_sincos(double x, double* s, double* c)
{
    const double TWO_OVER_PI = 0.63661977236758134308;
    const double PIO2_HI = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
    const double PIO2_LO = 6.07710050650619224932e-11; /* pi/2 - PIO2_HI */
    double kf = rint(x * TWO_OVER_PI);
    double r = (x - kf * PIO2_HI) - kf * PIO2_LO;
    double r2 = r * r;
    /* sin r = r * (1 - r2/3! + r2^2/5! - ...) up to r^19 */
    double ps = -1.0 / 121645100408832000.0;            /* -1/19! */
    ps = ps * r2 + 1.0 / 355687428096000.0;             /*  1/17! */
    ps = ps * r2 - 1.0 / 1307674368000.0;               /* -1/15! */
    ps = ps * r2 + 1.0 / 6227020800.0;                  /*  1/13! */
    ps = ps * r2 - 1.0 / 39916800.0;                    /* -1/11! */
    ps = ps * r2 + 1.0 / 362880.0;                      /*  1/9!  */
    ps = ps * r2 - 1.0 / 5040.0;                        /* -1/7!  */
    ps = ps * r2 + 1.0 / 120.0;                         /*  1/5!  */
    ps = ps * r2 - 1.0 / 6.0;                           /* -1/3!  */
    double sr = r + r * (r2 * ps);
    double pc = 1.0 / 6402373705728000.0;               /*  1/18! */
    pc = pc * r2 - 1.0 / 20922789888000.0;              /* -1/16! */
    pc = pc * r2 + 1.0 / 87178291200.0;                 /*  1/14! */
    pc = pc * r2 - 1.0 / 479001600.0;                   /* -1/12! */
    pc = pc * r2 + 1.0 / 3628800.0;                     /*  1/10! */
    pc = pc * r2 - 1.0 / 40320.0;                       /* -1/8!  */
    pc = pc * r2 + 1.0 / 720.0;                         /*  1/6!  */
    pc = pc * r2 - 1.0 / 24.0;                          /* -1/4!  */
    pc = pc * r2 + 0.5;                                 /*  1/2!  */
    double cr = 1.0 - r2 * pc;
    long k = (long)kf;
    switch (k & 3) {
    case 0: *s = sr;  *c = cr;  break;
    case 1: *s = cr;  *c = -sr; break;
    case 2: *s = -sr; *c = -cr; break;
    default:*s = -cr; *c = sr;  break;
    }
}

/* acos by 6 Newton steps on cos(theta) = c (fixed count => deterministic). */
double orc_acos(double c)
{
    const double PI = 3.14159265358979323846;
    if (c >= 1.0) return 0.0;
    if (c <= -1.0) return PI;
    int neg = c < 0.0;
    double a = neg ? -c : c;
    double th = sqrt(2.0 * (1.0 - a));
    for (int it = 0; it < 6; it++) {
        double s, cc;
        orc_sincos(th, &s, &cc);
        th = th + (cc - a) / s;
    }
    return neg ? PI - th : th;
}

/* ---- [UPSTREAM] lapack.cpp JacobiSVDImpl_<double>(At, astep, W, Vt, vstep, m, n, n1, DBL_MIN, DBL_EPSILON*10)
 * Departure: std::hypot -> orc_hypot (deterministic on host and device). ---- */
void orc_jacobi_svd(double* At, int astep, double* _W, double* Vt, int vstep, int m, int n, int n1)
{
    const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
    double Wbuf[64];
    double* W = n <= 64 ? Wbuf : (double*)malloc(sizeof(double) * n);
    int i, j, k, iter, max_iter = m > 30 ? m : 30;
    double c, s, sd;
    if (!Vt) n1 = 0;

    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) { double t = At[i*astep + k]; sd += t*t; }
        W[i] = sd;
        if (Vt) { for (k = 0; k < n; k++) Vt[i*vstep + k] = 0; Vt[i*vstep + i] = 1; }
    }

    for (iter = 0; iter < max_iter; iter++) {
        int changed = 0;
        for (i = 0; i < n-1; i++)
            for (j = i+1; j < n; j++) {
                double *Ai = At + i*astep, *Aj = At + j*astep;
                double a = W[i], p = 0, b = W[j];
                for (k = 0; k < m; k++) p += Ai[k]*Aj[k];
                if (fabs(p) <= eps*sqrt(a*b)) continue;
                p *= 2;
                double beta = a - b, gamma = orc_hypot(p, beta);
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
                    double t0 = c*Ai[k] + s*Aj[k];
                    double t1 = -s*Ai[k] + c*Aj[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += t0*t0; b += t1*t1;
                }
                W[i] = a; W[j] = b;
                changed = 1;
                if (Vt) {
                    double *Vi = Vt + i*vstep, *Vj = Vt + j*vstep;
                    for (k = 0; k < n; k++) {
                        double t0 = c*Vi[k] + s*Vj[k];
                        double t1 = -s*Vi[k] + c*Vj[k];
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
            if (Vt) {
                for (k = 0; k < m; k++) { t = At[i*astep+k]; At[i*astep+k] = At[j*astep+k]; At[j*astep+k] = t; }
                for (k = 0; k < n; k++) { t = Vt[i*vstep+k]; Vt[i*vstep+k] = Vt[j*vstep+k]; Vt[j*vstep+k] = t; }
            }
        }
    }
    for (i = 0; i < n; i++) _W[i] = W[i];

    if (Vt) {
        orc_rng rng; orc_rng_init(&rng, 0x12345678);
        for (i = 0; i < n1; i++) {
            sd = i < n ? W[i] : 0;
            for (int ii = 0; ii < 100 && sd <= minval; ii++) {
                /* zero singular value: random vector orthogonalised against the previous ones */
                const double val0 = 1./m;
                for (k = 0; k < m; k++) {
                    double val = (orc_rng_next(&rng) & 256) != 0 ? val0 : -val0;
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
    if (W != Wbuf) free(W);
}

/* [UPSTREAM] lapack.cpp _SVDcompute(src, w, u, vt, flags = 0) */
void orc_svd(const double* A, int m0, int n0, double* w, double* u, double* vt)
{
    int m = m0, n = n0, at = 0, i, j;
    if (m < n) { int t = m; m = n; n = t; at = 1; }
    double* temp_a = (double*)malloc(sizeof(double) * (size_t)n * m);   /* n rows of m */
    double* temp_v = (double*)malloc(sizeof(double) * (size_t)n * n);
    if (!at) { for (i = 0; i < m; i++) for (j = 0; j < n; j++) temp_a[j*m + i] = A[i*n0 + j]; }
    else     { memcpy(temp_a, A, sizeof(double) * (size_t)n * m); }
    orc_jacobi_svd(temp_a, m, w, temp_v, n, m, n, n);
    if (!at) {
        if (u)  for (i = 0; i < n; i++) for (j = 0; j < m; j++) u[j*n + i] = temp_a[i*m + j];   /* u = temp_u^T : m x n */
        if (vt) memcpy(vt, temp_v, sizeof(double) * (size_t)n * n);
    } else {
        if (u)  for (i = 0; i < n; i++) for (j = 0; j < n; j++) u[j*n + i] = temp_v[i*n + j];   /* u = temp_v^T : n x n (= m0 x min) */
        if (vt) memcpy(vt, temp_a, sizeof(double) * (size_t)n * m);                             /* vt = temp_u : n x m */
    }
    free(temp_a); free(temp_v);
}

/* [UPSTREAM] lapack.cpp SVBkSbImpl_<double>, nb == 1 with b, or b == NULL (inverse). eps = 2*DBL_EPSILON.
 * u is given transposed (uT = true: row i of ut = i-th left vector), v transposed too. */
static void svbksb(int m, int n, const double* w, const double* ut, int ldu, const double* vt, int ldv,
                   const double* b, int nb, double* x, int ldx)
{
    double threshold = 0; int i, j, k, nm = m < n ? m : n;
    if (!b) nb = m;
    for (i = 0; i < n; i++) for (j = 0; j < nb; j++) x[i*ldx + j] = 0;
    for (i = 0; i < nm; i++) threshold += w[i];
    threshold *= DBL_EPSILON * 2;
    for (i = 0; i < nm; i++) {
        const double* u = ut + (size_t)i*ldu; const double* v = vt + (size_t)i*ldv;
        double wi = w[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1/wi;
        if (nb == 1) {
            double s = 0;
            if (b) for (j = 0; j < m; j++) s += u[j]*b[j]; else s = u[0];
            s *= wi;
            for (j = 0; j < n; j++) x[j*ldx] = x[j*ldx] + s*v[j];
        } else {
            /* b == NULL: x += v * (u^T * wi)   (MatrAXPY) */
            double buffer[16];
            for (j = 0; j < nb; j++) buffer[j] = u[j]*wi;
            for (k = 0; k < n; k++) { double sv = v[k]; for (j = 0; j < nb; j++) x[k*ldx + j] = x[k*ldx + j] + sv*buffer[j]; }
        }
    }
}

/* [UPSTREAM] lapack.cpp cv::solve(..., DECOMP_SVD): a = src^T; JacobiSVD(a, w, v, m, n); SVBkSb */
void orc_solve_svd(const double* A, int m, int n, const double* b, double* x)
{
    double a[16*16], v[16*16], w[16]; int i, j;
    for (i = 0; i < m; i++) for (j = 0; j < n; j++) a[j*m + i] = A[i*n + j];
    orc_jacobi_svd(a, m, w, v, n, m, n, n);
    svbksb(m, n, w, a, m, v, n, b, 1, x, 1);
}

/* [UPSTREAM] lapack.cpp cv::invert(3x3, DECOMP_SVD): SVD::compute + SVD::backSubst(rhs = empty) */
void orc_invert3_svd(const double* A, double* Ainv)
{
    double a[9], v[9], w[3]; int i, j;
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) a[j*3 + i] = A[i*3 + j];
    orc_jacobi_svd(a, 3, w, v, 3, 3, 3, 3);
    /* u (m x nm) = a^T; backSubst uses u non-transposed; our svbksb takes u^T = a directly */
    svbksb(3, 3, w, a, 3, v, 3, NULL, 3, Ainv, 3);
}

/* [UPSTREAM] matmul MulTransposedR<double,double> (dst < gemm_level) + completeSymm */
void orc_mul_transposed(const double* src, int rows, int cols, double* dst)
{
    int i, j, k;
    for (i = 0; i < cols; i++)
        for (j = i; j < cols; j++) {
            double s0 = 0;
            for (k = 0; k < rows; k++) s0 += src[k*cols + i] * src[k*cols + j];
            dst[i*cols + j] = s0;   /* *scale(=1) */
        }
    for (i = 0; i < cols; i++) for (j = 0; j < i; j++) dst[i*cols + j] = dst[j*cols + i];
}

/* ---- MU:65-86 compute_median ---- */
static int cmp_double(const void* a, const void* b)
{
    double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}
double orc_compute_median(const double* v, int n)
{
    if (n == 0) return 0.0;
    double* t = (double*)malloc(sizeof(double) * n);
    memcpy(t, v, sizeof(double) * n);
    qsort(t, n, sizeof(double), cmp_double);
    double r;
    if (n % 2 == 0) { int mid = n / 2; r = (t[mid - 1] + t[mid]) / 2.0; }
    else r = t[n / 2];
    free(t);
    return r;
}

/* ---- MU:35-56 compute_mean_and_variance ---- */
void orc_compute_mean_and_variance(const double* v, int N, double* mv)
{
    double sum = 0.0, sumOfSquares = 0.0;
    for (int i = 0; i < N; i++) { double value = v[i]; sum += value; sumOfSquares += value * value; }
    double mean = sum / N;
    double variance = (sumOfSquares / N) - (mean * mean);
    mv[0] = mean; mv[1] = variance;
}
