/*
 * o_fivepoint.c -- CPU ORACLE (test infrastructure): the minimal solver and error of
 * cv::findEssentialMat (reference call: VOU:147) -- [UPSTREAM] calib3d/src/five-point.cpp
 * EMEstimatorCallback::runKernel / computeError, core solvePoly, LU-based invert, SVD::solveZ.
 * SURVEY.md App. A.4.  PARITY UNPINNED vs OpenCV.
 *
 * Structure follows OpenCV's Nister/Stewenius formulation: 5x9 epipolar system -> 4-D null space
 * (SVD with FULL_UV: the four extra rows of Vt come from JacobiSVD's zero-singular-value completion
 * with RNG(0x12345678)) -> 10 cubic constraints in (x,y,z) over 20 monomials -> Gauss-Jordan
 * (inv(A[:, :10]) * A[:, 10:]) -> 3x13 polynomial matrix B(z) -> det B(z) = 0 (degree 10) -> real
 * roots (Durand-Kerner, 300 iterations) -> (x,y) from the null vector of B(z) -> E, unit Frobenius
 * norm.  OpenCV expands the constraint coefficients and the determinant polynomial with generated
 * formulas; here they are produced by small polynomial routines (same polynomials, the order of
 * the additions inside one coefficient differs) -- a documented departure that cannot be pinned here.
 */
#include "uvo_oracle.h"
#include <math.h>
#include <float.h>
#include <string.h>

/* monomial order of the 20 columns ([UPSTREAM] five-point.cpp / Stewenius):
 * x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy | xz^2 xz x yz^2 yz y z^3 z^2 z 1 */
static const int MONO[20][3] = {
    {3,0,0},{0,3,0},{2,1,0},{1,2,0},{2,0,1},{2,0,0},{0,2,1},{0,2,0},{1,1,1},{1,1,0},
    {1,0,2},{1,0,1},{1,0,0},{0,1,2},{0,1,1},{0,1,0},{0,0,3},{0,0,2},{0,0,1},{0,0,0} };
static int mono_index(int a, int b, int c)
{
    for (int i = 0; i < 20; i++) if (MONO[i][0] == a && MONO[i][1] == b && MONO[i][2] == c) return i;
    return -1;
}
/* polynomials in (x,y,z) of total degree <= 3 as 20 coefficients in MONO order */
typedef struct { double c[20]; } poly3;
static void p_zero(poly3* p) { memset(p, 0, sizeof(*p)); }
static void p_lin(poly3* p, double x, double y, double z, double w)
{
    p_zero(p);
    p->c[12] = x; p->c[15] = y; p->c[18] = z; p->c[19] = w;
}
static void p_mul(const poly3* a, const poly3* b, poly3* out)
{
    poly3 r; p_zero(&r);
    for (int i = 0; i < 20; i++) {
        if (a->c[i] == 0) continue;
        for (int j = 0; j < 20; j++) {
            if (b->c[j] == 0) continue;
            int e0 = MONO[i][0] + MONO[j][0], e1 = MONO[i][1] + MONO[j][1], e2 = MONO[i][2] + MONO[j][2];
            if (e0 + e1 + e2 > 3) continue;           /* never happens for the products formed below */
            r.c[mono_index(e0, e1, e2)] += a->c[i] * b->c[j];
        }
    }
    *out = r;
}
static void p_axpy(poly3* y, double a, const poly3* x) { for (int i = 0; i < 20; i++) y->c[i] += a * x->c[i]; }

/* [UPSTREAM] getCoeffMat: rows = det(E) and the nine entries of 2 E E^T E - trace(E E^T) E,
 * E = x X + y Y + z Z + W with X,Y,Z,W the null-space basis (row-major 3x3). */
static void get_coeff_mat(const double* e /* 4 x 9 */, double* A /* 10 x 20 */)
{
    poly3 E[3][3], EEt[3][3], t, acc, tr;
    for (int i = 0; i < 9; i++) p_lin(&E[i/3][i%3], e[0*9 + i], e[1*9 + i], e[2*9 + i], e[3*9 + i]);
    /* det(E) */
    poly3 m0, m1, m2, d;
    p_mul(&E[1][1], &E[2][2], &m0); p_mul(&E[1][2], &E[2][1], &t); p_axpy(&m0, -1.0, &t);
    p_mul(&E[1][0], &E[2][2], &m1); p_mul(&E[1][2], &E[2][0], &t); p_axpy(&m1, -1.0, &t);
    p_mul(&E[1][0], &E[2][1], &m2); p_mul(&E[1][1], &E[2][0], &t); p_axpy(&m2, -1.0, &t);
    p_mul(&E[0][0], &m0, &d);
    p_mul(&E[0][1], &m1, &t); p_axpy(&d, -1.0, &t);
    p_mul(&E[0][2], &m2, &t); p_axpy(&d, 1.0, &t);
    memcpy(A, d.c, sizeof(double) * 20);
    /* E E^T and its trace */
    p_zero(&tr);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        p_zero(&acc);
        for (int k = 0; k < 3; k++) { p_mul(&E[i][k], &E[j][k], &t); p_axpy(&acc, 1.0, &t); }
        EEt[i][j] = acc;
        if (i == j) p_axpy(&tr, 1.0, &acc);
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        p_zero(&acc);
        for (int k = 0; k < 3; k++) { p_mul(&EEt[i][k], &E[k][j], &t); p_axpy(&acc, 2.0, &t); }
        p_mul(&tr, &E[i][j], &t); p_axpy(&acc, -1.0, &t);
        memcpy(A + (1 + i*3 + j) * 20, acc.c, sizeof(double) * 20);
    }
}

/* [UPSTREAM] lapack.cpp LUImpl<double> (partial pivoting, eps = DBL_EPSILON*100); b: m x n right-hand sides */
static int lu_solve(double* A, int astep, int m, double* b, int bstep, int n)
{
    const double eps = DBL_EPSILON * 100;
    int i, j, k, p = 1;
    for (i = 0; i < m; i++) {
        k = i;
        for (j = i+1; j < m; j++) if (fabs(A[j*astep + i]) > fabs(A[k*astep + i])) k = j;
        if (fabs(A[k*astep + i]) < eps) return 0;
        if (k != i) {
            for (j = i; j < m; j++) { double t = A[i*astep + j]; A[i*astep + j] = A[k*astep + j]; A[k*astep + j] = t; }
            if (b) for (j = 0; j < n; j++) { double t = b[i*bstep + j]; b[i*bstep + j] = b[k*bstep + j]; b[k*bstep + j] = t; }
            p = -p;
        }
        double d = -1/A[i*astep + i];
        for (j = i+1; j < m; j++) {
            double alpha = A[j*astep + i]*d;
            for (k = i+1; k < m; k++) A[j*astep + k] += alpha*A[i*astep + k];
            if (b) for (k = 0; k < n; k++) b[j*bstep + k] += alpha*b[i*bstep + k];
        }
    }
    if (b) {
        for (i = m-1; i >= 0; i--)
            for (j = 0; j < n; j++) {
                double s = b[i*bstep + j];
                for (k = i+1; k < m; k++) s -= A[i*astep + k]*b[k*bstep + j];
                b[i*bstep + j] = s/A[i*astep + i];
            }
    }
    return p;
}

/* [UPSTREAM] core mathfuncs.cpp solvePoly (real coefficients c[0..n], 300 iterations of Durand-Kerner
 * with the repeated-root branch omitted: it only triggers when two iterates coincide exactly) */
typedef struct { double re, im; } cplx;
static cplx c_mul(cplx a, cplx b) { cplx r = { a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re }; return r; }
static cplx c_sub(cplx a, cplx b) { cplx r = { a.re - b.re, a.im - b.im }; return r; }
static cplx c_add(cplx a, cplx b) { cplx r = { a.re + b.re, a.im + b.im }; return r; }
static cplx c_div(cplx a, cplx b)
{
    double t = 1./(b.re*b.re + b.im*b.im);
    cplx r = { (a.re*b.re + a.im*b.im)*t, (-a.re*b.im + a.im*b.re)*t };
    return r;
}
int orc_solve_poly(const double* coeffs0, int n0, double* roots_re, double* roots_im)
{
    cplx coeffs[16], roots[16];
    int n = n0, i, j, iter;
    for (i = 0; i <= n; i++) { coeffs[i].re = coeffs0[i]; coeffs[i].im = 0; }
    for (; n > 1; n--) if (fabs(coeffs[n].re) + fabs(coeffs[n].im) > DBL_EPSILON) break;
    cplx p = {1, 0}, r = {1, 1};
    for (i = 0; i < n; i++) { roots[i] = p; p = c_mul(p, r); }
    for (iter = 0; iter < 300; iter++) {
        double maxDiff = 0;
        for (i = 0; i < n; i++) {
            p = roots[i];
            cplx num = coeffs[n], denom = coeffs[n];
            for (j = 0; j < n; j++) {
                num = c_add(c_mul(num, p), coeffs[n-j-1]);
                if (j != i) {
                    cplx df = c_sub(p, roots[j]);
                    if (df.re != 0 || df.im != 0) denom = c_mul(denom, df);
                }
            }
            num = c_div(num, denom);
            roots[i] = c_sub(p, num);
            double a = sqrt(num.re*num.re + num.im*num.im);
            if (a > maxDiff) maxDiff = a;
        }
        if (maxDiff <= 0) break;
    }
    for (i = 0; i < n; i++) if (fabs(roots[i].im) < 1e-100) roots[i].im = 0;
    for (; n < n0; n++) roots[n] = roots[n-1];          /* degenerate leading coefficients: repeat the last root */
    for (i = 0; i < n0; i++) { roots_re[i] = roots[i].re; roots_im[i] = roots[i].im; }
    return n0;
}

/* polynomials in z, c[k] multiplies z^k */
static void pz_mul(const double* a, int da, const double* b, int db, double* out)
{
    for (int i = 0; i <= da + db; i++) out[i] = 0;
    for (int i = 0; i <= da; i++) for (int j = 0; j <= db; j++) out[i + j] += a[i] * b[j];
}

static double g_last_poly[11];
void orc_five_point_last_poly(double* c) { memcpy(c, g_last_poly, sizeof(g_last_poly)); }   /* test/debug hook */

/* [UPSTREAM] EMEstimatorCallback::runKernel for 5 normalised correspondences (q1, q2: 5 x 2 doubles).
 * models: up to 10 matrices of 9 doubles (row-major, x2^T E x1 = 0).  Returns their number. */
int orc_five_point(const double* q1, const double* q2, double* models)
{
    const int n = 5;
    double Q[5 * 9];
    for (int i = 0; i < n; i++) {
        double x1 = q1[2*i], y1 = q1[2*i+1], x2 = q2[2*i], y2 = q2[2*i+1];
        double* r = Q + i*9;
        r[0] = x2*x1; r[1] = x2*y1; r[2] = x2; r[3] = y2*x1; r[4] = y2*y1; r[5] = y2; r[6] = x1; r[7] = y1; r[8] = 1.0;
    }
    /* SVD::compute(Q, W, U, Vt, MODIFY_A | FULL_UV): m < n => At = Q (5 rows of 9), urows = 9, the last four
     * rows of Vt are JacobiSVD's completion of the zero singular values */
    double At[9 * 9], W[9], V5[5 * 5];
    memset(At, 0, sizeof(At));
    memcpy(At, Q, sizeof(Q));
    orc_jacobi_svd(At, 9, W, V5, 5, 9, 5, 9);
    const double* EE = At + 5 * 9;                       /* 4 x 9: rows 5..8 of Vt */
    double A[10 * 20];
    get_coeff_mat(EE, A);
    /* A = A[:, 0:10].inv() * A[:, 10:20]  (invert DECOMP_LU: LU with identity right-hand side, then gemm) */
    double Al[100], Ainv[100], Ar[100], Ap[100];
    for (int i = 0; i < 10; i++) for (int j = 0; j < 10; j++) { Al[i*10 + j] = A[i*20 + j]; Ar[i*10 + j] = A[i*20 + 10 + j]; Ainv[i*10 + j] = i == j; }
    if (lu_solve(Al, 10, 10, Ainv, 10, 10) == 0) memset(Ainv, 0, sizeof(Ainv));     /* singular: invert() yields zeros */
    for (int i = 0; i < 10; i++) for (int j = 0; j < 10; j++) {
        double s = 0;
        for (int k = 0; k < 10; k++) s += Ainv[i*10 + k] * Ar[k*10 + j];
        Ap[i*10 + j] = s;
    }
    /* B (3 x 13): row(x^2 z) - z*row(x^2), row(y^2 z) - z*row(y^2), row(xyz) - z*row(xy) */
    double b[3 * 13];
    for (int i = 0; i < 3; i++) {
        const double* a1 = Ap + (i*2 + 4) * 10; const double* a2 = Ap + (i*2 + 5) * 10;
        double row1[13] = {0}, row2[13] = {0};
        for (int k = 0; k < 3; k++) { row1[1 + k] = a1[k]; row1[5 + k] = a1[3 + k]; }
        for (int k = 0; k < 4; k++) row1[9 + k] = a1[6 + k];
        for (int k = 0; k < 3; k++) { row2[k] = a2[k]; row2[4 + k] = a2[3 + k]; }
        for (int k = 0; k < 4; k++) row2[8 + k] = a2[6 + k];
        for (int k = 0; k < 13; k++) b[i*13 + k] = row1[k] - row2[k];
    }
    /* det B(z): entries (r,0), (r,1) cubic, (r,2) quartic; stored high-to-low in b, polynomials low-to-high */
    double e[3][3][5];
    for (int r = 0; r < 3; r++) {
        for (int k = 0; k < 4; k++) { e[r][0][k] = b[r*13 + 3 - k]; e[r][1][k] = b[r*13 + 7 - k]; }
        e[r][0][4] = e[r][1][4] = 0;
        for (int k = 0; k < 5; k++) e[r][2][k] = b[r*13 + 12 - k];
    }
    double c[11], t1[12], t2[12], m[12];
    for (int k = 0; k < 11; k++) c[k] = 0;
    /* + e00*(e11*e22 - e12*e21) - e01*(e10*e22 - e12*e20) + e02*(e10*e21 - e11*e20) */
    pz_mul(e[1][1], 3, e[2][2], 4, t1); pz_mul(e[1][2], 4, e[2][1], 3, t2);
    for (int k = 0; k <= 7; k++) m[k] = t1[k] - t2[k];
    pz_mul(e[0][0], 3, m, 7, t1); for (int k = 0; k <= 10; k++) c[k] += t1[k];
    pz_mul(e[1][0], 3, e[2][2], 4, t1); pz_mul(e[1][2], 4, e[2][0], 3, t2);
    for (int k = 0; k <= 7; k++) m[k] = t1[k] - t2[k];
    pz_mul(e[0][1], 3, m, 7, t1); for (int k = 0; k <= 10; k++) c[k] -= t1[k];
    pz_mul(e[1][0], 3, e[2][1], 3, t1); pz_mul(e[1][1], 3, e[2][0], 3, t2);
    for (int k = 0; k <= 6; k++) m[k] = t1[k] - t2[k];
    pz_mul(e[0][2], 4, m, 6, t1); for (int k = 0; k <= 10; k++) c[k] += t1[k];

    memcpy(g_last_poly, c, sizeof(g_last_poly));
    double rre[10], rim[10];
    orc_solve_poly(c, 10, rre, rim);
    int count = 0;
    for (int i = 0; i < 10; i++) {
        if (fabs(rim[i]) > 1e-10) continue;
        double z1 = rre[i], z2 = z1*z1, z3 = z2*z1, z4 = z3*z1;
        double bz[9], w[3], u[9], vt[9];
        for (int j = 0; j < 3; j++) {
            const double* br = b + j*13;
            bz[j*3 + 0] = br[0]*z3 + br[1]*z2 + br[2]*z1 + br[3];
            bz[j*3 + 1] = br[4]*z3 + br[5]*z2 + br[6]*z1 + br[7];
            bz[j*3 + 2] = br[8]*z4 + br[9]*z3 + br[10]*z2 + br[11]*z1 + br[12];
        }
        orc_svd(bz, 3, 3, w, u, vt);                      /* SVD::solveZ: last row of vt */
        if (fabs(vt[8]) < 1e-10) continue;
        double xs = vt[6] / vt[8], ys = vt[7] / vt[8];
        double Ev[9], nrm = 0;
        for (int k = 0; k < 9; k++) { Ev[k] = EE[0*9 + k]*xs + EE[1*9 + k]*ys + EE[2*9 + k]*z1 + EE[3*9 + k]; }
        for (int k = 0; k < 9; k++) nrm += Ev[k]*Ev[k];
        nrm = sqrt(nrm);
        for (int k = 0; k < 9; k++) models[count*9 + k] = Ev[k] / nrm;
        count++;
    }
    return count;
}

/* [UPSTREAM] EMEstimatorCallback::computeError: Sampson distance, double -> float */
void orc_sampson_error(const double* p1, const double* p2, int n, const double* E, float* err)
{
    for (int i = 0; i < n; i++) {
        double x1 = p1[2*i], y1 = p1[2*i+1], x2 = p2[2*i], y2 = p2[2*i+1];
        double Ex1[3] = { E[0]*x1 + E[1]*y1 + E[2]*1., E[3]*x1 + E[4]*y1 + E[5]*1., E[6]*x1 + E[7]*y1 + E[8]*1. };
        double Etx2[2] = { E[0]*x2 + E[3]*y2 + E[6]*1., E[1]*x2 + E[4]*y2 + E[7]*1. };
        double x2tEx1 = x2*Ex1[0] + y2*Ex1[1] + 1.*Ex1[2];
        double a = Ex1[0]*Ex1[0], b = Ex1[1]*Ex1[1], c = Etx2[0]*Etx2[0], d = Etx2[1]*Etx2[1];
        err[i] = (float)(x2tEx1*x2tEx1 / (a + b + c + d));
    }
}
