/*
 * o_homography.c -- CPU ORACLE (test infrastructure): cv::findHomography (VOU:152) and
 * cv::decomposeHomographyMat (VOU:585).  Restates [UPSTREAM] calib3d/src/fundam.cpp
 * (HomographyEstimatorCallback: checkSubset / runKernel / computeError, HomographyRefineCallback,
 * findHomography), levmarq.cpp (LMSolverImpl::run), core lapack.cpp (JacobiImpl_ symmetric eigen
 * solver, solve / invert with DECOMP_EIG), homography_decomp.cpp (HomographyDecompInria).
 * SURVEY.md App. A.5.  PARITY UNPINNED vs OpenCV.
 */
#include "uvo_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

int orc_homography_robust(const float* src, const float* dst, int n, int method, double threshold, int maxIters, double confidence,
                          double* H, uint8_t* mask);

/* [UPSTREAM] lapack.cpp JacobiImpl_<double>: eigenvalues descending in W, eigenvectors in the rows of V.
 * std::hypot -> orc_hypot (shared with the HIP path). */
void orc_jacobi_eigen(double* A, int n, double* W, double* V)
{
    const double eps = DBL_EPSILON;
    int i, j, k, m, indR[16], indC[16];
    double mv = 0;
    for (i = 0; i < n; i++) { for (j = 0; j < n; j++) V[i*n + j] = 0; V[i*n + i] = 1; }
    int iters, maxIters = n*n*30;
    for (k = 0; k < n; k++) {
        W[k] = A[(n + 1)*k];
        if (k < n - 1) {
            for (m = k+1, mv = fabs(A[n*k + m]), i = k+2; i < n; i++) { double val = fabs(A[n*k+i]); if (mv < val) mv = val, m = i; }
            indR[k] = m;
        }
        if (k > 0) {
            for (m = 0, mv = fabs(A[k]), i = 1; i < k; i++) { double val = fabs(A[n*i+k]); if (mv < val) mv = val, m = i; }
            indC[k] = m;
        }
    }
    if (n > 1) for (iters = 0; iters < maxIters; iters++) {
        for (k = 0, mv = fabs(A[indR[0]]), i = 1; i < n-1; i++) { double val = fabs(A[n*i + indR[i]]); if (mv < val) mv = val, k = i; }
        int l = indR[k];
        for (i = 1; i < n; i++) { double val = fabs(A[n*indC[i] + i]); if (mv < val) mv = val, k = indC[i], l = i; }
        double p = A[n*k + l];
        if (fabs(p) <= eps) break;
        double y = (W[l] - W[k])*0.5;
        double t = fabs(y) + orc_hypot(p, y);
        double s = orc_hypot(p, t);
        double c = t/s;
        s = p/s; t = (p/t)*p;
        if (y < 0) s = -s, t = -t;
        A[n*k + l] = 0;
        W[k] -= t;
        W[l] += t;
        double a0, b0;
#define ROT(v0, v1) a0 = v0, b0 = v1, v0 = a0*c - b0*s, v1 = a0*s + b0*c
        for (i = 0; i < k; i++) ROT(A[n*i+k], A[n*i+l]);
        for (i = k+1; i < l; i++) ROT(A[n*k+i], A[n*i+l]);
        for (i = l+1; i < n; i++) ROT(A[n*k+i], A[n*l+i]);
        for (i = 0; i < n; i++) ROT(V[n*k+i], V[n*l+i]);
#undef ROT
        for (j = 0; j < 2; j++) {
            int idx = j == 0 ? k : l;
            if (idx < n - 1) {
                for (m = idx+1, mv = fabs(A[n*idx + m]), i = idx+2; i < n; i++) { double val = fabs(A[n*idx+i]); if (mv < val) mv = val, m = i; }
                indR[idx] = m;
            }
            if (idx > 0) {
                for (m = 0, mv = fabs(A[idx]), i = 1; i < idx; i++) { double val = fabs(A[n*i+idx]); if (mv < val) mv = val, m = i; }
                indC[idx] = m;
            }
        }
    }
    for (k = 0; k < n-1; k++) {
        m = k;
        for (i = k+1; i < n; i++) if (W[m] < W[i]) m = i;
        if (k != m) {
            double t = W[m]; W[m] = W[k]; W[k] = t;
            for (i = 0; i < n; i++) { t = V[n*m + i]; V[n*m + i] = V[n*k + i]; V[n*k + i] = t; }
        }
    }
}

/* ---- HomographyEstimatorCallback ---- */
static int have_collinear_points(const float* m, int count)
{
    int j, k, i = count - 1;
    for (j = 0; j < i; j++) {
        double dx1 = m[2*j] - m[2*i], dy1 = m[2*j+1] - m[2*i+1];
        for (k = 0; k < j; k++) {
            double dx2 = m[2*k] - m[2*i], dy2 = m[2*k+1] - m[2*i+1];
            if (fabs(dx2*dy1 - dy2*dx1) <= FLT_EPSILON*(fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return 1;
        }
    }
    return 0;
}
static double det3d(double a0, double a1, double a2, double a3, double a4, double a5, double a6, double a7, double a8)
{
    return a0*(a4*a8 - a5*a7) - a1*(a3*a8 - a5*a6) + a2*(a3*a7 - a4*a6);
}
int orc_homography_check_subset(const float* ms1, const float* ms2, int count)
{
    if (have_collinear_points(ms1, count) || have_collinear_points(ms2, count)) return 0;
    if (count == 4) {
        static const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
        int negative = 0;
        for (int i = 0; i < 4; i++) {
            const int* t = tt[i];
            double dA = det3d(ms1[2*t[0]], ms1[2*t[0]+1], 1., ms1[2*t[1]], ms1[2*t[1]+1], 1., ms1[2*t[2]], ms1[2*t[2]+1], 1.);
            double dB = det3d(ms2[2*t[0]], ms2[2*t[0]+1], 1., ms2[2*t[1]], ms2[2*t[1]+1], 1., ms2[2*t[2]], ms2[2*t[2]+1], 1.);
            negative += dA*dB < 0;
        }
        if (negative != 0 && negative != 4) return 0;
    }
    return 1;
}

/* runKernel: normalised DLT, 9x9 L^T L, eigenvector of the smallest eigenvalue */
int orc_homography_kernel(const float* M, const float* m, int count, double* Hout)
{
    double LtL[81], W[9], V[81];
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    int i;
    for (i = 0; i < count; i++) { cmx += m[2*i]; cmy += m[2*i+1]; cMx += M[2*i]; cMy += M[2*i+1]; }
    cmx /= count; cmy /= count; cMx /= count; cMy /= count;
    for (i = 0; i < count; i++) {
        smx += fabs(m[2*i] - cmx); smy += fabs(m[2*i+1] - cmy);
        sMx += fabs(M[2*i] - cMx); sMy += fabs(M[2*i+1] - cMy);
    }
    if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON) return 0;
    smx = count/smx; smy = count/smy; sMx = count/sMx; sMy = count/sMy;
    double invHnorm[9] = { 1./smx, 0, cmx, 0, 1./smy, cmy, 0, 0, 1 };
    double Hnorm2[9] = { sMx, 0, -cMx*sMx, 0, sMy, -cMy*sMy, 0, 0, 1 };
    memset(LtL, 0, sizeof(LtL));
    for (i = 0; i < count; i++) {
        double x = (m[2*i] - cmx)*smx, y = (m[2*i+1] - cmy)*smy;
        double X = (M[2*i] - cMx)*sMx, Y = (M[2*i+1] - cMy)*sMy;
        double Lx[9] = { X, Y, 1, 0, 0, 0, -x*X, -x*Y, -x };
        double Ly[9] = { 0, 0, 0, X, Y, 1, -y*X, -y*Y, -y };
        for (int j = 0; j < 9; j++) for (int k = j; k < 9; k++) LtL[j*9 + k] += Lx[j]*Lx[k] + Ly[j]*Ly[k];
    }
    for (int j = 0; j < 9; j++) for (int k = 0; k < j; k++) LtL[j*9 + k] = LtL[k*9 + j];      /* completeSymm */
    orc_jacobi_eigen(LtL, 9, W, V);
    const double* H0 = V + 8*9;
    double Ht[9], H1[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {              /* _Htemp = _invHnorm * _H0 (gemm small 3x3 case) */
        Ht[r*3 + c] = invHnorm[r*3]*H0[c] + invHnorm[r*3+1]*H0[3 + c] + invHnorm[r*3+2]*H0[6 + c];
    }
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {              /* _H0 = _Htemp * _Hnorm2 */
        H1[r*3 + c] = Ht[r*3]*Hnorm2[c] + Ht[r*3+1]*Hnorm2[3 + c] + Ht[r*3+2]*Hnorm2[6 + c];
    }
    double sc = 1./H1[8];
    for (int k = 0; k < 9; k++) Hout[k] = H1[k]*sc;                        /* convertTo(.., 1./H(2,2)) */
    return 1;
}

void orc_homography_error(const float* M, const float* m, int count, const double* H, float* err)
{
    float Hf[8];
    for (int k = 0; k < 8; k++) Hf[k] = (float)H[k];
    for (int i = 0; i < count; i++) {
        float Mx = M[2*i], My = M[2*i+1];
        float ww = 1.f/(Hf[6]*Mx + Hf[7]*My + 1.f);
        float dx = (Hf[0]*Mx + Hf[1]*My + Hf[2])*ww - m[2*i];
        float dy = (Hf[3]*Mx + Hf[4]*My + Hf[5])*ww - m[2*i+1];
        err[i] = dx*dx + dy*dy;
    }
}

/* ---- HomographyRefineCallback::compute + LMSolverImpl::run (maxIters 10, eps FLT_EPSILON) ---- */
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
static double norm_l2sqr(const double* a, int n)      /* normL2Sqr<double,double>, unrolled by 4 */
{
    double s = 0; int i = 0;
    for (; i <= n - 4; i += 4) { double v0 = a[i], v1 = a[i+1], v2 = a[i+2], v3 = a[i+3]; s += v0*v0 + v1*v1 + v2*v2 + v3*v3; }
    for (; i < n; i++) { double v = a[i]; s += v*v; }
    return s;
}
static double dot_n(const double* a, const double* b, int n)   /* dotProd_<double>, unrolled by 4 */
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
/* cv::solve(A, b, x, DECOMP_EIG) for symmetric 8x8: Jacobi eigen + SVBkSb with u = v = eigenvectors */
static void solve_eig8(const double* A, const double* b, double* x)
{
    double a[64], w[8], v[64];
    memcpy(a, A, sizeof(a));
    orc_jacobi_eigen(a, 8, w, v);
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
    double a[64], w[8], v[64];
    memcpy(a, A, sizeof(a));
    orc_jacobi_eigen(a, 8, w, v);
    double threshold = 0;
    for (int i = 0; i < 64; i++) Ainv[i] = 0;
    for (int i = 0; i < 8; i++) threshold += w[i];
    threshold *= DBL_EPSILON * 2;
    for (int i = 0; i < 8; i++) {
        double wi = w[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1/wi;
        double buffer[8];
        for (int j = 0; j < 8; j++) buffer[j] = v[i*8 + j]*wi;          /* u = vt^T: u(j,i) = v[i][j] */
        for (int k = 0; k < 8; k++) { double sv = v[i*8 + k]; for (int j = 0; j < 8; j++) Ainv[k*8 + j] = Ainv[k*8 + j] + sv*buffer[j]; }
    }
}
static void lm_refine_homography(const float* M, const float* m, int count, double* h /* 8 */)
{
    const int lx = 8, maxIters = 10; const double epsx = FLT_EPSILON, epsf = FLT_EPSILON;
    int rows = 2*count;
    double* r = (double*)malloc(sizeof(double)*rows); double* rd = (double*)malloc(sizeof(double)*rows);
    double* J = (double*)malloc(sizeof(double)*rows*8);
    double x[8], xd[8], d[8], v[8], A[64], Ap[64], D[8], temp_d[8];
    memcpy(x, h, sizeof(x));
    refine_compute(M, m, count, x, r, J);
    double S = norm_l2sqr(r, rows);
    jtj_jtr(J, r, rows, A, v);
    for (int i = 0; i < lx; i++) D[i] = A[i*8 + i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    int iter = 0;
    for (;;) {
        memcpy(Ap, A, sizeof(Ap));
        for (int i = 0; i < lx; i++) Ap[i*8 + i] += lambda*D[i];
        solve_eig8(Ap, v, d);
        for (int i = 0; i < lx; i++) xd[i] = x[i] - d[i];
        refine_compute(M, m, count, xd, rd, NULL);
        double Sd = norm_l2sqr(rd, rows);
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
            refine_compute(M, m, count, x, r, J);
            jtj_jtr(J, r, rows, A, v);
        }
        iter++;
        int proceed = iter < maxIters && norm_inf(d, lx) >= epsx && norm_inf(r, rows) >= epsf;
        if (!proceed) break;
    }
    memcpy(h, x, sizeof(x));
    free(r); free(rd); free(J);
}

/* cv::findHomography(points1, points2, method (4|8), ransacReprojThreshold, mask, maxIters, confidence).
 * Returns 1 with H (9 doubles) or 0 (OpenCV returns an empty matrix and a zero mask). */
int orc_find_homography(const orc_point2f* p1, const orc_point2f* p2, int npoints, int method, double thr, int maxIters, double confidence,
                        double* H, uint8_t* mask)
{
    if (thr <= 0) thr = 3;
    float* src = (float*)malloc(sizeof(float)*2*(npoints + 1)); float* dst = (float*)malloc(sizeof(float)*2*(npoints + 1));
    for (int i = 0; i < npoints; i++) { src[2*i] = p1[i].x; src[2*i+1] = p1[i].y; dst[2*i] = p2[i].x; dst[2*i+1] = p2[i].y; }
    int result;
    if (npoints == 4) { memset(mask, 1, npoints); result = orc_homography_kernel(src, dst, npoints, H) > 0; }
    else result = orc_homography_robust(src, dst, npoints, method, thr, maxIters, confidence, H, mask);
    if (result && npoints > 4) {
        int n = 0;                                                      /* compressElems */
        for (int i = 0; i < npoints; i++) if (mask[i]) { src[2*n] = src[2*i]; src[2*n+1] = src[2*i+1]; dst[2*n] = dst[2*i]; dst[2*n+1] = dst[2*i+1]; n++; }
        if (n > 0) {
            orc_homography_kernel(src, dst, n, H);                      /* refit on the inliers (RANSAC / LMEDS) */
            lm_refine_homography(src, dst, n, H);                       /* H8 aliases H[0..7]; H[8] stays 1 */
        }
    }
    if (!result) memset(mask, 0, npoints);
    free(src); free(dst);
    return result;
}

/* ---- cv::decomposeHomographyMat (HomographyDecompInria) ---- */
static void m3mul(const double* a, const double* b, double* out)
{
    double r[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += a[i*3+k]*b[k*3+j]; r[i*3+j] = s; }
    memcpy(out, r, sizeof(r));
}
static double m3det(const double* m) { return m[0]*(m[4]*m[8] - m[5]*m[7]) - m[1]*(m[3]*m[8] - m[5]*m[6]) + m[2]*(m[3]*m[7] - m[4]*m[6]); }
/* Matx33d::inv() -> Matx_FastInvOp<double,3>: adjugate / determinant */
static void m3inv(const double* a, double* b)
{
    double d = m3det(a);
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
    m3mul(Hn, T, R);
    if (m3det(R) < 0) for (int i = 0; i < 9; i++) R[i] *= -1;
}
int orc_decompose_homography_mat(const double* H, const double* K, double* Rs /* 4x9 */, double* ts /* 4x3 */, double* ns /* 4x3 */)
{
    double Kinv[9], Hn[9], tmp[9], w[3], u[9], vt[9];
    m3inv(K, Kinv);
    m3mul(Kinv, H, tmp); m3mul(tmp, K, Hn);
    orc_svd(Hn, 3, 3, w, u, vt);                                          /* removeScale */
    { double s = 1.0/w[1]; for (int i = 0; i < 9; i++) Hn[i] = Hn[i]*s; }
    const double epsilon = 0.001;
    double S[9], Ht[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Ht[i*3 + j] = Hn[j*3 + i];
    m3mul(Ht, Hn, S);
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
