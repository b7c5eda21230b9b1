/*
 * o_pnp.c -- CPU ORACLE (test infrastructure): solvePnPRansac with the EPnP kernel.
 * Follows the call at VO:647-648 (flags = PNP_METHOD_FLAG = 1 = SOLVEPNP_EPNP,
 * useExtrinsicGuess = false, distCoeffs = zeros(4,1)) and restates [UPSTREAM] calib3d
 * solvepnp.cpp (solvePnPRansac, PnPRansacCallback, solvePnPGeneric EPNP branch),
 * ptsetreg.cpp (RANSACPointSetRegistrator::run/getSubset/findInliers,
 * RANSACUpdateNumIters), epnp.cpp, undistortPoints with zero distortion.
 * SURVEY.md App. A.3, A.7.  PARITY UNPINNED vs OpenCV.
 */
#include "uvo_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

/* [UPSTREAM] ptsetreg.cpp RANSACUpdateNumIters */
int orc_ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters)
{
    p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= maxIters*(-denom) ? maxIters : orc_cvRound(num/denom);
}

/* ------------------------------- EPnP ([UPSTREAM] epnp.cpp) ------------------------------- */
typedef struct {
    double uc, vc, fu, fv;
    int n;
    const double* pws; const double* us;
    double* alphas; double* pcs;
    double cws[4][3], ccs[4][3];
} epnp_t;

static double dot3(const double* a, const double* b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
static double dist2(const double* p1, const double* p2)
{
    return (p1[0]-p2[0])*(p1[0]-p2[0]) + (p1[1]-p2[1])*(p1[1]-p2[1]) + (p1[2]-p2[2])*(p1[2]-p2[2]);
}

static void choose_control_points(epnp_t* e)
{
    int n = e->n;
    e->cws[0][0] = e->cws[0][1] = e->cws[0][2] = 0;
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) e->cws[0][j] += e->pws[3*i + j];
    for (int j = 0; j < 3; j++) e->cws[0][j] /= n;
    double* PW0 = (double*)malloc(sizeof(double) * 3 * n);
    double pw0tpw0[9], dc[3], uct[9], u[9];
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) PW0[3*i + j] = e->pws[3*i + j] - e->cws[0][j];
    orc_mul_transposed(PW0, n, 3, pw0tpw0);
    orc_svd(pw0tpw0, 3, 3, dc, u, NULL);                  /* cvSVD(.., U_T): uct = u^T */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) uct[i*3 + j] = u[j*3 + i];
    free(PW0);
    for (int i = 1; i < 4; i++) {
        double k = sqrt(dc[i-1] / n);
        for (int j = 0; j < 3; j++) e->cws[i][j] = e->cws[0][j] + k * uct[3*(i-1) + j];
    }
}

static void compute_barycentric_coordinates(epnp_t* e)
{
    double cc[9], ci[9];
    for (int i = 0; i < 3; i++) for (int j = 1; j < 4; j++) cc[3*i + j - 1] = e->cws[j][i] - e->cws[0][i];
    orc_invert3_svd(cc, ci);
    for (int i = 0; i < e->n; i++) {
        const double* pi = e->pws + 3*i;
        double* a = e->alphas + 4*i;
        for (int j = 0; j < 3; j++)
            a[1 + j] = ci[3*j] * (pi[0] - e->cws[0][0]) + ci[3*j + 1] * (pi[1] - e->cws[0][1]) + ci[3*j + 2] * (pi[2] - e->cws[0][2]);
        a[0] = 1.0f - a[1] - a[2] - a[3];
    }
}

static void fill_M(const epnp_t* e, double* M, int row, const double* as, double u, double v)
{
    double* M1 = M + row*12; double* M2 = M1 + 12;
    for (int i = 0; i < 4; i++) {
        M1[3*i] = as[i] * e->fu; M1[3*i + 1] = 0.0; M1[3*i + 2] = as[i] * (e->uc - u);
        M2[3*i] = 0.0; M2[3*i + 1] = as[i] * e->fv; M2[3*i + 2] = as[i] * (e->vc - v);
    }
}

static void compute_L_6x10(const double* ut, double* l_6x10)
{
    const double* v[4] = { ut + 12*11, ut + 12*10, ut + 12*9, ut + 12*8 };
    double dv[4][6][3];
    for (int i = 0; i < 4; i++) {
        int a = 0, b = 1;
        for (int j = 0; j < 6; j++) {
            dv[i][j][0] = v[i][3*a] - v[i][3*b];
            dv[i][j][1] = v[i][3*a + 1] - v[i][3*b + 1];
            dv[i][j][2] = v[i][3*a + 2] - v[i][3*b + 2];
            b++;
            if (b > 3) { a++; b = a + 1; }
        }
    }
    for (int i = 0; i < 6; i++) {
        double* row = l_6x10 + 10*i;
        row[0] =        dot3(dv[0][i], dv[0][i]);
        row[1] = 2.0f * dot3(dv[0][i], dv[1][i]);
        row[2] =        dot3(dv[1][i], dv[1][i]);
        row[3] = 2.0f * dot3(dv[0][i], dv[2][i]);
        row[4] = 2.0f * dot3(dv[1][i], dv[2][i]);
        row[5] =        dot3(dv[2][i], dv[2][i]);
        row[6] = 2.0f * dot3(dv[0][i], dv[3][i]);
        row[7] = 2.0f * dot3(dv[1][i], dv[3][i]);
        row[8] = 2.0f * dot3(dv[2][i], dv[3][i]);
        row[9] =        dot3(dv[3][i], dv[3][i]);
    }
}

static void compute_rho(const epnp_t* e, double* rho)
{
    rho[0] = dist2(e->cws[0], e->cws[1]); rho[1] = dist2(e->cws[0], e->cws[2]); rho[2] = dist2(e->cws[0], e->cws[3]);
    rho[3] = dist2(e->cws[1], e->cws[2]); rho[4] = dist2(e->cws[1], e->cws[3]); rho[5] = dist2(e->cws[2], e->cws[3]);
}

static void find_betas_approx_1(const double* L, const double* rho, double* betas)
{
    double l_6x4[24], b4[4];
    for (int i = 0; i < 6; i++) { l_6x4[4*i] = L[10*i]; l_6x4[4*i+1] = L[10*i+1]; l_6x4[4*i+2] = L[10*i+3]; l_6x4[4*i+3] = L[10*i+6]; }
    orc_solve_svd(l_6x4, 6, 4, rho, b4);
    if (b4[0] < 0) { betas[0] = sqrt(-b4[0]); betas[1] = -b4[1] / betas[0]; betas[2] = -b4[2] / betas[0]; betas[3] = -b4[3] / betas[0]; }
    else           { betas[0] = sqrt(b4[0]);  betas[1] = b4[1] / betas[0];  betas[2] = b4[2] / betas[0];  betas[3] = b4[3] / betas[0]; }
}
static void find_betas_approx_2(const double* L, const double* rho, double* betas)
{
    double l_6x3[18], b3[3];
    for (int i = 0; i < 6; i++) { l_6x3[3*i] = L[10*i]; l_6x3[3*i+1] = L[10*i+1]; l_6x3[3*i+2] = L[10*i+2]; }
    orc_solve_svd(l_6x3, 6, 3, rho, b3);
    if (b3[0] < 0) { betas[0] = sqrt(-b3[0]); betas[1] = (b3[2] < 0) ? sqrt(-b3[2]) : 0.0; }
    else           { betas[0] = sqrt(b3[0]);  betas[1] = (b3[2] > 0) ? sqrt(b3[2]) : 0.0; }
    if (b3[1] < 0) betas[0] = -betas[0];
    betas[2] = 0.0; betas[3] = 0.0;
}
static void find_betas_approx_3(const double* L, const double* rho, double* betas)
{
    double l_6x5[30], b5[5];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 5; j++) l_6x5[5*i + j] = L[10*i + j];
    orc_solve_svd(l_6x5, 6, 5, rho, b5);
    if (b5[0] < 0) { betas[0] = sqrt(-b5[0]); betas[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0; }
    else           { betas[0] = sqrt(b5[0]);  betas[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0; }
    if (b5[1] < 0) betas[0] = -betas[0];
    betas[2] = b5[3] / betas[0];
    betas[3] = 0.0;
}

static void compute_A_and_b_gauss_newton(const double* l_6x10, const double* rho, const double betas[4], double* A, double* b)
{
    for (int i = 0; i < 6; i++) {
        const double* rowL = l_6x10 + i*10;
        double* rowA = A + i*4;
        rowA[0] = 2*rowL[0]*betas[0] +   rowL[1]*betas[1] +   rowL[3]*betas[2] +   rowL[6]*betas[3];
        rowA[1] =   rowL[1]*betas[0] + 2*rowL[2]*betas[1] +   rowL[4]*betas[2] +   rowL[7]*betas[3];
        rowA[2] =   rowL[3]*betas[0] +   rowL[4]*betas[1] + 2*rowL[5]*betas[2] +   rowL[8]*betas[3];
        rowA[3] =   rowL[6]*betas[0] +   rowL[7]*betas[1] +   rowL[8]*betas[2] + 2*rowL[9]*betas[3];
        b[i] = rho[i] -
            (rowL[0]*betas[0]*betas[0] + rowL[1]*betas[0]*betas[1] + rowL[2]*betas[1]*betas[1] +
             rowL[3]*betas[0]*betas[2] + rowL[4]*betas[1]*betas[2] + rowL[5]*betas[2]*betas[2] +
             rowL[6]*betas[0]*betas[3] + rowL[7]*betas[1]*betas[3] + rowL[8]*betas[2]*betas[3] +
             rowL[9]*betas[3]*betas[3]);
    }
}

/* [UPSTREAM] epnp.cpp qr_solve (Householder, in place; includes the original's eta scan quirk) */
static void qr_solve(double* pA, double* pb, double* pX, int nr, int nc)
{
    double A1[6], A2[6];
    double* ppAkk = pA;
    for (int k = 0; k < nc; k++) {
        double* ppAik1 = ppAkk; double eta = fabs(*ppAik1);
        for (int i = k + 1; i < nr; i++) { double elt = fabs(*ppAik1); if (eta < elt) eta = elt; ppAik1 += nc; }
        if (eta == 0) { A1[k] = A2[k] = 0.0; return; }
        else {
            double* ppAik2 = ppAkk; double sum2 = 0.0, inv_eta = 1. / eta;
            for (int i = k; i < nr; i++) { *ppAik2 *= inv_eta; sum2 += *ppAik2 * *ppAik2; ppAik2 += nc; }
            double sigma = sqrt(sum2);
            if (*ppAkk < 0) sigma = -sigma;
            *ppAkk += sigma;
            A1[k] = sigma * *ppAkk;
            A2[k] = -eta * sigma;
            for (int j = k + 1; j < nc; j++) {
                double* ppAik = ppAkk; double sum = 0;
                for (int i = k; i < nr; i++) { sum += *ppAik * ppAik[j - k]; ppAik += nc; }
                double tau = sum / A1[k];
                ppAik = ppAkk;
                for (int i = k; i < nr; i++) { ppAik[j - k] -= tau * *ppAik; ppAik += nc; }
            }
        }
        ppAkk += nc + 1;
    }
    double* ppAjj = pA;
    for (int j = 0; j < nc; j++) {
        double* ppAij = ppAjj; double tau = 0;
        for (int i = j; i < nr; i++) { tau += *ppAij * pb[i]; ppAij += nc; }
        tau /= A1[j];
        ppAij = ppAjj;
        for (int i = j; i < nr; i++) { pb[i] -= tau * *ppAij; ppAij += nc; }
        ppAjj += nc + 1;
    }
    pX[nc - 1] = pb[nc - 1] / A2[nc - 1];
    for (int i = nc - 2; i >= 0; i--) {
        double* ppAij = pA + i*nc + (i + 1); double sum = 0;
        for (int j = i + 1; j < nc; j++) { sum += *ppAij * pX[j]; ppAij++; }
        pX[i] = (pb[i] - sum) / A2[i];
    }
}

static void gauss_newton(const double* L, const double* rho, double betas[4])
{
    double a[24], b[6], x[4] = {0, 0, 0, 0};
    for (int k = 0; k < 5; k++) {
        compute_A_and_b_gauss_newton(L, rho, betas, a, b);
        qr_solve(a, b, x, 6, 4);
        for (int i = 0; i < 4; i++) betas[i] += x[i];
    }
}

static void compute_ccs(epnp_t* e, const double* betas, const double* ut)
{
    for (int i = 0; i < 4; i++) e->ccs[i][0] = e->ccs[i][1] = e->ccs[i][2] = 0.0f;
    for (int i = 0; i < 4; i++) {
        const double* v = ut + 12*(11 - i);
        for (int j = 0; j < 4; j++) for (int k = 0; k < 3; k++) e->ccs[j][k] += betas[i] * v[3*j + k];
    }
}
static void compute_pcs(epnp_t* e)
{
    for (int i = 0; i < e->n; i++) {
        const double* a = e->alphas + 4*i; double* pc = e->pcs + 3*i;
        for (int j = 0; j < 3; j++) pc[j] = a[0]*e->ccs[0][j] + a[1]*e->ccs[1][j] + a[2]*e->ccs[2][j] + a[3]*e->ccs[3][j];
    }
}
static void solve_for_sign(epnp_t* e)
{
    if (e->pcs[2] < 0.0) {
        for (int i = 0; i < 4; i++) for (int j = 0; j < 3; j++) e->ccs[i][j] = -e->ccs[i][j];
        for (int i = 0; i < e->n; i++) { e->pcs[3*i] = -e->pcs[3*i]; e->pcs[3*i+1] = -e->pcs[3*i+1]; e->pcs[3*i+2] = -e->pcs[3*i+2]; }
    }
}
static void estimate_R_and_t(epnp_t* e, double R[3][3], double t[3])
{
    double pc0[3] = {0,0,0}, pw0[3] = {0,0,0};
    int n = e->n;
    for (int i = 0; i < n; i++) {
        const double* pc = e->pcs + 3*i; const double* pw = e->pws + 3*i;
        for (int j = 0; j < 3; j++) { pc0[j] += pc[j]; pw0[j] += pw[j]; }
    }
    for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
    double abt[9] = {0}, abt_d[3], abt_u[9], abt_vt[9], abt_v[9];
    for (int i = 0; i < n; i++) {
        const double* pc = e->pcs + 3*i; const double* pw = e->pws + 3*i;
        for (int j = 0; j < 3; j++) {
            abt[3*j]     += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
            abt[3*j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
            abt[3*j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
        }
    }
    orc_svd(abt, 3, 3, abt_d, abt_u, abt_vt);          /* cvSVD(.., V not transposed): v = vt^T */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) abt_v[i*3 + j] = abt_vt[j*3 + i];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i][j] = dot3(abt_u + 3*i, abt_v + 3*j);
    const double det =
        R[0][0]*R[1][1]*R[2][2] + R[0][1]*R[1][2]*R[2][0] + R[0][2]*R[1][0]*R[2][1] -
        R[0][2]*R[1][1]*R[2][0] - R[0][1]*R[1][0]*R[2][2] - R[0][0]*R[1][2]*R[2][1];
    if (det < 0) { R[2][0] = -R[2][0]; R[2][1] = -R[2][1]; R[2][2] = -R[2][2]; }
    t[0] = pc0[0] - dot3(R[0], pw0);
    t[1] = pc0[1] - dot3(R[1], pw0);
    t[2] = pc0[2] - dot3(R[2], pw0);
}
static double reprojection_error(const epnp_t* e, const double R[3][3], const double t[3])
{
    double sum2 = 0.0;
    for (int i = 0; i < e->n; i++) {
        const double* pw = e->pws + 3*i;
        double Xc = dot3(R[0], pw) + t[0];
        double Yc = dot3(R[1], pw) + t[1];
        double inv_Zc = 1.0 / (dot3(R[2], pw) + t[2]);
        double ue = e->uc + e->fu * Xc * inv_Zc;
        double ve = e->vc + e->fv * Yc * inv_Zc;
        double u = e->us[2*i], v = e->us[2*i + 1];
        sum2 += sqrt((u - ue)*(u - ue) + (v - ve)*(v - ve));
    }
    return sum2 / e->n;
}
static double compute_R_and_t(epnp_t* e, const double* ut, const double* betas, double R[3][3], double t[3])
{
    compute_ccs(e, betas, ut);
    compute_pcs(e);
    solve_for_sign(e);
    estimate_R_and_t(e, R, t);
    return reprojection_error(e, R, t);
}

/* [UPSTREAM] epnp.cpp epnp::compute_pose */
void orc_epnp(const double* pws, const double* us, int n, double fu, double fv, double uc, double vc,
              double* Rout, double* tout)
{
    epnp_t e; e.uc = uc; e.vc = vc; e.fu = fu; e.fv = fv; e.n = n; e.pws = pws; e.us = us;
    e.alphas = (double*)malloc(sizeof(double) * 4 * n);
    e.pcs = (double*)malloc(sizeof(double) * 3 * n);
    choose_control_points(&e);
    compute_barycentric_coordinates(&e);
    double* M = (double*)malloc(sizeof(double) * 2 * n * 12);
    for (int i = 0; i < n; i++) fill_M(&e, M, 2*i, e.alphas + 4*i, us[2*i], us[2*i + 1]);
    double mtm[144], d[12], u[144], ut[144];
    orc_mul_transposed(M, 2*n, 12, mtm);
    orc_svd(mtm, 12, 12, d, u, NULL);                  /* cvSVD(MtM, D, Ut, 0, MODIFY_A | U_T) */
    for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) ut[i*12 + j] = u[j*12 + i];
    free(M);
    double l_6x10[60], rho[6];
    compute_L_6x10(ut, l_6x10);
    compute_rho(&e, rho);
    double Betas[4][4], rep_errors[4], Rs[4][3][3], ts[4][3];
    find_betas_approx_1(l_6x10, rho, Betas[1]);
    gauss_newton(l_6x10, rho, Betas[1]);
    rep_errors[1] = compute_R_and_t(&e, ut, Betas[1], Rs[1], ts[1]);
    find_betas_approx_2(l_6x10, rho, Betas[2]);
    gauss_newton(l_6x10, rho, Betas[2]);
    rep_errors[2] = compute_R_and_t(&e, ut, Betas[2], Rs[2], ts[2]);
    find_betas_approx_3(l_6x10, rho, Betas[3]);
    gauss_newton(l_6x10, rho, Betas[3]);
    rep_errors[3] = compute_R_and_t(&e, ut, Betas[3], Rs[3], ts[3]);
    int N = 1;
    if (rep_errors[2] < rep_errors[1]) N = 2;
    if (rep_errors[3] < rep_errors[N]) N = 3;
    memcpy(tout, ts[N], sizeof(double)*3);
    memcpy(Rout, Rs[N], sizeof(double)*9);
    free(e.alphas); free(e.pcs);
}

/* [UPSTREAM] solvePnPGeneric, EPNP branch: undistortPoints (zero distortion => (u-cx)*(1/fx), stored in
 * the depth of the image points), epnp, Rodrigues(R -> rvec).  is_f32: RANSAC-kernel call (float
 * points); otherwise the refit on double points. */
static void solve_pnp_epnp(const double* obj, const double* img, int n, int img_is_f32, const double* K,
                           double* rvec, double* tvec)
{
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double ifx = 1./fx, ify = 1./fy;
    double* us = (double*)malloc(sizeof(double) * 2 * n);
    for (int i = 0; i < n; i++) {
        double x = (img[2*i] - cx)*ifx, y = (img[2*i+1] - cy)*ify;
        if (img_is_f32) { x = (double)(float)x; y = (double)(float)y; }   /* undistorted points stored as CV_32FC2 */
        us[2*i] = x*fx + cx;            /* epnp::init_points: ipoints.x*fu + uc */
        us[2*i+1] = y*fy + cy;
    }
    double R[9];
    orc_epnp(obj, us, n, fx, fy, cx, cy, R, tvec);
    orc_rodrigues_mat2vec(R, rvec);
    free(us);
}

/* [UPSTREAM] solvepnp.cpp solvePnPRansac (model_points = 5, EPnP kernel) + ptsetreg.cpp RANSAC run */
int orc_solve_pnp_ransac(const double* obj64, const orc_point2f* img, int npoints, const double* K,
                         int iterationsCount, float reprojectionError, double confidence,
                         double* rvec_out, double* tvec_out, int* inliers, int* n_inliers)
{
    *n_inliers = 0;
    if (npoints < 4) return 0;                      /* CV_Assert(npoints >= 4) would throw */
    const int modelPoints = 5;
    /* opoints converted to CV_32F on entry */
    float* opoints = (float*)malloc(sizeof(float) * 3 * npoints);
    for (int i = 0; i < 3*npoints; i++) opoints[i] = (float)obj64[i];
    double* od = (double*)malloc(sizeof(double) * 3 * npoints);
    double* id = (double*)malloc(sizeof(double) * 2 * npoints);
    double rvec[3] = {0,0,0}, tvec[3] = {0,0,0};
    int result = 0;

    if (npoints == 4) { free(opoints); free(od); free(id); return 0; }  /* P3P path: not used by the reference config; unsupported */

    if (npoints == modelPoints) {
        for (int i = 0; i < npoints; i++) { od[3*i] = opoints[3*i]; od[3*i+1] = opoints[3*i+1]; od[3*i+2] = opoints[3*i+2]; id[2*i] = img[i].x; id[2*i+1] = img[i].y; }
        solve_pnp_epnp(od, id, npoints, 1, K, rvec_out, tvec_out);
        for (int i = 0; i < npoints; i++) inliers[i] = i;
        *n_inliers = npoints;
        free(opoints); free(od); free(id);
        return 1;
    }

    uint8_t* mask = (uint8_t*)malloc(npoints);
    uint8_t* bestMask = (uint8_t*)malloc(npoints);
    float* proj = (float*)malloc(sizeof(float) * 2 * npoints);
    double bestModel[6] = {0};
    int niters = iterationsCount > 1 ? iterationsCount : 1;
    int maxGoodCount = 0;
    double threshold = reprojectionError;            /* param1 (float -> double) */
    orc_rng rng; orc_rng_init(&rng, (uint64_t)-1);
    for (int iter = 0; iter < niters; iter++) {
        int idx[5];
        /* getSubset: checkSubset is the default (always true) for PnPRansacCallback */
        for (int i = 0; i < modelPoints; i++) {
            int idx_i;
            for (;;) {
                idx_i = orc_rng_uniform(&rng, 0, npoints);
                int dup = 0; for (int q = 0; q < i; q++) if (idx[q] == idx_i) dup = 1;
                if (!dup) break;
            }
            idx[i] = idx_i;
            od[3*i] = opoints[3*idx_i]; od[3*i+1] = opoints[3*idx_i+1]; od[3*i+2] = opoints[3*idx_i+2];
            id[2*i] = img[idx_i].x; id[2*i+1] = img[idx_i].y;
        }
        /* runKernel: solvePnP(EPNP) -> model = [rvec | tvec] (always one model) */
        solve_pnp_epnp(od, id, modelPoints, 1, K, rvec, tvec);
        /* computeError: projectPoints(float out), err = |ipt - proj|^2 in float */
        double R[9];
        orc_rodrigues_vec2mat(rvec, R);
        orc_project_points_f32(opoints, npoints, R, tvec, K, proj);
        float t = (float)(threshold*threshold);
        int goodCount = 0;
        for (int i = 0; i < npoints; i++) {
            float dx = img[i].x - proj[2*i], dy = img[i].y - proj[2*i+1];
            float err = dx*dx + dy*dy;
            int f = err <= t;
            mask[i] = (uint8_t)f; goodCount += f;
        }
        if (goodCount > (maxGoodCount > modelPoints-1 ? maxGoodCount : modelPoints-1)) {
            uint8_t* tmp = mask; mask = bestMask; bestMask = tmp;
            memcpy(bestModel, rvec, sizeof(double)*3); memcpy(bestModel+3, tvec, sizeof(double)*3);
            maxGoodCount = goodCount;
            niters = orc_ransac_update_num_iters(confidence, (double)(npoints - goodCount)/npoints, modelPoints, niters);
        }
    }
    if (maxGoodCount > 0) {
        /* refit on inliers with double points (values are the float-rounded ones) */
        int n1 = 0;
        for (int i = 0; i < npoints; i++) if (bestMask[i]) {
            od[3*n1] = opoints[3*i]; od[3*n1+1] = opoints[3*i+1]; od[3*n1+2] = opoints[3*i+2];
            id[2*n1] = img[i].x; id[2*n1+1] = img[i].y; n1++;
        }
        solve_pnp_epnp(od, id, n1, 0, K, rvec_out, tvec_out);
        int k = 0;
        for (int i = 0; i < npoints; i++) if (bestMask[i]) inliers[k++] = i;
        *n_inliers = k;
        result = 1;
    } else {
        /* RANSAC failed: OpenCV assigns its local rvec/tvec, whose buffers the callback shares and
         * last wrote in runKernel (= the last hypothesis evaluated), and releases inliers */
        memcpy(rvec_out, rvec, sizeof(double)*3); memcpy(tvec_out, tvec, sizeof(double)*3);
        result = 0;
    }
    free(mask); free(bestMask); free(proj); free(opoints); free(od); free(id);
    return result;
}
