// uvo_mono.h -- small fp64 routines of the mono path shared by host and device code of libuvo_hip:
// symmetric Jacobi eigen-solver (OpenCV lapack.cpp JacobiImpl_), the normalised-DLT homography
// kernel and its reprojection error (fundam.cpp HomographyEstimatorCallback), decomposeEssentialMat
// (five-point.cpp), Sampson error.  Same operation order as OpenCV 4.5; std::hypot -> det_hypot.
// Reference call sites: findHomography VO_utility.cpp:152, findEssentialMat / recoverPose
// VO_utility.cpp:147-149.
#pragma once
#include "uvo_linalg.h"

namespace uvo {

// JacobiImpl_<double>(A, n, W, V): eigenvalues descending in W, eigenvectors in the rows of V.
// A (n x n, tight) is destroyed.  indR/indC: 2n ints of scratch.
template <class A, class IA>
__host__ __device__ void jacobi_eigen(A Am, int n, A W, A V, IA indR, IA indC)
{
    const double eps = DBL_EPSILON;
    int i, j, k, m;
    double mv = 0;
    for (i = 0; i < n; i++) { for (j = 0; j < n; j++) V[i*n + j] = 0; V[i*n + i] = 1; }
    int iters, maxIters = n*n*30;
    for (k = 0; k < n; k++) {
        W[k] = Am[(n + 1)*k];
        if (k < n - 1) {
            for (m = k+1, mv = fabs(Am[n*k + m]), i = k+2; i < n; i++) { double val = fabs(Am[n*k+i]); if (mv < val) mv = val, m = i; }
            indR[k] = m;
        }
        if (k > 0) {
            for (m = 0, mv = fabs(Am[k]), i = 1; i < k; i++) { double val = fabs(Am[n*i+k]); if (mv < val) mv = val, m = i; }
            indC[k] = m;
        }
    }
    if (n > 1) for (iters = 0; iters < maxIters; iters++) {
        for (k = 0, mv = fabs(Am[indR[0]]), i = 1; i < n-1; i++) { double val = fabs(Am[n*i + indR[i]]); if (mv < val) mv = val, k = i; }
        int l = indR[k];
        for (i = 1; i < n; i++) { double val = fabs(Am[n*indC[i] + i]); if (mv < val) mv = val, k = indC[i], l = i; }
        double p = Am[n*k + l];
        if (fabs(p) <= eps) break;
        double y = (W[l] - W[k])*0.5;
        double t = fabs(y) + det_hypot(p, y);
        double s = det_hypot(p, t);
        double c = t/s;
        s = p/s; t = (p/t)*p;
        if (y < 0) s = -s, t = -t;
        Am[n*k + l] = 0;
        W[k] -= t;
        W[l] += t;
        double a0, b0;
#define UVO_ROT(v0, v1) a0 = v0, b0 = v1, v0 = a0*c - b0*s, v1 = a0*s + b0*c
        for (i = 0; i < k; i++) UVO_ROT(Am[n*i+k], Am[n*i+l]);
        for (i = k+1; i < l; i++) UVO_ROT(Am[n*k+i], Am[n*i+l]);
        for (i = l+1; i < n; i++) UVO_ROT(Am[n*k+i], Am[n*l+i]);
        for (i = 0; i < n; i++) UVO_ROT(V[n*k+i], V[n*l+i]);
#undef UVO_ROT
        for (j = 0; j < 2; j++) {
            int idx = j == 0 ? k : l;
            if (idx < n - 1) {
                for (m = idx+1, mv = fabs(Am[n*idx + m]), i = idx+2; i < n; i++) { double val = fabs(Am[n*idx+i]); if (mv < val) mv = val, m = i; }
                indR[idx] = m;
            }
            if (idx > 0) {
                for (m = 0, mv = fabs(Am[idx]), i = 1; i < idx; i++) { double val = fabs(Am[n*i+idx]); if (mv < val) mv = val, m = i; }
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

// HomographyEstimatorCallback::runKernel: M -> m, `count` Point2f pairs (interleaved x,y floats).
// scratch: LtL(81) W(9) V(81) doubles + 18 ints.  Returns 0 when the points are degenerate.
template <class A, class IA>
__host__ __device__ int homography_kernel(const float* M, const float* m, int count, double* Hout, A LtL, A W, A V, IA ind)
{
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
    const double invHnorm[9] = { 1./smx, 0, cmx, 0, 1./smy, cmy, 0, 0, 1 };
    const double Hnorm2[9] = { sMx, 0, -cMx*sMx, 0, sMy, -cMy*sMy, 0, 0, 1 };
    for (i = 0; i < 81; i++) LtL[i] = 0;
    for (i = 0; i < count; i++) {
        double x = (m[2*i] - cmx)*smx, y = (m[2*i+1] - cmy)*smy;
        double X = (M[2*i] - cMx)*sMx, Y = (M[2*i+1] - cMy)*sMy;
        const double Lx[9] = { X, Y, 1, 0, 0, 0, -x*X, -x*Y, -x };
        const double Ly[9] = { 0, 0, 0, X, Y, 1, -y*X, -y*Y, -y };
        for (int j = 0; j < 9; j++) for (int k = j; k < 9; k++) LtL[j*9 + k] += Lx[j]*Lx[k] + Ly[j]*Ly[k];
    }
    for (int j = 0; j < 9; j++) for (int k = 0; k < j; k++) LtL[j*9 + k] = LtL[k*9 + j];
    jacobi_eigen(LtL, 9, W, V, ind, ind + 9);
    double H0[9], Ht[9], H1[9];
    for (int k = 0; k < 9; k++) H0[k] = V[8*9 + k];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
        Ht[r*3 + c] = invHnorm[r*3]*H0[c] + invHnorm[r*3+1]*H0[3 + c] + invHnorm[r*3+2]*H0[6 + c];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
        H1[r*3 + c] = Ht[r*3]*Hnorm2[c] + Ht[r*3+1]*Hnorm2[3 + c] + Ht[r*3+2]*Hnorm2[6 + c];
    double sc = 1./H1[8];
    for (int k = 0; k < 9; k++) Hout[k] = H1[k]*sc;
    return 1;
}

// HomographyEstimatorCallback::computeError for one correspondence (float arithmetic)
__host__ __device__ __forceinline__ float homography_error1(const float* Hf, float Mx, float My, float mx, float my)
{
    float ww = 1.f/(Hf[6]*Mx + Hf[7]*My + 1.f);
    float dx = (Hf[0]*Mx + Hf[1]*My + Hf[2])*ww - mx;
    float dy = (Hf[3]*Mx + Hf[4]*My + Hf[5])*ww - my;
    return dx*dx + dy*dy;
}

// EMEstimatorCallback::computeError for one correspondence (Sampson distance, double -> float)
__host__ __device__ __forceinline__ float sampson_error1(const double* E, double x1, double y1, double x2, double y2)
{
    double Ex1_0 = E[0]*x1 + E[1]*y1 + E[2]*1., Ex1_1 = E[3]*x1 + E[4]*y1 + E[5]*1., Ex1_2 = E[6]*x1 + E[7]*y1 + E[8]*1.;
    double Etx2_0 = E[0]*x2 + E[3]*y2 + E[6]*1., Etx2_1 = E[1]*x2 + E[4]*y2 + E[7]*1.;
    double x2tEx1 = x2*Ex1_0 + y2*Ex1_1 + 1.*Ex1_2;
    double a = Ex1_0*Ex1_0, b = Ex1_1*Ex1_1, c = Etx2_0*Etx2_0, d = Etx2_1*Etx2_1;
    return (float)(x2tEx1*x2tEx1 / (a + b + c + d));
}

__host__ __device__ __forceinline__ double det3(const double* m)
{
    return m[0]*(m[4]*m[8] - m[5]*m[7]) - m[1]*(m[3]*m[8] - m[5]*m[6]) + m[2]*(m[3]*m[7] - m[4]*m[6]);
}
__host__ __device__ inline void mat3_mul(const double* a, const double* b, double* out)
{
    double r[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += a[i*3+k]*b[k*3+j];
        r[i*3+j] = s;
    }
    for (int i = 0; i < 9; i++) out[i] = r[i];
}

}  // namespace uvo
