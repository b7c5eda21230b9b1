/*
 * o_geom.c -- CPU ORACLE (test infrastructure): triangulation, projection, Rodrigues,
 * and the reference's extract_3Dpoints / reproject_errors.
 * Follows VOU:188-237, VOU:632-651, VO:631, VO:673 and restates [UPSTREAM] calib3d
 * triangulate.cpp (icvTriangulatePoints), fundam.cpp (convertPointsFromHomogeneous),
 * calibration.cpp (cvProjectPoints2 without distortion, cvRodrigues2).  SURVEY.md App. A.6.
 * PARITY UNPINNED vs OpenCV.
 */
#include "uvo_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

/* [UPSTREAM] triangulate.cpp: per point A(4x4) rows x*P[2]-P[0], y*P[2]-P[1] per view;
 * cv::SVD::compute(A, w, u, vt); X = vt row 3; stored as float (Point2f input). VO:631 */
void orc_triangulate_points(const double* P1, const double* P2, const orc_point2f* x1, const orc_point2f* x2,
                            int n, float* out)
{
    const double* P[2] = { P1, P2 };
    for (int i = 0; i < n; i++) {
        double A[16], w[4], u[16], vt[16];
        for (int j = 0; j < 2; j++) {
            double x = j == 0 ? x1[i].x : x2[i].x;
            double y = j == 0 ? x1[i].y : x2[i].y;
            for (int k = 0; k < 4; k++) {
                A[(j*2+0)*4 + k] = x * P[j][2*4 + k] - P[j][0*4 + k];
                A[(j*2+1)*4 + k] = y * P[j][2*4 + k] - P[j][1*4 + k];
            }
        }
        orc_svd(A, 4, 4, w, u, vt);
        out[0*n + i] = (float)vt[3*4 + 0];
        out[1*n + i] = (float)vt[3*4 + 1];
        out[2*n + i] = (float)vt[3*4 + 2];
        out[3*n + i] = (float)vt[3*4 + 3];
    }
}

/* [UPSTREAM] calibration.cpp cvRodrigues2, vector -> matrix (sin/cos via orc_sincos) */
void orc_rodrigues_vec2mat(const double* rv, double* R)
{
    double rx = rv[0], ry = rv[1], rz = rv[2];
    double theta = sqrt(rx*rx + ry*ry + rz*rz);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = 0; R[0] = R[4] = R[8] = 1;
        return;
    }
    double s, c; orc_sincos(theta, &s, &c);
    double c1 = 1. - c;
    double itheta = theta ? 1./theta : 0.;
    rx *= itheta; ry *= itheta; rz *= itheta;
    double rrt[9] = { rx*rx, rx*ry, rx*rz, rx*ry, ry*ry, ry*rz, rx*rz, ry*rz, rz*rz };
    double r_x[9] = { 0, -rz, ry, rz, 0, -rx, -ry, rx, 0 };
    double eye[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (int k = 0; k < 9; k++) R[k] = c*eye[k] + c1*rrt[k] + s*r_x[k];
}

/* [UPSTREAM] calibration.cpp cvRodrigues2, matrix -> vector (acos via orc_acos) */
void orc_rodrigues_mat2vec(const double* Rin, double* rv)
{
    double R[9], w[3], u[9], vt[9];
    for (int i = 0; i < 9; i++) {
        if (!(Rin[i] >= -100 && Rin[i] < 100)) { rv[0] = rv[1] = rv[2] = 0; return; }  /* checkRange */
    }
    orc_svd(Rin, 3, 3, w, u, vt);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double s = 0; for (int k = 0; k < 3; k++) s += u[i*3+k]*vt[k*3+j];
        R[i*3+j] = s;
    }
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx*rx + ry*ry + rz*rz)*0.25);
    double c = (R[0] + R[4] + R[8] - 1)*0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = orc_acos(c);
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

/* [UPSTREAM] calibration.cpp cvProjectPoints2, k = 0, no tilt: x = (R X + t).x * (1/z); u = x*fx + cx */
static inline void project_one(double X, double Y, double Z, const double* R, const double* t, const double* K,
                               double* u, double* v)
{
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double x = R[0]*X + R[1]*Y + R[2]*Z + t[0];
    double y = R[3]*X + R[4]*Y + R[5]*Z + t[1];
    double z = R[6]*X + R[7]*Y + R[8]*Z + t[2];
    z = z ? 1./z : 1;
    x *= z; y *= z;
    *u = x*fx + cx;
    *v = y*fy + cy;
}
void orc_project_points_f64(const double* X, int n, const double* R, const double* t, const double* K, double* out)
{
    for (int i = 0; i < n; i++) project_one(X[3*i], X[3*i+1], X[3*i+2], R, t, K, &out[2*i], &out[2*i+1]);
}
void orc_project_points_f32(const float* X, int n, const double* R, const double* t, const double* K, float* out)
{
    for (int i = 0; i < n; i++) {
        double u, v;
        project_one(X[3*i], X[3*i+1], X[3*i+2], R, t, K, &u, &v);
        out[2*i] = (float)u; out[2*i+1] = (float)v;
    }
}

/* VOU:632-651 reproject_errors */
void orc_reproject_errors(const double* world, int n, const double* R, const double* t, const double* K,
                          const orc_point2f* img, double* err)
{
    for (int i = 0; i < n; i++) {
        double u, v;
        project_one(world[3*i], world[3*i+1], world[3*i+2], R, t, K, &u, &v);
        double dx = img[i].x - u;
        double dy = img[i].y - v;
        err[i] = sqrt(dx * dx + dy * dy);
    }
}

/* VOU:188-237 extract_3Dpoints (+ [UPSTREAM] convertPointsFromHomogeneous float 4->3) */
int orc_extract_3Dpoints(const orc_point2f* k1, const orc_point2f* k2, int n,
                         const double* R1, const double* t1, const double* R2, const double* t2,
                         const double* K1, const double* K2, const float* points4D,
                         int MIN_NUM_3DPOINTS, double REPROJECTION_TOLERANCE, double* pts, int* idx)
{
    if (n <= 0) return 0;
    double* cam1 = (double*)malloc(sizeof(double) * 3 * n);
    for (int i = 0; i < n; i++) {             /* VOU:191-201 */
        float w = points4D[3*n + i];
        float scale = w != 0.f ? 1.f/w : 1.f;
        cam1[3*i]   = (double)(points4D[0*n + i]*scale);
        cam1[3*i+1] = (double)(points4D[1*n + i]*scale);
        cam1[3*i+2] = (double)(points4D[2*n + i]*scale);
    }
    int ngood = 0, G = 0;
    int* good_idx = (int*)malloc(sizeof(int) * n);
    double* good_z = (double*)malloc(sizeof(double) * n);
    if (n >= MIN_NUM_3DPOINTS) {              /* VOU:203-220 */
        double* e1 = (double*)malloc(sizeof(double) * n);
        double* e2 = (double*)malloc(sizeof(double) * n);
        orc_reproject_errors(cam1, n, R1, t1, K1, k1, e1);
        orc_reproject_errors(cam1, n, R2, t2, K2, k2, e2);
        for (int i = 0; i < n; i++) {
            double mean = (e1[i] + e2[i]) / 2.0;
            if ((mean < REPROJECTION_TOLERANCE) && (cam1[3*i+2] > 0)) { good_idx[ngood] = i; good_z[ngood] = cam1[3*i+2]; ngood++; }
        }
        free(e1); free(e2);
    }
    if (ngood >= MIN_NUM_3DPOINTS && ngood > 0) {          /* VOU:222-236 */
        double mv[2];
        orc_compute_mean_and_variance(good_z, ngood, mv);
        for (int i = 0; i < ngood; i++) {
            double z = good_z[i];
            if ((z <= mv[0] + 3.0*sqrt(mv[1])) && (z >= mv[0] - 3.0*sqrt(mv[1]))) {
                int src = good_idx[i];
                idx[G] = src;
                pts[3*G] = cam1[3*src]; pts[3*G+1] = cam1[3*src+1]; pts[3*G+2] = cam1[3*src+2];
                G++;
            }
        }
    }
    free(cam1); free(good_idx); free(good_z);
    return G;
}
