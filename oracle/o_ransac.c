/*
 * o_ransac.c -- CPU ORACLE (test infrastructure): OpenCV's robust point-set registrators and the two
 * estimators the mono node uses through them: cv::findEssentialMat (VOU:147) and the robust loop of
 * cv::findHomography (VOU:152).  Restates [UPSTREAM] calib3d/src/ptsetreg.cpp
 * (RANSACPointSetRegistrator::run/getSubset/findInliers, LMeDSPointSetRegistrator::run),
 * five-point.cpp (findEssentialMat, recoverPose, decomposeEssentialMat).  SURVEY.md App. A.3-A.4.
 * PARITY UNPINNED vs OpenCV.
 */
#include "uvo_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

int orc_five_point(const double* q1, const double* q2, double* models);
void orc_sampson_error(const double* p1, const double* p2, int n, const double* E, float* err);
int orc_homography_kernel(const float* M, const float* m, int count, double* H);
void orc_homography_error(const float* M, const float* m, int count, const double* H, float* err);
int orc_homography_check_subset(const float* ms1, const float* ms2, int count);

/* callback table: E works on Point2d (2 doubles), H on Point2f (2 floats) */
typedef struct {
    int model_points, elem_size /* bytes of one point */, max_models;
    int (*run_kernel)(const void* ms1, const void* ms2, int count, double* models);
    void (*compute_error)(const void* m1, const void* m2, int count, const double* model, float* err);
    int (*check_subset)(const void* ms1, const void* ms2, int count);
} reg_cb;

static int e_kernel(const void* a, const void* b, int count, double* models) { (void)count; return orc_five_point((const double*)a, (const double*)b, models); }
static void e_error(const void* a, const void* b, int count, const double* model, float* err) { orc_sampson_error((const double*)a, (const double*)b, count, model, err); }
static int h_kernel(const void* a, const void* b, int count, double* models) { return orc_homography_kernel((const float*)a, (const float*)b, count, models); }
static void h_error(const void* a, const void* b, int count, const double* model, float* err) { orc_homography_error((const float*)a, (const float*)b, count, model, err); }
static int h_check(const void* a, const void* b, int count) { return orc_homography_check_subset((const float*)a, (const float*)b, count); }
static const reg_cb CB_E = { 5, 2 * sizeof(double), 10, e_kernel, e_error, NULL };
static const reg_cb CB_H = { 4, 2 * sizeof(float), 1, h_kernel, h_error, h_check };

/* [UPSTREAM] getSubset: draw modelPoints distinct indices (redraw on duplicate), then checkSubset */
static int get_subset(const reg_cb* cb, const char* m1, const char* m2, int count, char* ms1, char* ms2, orc_rng* rng, int maxAttempts)
{
    int idx[8];
    for (int iters = 0; iters < maxAttempts; ++iters) {
        int i;
        for (i = 0; i < cb->model_points; ++i) {
            int idx_i;
            for (;;) {
                idx_i = orc_rng_uniform(rng, 0, count);
                int dup = 0;
                for (int q = 0; q < i; q++) if (idx[q] == idx_i) dup = 1;
                if (!dup) break;
            }
            idx[i] = idx_i;
            memcpy(ms1 + (size_t)i * cb->elem_size, m1 + (size_t)idx_i * cb->elem_size, cb->elem_size);
            memcpy(ms2 + (size_t)i * cb->elem_size, m2 + (size_t)idx_i * cb->elem_size, cb->elem_size);
        }
        if (!cb->check_subset || cb->check_subset(ms1, ms2, i)) return 1;
    }
    return 0;
}

static int find_inliers(const reg_cb* cb, const void* m1, const void* m2, int count, const double* model, float* err, uint8_t* mask, double thresh)
{
    cb->compute_error(m1, m2, count, model, err);
    float t = (float)(thresh * thresh);
    int nz = 0;
    for (int i = 0; i < count; i++) { int f = err[i] <= t; mask[i] = (uint8_t)f; nz += f; }
    return nz;
}

/* [UPSTREAM] RANSACPointSetRegistrator::run.  model: 9 doubles.  Returns 1 on success. */
static int ransac_run(const reg_cb* cb, const void* m1, const void* m2, int count, double threshold, double confidence, int maxIters,
                      double* model_out, uint8_t* mask_out)
{
    int modelPoints = cb->model_points, niters = maxIters > 1 ? maxIters : 1, maxGoodCount = 0;
    if (count < modelPoints) return 0;
    double models[10 * 9], bestModel[9];
    if (count == modelPoints) {
        if (cb->run_kernel(m1, m2, count, models) <= 0) return 0;
        memcpy(model_out, models, sizeof(double) * 9);
        memset(mask_out, 1, count);
        return 1;
    }
    float* err = (float*)malloc(sizeof(float) * count);
    uint8_t* mask = (uint8_t*)malloc(count);
    char ms1[8 * 16], ms2[8 * 16];
    orc_rng rng; orc_rng_init(&rng, (uint64_t)-1);
    for (int iter = 0; iter < niters; iter++) {
        if (!get_subset(cb, (const char*)m1, (const char*)m2, count, ms1, ms2, &rng, 10000)) {
            if (iter == 0) { free(err); free(mask); return 0; }
            break;
        }
        int nmodels = cb->run_kernel(ms1, ms2, modelPoints, models);
        if (nmodels <= 0) continue;
        for (int i = 0; i < nmodels; i++) {
            int goodCount = find_inliers(cb, m1, m2, count, models + i * 9, err, mask, threshold);
            if (goodCount > (maxGoodCount > modelPoints - 1 ? maxGoodCount : modelPoints - 1)) {
                memcpy(mask_out, mask, count);
                memcpy(bestModel, models + i * 9, sizeof(bestModel));
                maxGoodCount = goodCount;
                niters = orc_ransac_update_num_iters(confidence, (double)(count - goodCount) / count, modelPoints, niters);
            }
        }
    }
    free(err); free(mask);
    if (maxGoodCount > 0) { memcpy(model_out, bestModel, sizeof(bestModel)); return 1; }
    return 0;
}

static int cmp_float(const void* a, const void* b) { float x = *(const float*)a, y = *(const float*)b; return (x > y) - (x < y); }

/* [UPSTREAM] LMeDSPointSetRegistrator::run (median = sorted middle, mean of the two middles for even counts) */
static int lmeds_run(const reg_cb* cb, const void* m1, const void* m2, int count, double confidence, int maxIters,
                     double* model_out, uint8_t* mask_out)
{
    const double outlierRatio = 0.45;
    int modelPoints = cb->model_points;
    if (count < modelPoints) return 0;
    double models[10 * 9], bestModel[9], minMedian = DBL_MAX;
    if (count == modelPoints) {
        if (cb->run_kernel(m1, m2, count, models) <= 0) return 0;
        memcpy(model_out, models, sizeof(double) * 9);
        memset(mask_out, 1, count);
        return 1;
    }
    int niters = orc_ransac_update_num_iters(confidence, outlierRatio, modelPoints, maxIters);
    niters = niters > 3 ? niters : 3;
    float* err = (float*)malloc(sizeof(float) * count);
    char ms1[8 * 16], ms2[8 * 16];
    orc_rng rng; orc_rng_init(&rng, (uint64_t)-1);
    for (int iter = 0; iter < niters; iter++) {
        if (!get_subset(cb, (const char*)m1, (const char*)m2, count, ms1, ms2, &rng, 1000)) {
            if (iter == 0) { free(err); return 0; }
            break;
        }
        int nmodels = cb->run_kernel(ms1, ms2, modelPoints, models);
        if (nmodels <= 0) continue;
        for (int i = 0; i < nmodels; i++) {
            cb->compute_error(m1, m2, count, models + i * 9, err);
            qsort(err, count, sizeof(float), cmp_float);
            double median = count % 2 != 0 ? err[count/2] : (err[count/2 - 1] + err[count/2]) * 0.5;
            if (median < minMedian) { minMedian = median; memcpy(bestModel, models + i * 9, sizeof(bestModel)); }
        }
    }
    int result = 0;
    if (minMedian < DBL_MAX) {
        double sigma = 2.5 * 1.4826 * (1 + 5. / (count - modelPoints)) * sqrt(minMedian);
        sigma = sigma > 0.001 ? sigma : 0.001;
        int good = find_inliers(cb, m1, m2, count, bestModel, err, mask_out, sigma);
        memcpy(model_out, bestModel, sizeof(bestModel));
        result = good >= modelPoints;
    }
    free(err);
    return result;
}

/* cv::findEssentialMat(points1, points2, K, method, prob, threshold, maxIters, mask)  (VOU:147).
 * Returns 1 and E (9 doubles) on success; 0 means OpenCV would return an empty matrix. */
int orc_find_essential_mat(const orc_point2f* p1, const orc_point2f* p2, int n, const double* K, int method,
                           double prob, double threshold, int maxIters, double* E, uint8_t* mask)
{
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double* q1 = (double*)malloc(sizeof(double) * 2 * (n + 1));
    double* q2 = (double*)malloc(sizeof(double) * 2 * (n + 1));
    /* (points.col(0) - cx) / fx as the MatExpr evaluates it: col * (1/fx) + (-cx * (1/fx)) */
    double ax = 1. / fx, bx = -cx * ax, ay = 1. / fy, by = -cy * ay;
    for (int i = 0; i < n; i++) {
        q1[2*i] = p1[i].x * ax + bx; q1[2*i+1] = p1[i].y * ay + by;
        q2[2*i] = p2[i].x * ax + bx; q2[2*i+1] = p2[i].y * ay + by;
    }
    threshold /= (fx + fy) / 2;
    memset(mask, 0, n);
    int ok;
    if (method == 8) ok = ransac_run(&CB_E, q1, q2, n, threshold, prob, maxIters, E, mask);
    else             ok = lmeds_run(&CB_E, q1, q2, n, prob, maxIters, E, mask);
    free(q1); free(q2);
    return ok;
}

/* robust part of cv::findHomography: returns 1 with H (pre-refinement) and the loop's mask */
int orc_homography_robust(const float* src, const float* dst, int n, int method, double threshold, int maxIters, double confidence,
                          double* H, uint8_t* mask)
{
    memset(mask, 0, n);
    if (method == 8) return ransac_run(&CB_H, src, dst, n, threshold, confidence, maxIters, H, mask);
    return lmeds_run(&CB_H, src, dst, n, confidence, maxIters, H, mask);
}

/* [UPSTREAM] five-point.cpp decomposeEssentialMat */
static double det3(const double* m)
{
    return m[0]*(m[4]*m[8] - m[5]*m[7]) - m[1]*(m[3]*m[8] - m[5]*m[6]) + m[2]*(m[3]*m[7] - m[4]*m[6]);
}
static void mat3_mul(const double* a, const double* b, double* out)
{
    double r[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += a[i*3+k]*b[k*3+j];
        r[i*3+j] = s;
    }
    memcpy(out, r, sizeof(r));
}
void orc_decompose_essential_mat(const double* E, double* R1, double* R2, double* t)
{
    double D[3], U[9], Vt[9];
    orc_svd(E, 3, 3, D, U, Vt);
    if (det3(U) < 0) for (int i = 0; i < 9; i++) U[i] *= -1.;
    if (det3(Vt) < 0) for (int i = 0; i < 9; i++) Vt[i] *= -1.;
    const double W[9] = { 0, 1, 0, -1, 0, 0, 0, 0, 1 }, Wt[9] = { 0, -1, 0, 1, 0, 0, 0, 0, 1 };
    double UW[9];
    mat3_mul(U, W, UW); mat3_mul(UW, Vt, R1);
    mat3_mul(U, Wt, UW); mat3_mul(UW, Vt, R2);
    t[0] = U[2]; t[1] = U[5]; t[2] = U[8];
}

/* triangulatePoints with double points (recoverPose converts its inputs to CV_64F): 4 x n doubles */
static void triangulate_f64(const double* P1, const double* P2, const double* x1, const double* x2, int n, double* out)
{
    const double* P[2] = { P1, P2 };
    for (int i = 0; i < n; i++) {
        double A[16], w[4], u[16], vt[16];
        for (int j = 0; j < 2; j++) {
            double x = j == 0 ? x1[2*i] : x2[2*i], y = j == 0 ? x1[2*i+1] : x2[2*i+1];
            for (int k = 0; k < 4; k++) {
                A[(j*2+0)*4 + k] = x * P[j][2*4 + k] - P[j][0*4 + k];
                A[(j*2+1)*4 + k] = y * P[j][2*4 + k] - P[j][1*4 + k];
            }
        }
        orc_svd(A, 4, 4, w, u, vt);
        for (int k = 0; k < 4; k++) out[k*n + i] = vt[3*4 + k];
    }
}

/* cv::recoverPose(E, points1, points2, K, R, t, mask) with distanceThresh = 50  (VOU:149).
 * mask is in/out (AND-ed with the cheirality test of the winning candidate); returns the inlier count. */
int orc_recover_pose(const double* E, const orc_point2f* p1, const orc_point2f* p2, int n, const double* K,
                     double* R, double* t, uint8_t* mask)
{
    const double distanceThresh = 50;
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double ax = 1. / fx, bx = -cx * ax, ay = 1. / fy, by = -cy * ay;
    double* q1 = (double*)malloc(sizeof(double) * 2 * (n + 1));
    double* q2 = (double*)malloc(sizeof(double) * 2 * (n + 1));
    for (int i = 0; i < n; i++) {
        q1[2*i] = p1[i].x * ax + bx; q1[2*i+1] = p1[i].y * ay + by;
        q2[2*i] = p2[i].x * ax + bx; q2[2*i+1] = p2[i].y * ay + by;
    }
    double R1[9], R2[9], tt[3];
    orc_decompose_essential_mat(E, R1, R2, tt);
    const double P0[12] = { 1,0,0,0, 0,1,0,0, 0,0,1,0 };
    const double* Rc[4] = { R1, R2, R1, R2 };
    const double sg[4] = { 1, 1, -1, -1 };
    double* Q = (double*)malloc(sizeof(double) * 4 * (n + 1));
    uint8_t* masks = (uint8_t*)malloc((size_t)4 * (n + 1));
    int good[4];
    for (int c = 0; c < 4; c++) {
        double P[12];
        for (int i = 0; i < 3; i++) { P[i*4] = Rc[c][i*3]; P[i*4+1] = Rc[c][i*3+1]; P[i*4+2] = Rc[c][i*3+2]; P[i*4+3] = sg[c] * tt[i]; }
        triangulate_f64(P0, P, q1, q2, n, Q);
        uint8_t* mk = masks + (size_t)c * n;
        good[c] = 0;
        for (int i = 0; i < n; i++) {
            double X = Q[i], Y = Q[n + i], Z = Q[2*n + i], Wv = Q[3*n + i];
            int m = Z * Wv > 0;
            X /= Wv; Y /= Wv; Z /= Wv; Wv /= Wv;
            m = (Z < distanceThresh) & m;
            double z2 = P[8]*X + P[9]*Y + P[10]*Z + P[11]*Wv;         /* Q = P * Q, row 2 */
            m = (z2 > 0) & m;
            m = (z2 < distanceThresh) & m;
            m = m && mask[i];
            mk[i] = (uint8_t)(m ? 1 : 0);
            good[c] += m ? 1 : 0;
        }
    }
    int best;
    if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3]) best = 0;
    else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3]) best = 1;
    else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3]) best = 2;
    else best = 3;
    memcpy(R, Rc[best], sizeof(double) * 9);
    for (int i = 0; i < 3; i++) t[i] = sg[best] * tt[i];
    memcpy(mask, masks + (size_t)best * n, n);
    int g = good[best];
    free(q1); free(q2); free(Q); free(masks);
    return g;
}
