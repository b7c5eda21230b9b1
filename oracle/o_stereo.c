/*
 * o_stereo.c -- CPU ORACLE (test infrastructure): ROS-free restatement of the stereo
 * per-frame-pair loop visual_odometry_node::stereo_VO (VO:406-741): init phase
 * VO:474-520, main loop body VO:531-739, output VO:148-159, state carry VO:723-733.
 * get_image preprocessing (VO:482-483, 542-543) is outside this path: the caller passes
 * the gray images and the *new* camera matrices.
 * PARITY UNPINNED vs OpenCV.
 */
#include "uvo_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct orc_stereo {
    orc_vo_params p;
    double K_left[9], K_right[9], R_right[9], t_right[3];
    double P_eye_left[12], P_right[12];          /* VO:460, 462 */
    int cap;
    int vo_initialized;
    /* VO:468: results_match_prev lives across init attempts and match_features appends */
    orc_dmatch* results_match_prev; int n_match_prev;
    /* carried state VO:515-520 / VO:727-733 */
    orc_keypoint *prevL_kps_as, *prevR_kps_as; int n_prevL_as, n_prevR_as;
    float* prevL_desc_as; int n_prevL_desc_as;
    double t_prev_curr[3], rvec[3], tvec[3];
    /* last-step intermediates for parity tests */
    orc_keypoint *kpsL, *kpsR; float *descL, *descR; int nL, nR;
    orc_dmatch *m_curr, *m_pc; int n_m_curr, n_m_pc;
    float* points4D; int nT;
    double* good_pts; int* good_idx; int G;
    int* inliers; int n_inl;
    int detector;              /* the reference's global FEATURE_DETECTOR (VOH:25): 0 "SURF", 1 "SIFT", 2 "AKAZE", 3 "ORB" */
    int orb_pattern[1024];     /* "ORB": the sampling table, OpenCV's bit_pattern_31_ layout (o_orb.c: an input) */
};
void orc_stereo_use_sift(orc_stereo* s, int on) { s->detector = on ? 1 : 0; }
/* detector 0..3 as above; pattern: 1024 ints for "ORB" (ignored otherwise) */
void orc_stereo_use_detector(orc_stereo* s, int detector, const int* pattern)
{
    s->detector = detector;
    if (detector == 3 && pattern) memcpy(s->orb_pattern, pattern, sizeof(s->orb_pattern));
}

/* VOU:9-15 compute_projection_matrix: K * [R|t] ([UPSTREAM] gemm small-matrix case, len 3) */
static void compute_projection_matrix(const double* R, const double* t, const double* K, double* P)
{
    double Rt[12];
    for (int i = 0; i < 3; i++) { Rt[i*4] = R[i*3]; Rt[i*4+1] = R[i*3+1]; Rt[i*4+2] = R[i*3+2]; Rt[i*4+3] = t[i]; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++)
        P[i*4 + j] = K[i*3]*Rt[j] + K[i*3+1]*Rt[4 + j] + K[i*3+2]*Rt[8 + j];
}

orc_stereo* orc_stereo_create(const orc_vo_params* p, const double* K_left, const double* K_right,
                              const double* R_right, const double* t_right, int max_kpts)
{
    orc_stereo* s = (orc_stereo*)calloc(1, sizeof(orc_stereo));
    s->p = *p; s->cap = max_kpts;
    memcpy(s->K_left, K_left, sizeof(double)*9); memcpy(s->K_right, K_right, sizeof(double)*9);
    memcpy(s->R_right, R_right, sizeof(double)*9); memcpy(s->t_right, t_right, sizeof(double)*3);
    double I[9] = {1,0,0,0,1,0,0,0,1}, z[3] = {0,0,0};
    compute_projection_matrix(I, z, K_left, s->P_eye_left);          /* VO:460 */
    compute_projection_matrix(R_right, t_right, K_right, s->P_right);/* VO:462 */
    size_t c = (size_t)max_kpts;
    s->results_match_prev = (orc_dmatch*)malloc(sizeof(orc_dmatch) * c * 4);
    s->prevL_kps_as = (orc_keypoint*)malloc(sizeof(orc_keypoint) * c * 4);
    s->prevR_kps_as = (orc_keypoint*)malloc(sizeof(orc_keypoint) * c * 4);
    s->prevL_desc_as = (float*)malloc(sizeof(float) * 128 * c * 4);
    s->kpsL = (orc_keypoint*)malloc(sizeof(orc_keypoint) * c); s->kpsR = (orc_keypoint*)malloc(sizeof(orc_keypoint) * c);
    s->descL = (float*)malloc(sizeof(float) * 128 * c); s->descR = (float*)malloc(sizeof(float) * 128 * c);
    s->m_curr = (orc_dmatch*)malloc(sizeof(orc_dmatch) * c); s->m_pc = (orc_dmatch*)malloc(sizeof(orc_dmatch) * c * 4);
    s->points4D = (float*)malloc(sizeof(float) * 4 * c * 4);
    s->good_pts = (double*)malloc(sizeof(double) * 3 * c * 4); s->good_idx = (int*)malloc(sizeof(int) * c * 4);
    s->inliers = (int*)malloc(sizeof(int) * c * 4);
    return s;
}
void orc_stereo_destroy(orc_stereo* s)
{
    if (!s) return;
    free(s->results_match_prev); free(s->prevL_kps_as); free(s->prevR_kps_as); free(s->prevL_desc_as);
    free(s->kpsL); free(s->kpsR); free(s->descL); free(s->descR); free(s->m_curr); free(s->m_pc);
    free(s->points4D); free(s->good_pts); free(s->good_idx); free(s->inliers); free(s);
}

static int detect(orc_stereo* s, const uint8_t* img, int w, int h, int stride, orc_keypoint* kps, float* desc)
{
    if (s->detector == 2) {     /* VOU:93-98: AKAZE::create()->detectAndCompute; rows of 61 bytes */
        int n = orc_akaze_detect_and_compute(img, w, h, stride, kps, (uint8_t*)desc, s->cap);
        return n < 0 ? s->cap : n;
    }
    if (s->detector == 3) {     /* VOU:100-105: ORB::create(10000, 1.2, 8, 31, 0, 2, ORB::HARRIS_SCORE, 31, 10)->detectAndCompute; rows of 32 bytes */
        int n = orc_orb_detect_and_compute(img, w, h, stride, 10000, 1.2f, 8, 31, 0, 31, 10, s->orb_pattern, kps, (uint8_t*)desc, s->cap);
        return n < 0 ? s->cap : n;
    }
    if (s->detector == 1) {          /* VOU:107-112: SIFT::create(10000, 3, 0.03, 10, 1.6)->detectAndCompute */
        int n = orc_sift_detect_and_compute(img, w, h, stride, 10000, 3, 0.03, 10, 1.6, kps, desc, s->cap);
        return n < 0 ? s->cap : n;
    }
    /* VOU:114-119 */
    orc_surf_params sp = { (double)s->p.SURF_MIN_HESSIAN, s->p.SURF_OCTAVES_NUMBER, s->p.SURF_OCTAVES_LAYERS,
                           s->p.SURF_EXTENDED, s->p.SURF_UPRIGHT };
    int n = orc_surf_detect_and_compute(img, w, h, stride, &sp, kps, desc, s->cap);
    return n < 0 ? s->cap : n;      /* capacity overflow: truncated to cap (flagged by the caller's sizing) */
}

/* VOU:683-696 / VOU:704-717 index gathers with the reference's bounds checks */
static int gather_keypoints(const orc_keypoint* src, int nsrc, const int* idx, int nidx, orc_keypoint* dst)
{
    int k = 0;
    for (int i = 0; i < nidx; i++) if (idx[i] >= 0 && idx[i] < nsrc) dst[k++] = src[idx[i]];
    return k;
}
static void gather_descriptors(const float* src_, int nsrc, const int* idx, int nidx, float* dst_, int row_bytes)
{
    const uint8_t* src = (const uint8_t*)src_; uint8_t* dst = (uint8_t*)dst_;   /* descriptors.row(idx).copyTo(...): CV_32F or CV_8U rows alike */
    for (int i = 0; i < nidx; i++) {
        if (idx[i] >= 0 && idx[i] < nsrc) memcpy(dst + (size_t)i*row_bytes, src + (size_t)idx[i]*row_bytes, (size_t)row_bytes);
        else memset(dst + (size_t)i*row_bytes, 0, (size_t)row_bytes);   /* reference leaves the row uninitialised */
    }
}
/* descriptorSize() in elements: SURF 64 (128 with `extended`), SIFT 128 floats; AKAZE 61, ORB 32 bytes */
static int desc_dim(const orc_stereo* s) { return s->detector == 2 ? 61 : s->detector == 3 ? 32 : (s->detector == 1 || s->p.SURF_EXTENDED ? 128 : 64); }
static int desc_row_bytes(const orc_stereo* s) { return s->detector >= 2 ? desc_dim(s) : desc_dim(s) * (int)sizeof(float); }
/* match_features (VOU:515-543): BFMatcher(NORM_HAMMING) for "AKAZE" / "ORB", BFMatcher(NORM_L2) for "SURF" / "SIFT", then the ratio test */
static void match(const orc_stereo* s, const float* d1, int n1, const float* d2, int n2, orc_dmatch* out, int cap, int* m)
{
    if (s->detector >= 2) orc_match_knn2_ratio_hamming((const uint8_t*)d1, n1, (const uint8_t*)d2, n2, desc_dim(s), (float)s->p.LOWE_RATIO_THRESHOLD, out, cap, m);
    else orc_match_knn2_ratio(d1, n1, d2, n2, desc_dim(s), (float)s->p.LOWE_RATIO_THRESHOLD, out, cap, m);
}

int orc_stereo_step(orc_stereo* s, const uint8_t* left, const uint8_t* right, int w, int h, int stride,
                    double dt, orc_stereo_result* out)
{
    const orc_vo_params* p = &s->p;
    memset(out, 0, sizeof(*out));
    s->n_m_curr = s->n_m_pc = s->nT = s->G = s->n_inl = 0;
    s->nL = detect(s, left, w, h, stride, s->kpsL, s->descL);
    s->nR = detect(s, right, w, h, stride, s->kpsR, s->descR);
    out->n_left = s->nL; out->n_right = s->nR;
    int* ia = (int*)malloc(sizeof(int) * (size_t)s->cap * 4);
    int* ib = (int*)malloc(sizeof(int) * (size_t)s->cap * 4);

    if (!s->vo_initialized) {
        /* ---- VO:474-520 ---- */
        if (s->nL >= p->MIN_NUM_FEATURES && s->nR >= p->MIN_NUM_FEATURES) {
            match(s, s->descL, s->nL, s->descR, s->nR, s->results_match_prev, s->cap * 4, &s->n_match_prev);
            if (s->n_match_prev > p->MIN_NUM_FEATURES) s->vo_initialized = 1;
        }
        out->n_stereo_matches = s->n_match_prev;
        if (s->vo_initialized) {
            for (int i = 0; i < s->n_match_prev; i++) { ia[i] = s->results_match_prev[i].queryIdx; ib[i] = s->results_match_prev[i].trainIdx; }
            gather_descriptors(s->descL, s->nL, ia, s->n_match_prev, s->prevL_desc_as, desc_row_bytes(s)); s->n_prevL_desc_as = s->n_match_prev;
            s->n_prevL_as = gather_keypoints(s->kpsL, s->nL, ia, s->n_match_prev, s->prevL_kps_as);
            s->n_prevR_as = gather_keypoints(s->kpsR, s->nR, ib, s->n_match_prev, s->prevR_kps_as);
            memcpy(s->m_curr, s->results_match_prev, sizeof(orc_dmatch) * (size_t)(s->n_match_prev < s->cap ? s->n_match_prev : s->cap));
            s->n_m_curr = s->n_match_prev < s->cap ? s->n_match_prev : s->cap;
        }
        out->initialized = 0; out->valid = 0;
        free(ia); free(ib);
        return 0;
    }

    /* ---- VO:531-739 ---- */
    out->initialized = 1;
    int valid = 0;
    orc_keypoint* currL_kps_as = (orc_keypoint*)malloc(sizeof(orc_keypoint) * (size_t)s->cap);
    orc_keypoint* currR_kps_as = (orc_keypoint*)malloc(sizeof(orc_keypoint) * (size_t)s->cap);
    float* currL_desc_as = (float*)malloc(sizeof(float) * 128 * (size_t)s->cap);
    const int dim = desc_row_bytes(s);
    int n_currL_as = 0, n_currR_as = 0, n_currL_desc_as = 0;

    if (s->nL >= p->MIN_NUM_FEATURES && s->nR >= p->MIN_NUM_FEATURES) {                     /* VO:556 */
        match(s, s->descL, s->nL, s->descR, s->nR, s->m_curr, s->cap, &s->n_m_curr);         /* VO:558 */
        if (s->n_m_curr > p->MIN_NUM_FEATURES) {                                             /* VO:567 */
            for (int i = 0; i < s->n_m_curr; i++) { ia[i] = s->m_curr[i].queryIdx; ib[i] = s->m_curr[i].trainIdx; }
            gather_descriptors(s->descL, s->nL, ia, s->n_m_curr, currL_desc_as, dim); n_currL_desc_as = s->n_m_curr;   /* VO:576 */
            n_currL_as = gather_keypoints(s->kpsL, s->nL, ia, s->n_m_curr, currL_kps_as);                         /* VO:578 */
            n_currR_as = gather_keypoints(s->kpsR, s->nR, ib, s->n_m_curr, currR_kps_as);                         /* VO:579 */

            /* triangular matching VO:592 */
            match(s, s->prevL_desc_as, s->n_prevL_desc_as, s->descL, s->nL, s->m_pc, s->cap * 4, &s->n_m_pc);
            int T = s->n_m_pc;
            for (int i = 0; i < T; i++) { ia[i] = s->m_pc[i].queryIdx; ib[i] = s->m_pc[i].trainIdx; }
            orc_keypoint* pl = (orc_keypoint*)malloc(sizeof(orc_keypoint) * (size_t)(T + 1));
            orc_keypoint* pr = (orc_keypoint*)malloc(sizeof(orc_keypoint) * (size_t)(T + 1));
            orc_keypoint* cu = (orc_keypoint*)malloc(sizeof(orc_keypoint) * (size_t)(T + 1));
            int npl = gather_keypoints(s->prevL_kps_as, s->n_prevL_as, ia, T, pl);          /* VO:610 */
            int npr = gather_keypoints(s->prevR_kps_as, s->n_prevR_as, ia, T, pr);          /* VO:611 */
            int ncu = gather_keypoints(s->kpsL, s->nL, ib, T, cu);                          /* VO:613 */
            (void)ncu;
            if (T > p->MIN_NUM_FEATURES && npl == T && npr == T) {                           /* VO:626 */
                orc_point2f* x1 = (orc_point2f*)malloc(sizeof(orc_point2f) * T);
                orc_point2f* x2 = (orc_point2f*)malloc(sizeof(orc_point2f) * T);
                for (int i = 0; i < T; i++) { x1[i].x = pl[i].x; x1[i].y = pl[i].y; x2[i].x = pr[i].x; x2[i].y = pr[i].y; }  /* VO:616-617 */
                orc_triangulate_points(s->P_eye_left, s->P_right, x1, x2, T, s->points4D);  /* VO:631 */
                s->nT = T;
                double I[9] = {1,0,0,0,1,0,0,0,1}, z[3] = {0,0,0};
                s->G = orc_extract_3Dpoints(x1, x2, T, I, z, s->R_right, s->t_right, s->K_left, s->K_right, s->points4D,
                                            p->MIN_NUM_3DPOINTS, p->REPROJECTION_TOLERANCE, s->good_pts, s->good_idx);  /* VO:632 */
                if (s->G > p->MIN_NUM_3DPOINTS) {                                            /* VO:634 */
                    orc_point2f* ci = (orc_point2f*)malloc(sizeof(orc_point2f) * s->G);
                    for (int i = 0; i < s->G; i++) { ci[i].x = cu[s->good_idx[i]].x; ci[i].y = cu[s->good_idx[i]].y; }   /* VO:638-640 */
                    orc_solve_pnp_ransac(s->good_pts, ci, s->G, s->K_left, p->ITERATIONS_COUNT,
                                         (float)p->REPROJECTION_ERROR_THRESHOLD, p->CONFIDENCE,
                                         s->rvec, s->tvec, s->inliers, &s->n_inl);           /* VO:647-648 */
                    if (s->n_inl < p->MIN_NUM_INLIERS) valid = 0;                            /* VO:665 */
                    else {
                        double R[9];
                        orc_rodrigues_vec2mat(s->rvec, R);                                   /* VO:673 */
                        for (int i = 0; i < 3; i++) {                                        /* VO:675: -R^T t */
                            double acc = 0;
                            for (int k = 0; k < 3; k++) acc += R[k*3 + i] * s->tvec[k];
                            s->t_prev_curr[i] = acc * -1.0;
                        }
                        valid = 1;
                    }
                    free(ci);
                }
                free(x1); free(x2);
            }
            free(pl); free(pr); free(cu);
        }
    }
    /* output VO:717 -> VO:148-159 */
    out->valid = valid;
    out->n_stereo_matches = s->n_m_curr; out->n_tri_matches = s->n_m_pc; out->n_good3d = s->G; out->n_inliers = s->n_inl;
    for (int i = 0; i < 3; i++) { out->rvec[i] = s->rvec[i]; out->tvec[i] = s->tvec[i]; out->t_prev_curr[i] = s->t_prev_curr[i];
                                  out->velocity[i] = s->t_prev_curr[i] / dt; }
    /* state carry VO:727-733 (happens on failure too, with possibly empty sets) */
    memcpy(s->prevL_kps_as, currL_kps_as, sizeof(orc_keypoint) * (size_t)n_currL_as); s->n_prevL_as = n_currL_as;
    memcpy(s->prevR_kps_as, currR_kps_as, sizeof(orc_keypoint) * (size_t)n_currR_as); s->n_prevR_as = n_currR_as;
    memcpy(s->prevL_desc_as, currL_desc_as, (size_t)dim * (size_t)n_currL_desc_as); s->n_prevL_desc_as = n_currL_desc_as;
    free(currL_kps_as); free(currR_kps_as); free(currL_desc_as); free(ia); free(ib);
    return 0;
}

int orc_stereo_get(orc_stereo* s, const char* what, void* out, int cap_bytes)
{
    const void* src = NULL; size_t nb = 0; int count = 0;
#define CASE(name, ptr, cnt, esz) if (!strcmp(what, name)) { src = (ptr); count = (cnt); nb = (size_t)(cnt) * (esz); }
    CASE("kps_left", s->kpsL, s->nL, sizeof(orc_keypoint))
    CASE("kps_right", s->kpsR, s->nR, sizeof(orc_keypoint))
    CASE("desc_left", s->descL, s->nL, desc_row_bytes(s))
    CASE("desc_right", s->descR, s->nR, desc_row_bytes(s))
    CASE("matches_stereo", s->m_curr, s->n_m_curr, sizeof(orc_dmatch))
    CASE("matches_tri", s->m_pc, s->n_m_pc, sizeof(orc_dmatch))
    CASE("points4d", s->points4D, s->nT, 4*sizeof(float))
    CASE("good_pts", s->good_pts, s->G, 3*sizeof(double))
    CASE("good_idx", s->good_idx, s->G, sizeof(int))
    CASE("inliers", s->inliers, s->n_inl, sizeof(int))
#undef CASE
    if (!src && nb == 0 && count == 0 && strcmp(what, "kps_left")) { /* unknown or empty */ }
    if ((int)nb > cap_bytes) return -count;
    if (nb) memcpy(out, src, nb);
    return count;
}
