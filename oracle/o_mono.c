/*
 * o_mono.c -- CPU ORACLE (test infrastructure): the mono-path functions of uvo_libraries and a
 * ROS-free restatement of visual_odometry_node::mono_VO (VO:167-398).
 *   select_estimation_method  VOU:725-748      estimate_relative_pose    VOU:134-180
 *   extract_inliers           VOU:306-329      recover_pose_homography   VOU:581-624 (+ VOU:71-83, 9-15)
 *   convert_3Dpoints_camera   VOU:46-63        compute_scale_factor      VOU:23-38
 *   match_features (7-arg)    VOU:551-573      mono_output_computation   VO:126-140
 * PARITY UNPINNED vs OpenCV.
 */
#include "uvo_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

int orc_find_essential_mat(const orc_point2f* p1, const orc_point2f* p2, int n, const double* K, int method,
                           double prob, double threshold, int maxIters, double* E, uint8_t* mask);
int orc_recover_pose(const double* E, const orc_point2f* p1, const orc_point2f* p2, int n, const double* K, double* R, double* t, uint8_t* mask);
int orc_find_homography(const orc_point2f* p1, const orc_point2f* p2, int npoints, int method, double thr, int maxIters, double confidence,
                        double* H, uint8_t* mask);
int orc_decompose_homography_mat(const double* H, const double* K, double* Rs, double* ts, double* ns);

/* VOU:725-748: returns 1 for "use essential" (median pixel displacement >= DISTANCE) */
int orc_select_estimation_method(const orc_point2f* k1, const orc_point2f* k2, int n, int DISTANCE)
{
    double* d = (double*)malloc(sizeof(double) * (n + 1));
    for (int i = 0; i < n; i++) {
        double dx = k1[i].x - k2[i].x, dy = k1[i].y - k2[i].y;
        d[i] = sqrt(dx * dx + dy * dy);
    }
    double med = orc_compute_median(d, n);
    free(d);
    return med < DISTANCE ? 0 : 1;
}

/* VOU:306-329 */
int orc_extract_inliers(const orc_point2f* k1, const orc_point2f* k2, const uint8_t* mask, int n, orc_point2f* in1, orc_point2f* in2)
{
    int k = 0;
    for (int i = 0; i < n; i++) if (mask[i] != 0) { in1[k] = k1[i]; in2[k] = k2[i]; k++; }
    return k;       /* inlier_matches[k] = DMatch(k, k) is implied */
}

static void projection_matrix(const double* R, const double* t, const double* K, double* P)      /* VOU:9-15 */
{
    double Rt[12];
    for (int i = 0; i < 3; i++) { Rt[i*4] = R[i*3]; Rt[i*4+1] = R[i*3+1]; Rt[i*4+2] = R[i*3+2]; Rt[i*4+3] = t[i]; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) P[i*4 + j] = K[i*3]*Rt[j] + K[i*3+1]*Rt[4 + j] + K[i*3+2]*Rt[8 + j];
}

/* VOU:581-624.  The reference reads point.at<double>(2) from a 3x1 column view of a CV_32F matrix
 * (VOU:601-602): the 8 bytes starting at z_j, i.e. (z_j as low word, z_{j+1} as high word) of a double.
 * That reinterpretation is reproduced; for the last column it reads past the buffer (undefined) and
 * that column is not counted here.  Returns max_good_points; R, t are written only when a solution wins. */
int orc_recover_pose_homography(const double* H, const orc_point2f* p1, const orc_point2f* p2, int n, const double* K,
                                double HOMOGRAPHY_DISTANCE, double* R, double* t)
{
    double Rs[36], ts[12], ns[12];
    int solutions = orc_decompose_homography_mat(H, K, Rs, ts, ns);
    const double I[9] = {1,0,0,0,1,0,0,0,1}, z[3] = {0,0,0};
    double proj_std[12];
    projection_matrix(I, z, K, proj_std);
    int best = -1, max_good = 0;
    float* p4 = (float*)malloc(sizeof(float) * 4 * (n + 1));
    float* zrow = (float*)malloc(sizeof(float) * (n + 1));
    for (int i = 0; i < solutions; i++) {
        double P[12];
        projection_matrix(Rs + 9*i, ts + 3*i, K, P);
        orc_triangulate_points(proj_std, P, p1, p2, n, p4);
        for (int j = 0; j < n; j++) {                       /* convert_from_homogeneous_coords, VOU:71-83: col / w as float */
            float inv = (float)(1.0 / (double)p4[3*n + j]);
            zrow[j] = p4[2*n + j] * inv + 0.f;
        }
        int good = 0;
        for (int j = 0; j + 1 < n; j++) {
            uint32_t lo, hi; memcpy(&lo, &zrow[j], 4); memcpy(&hi, &zrow[j + 1], 4);
            uint64_t bits = ((uint64_t)hi << 32) | lo;
            double v; memcpy(&v, &bits, 8);
            if (v > 0 && v < HOMOGRAPHY_DISTANCE) good++;
        }
        if (good > max_good) { best = i; max_good = good; }
    }
    if (best != -1) {
        const double* tb = ts + 3*best;
        double nrm = sqrt(tb[0]*tb[0] + tb[1]*tb[1] + tb[2]*tb[2]);
        double inv = 1.0 / nrm;                               /* Mat / scalar = Mat * (1/scalar) */
        memcpy(R, Rs + 9*best, sizeof(double)*9);
        for (int k = 0; k < 3; k++) t[k] = tb[k] * inv;
    }
    free(p4); free(zrow);
    return max_good;
}

/* VOU:134-180.  *use_essential is the reference's global (in/out).  OpenCV would throw (and the node
 * die) if findEssentialMat / findHomography returned an empty matrix; here that attempt simply fails. */
int orc_estimate_relative_pose(const orc_vo_params* p, int* use_essential, const orc_point2f* k1, const orc_point2f* k2, int n,
                               const double* K, double* R, double* t, orc_point2f* in1, orc_point2f* in2, int* n_in, uint8_t* mask_out)
{
    int estimate_completed = 0, switch_method = 0, success = 0;
    uint8_t* mask = (uint8_t*)calloc(n + 1, 1);
    while (!estimate_completed) {
        int valid_inliers = 0;
        memset(mask, 0, n);
        if (*use_essential) {
            double E[9];
            int ok = orc_find_essential_mat(k1, k2, n, K, p->ESSENTIAL_OUTLIER_METHOD, p->ESSENTIAL_CONFIDENCE, p->ESSENTIAL_THRESHOLD,
                                            (int)p->ESSENTIAL_MAX_ITERS, E, mask);
            *n_in = orc_extract_inliers(k1, k2, mask, n, in1, in2);
            if (ok) orc_recover_pose(E, k1, k2, n, K, R, t, mask);
            else memset(mask, 0, n);
        } else {
            double H[9];
            int ok = orc_find_homography(k1, k2, n, p->HOMOGRAPHY_OUTLIER_METHOD, p->HOMOGRAPHY_THRESHOLD, (int)p->HOMOGRAPHY_MAX_ITERS,
                                         p->HOMOGRAPHY_CONFIDENCE, H, mask);
            *n_in = orc_extract_inliers(k1, k2, mask, n, in1, in2);
            if (ok) orc_recover_pose_homography(H, k1, k2, n, K, p->HOMOGRAPHY_DISTANCE, R, t);
        }
        for (int i = 0; i < n; i++) valid_inliers += mask[i] != 0;
        double valid_point_fraction = (double)valid_inliers / n;
        if (valid_point_fraction >= p->VPF_THRESHOLD && valid_inliers >= p->MIN_NUM_INLIERS) { success = 1; estimate_completed = 1; }
        else {
            if (switch_method) break;
            switch_method = 1;
            *use_essential = !*use_essential;
        }
    }
    if (mask_out) memcpy(mask_out, mask, n);
    free(mask);
    return success;
}

/* VOU:46-63: keeps the ORIGINAL row i when (R p_i + t).z > 0; out is G' x 3 (the reference returns its transpose) */
int orc_convert_3Dpoints_camera(const double* pts, int n, const double* R, const double* t, double* out)
{
    int k = 0;
    for (int i = 0; i < n; i++) {
        const double* p = pts + 3*i;
        double z = (R[6]*p[0] + R[7]*p[1] + R[8]*p[2]) * 1.0 + t[2] * 1.0;
        if (z > 0) { out[3*k] = p[0]; out[3*k+1] = p[1]; out[3*k+2] = p[2]; k++; }
    }
    return k;
}

/* VOU:23-38: `distance` is a float parameter; world_points is 3 x n (rows < 3 or empty => 0.0) */
double orc_compute_scale_factor(float distance, const double* pts_nx3, int n)
{
    if (n <= 0) return 0.0;
    double* z = (double*)malloc(sizeof(double) * n);
    for (int i = 0; i < n; i++) z[i] = pts_nx3[3*i + 2];
    double med = orc_compute_median(z, n);
    free(z);
    return distance / med;
}

/* ---- mono VO state machine (VO:167-398) ---- */
struct orc_mono {
    orc_vo_params p;
    double K[9];
    int cap, vo_initialized, use_essential, detector;      /* detector: the reference's global FEATURE_DETECTOR (VOH:25): 0 "SURF", 1 "SIFT", 2 "AKAZE", 3 "ORB" */
    int orb_pattern[1024];                                  /* "ORB": the sampling table (o_orb.c: an input) */
    orc_keypoint* prev_kps; float* prev_desc; int n_prev;
    double R[9], t[3], SF;
    /* last-step intermediates */
    orc_keypoint* kps; float* desc; int n_kps;
    orc_dmatch* matches; int n_matches;
    uint8_t* mask; int n_inl;
    double* good_pts; int G;
};

/* orc_mono_result.published == 0: the frame was skipped with `continue` (VO:276-307): nothing is published */

orc_mono* orc_mono_create(const orc_vo_params* p, const double* K, int max_kpts)
{
    orc_mono* s = (orc_mono*)calloc(1, sizeof(orc_mono));
    s->p = *p; memcpy(s->K, K, sizeof(s->K)); s->cap = max_kpts; s->use_essential = 1; s->SF = 1.0;
    s->R[0] = s->R[4] = s->R[8] = 1.0;
    size_t c = (size_t)max_kpts;
    s->prev_kps = (orc_keypoint*)malloc(sizeof(orc_keypoint)*c); s->prev_desc = (float*)malloc(sizeof(float)*128*c);
    s->kps = (orc_keypoint*)malloc(sizeof(orc_keypoint)*c); s->desc = (float*)malloc(sizeof(float)*128*c);
    s->matches = (orc_dmatch*)malloc(sizeof(orc_dmatch)*c); s->mask = (uint8_t*)malloc(c);
    s->good_pts = (double*)malloc(sizeof(double)*3*c);
    return s;
}
void orc_mono_destroy(orc_mono* s)
{
    if (!s) return;
    free(s->prev_kps); free(s->prev_desc); free(s->kps); free(s->desc); free(s->matches); free(s->mask); free(s->good_pts); free(s);
}

void orc_mono_use_sift(orc_mono* s, int on) { s->detector = on ? 1 : 0; }
void orc_mono_use_detector(orc_mono* s, int detector, const int* pattern)
{
    s->detector = detector;
    if (detector == 3 && pattern) memcpy(s->orb_pattern, pattern, sizeof(s->orb_pattern));
}

int orc_mono_step(orc_mono* s, const uint8_t* img, int w, int h, int stride, double range, double dt, orc_mono_result* out)
{
    const orc_vo_params* p = &s->p;
    memset(out, 0, sizeof(*out));
    s->n_matches = s->n_inl = s->G = 0;
    orc_surf_params sp = { (double)p->SURF_MIN_HESSIAN, p->SURF_OCTAVES_NUMBER, p->SURF_OCTAVES_LAYERS, p->SURF_EXTENDED, p->SURF_UPRIGHT };
    const int ddim = s->detector == 2 ? 61 : s->detector == 3 ? 32 : (s->detector == 1 || p->SURF_EXTENDED ? 128 : 64);
    int n;
    if (s->detector >= 2) {
        /* VOU:93-105: CV_8U rows.  The mono loop's match_features overload (VOU:551-573) constructs BFMatcher(NORM_L2) whatever the detector:
         * on CV_8U rows OpenCV sums the squared byte differences in integers and takes the float square root -- every sum is below 2^24,
         * so the float matcher on the bytes widened to float returns the same distances; the rows are kept widened */
        uint8_t* b = (uint8_t*)malloc((size_t)s->cap * 64);
        n = s->detector == 2 ? orc_akaze_detect_and_compute(img, w, h, stride, s->kps, b, s->cap)
                             : orc_orb_detect_and_compute(img, w, h, stride, 10000, 1.2f, 8, 31, 0, 31, 10, s->orb_pattern, s->kps, b, s->cap);
        const int nn = n < 0 ? s->cap : n;
        for (size_t i = 0; i < (size_t)nn * ddim; i++) s->desc[i] = (float)b[i];
        free(b);
    } else
    n = s->detector == 1 ? orc_sift_detect_and_compute(img, w, h, stride, 10000, 3, 0.03, 10, 1.6, s->kps, s->desc, s->cap)      /* VOU:107-112 */
                         : orc_surf_detect_and_compute(img, w, h, stride, &sp, s->kps, s->desc, s->cap);
    if (n < 0) n = s->cap;
    s->n_kps = n; out->n_kps = n;
    if (!s->vo_initialized) {                                            /* VO:227-245 */
        memcpy(s->prev_kps, s->kps, sizeof(orc_keypoint)*(size_t)n); memcpy(s->prev_desc, s->desc, sizeof(float)*ddim*(size_t)n); s->n_prev = n;
        if (n >= p->MIN_NUM_FEATURES) s->vo_initialized = 1;
        return 0;
    }
    out->initialized = 1;
#define ROLL_STATE() do { memcpy(s->prev_kps, s->kps, sizeof(orc_keypoint)*(size_t)n); memcpy(s->prev_desc, s->desc, sizeof(float)*ddim*(size_t)n); s->n_prev = n; } while (0)
    if (n < p->MIN_NUM_FEATURES) { ROLL_STATE(); return 0; }             /* VO:276-284 */
    orc_match_knn2_ratio(s->prev_desc, s->n_prev, s->desc, n, ddim, (float)p->LOWE_RATIO_THRESHOLD, s->matches, s->cap, &s->n_matches);   /* VO:287 */
    int M = s->n_matches;
    out->n_matches = M;
    if (M < p->MIN_NUM_FEATURES) { ROLL_STATE(); return 0; }             /* VO:299-307 */
    orc_point2f* k1 = (orc_point2f*)malloc(sizeof(orc_point2f)*M); orc_point2f* k2 = (orc_point2f*)malloc(sizeof(orc_point2f)*M);
    orc_point2f* in1 = (orc_point2f*)malloc(sizeof(orc_point2f)*M); orc_point2f* in2 = (orc_point2f*)malloc(sizeof(orc_point2f)*M);
    for (int i = 0; i < M; i++) {                                         /* VOU:567-568 */
        const orc_keypoint* a = &s->prev_kps[s->matches[i].queryIdx]; const orc_keypoint* b = &s->kps[s->matches[i].trainIdx];
        k1[i].x = a->x; k1[i].y = a->y; k2[i].x = b->x; k2[i].y = b->y;
    }
    s->use_essential = orc_select_estimation_method(k1, k2, M, p->DISTANCE);      /* VO:310-317 */
    int n_in = 0;
    int success = orc_estimate_relative_pose(p, &s->use_essential, k1, k2, M, s->K, s->R, s->t, in1, in2, &n_in, s->mask);   /* VO:323 */
    out->success = success; out->used_essential = s->use_essential; out->n_inliers = n_in; s->n_inl = n_in;
    int valid = success ? 1 : 0;                                          /* VO:335-344 */
    if (success) {                                                        /* VO:351-376 */
        const double I[9] = {1,0,0,0,1,0,0,0,1}, z[3] = {0,0,0};
        double P_prev[12], P_curr[12];
        projection_matrix(I, z, s->K, P_prev);
        projection_matrix(s->R, s->t, s->K, P_curr);
        float* p4 = (float*)malloc(sizeof(float)*4*(n_in + 1));
        int* idx = (int*)malloc(sizeof(int)*(n_in + 1));
        orc_triangulate_points(P_prev, P_curr, in1, in2, n_in, p4);
        s->G = orc_extract_3Dpoints(in1, in2, n_in, I, z, s->R, s->t, s->K, s->K, p4, p->MIN_NUM_3DPOINTS, p->REPROJECTION_TOLERANCE, s->good_pts, idx);
        out->n_good3d = s->G;
        if (s->G < p->MIN_NUM_3DPOINTS) valid = 0;
        else {
            double* front = (double*)malloc(sizeof(double)*3*(s->G + 1));
            int nf = orc_convert_3Dpoints_camera(s->good_pts, s->G, s->R, s->t, front);
            out->n_front = nf;
            if (nf > 0) s->SF = orc_compute_scale_factor((float)range, front, nf);
            else valid = 0;
            free(front);
        }
        free(p4); free(idx);
    }
    /* VO:126-140: -SF * R^T * t / dt  (gemm with alpha = (-SF) * (1/dt)) */
    double alpha = (-s->SF) * (1.0 / dt);
    for (int i = 0; i < 3; i++) {
        double acc = 0;
        for (int k = 0; k < 3; k++) acc += s->R[k*3 + i] * s->t[k];
        out->velocity[i] = acc * alpha;
    }
    out->published = 1; out->valid = valid; out->SF = s->SF;
    memcpy(out->R, s->R, sizeof(out->R)); memcpy(out->t, s->t, sizeof(out->t));
    ROLL_STATE();                                                         /* VO:392-395 */
#undef ROLL_STATE
    free(k1); free(k2); free(in1); free(in2);
    return 0;
}

int orc_mono_get(orc_mono* s, const char* what, void* out, int cap_bytes)
{
    const void* src = NULL; size_t nb = 0; int count = 0;
    if (!strcmp(what, "kps")) { src = s->kps; count = s->n_kps; nb = (size_t)count*sizeof(orc_keypoint); }
    else if (!strcmp(what, "matches")) { src = s->matches; count = s->n_matches; nb = (size_t)count*sizeof(orc_dmatch); }
    else if (!strcmp(what, "mask")) { src = s->mask; count = s->n_matches; nb = (size_t)count; }
    else if (!strcmp(what, "good_pts")) { src = s->good_pts; count = s->G; nb = (size_t)count*3*sizeof(double); }
    if ((int)nb > cap_bytes) return -count;
    if (nb) memcpy(out, src, nb);
    return count;
}
