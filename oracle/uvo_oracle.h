/*
 * uvo_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the UVO feature-match + relative-pose hot path
 * (reference: team-ergo-unipi/ergo_uvo, uvo_libraries/src/VO_utility.cpp,
 * uvo/include/visual_odometry.h) and of the OpenCV 4.5.x / opencv_contrib
 * routines those files call.  OpenCV is an un-vendored third-party dependency
 * of the reference ("OpenCV 4.5" + contrib xfeatures2d, README.md:60,70-72;
 * find_package(OpenCV 4 REQUIRED), uvo_libraries/CMakeLists.txt:15) and is not
 * available in this environment, and the reference ships no tests, golden
 * vectors or data.
 *
 *   >>> PARITY UNPINNED vs OpenCV: nothing here could be checked against a
 *   >>> run of the reference.  What pins this oracle is analytic known-answer
 *   >>> tests (tests/test_oracle_*.py) and self-generated fixtures
 *   >>> (tests/golden/).  Parity shown elsewhere is HIP <-> this restatement.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (ergo_uvo_amd/) never includes, links or
 * calls anything in oracle/.
 *
 * Every function cites the reference line(s) it follows (VOU = VO_utility.cpp,
 * VO = visual_odometry.h, MU = math_utility.cpp) and/or the upstream OpenCV
 * routine it restates ([UPSTREAM], recalled behaviour).
 */
#ifndef UVO_ORACLE_H
#define UVO_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- boundary PODs (same memory layout as cv::KeyPoint / cv::DMatch / cv::Point2f) ---- */
typedef struct { float x, y, size, angle, response; int octave, class_id; } orc_keypoint; /* 28 B */
typedef struct { int queryIdx, trainIdx, imgIdx; float distance; } orc_dmatch;           /* 16 B */
typedef struct { float x, y; } orc_point2f;

/* ---- scalar helpers ([UPSTREAM] core/fast_math.hpp, core/operations.hpp) ---- */
int      orc_cvRound(double v);          /* lrint, round-half-even */
int      orc_cvRoundf(float v);
int      orc_cvFloor(double v);
int      orc_cvCeil(double v);
typedef struct { uint64_t state; } orc_rng;
void     orc_rng_init(orc_rng* r, uint64_t seed);
uint32_t orc_rng_next(orc_rng* r);
int      orc_rng_uniform(orc_rng* r, int a, int b);

/* deterministic elementary functions shared (operation for operation) with the
 * HIP path: plain IEEE double arithmetic, no libm, so host and device agree
 * bit for bit.  <= 2 ulp from libm (documented departure). */
double orc_hypot(double a, double b);
void   orc_sincos(double x, double* s, double* c);
double orc_acos(double c);

/* ---- dense linear algebra ([UPSTREAM] core/src/lapack.cpp, matmul) ---- */
/* One-sided Jacobi SVD exactly as JacobiSVDImpl_<double>: At is n rows of m
 * (rows = columns of A), W n values, Vt n x n (may be NULL), n1 rows of U^T
 * completed.  Strides in elements. */
void orc_jacobi_svd(double* At, int astep, double* W, double* Vt, int vstep, int m, int n, int n1);
/* cv::SVD::compute(A[m x n]) -> w[min], u[m x min], vt[min x n] (row-major, tight) */
void orc_svd(const double* A, int m, int n, double* w, double* u, double* vt);
/* cv::solve(A[m x n], b[m], DECOMP_SVD) -> x[n] */
void orc_solve_svd(const double* A, int m, int n, const double* b, double* x);
/* cv::invert(A[3x3], DECOMP_SVD) */
void orc_invert3_svd(const double* A, double* Ainv);
/* cv::mulTransposed(src[rows x cols], aTa=true) -> dst[cols x cols] */
void orc_mul_transposed(const double* src, int rows, int cols, double* dst);

/* ---- SURF ([UPSTREAM] opencv_contrib xfeatures2d/src/surf.cpp; VOU:114-119) ---- */
typedef struct {
    double hessianThreshold; int nOctaves, nOctaveLayers, extended, upright;
} orc_surf_params;
void orc_integral_u8(const uint8_t* img, int w, int h, int stride, int32_t* sum /* (h+1)*(w+1) */);
/* one det/trace layer: planes are (h/step) x (w/step), caller-zeroed */
void orc_surf_layer(const int32_t* sum, int w, int h, int size, int step, float* det, float* trace);
/* full detectAndCompute; returns number of keypoints (<= cap, or -needed if cap too small) */
int  orc_surf_detect_and_compute(const uint8_t* img, int w, int h, int stride, const orc_surf_params* p,
                                 orc_keypoint* kps, float* desc, int cap);
/* cv::resize(u8 src sw x sh -> dst dw x dh, INTER_AREA), downscale only */
void orc_resize_area_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh);
void orc_gaussian_kernel_f32(int n, double sigma, float* out);

/* ---- matching ([UPSTREAM] features2d BFMatcher / core batchDistance; VOU:515-573) ---- */
float orc_l2_distance_f32(const float* a, const float* b, int n);
/* knnMatch k=2 + Lowe ratio; appends to out starting at *m (VOU:533-540 appends). */
int  orc_match_knn2_ratio(const float* d1, int n1, const float* d2, int n2, int dim, float ratio,
                          orc_dmatch* out, int cap, int* m);
/* raw knn (for tests): idx[2*n1], dist[2*n1]; idx=-1 when absent */
void orc_knn2(const float* d1, int n1, const float* d2, int n2, int dim, int* idx, float* dist);
/* VOU:520-524, the AKAZE / ORB branch: BFMatcher(NORM_HAMMING) on `bytes`-byte binary descriptors */
void orc_knn2_hamming(const uint8_t* d1, int n1, const uint8_t* d2, int n2, int bytes, int* idx, float* dist);
int  orc_match_knn2_ratio_hamming(const uint8_t* d1, int n1, const uint8_t* d2, int n2, int bytes, float ratio,
                                  orc_dmatch* out, int cap, int* m);

/* ---- geometry ([UPSTREAM] calib3d triangulate.cpp, calibration.cpp) ---- */
void orc_triangulate_points(const double* P1, const double* P2, const orc_point2f* x1, const orc_point2f* x2,
                            int n, float* out4xN);
void orc_rodrigues_vec2mat(const double* r, double* R);
void orc_rodrigues_mat2vec(const double* R, double* r);
/* cv::projectPoints, no distortion; R is a 3x3 matrix */
void orc_project_points_f64(const double* X, int n, const double* R, const double* t, const double* K, double* out2n);
void orc_project_points_f32(const float* X, int n, const double* R, const double* t, const double* K, float* out2n);

/* ---- uvo_libraries functions (VOU / MU) ---- */
typedef struct {
    /* vo_params */
    int    DISTANCE;
    double LOWE_RATIO_THRESHOLD;
    int    ESSENTIAL_OUTLIER_METHOD; double ESSENTIAL_MAX_ITERS, ESSENTIAL_CONFIDENCE, ESSENTIAL_THRESHOLD;
    int    HOMOGRAPHY_OUTLIER_METHOD; double HOMOGRAPHY_MAX_ITERS, HOMOGRAPHY_CONFIDENCE, HOMOGRAPHY_THRESHOLD, HOMOGRAPHY_DISTANCE;
    double VPF_THRESHOLD, REPROJECTION_TOLERANCE;
    int    MIN_NUM_FEATURES, MIN_NUM_3DPOINTS, MIN_NUM_INLIERS;
    int    ITERATIONS_COUNT; double REPROJECTION_ERROR_THRESHOLD, CONFIDENCE; int USE_EXTRINSIC_GUESS, PNP_METHOD_FLAG;
    int    SURF_MIN_HESSIAN, SURF_OCTAVES_NUMBER, SURF_OCTAVES_LAYERS, SURF_EXTENDED, SURF_UPRIGHT;
} orc_vo_params;

double orc_compute_median(const double* v, int n);                        /* MU:65-86 */
void   orc_compute_mean_and_variance(const double* v, int n, double* mv); /* MU:35-56 */
/* VOU:632-651 */
void   orc_reproject_errors(const double* world, int n, const double* R, const double* t, const double* K,
                            const orc_point2f* img, double* err);
/* VOU:188-237; returns G, fills pts[G*3] (f64) and idx[G] */
int    orc_extract_3Dpoints(const orc_point2f* k1, const orc_point2f* k2, int n,
                            const double* R1, const double* t1, const double* R2, const double* t2,
                            const double* K1, const double* K2, const float* points4D /* 4 x n */,
                            int min_num_3dpoints, double reproj_tol, double* pts, int* idx);

/* cv::solvePnPRansac(..., flags = SOLVEPNP_EPNP) as called at VO:647-648.
 * obj: n x 3 f64, img: n Point2f.  Returns 1/0 (OpenCV bool); inliers ascending. */
int    orc_solve_pnp_ransac(const double* obj, const orc_point2f* img, int n, const double* K,
                            int iterationsCount, float reprojectionError, double confidence,
                            double* rvec, double* tvec, int* inliers, int* n_inliers);
/* one EPnP solve on double points (normalised image coords xn,yn given as us = x*fu+uc) */
void   orc_epnp(const double* pws, const double* us, int n, double fu, double fv, double uc, double vc,
                double* R, double* t);
int    orc_ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters);

/* ---- stereo VO state machine (VO:406-741), ROS-free ---- */
typedef struct orc_stereo orc_stereo;
typedef struct {
    int    valid;                 /* successful_estimate.data */
    int    initialized;           /* 0 while in the init loop (VO:474-506) */
    int    n_left, n_right;       /* keypoints per image */
    int    n_stereo_matches;      /* results_match_curr.size() */
    int    n_tri_matches;         /* results_match_prev_curr.size() */
    int    n_good3d;              /* good_prevCam_points.rows */
    int    n_inliers;             /* inliers_idx.rows */
    double rvec[3], tvec[3];      /* R_currCam_prevCam_Vec, t_currCam_prevCam */
    double t_prev_curr[3];        /* t_prevCam_currCam (kept on failure) */
    double velocity[3];           /* t_prevCam_currCam / dt  (VO:148-159) */
} orc_stereo_result;
orc_stereo* orc_stereo_create(const orc_vo_params* p, const double* K_left, const double* K_right,
                              const double* R_right, const double* t_right, int max_kpts);
void        orc_stereo_destroy(orc_stereo* s);
void        orc_stereo_use_detector(orc_stereo* s, int detector /* 0 "SURF", 1 "SIFT", 2 "AKAZE", 3 "ORB" */, const int* orb_pattern /* 1024 ints for "ORB" */);
void        orc_stereo_use_sift(orc_stereo* s, int on);    /* FEATURE_DETECTOR = "SIFT" instead of "SURF" (detect_features VOU:107-112, match_features VOU:525-529) */
/* o_orb.c: ORB::create(nfeatures, scaleFactor, nlevels, edgeThreshold, 0, 2, HARRIS_SCORE, patchSize, fastThreshold)->detectAndCompute
 * (VO_utility.cpp:100-105).  pattern: 1024 ints in the layout of OpenCV's bit_pattern_31_ (x0, y0, x1, y1 per bit) or NULL (keypoints
 * only); desc: cap x 32 bytes; -(count) if cap is too small */
int         orc_orb_detect_and_compute(const uint8_t* img, int w, int h, int stride, int nfeatures, float scaleFactor, int nlevels, int edgeThreshold,
                                       int firstLevel, int patchSize, int fastThreshold, const int* pattern, orc_keypoint* kps, uint8_t* desc, int cap);
int         orc_orb_levels(int img_w, int img_h, int nfeatures, float scaleFactor, int nlevels, int edgeThreshold, int firstLevel, int patchSize,
                           int* out /* [nlevels][3]: w, h, features wanted */, float* scale);
void        orc_resize_linear_exact_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride);
void        orc_fast_scores(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score);
int         orc_fast_detect(const uint8_t* img, int w, int h, int stride, int threshold, orc_keypoint* kps, int cap);
int         orc_retain_best(const float* responses, int n, int n_points, int* perm);
float       orc_orb_harris(const uint8_t* center, int step);
void        orc_orb_umax(int halfPatchSize, int* umax);
float       orc_orb_ic_angle(const uint8_t* center, int step, const int* umax, int half_k);
void        orc_orb_blur_kernel(int* k7);
void        orc_orb_blur_u8(const uint8_t* src, int w, int h, uint8_t* dst);
void        orc_orb_random_pattern(int patchSize, int* pattern, int npoints);
void        orc_orb_describe(const uint8_t* center, int step, float angle_deg, const int* pattern, uint8_t* desc);
int         orc_orb_level_image(const uint8_t* img, int w, int h, int stride, float scaleFactor, int nlevels, int level, int blurred, uint8_t* out, int cap, int* ow, int* oh);
/* o_akaze.c: AKAZE::create()->detectAndCompute (VO_utility.cpp:93-98); desc: cap x 61 bytes (M-LDB, 486 bits) or NULL; -(count) if cap is too small */
int         orc_akaze_detect_and_compute(const uint8_t* img, int w, int h, int stride, orc_keypoint* kps, uint8_t* desc, int cap);
int         orc_akaze_fed_tau(float T, float tau_max, float* tau /* >= 256 */);
int         orc_akaze_levels(int img_w, int img_h, int* out /* [n][6]: w, h, octave, sigma_size, border, nsteps */, float* esigma);
int         orc_akaze_plane(const uint8_t* img, int w, int h, int stride, int level, int what /* 0 Lt, 1 Lsmooth, 2 Lx, 3 Ly, 4 Ldet */, float* out,
                            int* ow, int* oh, float* kcontrast);
void        orc_akaze_scharr(const float* src, int w, int h, int xorder, float* dst);
void        orc_akaze_pm_g2(const float* lx, const float* ly, int n, float k, float* dst);
void        orc_akaze_nld_step(const float* lt, const float* lf, int w, int h, float step_size, float* out);
/* o_sift.c: SIFT::create(nfeatures, nOctaveLayers, contrastThreshold, edgeThreshold, sigma)->detectAndCompute; desc: cap x 128 floats or NULL */
int         orc_sift_detect_and_compute(const uint8_t* img, int w, int h, int stride, int nfeatures, int nOctaveLayers, double contrastThreshold,
                                        double edgeThreshold, double sigma, orc_keypoint* kps, float* desc, int cap);
int         orc_stereo_step(orc_stereo* s, const uint8_t* left, const uint8_t* right, int w, int h, int stride,
                            double dt, orc_stereo_result* out);
/* introspection for parity tests: last step's intermediate arrays */
int         orc_stereo_get(orc_stereo* s, const char* what, void* out, int cap_bytes);

/* ---- mono path ([UPSTREAM] five-point.cpp, fundam.cpp, levmarq.cpp, homography_decomp.cpp; VOU / VO mono) ---- */
int    orc_solve_poly(const double* coeffs, int degree, double* roots_re, double* roots_im);
void   orc_jacobi_eigen(double* A, int n, double* W, double* V);
int    orc_five_point(const double* q1, const double* q2, double* models /* <= 10 x 9 */);
void   orc_sampson_error(const double* p1, const double* p2, int n, const double* E, float* err);
int    orc_find_essential_mat(const orc_point2f* p1, const orc_point2f* p2, int n, const double* K, int method,
                              double prob, double threshold, int maxIters, double* E, uint8_t* mask);
void   orc_decompose_essential_mat(const double* E, double* R1, double* R2, double* t);
int    orc_recover_pose(const double* E, const orc_point2f* p1, const orc_point2f* p2, int n, const double* K,
                        double* R, double* t, uint8_t* mask);
int    orc_homography_kernel(const float* M, const float* m, int count, double* H);
void   orc_homography_error(const float* M, const float* m, int count, const double* H, float* err);
int    orc_homography_check_subset(const float* ms1, const float* ms2, int count);
int    orc_find_homography(const orc_point2f* p1, const orc_point2f* p2, int n, int method, double thr, int maxIters,
                           double confidence, double* H, uint8_t* mask);
int    orc_decompose_homography_mat(const double* H, const double* K, double* Rs, double* ts, double* ns);
int    orc_select_estimation_method(const orc_point2f* k1, const orc_point2f* k2, int n, int DISTANCE);
int    orc_extract_inliers(const orc_point2f* k1, const orc_point2f* k2, const uint8_t* mask, int n, orc_point2f* in1, orc_point2f* in2);
int    orc_recover_pose_homography(const double* H, const orc_point2f* p1, const orc_point2f* p2, int n, const double* K,
                                   double HOMOGRAPHY_DISTANCE, double* R, double* t);
int    orc_estimate_relative_pose(const orc_vo_params* p, int* use_essential, const orc_point2f* k1, const orc_point2f* k2, int n,
                                  const double* K, double* R, double* t, orc_point2f* in1, orc_point2f* in2, int* n_in, uint8_t* mask_out);
int    orc_convert_3Dpoints_camera(const double* pts, int n, const double* R, const double* t, double* out);
double orc_compute_scale_factor(float distance, const double* pts_nx3, int n);
typedef struct orc_mono orc_mono;
typedef struct {
    int published, valid, initialized, used_essential, success;
    int n_kps, n_matches, n_inliers, n_good3d, n_front;
    double R[9], t[3], SF, velocity[3];
} orc_mono_result;
orc_mono* orc_mono_create(const orc_vo_params* p, const double* K, int max_kpts);
void      orc_mono_destroy(orc_mono* s);
void      orc_mono_use_sift(orc_mono* s, int on);       /* FEATURE_DETECTOR = "SIFT" */
void        orc_mono_use_detector(orc_mono* s, int detector /* 0 "SURF", 1 "SIFT", 2 "AKAZE", 3 "ORB" */, const int* orb_pattern /* 1024 ints for "ORB" */);
int       orc_mono_step(orc_mono* s, const uint8_t* img, int w, int h, int stride, double range, double dt, orc_mono_result* out);
int       orc_mono_get(orc_mono* s, const char* what, void* out, int cap_bytes);

#ifdef __cplusplus
}
#endif

/* ---- get_image preprocessing (o_preproc.c) ---- */
void orc_rgb2gray_u8(const uint8_t* rgb, int w, int h, int stride, uint8_t* gray);
int  orc_resize_area_u8c3(const uint8_t* src, int sw, int sh, int stride, uint8_t* dst, int dw, int dh);
void orc_init_undistort_map(const double* A, const double* dist4, const double* Ar, int cols, int rows, int16_t* map1, uint16_t* map2);
void orc_remap_bilinear_u8(const uint8_t* src, int sw, int sh, const int16_t* map1, const uint16_t* map2, uint8_t* dst, int dw, int dh);
void orc_undistort_u8(const uint8_t* src, int w, int h, const double* K, const double* dist4, const double* newK, uint8_t* dst);
void orc_clahe_u8(const uint8_t* src, int w, int h, double clip_limit, uint8_t* dst);
int  orc_get_image(const uint8_t* rgb, int w, int h, int stride, int desired_width, const double* K, const double* dist4,
                   const double* newK, int clahe_on, int clip_limit, uint8_t* out, int* out_w, int* out_h);
/* reference VO_utility.cpp:658-675 (resize_camera_matrix): K scaled in place, newK = getOptimalNewCameraMatrix(alpha = 0) */
int  orc_resize_camera_matrix(int original_width, int original_height, int desired_width, double* K, const double* dist4, double* newK,
                              int* desired_height_out);

#endif
