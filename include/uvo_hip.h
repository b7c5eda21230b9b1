/*
 * uvo_hip.h -- C ABI of libuvo_hip.so: the MI355X (gfx950) implementation of UVO's per-frame
 * feature-match + relative-pose hot path.
 *
 * The reference (team-ergo-unipi/ergo_uvo) has no FFI: its boundary is the C++ shared library
 * `uvo_libraries` (free functions, uvo_libraries/include/uvo_libraries/VO_utility.h:96-117, with
 * parameters in the mutable globals of VO_utility.h:25-89) plus the direct OpenCV calls in
 * uvo/include/visual_odometry.h.  Each entry point below names the reference interface it
 * replaces.  Everything is POD: plain pointers, sizes and status codes; nothing throws across
 * this boundary (the reference lets cv::Exception kill the process and relies on roslaunch
 * respawn, uvo/launch/UVO_node.launch:24,38).
 *
 * Memory: `mem` says where caller buffers live (UVO_MEM_HOST or UVO_MEM_DEVICE = HIP device
 * memory of the context's GPU).  Outputs are caller-allocated with a capacity; counts come back
 * through pointers.  A context is single-caller, owns one HIP stream and all device workspaces
 * (no per-call allocation); distinct contexts are independent.
 */
#ifndef UVO_HIP_H
#define UVO_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct uvo_ctx uvo_ctx;

typedef enum {
    UVO_OK = 0,
    UVO_INVALID_ARG = 1,
    UVO_TOO_FEW_POINTS = 2,
    UVO_CAPACITY = 3,          /* a caller buffer or the context's max_kpts was too small */
    UVO_HIP_ERROR = 4,
    UVO_NO_DEVICE = 5
} uvo_status;

/* UVO_MEM_DEVICE buffers are read/written on the context's own (non-blocking) streams.  Ordering: either the data is
 * complete before the call (the caller synchronised the stream that produced it), or the producing stream is declared with
 * uvo_ctx_set_producer_stream and the context orders its reads after the work queued on that stream at call time.
 * Lifetime: a buffer given to a synchronous call may be reused when the call returns; a buffer given to uvo_stereo_submit /
 * uvo_mono_submit is read asynchronously and must stay valid and unmodified until the matching collect returns (a device image with
 * tight pitch and a 16-byte aligned base is not even copied: the detector reads it in place). */
enum { UVO_MEM_HOST = 0, UVO_MEM_DEVICE = 1 };

/* cv::KeyPoint (28 B), cv::DMatch (16 B), cv::Point2f -- same field order and size */
typedef struct { float x, y, size, angle, response; int octave, class_id; } uvo_keypoint;
typedef struct { int queryIdx, trainIdx, imgIdx; float distance; } uvo_dmatch;
typedef struct { float x, y; } uvo_point2f;

/* The parameter globals of VO_utility.h:25-89 that the hot path reads (key names: SURVEY.md 5.6,
 * loaded by get_VO_parameters, VO_utility.cpp:461-507). */
typedef struct uvo_params {
    int    DISTANCE;                                   /* VOH:47  /vo_params/distance */
    double LOWE_RATIO_THRESHOLD;                       /* VOH:65 */
    int    ESSENTIAL_OUTLIER_METHOD;                   /* VOH:52-55 */
    double ESSENTIAL_MAX_ITERS, ESSENTIAL_CONFIDENCE, ESSENTIAL_THRESHOLD;
    int    HOMOGRAPHY_OUTLIER_METHOD;                  /* VOH:57-61 */
    double HOMOGRAPHY_MAX_ITERS, HOMOGRAPHY_CONFIDENCE, HOMOGRAPHY_THRESHOLD, HOMOGRAPHY_DISTANCE;
    double VPF_THRESHOLD, REPROJECTION_TOLERANCE;      /* VOH:63-64 */
    int    MIN_NUM_FEATURES, MIN_NUM_3DPOINTS, MIN_NUM_INLIERS;   /* VOH:67-69 */
    int    ITERATIONS_COUNT;                           /* VOH:72-76 */
    double REPROJECTION_ERROR_THRESHOLD, CONFIDENCE;
    int    USE_EXTRINSIC_GUESS, PNP_METHOD_FLAG;
    int    SURF_MIN_HESSIAN, SURF_OCTAVES_NUMBER, SURF_OCTAVES_LAYERS, SURF_EXTENDED, SURF_UPRIGHT; /* VOH:83-87 */
} uvo_params;

/* uvo/config/stereo_VO_parameters.yaml / mono_VO_parameters.yaml values */
void uvo_params_default_stereo(uvo_params* p);
void uvo_params_default_mono(uvo_params* p);

/* ---- context ---- */
uvo_status  uvo_ctx_create(const uvo_params* p, int device, int max_w, int max_h, int max_kpts, uvo_ctx** out);
void        uvo_ctx_destroy(uvo_ctx* c);
const char* uvo_last_error(const uvo_ctx* c);           /* message of the last non-OK status */
void*       uvo_ctx_stream(uvo_ctx* c);                 /* the context's hipStream_t */
uvo_status  uvo_ctx_set_params(uvo_ctx* c, const uvo_params* p);
/* The hipStream_t on which the caller produces its UVO_MEM_DEVICE inputs (NULL = the default stream): while `enabled`,
 * every image upload first waits for the work queued on that stream so far.  May be changed between any two calls. */
uvo_status  uvo_ctx_set_producer_stream(uvo_ctx* c, void* hip_stream, int enabled);
/* Non-fatal advice about the process environment noticed at context creation ("" when there is none), e.g. a pipeline
 * deeper than two lanes with GPU_MAX_HW_QUEUES left at the ROCm default of 4 (lanes then share hardware queues). */
const char* uvo_ctx_warning(const uvo_ctx* c);
/* Entries submitted (uvo_stereo_submit / uvo_mono_submit, init pairs included) and not yet collected.  A collect that fails
 * with "nothing submitted" or "wrong kind" leaves it unchanged; every other collect, successful or not, dequeues one. */
int         uvo_ctx_pending(const uvo_ctx* c);
/* How this context's host side waits and who drives the PnP stage of pipelined pairs at its current depth, as text:
 * "wait=poll|timed-sleep+poll|interrupt stage_b=worker|device cpu_budget=<logical CPUs, -1 unlimited> depth=<lanes>".  Decided from
 * the process's affinity mask, the cgroup CPU quota and the environment (UVO_CPU_BUDGET = this process's share of the quota,
 * UVO_WORKER_WAIT = spin | sleep | block-all, UVO_STAGE_B = worker | device): with a CPU per lane the lanes' worker threads poll and
 * run the PnP stage; below that the first RANSAC round runs device-driven and uvo_stereo_collect confirms it, so that a rank keeps
 * one host thread busy.  Results do not depend on the choice.  Valid until the next call. */
const char* uvo_ctx_host_policy(uvo_ctx* c);

/* The reference's global FEATURE_DETECTOR (VO_utility.h:25, /vo_params/feature_detector) for the fused steps: "SURF" (default) or
 * "SIFT" -- uvo_stereo_step / submit and uvo_mono_step / submit then take detect_features' SIFT branch (VO_utility.cpp:107-112,
 * SIFT::create(10000, 3, 0.03, 10, 1.6)) and match 128-float rows (VO_utility.cpp:525-529); everything after the matcher is unchanged.
 * Refused while pairs are in flight or a VO sequence is running (reset first).  A frame with more than max_kpts SIFT keypoints is a
 * UVO_CAPACITY error, as for SURF: create the context with max_kpts >= 10000 + ties for 1080p frames. */
uvo_status uvo_ctx_set_feature_detector(uvo_ctx* c, const char* name);

/* ---- detect_features, SURF branch (VO_utility.h:100 -> VO_utility.cpp:114-119) ----
 * gray: 8-bit single channel, `stride` bytes per row.  kps/desc are host buffers of capacity `cap`, either may be NULL;
 * desc is n x 64 f32, or n x 128 when the context's SURF_EXTENDED is set.  SURF_UPRIGHT = 0 runs the orientation assignment
 * and samples the descriptor window rotated (keypoint.angle = the orientation, 270 when upright).  Overwrites outputs (as
 * detectAndCompute does).  The matcher, the gathers and the stereo / mono steps take the row width from the context's
 * SURF_EXTENDED (64 or 128 floats) end to end. */
uvo_status uvo_surf_detect(uvo_ctx* c, const uint8_t* gray, int w, int h, int stride, int mem,
                           uvo_keypoint* kps, float* desc, int cap, int* n);
/* ---- detect_features, FEATURE_DETECTOR == "SIFT" (VO_utility.cpp:107-112):
 *     SIFT::create(10000, 3, 0.03, 10, 1.6)->detectAndCompute(img, noArray(), keypoints, descriptors)
 * the five arguments of SIFT::create are passed through (nfeatures <= 0 keeps every keypoint).  gray as for uvo_surf_detect.
 * kps / desc: host buffers (desc: cap x 128 floats, the integer-valued rows cv::SIFT writes as CV_32F); keypoints come out in
 * KeyPoint_LessThan order (x, y, size, angle ...) after duplicate removal, with OpenCV's packed octave / layer / offset in
 * `octave`.  A standalone operator: the stereo / mono steps of this library run on SURF (the shipped parameter files). */
uvo_status uvo_sift_detect(uvo_ctx* c, const uint8_t* gray, int w, int h, int stride, int mem, int nfeatures, int n_octave_layers,
                           double contrast_threshold, double edge_threshold, double sigma, uvo_keypoint* kps, float* desc, int cap, int* n);
/* detect_features, FEATURE_DETECTOR == "AKAZE" (VO_utility.cpp:93-98): AKAZE::create()->detectAndCompute(img, noArray(), keypoints,
 * descriptors) -- DESCRIPTOR_MLDB at full length (486 bits: rows of 61 bytes, for uvo_match_knn2_ratio_hamming), 3 channels, threshold
 * 0.001, 4 octaves of 4 sublevels, DIFF_PM_G2.  Keypoints in OpenCV's order (evolution level by level, row-major), `size` the diameter,
 * `angle` in degrees, `class_id` the evolution level.  The non-linear scale space, the Hessian response, orientation and descriptors run
 * on the device; the sequential duplicate suppression of Find_Scale_Space_Extrema runs on the host over the candidate list.  A
 * standalone operator: the fused stereo / mono steps run on SURF or SIFT. */
uvo_status uvo_akaze_detect(uvo_ctx* c, const uint8_t* gray, int w, int h, int stride, int mem, uvo_keypoint* kps, uint8_t* desc, int cap, int* n);
/* test hook: plane `what` (0 Lt, 1 Lsmooth, 2 Lx, 3 Ly -- the multiscale derivatives --, 4 Ldet) of evolution level `level` of the last
 * uvo_akaze_detect, row-major floats to a host buffer */
uvo_status uvo_akaze_plane(uvo_ctx* c, int level, int what, float* out, int cap_floats, int* w, int* h);
/* detect_features, FEATURE_DETECTOR == "ORB" (VO_utility.cpp:100-105):
 *     ORB::create(10000, 1.2, 8, 31, 0, 2, ORB::HARRIS_SCORE, 31, 10)->detectAndCompute(img, noArray(), keypoints, descriptors)
 * -- those arguments are the context's defaults (firstLevel 0, WTA_K 2 and HARRIS_SCORE are fixed); uvo_orb_configure changes the
 * others.  Rows of 32 bytes for uvo_match_knn2_ratio_hamming.  Keypoints in OpenCV's order: level by level, within a level the order
 * KeyPointsFilter::retainBest (libstdc++'s std::nth_element + std::partition) leaves; `size` = patchSize x 1.2^level, `angle` in
 * degrees (intensity centroid), `response` the Harris measure, `octave` the level.  Pyramid (INTER_LINEAR_EXACT), FAST-9/16 scores and
 * maxima, Harris responses, angles, the 7 x 7 blur and the descriptors run on the device; the two retainBest rankings per level run
 * on the host over the response arrays.
 * THE SAMPLING TABLE IS THE CALLER'S: with patchSize 31 OpenCV reads its 256 test pairs from `bit_pattern_31_` (orb.cpp: 1024 integers
 * learned offline), which this library cannot restate and the reference does not hold.  uvo_orb_set_pattern takes it in OpenCV's own
 * layout -- x0, y0, x1, y1 per descriptor bit, |coordinate| <= patchSize / 2 -- and keeps it for the context (NULL forgets it).  Without a
 * table uvo_orb_detect returns keypoints only (desc = NULL) and refuses descriptors with UVO_INVALID_ARG.  (For patch sizes other than
 * 31 OpenCV draws the table itself with cv::RNG(0x34985739): INTEGRATION.md shows the eight lines.)
 * A standalone operator: the fused stereo / mono steps run on SURF or SIFT. */
uvo_status uvo_orb_configure(uvo_ctx* c, int nfeatures, float scale_factor, int nlevels, int edge_threshold, int patch_size, int fast_threshold);
uvo_status uvo_orb_set_pattern(uvo_ctx* c, const int* pattern /* 1024 ints or NULL */);
uvo_status uvo_orb_detect(uvo_ctx* c, const uint8_t* gray, int w, int h, int stride, int mem, uvo_keypoint* kps, uint8_t* desc, int cap, int* n);
/* test hook: level `level` of the last uvo_orb_detect as bytes to a host buffer: what = 0 the resized image (levels >= 1), 1 its blurred
 * copy (after a call with descriptors), 2 its FAST score map (0 = no corner) */
uvo_status uvo_orb_plane(uvo_ctx* c, int level, int what, uint8_t* out, int cap_bytes, int* w, int* h);
/* test hook: Gaussian (dog = 0, layers 0 .. n_octave_layers + 2) or difference (dog = 1, layers 0 .. n_octave_layers + 1) layer of
 * octave `octave` (0 = the doubled image) of the last uvo_sift_detect, row-major floats to a host buffer; out = NULL reports the size */
uvo_status uvo_sift_layer(uvo_ctx* c, int octave, int layer, int dog, float* out, int cap_floats, int* w, int* h);
/* test hooks into the detector's first stages (host outputs): integral image (h+1)x(w+1) s32 of the
 * last image given to uvo_surf_detect / uvo_integral, and one Hessian det/trace layer of it */
uvo_status uvo_integral(uvo_ctx* c, const uint8_t* gray, int w, int h, int stride, int mem, int32_t* sum);
uvo_status uvo_hessian_layer(uvo_ctx* c, int octave, int layer, float* det, float* trace);

/* ---- match_features (VO_utility.h:109-110 -> VO_utility.cpp:515-573): BFMatcher(NORM_L2).knnMatch
 * k=2 + Lowe ratio.  d1: n1 x 64, d2: n2 x 64 (f32, `mem`).  Matches are APPENDED at out[*m]
 * (the reference's vector is appended to, VO_utility.cpp:538); out is a host buffer. */
uvo_status uvo_match_knn2_ratio(uvo_ctx* c, const float* d1, int n1, const float* d2, int n2, int mem,
                                float ratio, uvo_dmatch* out, int cap, int* m);
/* raw 2-NN (host outputs idx[2*n1], dist[2*n1]; idx = -1 when the train set is too small) */
uvo_status uvo_match_knn2(uvo_ctx* c, const float* d1, int n1, const float* d2, int n2, int mem,
                          int* idx, float* dist);
/* The same two operators on rows of `dim` floats (64 or 128) whatever the context's SURF_EXTENDED says: match_features sends
 * FEATURE_DETECTOR == "SIFT" descriptors (128 floats per row) to the same BFMatcher(NORM_L2) (VO_utility.cpp:525-529). */
uvo_status uvo_match_knn2_ratio_dim(uvo_ctx* c, const float* d1, int n1, const float* d2, int n2, int dim, int mem,
                                    float ratio, uvo_dmatch* out, int cap, int* m);
uvo_status uvo_match_knn2_dim(uvo_ctx* c, const float* d1, int n1, const float* d2, int n2, int dim, int mem,
                              int* idx, float* dist);

/* The AKAZE / ORB branch of the same function (VO_utility.cpp:520-524): BFMatcher(NORM_HAMMING).knnMatch k=2 + Lowe ratio on binary
 * descriptors of `bytes` bytes per row (1..64: ORB 32, AKAZE 61), u8, `mem`.  DMatch::distance = the number of differing bits.
 * (Rows from uvo_akaze_detect / uvo_orb_detect, or from the caller's own detector.) */
uvo_status uvo_match_knn2_ratio_hamming(uvo_ctx* c, const uint8_t* d1, int n1, const uint8_t* d2, int n2, int bytes, int mem,
                                        float ratio, uvo_dmatch* out, int cap, int* m);
uvo_status uvo_match_knn2_hamming(uvo_ctx* c, const uint8_t* d1, int n1, const uint8_t* d2, int n2, int bytes, int mem,
                                  int* idx, float* dist);

/* ---- cv::triangulatePoints as called at visual_odometry.h:355, 631 and VO_utility.cpp:595 ----
 * P1, P2: 3x4 f64 row-major; x1, x2: n Point2f (host); out: 4 x n f32 (host). */
uvo_status uvo_triangulate_points(uvo_ctx* c, const double* P1, const double* P2,
                                  const uvo_point2f* x1, const uvo_point2f* x2, int n, float* out4xn);

/* ---- extract_3Dpoints (VO_utility.h:102 -> VO_utility.cpp:188-237), thresholds from the params ----
 * pts: g x 3 f64, idx: g s32 (host, capacity n). */
uvo_status uvo_extract_3d_points(uvo_ctx* c, const uvo_point2f* k1, const uvo_point2f* k2, int n,
                                 const double* R1, const double* t1, const double* R2, const double* t2,
                                 const double* K1, const double* K2, const float* points4d,
                                 double* pts, int* idx, int* g);

/* ---- reproject_errors (VO_utility.h:112 -> VO_utility.cpp:632-651): cv::projectPoints without distortion,
 * then the pixel distance to img[i].  world: n x 3 f64, R: 3x3 f64, t: 3, K: 3x3 (host); err: n f64 (host). */
uvo_status uvo_reproject_errors(uvo_ctx* c, const double* world, int n, const double* R, const double* t,
                                const double* K, const uvo_point2f* img, double* err);

/* ---- cv::solvePnPRansac(..., SOLVEPNP_EPNP) as called at visual_odometry.h:647-648 ----
 * obj: n x 3 f64, img: n Point2f, K 3x3 f64 (host).  inliers: host, capacity n, ascending.
 * *ok is OpenCV's bool return. */
uvo_status uvo_solve_pnp_ransac(uvo_ctx* c, const double* obj, const uvo_point2f* img, int n, const double* K,
                                int iterations_count, float reprojection_error, double confidence,
                                double* rvec, double* tvec, int* inliers, int* n_inliers, int* ok);

/* cv::Rodrigues (visual_odometry.h:673): n_in = 3 (vector -> 3x3) or 9 (3x3 -> vector) */
uvo_status uvo_rodrigues(const double* in, int n_in, double* out);

/* ---- stereo loop body, visual_odometry_node::stereo_VO (visual_odometry.h:474-520 init,
 * 531-739 main loop, 148-159 output) with all intermediates kept on the device ---- */
typedef struct {
    int    valid;                 /* successful_estimate.data */
    int    initialized;           /* 0 for steps consumed by the init loop */
    int    n_left, n_right, n_stereo_matches, n_tri_matches, n_good3d, n_inliers;
    double rvec[3], tvec[3];      /* R_currCam_prevCam_Vec, t_currCam_prevCam */
    double t_prev_curr[3];        /* t_prevCam_currCam (kept from the last valid step on failure) */
    double velocity[3];           /* t_prevCam_currCam / dt */
} uvo_stereo_result;

/* new_camera_matrix_left/right, R_right, t_right (visual_odometry.h:447-463) */
uvo_status uvo_stereo_set_rig(uvo_ctx* c, const double* K_left, const double* K_right,
                              const double* R_right, const double* t_right);
uvo_status uvo_stereo_reset(uvo_ctx* c);
uvo_status uvo_stereo_step(uvo_ctx* c, const uint8_t* left, const uint8_t* right, int w, int h, int stride,
                           int mem, double dt, uvo_stereo_result* out);
/* The same step split for throughput: uvo_stereo_submit enqueues the device work of a pair up to
 * extract_3Dpoints on its pipeline lane and returns at once; the lane's worker thread runs PnP-RANSAC as soon as
 * that work finishes; uvo_stereo_collect returns the result of the OLDEST submitted pair.  Up to `depth` pairs may
 * be in flight (uvo_stereo_set_depth), so the detector of pairs k+1.. overlaps the pose solve of pair k; results
 * are identical to uvo_stereo_step's.  Submit paces the pipeline: it may block until the detector stage of the pair submitted two
 * pairs earlier has drained (at most two detector stages run side by side).  The pairs consumed by the init loop (VO:474-520) run synchronously inside
 * uvo_stereo_submit; their results queue like any other, so a caller need not know when the loop initialises.
 * Pipeline depth: 1..16, default 2 (the stereo loop is fastest at 6, the mono loop at about 14).  Each unit of depth is one more set of device buffers, two more HIP streams and
 * one more host worker thread; changing it restarts nothing unless the lane holding the previous pair is removed. */
uvo_status uvo_stereo_set_depth(uvo_ctx* c, int depth);
uvo_status uvo_stereo_submit(uvo_ctx* c, const uint8_t* left, const uint8_t* right, int w, int h, int stride, int mem);
uvo_status uvo_stereo_collect(uvo_ctx* c, double dt, uvo_stereo_result* out);
/* intermediates of the pair returned by the last uvo_stereo_step / uvo_stereo_collect, read from its lane (valid until the
 * next uvo_stereo_submit reuses that lane, i.e. call it right after the collect): "kps_left", "kps_right", "desc_left", "desc_right",
 * "matches_stereo", "matches_tri", "points4d", "good_pts", "good_idx", "inliers".
 * Returns the element count, or -(count) if cap_bytes is too small. */
int        uvo_stereo_get(uvo_ctx* c, const char* what, void* out, int cap_bytes);

/* ---- mono relative pose: estimate_relative_pose (VO_utility.h:101 -> VO_utility.cpp:134-180) and the OpenCV calls
 * it makes.  Points are host Point2f arrays, K 3x3 f64, masks n bytes (0/1), methods 4 = LMEDS, 8 = RANSAC. ---- */
/* cv::findEssentialMat(p1, p2, K, method, prob, threshold, maxIters, mask)  (VO_utility.cpp:147); *ok = 0: OpenCV's empty E */
uvo_status uvo_find_essential_mat(uvo_ctx* c, const uvo_point2f* p1, const uvo_point2f* p2, int n, const double* K, int method,
                                  double prob, double threshold, int max_iters, double* E, uint8_t* mask, int* ok);
/* cv::recoverPose(E, p1, p2, K, R, t, mask), distance threshold 50 (VO_utility.cpp:149); mask is in/out */
uvo_status uvo_recover_pose(uvo_ctx* c, const double* E, const uvo_point2f* p1, const uvo_point2f* p2, int n, const double* K,
                            double* R, double* t, uint8_t* mask, int* good);
/* cv::findHomography(p1, p2, method, threshold, mask, maxIters, confidence)  (VO_utility.cpp:152) */
uvo_status uvo_find_homography(uvo_ctx* c, const uvo_point2f* p1, const uvo_point2f* p2, int n, int method, double threshold,
                               int max_iters, double confidence, double* H, uint8_t* mask, int* ok);
/* cv::decomposeHomographyMat(H, K, Rs, ts, ns)  (VO_utility.cpp:585): up to 4 solutions, Rs 4x9, ts 4x3, ns 4x3 (host) */
uvo_status uvo_decompose_homography_mat(const double* H, const double* K, double* Rs, double* ts, double* ns, int* n_solutions);
/* recover_pose_homography (VO_utility.h:111 -> VO_utility.cpp:581-624); R, t written only when a candidate wins */
uvo_status uvo_recover_pose_homography(uvo_ctx* c, const double* H, const uvo_point2f* p1, const uvo_point2f* p2, int n,
                                       const double* K, double* R, double* t, int* max_good);
/* select_estimation_method (VO_utility.h:116 -> VO_utility.cpp:725-748): 1 = essential, 0 = homography; -1 = invalid arguments or
 * no host memory for the median's scratch (the entry has no context to carry a status) */
int        uvo_select_estimation_method(const uvo_point2f* k1, const uvo_point2f* k2, int n, int distance);
/* estimate_relative_pose: *use_essential is the reference's global of that name (in/out), R and t are in/out,
 * in1/in2 (capacity n) receive extract_inliers' output, mask the final mask (VO_utility.cpp:157) */
uvo_status uvo_estimate_relative_pose(uvo_ctx* c, const uvo_point2f* k1, const uvo_point2f* k2, int n, const double* K,
                                      int* use_essential, double* R, double* t, uvo_point2f* in1, uvo_point2f* in2, int* n_in,
                                      uint8_t* mask, int* success);

/* ---- mono loop body, visual_odometry_node::mono_VO (visual_odometry.h:227-245 init, 247-397 main loop, 126-140 output) ---- */
typedef struct {
    int    published;             /* 0: frame skipped with `continue` (visual_odometry.h:276-307) -- nothing is published */
    int    valid, initialized, used_essential, success;
    int    n_kps, n_matches, n_inliers, n_good3d, n_front;
    double R[9], t[3], SF, velocity[3];
} uvo_mono_result;
uvo_status uvo_mono_set_camera(uvo_ctx* c, const double* K);     /* new_camera_matrix (visual_odometry.h:221-222) */
uvo_status uvo_mono_reset(uvo_ctx* c);
uvo_status uvo_mono_step(uvo_ctx* c, const uint8_t* img, int w, int h, int stride, int mem, double range, double dt,
                         uvo_mono_result* out);
/* The same loop body, pipelined like uvo_stereo_submit / uvo_stereo_collect (uvo_stereo_set_depth sets the number of lanes):
 * submit enqueues detection and the matching against the previous frame on a lane and returns; the lane's worker runs the
 * host-orchestrated pose stage (method selection, RANSAC / LMedS, recoverPose, triangulation, scale) beside the next frames'
 * detection; collect returns the oldest frame's result and applies the sequential state (R, t kept when no estimate wrote
 * them; the scale factor) in order.  Results equal uvo_mono_step's.  The first frames (until one has MIN_NUM_FEATURES
 * keypoints) are processed synchronously inside submit. */
uvo_status uvo_mono_submit(uvo_ctx* c, const uint8_t* img, int w, int h, int stride, int mem, double range);
uvo_status uvo_mono_collect(uvo_ctx* c, double dt, uvo_mono_result* out);
/* last step's intermediates: "kps", "matches", "mask", "good_pts" */
int        uvo_mono_get(uvo_ctx* c, const char* what, void* out, int cap_bytes);

/* ---- get_image (VO_utility.h:105 -> VO_utility.cpp:337-379), the preprocessing in front of detect_features:
 * cv::resize(INTER_AREA) to desired_width x (int)(h / (w / desired_width)) (skipped when the size already matches),
 * cv::cvtColor(COLOR_RGB2GRAY), cv::undistort(K, dist, newK), and cv::CLAHE(clip_limit, 8x8 tiles) when `clahe` != 0
 * (the reference's globals DESIRED_WIDTH, CLAHE_CORRECTION, CLIP_LIMIT, VO_utility.h:41-44).
 * rgb: h x w x 3 interleaved u8, `stride` bytes per row, host or device per `mem`; dist4 = (k1, k2, p1, p2);
 * out: tight-pitch u8 written to host or device memory per `out_mem` (feed it to uvo_surf_detect / uvo_stereo_submit with
 * UVO_MEM_DEVICE to keep the frame in HBM).  The undistortion maps are cached per (K, dist, newK, size). */
uvo_status uvo_get_image(uvo_ctx* c, const uint8_t* rgb, int w, int h, int stride, int mem, const double* K, const double* dist4,
                         const double* newK, int desired_width, int clahe, int clip_limit, uint8_t* out, int out_mem,
                         int* out_w, int* out_h);

/* ---- compressed-image ingest: from_ros_to_cv_image (math_utility.h -> uvo_libraries/src/math_utility.cpp:154-173) =
 * cv_bridge::toCvCopy(sensor_msgs/CompressedImage) -> cv::imdecode, then cv::cvtColor(COLOR_BayerBGGR2BGR) when the message's
 * `format` contains "bayer".  data: the message payload, recognised by its signature as cv::imdecode does:
 *   JPEG  baseline / sequential Huffman, 8 bit, 1 or 3 components, sampling factors <= 2 (progressive and arithmetic-coded files are
 *         refused with UVO_INVALID_ARG).  Entropy decoding on the host; dequantisation + IDCT + chroma upsampling + colour
 *         conversion on the device, byte-identical to libjpeg's defaults (JDCT_ISLOW, fancy upsampling).
 *   PNG   non-interlaced; grey and palette at 1 / 2 / 4 / 8 bits, RGB and RGBA at 8 bits -> 1, 3, 3, 4 channels as
 *         imdecode(IMREAD_UNCHANGED) returns them (16-bit samples, grey + alpha, palettes with tRNS and Adam7 are refused).  zlib
 *         inflate and the scanline filters on the host, sample expansion and channel order on the device.
 * A "bayer" message is demosaiced on the device after decoding.  out: h x w x channels u8 (B G R [A] order) in host or device
 * memory per out_mem; out = NULL only reports the size (headers only, nothing is decoded).  Feed the result to uvo_get_image. */
uvo_status uvo_decode_image(uvo_ctx* c, const uint8_t* data, size_t n, const char* format, uint8_t* out, size_t cap_bytes, int out_mem,
                            int* w, int* h, int* channels);
/* cv::cvtColor(src, dst, COLOR_BayerBGGR2BGR) of an 8-bit mosaic (bilinear; borders copy their neighbour) */
uvo_status uvo_bayer_bggr2bgr(uvo_ctx* c, const uint8_t* bayer, int w, int h, int stride, int mem, uint8_t* out_bgr, int out_mem);

/* ---- resize_camera_matrix (VO_utility.h:112 -> VO_utility.cpp:658-675), once per run, host arithmetic only (no context):
 * K (3x3 row-major, in/out) is divided by ratio = original_width / desired_width with the skew K[0][1] kept and K[2][2] = 1;
 * newK (3x3 out) = cv::getOptimalNewCameraMatrix(K, dist4, Size(desired_width, desired_height), alpha = 0, same size), where
 * desired_height = (int)(original_height / ratio) is also returned.  dist4 = (k1, k2, p1, p2). */
uvo_status uvo_resize_camera_matrix(int original_width, int original_height, int desired_width, double* K, const double* dist4,
                                    double* newK, int* desired_height);

/* ---- per-stage device timing (HIP events on the context's stream) for bench.py ---- */
uvo_status uvo_timing_enable(uvo_ctx* c, int on);
int        uvo_timing_count(uvo_ctx* c);
const char* uvo_timing_name(uvo_ctx* c, int i);
/* accumulated milliseconds and number of launches of stage i since the last reset */
uvo_status uvo_timing_get(uvo_ctx* c, int i, double* ms, long long* launches);
uvo_status uvo_timing_reset(uvo_ctx* c);


/* ---- pipeline trace (diagnostics; bench.py reports it for short runs): per pipelined pair, when its phases ran.
 * dev_ms: HIP-event times on the lane's streams -- stage A begins (the pair's first kernel may start), detector done, stage A done
 * (VO:548-632), PnP stage begins, hypotheses scored, PnP stage done (VO:647-648) -- in ms after the first traced pair's stage-A
 * begin; the last three are -1 when the pair's gates skipped solvePnPRansac.  host_ms: the submitting thread entered
 * uvo_stereo_submit, its pacing wait ended, it returned; the lane's worker saw stage A's end, got a PnP slot, finished -- steady
 * clock, ms after the first traced pair's submit (-1: not reached).  dev_ms[6], [7]: begin and end of the pair's detection launch
 * (k_hessian_nms_all; bench.py's roofline.frac_pipelined).  The ring keeps the last 256 pairs per lane. */
typedef struct uvo_trace_row { long long pair; int lane; int b_used; float dev_ms[8]; double host_ms[6]; } uvo_trace_row;
uvo_status uvo_trace_enable(uvo_ctx* c, int on);           /* not while pairs are in flight; enabling clears the ring */
int        uvo_trace_read(uvo_ctx* c, uvo_trace_row* rows, int cap);   /* waits for the device; rows sorted by pair; returns the
                                                                         number of traced pairs (may exceed cap), -1 on misuse */

#ifdef __cplusplus
}
#endif
#endif
