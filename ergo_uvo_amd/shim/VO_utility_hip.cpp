// VO_utility_hip.cpp -- the uvo_libraries function surface on top of the C ABI of libuvo_hip.so.
//
// Each function marshals cv-style containers into the plain buffers of include/uvo_hip.h, calls the device path and
// marshals back.  There is no arithmetic of the hot path here and no CPU fallback: without a GPU the first call
// throws uvo_hip::Error(UVO_NO_DEVICE).  Reference: uvo_libraries/src/VO_utility.cpp (cited as VOU).
#include "uvo_libraries_hip/VO_utility_hip.h"

#include <new>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>

using namespace uvocv;
using std::vector;

// ---- parameter globals; defaults are the stereo launch file's values (uvo_params_default_stereo) ---------------
std::string FEATURE_DETECTOR = "SURF";
int    DESIRED_WIDTH = 640;        bool CLAHE_CORRECTION = true;   int CLIP_LIMIT = 8;          // uvo/config/stereo_VO_parameters.yaml
int    DISTANCE;
int    ESSENTIAL_OUTLIER_METHOD;   double ESSENTIAL_MAX_ITERS, ESSENTIAL_CONFIDENCE, ESSENTIAL_THRESHOLD;
int    HOMOGRAPHY_OUTLIER_METHOD;  double HOMOGRAPHY_MAX_ITERS, HOMOGRAPHY_CONFIDENCE, HOMOGRAPHY_THRESHOLD, HOMOGRAPHY_DISTANCE;
double VPF_THRESHOLD, REPROJECTION_TOLERANCE, LOWE_RATIO_THRESHOLD;
int    MIN_NUM_FEATURES, MIN_NUM_3DPOINTS, MIN_NUM_INLIERS;
int    ITERATIONS_COUNT;           double REPROJECTION_ERROR_THRESHOLD, CONFIDENCE;
bool   USE_EXTRINSIC_GUESS;        int PNP_METHOD_FLAG;
int    SURF_MIN_HESSIAN, SURF_OCTAVES_NUMBER, SURF_OCTAVES_LAYERS;
bool   SURF_EXTENDED, SURF_UPRIGHT;
bool   use_essential = true;

namespace {

void globals_from(const uvo_params& p)
{
    DISTANCE = p.DISTANCE; LOWE_RATIO_THRESHOLD = p.LOWE_RATIO_THRESHOLD;
    ESSENTIAL_OUTLIER_METHOD = p.ESSENTIAL_OUTLIER_METHOD; ESSENTIAL_MAX_ITERS = p.ESSENTIAL_MAX_ITERS;
    ESSENTIAL_CONFIDENCE = p.ESSENTIAL_CONFIDENCE; ESSENTIAL_THRESHOLD = p.ESSENTIAL_THRESHOLD;
    HOMOGRAPHY_OUTLIER_METHOD = p.HOMOGRAPHY_OUTLIER_METHOD; HOMOGRAPHY_MAX_ITERS = p.HOMOGRAPHY_MAX_ITERS;
    HOMOGRAPHY_CONFIDENCE = p.HOMOGRAPHY_CONFIDENCE; HOMOGRAPHY_THRESHOLD = p.HOMOGRAPHY_THRESHOLD;
    HOMOGRAPHY_DISTANCE = p.HOMOGRAPHY_DISTANCE; VPF_THRESHOLD = p.VPF_THRESHOLD;
    REPROJECTION_TOLERANCE = p.REPROJECTION_TOLERANCE; MIN_NUM_FEATURES = p.MIN_NUM_FEATURES;
    MIN_NUM_3DPOINTS = p.MIN_NUM_3DPOINTS; MIN_NUM_INLIERS = p.MIN_NUM_INLIERS; ITERATIONS_COUNT = p.ITERATIONS_COUNT;
    REPROJECTION_ERROR_THRESHOLD = p.REPROJECTION_ERROR_THRESHOLD; CONFIDENCE = p.CONFIDENCE;
    USE_EXTRINSIC_GUESS = p.USE_EXTRINSIC_GUESS != 0; PNP_METHOD_FLAG = p.PNP_METHOD_FLAG;
    SURF_MIN_HESSIAN = p.SURF_MIN_HESSIAN; SURF_OCTAVES_NUMBER = p.SURF_OCTAVES_NUMBER;
    SURF_OCTAVES_LAYERS = p.SURF_OCTAVES_LAYERS; SURF_EXTENDED = p.SURF_EXTENDED != 0; SURF_UPRIGHT = p.SURF_UPRIGHT != 0;
}
uvo_params globals_to()
{
    uvo_params p;
    memset(&p, 0, sizeof(p));
    p.DISTANCE = DISTANCE; p.LOWE_RATIO_THRESHOLD = LOWE_RATIO_THRESHOLD;
    p.ESSENTIAL_OUTLIER_METHOD = ESSENTIAL_OUTLIER_METHOD; p.ESSENTIAL_MAX_ITERS = ESSENTIAL_MAX_ITERS;
    p.ESSENTIAL_CONFIDENCE = ESSENTIAL_CONFIDENCE; p.ESSENTIAL_THRESHOLD = ESSENTIAL_THRESHOLD;
    p.HOMOGRAPHY_OUTLIER_METHOD = HOMOGRAPHY_OUTLIER_METHOD; p.HOMOGRAPHY_MAX_ITERS = HOMOGRAPHY_MAX_ITERS;
    p.HOMOGRAPHY_CONFIDENCE = HOMOGRAPHY_CONFIDENCE; p.HOMOGRAPHY_THRESHOLD = HOMOGRAPHY_THRESHOLD;
    p.HOMOGRAPHY_DISTANCE = HOMOGRAPHY_DISTANCE; p.VPF_THRESHOLD = VPF_THRESHOLD;
    p.REPROJECTION_TOLERANCE = REPROJECTION_TOLERANCE; p.MIN_NUM_FEATURES = MIN_NUM_FEATURES;
    p.MIN_NUM_3DPOINTS = MIN_NUM_3DPOINTS; p.MIN_NUM_INLIERS = MIN_NUM_INLIERS; p.ITERATIONS_COUNT = ITERATIONS_COUNT;
    p.REPROJECTION_ERROR_THRESHOLD = REPROJECTION_ERROR_THRESHOLD; p.CONFIDENCE = CONFIDENCE;
    p.USE_EXTRINSIC_GUESS = USE_EXTRINSIC_GUESS; p.PNP_METHOD_FLAG = PNP_METHOD_FLAG;
    p.SURF_MIN_HESSIAN = SURF_MIN_HESSIAN; p.SURF_OCTAVES_NUMBER = SURF_OCTAVES_NUMBER;
    p.SURF_OCTAVES_LAYERS = SURF_OCTAVES_LAYERS; p.SURF_EXTENDED = SURF_EXTENDED; p.SURF_UPRIGHT = SURF_UPRIGHT;
    return p;
}
struct GlobalsInit { GlobalsInit() { uvo_params p; uvo_params_default_stereo(&p); globals_from(p); } } g_globals_init;

struct State {
    std::mutex mu;
    uvo_ctx* ctx = nullptr;
    uvo_params applied;
    int device = 0, max_w = 1920, max_h = 1200, max_kpts = 8192;
    std::vector<int> orb_pattern;                 // OpenCV's bit_pattern_31_ (set_orb_pattern / UVO_ORB_PATTERN_FILE); empty = not supplied
    bool orb_pattern_sent = false;                // ... and handed to the current context
} g;

[[noreturn]] void raise(uvo_status st, const char* where)
{
    const char* msg = g.ctx ? uvo_last_error(g.ctx) : "";
    throw uvo_hip::Error(st, std::string(where) + ": uvo_status " + std::to_string((int)st) + (msg && *msg ? std::string(" (") + msg + ")" : ""));
}
#define SHIM_TRY(expr, where) do { uvo_status st_ = (expr); if (st_ != UVO_OK) raise(st_, where); } while (0)

// the context with the current value of the parameter globals applied
uvo_ctx* ctx_now()
{
    std::lock_guard<std::mutex> lk(g.mu);
    uvo_params p = globals_to();
    if (!g.ctx) {
        uvo_status st = uvo_ctx_create(&p, g.device, g.max_w, g.max_h, g.max_kpts, &g.ctx);
        if (st != UVO_OK) { g.ctx = nullptr; raise(st, "uvo_ctx_create"); }
        g.applied = p;
    } else if (memcmp(&p, &g.applied, sizeof(p)) != 0) {
        SHIM_TRY(uvo_ctx_set_params(g.ctx, &p), "uvo_ctx_set_params");
        g.applied = p;
    }
    return g.ctx;
}

void require(bool cond, const char* what) { if (!cond) throw uvo_hip::Error(UVO_INVALID_ARG, what); }

// 3x3 / 3x1 / 3x4 CV_64F -> tight row-major doubles
void doubles_of(const Mat& m, int rows, int cols, double* out, const char* what)
{
    require(!m.empty() && m.type() == CV_64FC1 && m.rows * m.cols == rows * cols, what);
    int k = 0;
    for (int i = 0; i < m.rows; i++) for (int j = 0; j < m.cols; j++) out[k++] = m.at<double>(i, j);
}
const uvo_point2f* pts_of(const vector<Point2f>& v) { return reinterpret_cast<const uvo_point2f*>(v.data()); }
uvo_point2f* pts_of(vector<Point2f>& v) { return reinterpret_cast<uvo_point2f*>(v.data()); }
static_assert(sizeof(Point2f) == sizeof(uvo_point2f) && sizeof(KeyPoint) == sizeof(uvo_keypoint) && sizeof(DMatch) == sizeof(uvo_dmatch),
              "cv value types must match the C ABI's PODs");

// N x 3 CV_64F (continuous or not) -> tight
vector<double> rows3_of(const Mat& m, const char* what)
{
    require(m.empty() || (m.type() == CV_64FC1 && m.cols == 3), what);
    vector<double> v((size_t)m.rows * 3);
    for (int i = 0; i < m.rows; i++) for (int j = 0; j < 3; j++) v[(size_t)i * 3 + j] = m.at<double>(i, j);
    return v;
}
Mat mat_from(const double* v, int rows, int cols)
{
    Mat m(rows, cols, CV_64FC1);
    for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) m.at<double>(i, j) = v[i * cols + j];
    return m;
}
double median_of(vector<double> v)                       // compute_median (math_utility.cpp:11-27)
{
    if (v.empty()) return 0.0;
    const size_t n = v.size();
    std::sort(v.begin(), v.end());
    return n % 2 == 0 ? (v[n/2 - 1] + v[n/2]) / 2.0 : v[n/2];
}

}  // namespace

// ================================================================================================================
namespace uvo_hip {

void configure(int device, int max_w, int max_h, int max_kpts)
{
    shutdown();
    std::lock_guard<std::mutex> lk(g.mu);
    g.device = device; g.max_w = max_w; g.max_h = max_h; g.max_kpts = max_kpts;
}
uvo_ctx* context() { return ctx_now(); }
void shutdown()
{
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.ctx) { uvo_ctx_destroy(g.ctx); g.ctx = nullptr; }
    g.orb_pattern_sent = false;
}
void set_orb_pattern(const int* pattern1024)
{
    std::lock_guard<std::mutex> lk(g.mu);
    if (pattern1024) g.orb_pattern.assign(pattern1024, pattern1024 + 1024); else g.orb_pattern.clear();
    g.orb_pattern_sent = false;
}

void triangulatePoints(const Mat& P1, const Mat& P2, const vector<Point2f>& x1, const vector<Point2f>& x2, Mat& points4D)
{
    double p1[12], p2[12];
    doubles_of(P1, 3, 4, p1, "triangulatePoints: projMatr1 must be 3x4 CV_64F");
    doubles_of(P2, 3, 4, p2, "triangulatePoints: projMatr2 must be 3x4 CV_64F");
    require(x1.size() == x2.size(), "triangulatePoints: point counts differ");
    const int n = (int)x1.size();
    points4D.create(4, n, CV_32FC1);
    if (n == 0) return;
    vector<float> out((size_t)4 * n);
    SHIM_TRY(uvo_triangulate_points(ctx_now(), p1, p2, pts_of(x1), pts_of(x2), n, out.data()), "uvo_triangulate_points");
    for (int r = 0; r < 4; r++) memcpy(points4D.ptr<float>(r), out.data() + (size_t)r * n, sizeof(float) * n);
}

bool solvePnPRansac(const Mat& objectPoints, const vector<Point2f>& imagePoints, const Mat& cameraMatrix, const Mat& distCoeffs,
                    Mat& rvec, Mat& tvec, bool useExtrinsicGuess, int iterationsCount, float reprojectionError, double confidence,
                    Mat& inliers, int flags)
{
    (void)useExtrinsicGuess;                              // SOLVEPNP_EPNP ignores the guess (calib3d solvepnp.cpp)
    require(flags == 1, "solvePnPRansac: only SOLVEPNP_EPNP (1) is provided (visual_odometry.h:647-648)");
    if (!distCoeffs.empty())
        for (int i = 0; i < distCoeffs.rows; i++) for (int j = 0; j < distCoeffs.cols; j++)
            require(distCoeffs.at<double>(i, j) == 0.0, "solvePnPRansac: distortion must be zero (images are undistorted by get_image)");
    double K[9];
    doubles_of(cameraMatrix, 3, 3, K, "solvePnPRansac: cameraMatrix must be 3x3 CV_64F");
    vector<double> obj = rows3_of(objectPoints, "solvePnPRansac: objectPoints must be N x 3 CV_64F");
    const int n = objectPoints.rows;
    require(n == (int)imagePoints.size(), "solvePnPRansac: point counts differ");
    double rv[3] = {0, 0, 0}, tv[3] = {0, 0, 0};
    vector<int> inl((size_t)(n > 0 ? n : 1));
    int ni = 0, ok = 0;
    SHIM_TRY(uvo_solve_pnp_ransac(ctx_now(), obj.data(), pts_of(imagePoints), n, K, iterationsCount, reprojectionError, confidence,
                                  rv, tv, inl.data(), &ni, &ok), "uvo_solve_pnp_ransac");
    if (ok) { rvec = mat_from(rv, 3, 1); tvec = mat_from(tv, 3, 1); }
    inliers.create(ni, 1, CV_32SC1);
    for (int i = 0; i < ni; i++) inliers.at<int>(i, 0) = inl[i];
    return ok != 0;
}

void Rodrigues(const Mat& src, Mat& dst)
{
    require(!src.empty() && src.type() == CV_64FC1 && (src.rows * src.cols == 3 || (src.rows == 3 && src.cols == 3)),
            "Rodrigues: 3x1, 1x3 or 3x3 CV_64F expected");
    double in[9], out[9];
    const int nin = src.rows * src.cols;
    doubles_of(src, src.rows, src.cols, in, "Rodrigues");
    SHIM_TRY(uvo_rodrigues(in, nin, out), "uvo_rodrigues");
    dst = nin == 3 ? mat_from(out, 3, 3) : mat_from(out, 3, 1);
}

}  // namespace uvo_hip

// ================================================================================================================
// VOU:9-15: K * [R | t]; each entry is the 3-term dot product in index order (cv::gemm on 3x3 * 3x4)
Mat compute_projection_matrix(const Mat& R, const Mat& t, const Mat& cameraIntrinsic)
{
    double r[9], tt[3], K[9], P[12];
    doubles_of(R, 3, 3, r, "compute_projection_matrix: R must be 3x3 CV_64F");
    doubles_of(t, 3, 1, tt, "compute_projection_matrix: t must be 3x1 CV_64F");
    doubles_of(cameraIntrinsic, 3, 3, K, "compute_projection_matrix: K must be 3x3 CV_64F");
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += K[i*3 + k] * (j < 3 ? r[k*3 + j] : tt[k]);
        P[i*4 + j] = s;
    }
    return mat_from(P, 3, 4);
}

// VOU:23-38: distance / median(z row)
double compute_scale_factor(float distance, const Mat& world_points)
{
    if (world_points.empty() || world_points.rows < 3) return 0.0;
    vector<double> z((size_t)world_points.cols);
    for (int j = 0; j < world_points.cols; j++) z[j] = world_points.at<double>(2, j);
    return distance / median_of(z);
}

// VOU:46-63: keep the ORIGINAL rows whose transformed z is positive; result is 3 x M (empty when none)
Mat convert_3Dpoints_camera(const Mat& points_to_convert, const Mat& R_to_from, const Mat& t_to_from)
{
    double R[9], t[3];
    doubles_of(R_to_from, 3, 3, R, "convert_3Dpoints_camera: R must be 3x3 CV_64F");
    doubles_of(t_to_from, 3, 1, t, "convert_3Dpoints_camera: t must be 3x1 CV_64F");
    vector<double> P = rows3_of(points_to_convert, "convert_3Dpoints_camera: points must be N x 3 CV_64F");
    vector<int> keep;
    for (int i = 0; i < points_to_convert.rows; i++) {
        const double* q = &P[(size_t)i * 3];
        double zt = (R[6]*q[0] + R[7]*q[1] + R[8]*q[2]) * 1.0 + t[2] * 1.0;       // transform_coordinates: gemm(R, p, 1, t, 1)
        if (zt > 0) keep.push_back(i);
    }
    if (keep.empty()) return Mat();
    Mat out(3, (int)keep.size(), CV_64FC1);
    for (size_t j = 0; j < keep.size(); j++) for (int r = 0; r < 3; r++) out.at<double>(r, (int)j) = P[(size_t)keep[j] * 3 + r];
    return out;
}

// VOU:71-83: columns divided by their 4th entry (float)
Mat convert_from_homogeneous_coords(const Mat& points4d)
{
    require(points4d.rows == 4 && points4d.type() == CV_32FC1, "convert_from_homogeneous_coords: 4 x N CV_32F expected");
    Mat out(3, points4d.cols, CV_32FC1);
    for (int j = 0; j < points4d.cols; j++) {
        float w = points4d.at<float>(3, j);
        for (int r = 0; r < 3; r++) out.at<float>(r, j) = points4d.at<float>(r, j) / w;
    }
    return out;
}

// VOU:658-675: cameraMatrix /= ratio (skew kept, K[2][2] = 1); newCamMatrix = getOptimalNewCameraMatrix(K, dist, size, 0, size, 0)
void resize_camera_matrix(Mat original_image, Mat& cameraMatrix, Mat distortionCoeff, Mat& newCamMatrix)
{
    require(!original_image.empty(), "resize_camera_matrix: empty image");
    double K[9], newK[9], d4[4] = {0, 0, 0, 0};
    doubles_of(cameraMatrix, 3, 3, K, "resize_camera_matrix: cameraMatrix must be 3x3 CV_64F");
    require(!distortionCoeff.empty() && distortionCoeff.type() == CV_64FC1 && distortionCoeff.rows * distortionCoeff.cols == 4,
            "resize_camera_matrix: distortion must be (k1, k2, p1, p2) CV_64F");
    for (int i = 0, k = 0; i < distortionCoeff.rows; i++) for (int j = 0; j < distortionCoeff.cols; j++) d4[k++] = distortionCoeff.at<double>(i, j);
    int dh = 0;
    SHIM_TRY(uvo_resize_camera_matrix(original_image.cols, original_image.rows, DESIRED_WIDTH, K, d4, newK, &dh), "uvo_resize_camera_matrix");
    if (cameraMatrix.empty() || cameraMatrix.type() != CV_64FC1) cameraMatrix = Mat(3, 3, CV_64FC1);
    newCamMatrix = Mat(3, 3, CV_64FC1);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { cameraMatrix.at<double>(i, j) = K[3*i + j]; newCamMatrix.at<double>(i, j) = newK[3*i + j]; }
}

// VOU:337-379: resize INTER_AREA to DESIRED_WIDTH -> RGB2GRAY -> undistort -> CLAHE(CLIP_LIMIT) when CLAHE_CORRECTION
Mat get_image(const Mat& current_img, const Mat& cameraMatrix, const Mat& distortionCoeff, const Mat& newCamMatrix)
{
    require(!current_img.empty() && current_img.type() == CV_8UC3, "get_image: CV_8UC3 image expected");
    double K[9], newK[9], d4[4] = {0, 0, 0, 0};
    doubles_of(cameraMatrix, 3, 3, K, "get_image: cameraMatrix must be 3x3 CV_64F");
    doubles_of(newCamMatrix, 3, 3, newK, "get_image: newCamMatrix must be 3x3 CV_64F");
    require(!distortionCoeff.empty() && distortionCoeff.type() == CV_64FC1 && distortionCoeff.rows * distortionCoeff.cols == 4,
            "get_image: distortion must be (k1, k2, p1, p2) CV_64F");
    for (int i = 0, k = 0; i < distortionCoeff.rows; i++) for (int j = 0; j < distortionCoeff.cols; j++) d4[k++] = distortionCoeff.at<double>(i, j);
    const int w = current_img.cols, h = current_img.rows;
    const int stride = h > 1 ? (int)(current_img.ptr<uint8_t>(1) - current_img.ptr<uint8_t>(0)) : w * 3;
    const int dh = (int)(h / ((double)w / (double)DESIRED_WIDTH));
    require(dh > 0, "get_image: DESIRED_WIDTH");
    Mat out(dh, DESIRED_WIDTH, CV_8UC1);
    std::vector<uint8_t> tight((size_t)dh * DESIRED_WIDTH);
    int ow = 0, oh = 0;
    SHIM_TRY(uvo_get_image(ctx_now(), current_img.ptr<uint8_t>(0), w, h, stride, UVO_MEM_HOST, K, d4, newK, DESIRED_WIDTH, CLAHE_CORRECTION ? 1 : 0,
                           CLIP_LIMIT, tight.data(), UVO_MEM_HOST, &ow, &oh), "uvo_get_image");
    for (int y = 0; y < oh; y++) memcpy(out.ptr<uint8_t>(y), tight.data() + (size_t)y * ow, (size_t)ow);
    return out;
}

// VOU:91-126: the "SURF" branch (SURF::create(...)->detectAndCompute), the "SIFT" branch (SIFT::create(10000, 3, 0.03, 10, 1.6)->
// detectAndCompute, VOU:107-112) and the "AKAZE" branch (AKAZE::create()->detectAndCompute, VOU:93-98: CV_8U rows of 61 bytes).
// The "ORB" branch (ORB::create(10000, 1.2, 8, 31, 0, 2, ORB::HARRIS_SCORE, 31, 10)->detectAndCompute, VOU:100-105: CV_8U rows of 32 bytes)
// needs OpenCV's learned sampling table bit_pattern_31_, which cannot be restated: uvo_hip::set_orb_pattern(table) or a text file of its
// 1024 integers named by UVO_ORB_PATTERN_FILE; without one the branch throws and says so.
static void orb_pattern_to(uvo_ctx* c)
{
    std::lock_guard<std::mutex> lk(g.mu);
    if (g.orb_pattern.empty())
        if (const char* path = getenv("UVO_ORB_PATTERN_FILE"))
            if (FILE* f = fopen(path, "r")) {
                std::vector<int> v; int x; char sep;
                while (fscanf(f, " %d", &x) == 1) { v.push_back(x); if (fscanf(f, " %c", &sep) == 1 && sep != ',' && sep != ';') ungetc(sep, f); }
                fclose(f);
                if (v.size() == 1024) g.orb_pattern = v;
            }
    if (g.orb_pattern.size() != 1024)
        throw uvo_hip::Error(UVO_INVALID_ARG, "detect_features: FEATURE_DETECTOR \"ORB\" needs OpenCV's sampling table bit_pattern_31_ (features2d/src/orb.cpp, 1024 integers): "
                                              "uvo_hip::set_orb_pattern(table) or UVO_ORB_PATTERN_FILE=<text file of the integers>");
    if (!g.orb_pattern_sent) { SHIM_TRY(uvo_orb_set_pattern(c, g.orb_pattern.data()), "uvo_orb_set_pattern"); g.orb_pattern_sent = true; }
}
void detect_features(Mat img, vector<KeyPoint>& keypoints, Mat& descriptors)
{
    const bool sift = FEATURE_DETECTOR == "SIFT", akaze = FEATURE_DETECTOR == "AKAZE", orb = FEATURE_DETECTOR == "ORB";
    if (!sift && !akaze && !orb && FEATURE_DETECTOR != "SURF") throw uvo_hip::Error(UVO_INVALID_ARG, "detect_features: FEATURE_DETECTOR must be \"SURF\", \"SIFT\", \"AKAZE\" or \"ORB\"");   // VOU:121-124: "WRONG SELECTION OF DETECTOR"
    require(!img.empty() && img.type() == CV_8UC1, "detect_features: CV_8UC1 image expected");
    uvo_ctx* c = ctx_now();
    if (orb) {
        orb_pattern_to(c);
        const int cap = std::max(g.max_kpts, 10000 + 2048);                     // retainBest(10000 in all) bounds the output, ties at the cuts aside
        vector<uvo_keypoint> kps((size_t)cap);
        vector<uint8_t> desc((size_t)cap * 32);
        int n = 0;
        const int stride = img.rows > 1 ? (int)(img.ptr<uint8_t>(1) - img.ptr<uint8_t>(0)) : img.cols;
        SHIM_TRY(uvo_orb_detect(c, img.ptr<uint8_t>(0), img.cols, img.rows, stride, UVO_MEM_HOST, kps.data(), desc.data(), cap, &n), "uvo_orb_detect");
        keypoints.resize((size_t)n);
        if (n) memcpy(static_cast<void*>(keypoints.data()), kps.data(), sizeof(uvo_keypoint) * n);
        descriptors.create(n, 32, CV_8UC1);                                  // ORB::descriptorType() == CV_8U, descriptorSize() == 32
        for (int i = 0; i < n; i++) memcpy(descriptors.ptr<uint8_t>(i), desc.data() + (size_t)i * 32, 32);
        return;
    }
    if (akaze) {
        const int cap = g.max_kpts;
        vector<uvo_keypoint> kps((size_t)cap);
        vector<uint8_t> desc((size_t)cap * 61);
        int n = 0;
        const int stride = img.rows > 1 ? (int)(img.ptr<uint8_t>(1) - img.ptr<uint8_t>(0)) : img.cols;
        SHIM_TRY(uvo_akaze_detect(c, img.ptr<uint8_t>(0), img.cols, img.rows, stride, UVO_MEM_HOST, kps.data(), desc.data(), cap, &n), "uvo_akaze_detect");
        keypoints.resize((size_t)n);
        if (n) memcpy(static_cast<void*>(keypoints.data()), kps.data(), sizeof(uvo_keypoint) * n);
        descriptors.create(n, 61, CV_8UC1);                                  // AKAZE::descriptorType() == CV_8U, descriptorSize() == 61
        for (int i = 0; i < n; i++) memcpy(descriptors.ptr<uint8_t>(i), desc.data() + (size_t)i * 61, 61);
        return;
    }
    const int cap = sift ? std::max(g.max_kpts, 10000) + 1024 : g.max_kpts;       // retainBest(10000) bounds SIFT's output, ties at the cut aside (room for them)
    vector<uvo_keypoint> kps((size_t)cap);
    const int dsize = sift ? 128 : (SURF_EXTENDED ? 128 : 64);             // descriptorSize()
    vector<float> desc((size_t)cap * dsize);
    int n = 0;
    // rows may be padded in a real cv::Mat: pass the row pitch
    const int stride = img.rows > 1 ? (int)(img.ptr<uint8_t>(1) - img.ptr<uint8_t>(0)) : img.cols;
    if (sift)
        SHIM_TRY(uvo_sift_detect(c, img.ptr<uint8_t>(0), img.cols, img.rows, stride, UVO_MEM_HOST, 10000, 3, 0.03, 10, 1.6, kps.data(), desc.data(), cap, &n),
                 "uvo_sift_detect");
    else
        SHIM_TRY(uvo_surf_detect(c, img.ptr<uint8_t>(0), img.cols, img.rows, stride, UVO_MEM_HOST, kps.data(), desc.data(), cap, &n),
                 "uvo_surf_detect");
    keypoints.resize((size_t)n);
    if (n) memcpy(static_cast<void*>(keypoints.data()), kps.data(), sizeof(uvo_keypoint) * n);
    descriptors.create(n, dsize, CV_32FC1);
    for (int i = 0; i < n; i++) memcpy(descriptors.ptr<float>(i), desc.data() + (size_t)i * dsize, sizeof(float) * dsize);
}

namespace {
vector<float> tight_descriptors(const Mat& d, const char* what, int dsize)
{
    require(d.empty() || (d.type() == CV_32FC1 && d.cols == dsize), what);
    vector<float> v((size_t)d.rows * dsize);
    for (int i = 0; i < d.rows; i++) memcpy(v.data() + (size_t)i * dsize, d.ptr<float>(i), sizeof(float) * dsize);
    return v;
}
// BFMatcher(NORM_L2).knnMatch(k = 2) + Lowe ratio; results are APPENDED to `matches` as the reference's push_back does
// dsize: SURF rows (64, or 128 with SURF_EXTENDED: what detect_features produced under the same parameters) or SIFT rows (128)
void match_impl(const Mat& d1, const Mat& d2, vector<DMatch>& matches, int dsize)
{
    vector<float> a = tight_descriptors(d1, "match_features: descriptors1 must be CV_32F with the detector's descriptor size (SURF: 64, or 128 with SURF_EXTENDED; SIFT: 128) columns", dsize);
    vector<float> b = tight_descriptors(d2, "match_features: descriptors2 must be CV_32F with the detector's descriptor size (SURF: 64, or 128 with SURF_EXTENDED; SIFT: 128) columns", dsize);
    const int n1 = d1.rows, n2 = d2.rows;
    if (n1 == 0) return;
    vector<uvo_dmatch> out((size_t)n1);
    int m = 0;
    SHIM_TRY(uvo_match_knn2_ratio_dim(ctx_now(), a.data(), n1, b.data(), n2, dsize, UVO_MEM_HOST, (float)LOWE_RATIO_THRESHOLD, out.data(), n1, &m),
             "uvo_match_knn2_ratio_dim");
    const size_t base = matches.size();
    matches.resize(base + (size_t)m);
    if (m) memcpy(static_cast<void*>(matches.data() + base), out.data(), sizeof(uvo_dmatch) * m);
}
}  // namespace

void match_features(vector<KeyPoint> keypoints1, vector<KeyPoint> keypoints2, Mat descriptors1, Mat descriptors2, vector<DMatch>& matches)
{
    (void)keypoints1; (void)keypoints2;
    if (FEATURE_DETECTOR == "AKAZE" || FEATURE_DETECTOR == "ORB") {                      // VOU:520-524: BFMatcher(NORM_HAMMING) on the caller's binary descriptors
        const Mat& d1 = descriptors1; const Mat& d2 = descriptors2;
        require(d1.empty() || d1.type() == CV_8UC1, "match_features: binary descriptors must be CV_8U");
        require(d2.empty() || (d2.type() == CV_8UC1 && (d1.empty() || d2.cols == d1.cols)), "match_features: binary descriptors must be CV_8U with equal widths");
        const int n1 = d1.rows, n2 = d2.rows, bytes = d1.empty() ? d2.cols : d1.cols;
        if (n1 == 0) return;
        vector<uint8_t> a((size_t)n1 * bytes), b((size_t)n2 * bytes);
        for (int i = 0; i < n1; i++) memcpy(a.data() + (size_t)i * bytes, d1.ptr<uint8_t>(i), (size_t)bytes);
        for (int i = 0; i < n2; i++) memcpy(b.data() + (size_t)i * bytes, d2.ptr<uint8_t>(i), (size_t)bytes);
        vector<uvo_dmatch> out((size_t)n1);
        int m = 0;
        SHIM_TRY(uvo_match_knn2_ratio_hamming(ctx_now(), a.data(), n1, b.data(), n2, bytes, UVO_MEM_HOST, (float)LOWE_RATIO_THRESHOLD, out.data(), n1, &m),
                 "uvo_match_knn2_ratio_hamming");
        const size_t base = matches.size();
        matches.resize(base + (size_t)m);
        if (m) memcpy(static_cast<void*>(matches.data() + base), out.data(), sizeof(uvo_dmatch) * m);
        return;
    }
    // VOU:525-529: "SURF" and "SIFT" share BFMatcher(NORM_L2); any other name matches nothing in the reference (knn_matches stays empty)
    if (FEATURE_DETECTOR == "SIFT") { match_impl(descriptors1, descriptors2, matches, 128); return; }
    if (FEATURE_DETECTOR != "SURF") return;
    match_impl(descriptors1, descriptors2, matches, SURF_EXTENDED ? 128 : 64);
}

void match_features(vector<KeyPoint> keypoints1, vector<KeyPoint> keypoints2, Mat descriptors1, Mat descriptors2, vector<DMatch>& matches,
                    vector<Point2f>& keypoints1_conv, vector<Point2f>& keypoints2_conv)
{
    const size_t base = matches.size();
    if ((!descriptors1.empty() && descriptors1.type() == CV_8UC1) || (!descriptors2.empty() && descriptors2.type() == CV_8UC1)) {
        // VOU:555-556 applies BFMatcher(NORM_L2) to AKAZE / ORB rows as well: for CV_8U rows OpenCV sums the squared byte differences in
        // integers and takes the float square root.  Every such sum is below 2^24, so the float matcher on the bytes widened to float
        // (rows padded with zeros to 64 columns: equal in both sets, no contribution) gives the same distances bit for bit.
        require(descriptors1.empty() || (descriptors1.type() == CV_8UC1 && descriptors1.cols <= 64), "match_features: CV_8U descriptors of at most 64 bytes expected");
        require(descriptors2.empty() || (descriptors2.type() == CV_8UC1 && (descriptors1.empty() || descriptors2.cols == descriptors1.cols)), "match_features: CV_8U descriptors with equal widths expected");
        auto widen = [](const Mat& d) { Mat f; f.create(d.rows, 64, CV_32FC1); for (int i = 0; i < d.rows; i++) { float* o = f.ptr<float>(i); const uint8_t* b = d.ptr<uint8_t>(i); for (int k = 0; k < 64; k++) o[k] = k < d.cols ? (float)b[k] : 0.f; } return f; };
        match_impl(widen(descriptors1), widen(descriptors2), matches, 64);
    } else
    match_impl(descriptors1, descriptors2, matches, descriptors1.empty() ? (SURF_EXTENDED ? 128 : 64) : descriptors1.cols);     // VOU:551-552: NORM_L2 whatever the detector
    for (size_t i = base; i < matches.size(); i++) {
        keypoints1_conv.push_back(keypoints1.at((size_t)matches[i].queryIdx).pt);      // query is keypoints1
        keypoints2_conv.push_back(keypoints2.at((size_t)matches[i].trainIdx).pt);      // train is keypoints2
    }
}

// VOU:632-651
vector<double> reproject_errors(const Mat& world_points, const Mat& R, const Mat& t, const Mat& cameraMatrix, const vector<Point2f>& img_points)
{
    double r[9], tt[3], K[9];
    doubles_of(R, 3, 3, r, "reproject_errors: R must be 3x3 CV_64F");
    doubles_of(t, 3, 1, tt, "reproject_errors: t must be 3x1 CV_64F");
    doubles_of(cameraMatrix, 3, 3, K, "reproject_errors: cameraMatrix must be 3x3 CV_64F");
    vector<double> w = rows3_of(world_points, "reproject_errors: world_points must be N x 3 CV_64F");
    const int n = (int)img_points.size();
    require(world_points.rows >= n, "reproject_errors: fewer world points than image points");
    vector<double> err((size_t)n);
    SHIM_TRY(uvo_reproject_errors(ctx_now(), w.data(), n, r, tt, K, pts_of(img_points), err.data()), "uvo_reproject_errors");
    return err;
}

namespace {
// shared body of extract_3Dpoints / extract_3Dpoints_and_reprojection
void extract_impl(const vector<Point2f>& k1, const vector<Point2f>& k2, const Mat& R1, const Mat& t1, const Mat& R2, const Mat& t2,
                  const Mat& K1m, const Mat& K2m, const Mat& points4D, Mat& very_good_cam1_points, Mat& very_good_indexes,
                  vector<double>* reproj)
{
    double r1[9], tt1[3], r2[9], tt2[3], K1[9], K2[9];
    doubles_of(R1, 3, 3, r1, "extract_3Dpoints: R1 must be 3x3 CV_64F"); doubles_of(t1, 3, 1, tt1, "extract_3Dpoints: t1 must be 3x1 CV_64F");
    doubles_of(R2, 3, 3, r2, "extract_3Dpoints: R2 must be 3x3 CV_64F"); doubles_of(t2, 3, 1, tt2, "extract_3Dpoints: t2 must be 3x1 CV_64F");
    doubles_of(K1m, 3, 3, K1, "extract_3Dpoints: cameraMatrix1 must be 3x3 CV_64F");
    doubles_of(K2m, 3, 3, K2, "extract_3Dpoints: cameraMatrix2 must be 3x3 CV_64F");
    require(points4D.empty() || (points4D.rows == 4 && points4D.type() == CV_32FC1), "extract_3Dpoints: points4D must be 4 x N CV_32F");
    const int n = points4D.cols;
    require((int)k1.size() >= n && (int)k2.size() >= n, "extract_3Dpoints: fewer image points than 3-D points");
    if (n == 0) return;
    vector<float> p4((size_t)4 * n);
    for (int r = 0; r < 4; r++) memcpy(p4.data() + (size_t)r * n, points4D.ptr<float>(r), sizeof(float) * n);
    vector<double> pts((size_t)3 * n);
    vector<int> idx((size_t)n);
    int G = 0;
    SHIM_TRY(uvo_extract_3d_points(ctx_now(), pts_of(k1), pts_of(k2), n, r1, tt1, r2, tt2, K1, K2, p4.data(), pts.data(), idx.data(), &G),
             "uvo_extract_3d_points");
    if (G == 0) return;
    Mat P(G, 3, CV_64FC1), I(G, 1, CV_32SC1);
    for (int i = 0; i < G; i++) { for (int j = 0; j < 3; j++) P.at<double>(i, j) = pts[(size_t)i * 3 + j]; I.at<int>(i, 0) = idx[i]; }
    very_good_cam1_points.push_back(P);
    very_good_indexes.push_back(I);
    if (reproj) {                                         // VOU:262-271: mean of the two views' errors, for the kept points
        vector<Point2f> s1((size_t)G), s2((size_t)G);
        for (int i = 0; i < G; i++) { s1[i] = k1[(size_t)idx[i]]; s2[i] = k2[(size_t)idx[i]]; }
        vector<double> e1((size_t)G), e2((size_t)G);
        SHIM_TRY(uvo_reproject_errors(ctx_now(), pts.data(), G, r1, tt1, K1, pts_of(s1), e1.data()), "uvo_reproject_errors");
        SHIM_TRY(uvo_reproject_errors(ctx_now(), pts.data(), G, r2, tt2, K2, pts_of(s2), e2.data()), "uvo_reproject_errors");
        for (int i = 0; i < G; i++) reproj->push_back((e1[i] + e2[i]) / 2.0);
    }
}
}  // namespace

void extract_3Dpoints(vector<Point2f> keypoints1_conv, vector<Point2f> keypoints2_conv, Mat R1, Mat t1, Mat R2, Mat t2, Mat cameraMatrix1,
                      Mat cameraMatrix2, Mat points4D, Mat& very_good_cam1_points, Mat& very_good_indexes)
{
    extract_impl(keypoints1_conv, keypoints2_conv, R1, t1, R2, t2, cameraMatrix1, cameraMatrix2, points4D, very_good_cam1_points,
                 very_good_indexes, nullptr);
}
void extract_3Dpoints_and_reprojection(vector<Point2f> keypoints1_conv, vector<Point2f> keypoints2_conv, Mat R1, Mat t1, Mat R2, Mat t2,
                                       Mat cameraMatrix1, Mat cameraMatrix2, Mat points4D, Mat& very_good_cam1_points,
                                       Mat& very_good_indexes, vector<double>& reprojection_errors_vector)
{
    extract_impl(keypoints1_conv, keypoints2_conv, R1, t1, R2, t2, cameraMatrix1, cameraMatrix2, points4D, very_good_cam1_points,
                 very_good_indexes, &reprojection_errors_vector);
}

// VOU:306-329
void extract_inliers(const vector<Point2f>& keypoints1_conv, const vector<Point2f>& keypoints2_conv, const Mat& mask,
                     vector<Point2f>& inliers1, vector<Point2f>& inliers2, vector<DMatch>& inlier_matches)
{
    inliers1.clear(); inliers2.clear(); inlier_matches.clear();
    for (int i = 0; i < mask.rows; i++) {
        if (mask.at<uint8_t>(i, 0) != 0) {
            inliers1.push_back(keypoints1_conv[(size_t)i]);
            inliers2.push_back(keypoints2_conv[(size_t)i]);
            DMatch m;
            m.queryIdx = (int)inliers1.size() - 1;
            m.trainIdx = (int)inliers2.size() - 1;
            inlier_matches.push_back(m);
        }
    }
}

// VOU:134-180: reads the global `use_essential`, and CLEARS it when the essential branch fails its inlier test and
// the homography branch is taken instead (VOU:160-163)
void estimate_relative_pose(vector<Point2f> keypoints1_conv, vector<Point2f> keypoints2_conv, Mat cameraMatrix, Mat& R_currCam_prevCam,
                            Mat& t_currCam_prevCam, vector<Point2f>& inliers1, vector<Point2f>& inliers2, vector<DMatch>& inlier_matches,
                            bool& success)
{
    double K[9];
    doubles_of(cameraMatrix, 3, 3, K, "estimate_relative_pose: cameraMatrix must be 3x3 CV_64F");
    require(keypoints1_conv.size() == keypoints2_conv.size(), "estimate_relative_pose: point counts differ");
    const int n = (int)keypoints1_conv.size();
    int ue = use_essential ? 1 : 0, n_in = 0, ok = 0;
    double R[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, t[3] = {0, 0, 0};      // in/out: recoverPose / the homography branch may leave them as they were
    const bool had_R = !R_currCam_prevCam.empty() && R_currCam_prevCam.type() == CV_64FC1 && R_currCam_prevCam.rows == 3 && R_currCam_prevCam.cols == 3;
    const bool had_t = !t_currCam_prevCam.empty() && t_currCam_prevCam.type() == CV_64FC1 && t_currCam_prevCam.rows * t_currCam_prevCam.cols == 3;
    if (had_R) doubles_of(R_currCam_prevCam, 3, 3, R, "estimate_relative_pose: R");
    if (had_t) doubles_of(t_currCam_prevCam, 3, 1, t, "estimate_relative_pose: t");
    vector<Point2f> in1((size_t)(n > 0 ? n : 1)), in2((size_t)(n > 0 ? n : 1));
    vector<uint8_t> mask((size_t)(n > 0 ? n : 1));
    SHIM_TRY(uvo_estimate_relative_pose(ctx_now(), pts_of(keypoints1_conv), pts_of(keypoints2_conv), n, K, &ue, R, t, pts_of(in1), pts_of(in2),
                                        &n_in, mask.data(), &ok), "uvo_estimate_relative_pose");
    use_essential = ue != 0;
    success = ok != 0;
    in1.resize((size_t)n_in); in2.resize((size_t)n_in);
    inliers1 = in1; inliers2 = in2;
    inlier_matches.clear();
    for (int i = 0; i < n_in; i++) { DMatch m; m.queryIdx = i; m.trainIdx = i; inlier_matches.push_back(m); }
    R_currCam_prevCam = mat_from(R, 3, 3); t_currCam_prevCam = mat_from(t, 3, 1);      // the last attempt's pose, also on failure (VOU:149, 154)
}

// VOU:581-624: returns the number of points in front of both cameras for the chosen decomposition
int recover_pose_homography(Mat H, vector<Point2f> inliers1, vector<Point2f> inliers2, Mat cameraMatrix, Mat& R, Mat& t)
{
    double h[9], K[9], r[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tt[3] = {0, 0, 0};        // in/out: written only when a candidate wins (VOU:611-617)
    if (!R.empty() && R.type() == CV_64FC1 && R.rows == 3 && R.cols == 3) doubles_of(R, 3, 3, r, "recover_pose_homography: R");
    if (!t.empty() && t.type() == CV_64FC1 && t.rows * t.cols == 3) doubles_of(t, 3, 1, tt, "recover_pose_homography: t");
    doubles_of(H, 3, 3, h, "recover_pose_homography: H must be 3x3 CV_64F");
    doubles_of(cameraMatrix, 3, 3, K, "recover_pose_homography: cameraMatrix must be 3x3 CV_64F");
    require(inliers1.size() == inliers2.size(), "recover_pose_homography: point counts differ");
    int best = 0;
    SHIM_TRY(uvo_recover_pose_homography(ctx_now(), h, pts_of(inliers1), pts_of(inliers2), (int)inliers1.size(), K, r, tt, &best),
             "uvo_recover_pose_homography");
    R = mat_from(r, 3, 3); t = mat_from(tt, 3, 1);
    return best;
}

// VOU:683-697: rows whose index is out of range are left as created
void select_desired_descriptors(const Mat& descriptors, Mat& descriptors_desired, const Mat& indexes)
{
    descriptors_desired.create(indexes.rows, descriptors.cols, descriptors.type());
    require(descriptors.type() == CV_32FC1 || descriptors.type() == CV_8UC1, "select_desired_descriptors: CV_32F (SURF, SIFT) or CV_8U (AKAZE, ORB) descriptors expected");
    const size_t row_bytes = (size_t)descriptors.cols * descriptors.elemSize();                 // descriptors.row(idx).copyTo(...): whatever the type
    for (int i = 0; i < indexes.rows; i++) {
        int idx = indexes.at<int>(i, 0);
        if (idx >= 0 && idx < descriptors.rows) memcpy(descriptors_desired.ptr<uint8_t>(i), descriptors.ptr<uint8_t>(idx), row_bytes);
    }
}

// VOU:704-717: appends
void select_desired_keypoints(const vector<KeyPoint>& keypoints, vector<KeyPoint>& keypoints_desired, const Mat& indexes)
{
    keypoints_desired.reserve(keypoints_desired.size() + (size_t)indexes.rows);
    for (int i = 0; i < indexes.rows; i++) {
        int idx = indexes.at<int>(i, 0);
        if (idx >= 0 && idx < (int)keypoints.size()) keypoints_desired.push_back(keypoints[(size_t)idx]);
    }
}

// VOU:725-748: true = essential matrix, false = homography (median pixel displacement below DISTANCE)
bool select_estimation_method(const vector<Point2f>& keypoints1_conv, const vector<Point2f>& keypoints2_conv)
{
    require(keypoints1_conv.size() == keypoints2_conv.size(), "select_estimation_method: point counts differ");
    const int essential = uvo_select_estimation_method(pts_of(keypoints1_conv), pts_of(keypoints2_conv), (int)keypoints1_conv.size(), DISTANCE);
    if (essential < 0) throw std::bad_alloc();            // the C entry reports what the reference's std::vector would have thrown
    return essential != 0;
}
