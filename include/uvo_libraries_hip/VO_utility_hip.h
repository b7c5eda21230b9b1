// VO_utility_hip.h -- the uvo_libraries function surface, served by libuvo_hip.so (MI355X).
//
// Drop-in for the hot-path part of uvo_libraries/include/uvo_libraries/VO_utility.h (functions VOH:96-117, parameter
// globals VOH:25-89): same names, same argument order and meaning, same in/out conventions (outputs that the
// reference fills with push_back are appended to here as well).  The nodes (uvo/include/visual_odometry.h) include
// this header instead and link libuvo_libraries_hip.so; INTEGRATION.md lists the three lines that change.
//
// Differences a maintainer should know about:
//   * the parameter globals are DECLARED here and DEFINED once in the library (the reference defines them in its
//     header); get_VO_parameters() keeps assigning to them as before, every call below reads their current value;
//   * detect_features serves all four names of FEATURE_DETECTOR; "ORB" needs OpenCV's learned sampling table bit_pattern_31_
//     (features2d/src/orb.cpp), which this library cannot restate: uvo_hip::set_orb_pattern(table), or the environment variable
//     UVO_ORB_PATTERN_FILE naming a text file of its 1024 integers -- without it the ORB branch throws and says so;
//   * OpenCV errors become uvo_hip::Error (a std::runtime_error) carrying the library's message;
//   * the cv:: calls the stereo/mono loops make directly (triangulatePoints, solvePnPRansac, Rodrigues) have
//     same-signature replacements in namespace uvo_hip.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>
#include "uvo_libraries_hip/cv_compat.h"
#include "uvo_hip.h"

// ---- parameter globals (VOH:25-89), defined in VO_utility_hip.cpp ---------------------------------------------
extern std::string FEATURE_DETECTOR;
extern int    DESIRED_WIDTH;              extern bool CLAHE_CORRECTION;   extern int CLIP_LIMIT;      // VOH:41-44
extern int    DISTANCE;
extern int    ESSENTIAL_OUTLIER_METHOD;   extern double ESSENTIAL_MAX_ITERS, ESSENTIAL_CONFIDENCE, ESSENTIAL_THRESHOLD;
extern int    HOMOGRAPHY_OUTLIER_METHOD;  extern double HOMOGRAPHY_MAX_ITERS, HOMOGRAPHY_CONFIDENCE, HOMOGRAPHY_THRESHOLD, HOMOGRAPHY_DISTANCE;
extern double VPF_THRESHOLD, REPROJECTION_TOLERANCE, LOWE_RATIO_THRESHOLD;
extern int    MIN_NUM_FEATURES, MIN_NUM_3DPOINTS, MIN_NUM_INLIERS;
extern int    ITERATIONS_COUNT;           extern double REPROJECTION_ERROR_THRESHOLD, CONFIDENCE;
extern bool   USE_EXTRINSIC_GUESS;        extern int PNP_METHOD_FLAG;
extern int    SURF_MIN_HESSIAN, SURF_OCTAVES_NUMBER, SURF_OCTAVES_LAYERS;
extern bool   SURF_EXTENDED, SURF_UPRIGHT;
extern bool   use_essential;

// ---- functions (VOH:96-117; implementation lines cite uvo_libraries/src/VO_utility.cpp = VOU) ------------------
uvocv::Mat compute_projection_matrix(const uvocv::Mat& R, const uvocv::Mat& t, const uvocv::Mat& cameraIntrinsic);      // VOU:9-15
double     compute_scale_factor(float distance, const uvocv::Mat& world_points);                                        // VOU:23-38
uvocv::Mat convert_3Dpoints_camera(const uvocv::Mat& points_to_convert, const uvocv::Mat& R_to_from, const uvocv::Mat& t_to_from); // VOU:46-63
uvocv::Mat convert_from_homogeneous_coords(const uvocv::Mat& points4d);                                                 // VOU:71-83
// VO_utility.h:112: scales cameraMatrix in place by original width / DESIRED_WIDTH, newCamMatrix = getOptimalNewCameraMatrix(alpha = 0)
void resize_camera_matrix(uvocv::Mat original_image, uvocv::Mat& cameraMatrix, uvocv::Mat distortionCoeff, uvocv::Mat& newCamMatrix);
uvocv::Mat get_image(const uvocv::Mat& current_img, const uvocv::Mat& cameraMatrix, const uvocv::Mat& distortionCoeff,
                     const uvocv::Mat& newCamMatrix);                                                                  // VOU:337-379
void detect_features(uvocv::Mat img, std::vector<uvocv::KeyPoint>& keypoints, uvocv::Mat& descriptors);                 // VOU:91-126
void estimate_relative_pose(std::vector<uvocv::Point2f> keypoints1_conv, std::vector<uvocv::Point2f> keypoints2_conv,
                            uvocv::Mat cameraMatrix, uvocv::Mat& R_currCam_prevCam, uvocv::Mat& t_currCam_prevCam,
                            std::vector<uvocv::Point2f>& inliers1, std::vector<uvocv::Point2f>& inliers2,
                            std::vector<uvocv::DMatch>& inlier_matches, bool& success);                                 // VOU:134-180
void extract_3Dpoints(std::vector<uvocv::Point2f> keypoints1_conv, std::vector<uvocv::Point2f> keypoints2_conv,
                      uvocv::Mat R1, uvocv::Mat t1, uvocv::Mat R2, uvocv::Mat t2, uvocv::Mat cameraMatrix1,
                      uvocv::Mat cameraMatrix2, uvocv::Mat points4D, uvocv::Mat& very_good_cam1_points,
                      uvocv::Mat& very_good_indexes);                                                                   // VOU:188-237
void extract_3Dpoints_and_reprojection(std::vector<uvocv::Point2f> keypoints1_conv, std::vector<uvocv::Point2f> keypoints2_conv,
                      uvocv::Mat R1, uvocv::Mat t1, uvocv::Mat R2, uvocv::Mat t2, uvocv::Mat cameraMatrix1,
                      uvocv::Mat cameraMatrix2, uvocv::Mat points4D, uvocv::Mat& very_good_cam1_points,
                      uvocv::Mat& very_good_indexes, std::vector<double>& reprojection_errors_vector);                  // VOU:245-298
void extract_inliers(const std::vector<uvocv::Point2f>& keypoints1_conv, const std::vector<uvocv::Point2f>& keypoints2_conv,
                     const uvocv::Mat& mask, std::vector<uvocv::Point2f>& inliers1, std::vector<uvocv::Point2f>& inliers2,
                     std::vector<uvocv::DMatch>& inlier_matches);                                                       // VOU:306-329
void match_features(std::vector<uvocv::KeyPoint> keypoints1, std::vector<uvocv::KeyPoint> keypoints2, uvocv::Mat descriptors1,
                    uvocv::Mat descriptors2, std::vector<uvocv::DMatch>& matches);                                      // VOU:515-543
void match_features(std::vector<uvocv::KeyPoint> keypoints1, std::vector<uvocv::KeyPoint> keypoints2, uvocv::Mat descriptors1,
                    uvocv::Mat descriptors2, std::vector<uvocv::DMatch>& matches,
                    std::vector<uvocv::Point2f>& keypoints1_conv, std::vector<uvocv::Point2f>& keypoints2_conv);        // VOU:551-573
int  recover_pose_homography(uvocv::Mat H, std::vector<uvocv::Point2f> inliers1, std::vector<uvocv::Point2f> inliers2,
                             uvocv::Mat cameraMatrix, uvocv::Mat& R, uvocv::Mat& t);                                    // VOU:581-624
std::vector<double> reproject_errors(const uvocv::Mat& world_points, const uvocv::Mat& R, const uvocv::Mat& t,
                                     const uvocv::Mat& cameraMatrix, const std::vector<uvocv::Point2f>& img_points);    // VOU:632-651
void select_desired_descriptors(const uvocv::Mat& descriptors, uvocv::Mat& descriptors_desired, const uvocv::Mat& indexes);   // VOU:683-697
void select_desired_keypoints(const std::vector<uvocv::KeyPoint>& keypoints, std::vector<uvocv::KeyPoint>& keypoints_desired,
                              const uvocv::Mat& indexes);                                                               // VOU:704-717
bool select_estimation_method(const std::vector<uvocv::Point2f>& keypoints1_conv,
                              const std::vector<uvocv::Point2f>& keypoints2_conv);                                      // VOU:725-748

namespace uvo_hip {

struct Error : std::runtime_error { uvo_status status; Error(uvo_status s, const std::string& m) : std::runtime_error(m), status(s) {} };

// The library keeps one process-wide device context, created on first use for images up to max_w x max_h and
// max_kpts keypoints per image.  Call configure() before the first function to choose other limits or another GPU.
void     configure(int device, int max_w, int max_h, int max_kpts);
uvo_ctx* context();
void     shutdown();
// OpenCV's bit_pattern_31_ for the "ORB" branch of detect_features: 1024 ints, x0, y0, x1, y1 per descriptor bit (nullptr forgets it)
void     set_orb_pattern(const int* pattern1024);

// Replacements for the cv:: functions the node loops call directly (visual_odometry.h:355, 631, 647-648, 673).
void triangulatePoints(const uvocv::Mat& projMatr1, const uvocv::Mat& projMatr2, const std::vector<uvocv::Point2f>& projPoints1,
                       const std::vector<uvocv::Point2f>& projPoints2, uvocv::Mat& points4D);
bool solvePnPRansac(const uvocv::Mat& objectPoints /* N x 3 CV_64F */, const std::vector<uvocv::Point2f>& imagePoints,
                    const uvocv::Mat& cameraMatrix, const uvocv::Mat& distCoeffs /* empty or zeros: images are undistorted upstream */,
                    uvocv::Mat& rvec, uvocv::Mat& tvec, bool useExtrinsicGuess, int iterationsCount, float reprojectionError,
                    double confidence, uvocv::Mat& inliers, int flags);
void Rodrigues(const uvocv::Mat& src, uvocv::Mat& dst);

}  // namespace uvo_hip
