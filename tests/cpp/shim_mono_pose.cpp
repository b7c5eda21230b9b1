// shim_mono_pose.cpp -- test driver: the mono node's relative-pose calls (visual_odometry.h:310-323) through the
// uvo_libraries function surface: select_estimation_method + estimate_relative_pose (+ the global `use_essential`).
//   usage: shim_mono_pose <input.bin> <output.bin>
//   input : int32 n, essential_method, homography_method; f64 K[9], essential_threshold, homography_threshold; n x (x1,y1) f32; n x (x2,y2) f32
//   output: int32 use_essential_in, use_essential_out, success, n_inliers; f64 R[9], t[3]; n_inliers x (x1,y1,x2,y2) f32
#include <cstdio>
#include <vector>
#include "uvo_libraries_hip/VO_utility_hip.h"
using namespace uvocv;

int main(int argc, char** argv)
{
    if (argc != 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[3]; double cam[11];
    if (fread(hdr, sizeof(int), 3, f) != 3 || fread(cam, sizeof(double), 11, f) != 11) return 2;
    const int n = hdr[0];
    std::vector<Point2f> k1((size_t)n), k2((size_t)n);
    if (fread(static_cast<void*>(k1.data()), sizeof(Point2f), n, f) != (size_t)n || fread(static_cast<void*>(k2.data()), sizeof(Point2f), n, f) != (size_t)n) return 2;
    fclose(f);
    // what get_VO_parameters would set from uvo/config/mono_VO_parameters.yaml
    uvo_params mp; uvo_params_default_mono(&mp);
    DISTANCE = mp.DISTANCE; VPF_THRESHOLD = mp.VPF_THRESHOLD; MIN_NUM_INLIERS = mp.MIN_NUM_INLIERS; HOMOGRAPHY_DISTANCE = mp.HOMOGRAPHY_DISTANCE;
    ESSENTIAL_MAX_ITERS = mp.ESSENTIAL_MAX_ITERS; ESSENTIAL_CONFIDENCE = mp.ESSENTIAL_CONFIDENCE;
    HOMOGRAPHY_MAX_ITERS = mp.HOMOGRAPHY_MAX_ITERS; HOMOGRAPHY_CONFIDENCE = mp.HOMOGRAPHY_CONFIDENCE;
    REPROJECTION_TOLERANCE = mp.REPROJECTION_TOLERANCE; MIN_NUM_3DPOINTS = mp.MIN_NUM_3DPOINTS; MIN_NUM_FEATURES = mp.MIN_NUM_FEATURES;
    ESSENTIAL_OUTLIER_METHOD = hdr[1]; HOMOGRAPHY_OUTLIER_METHOD = hdr[2];
    ESSENTIAL_THRESHOLD = cam[9]; HOMOGRAPHY_THRESHOLD = cam[10];
    Mat K(3, 3, CV_64FC1);
    for (int i = 0; i < 9; i++) K.at<double>(i / 3, i % 3) = cam[i];
    try {
        use_essential = select_estimation_method(k1, k2);                                   // VO:310-317
        const int ue_in = use_essential ? 1 : 0;
        Mat R = Mat::eye(3, 3, CV_64FC1), t = Mat::zeros(3, 1, CV_64FC1);                   // as the node initialises them
        std::vector<Point2f> in1, in2; std::vector<DMatch> im; bool success = false;
        estimate_relative_pose(k1, k2, K, R, t, in1, in2, im, success);                    // VO:323
        FILE* o = fopen(argv[2], "wb");
        int oh[4] = { ue_in, use_essential ? 1 : 0, success ? 1 : 0, (int)in1.size() };
        fwrite(oh, sizeof(int), 4, o);
        double rt[12];
        for (int i = 0; i < 9; i++) rt[i] = R.at<double>(i / 3, i % 3);
        for (int i = 0; i < 3; i++) rt[9 + i] = t.at<double>(i, 0);
        fwrite(rt, sizeof(double), 12, o);
        for (size_t i = 0; i < in1.size(); i++) { float v[4] = { in1[i].x, in1[i].y, in2[i].x, in2[i].y }; fwrite(v, sizeof(float), 4, o); }
        fclose(o);
        if (im.size() != in1.size() || (im.size() && (im.back().queryIdx != (int)im.size() - 1))) return 3;
    } catch (const uvo_hip::Error& e) {
        fprintf(stderr, "uvo_hip::Error: %s\n", e.what());
        return 1;
    }
    uvo_hip::shutdown();
    return 0;
}
