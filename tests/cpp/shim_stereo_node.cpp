// shim_stereo_node.cpp -- test driver: the stereo node's per-pair loop (uvo/include/visual_odometry.h:474-520 init,
// 531-739 main loop, 148-159 output) written against the uvo_libraries function surface exactly as the node uses it,
// but linked to libuvo_libraries_hip.so.  tests/test_shim.py feeds it synthetic pairs and compares every per-frame
// record bit for bit with the CPU oracle's stereo state machine.
//
//   usage: shim_stereo_node <input.bin> <output.bin>
//   input : int32 W, H, nframes, min_hessian (-1 / -2 / -3: FEATURE_DETECTOR = "SIFT" / "AKAZE" / "ORB"); f64 K_left[9], K_right[9], R_right[9], t_right[3]; then nframes x (L, R) u8
//   output: per frame 8 x int32 (valid, initialized, nL, nR, n_stereo, n_tri, G, n_inliers) + 9 x f64 (rvec, tvec, t_prev_curr)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "uvo_libraries_hip/VO_utility_hip.h"

using namespace uvocv;
using std::vector;

static Mat mat64(const double* v, int r, int c) { Mat m(r, c, CV_64FC1); for (int i = 0; i < r * c; i++) m.at<double>(i / c, i % c) = v[i]; return m; }
static vector<Point2f> points_of(const vector<KeyPoint>& k) { vector<Point2f> p; for (const KeyPoint& q : k) p.push_back(q.pt); return p; }   // KeyPoint::convert

int main(int argc, char** argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("input"); return 2; }
    int hdr[4]; double cam[30];
    if (fread(hdr, sizeof(int), 4, f) != 4 || fread(cam, sizeof(double), 30, f) != 30) { fprintf(stderr, "short header\n"); return 2; }
    const int W = hdr[0], H = hdr[1], nframes = hdr[2];
    if (hdr[3] == -1) FEATURE_DETECTOR = "SIFT";                      // min_hessian < 0: the loop on another detector's features -- SIFT (VOU:107-112, 525-529),
    else if (hdr[3] == -2) FEATURE_DETECTOR = "AKAZE";               // AKAZE (VOU:93-98) or ORB (VOU:100-105; the sampling table from UVO_ORB_PATTERN_FILE), both
    else if (hdr[3] == -3) FEATURE_DETECTOR = "ORB";                 // through match_features' Hamming branch (VOU:520-524)
    else SURF_MIN_HESSIAN = hdr[3];                                 // get_VO_parameters would set the globals
    Mat K_left = mat64(cam, 3, 3), K_right = mat64(cam + 9, 3, 3), R_right = mat64(cam + 18, 3, 3), t_right = mat64(cam + 27, 3, 1);
    Mat R_eye = Mat::eye(3, 3, CV_64FC1), t_zeros = Mat::zeros(3, 1, CV_64FC1), distCoeffs;
    Mat P_eye_left = compute_projection_matrix(R_eye, t_zeros, K_left);            // VO:460
    Mat P_right = compute_projection_matrix(R_right, t_right, K_right);            // VO:462
    uvo_hip::configure(0, W, H, hdr[3] == -3 ? 16384 : 8192);        // ORB::create(10000, ...) returns up to 10000 keypoints (and ties)

    FILE* out = fopen(argv[2], "wb");
    if (!out) { perror("output"); return 2; }
    bool vo_initialized = false;
    vector<DMatch> results_match_prev;                               // VO:468: lives across init attempts
    vector<KeyPoint> prevL_as, prevR_as; Mat prevL_desc_as;
    Mat rvec = Mat::zeros(3, 1, CV_64FC1), tvec = Mat::zeros(3, 1, CV_64FC1), t_prev_curr = Mat::zeros(3, 1, CV_64FC1);
    try {
        for (int fr = 0; fr < nframes; fr++) {
            Mat L(H, W, CV_8UC1), R(H, W, CV_8UC1);
            for (int y = 0; y < H; y++) if (fread(L.ptr<uint8_t>(y), 1, W, f) != (size_t)W) { fprintf(stderr, "short frame\n"); return 2; }
            for (int y = 0; y < H; y++) if (fread(R.ptr<uint8_t>(y), 1, W, f) != (size_t)W) { fprintf(stderr, "short frame\n"); return 2; }
            vector<KeyPoint> kL, kR; Mat dL, dR;
            detect_features(L, kL, dL);
            detect_features(R, kR, dR);
            int rec[8] = {0, vo_initialized ? 1 : 0, (int)kL.size(), (int)kR.size(), 0, 0, 0, 0};
            if (!vo_initialized) {                                  // VO:474-520
                if ((int)kL.size() >= MIN_NUM_FEATURES && (int)kR.size() >= MIN_NUM_FEATURES) {
                    match_features(kL, kR, dL, dR, results_match_prev);
                    if ((int)results_match_prev.size() > MIN_NUM_FEATURES) vo_initialized = true;
                }
                rec[4] = (int)results_match_prev.size();
                if (vo_initialized) {
                    Mat il, ir;
                    for (const DMatch& m : results_match_prev) { il.push_back(m.queryIdx); ir.push_back(m.trainIdx); }
                    select_desired_descriptors(dL, prevL_desc_as, il);
                    select_desired_keypoints(kL, prevL_as, il);
                    select_desired_keypoints(kR, prevR_as, ir);
                }
            } else {                                                // VO:531-739
                int valid = 0;
                vector<DMatch> m_curr, m_pc;
                vector<KeyPoint> currL_as, currR_as; Mat currL_desc_as;
                Mat good_pts, good_idx, inliers_idx;
                if ((int)kL.size() >= MIN_NUM_FEATURES && (int)kR.size() >= MIN_NUM_FEATURES) {
                    match_features(kL, kR, dL, dR, m_curr);
                    if ((int)m_curr.size() > MIN_NUM_FEATURES) {
                        Mat il, ir;
                        for (const DMatch& m : m_curr) { il.push_back(m.queryIdx); ir.push_back(m.trainIdx); }
                        select_desired_descriptors(dL, currL_desc_as, il);
                        select_desired_keypoints(kL, currL_as, il);
                        select_desired_keypoints(kR, currR_as, ir);
                        match_features(prevL_as, kL, prevL_desc_as, dL, m_pc);               // triangular matching, VO:592
                        Mat pl_idx, pr_idx;
                        for (const DMatch& m : m_pc) { pl_idx.push_back(m.queryIdx); pr_idx.push_back(m.trainIdx); }
                        vector<KeyPoint> pl, pr, cu;
                        select_desired_keypoints(prevL_as, pl, pl_idx);
                        select_desired_keypoints(prevR_as, pr, pl_idx);
                        select_desired_keypoints(kL, cu, pr_idx);
                        vector<Point2f> x1 = points_of(pl), x2 = points_of(pr);
                        if ((int)m_pc.size() > MIN_NUM_FEATURES) {
                            Mat points4D;
                            uvo_hip::triangulatePoints(P_eye_left, P_right, x1, x2, points4D);              // VO:631
                            extract_3Dpoints(x1, x2, R_eye, t_zeros, R_right, t_right, K_left, K_right, points4D, good_pts, good_idx);
                            if (good_pts.rows > MIN_NUM_3DPOINTS) {
                                vector<KeyPoint> good_cu;
                                select_desired_keypoints(cu, good_cu, good_idx);
                                vector<Point2f> ci = points_of(good_cu);
                                uvo_hip::solvePnPRansac(good_pts, ci, K_left, distCoeffs, rvec, tvec, USE_EXTRINSIC_GUESS, ITERATIONS_COUNT,
                                                        (float)REPROJECTION_ERROR_THRESHOLD, CONFIDENCE, inliers_idx, PNP_METHOD_FLAG);   // VO:647
                                if (inliers_idx.rows >= MIN_NUM_INLIERS) {
                                    Mat Rm;
                                    uvo_hip::Rodrigues(rvec, Rm);                                            // VO:673
                                    for (int i = 0; i < 3; i++) {                                            // VO:675: -R^T t
                                        double acc = 0;
                                        for (int k = 0; k < 3; k++) acc += Rm.at<double>(k, i) * tvec.at<double>(k, 0);
                                        t_prev_curr.at<double>(i, 0) = acc * -1.0;
                                    }
                                    valid = 1;
                                }
                            }
                        }
                    }
                }
                rec[0] = valid; rec[4] = (int)m_curr.size(); rec[5] = (int)m_pc.size(); rec[6] = good_pts.rows; rec[7] = inliers_idx.rows;
                prevL_as = currL_as; prevR_as = currR_as; prevL_desc_as = currL_desc_as.clone();             // VO:727-733
            }
            double vals[9];
            for (int i = 0; i < 3; i++) { vals[i] = rvec.at<double>(i, 0); vals[3 + i] = tvec.at<double>(i, 0); vals[6 + i] = t_prev_curr.at<double>(i, 0); }
            fwrite(rec, sizeof(int), 8, out); fwrite(vals, sizeof(double), 9, out);
        }
    } catch (const uvo_hip::Error& e) {
        fprintf(stderr, "uvo_hip::Error: %s\n", e.what());
        return 1;
    }
    fclose(out); fclose(f);
    uvo_hip::shutdown();
    return 0;
}
