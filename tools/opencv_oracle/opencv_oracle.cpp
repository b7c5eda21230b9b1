// opencv_oracle.cpp -- OPTIONAL dumper that pins this repository's CPU oracle (and through it the HIP path) to the real
// OpenCV 4.5.x + opencv_contrib (xfeatures2d, nonfree SURF) that team-ergo-unipi/ergo_uvo links.
//
// It is NOT built or run anywhere in this repository's pipeline (no OpenCV in the image; there is no network).  A third
// party with OpenCV 4.5 + contrib does:
//     python tools/opencv_oracle/make_inputs.py /tmp/uvo_inputs.npz            # deterministic inputs (numpy only)
//     g++ -O2 -std=c++17 tools/opencv_oracle/opencv_oracle.cpp -o opencv_oracle $(pkg-config --cflags --libs opencv4)
//     ./opencv_oracle /tmp/uvo_inputs.npz tests/golden/opencv_fixture.npz [bit_pattern_31.txt]
// (the optional third argument: a text file with the 1024 integers between the braces of `static int bit_pattern_31_[256*4]` in
//  modules/features2d/src/orb.cpp of the OpenCV tree that was built, comments removed -- stored in the fixture so that ORB's
//  DESCRIPTORS can be compared as well; without it the ORB keypoints are compared alone)
// Every OpenCV call below has the argument shapes of the reference's call site, cited next to it
// (VOU = uvo_libraries/src/VO_utility.cpp, VO = uvo/include/visual_odometry.h of the reference).
#include <opencv2/opencv.hpp>
#include <opencv2/xfeatures2d/nonfree.hpp>
#include "npz_io.h"

using namespace cv;

static Mat mat_u8(const npz::Array& a) { return Mat((int)a.shape[0], (int)a.shape[1], a.shape.size() == 3 ? CV_8UC3 : CV_8UC1, const_cast<unsigned char*>(a.bytes.data())).clone(); }
static Mat mat_f64(const npz::Array& a)
{
    const int r = (int)a.shape[0], c = a.shape.size() > 1 ? (int)a.shape[1] : 1;
    return Mat(r, c, CV_64F, const_cast<unsigned char*>(a.bytes.data())).clone();
}
static std::vector<Point2f> pts_f32(const npz::Array& a)
{
    std::vector<Point2f> v(a.shape[0]);
    const float* p = a.as<float>();
    for (size_t i = 0; i < v.size(); i++) v[i] = Point2f(p[2 * i], p[2 * i + 1]);
    return v;
}
static double scalar(const npz::File& f, const char* k) { const npz::Array& a = f.at(k); return a.dtype == "<f8" ? a.as<double>()[0] : (double)a.as<int32_t>()[0]; }
static npz::Array from_mat(const Mat& m)
{
    Mat c = m.isContinuous() ? m : m.clone();
    std::vector<size_t> shape = { (size_t)c.rows, (size_t)c.cols };
    if (c.channels() > 1) shape.push_back((size_t)c.channels());
    switch (c.depth()) {
    case CV_8U:  return npz::make(c.ptr<uint8_t>(), shape);
    case CV_32S: return npz::make(c.ptr<int32_t>(), shape);
    case CV_32F: return npz::make(c.ptr<float>(), shape);
    case CV_64F: return npz::make(c.ptr<double>(), shape);
    }
    throw std::runtime_error("from_mat: unsupported depth");
}
static void put_keypoints(npz::File& out, const std::string& key, const std::vector<KeyPoint>& kps)
{
    std::vector<float> f(kps.size() * 5); std::vector<int32_t> i(kps.size() * 2);
    for (size_t k = 0; k < kps.size(); k++) {
        f[5*k] = kps[k].pt.x; f[5*k + 1] = kps[k].pt.y; f[5*k + 2] = kps[k].size; f[5*k + 3] = kps[k].angle; f[5*k + 4] = kps[k].response;
        i[2*k] = kps[k].octave; i[2*k + 1] = kps[k].class_id;
    }
    out[key + "_f"] = npz::make(f.data(), { kps.size(), 5 });
    out[key + "_i"] = npz::make(i.data(), { kps.size(), 2 });
}

int main(int argc, char** argv)
{
    if (argc != 3 && argc != 4) { fprintf(stderr, "usage: %s inputs.npz fixture.npz [bit_pattern_31.txt]\n", argv[0]); return 2; }
    const npz::File in = npz::load(argv[1]);
    npz::File out;
    {
        const std::string v = CV_VERSION;
        out["opencv_version"] = npz::make(reinterpret_cast<const uint8_t*>(v.data()), { v.size() });
    }
    // ---------------- detect_features, SURF branch: VOU:114-119 ----------------
    const int min_hessian = (int)scalar(in, "min_hessian");
    std::vector<KeyPoint> kps[3]; Mat desc[3];
    const char* names[3] = { "left0", "right0", "left1" };
    for (int k = 0; k < 3; k++) {
        Mat img = mat_u8(in.at(names[k]));
        Ptr<xfeatures2d::SURF> detector = xfeatures2d::SURF::create(min_hessian, 4, 3, false, true);      // SURF_* of stereo_VO_parameters.yaml:43-47
        detector->detectAndCompute(img, noArray(), kps[k], desc[k]);
        put_keypoints(out, std::string("surf_") + names[k] + "_kps", kps[k]);
        out[std::string("surf_") + names[k] + "_desc"] = from_mat(desc[k]);
        if (k == 0) { Mat sum; integral(img, sum, CV_32S); out["integral_left0"] = from_mat(sum); }      // surf.cpp: integral(img, sum, CV_32S)
    }
    // ---------------- detect_features, SIFT branch: VOU:107-112 ----------------
    // (retainBest leaves the keypoints in std::nth_element's order when more than 10000 survive; the inputs stay below that, so the
    //  order is KeyPoint_LessThan's, the one removeDuplicatedSorted establishes)
    {
        Mat sdesc[2]; std::vector<KeyPoint> skps[2];
        for (int k = 0; k < 2; k++) {
            Mat img = mat_u8(in.at(names[k]));
            Ptr<SIFT> detector = SIFT::create(10000, 3, 0.03, 10, 1.6);
            detector->detectAndCompute(img, noArray(), skps[k], sdesc[k]);
            put_keypoints(out, std::string("sift_") + names[k] + "_kps", skps[k]);
            out[std::string("sift_") + names[k] + "_desc"] = from_mat(sdesc[k]);
        }
        // match_features with FEATURE_DETECTOR == "SIFT" (VOU:525-529): the same BFMatcher(NORM_L2), 128-float rows
        Ptr<DescriptorMatcher> matcher = DescriptorMatcher::create(DescriptorMatcher::BRUTEFORCE);
        std::vector<std::vector<DMatch>> knn;
        matcher->knnMatch(sdesc[0], sdesc[1], knn, 2);
        std::vector<int32_t> good; std::vector<float> gd;
        const float ratio = (float)scalar(in, "lowe_ratio");
        for (size_t i = 0; i < knn.size(); i++)
            if (knn[i].size() >= 2 && knn[i][0].distance < ratio * knn[i][1].distance) {
                good.push_back(knn[i][0].queryIdx); good.push_back(knn[i][0].trainIdx); gd.push_back(knn[i][0].distance);
            }
        out["sift_ratio_matches"] = npz::make(good.data(), { good.size() / 2, 2 });
        out["sift_ratio_dist"] = npz::make(gd.data(), { gd.size() });
    }
    // ---------------- detect_features, AKAZE branch: VOU:93-98, ORB branch: VOU:100-105; match_features' Hamming arm: VOU:520-524 ----------------
    {
        const float ratio = (float)scalar(in, "lowe_ratio");
        for (int which = 0; which < 2; which++) {
            const std::string tag = which == 0 ? "akaze_" : "orb_";
            Mat bdesc[2]; std::vector<KeyPoint> bkps[2];
            for (int k = 0; k < 2; k++) {
                Mat img = mat_u8(in.at(names[k]));
                if (which == 0) { Ptr<AKAZE> detector = AKAZE::create(); detector->detectAndCompute(img, noArray(), bkps[k], bdesc[k]); }
                else { Ptr<ORB> detector = ORB::create(10000, 1.2, 8, 31, 0, 2, ORB::HARRIS_SCORE, 31, 10); detector->detectAndCompute(img, noArray(), bkps[k], bdesc[k]); }
                put_keypoints(out, tag + names[k] + "_kps", bkps[k]);
                out[tag + names[k] + "_desc"] = from_mat(bdesc[k].empty() ? Mat(0, which == 0 ? 61 : 32, CV_8U) : bdesc[k]);
            }
            BFMatcher matcher(NORM_HAMMING, false);                                                          // VOU:522
            std::vector<std::vector<DMatch>> knn;
            if (!bdesc[0].empty() && !bdesc[1].empty()) matcher.knnMatch(bdesc[0], bdesc[1], knn, 2);
            std::vector<int32_t> good; std::vector<float> gd;
            for (size_t i = 0; i < knn.size(); i++)
                if (knn[i].size() >= 2 && knn[i][0].distance < ratio * knn[i][1].distance) {
                    good.push_back(knn[i][0].queryIdx); good.push_back(knn[i][0].trainIdx); gd.push_back(knn[i][0].distance);
                }
            out[tag + "ratio_matches"] = npz::make(good.data(), { good.size() / 2, 2 });
            out[tag + "ratio_dist"] = npz::make(gd.data(), { gd.size() });
        }
        if (argc == 4) {                                                                                     // the sampling table this OpenCV was built with
            std::vector<int32_t> pat; int v;
            if (FILE* f = fopen(argv[3], "r")) { while (fscanf(f, " %d", &v) == 1) { pat.push_back(v); int ch; while ((ch = fgetc(f)) != EOF && (ch == ',' || ch == ' ' || ch == '\n' || ch == '\r' || ch == '\t' || ch == ';')) {} if (ch != EOF) ungetc(ch, f); } fclose(f); }
            if (pat.size() != 1024) { fprintf(stderr, "%s: expected the 1024 integers of bit_pattern_31_, found %zu\n", argv[3], pat.size()); return 2; }
            out["orb_pattern"] = npz::make(pat.data(), { 256, 4 });
        }
    }
    // ---------------- match_features: VOU:515-543 (BFMatcher(NORM_L2).knnMatch k = 2, ratio test) ----------------
    {
        Ptr<DescriptorMatcher> matcher = DescriptorMatcher::create(DescriptorMatcher::BRUTEFORCE);        // VOU:526: NORM_L2 for SURF
        std::vector<std::vector<DMatch>> knn;
        matcher->knnMatch(desc[0], desc[1], knn, 2);                                                       // VOU:528
        std::vector<int32_t> idx(knn.size() * 2, -1); std::vector<float> dist(knn.size() * 2, 0.f);
        std::vector<int32_t> good; std::vector<float> gd;
        const float ratio = (float)scalar(in, "lowe_ratio");
        for (size_t i = 0; i < knn.size(); i++) {
            for (size_t j = 0; j < knn[i].size() && j < 2; j++) { idx[2*i + j] = knn[i][j].trainIdx; dist[2*i + j] = knn[i][j].distance; }
            if (knn[i].size() >= 2 && knn[i][0].distance < ratio * knn[i][1].distance) {                  // VOU:536
                good.push_back(knn[i][0].queryIdx); good.push_back(knn[i][0].trainIdx); gd.push_back(knn[i][0].distance);
            }
        }
        out["knn_idx"] = npz::make(idx.data(), { knn.size(), 2 });
        out["knn_dist"] = npz::make(dist.data(), { knn.size(), 2 });
        out["ratio_matches"] = npz::make(good.data(), { good.size() / 2, 2 });
        out["ratio_dist"] = npz::make(gd.data(), { gd.size() });
    }
    // ---------------- cv::triangulatePoints: VO:631 ----------------
    {
        Mat P1 = mat_f64(in.at("tri_P1")), P2 = mat_f64(in.at("tri_P2")), p4;
        std::vector<Point2f> x1 = pts_f32(in.at("tri_x1")), x2 = pts_f32(in.at("tri_x2"));
        triangulatePoints(P1, P2, x1, x2, p4);
        out["tri_points4d"] = from_mat(p4);
    }
    // ---------------- cv::solvePnPRansac(..., SOLVEPNP_EPNP): VO:647-648, cv::Rodrigues: VO:673 ----------------
    {
        Mat X = mat_f64(in.at("pnp_X")), K = mat_f64(in.at("pnp_K")), distc = Mat::zeros(4, 1, CV_64F), rvec, tvec, inl;
        std::vector<Point2f> x = pts_f32(in.at("pnp_x"));
        const bool ok = solvePnPRansac(X, x, K, distc, rvec, tvec, false, (int)scalar(in, "pnp_iterations"), (float)scalar(in, "pnp_reprojection_error"),
                                       scalar(in, "pnp_confidence"), inl, SOLVEPNP_EPNP);
        const int32_t okv = ok ? 1 : 0;
        out["pnp_ok"] = npz::make(&okv, { 1 });
        out["pnp_rvec"] = from_mat(rvec); out["pnp_tvec"] = from_mat(tvec);
        Mat inl32; if (!inl.empty()) inl.convertTo(inl32, CV_32S); else inl32 = Mat(0, 1, CV_32S);
        out["pnp_inliers"] = from_mat(inl32);
        Mat R; Rodrigues(rvec, R); out["pnp_R"] = from_mat(R);
    }
    // ---------------- findEssentialMat + recoverPose: VOU:147-149; findHomography: VOU:152; decomposeHomographyMat: VOU:585 ----------------
    {
        Mat K = mat_f64(in.at("mono_K"));
        std::vector<Point2f> e1 = pts_f32(in.at("e_x1")), e2 = pts_f32(in.at("e_x2")), h1 = pts_f32(in.at("h_x1")), h2 = pts_f32(in.at("h_x2"));
        for (int method : { (int)RANSAC, (int)LMEDS }) {
            const std::string m = std::to_string(method);
            const double thr = scalar(in, method == RANSAC ? "ransac_threshold" : "lmeds_threshold");
            Mat mask, E = findEssentialMat(e1, e2, K, method, scalar(in, "e_confidence"), thr, (int)scalar(in, "max_iters"), mask);
            out["E_" + m] = from_mat(E); out["E_mask_" + m] = from_mat(mask);
            if (E.rows == 3) {
                Mat R, t, m2 = mask.clone();
                const int32_t good = recoverPose(E, e1, e2, K, R, t, m2);
                out["rp_R_" + m] = from_mat(R); out["rp_t_" + m] = from_mat(t); out["rp_mask_" + m] = from_mat(m2); out["rp_good_" + m] = npz::make(&good, { 1 });
            }
            Mat hmask, H = findHomography(h1, h2, method, thr, hmask, (int)scalar(in, "max_iters"), scalar(in, "h_confidence"));
            out["H_" + m] = from_mat(H.empty() ? Mat::zeros(3, 3, CV_64F) : H); out["H_mask_" + m] = from_mat(hmask);
            if (method == RANSAC && !H.empty()) {
                std::vector<Mat> Rs, ts, ns;
                const int n = decomposeHomographyMat(H, K, Rs, ts, ns);
                Mat allR(0, 3, CV_64F), allt(0, 1, CV_64F), alln(0, 1, CV_64F);
                for (int i = 0; i < n; i++) { allR.push_back(Rs[i]); allt.push_back(ts[i]); alln.push_back(ns[i]); }
                out["Hdec_R"] = from_mat(allR); out["Hdec_t"] = from_mat(allt); out["Hdec_n"] = from_mat(alln);
            }
        }
    }
    // ---------------- get_image: VOU:337-379, resize_camera_matrix: VOU:658-675 ----------------
    {
        Mat rgb = mat_u8(in.at("pre_rgb")), K = mat_f64(in.at("pre_K")), dist4 = mat_f64(in.at("pre_dist")), newK = mat_f64(in.at("pre_newK"));
        const int dw = (int)scalar(in, "pre_width");
        const double ratio = (double)rgb.cols / (double)dw;
        const int dh = (int)(rgb.rows / ratio);
        Mat resized, gray, und, cl;
        resize(rgb, resized, Size(dw, dh), 0, 0, INTER_AREA);                                             // VOU:362
        cvtColor(resized, gray, COLOR_RGB2GRAY);                                                           // VOU:365
        undistort(gray, und, K, dist4, newK);                                                              // VOU:368
        Ptr<CLAHE> clahe = createCLAHE(); clahe->setClipLimit(scalar(in, "pre_clip_limit")); clahe->apply(und, cl);   // VOU:372-374
        out["pre_resized"] = from_mat(resized); out["pre_gray"] = from_mat(gray); out["pre_undistorted"] = from_mat(und); out["pre_clahe"] = from_mat(cl);
        Mat K0 = mat_f64(in.at("cam_K")), d0 = mat_f64(in.at("cam_dist"));
        const int ow = (int)scalar(in, "cam_width"), oh = (int)scalar(in, "cam_height"), cw = (int)scalar(in, "cam_desired_width");
        const double r2 = (double)ow / (double)cw; const int ch = (int)(oh / r2);
        Mat Ks = K0 / r2; Ks.at<double>(0, 1) = K0.at<double>(0, 1); Ks.at<double>(2, 2) = 1.0;           // VOU:664-668
        Mat opt = getOptimalNewCameraMatrix(Ks, d0, Size(cw, ch), 0, Size(cw, ch));                        // VOU:670
        out["cam_K_scaled"] = from_mat(Ks); out["cam_newK"] = from_mat(opt);
    }
    npz::save(argv[2], out);
    printf("wrote %zu arrays to %s (OpenCV %s)\n", out.size(), argv[2], CV_VERSION);
    return 0;
}
