// shim_vo_node.cpp -- test driver: the ROS-free node class (include/uvo_libraries_hip/visual_odometry_hip.h) fed from files,
// the way the ROS adapter feeds it from topics.  Parameters come from YAML files through the same loaders the node uses.
//
//   usage: shim_vo_node <mono|stereo> <camera_name> <frames.bin> <out.bin> <params.yaml> <intrinsics.yaml>
//   frames.bin: int32 W, H, n; then n x { f64 stamp, f64 range, u8 rgb[H][W][3] (mono) | left rgb, right rgb (stereo) }
//   out.bin   : per frame int32 x 6 (published, valid, n_kps, n_matches, n_inliers, n_good3d) + f64 x 4 (v[3], stamp)
//   --config-only as first argument: load the YAML files, print the globals as "NAME value" lines, touch no GPU.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "uvo_libraries_hip/visual_odometry_hip.h"

using namespace uvocv;

static int dump_config(int argc, char** argv)
{
    uvo_hip::ParamTree tree;
    for (int i = 4; i < argc; i++) tree.load_yaml_file(argv[i]);
    const std::string mode = argv[2], cam = argv[3];
    get_VO_parameters(tree);
    if (mode == "stereo") get_stereo_camera_parameters(tree, cam); else get_mono_camera_parameters(tree, cam);
#define P_I(n) printf(#n " %d\n", (int)n)
#define P_D(n) printf(#n " %.17g\n", (double)n)
    P_I(NODE_FREQ); P_I(DESIRED_WIDTH); P_I(CLAHE_CORRECTION); P_I(CLIP_LIMIT); P_I(DISTANCE); printf("FEATURE_DETECTOR %s\n", FEATURE_DETECTOR.c_str());
    P_D(LOWE_RATIO_THRESHOLD); P_I(ESSENTIAL_OUTLIER_METHOD); P_D(ESSENTIAL_MAX_ITERS); P_D(ESSENTIAL_CONFIDENCE); P_D(ESSENTIAL_THRESHOLD);
    P_I(HOMOGRAPHY_OUTLIER_METHOD); P_D(HOMOGRAPHY_MAX_ITERS); P_D(HOMOGRAPHY_CONFIDENCE); P_D(HOMOGRAPHY_THRESHOLD); P_D(HOMOGRAPHY_DISTANCE);
    P_D(VPF_THRESHOLD); P_D(REPROJECTION_TOLERANCE); P_I(MIN_NUM_FEATURES); P_I(MIN_NUM_3DPOINTS); P_I(MIN_NUM_INLIERS);
    P_I(ITERATIONS_COUNT); P_D(REPROJECTION_ERROR_THRESHOLD); P_D(CONFIDENCE); P_I(USE_EXTRINSIC_GUESS); P_I(PNP_METHOD_FLAG);
    P_I(FPS); P_I(SHOW_MATCHES); P_I(SURF_MIN_HESSIAN); P_I(SURF_OCTAVES_NUMBER); P_I(SURF_OCTAVES_LAYERS); P_I(SURF_EXTENDED); P_I(SURF_UPRIGHT);
    if (mode == "stereo") {
        P_D(fx_left); P_D(fy_left); P_D(ccx_left); P_D(ccy_left); P_D(fx_right); P_D(fy_right); P_D(ccx_right); P_D(ccy_right);
        P_D(k1_left); P_D(k2_left); P_D(p1_left); P_D(p2_left); P_D(k1_right); P_D(k2_right); P_D(p1_right); P_D(p2_right);
        printf("R_right"); for (int i = 0; i < 9 && !R_right.empty(); i++) printf(" %.17g", R_right.at<double>(i / 3, i % 3)); printf("\n");
        printf("t_right"); for (int i = 0; i < 3 && !t_right.empty(); i++) printf(" %.17g", t_right.at<double>(i, 0)); printf("\n");
        printf("R_left_rows %d t_left_rows %d\n", R_left.rows, t_left.rows);
    } else { P_D(fx); P_D(fy); P_D(ccx); P_D(ccy); P_D(k1); P_D(k2); P_D(p1); P_D(p2); }
    return 0;
}

static bool read_rgb(FILE* f, int W, int H, Mat& m)
{
    m = Mat(H, W, CV_8UC3);
    return fread(m.ptr<uint8_t>(0), 1, (size_t)W * H * 3, f) == (size_t)W * H * 3;
}

int main(int argc, char** argv)
{
    try {
        if (argc >= 5 && strcmp(argv[1], "--config-only") == 0) return dump_config(argc, argv);
        if (argc != 7) { fprintf(stderr, "usage: %s mono|stereo camera frames.bin out.bin params.yaml intrinsics.yaml\n", argv[0]); return 2; }
        const std::string mode = argv[1];
        uvo_hip::ParamTree tree;
        tree.load_yaml_file(argv[5]); tree.load_yaml_file(argv[6]);
        FILE* f = fopen(argv[3], "rb");
        if (!f) { perror("frames"); return 2; }
        int hdr[3];
        if (fread(hdr, sizeof(int), 3, f) != 3) { fprintf(stderr, "short header\n"); return 2; }
        const int W = hdr[0], H = hdr[1], n = hdr[2];
        const int max_kpts = getenv("UVO_TEST_MAX_KPTS") ? atoi(getenv("UVO_TEST_MAX_KPTS")) : 8192;       // (ORB::create(10000, ...) returns more than 8192 keypoints)
        uvo_hip::configure(0, W > 640 ? W : 640, H > 480 ? H : 480, max_kpts);
        uvo_hip::visual_odometry_core node(mode, tree, argv[2]);
        FILE* out = fopen(argv[4], "wb");
        if (!out) { perror("out"); return 2; }
        for (int k = 0; k < n; k++) {
            double meta[2];
            if (fread(meta, sizeof(double), 2, f) != 2) { fprintf(stderr, "short frame header\n"); return 2; }
            Mat a, b;
            if (!read_rgb(f, W, H, a)) { fprintf(stderr, "short frame\n"); return 2; }
            if (mode == "stereo") { if (!read_rgb(f, W, H, b)) { fprintf(stderr, "short frame\n"); return 2; } node.stereo_imgs_callback(a, b, meta[0]); }
            else { node.range_callback(meta[1]); node.mono_imgs_callback(a, meta[0]); }
            const uvo_hip::Published p = node.spin_once();
            const int rec[6] = { p.published, p.valid, p.n_kps, p.n_matches, p.n_inliers, p.n_good3d };
            const double vals[4] = { p.v[0], p.v[1], p.v[2], p.stamp };
            fwrite(rec, sizeof(int), 6, out); fwrite(vals, sizeof(double), 4, out);
        }
        fclose(out); fclose(f);
        uvo_hip::shutdown();
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
