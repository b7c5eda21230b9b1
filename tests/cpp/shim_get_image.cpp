// shim_get_image.cpp -- test driver: get_image (VO_utility.cpp:337-379) through the uvo_libraries function surface.
//   usage: shim_get_image <input.bin> <output.bin>
//   input : int32 W, H, DESIRED_WIDTH, CLAHE_CORRECTION, CLIP_LIMIT; f64 K[9], dist[4], newK[9]; H x W x 3 u8
//   output: int32 out_w, out_h; out_h x out_w u8; then f64 K_scaled[9], newK[9] from resize_camera_matrix(img, K, dist, .)
#include <cstdio>
#include <vector>
#include "uvo_libraries_hip/VO_utility_hip.h"
using namespace uvocv;

int main(int argc, char** argv)
{
    if (argc != 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[5]; double cam[22];
    if (fread(hdr, sizeof(int), 5, f) != 5 || fread(cam, sizeof(double), 22, f) != 22) return 2;
    const int W = hdr[0], H = hdr[1];
    DESIRED_WIDTH = hdr[2]; CLAHE_CORRECTION = hdr[3] != 0; CLIP_LIMIT = hdr[4];       // what get_VO_parameters would set
    Mat K(3, 3, CV_64FC1), D(4, 1, CV_64FC1), N(3, 3, CV_64FC1), img(H, W, CV_8UC3);
    for (int i = 0; i < 9; i++) { K.at<double>(i / 3, i % 3) = cam[i]; N.at<double>(i / 3, i % 3) = cam[13 + i]; }
    for (int i = 0; i < 4; i++) D.at<double>(i, 0) = cam[9 + i];
    for (int y = 0; y < H; y++) if (fread(img.ptr<uint8_t>(y), 1, (size_t)W * 3, f) != (size_t)W * 3) return 2;
    fclose(f);
    try {
        Mat out = get_image(img, K, D, N);
        FILE* o = fopen(argv[2], "wb");
        int dims[2] = { out.cols, out.rows };
        fwrite(dims, sizeof(int), 2, o);
        for (int y = 0; y < out.rows; y++) fwrite(out.ptr<uint8_t>(y), 1, (size_t)out.cols, o);
        // resize_camera_matrix (VO_utility.cpp:658-675) on the same camera: K is scaled in place, the optimal new matrix returned
        Mat Ks(3, 3, CV_64FC1), Nn;
        for (int i = 0; i < 9; i++) Ks.at<double>(i / 3, i % 3) = cam[i];
        resize_camera_matrix(img, Ks, D, Nn);
        double both[18];
        for (int i = 0; i < 9; i++) { both[i] = Ks.at<double>(i / 3, i % 3); both[9 + i] = Nn.at<double>(i / 3, i % 3); }
        fwrite(both, sizeof(double), 18, o);
        fclose(o);
    } catch (const uvo_hip::Error& e) {
        fprintf(stderr, "uvo_hip::Error: %s\n", e.what());
        return 1;
    }
    uvo_hip::shutdown();
    return 0;
}
