// shim_detect.cpp -- test driver: detect_features (VO_utility.cpp:91-126) through the uvo_libraries function surface, with the
// detector picked by the FEATURE_DETECTOR global as the reference does ("SURF": VOU:114-119, "SIFT": VOU:107-112).
//   usage: shim_detect <input.bin> <output.bin>
//   input : int32 w, h, min_hessian; char name[8] (zero padded); h x w bytes
//   output: int32 n, descriptor columns; n x KeyPoint (28 bytes); n x columns f32 (SURF, SIFT) or n x columns bytes (AKAZE, ORB -- the latter with UVO_ORB_PATTERN_FILE set)
//   exit 4 when detect_features throws for the name (an unserved detector)
#include <cstdio>
#include <cstring>
#include <vector>
#include "uvo_libraries_hip/VO_utility_hip.h"
using namespace uvocv;

int main(int argc, char** argv)
{
    if (argc != 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[3]; char name[9] = {0};
    if (fread(hdr, sizeof(int), 3, f) != 3 || fread(name, 1, 8, f) != 8) return 2;
    const int w = hdr[0], h = hdr[1];
    Mat img(h, w, CV_8UC1);
    for (int y = 0; y < h; y++) if (fread(img.ptr<unsigned char>(y), 1, (size_t)w, f) != (size_t)w) return 2;
    fclose(f);
    FEATURE_DETECTOR = name;
    SURF_MIN_HESSIAN = hdr[2];
    try {
        uvo_hip::configure(0, w, h, 8192);
        std::vector<KeyPoint> kps(3);                         // stale content: detectAndCompute overwrites its outputs
        Mat desc;
        detect_features(img, kps, desc);
        FILE* o = fopen(argv[2], "wb");
        const int n = (int)kps.size(), cols = desc.cols;
        if (desc.rows != n) return 3;
        fwrite(&n, sizeof(int), 1, o); fwrite(&cols, sizeof(int), 1, o);
        static_assert(sizeof(KeyPoint) == 28, "KeyPoint POD");
        if (n) fwrite(static_cast<const void*>(kps.data()), sizeof(KeyPoint), (size_t)n, o);
        if (desc.type() == CV_8UC1) for (int i = 0; i < n; i++) fwrite(desc.ptr<unsigned char>(i), 1, (size_t)cols, o);      // AKAZE: 61-byte rows, ORB: 32
        else for (int i = 0; i < n; i++) fwrite(desc.ptr<float>(i), sizeof(float), (size_t)cols, o);
        fclose(o);
    } catch (const uvo_hip::Error& e) {
        fprintf(stderr, "uvo_hip::Error: %s\n", e.what());
        return 4;
    }
    uvo_hip::shutdown();
    return 0;
}
