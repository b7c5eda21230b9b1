// shim_match_binary.cpp -- test driver: the AKAZE / ORB branch of match_features (VO_utility.cpp:520-524) through the uvo_libraries
// function surface: FEATURE_DETECTOR = "ORB", CV_8U descriptor matrices, results appended to `matches`.
//   usage: shim_match_binary <input.bin> <output.bin>
//   input : int32 n1, n2, bytes; f32 ratio; n1 x bytes u8; n2 x bytes u8
//           bytes < 0: the "SIFT" arm of the L2 branch instead (VO_utility.cpp:525-529): FEATURE_DETECTOR = "SIFT", rows of -bytes f32
//           n1 < 0   : -n1 query rows through the SEVEN-argument overload (VO_utility.cpp:551-573), which applies NORM_L2 to the u8 rows
//   output: int32 m; m x (queryIdx, trainIdx i32, distance f32)
#include <cstdio>
#include <vector>
#include "uvo_libraries_hip/VO_utility_hip.h"
using namespace uvocv;

int main(int argc, char** argv)
{
    if (argc != 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[3]; float ratio;
    if (fread(hdr, sizeof(int), 3, f) != 3 || fread(&ratio, sizeof(float), 1, f) != 1) return 2;
    const bool seven = hdr[0] < 0;
    const int n1 = seven ? -hdr[0] : hdr[0], n2 = hdr[1];
    const bool sift = hdr[2] < 0;
    const int nb = sift ? -hdr[2] : hdr[2];
    Mat d1(n1, nb, sift ? CV_32FC1 : CV_8UC1), d2(n2, nb, sift ? CV_32FC1 : CV_8UC1);
    const size_t rowb = (size_t)nb * (sift ? sizeof(float) : 1);
    for (int i = 0; i < n1; i++) if (fread(d1.ptr<unsigned char>(i), 1, rowb, f) != rowb) return 2;
    for (int i = 0; i < n2; i++) if (fread(d2.ptr<unsigned char>(i), 1, rowb, f) != rowb) return 2;
    fclose(f);
    FEATURE_DETECTOR = sift ? "SIFT" : "ORB";
    LOWE_RATIO_THRESHOLD = ratio;
    try {
        std::vector<KeyPoint> k1((size_t)n1), k2((size_t)n2);
        std::vector<DMatch> matches(1);                       // one stale entry: the reference appends (VOU:538)
        matches[0].queryIdx = -7;
        std::vector<Point2f> c1, c2;
        for (int i = 0; i < n1; i++) k1[(size_t)i].pt.x = (float)i;
        for (int i = 0; i < n2; i++) k2[(size_t)i].pt.x = (float)(1000000 + i);
        if (seven) match_features(k1, k2, d1, d2, matches, c1, c2); else match_features(k1, k2, d1, d2, matches);
        if (matches.empty() || matches[0].queryIdx != -7) return 3;
        if (seven) {                                          // the converted points of the appended matches (VOU:566-567)
            if (c1.size() + 1 != matches.size() || c2.size() != c1.size()) return 3;
            for (size_t i = 0; i < c1.size(); i++) if (c1[i].x != (float)matches[i + 1].queryIdx || c2[i].x != (float)(1000000 + matches[i + 1].trainIdx)) return 3;
        }
        FILE* o = fopen(argv[2], "wb");
        const int m = (int)matches.size() - 1;
        fwrite(&m, sizeof(int), 1, o);
        for (int i = 1; i <= m; i++) { int q[2] = { matches[(size_t)i].queryIdx, matches[(size_t)i].trainIdx }; fwrite(q, sizeof(int), 2, o); fwrite(&matches[(size_t)i].distance, sizeof(float), 1, o); }
        fclose(o);
    } catch (const uvo_hip::Error& e) {
        fprintf(stderr, "uvo_hip::Error: %s\n", e.what());
        return 1;
    }
    uvo_hip::shutdown();
    return 0;
}
