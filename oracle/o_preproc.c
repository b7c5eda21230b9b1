/*
 * o_preproc.c -- CPU ORACLE (test infrastructure): restatement of get_image (VOU:337-379), the per-frame image
 * preprocessing in front of the hot path:
 *     cv::resize(INTER_AREA) -> cv::cvtColor(COLOR_RGB2GRAY) -> cv::undistort -> optional cv::CLAHE::apply.
 * [UPSTREAM] imgproc: resize.cpp (resizeArea_, resizeAreaFast_ -- shared with o_surf.c), color_rgb (RGB2Gray<uchar>,
 * 15-bit coefficients), undistort.dispatch.cpp (undistort in stripes of (1<<12)/cols rows, initUndistortRectifyMap
 * baseline path with CV_16SC2 fixed-point maps), imgwarp.cpp (remapBilinear, INTER_BITS = 5, INTER_REMAP_COEF_BITS = 15,
 * BORDER_CONSTANT 0), clahe.cpp (8x8 tiles, clip + redistribute, float bilinear blend of the tile LUTs).
 * PARITY UNPINNED vs OpenCV: recalled behaviour; the items of medium confidence are the RGB2GRAY coefficient set
 * (9798/19235/3735 >> 15), the stripe-wise map evaluation of undistort and CLAHE's rounding (cvRound).  OpenCV builds
 * with AVX2 dispatch evaluate the map lines with vector code whose rounding may differ from this baseline path.
 */
#include "uvo_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

/* cvtColor(COLOR_RGB2GRAY), 8U: (R*9798 + G*19235 + B*3735 + (1 << 14)) >> 15 */
void orc_rgb2gray_u8(const uint8_t* rgb, int w, int h, int stride, uint8_t* gray)
{
    for (int y = 0; y < h; y++) {
        const uint8_t* s = rgb + (size_t)y * stride;
        uint8_t* d = gray + (size_t)y * w;
        for (int x = 0; x < w; x++) d[x] = (uint8_t)((s[3*x] * 9798 + s[3*x + 1] * 19235 + s[3*x + 2] * 3735 + (1 << 14)) >> 15);
    }
}

/* resize(INTER_AREA) of an interleaved 3-channel image: every channel is the single-channel resize of its plane
 * (resizeArea_ / resizeAreaFast_ keep one accumulator per destination element and channel, same tap order). */
int orc_resize_area_u8c3(const uint8_t* src, int sw, int sh, int stride, uint8_t* dst, int dw, int dh)
{
    if (dw > sw || dh > sh) return -1;               /* enlarging takes OpenCV's INTER_LINEAR path: outside this restatement */
    uint8_t* plane = (uint8_t*)malloc((size_t)sw * sh + (size_t)dw * dh);
    uint8_t* out = plane + (size_t)sw * sh;
    for (int c = 0; c < 3; c++) {
        for (int y = 0; y < sh; y++) for (int x = 0; x < sw; x++) plane[(size_t)y * sw + x] = src[(size_t)y * stride + 3*x + c];
        orc_resize_area_u8(plane, sw, sh, out, dw, dh);
        for (int i = 0; i < dw * dh; i++) dst[(size_t)3*i + c] = out[i];
    }
    free(plane);
    return 0;
}

/* cv::invert of a 3x3 double matrix (DECOMP_LU takes the closed form for n <= 3) */
static int invert3(const double* S, double* t)
{
    double d = S[0]*(S[4]*S[8] - S[5]*S[7]) - S[1]*(S[3]*S[8] - S[5]*S[6]) + S[2]*(S[3]*S[7] - S[4]*S[6]);
    if (d == 0) { memset(t, 0, sizeof(double) * 9); return 0; }
    d = 1./d;
    t[0] = (S[4]*S[8] - S[5]*S[7]) * d; t[1] = (S[2]*S[7] - S[1]*S[8]) * d; t[2] = (S[1]*S[5] - S[2]*S[4]) * d;
    t[3] = (S[5]*S[6] - S[3]*S[8]) * d; t[4] = (S[0]*S[8] - S[2]*S[6]) * d; t[5] = (S[2]*S[3] - S[0]*S[5]) * d;
    t[6] = (S[3]*S[7] - S[4]*S[6]) * d; t[7] = (S[1]*S[6] - S[0]*S[7]) * d; t[8] = (S[0]*S[4] - S[1]*S[3]) * d;
    return 1;
}

static int sat_int_d(double v) { return orc_cvRound(v); }    /* saturate_cast<int>(double) = cvRound (no overflow here) */

/* initUndistortRectifyMap(A, dist(k1,k2,p1,p2), R = I, Ar, size, CV_16SC2): map1 = (x, y) shorts, map2 = 5+5 fraction bits.
 * The running sums _x, _y, _w are advanced by one addition per column, as the reference loop does. */
void orc_init_undistort_map(const double* A, const double* dist4, const double* Ar, int cols, int rows, int16_t* map1, uint16_t* map2)
{
    double ir[9];
    invert3(Ar, ir);                                 /* (Ar * I).inv(DECOMP_LU) */
    const double u0 = A[2], v0 = A[5], fx = A[0], fy = A[4];
    const double k1 = dist4[0], k2 = dist4[1], p1 = dist4[2], p2 = dist4[3];
    const double k3 = 0, k4 = 0, k5 = 0, k6 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    for (int i = 0; i < rows; i++) {
        int16_t* m1 = map1 + (size_t)i * cols * 2;
        uint16_t* m2 = map2 + (size_t)i * cols;
        double _x = i*ir[1] + ir[2], _y = i*ir[4] + ir[5], _w = i*ir[7] + ir[8];
        for (int j = 0; j < cols; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
            double w = 1./_w, x = _x*w, y = _y*w;
            double x2 = x*x, y2 = y*y;
            double r2 = x2 + y2, _2xy = 2*x*y;
            double kr = (1 + ((k3*r2 + k2)*r2 + k1)*r2)/(1 + ((k6*r2 + k5)*r2 + k4)*r2);
            double xd = (x*kr + p1*_2xy + p2*(r2 + 2*x2) + s1*r2 + s2*r2*r2);
            double yd = (y*kr + p1*(r2 + 2*y2) + p2*_2xy + s3*r2 + s4*r2*r2);
            /* matTilt = identity: vecTilt = (xd, yd, 1), invProj = 1 */
            double invProj = 1.0;
            double u = fx*invProj*xd + u0;
            double v = fy*invProj*yd + v0;
            int iu = sat_int_d(u*32), iv = sat_int_d(v*32);
            m1[j*2] = (int16_t)(iu >> 5); m1[j*2 + 1] = (int16_t)(iv >> 5);
            m2[j] = (uint16_t)((iv & 31)*32 + (iu & 31));
        }
    }
}

/* remap(INTER_LINEAR, BORDER_CONSTANT = 0) of an 8UC1 image with fixed-point maps.  Weights (32-fx)(32-fy)*32 etc. are
 * exact in OpenCV's short table, so no sum correction applies; FixedPtCast<int, uchar, 15>. */
void orc_remap_bilinear_u8(const uint8_t* src, int sw, int sh, const int16_t* map1, const uint16_t* map2,
                           uint8_t* dst, int dw, int dh)
{
    for (int dy = 0; dy < dh; dy++)
        for (int dx = 0; dx < dw; dx++) {
            int sx = map1[((size_t)dy * dw + dx) * 2], sy = map1[((size_t)dy * dw + dx) * 2 + 1];
            int f = map2[(size_t)dy * dw + dx], fxi = f & 31, fyi = f >> 5;
            int w00 = (32 - fxi) * (32 - fyi) * 32, w01 = fxi * (32 - fyi) * 32, w10 = (32 - fxi) * fyi * 32, w11 = fxi * fyi * 32;
            int v00 = 0, v01 = 0, v10 = 0, v11 = 0;                  /* border value 0 for taps outside the image */
            if (sy >= 0 && sy < sh) { if (sx >= 0 && sx < sw) v00 = src[(size_t)sy * sw + sx]; if (sx + 1 >= 0 && sx + 1 < sw) v01 = src[(size_t)sy * sw + sx + 1]; }
            if (sy + 1 >= 0 && sy + 1 < sh) { if (sx >= 0 && sx < sw) v10 = src[(size_t)(sy + 1) * sw + sx]; if (sx + 1 >= 0 && sx + 1 < sw) v11 = src[(size_t)(sy + 1) * sw + sx + 1]; }
            int val = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
            dst[(size_t)dy * dw + dx] = (uint8_t)(val < 0 ? 0 : val > 255 ? 255 : val);
        }
}

/* cv::undistort(src, dst, K, dist, newK): stripes of max(1, 4096/cols) rows, each with its own map (Ar(1,2) = v0 - y) */
void orc_undistort_u8(const uint8_t* src, int w, int h, const double* K, const double* dist4, const double* newK, uint8_t* dst)
{
    int stripe0 = (1 << 12) / (w > 1 ? w : 1);
    if (stripe0 < 1) stripe0 = 1;
    if (stripe0 > h) stripe0 = h;
    int16_t* m1 = (int16_t*)malloc(sizeof(int16_t) * 2 * (size_t)stripe0 * w);
    uint16_t* m2 = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)stripe0 * w);
    double Ar[9];
    memcpy(Ar, newK, sizeof(Ar));
    const double v0 = Ar[5];
    for (int y = 0; y < h; y += stripe0) {
        int ss = stripe0 < h - y ? stripe0 : h - y;
        Ar[5] = v0 - y;
        orc_init_undistort_map(K, dist4, Ar, w, ss, m1, m2);
        orc_remap_bilinear_u8(src, w, h, m1, m2, dst + (size_t)y * w, w, ss);
    }
    free(m1); free(m2);
}

static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * (len - 1) - p; }
    return p;
}

/* cv::createCLAHE(clipLimit, Size(8,8))->apply on 8UC1 */
void orc_clahe_u8(const uint8_t* src, int w, int h, double clip_limit, uint8_t* dst)
{
    const int tilesX = 8, tilesY = 8, histSize = 256;
    int ew = w, eh = h;
    const uint8_t* lut_src = src;
    uint8_t* ext = NULL;
    if (w % tilesX != 0 || h % tilesY != 0) {
        ew = w + (tilesX - (w % tilesX)); eh = h + (tilesY - (h % tilesY));       /* copyMakeBorder(..., BORDER_REFLECT_101) */
        ext = (uint8_t*)malloc((size_t)ew * eh);
        for (int y = 0; y < eh; y++) for (int x = 0; x < ew; x++) ext[(size_t)y * ew + x] = src[(size_t)reflect101(y, h) * w + reflect101(x, w)];
        lut_src = ext;
    }
    const int tw = ew / tilesX, th = eh / tilesY;
    const int tileSizeTotal = tw * th;
    const float lutScale = (float)(histSize - 1) / tileSizeTotal;
    int clipLimit = 0;
    if (clip_limit > 0.0) { clipLimit = (int)(clip_limit * tileSizeTotal / histSize); if (clipLimit < 1) clipLimit = 1; }
    uint8_t* lut = (uint8_t*)malloc((size_t)tilesX * tilesY * histSize);
    for (int k = 0; k < tilesX * tilesY; k++) {
        const int ty = k / tilesX, tx = k % tilesX;
        int hist[256];
        memset(hist, 0, sizeof(hist));
        for (int y = 0; y < th; y++) for (int x = 0; x < tw; x++) hist[lut_src[(size_t)(ty * th + y) * ew + tx * tw + x]]++;
        if (clipLimit > 0) {
            int clipped = 0;
            for (int i = 0; i < histSize; i++) if (hist[i] > clipLimit) { clipped += hist[i] - clipLimit; hist[i] = clipLimit; }
            int redistBatch = clipped / histSize;
            int residual = clipped - redistBatch * histSize;
            for (int i = 0; i < histSize; i++) hist[i] += redistBatch;
            if (residual != 0) {
                int residualStep = histSize / residual; if (residualStep < 1) residualStep = 1;
                for (int i = 0; i < histSize && residual > 0; i += residualStep, residual--) hist[i]++;
            }
        }
        int sum = 0;
        uint8_t* tl = lut + (size_t)k * histSize;
        for (int i = 0; i < histSize; i++) {
            sum += hist[i];
            int v = orc_cvRoundf(sum * lutScale);
            tl[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    for (int y = 0; y < h; y++) {
        float tyf = y * inv_th - 0.5f;
        int ty1 = orc_cvFloor(tyf), ty2 = ty1 + 1;
        float ya = tyf - ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > tilesY - 1) ty2 = tilesY - 1;
        const uint8_t* p1 = lut + (size_t)ty1 * tilesX * histSize;
        const uint8_t* p2 = lut + (size_t)ty2 * tilesX * histSize;
        for (int x = 0; x < w; x++) {
            float txf = x * inv_tw - 0.5f;
            int tx1 = orc_cvFloor(txf), tx2 = tx1 + 1;
            float xa = txf - tx1, xa1 = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > tilesX - 1) tx2 = tilesX - 1;
            int sv = src[(size_t)y * w + x];
            int ind1 = tx1 * histSize + sv, ind2 = tx2 * histSize + sv;
            float res = (p1[ind1] * xa1 + p1[ind2] * xa) * ya1 + (p2[ind1] * xa1 + p2[ind2] * xa) * ya;
            int v = orc_cvRoundf(res);
            dst[(size_t)y * w + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    free(lut); free(ext);
}

/* get_image (VOU:337-379).  rgb: h x w x 3 interleaved; out: desired_height x desired_width gray.  Returns 0, or -1
 * when the image would have to be enlarged. */
int orc_get_image(const uint8_t* rgb, int w, int h, int stride, int desired_width, const double* K, const double* dist4,
                  const double* newK, int clahe_on, int clip_limit, uint8_t* out, int* out_w, int* out_h)
{
    double ratio = (double)w / (double)desired_width;
    int desired_height = (int)(h / ratio);
    *out_w = desired_width; *out_h = desired_height;
    uint8_t* gray = (uint8_t*)malloc((size_t)desired_width * desired_height);
    if (w == desired_width && h == desired_height) {
        orc_rgb2gray_u8(rgb, w, h, stride, gray);
    } else {
        uint8_t* small = (uint8_t*)malloc((size_t)desired_width * desired_height * 3);
        if (orc_resize_area_u8c3(rgb, w, h, stride, small, desired_width, desired_height) != 0) { free(small); free(gray); return -1; }
        orc_rgb2gray_u8(small, desired_width, desired_height, desired_width * 3, gray);
        free(small);
    }
    orc_undistort_u8(gray, desired_width, desired_height, K, dist4, newK, out);
    if (clahe_on) {
        memcpy(gray, out, (size_t)desired_width * desired_height);        /* clahe->apply(img, img): in place */
        orc_clahe_u8(gray, desired_width, desired_height, (double)clip_limit, out);
    }
    free(gray);
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * resize_camera_matrix (reference uvo_libraries/src/VO_utility.cpp:658-675): K scaled in place by the width ratio (skew and
 * K[2][2] restored), then cv::getOptimalNewCameraMatrix(K, dist, Size(dw, dh), alpha = 0, Size(dw, dh), validPixROI = 0,
 * centerPrincipalPoint = false).  OpenCV 4.5 calib3d (calibration.cpp cvGetOptimalNewCameraMatrix / icvGetRectangles,
 * undistort.dispatch.cpp cvUndistortPointsInternal), restated from the published algorithm -- PARITY UNPINNED, like the
 * rest of this file: a 9 x 9 grid of image points (x*(w-1)/8, y*(h-1)/8) is undistorted to normalised coordinates (five
 * fixed-point iterations, no R, no P), the inscribed rectangle of the grid's border is mapped onto the viewport.
 * ------------------------------------------------------------------------------------------- */
static void undistort_point_normalised(double u, double v, const double* K, const double* d4, double* xo, double* yo)
{
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5], ifx = 1. / fx, ify = 1. / fy;
    const double k1 = d4[0], k2 = d4[1], p1 = d4[2], p2 = d4[3];
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        /* k3..k6 = 0: icdist = (1 + ((k[7]*r2 + k[6])*r2 + k[5])*r2) / (1 + ((k[4]*r2 + k[1])*r2 + k[0])*r2) */
        double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((0 * r2 + k2) * r2 + k1) * r2);
        if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }        /* "test: undistortPoints.regression_14583" */
        const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x) + 0 * r2 + 0 * r2 * r2;
        const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y + 0 * r2 + 0 * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    *xo = x; *yo = y;
}

int orc_resize_camera_matrix(int original_width, int original_height, int desired_width, double* K, const double* dist4, double* newK,
                             int* desired_height_out)
{
    if (desired_width <= 0 || original_width <= 0 || original_height <= 0) return -1;
    const double ratio = (double)original_width / (double)desired_width;
    const int desired_height = (int)(original_height / ratio);
    if (desired_height_out) *desired_height_out = desired_height;
    const double skew = K[1];
    for (int i = 0; i < 9; i++) K[i] = K[i] / ratio;
    K[1] = skew; K[8] = 1;
    /* icvGetRectangles */
    const int N = 9;
    double iX0 = -DBL_MAX, iX1 = DBL_MAX, iY0 = -DBL_MAX, iY1 = DBL_MAX;
    double oX0 = DBL_MAX, oX1 = -DBL_MAX, oY0 = DBL_MAX, oY1 = -DBL_MAX;
    for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++) {
            double px, py;
            undistort_point_normalised((double)x * (desired_width - 1) / (N - 1), (double)y * (desired_height - 1) / (N - 1), K, dist4, &px, &py);
            oX0 = oX0 < px ? oX0 : px; oX1 = oX1 > px ? oX1 : px; oY0 = oY0 < py ? oY0 : py; oY1 = oY1 > py ? oY1 : py;
            if (x == 0) iX0 = iX0 > px ? iX0 : px;
            if (x == N - 1) iX1 = iX1 < px ? iX1 : px;
            if (y == 0) iY0 = iY0 > py ? iY0 : py;
            if (y == N - 1) iY1 = iY1 < py ? iY1 : py;
        }
    const double inner_x = iX0, inner_y = iY0, inner_w = iX1 - iX0, inner_h = iY1 - iY0;
    const double outer_x = oX0, outer_y = oY0, outer_w = oX1 - oX0, outer_h = oY1 - oY0;
    const double alpha = 0;
    const double fx0 = (desired_width - 1) / inner_w, fy0 = (desired_height - 1) / inner_h;
    const double cx0 = -fx0 * inner_x, cy0 = -fy0 * inner_y;
    const double fx1 = (desired_width - 1) / outer_w, fy1 = (desired_height - 1) / outer_h;
    const double cx1 = -fx1 * outer_x, cy1 = -fy1 * outer_y;
    for (int i = 0; i < 9; i++) newK[i] = K[i];                      /* cvConvert(cameraMatrix, &matM): the other entries are K's */
    newK[0] = fx0 * (1 - alpha) + fx1 * alpha;
    newK[4] = fy0 * (1 - alpha) + fy1 * alpha;
    newK[2] = cx0 * (1 - alpha) + cx1 * alpha;
    newK[5] = cy0 * (1 - alpha) + cy1 * alpha;
    return 0;
}
