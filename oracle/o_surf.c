/*
 * o_surf.c -- CPU ORACLE (test infrastructure): upright SURF-64 detect + describe.
 * Follows VOU:114-119 (SURF::create(minHessian, nOctaves, nOctaveLayers, extended,
 * upright)->detectAndCompute(img, noArray(), kps, desc)) and restates [UPSTREAM]
 * opencv_contrib xfeatures2d/src/surf.cpp (fastHessianDetector, calcLayerDetAndTrace,
 * resizeHaarPattern, calcHaarPattern, findMaximaInLayer, interpolateKeypoint,
 * KeypointGreater, SURFInvoker), imgproc integral (CV_32S) and resize INTER_AREA (u8).
 * SURVEY.md App. A.1.  PARITY UNPINNED vs OpenCV (see uvo_oracle.h).
 * The branch the reference configures is extended=false, upright=true; the other values of the two flags
 * (orientation assignment with the rotated sampling window, the 128-element descriptor: SURVEY.md 8(f) N4) are
 * restated as well, from the same SURFInvoker.  sin / cos of the orientation are orc_sincos (shared with the HIP
 * path operation for operation) narrowed to float, where OpenCV calls std::sin / std::cos on a float.
 */
#include "uvo_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

#define SURF_HAAR_SIZE0    9
#define SURF_HAAR_SIZE_INC 6
#define PATCH_SZ           20
#define SURF_DESC_SIGMA    3.3f

/* [UPSTREAM] imgproc integral(): sum is (h+1) x (w+1), first row/col zero */
void orc_integral_u8(const uint8_t* img, int w, int h, int stride, int32_t* sum)
{
    int sw = w + 1;
    memset(sum, 0, sizeof(int32_t) * sw);
    for (int y = 0; y < h; y++) {
        int32_t s = 0;
        const uint8_t* src = img + (size_t)y * stride;
        int32_t* prev = sum + (size_t)y * sw;
        int32_t* cur = prev + sw;
        cur[0] = 0;
        for (int x = 0; x < w; x++) { s += src[x]; cur[x + 1] = prev[x + 1] + s; }
    }
}

typedef struct { int p0, p1, p2, p3; float w; } SurfHF;

/* [UPSTREAM] surf.cpp resizeHaarPattern */
static void resize_haar_pattern(const int src[][5], SurfHF* dst, int n, int oldSize, int newSize, int widthStep)
{
    float ratio = (float)newSize / oldSize;
    for (int k = 0; k < n; k++) {
        int dx1 = orc_cvRoundf(ratio * src[k][0]);
        int dy1 = orc_cvRoundf(ratio * src[k][1]);
        int dx2 = orc_cvRoundf(ratio * src[k][2]);
        int dy2 = orc_cvRoundf(ratio * src[k][3]);
        dst[k].p0 = dy1 * widthStep + dx1;
        dst[k].p1 = dy2 * widthStep + dx1;
        dst[k].p2 = dy1 * widthStep + dx2;
        dst[k].p3 = dy2 * widthStep + dx2;
        dst[k].w = src[k][4] / ((float)(dx2 - dx1) * (dy2 - dy1));
    }
}

/* [UPSTREAM] surf.cpp calcHaarPattern: int box sum * float weight, accumulated in double */
static inline float calc_haar_pattern(const int32_t* origin, const SurfHF* f, int n)
{
    double d = 0;
    for (int k = 0; k < n; k++)
        d += (origin[f[k].p0] + origin[f[k].p3] - origin[f[k].p1] - origin[f[k].p2]) * f[k].w;
    return (float)d;
}

/* [UPSTREAM] surf.cpp calcLayerDetAndTrace.  sum is (h+1) x (w+1). */
void orc_surf_layer(const int32_t* sum, int w, int h, int size, int sampleStep, float* det, float* trace)
{
    static const int dx_s[3][5]  = { {0, 2, 3, 7, 1}, {3, 2, 6, 7, -2}, {6, 2, 9, 7, 1} };
    static const int dy_s[3][5]  = { {2, 0, 7, 3, 1}, {2, 3, 7, 6, -2}, {2, 6, 7, 9, 1} };
    static const int dxy_s[4][5] = { {1, 1, 4, 4, 1}, {5, 1, 8, 4, -1}, {1, 5, 4, 8, -1}, {5, 5, 8, 8, 1} };
    SurfHF Dx[3], Dy[3], Dxy[4];
    int sum_rows = h + 1, sum_cols = w + 1;
    if (size > sum_rows - 1 || size > sum_cols - 1) return;
    resize_haar_pattern(dx_s, Dx, 3, 9, size, sum_cols);
    resize_haar_pattern(dy_s, Dy, 3, 9, size, sum_cols);
    resize_haar_pattern(dxy_s, Dxy, 4, 9, size, sum_cols);
    int samples_i = 1 + (sum_rows - 1 - size) / sampleStep;
    int samples_j = 1 + (sum_cols - 1 - size) / sampleStep;
    int margin = (size / 2) / sampleStep;
    int cols = w / sampleStep;
    for (int i = 0; i < samples_i; i++) {
        const int32_t* sum_ptr = sum + (size_t)(i * sampleStep) * sum_cols;
        float* det_ptr = det + (size_t)(i + margin) * cols + margin;
        float* trace_ptr = trace + (size_t)(i + margin) * cols + margin;
        for (int j = 0; j < samples_j; j++) {
            float dx  = calc_haar_pattern(sum_ptr, Dx, 3);
            float dy  = calc_haar_pattern(sum_ptr, Dy, 3);
            float dxy = calc_haar_pattern(sum_ptr, Dxy, 4);
            sum_ptr += sampleStep;
            det_ptr[j] = dx * dy - 0.81f * dxy * dxy;
            trace_ptr[j] = dx + dy;
        }
    }
}

/* [UPSTREAM] Matx33f::solve(b, DECOMP_LU) -> Matx_FastSolveOp<float,3,3,1> (Cramer, float) */
static int solve3f(const float a[3][3], const float b[3], float x[3])
{
    float d = (float)(double)(a[0][0]*(a[1][1]*a[2][2] - a[2][1]*a[1][2]) -
                              a[0][1]*(a[1][0]*a[2][2] - a[2][0]*a[1][2]) +
                              a[0][2]*(a[1][0]*a[2][1] - a[2][0]*a[1][1]));
    if (d == 0) { x[0] = x[1] = x[2] = 0; return 0; }
    d = 1/d;
    x[0] = d*(b[0]*(a[1][1]*a[2][2] - a[1][2]*a[2][1]) -
              a[0][1]*(b[1]*a[2][2] - a[1][2]*b[2]) +
              a[0][2]*(b[1]*a[2][1] - a[1][1]*b[2]));
    x[1] = d*(a[0][0]*(b[1]*a[2][2] - a[1][2]*b[2]) -
              b[0]*(a[1][0]*a[2][2] - a[1][2]*a[2][0]) +
              a[0][2]*(a[1][0]*b[2] - b[1]*a[2][0]));
    x[2] = d*(a[0][0]*(a[1][1]*b[2] - b[1]*a[2][1]) -
              a[0][1]*(a[1][0]*b[2] - b[1]*a[2][0]) +
              b[0]*(a[1][0]*a[2][1] - a[1][1]*a[2][0]));
    return 1;
}

/* [UPSTREAM] surf.cpp interpolateKeypoint */
static int interpolate_keypoint(float N9[3][9], int dx, int dy, int ds, orc_keypoint* kpt)
{
    float b[3] = { -(N9[1][5]-N9[1][3])/2, -(N9[1][7]-N9[1][1])/2, -(N9[2][4]-N9[0][4])/2 };
    float A[3][3] = {
        { N9[1][3]-2*N9[1][4]+N9[1][5], (N9[1][8]-N9[1][6]-N9[1][2]+N9[1][0])/4, (N9[2][5]-N9[2][3]-N9[0][5]+N9[0][3])/4 },
        { (N9[1][8]-N9[1][6]-N9[1][2]+N9[1][0])/4, N9[1][1]-2*N9[1][4]+N9[1][7], (N9[2][7]-N9[2][1]-N9[0][7]+N9[0][1])/4 },
        { (N9[2][5]-N9[2][3]-N9[0][5]+N9[0][3])/4, (N9[2][7]-N9[2][1]-N9[0][7]+N9[0][1])/4, N9[0][4]-2*N9[1][4]+N9[2][4] } };
    float x[3];
    solve3f(A, b, x);
    int ok = (x[0] != 0 || x[1] != 0 || x[2] != 0) &&
             fabsf(x[0]) <= 1 && fabsf(x[1]) <= 1 && fabsf(x[2]) <= 1;
    if (ok) {
        kpt->x += x[0]*dx;
        kpt->y += x[1]*dy;
        kpt->size = (float)orc_cvRoundf(kpt->size + x[2]*ds);
    }
    return ok;
}

/* [UPSTREAM] surf.cpp KeypointGreater */
static int keypoint_greater(const orc_keypoint* a, const orc_keypoint* b)
{
    if (a->response > b->response) return 1;
    if (a->response < b->response) return 0;
    if (a->size > b->size) return 1;
    if (a->size < b->size) return 0;
    if (a->octave > b->octave) return 1;
    if (a->octave < b->octave) return 0;
    if (a->y < b->y) return 0;
    if (a->y > b->y) return 1;
    return a->x < b->x;
}
static int kp_cmp(const void* pa, const void* pb)
{
    const orc_keypoint* a = (const orc_keypoint*)pa; const orc_keypoint* b = (const orc_keypoint*)pb;
    if (keypoint_greater(a, b)) return -1;
    if (keypoint_greater(b, a)) return 1;
    return (a->class_id > b->class_id) - (a->class_id < b->class_id); /* total order for identical keys */
}

/* [UPSTREAM] imgproc getGaussianKernel(n, sigma, CV_32F) (4.5: double kernel, cast at the end) */
void orc_gaussian_kernel_f32(int n, double sigma, float* out)
{
    double t[64], sum = 0;
    double scale2X = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; i++) { double x = i - (n - 1) * 0.5; t[i] = exp(scale2X * x * x); sum += t[i]; }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) out[i] = (float)(t[i] * sum);
}

/* [UPSTREAM] imgproc resize.cpp: INTER_AREA, u8, 1 channel, scale >= 1 in both directions */
typedef struct { int si, di; float alpha; } DecimateAlpha;
static int compute_resize_area_tab(int ssize, int dsize, double scale, DecimateAlpha* tab)
{
    int k = 0;
    for (int dx = 0; dx < dsize; dx++) {
        double fsx1 = dx * scale;
        double fsx2 = fsx1 + scale;
        double cellWidth = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = orc_cvCeil(fsx1), sx2 = orc_cvFloor(fsx2);
        sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
        sx1 = sx1 < sx2 ? sx1 : sx2;
        if (sx1 - fsx1 > 1e-3) { tab[k].di = dx; tab[k].si = sx1 - 1; tab[k++].alpha = (float)((sx1 - fsx1) / cellWidth); }
        for (int sx = sx1; sx < sx2; sx++) { tab[k].di = dx; tab[k].si = sx; tab[k++].alpha = (float)(1.0 / cellWidth); }
        if (fsx2 - sx2 > 1e-3) {
            double a = fsx2 - sx2; if (a > 1.) a = 1.; if (a > cellWidth) a = cellWidth;
            tab[k].di = dx; tab[k].si = sx2; tab[k++].alpha = (float)(a / cellWidth);
        }
    }
    return k;
}
static uint8_t sat_u8_f(float v) { int iv = orc_cvRoundf(v); return (uint8_t)(iv < 0 ? 0 : iv > 255 ? 255 : iv); }

void orc_resize_area_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int iscale_x = orc_cvRound(scale_x), iscale_y = orc_cvRound(scale_y);
    int is_area_fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
    if (is_area_fast) {
        /* resizeAreaFast_<uchar,int,...>; 2x2 uses (sum+2)>>2, otherwise saturate_cast(sum * (1.f/area)) */
        int area = iscale_x * iscale_y;
        float scale = 1.f / (area);
        int dwidth1 = (sw / iscale_x);
        for (int dy = 0; dy < dh; dy++) {
            uint8_t* D = dst + (size_t)dy * dw;
            int sy0 = dy * iscale_y;
            int w = sy0 + iscale_y <= sh ? dwidth1 : 0;
            if (sy0 >= sh) { for (int dx = 0; dx < dw; dx++) D[dx] = 0; continue; }
            int dx = 0;
            if (iscale_x == 2 && iscale_y == 2) {
                for (; dx < w; dx++) {
                    const uint8_t* S = src + (size_t)sy0 * sw + dx * 2;
                    D[dx] = (uint8_t)((S[0] + S[1] + S[sw] + S[sw + 1] + 2) >> 2);
                }
            }
            for (; dx < w; dx++) {
                const uint8_t* S = src + (size_t)sy0 * sw + dx * iscale_x;
                int sum = 0;
                for (int sy = 0; sy < iscale_y; sy++) for (int sx = 0; sx < iscale_x; sx++) sum += S[sy * sw + sx];
                D[dx] = sat_u8_f(sum * scale);
            }
            for (; dx < dw; dx++) {
                int sum = 0, count = 0, sx0 = dx * iscale_x;
                if (sx0 >= sw) D[dx] = 0;
                for (int sy = 0; sy < iscale_y; sy++) {
                    if (sy0 + sy >= sh) break;
                    for (int sx = 0; sx < iscale_x; sx++) {
                        if (sx0 + sx >= sw) break;
                        sum += src[(size_t)(sy0 + sy) * sw + sx0 + sx]; count++;
                    }
                }
                D[dx] = sat_u8_f((float)sum / count);
            }
        }
        return;
    }
    DecimateAlpha* xtab = (DecimateAlpha*)malloc(sizeof(DecimateAlpha) * (size_t)(sw + sh) * 2);
    DecimateAlpha* ytab = xtab + sw * 2;
    int xtab_size = compute_resize_area_tab(sw, dw, scale_x, xtab);
    int ytab_size = compute_resize_area_tab(sh, dh, scale_y, ytab);
    float* buf = (float*)malloc(sizeof(float) * dw * 2);
    float* sum = buf + dw;
    int prev_dy = ytab[0].di;
    for (int dx = 0; dx < dw; dx++) sum[dx] = 0;
    for (int j = 0; j < ytab_size; j++) {
        float beta = ytab[j].alpha;
        int dy = ytab[j].di, sy = ytab[j].si;
        const uint8_t* S = src + (size_t)sy * sw;
        for (int dx = 0; dx < dw; dx++) buf[dx] = 0;
        for (int k = 0; k < xtab_size; k++) { int dxn = xtab[k].di; float alpha = xtab[k].alpha; buf[dxn] += S[xtab[k].si] * alpha; }
        if (dy != prev_dy) {
            uint8_t* D = dst + (size_t)prev_dy * dw;
            for (int dx = 0; dx < dw; dx++) { D[dx] = sat_u8_f(sum[dx]); sum[dx] = beta * buf[dx]; }
            prev_dy = dy;
        } else {
            for (int dx = 0; dx < dw; dx++) sum[dx] += beta * buf[dx];
        }
    }
    { uint8_t* D = dst + (size_t)prev_dy * dw; for (int dx = 0; dx < dw; dx++) D[dx] = sat_u8_f(sum[dx]); }
    free(buf); free(xtab);
}

#define ORI_RADIUS          6
#define ORI_WIN             60
#define SURF_ORI_SEARCH_INC 5
#define SURF_ORI_SIGMA      2.5f
#define ORI_MAX_SAMPLES     ((2 * ORI_RADIUS + 1) * (2 * ORI_RADIUS + 1))

/* [UPSTREAM] core mathfuncs_core: cv::fastAtan2 (degrees), also what cv::phase(.., angleInDegrees = true) evaluates per element */
float orc_fast_atan2(float y, float x)
{
    const float s = (float)(180 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s, p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)DBL_EPSILON); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)DBL_EPSILON); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

typedef struct { int n; int x[ORI_MAX_SAMPLES], y[ORI_MAX_SAMPLES]; float w[ORI_MAX_SAMPLES]; } OriTab;

/* SURFInvoker ctor: the sampling pattern of the orientation assignment (points of the 13 x 13 grid inside radius 6, Gaussian weights) */
static void make_ori_tab(OriTab* t)
{
    float G[2 * ORI_RADIUS + 1];
    orc_gaussian_kernel_f32(2 * ORI_RADIUS + 1, SURF_ORI_SIGMA, G);
    t->n = 0;
    for (int i = -ORI_RADIUS; i <= ORI_RADIUS; i++)
        for (int j = -ORI_RADIUS; j <= ORI_RADIUS; j++)
            if (i * i + j * j <= ORI_RADIUS * ORI_RADIUS) { t->x[t->n] = i; t->y[t->n] = j; t->w[t->n] = G[i + ORI_RADIUS] * G[j + ORI_RADIUS]; t->n++; }
}

/* [UPSTREAM] surf.cpp SURFInvoker::operator() for one keypoint: orientation (unless upright), sampling window, INTER_AREA
 * resize to 21 x 21, Haar responses, 4 x 4 cells of 4 (or, extended, 8) sums, normalisation.  `sum` is the integral image.
 * Returns 0 if the keypoint is marked for deletion (size = -1). */
static int surf_describe(const uint8_t* img, const int32_t* sum, int w, int h, int stride, const float* DW, const OriTab* ot, int upright, int extended,
                         orc_keypoint* kp, float* vec, uint8_t* winbuf)
{
    static const int dx_s[2][5] = { {0, 0, 2, 4, -1}, {2, 0, 4, 4, 1} };
    static const int dy_s[2][5] = { {0, 0, 4, 2, 1}, {0, 2, 4, 4, -1} };
    uint8_t PATCH[PATCH_SZ + 1][PATCH_SZ + 1];
    float DX[PATCH_SZ][PATCH_SZ], DY[PATCH_SZ][PATCH_SZ];
    const int dsize = extended ? 128 : 64;
    float size = kp->size;
    float s = size * 1.2f / 9.0f;
    int grad_wav_size = 2 * orc_cvRoundf(2 * s);
    const int sum_rows = h + 1, sum_cols = w + 1;
    if (sum_rows < grad_wav_size || sum_cols < grad_wav_size) { kp->size = -1; return 0; }
    float descriptor_dir = 360.f - 90.f;
    if (!upright) {
        SurfHF dx_t[2], dy_t[2];
        float X[ORI_MAX_SAMPLES], Y[ORI_MAX_SAMPLES], angle[ORI_MAX_SAMPLES];
        resize_haar_pattern(dx_s, dx_t, 2, 4, grad_wav_size, sum_cols);
        resize_haar_pattern(dy_s, dy_t, 2, 4, grad_wav_size, sum_cols);
        int nangle = 0;
        for (int kk = 0; kk < ot->n; kk++) {
            int x = orc_cvRoundf(kp->x + ot->x[kk] * s - (float)(grad_wav_size - 1) / 2);
            int y = orc_cvRoundf(kp->y + ot->y[kk] * s - (float)(grad_wav_size - 1) / 2);
            if (y < 0 || y >= sum_rows - grad_wav_size || x < 0 || x >= sum_cols - grad_wav_size) continue;
            const int32_t* ptr = sum + (size_t)y * sum_cols + x;
            float vx = calc_haar_pattern(ptr, dx_t, 2), vy = calc_haar_pattern(ptr, dy_t, 2);
            X[nangle] = vx * ot->w[kk]; Y[nangle] = vy * ot->w[kk];
            nangle++;
        }
        if (nangle == 0) { kp->size = -1; return 0; }        /* no gradient could be sampled: the keypoint is dropped */
        for (int k = 0; k < nangle; k++) angle[k] = orc_fast_atan2(Y[k], X[k]);      /* cv::phase(X, Y, angle, true) */
        float bestx = 0, besty = 0, descriptor_mod = 0;
        for (int i = 0; i < 360; i += SURF_ORI_SEARCH_INC) {
            float sumx = 0, sumy = 0, temp_mod;
            for (int j = 0; j < nangle; j++) {
                int d = abs(orc_cvRoundf(angle[j]) - i);
                if (d < ORI_WIN / 2 || d > 360 - ORI_WIN / 2) { sumx += X[j]; sumy += Y[j]; }
            }
            temp_mod = sumx * sumx + sumy * sumy;
            if (temp_mod > descriptor_mod) { descriptor_mod = temp_mod; bestx = sumx; besty = sumy; }
        }
        descriptor_dir = orc_fast_atan2(-besty, bestx);
    }
    kp->angle = descriptor_dir;
    int win_size = (int)((PATCH_SZ + 1) * s);
    if (!upright) {
        descriptor_dir *= (float)(3.14159265358979323846 / 180);
        double sd, cd;
        orc_sincos((double)descriptor_dir, &sd, &cd);
        float sin_dir = -(float)sd, cos_dir = (float)cd;
        float win_offset = -(float)(win_size - 1) / 2;
        float start_x = kp->x + win_offset * cos_dir + win_offset * sin_dir;
        float start_y = kp->y - win_offset * sin_dir + win_offset * cos_dir;
        const int ncols1 = w - 1, nrows1 = h - 1;
        for (int i = 0; i < win_size; i++, start_x += sin_dir, start_y += cos_dir) {
            double pixel_x = start_x, pixel_y = start_y;
            for (int j = 0; j < win_size; j++, pixel_x += cos_dir, pixel_y -= sin_dir) {
                int ix = orc_cvFloor(pixel_x), iy = orc_cvFloor(pixel_y);
                if ((unsigned)ix < (unsigned)ncols1 && (unsigned)iy < (unsigned)nrows1) {
                    float a = (float)(pixel_x - ix), b = (float)(pixel_y - iy);
                    const uint8_t* ip = img + (size_t)iy * stride + ix;
                    winbuf[i * win_size + j] = (uint8_t)orc_cvRoundf(ip[0] * (1.f - a) * (1.f - b) + ip[1] * a * (1.f - b) + ip[stride] * (1.f - a) * b + ip[stride + 1] * a * b);
                } else {
                    int x = orc_cvRound(pixel_x), y = orc_cvRound(pixel_y);
                    x = x > 0 ? x : 0; x = x < ncols1 ? x : ncols1;
                    y = y > 0 ? y : 0; y = y < nrows1 ? y : nrows1;
                    winbuf[i * win_size + j] = img[(size_t)y * stride + x];
                }
            }
        }
    } else {
        float win_offset = -(float)(win_size - 1) / 2;
        int start_x = orc_cvRoundf(kp->x + win_offset);
        int start_y = orc_cvRoundf(kp->y - win_offset);
        for (int i = 0; i < win_size; i++, start_x++) {
            int pixel_x = start_x, pixel_y = start_y;
            for (int j = 0; j < win_size; j++, pixel_y--) {
                int x = pixel_x > 0 ? pixel_x : 0;
                int y = pixel_y > 0 ? pixel_y : 0;
                x = x < w - 1 ? x : w - 1;
                y = y < h - 1 ? y : h - 1;
                winbuf[i * win_size + j] = img[(size_t)y * stride + x];
            }
        }
    }
    orc_resize_area_u8(winbuf, win_size, win_size, &PATCH[0][0], PATCH_SZ + 1, PATCH_SZ + 1);
    for (int i = 0; i < PATCH_SZ; i++)
        for (int j = 0; j < PATCH_SZ; j++) {
            float dw = DW[i * PATCH_SZ + j];
            float vx = (PATCH[i][j+1] - PATCH[i][j] + PATCH[i+1][j+1] - PATCH[i+1][j]) * dw;
            float vy = (PATCH[i+1][j] - PATCH[i][j] + PATCH[i+1][j+1] - PATCH[i][j+1]) * dw;
            DX[i][j] = vx; DY[i][j] = vy;
        }
    for (int kk = 0; kk < dsize; kk++) vec[kk] = 0;
    double square_mag = 0;
    float* v = vec;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            for (int y = i*5; y < i*5+5; y++)
                for (int x = j*5; x < j*5+5; x++) {
                    float tx = DX[y][x], ty = DY[y][x];
                    if (!extended) {
                        v[0] += tx; v[1] += ty;
                        v[2] += (float)fabs(tx); v[3] += (float)fabs(ty);
                    } else {
                        if (ty >= 0) { v[0] += tx; v[1] += (float)fabs(tx); } else { v[2] += tx; v[3] += (float)fabs(tx); }
                        if (tx >= 0) { v[4] += ty; v[5] += (float)fabs(ty); } else { v[6] += ty; v[7] += (float)fabs(ty); }
                    }
                }
            const int per = extended ? 8 : 4;
            for (int kk = 0; kk < per; kk++) square_mag += v[kk] * v[kk];
            v += per;
        }
    float scale = (float)(1. / (sqrt(square_mag) + FLT_EPSILON));
    for (int kk = 0; kk < dsize; kk++) vec[kk] *= scale;
    return 1;
}

/* [UPSTREAM] surf.cpp SURF_Impl::detectAndCompute + fastHessianDetector + SURFFindInvoker */
int orc_surf_detect_and_compute(const uint8_t* img, int w, int h, int stride, const orc_surf_params* p,
                                orc_keypoint* kps_out, float* desc_out, int cap)
{
    int nOctaves = p->nOctaves, nOctaveLayers = p->nOctaveLayers;
    float hessianThreshold = (float)p->hessianThreshold;
    int nTotalLayers = (nOctaveLayers + 2) * nOctaves;
    int32_t* sum = (int32_t*)malloc(sizeof(int32_t) * (size_t)(w + 1) * (h + 1));
    orc_integral_u8(img, w, h, stride, sum);

    float** dets = (float**)calloc(nTotalLayers, sizeof(float*));
    float** traces = (float**)calloc(nTotalLayers, sizeof(float*));
    int* sizes = (int*)malloc(sizeof(int) * nTotalLayers);
    int* sampleSteps = (int*)malloc(sizeof(int) * nTotalLayers);
    int index = 0, step = 1;
    for (int octave = 0; octave < nOctaves; octave++) {
        for (int layer = 0; layer < nOctaveLayers + 2; layer++) {
            size_t n = (size_t)(h / step) * (w / step);
            dets[index] = (float*)calloc(n ? n : 1, sizeof(float));     /* OpenCV leaves these uninitialised; never read outside the written region */
            traces[index] = (float*)calloc(n ? n : 1, sizeof(float));
            sizes[index] = (SURF_HAAR_SIZE0 + SURF_HAAR_SIZE_INC * layer) << octave;
            sampleSteps[index] = step;
            index++;
        }
        step *= 2;
    }
    for (int i = 0; i < nTotalLayers; i++)
        orc_surf_layer(sum, w, h, sizes[i], sampleSteps[i], dets[i], traces[i]);

    int kcap = 1024, nk = 0;
    orc_keypoint* kps = (orc_keypoint*)malloc(sizeof(orc_keypoint) * kcap);
    for (int octave = 0; octave < nOctaves; octave++)
        for (int l = 1; l <= nOctaveLayers; l++) {
            int layer = octave * (nOctaveLayers + 2) + l;
            int size = sizes[layer], sampleStep = sampleSteps[layer];
            int layer_rows = h / sampleStep, layer_cols = w / sampleStep;
            int margin = (sizes[layer + 1] / 2) / sampleStep + 1;
            int stp = layer_cols;
            for (int i = margin; i < layer_rows - margin; i++) {
                const float* det_ptr = dets[layer] + (size_t)i * stp;
                const float* trace_ptr = traces[layer] + (size_t)i * stp;
                for (int j = margin; j < layer_cols - margin; j++) {
                    float val0 = det_ptr[j];
                    if (val0 > hessianThreshold) {
                        int sum_i = sampleStep * (i - (size / 2) / sampleStep);
                        int sum_j = sampleStep * (j - (size / 2) / sampleStep);
                        const float* det1 = dets[layer - 1] + (size_t)i * stp + j;
                        const float* det2 = dets[layer] + (size_t)i * stp + j;
                        const float* det3 = dets[layer + 1] + (size_t)i * stp + j;
                        float N9[3][9] = {
                            { det1[-stp-1], det1[-stp], det1[-stp+1], det1[-1], det1[0], det1[1], det1[stp-1], det1[stp], det1[stp+1] },
                            { det2[-stp-1], det2[-stp], det2[-stp+1], det2[-1], det2[0], det2[1], det2[stp-1], det2[stp], det2[stp+1] },
                            { det3[-stp-1], det3[-stp], det3[-stp+1], det3[-1], det3[0], det3[1], det3[stp-1], det3[stp], det3[stp+1] } };
                        int is_max = 1;
                        for (int a = 0; a < 3 && is_max; a++)
                            for (int b = 0; b < 9; b++) {
                                if (a == 1 && b == 4) continue;
                                if (!(val0 > N9[a][b])) { is_max = 0; break; }
                            }
                        if (is_max) {
                            float center_i = sum_i + (size - 1) * 0.5f;
                            float center_j = sum_j + (size - 1) * 0.5f;
                            orc_keypoint kpt;
                            kpt.x = center_j; kpt.y = center_i; kpt.size = (float)sizes[layer];
                            kpt.angle = -1; kpt.response = val0; kpt.octave = octave;
                            kpt.class_id = (trace_ptr[j] > 0) - (trace_ptr[j] < 0);
                            int ds = size - sizes[layer - 1];
                            if (interpolate_keypoint(N9, sampleStep, sampleStep, ds, &kpt)) {
                                if (nk == kcap) { kcap *= 2; kps = (orc_keypoint*)realloc(kps, sizeof(orc_keypoint) * kcap); }
                                kps[nk++] = kpt;
                            }
                        }
                    }
                }
            }
        }
    qsort(kps, nk, sizeof(orc_keypoint), kp_cmp);

    for (int i = 0; i < nTotalLayers; i++) { free(dets[i]); free(traces[i]); }
    free(dets); free(traces); free(sizes); free(sampleSteps); free(sum);

    /* descriptors + removal of keypoints marked for deletion */
    float G[PATCH_SZ], DW[PATCH_SZ * PATCH_SZ];
    orc_gaussian_kernel_f32(PATCH_SZ, SURF_DESC_SIGMA, G);
    for (int i = 0; i < PATCH_SZ; i++) for (int j = 0; j < PATCH_SZ; j++) DW[i * PATCH_SZ + j] = G[i] * G[j];
    float maxSize = 0;
    for (int k = 0; k < nk; k++) if (kps[k].size > maxSize) maxSize = kps[k].size;
    int imaxSize = orc_cvCeil((PATCH_SZ + 1) * maxSize * 1.2f / 9.0f); if (imaxSize < 1) imaxSize = 1;
    uint8_t* winbuf = (uint8_t*)malloc((size_t)imaxSize * imaxSize);
    int j = 0, overflow = 0;
    float vec[128];
    const int dsize = p->extended ? 128 : 64;
    OriTab ot;
    make_ori_tab(&ot);
    int32_t* isum = NULL;
    if (!p->upright) { isum = (int32_t*)malloc(sizeof(int32_t) * (size_t)(w + 1) * (h + 1)); orc_integral_u8(img, w, h, stride, isum); }
    for (int k = 0; k < nk; k++) {
        if (surf_describe(img, isum, w, h, stride, DW, &ot, p->upright, p->extended, &kps[k], vec, winbuf)) {
            if (j < cap) { kps_out[j] = kps[k]; if (desc_out) memcpy(desc_out + (size_t)j * dsize, vec, sizeof(float) * dsize); }
            else overflow = 1;
            j++;
        }
    }
    free(isum);
    free(winbuf); free(kps);
    return overflow ? -j : j;
}
