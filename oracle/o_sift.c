/* o_sift.c -- TEST INFRASTRUCTURE (CPU oracle; never linked into the product).
 *
 * The SIFT branch of detect_features (uvo_libraries/src/VO_utility.cpp:107-112):
 *     Ptr<SIFT> detector = SIFT::create(10000, 3, 0.03, 10, 1.6);  detector->detectAndCompute(img, noArray(), keypoints, descriptors);
 * restated from memory of OpenCV 4.5 features2d (sift.dispatch.cpp: createInitialImage, buildGaussianPyramid, buildDoGPyramid,
 * findScaleSpaceExtrema, KeyPointsFilter; sift.simd.hpp: adjustLocalExtrema, calcOrientationHist, calcSIFTDescriptor -- their SCALAR
 * paths, float `sift_wt`, SIFT_FIXPT_SCALE = 1) and of the imgproc / core routines they call (resize INTER_LINEAR of CV_32F,
 * GaussianBlur with the symmetric separable filters and BORDER_REFLECT_101, getGaussianKernel, hal::exp32f's table + polynomial,
 * hal::fastAtan2, hal::magnitude32f).
 *
 * PARITY UNPINNED, confidence MEDIUM TO LOW: OpenCV is absent here, the reference holds no SIFT vectors, and several of the routines
 * above have SIMD paths that round differently from their scalar ones (FMA in the separable filters and in magnitude, float
 * instead of double polynomial in exp32f).  What this file pins is the HIP implementation to one fixed operation order.  Stated
 * departures, shared with the HIP path so that host and device agree bit for bit: cosf / sinf / powf(2, x) are the deterministic
 * double series of o_core.c rounded to float; when more than nfeatures keypoints survive, the ones whose response is at least the
 * nfeatures-th largest are kept IN SORTED ORDER (KeyPointsFilter::retainBest leaves an nth_element-defined order). */
#include "uvo_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SIFT_DESCR_WIDTH 4
#define SIFT_DESCR_HIST_BINS 8
#define SIFT_INIT_SIGMA 0.5f
#define SIFT_IMG_BORDER 5
#define SIFT_MAX_INTERP_STEPS 5
#define SIFT_ORI_HIST_BINS 36
#define SIFT_ORI_SIG_FCTR 1.5f
#define SIFT_ORI_RADIUS 4.5f                 /* 3 * SIFT_ORI_SIG_FCTR */
#define SIFT_ORI_PEAK_RATIO 0.8f
#define SIFT_DESCR_SCL_FCTR 3.f
#define SIFT_DESCR_MAG_THR 0.2f
#define SIFT_INT_DESCR_FCTR 512.f

typedef struct { int w, h; float* d; } img_t;
float orc_fast_atan2(float y, float x);      /* o_surf.c: cv::fastAtan2 (degrees) */

/* ---- hal::exp32f, scalar path: 2^(x log2 e) with a 64-entry table and a degree-4 polynomial evaluated in double ---- */
#define EXPTAB_SCALE 6
#define EXPTAB_MASK ((1 << EXPTAB_SCALE) - 1)
#define EXPPOLY_32F_A0 .9670371139572337719125840413672004409288e-2
static float g_exptab[1 << EXPTAB_SCALE];
static int g_exptab_ready = 0;
static void exptab_init(void)
{
    if (g_exptab_ready) return;
    for (int i = 0; i < (1 << EXPTAB_SCALE); i++) g_exptab[i] = (float)(pow(2.0, (double)i / (1 << EXPTAB_SCALE)) * EXPPOLY_32F_A0);
    g_exptab_ready = 1;
}
const float* orc_sift_exptab(void) { exptab_init(); return g_exptab; }
float orc_exp32f(float x)
{
    static const double exp_prescale = 1.4426950408889634073599246810019 * (1 << EXPTAB_SCALE);
    static const double exp_postscale = 1. / (1 << EXPTAB_SCALE);
    static const double exp_max_val = 3000. * (1 << EXPTAB_SCALE);
    const float A4 = (float)(1.000000000000002438532970795181890933776 / EXPPOLY_32F_A0), A3 = (float)(.6931471805521448196800669615864773144641 / EXPPOLY_32F_A0),
                A2 = (float)(.2402265109513301490103372422686535526573 / EXPPOLY_32F_A0), A1 = (float)(.5550339366753125211915322047004666939128e-1 / EXPPOLY_32F_A0);
    exptab_init();
    double x0 = (double)x * exp_prescale;
    if (x0 < -exp_max_val) x0 = -exp_max_val;
    if (x0 > exp_max_val) x0 = exp_max_val;
    const int val0 = orc_cvRound(x0);
    int t = (val0 >> EXPTAB_SCALE) + 127;
    t = !(t & ~255) ? t : (t < 0 ? 0 : 255);
    union { int i; float f; } buf;
    buf.i = t << 23;
    x0 = (x0 - val0) * exp_postscale;
    return (float)((double)buf.f * (double)g_exptab[val0 & EXPTAB_MASK] * ((((x0 + A1) * x0 + A2) * x0 + A3) * x0 + A4));
}
/* 2^x for the keypoint size (powf(2.f, ...) in the reference): e^(x ln 2) by its Taylor series in double on the fractional part */
float orc_exp2f_det(float x)
{
    const double xd = (double)x, fl = floor(xd), fr = (xd - fl) * 0.69314718055994530942;
    double term = 1, sum = 1;
    for (int k = 1; k <= 24; k++) { term = term * fr / k; sum += term; }
    return (float)ldexp(sum, (int)fl);
}

/* ---- getGaussianKernel(n, sigma, CV_32F): exp in double, normalised in double, cast (as the SURF descriptor weights in o_surf.c) ---- */
int orc_sift_gauss_kernel(double sigma, float* k /* >= 64 */)
{
    int n = orc_cvRound(sigma * 4 * 2 + 1) | 1;              /* GaussianBlur, CV_32F: 4 sigma either side */
    if (n > 63) n = 63;
    const double scale2X = -0.5 / (sigma * sigma);
    double t[64], sum = 0;
    for (int i = 0; i < n; i++) { const double x = i - (n - 1) * 0.5; t[i] = exp(scale2X * x * x); sum += t[i]; }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] = (float)(t[i] * sum);
    return n;
}
static int reflect101(int p, int n) { if (n == 1) return 0; while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; } return p; }

/* GaussianBlur(src, dst, Size(), sigma) on floats = sepFilter2D through the filter engine (filter.dispatch.cpp, getLinearRowFilter /
 * getLinearColumnFilter): kernels wider than 5 get the GENERIC RowFilter<float, float> -- s = k[0] x[-r]; s += k[t] x[-r+t], left to
 * right, no pairing -- and, the kernel being symmetrical, SymmColumnFilter -- s = k0 x0 (+ delta = 0); s += ki (x[+i] + x[-i]).
 * (SIFT's sigmas give 11 .. 27 taps; SymmRowSmallFilter only serves ksize <= 5.)  BORDER_REFLECT_101. */
static void gaussian_blur(const img_t* src, img_t* dst, double sigma)
{
    float k[64];
    const int n = orc_sift_gauss_kernel(sigma, k), r = n / 2, w = src->w, h = src->h;
    float* tmp = (float*)malloc(sizeof(float) * (size_t)w * h);
    for (int y = 0; y < h; y++) {
        const float* s = src->d + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            float acc = k[0] * s[reflect101(x - r, w)];
            for (int t = 1; t < n; t++) acc += k[t] * s[reflect101(x - r + t, w)];
            tmp[(size_t)y * w + x] = acc;
        }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float acc = k[r] * tmp[(size_t)y * w + x];
            for (int i = 1; i <= r; i++) acc += k[r + i] * (tmp[(size_t)reflect101(y + i, h) * w + x] + tmp[(size_t)reflect101(y - i, h) * w + x]);
            dst->d[(size_t)y * w + x] = acc;
        }
    free(tmp);
}

/* resize(src, dst, 2w x 2h, INTER_LINEAR) of CV_32F: horizontal then vertical two-tap interpolation, coefficients in float */
static void lin_coef(int d, int ssize, int* s0, float* a0, float* a1)
{
    float f = (float)((d + 0.5) * 0.5 - 0.5);
    int s = orc_cvFloor(f);
    f -= s;
    if (s < 0) { f = 0; s = 0; }
    if (s + 1 >= ssize) { f = 0; s = ssize - 1; }
    *s0 = s; *a0 = 1.f - f; *a1 = f;
}
static void resize2x_linear(const uint8_t* img, int w, int h, int stride, img_t* dst)
{
    const int W = 2 * w, H = 2 * h;
    float* hrow = (float*)malloc(sizeof(float) * (size_t)W * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < W; x++) {
            int s; float a0, a1;
            lin_coef(x, w, &s, &a0, &a1);
            const int s1 = s + 1 < w ? s + 1 : s;
            hrow[(size_t)y * W + x] = (float)img[(size_t)y * stride + s] * a0 + (float)img[(size_t)y * stride + s1] * a1;
        }
    for (int y = 0; y < H; y++) {
        int s; float b0, b1;
        lin_coef(y, h, &s, &b0, &b1);
        const int s1 = s + 1 < h ? s + 1 : s;
        for (int x = 0; x < W; x++) dst->d[(size_t)y * W + x] = hrow[(size_t)s * W + x] * b0 + hrow[(size_t)s1 * W + x] * b1;
    }
    free(hrow);
}

typedef struct {
    int nOctaves, nLayers, gw[16], gh[16];
    img_t* gauss;      /* nOctaves * (nLayers + 3) */
    img_t* dog;        /* nOctaves * (nLayers + 2) */
} pyr_t;

static img_t img_new(int w, int h) { img_t m; m.w = w; m.h = h; m.d = (float*)malloc(sizeof(float) * (size_t)w * h); return m; }

/* sigma of the blur that takes layer i-1 to layer i (buildGaussianPyramid) */
void orc_sift_layer_sigmas(int nLayers, double sigma, double* sig /* nLayers + 3 */)
{
    sig[0] = sigma;
    const double k = pow(2., 1. / nLayers);
    for (int i = 1; i < nLayers + 3; i++) {
        const double sig_prev = pow(k, (double)(i - 1)) * sigma, sig_total = sig_prev * k;
        sig[i] = sqrt(sig_total * sig_total - sig_prev * sig_prev);
    }
}
int orc_sift_num_octaves(int w, int h) { const int m = 2 * (w < h ? w : h); return orc_cvRound(log((double)m) / log(2.) - 2) + 1; }     /* firstOctave = -1 */

static void build_pyramid(const uint8_t* img, int w, int h, int stride, int nLayers, double sigma, pyr_t* p)
{
    p->nLayers = nLayers;
    p->nOctaves = orc_sift_num_octaves(w, h);
    if (p->nOctaves > 16) p->nOctaves = 16;
    if (p->nOctaves < 1) p->nOctaves = 1;
    p->gauss = (img_t*)calloc((size_t)p->nOctaves * (nLayers + 3), sizeof(img_t));
    p->dog = (img_t*)calloc((size_t)p->nOctaves * (nLayers + 2), sizeof(img_t));
    double sig[16];
    orc_sift_layer_sigmas(nLayers, sigma, sig);
    /* createInitialImage: u8 -> float, doubled with INTER_LINEAR, blurred to sigma */
    img_t dbl = img_new(2 * w, 2 * h);
    resize2x_linear(img, w, h, stride, &dbl);
    const float sd2 = (float)sigma * (float)sigma - SIFT_INIT_SIGMA * SIFT_INIT_SIGMA * 4;
    const float sig_diff = sqrtf(sd2 > 0.01f ? sd2 : 0.01f);
    for (int o = 0; o < p->nOctaves; o++) {
        for (int i = 0; i < nLayers + 3; i++) {
            img_t* dst = &p->gauss[o * (nLayers + 3) + i];
            if (o == 0 && i == 0) { *dst = img_new(dbl.w, dbl.h); gaussian_blur(&dbl, dst, (double)sig_diff); }
            else if (i == 0) {
                const img_t* src = &p->gauss[(o - 1) * (nLayers + 3) + nLayers];
                *dst = img_new(src->w / 2, src->h / 2);
                for (int y = 0; y < dst->h; y++) for (int x = 0; x < dst->w; x++) dst->d[(size_t)y * dst->w + x] = src->d[(size_t)(2 * y) * src->w + 2 * x];    /* INTER_NEAREST */
            } else {
                const img_t* src = &p->gauss[o * (nLayers + 3) + i - 1];
                *dst = img_new(src->w, src->h);
                gaussian_blur(src, dst, sig[i]);
            }
        }
        p->gw[o] = p->gauss[o * (nLayers + 3)].w; p->gh[o] = p->gauss[o * (nLayers + 3)].h;
        for (int i = 0; i < nLayers + 2; i++) {
            const img_t* a = &p->gauss[o * (nLayers + 3) + i]; const img_t* b = a + 1;
            img_t* d = &p->dog[o * (nLayers + 2) + i];
            *d = img_new(a->w, a->h);
            for (size_t e = 0; e < (size_t)a->w * a->h; e++) d->d[e] = b->d[e] - a->d[e];
        }
    }
    free(dbl.d);
}
static void free_pyramid(pyr_t* p)
{
    for (int i = 0; i < p->nOctaves * (p->nLayers + 3); i++) free(p->gauss[i].d);
    for (int i = 0; i < p->nOctaves * (p->nLayers + 2); i++) free(p->dog[i].d);
    free(p->gauss); free(p->dog);
}

#define AT(m, r, c) ((m)->d[(size_t)(r) * (m)->w + (c)])

static int solve3f(const float a[3][3], const float b[3], float x[3])      /* Matx33f::solve(b, DECOMP_LU): Cramer in float (as o_surf.c) */
{
    float d = (float)(double)(a[0][0]*(a[1][1]*a[2][2] - a[2][1]*a[1][2]) - a[0][1]*(a[1][0]*a[2][2] - a[2][0]*a[1][2]) + a[0][2]*(a[1][0]*a[2][1] - a[2][0]*a[1][1]));
    if (d == 0) { x[0] = x[1] = x[2] = 0; return 0; }
    d = 1/d;
    x[0] = d*(b[0]*(a[1][1]*a[2][2] - a[1][2]*a[2][1]) - a[0][1]*(b[1]*a[2][2] - a[1][2]*b[2]) + a[0][2]*(b[1]*a[2][1] - a[1][1]*b[2]));
    x[1] = d*(a[0][0]*(b[1]*a[2][2] - a[1][2]*b[2]) - b[0]*(a[1][0]*a[2][2] - a[1][2]*a[2][0]) + a[0][2]*(a[1][0]*b[2] - b[1]*a[2][0]));
    x[2] = d*(a[0][0]*(a[1][1]*b[2] - b[1]*a[2][1]) - a[0][1]*(a[1][0]*b[2] - b[1]*a[2][0]) + b[0]*(a[1][0]*a[2][1] - a[1][1]*a[2][0]));
    return 1;
}

/* adjustLocalExtrema: quadratic refinement (up to five steps), contrast and edge tests */
static int adjust_local_extrema(const pyr_t* p, orc_keypoint* kpt, int octv, int* player, int* pr, int* pc, float contrastThreshold, float edgeThreshold, float sigma)
{
    const int nL = p->nLayers;
    const float img_scale = 1.f / 255, deriv_scale = img_scale * 0.5f, second_deriv_scale = img_scale, cross_deriv_scale = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0, contr = 0;
    int i = 0, layer = *player, r = *pr, c = *pc;
    for (; i < SIFT_MAX_INTERP_STEPS; i++) {
        const img_t* img = &p->dog[octv * (nL + 2) + layer]; const img_t* prev = img - 1; const img_t* next = img + 1;
        const float dD[3] = { (AT(img, r, c + 1) - AT(img, r, c - 1)) * deriv_scale, (AT(img, r + 1, c) - AT(img, r - 1, c)) * deriv_scale,
                              (AT(next, r, c) - AT(prev, r, c)) * deriv_scale };
        const float v2 = AT(img, r, c) * 2;
        const float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * second_deriv_scale;
        const float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * second_deriv_scale;
        const float dss = (AT(next, r, c) + AT(prev, r, c) - v2) * second_deriv_scale;
        const float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cross_deriv_scale;
        const float dxs = (AT(next, r, c + 1) - AT(next, r, c - 1) - AT(prev, r, c + 1) + AT(prev, r, c - 1)) * cross_deriv_scale;
        const float dys = (AT(next, r + 1, c) - AT(next, r - 1, c) - AT(prev, r + 1, c) + AT(prev, r - 1, c)) * cross_deriv_scale;
        const float H[3][3] = { { dxx, dxy, dxs }, { dxy, dyy, dys }, { dxs, dys, dss } };
        float X[3];
        solve3f(H, dD, X);
        xi = -X[2]; xr = -X[1]; xc = -X[0];
        if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
        if (fabsf(xi) > (float)(2147483647 / 3) || fabsf(xr) > (float)(2147483647 / 3) || fabsf(xc) > (float)(2147483647 / 3)) return 0;
        c += orc_cvRoundf(xc); r += orc_cvRoundf(xr); layer += orc_cvRoundf(xi);
        if (layer < 1 || layer > nL || c < SIFT_IMG_BORDER || c >= img->w - SIFT_IMG_BORDER || r < SIFT_IMG_BORDER || r >= img->h - SIFT_IMG_BORDER) return 0;
    }
    if (i >= SIFT_MAX_INTERP_STEPS) return 0;
    {
        const img_t* img = &p->dog[octv * (nL + 2) + layer]; const img_t* prev = img - 1; const img_t* next = img + 1;
        const float dD[3] = { (AT(img, r, c + 1) - AT(img, r, c - 1)) * deriv_scale, (AT(img, r + 1, c) - AT(img, r - 1, c)) * deriv_scale,
                              (AT(next, r, c) - AT(prev, r, c)) * deriv_scale };
        const float t = dD[0] * xc + dD[1] * xr + dD[2] * xi;             /* Matx31f::dot: accumulated in float, in this order */
        contr = AT(img, r, c) * img_scale + t * 0.5f;
        if (fabsf(contr) * nL < contrastThreshold) return 0;
        const float v2 = AT(img, r, c) * 2.f;
        const float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * second_deriv_scale;
        const float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * second_deriv_scale;
        const float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cross_deriv_scale;
        const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        if (det <= 0 || tr * tr * edgeThreshold >= (edgeThreshold + 1) * (edgeThreshold + 1) * det) return 0;
    }
    kpt->x = (c + xc) * (1 << octv);
    kpt->y = (r + xr) * (1 << octv);
    kpt->octave = octv + (layer << 8) + (orc_cvRound((xi + 0.5) * 255) << 16);
    kpt->size = sigma * orc_exp2f_det((layer + xi) / nL) * (1 << octv) * 2;
    kpt->response = fabsf(contr);
    kpt->class_id = -1;
    kpt->angle = -1;
    *player = layer; *pr = r; *pc = c;
    return 1;
}

/* calcOrientationHist: 36-bin gradient histogram around (px, py) of a Gaussian layer, smoothed; returns its maximum */
static float calc_orientation_hist(const img_t* img, int px, int py, int radius, float sigma, float* hist, int n)
{
    const float expf_scale = -1.f / (2.f * sigma * sigma);
    float temphist[SIFT_ORI_HIST_BINS + 4];
    float* th = temphist + 2;
    for (int i = 0; i < n; i++) th[i] = 0.f;
    for (int i = -radius; i <= radius; i++) {
        const int y = py + i;
        if (y <= 0 || y >= img->h - 1) continue;
        for (int j = -radius; j <= radius; j++) {
            const int x = px + j;
            if (x <= 0 || x >= img->w - 1) continue;
            const float dx = AT(img, y, x + 1) - AT(img, y, x - 1), dy = AT(img, y - 1, x) - AT(img, y + 1, x);
            const float wgt = orc_exp32f((i * i + j * j) * expf_scale);
            const float ori = orc_fast_atan2(dy, dx), mag = sqrtf(dx * dx + dy * dy);
            int bin = orc_cvRoundf((n / 360.f) * ori);
            if (bin >= n) bin -= n;
            if (bin < 0) bin += n;
            th[bin] += wgt * mag;
        }
    }
    th[-1] = th[n - 1]; th[-2] = th[n - 2]; th[n] = th[0]; th[n + 1] = th[1];
    for (int i = 0; i < n; i++)
        hist[i] = (th[i - 2] + th[i + 2]) * (1.f / 16.f) + (th[i - 1] + th[i + 1]) * (4.f / 16.f) + th[i] * (6.f / 16.f);
    float maxval = hist[0];
    for (int i = 1; i < n; i++) maxval = maxval > hist[i] ? maxval : hist[i];
    return maxval;
}

static int kp_less(const void* a_, const void* b_)          /* KeyPoint_LessThan (keypoint.cpp) */
{
    const orc_keypoint* a = (const orc_keypoint*)a_; const orc_keypoint* b = (const orc_keypoint*)b_;
    if (a->x != b->x) return a->x < b->x ? -1 : 1;
    if (a->y != b->y) return a->y < b->y ? -1 : 1;
    if (a->size != b->size) return a->size > b->size ? -1 : 1;
    if (a->angle != b->angle) return a->angle < b->angle ? -1 : 1;
    if (a->response != b->response) return a->response > b->response ? -1 : 1;
    if (a->octave != b->octave) return a->octave > b->octave ? -1 : 1;
    if (a->class_id != b->class_id) return a->class_id > b->class_id ? -1 : 1;
    return 0;
}
static int float_desc(const void* a, const void* b) { const float x = *(const float*)a, y = *(const float*)b; return x > y ? -1 : (x < y ? 1 : 0); }

/* removeDuplicatedSorted + retainBest + the scaling back of firstOctave = -1, on an unordered candidate list (in place); returns the count */
int orc_sift_finish_keypoints(orc_keypoint* k, int n, int nfeatures)
{
    if (n == 0) return 0;
    qsort(k, (size_t)n, sizeof(*k), kp_less);
    int m = 1;
    for (int i = 1; i < n; i++) {
        const orc_keypoint* a = &k[m - 1]; const orc_keypoint* b = &k[i];
        if (a->x != b->x || a->y != b->y || a->size != b->size || a->angle != b->angle) k[m++] = k[i];
    }
    n = m;
    if (nfeatures > 0 && n > nfeatures) {
        float* resp = (float*)malloc(sizeof(float) * (size_t)n);
        for (int i = 0; i < n; i++) resp[i] = k[i].response;
        qsort(resp, (size_t)n, sizeof(float), float_desc);
        const float amb = resp[nfeatures - 1];
        free(resp);
        m = 0;
        for (int i = 0; i < n; i++) if (k[i].response >= amb) k[m++] = k[i];
        n = m;
    }
    for (int i = 0; i < n; i++) {
        k[i].octave = (k[i].octave & ~255) | ((k[i].octave + -1) & 255);
        k[i].x *= 0.5f; k[i].y *= 0.5f; k[i].size *= 0.5f;
    }
    return n;
}

/* calcSIFTDescriptor: 4 x 4 x 8 histogram of a rotated, Gaussian-weighted window, trilinear votes in sample order */
static void calc_descriptor(const img_t* img, float ptx, float pty, float ori, float scl, float* dst)
{
    const int d = SIFT_DESCR_WIDTH, n = SIFT_DESCR_HIST_BINS;
    const int px = orc_cvRoundf(ptx), py = orc_cvRoundf(pty);
    double sd, cd;
    orc_sincos((double)(ori * (float)(3.14159265358979323846 / 180)), &sd, &cd);
    float cos_t = (float)cd, sin_t = (float)sd;
    const float bins_per_rad = n / 360.f, exp_scale = -1.f / (d * d * 0.5f), hist_width = SIFT_DESCR_SCL_FCTR * scl;
    int radius = orc_cvRoundf(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
    const int rmax = (int)sqrt(((double)img->w) * img->w + ((double)img->h) * img->h);
    radius = radius < rmax ? radius : rmax;
    cos_t /= hist_width; sin_t /= hist_width;
    float hist[(SIFT_DESCR_WIDTH + 2) * (SIFT_DESCR_WIDTH + 2) * (SIFT_DESCR_HIST_BINS + 2)];
    memset(hist, 0, sizeof(hist));
    const int rows = img->h, cols = img->w;
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            const float c_rot = j * cos_t - i * sin_t, r_rot = j * sin_t + i * cos_t;
            float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
            const int r = py + i, c = px + j;
            if (!(rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < rows - 1 && c > 0 && c < cols - 1)) continue;
            const float dx = AT(img, r, c + 1) - AT(img, r, c - 1), dy = AT(img, r - 1, c) - AT(img, r + 1, c);
            const float wgt = orc_exp32f((c_rot * c_rot + r_rot * r_rot) * exp_scale);
            float obin = (orc_fast_atan2(dy, dx) - ori) * bins_per_rad;
            const float mag = sqrtf(dx * dx + dy * dy) * wgt;
            const int r0 = orc_cvFloor(rbin), c0 = orc_cvFloor(cbin);
            int o0 = orc_cvFloor(obin);
            rbin -= r0; cbin -= c0; obin -= o0;
            if (o0 < 0) o0 += n;
            if (o0 >= n) o0 -= n;
            const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
            const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11, v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
            const float v_rco111 = v_rc11 * obin, v_rco110 = v_rc11 - v_rco111, v_rco101 = v_rc10 * obin, v_rco100 = v_rc10 - v_rco101;
            const float v_rco011 = v_rc01 * obin, v_rco010 = v_rc01 - v_rco011, v_rco001 = v_rc00 * obin, v_rco000 = v_rc00 - v_rco001;
            const int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0;
            hist[idx] += v_rco000; hist[idx + 1] += v_rco001;
            hist[idx + (n + 2)] += v_rco010; hist[idx + (n + 3)] += v_rco011;
            hist[idx + (d + 2) * (n + 2)] += v_rco100; hist[idx + (d + 2) * (n + 2) + 1] += v_rco101;
            hist[idx + (d + 3) * (n + 2)] += v_rco110; hist[idx + (d + 3) * (n + 2) + 1] += v_rco111;
        }
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) {
            const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
            hist[idx] += hist[idx + n];
            hist[idx + 1] += hist[idx + n + 1];
            for (int k = 0; k < n; k++) dst[(i * d + j) * n + k] = hist[idx + k];
        }
    const int len = d * d * n;
    float nrm2 = 0;
    for (int k = 0; k < len; k++) nrm2 += dst[k] * dst[k];
    const float thr = sqrtf(nrm2) * SIFT_DESCR_MAG_THR;
    nrm2 = 0;
    for (int k = 0; k < len; k++) { const float val = dst[k] < thr ? dst[k] : thr; dst[k] = val; nrm2 += val * val; }
    const float sq = sqrtf(nrm2);
    nrm2 = SIFT_INT_DESCR_FCTR / (sq > FLT_EPSILON ? sq : FLT_EPSILON);
    for (int k = 0; k < len; k++) {
        int v = orc_cvRoundf(dst[k] * nrm2);                                      /* saturate_cast<uchar> */
        dst[k] = (float)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

/* SIFT::detectAndCompute(img, noArray(), kps, desc).  kps: capacity cap; desc: cap x 128 floats or NULL.  Returns the count,
 * or -(count needed) when cap is too small. */
int orc_sift_detect_and_compute(const uint8_t* img, int w, int h, int stride, int nfeatures, int nOctaveLayers, double contrastThreshold,
                                double edgeThreshold, double sigma, orc_keypoint* kps, float* desc, int cap)
{
    pyr_t p;
    build_pyramid(img, w, h, stride, nOctaveLayers, sigma, &p);
    const int nL = nOctaveLayers, n = SIFT_ORI_HIST_BINS;
    const int threshold = orc_cvFloor(0.5 * contrastThreshold / nL * 255);
    int cnt = 0, capk = 1 << 16;
    orc_keypoint* all = (orc_keypoint*)malloc(sizeof(orc_keypoint) * (size_t)capk);
    for (int o = 0; o < p.nOctaves; o++)
        for (int i = 1; i <= nL; i++) {
            const img_t* img1 = &p.dog[o * (nL + 2) + i]; const img_t* prev = img1 - 1; const img_t* next = img1 + 1;
            const int rows = img1->h, cols = img1->w;
            for (int r = SIFT_IMG_BORDER; r < rows - SIFT_IMG_BORDER; r++)
                for (int c = SIFT_IMG_BORDER; c < cols - SIFT_IMG_BORDER; c++) {
                    const float val = AT(img1, r, c);
                    if (!(fabsf(val) > threshold)) continue;
                    int ext = 1;
                    for (int dr = -1; dr <= 1 && ext; dr++)
                        for (int dc = -1; dc <= 1 && ext; dc++) {
                            const float a = AT(img1, r + dr, c + dc), b = AT(prev, r + dr, c + dc), cc = AT(next, r + dr, c + dc);
                            if (val > 0) ext = val >= a && val >= b && val >= cc; else ext = val <= a && val <= b && val <= cc;
                        }
                    if (!ext) continue;
                    orc_keypoint kpt;
                    int r1 = r, c1 = c, layer = i;
                    if (!adjust_local_extrema(&p, &kpt, o, &layer, &r1, &c1, (float)contrastThreshold, (float)edgeThreshold, (float)sigma)) continue;
                    const float scl_octv = kpt.size * 0.5f / (1 << o);
                    float hist[SIFT_ORI_HIST_BINS];
                    const float omax = calc_orientation_hist(&p.gauss[o * (nL + 3) + layer], c1, r1, orc_cvRoundf(SIFT_ORI_RADIUS * scl_octv),
                                                             SIFT_ORI_SIG_FCTR * scl_octv, hist, n);
                    const float mag_thr = omax * SIFT_ORI_PEAK_RATIO;
                    for (int j = 0; j < n; j++) {
                        const int l = j > 0 ? j - 1 : n - 1, r2 = j < n - 1 ? j + 1 : 0;
                        if (hist[j] > hist[l] && hist[j] > hist[r2] && hist[j] >= mag_thr) {
                            float bin = j + 0.5f * (hist[l] - hist[r2]) / (hist[l] - 2 * hist[j] + hist[r2]);
                            bin = bin < 0 ? n + bin : (bin >= n ? bin - n : bin);
                            kpt.angle = 360.f - (360.f / n) * bin;
                            if (fabsf(kpt.angle - 360.f) < FLT_EPSILON) kpt.angle = 0.f;
                            if (cnt == capk) { capk *= 2; all = (orc_keypoint*)realloc(all, sizeof(orc_keypoint) * (size_t)capk); }
                            all[cnt++] = kpt;
                        }
                    }
                }
        }
    cnt = orc_sift_finish_keypoints(all, cnt, nfeatures);
    if (cnt > cap) { free(all); free_pyramid(&p); return -cnt; }
    for (int k = 0; k < cnt; k++) {
        kps[k] = all[k];
        if (desc) {
            /* calcDescriptors: unpackOctave; the window is taken in the keypoint's own octave and layer */
            int octave = all[k].octave & 255; const int layer = (all[k].octave >> 8) & 255;
            octave = octave < 128 ? octave : (-128 | octave);
            const float scale = octave >= 0 ? 1.f / (1 << octave) : (float)(1 << -octave);
            const float size = all[k].size * scale;
            const img_t* gimg = &p.gauss[(octave + 1) * (nL + 3) + layer];                /* (octave - firstOctave) */
            float angle = 360.f - all[k].angle;
            if (fabsf(angle - 360.f) < FLT_EPSILON) angle = 0.f;
            calc_descriptor(gimg, all[k].x * scale, all[k].y * scale, angle, size * 0.5f, desc + (size_t)k * 128);
        }
    }
    free(all);
    free_pyramid(&p);
    return cnt;
}

/* test hook: Gaussian layer `layer` of octave `o` (floats, row-major); returns its width * height, or 0 */
int orc_sift_gauss_layer(const uint8_t* img, int w, int h, int stride, int nOctaveLayers, double sigma, int o, int layer, float* out, int* ow, int* oh)
{
    pyr_t p;
    build_pyramid(img, w, h, stride, nOctaveLayers, sigma, &p);
    int n = 0;
    if (o >= 0 && o < p.nOctaves && layer >= 0 && layer < nOctaveLayers + 3) {
        const img_t* m = &p.gauss[o * (nOctaveLayers + 3) + layer];
        *ow = m->w; *oh = m->h; n = m->w * m->h;
        if (out) memcpy(out, m->d, sizeof(float) * (size_t)n);
    }
    free_pyramid(&p);
    return n;
}
