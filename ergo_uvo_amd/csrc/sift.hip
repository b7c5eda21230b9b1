// sift.hip -- the SIFT branch of detect_features (uvo_libraries/src/VO_utility.cpp:107-112):
//     Ptr<SIFT> detector = SIFT::create(10000, 3, 0.03, 10, 1.6);  detector->detectAndCompute(img, noArray(), keypoints, descriptors);
// SURVEY.md 8(f) N4 (the shipped parameter files select SURF; this is the next detector of the switch).  OpenCV 4.5 features2d:
// createInitialImage (u8 -> float, doubled with INTER_LINEAR, blurred to sigma), buildGaussianPyramid (nOctaveLayers + 3 blurs per
// octave, INTER_NEAREST halving), buildDoGPyramid, findScaleSpaceExtrema (26-neighbour extrema, adjustLocalExtrema,
// calcOrientationHist), KeyPointsFilter (duplicates, retainBest), calcSIFTDescriptor -- in the operation order of their scalar paths
// (parity vs OpenCV itself is UNPINNED: DESIGN.md section 0; cosf / sinf / powf(2, x) are this library's deterministic double series).
//
// Byte / float streaming work, HBM-bound: the doubled base image, the blurs (separable, one thread per pixel, taps in the
// symmetric filter's order), the differences and the extrema test are coalesced passes over float images that stay in HBM
// (0.5 GB of pyramid at 1080p).  The per-extremum work (refinement, 36-bin orientation histogram) and the per-keypoint
// descriptor (4 x 4 x 8 trilinear histogram, votes added in sample order) are sequential by definition of their float sums: a
// thread per extremum / keypoint.  Sorting, duplicate removal and retainBest (a few thousand 28-byte records) run on the host.
// A standalone operator (uvo_sift_detect): the stereo / mono loops of this library run on SURF.
#include "uvo_ctx.h"
#include "uvo_math.h"
#include <algorithm>
#include <string.h>
#include <math.h>
#include <float.h>
#include <vector>

namespace uvo {

static const int kSiftMaxLayers = 8, kSiftMaxOctaves = 16, kSiftMaxTaps = 64;
static const int SIFT_IMG_BORDER = 5, SIFT_MAX_INTERP_STEPS = 5, SIFT_ORI_HIST_BINS = 36;

struct SiftCand { int o, layer, r, c; };
struct SiftWs {
    int w = 0, h = 0, nL = 0, nOct = 0;
    int ow[kSiftMaxOctaves], oh[kSiftMaxOctaves];
    float* gauss[kSiftMaxOctaves * (kSiftMaxLayers + 3)] = {nullptr};
    float* dog[kSiftMaxOctaves * (kSiftMaxLayers + 2)] = {nullptr};
    float* tmp = nullptr; uint8_t* d_img = nullptr;
    float* d_exptab = nullptr;
    SiftCand* d_cand = nullptr; uvo_keypoint* d_raw = nullptr; uvo_keypoint* d_kps = nullptr; float* d_desc = nullptr;
    int* d_cnt = nullptr;         // [0] candidates, [1] raw keypoints
    int cand_cap = 0, raw_cap = 0, kp_cap = 0;
};
static void sift_ws_release(SiftWs* s)
{
    for (float*& p : s->gauss) { (void)hipFree(p); p = nullptr; }
    for (float*& p : s->dog) { (void)hipFree(p); p = nullptr; }
    (void)hipFree(s->tmp); (void)hipFree(s->d_img); (void)hipFree(s->d_cand); (void)hipFree(s->d_raw); (void)hipFree(s->d_kps); (void)hipFree(s->d_desc);
    (void)hipFree(s->d_cnt); (void)hipFree(s->d_exptab);
    s->tmp = nullptr; s->d_img = nullptr; s->d_cand = nullptr; s->d_raw = nullptr; s->d_kps = nullptr; s->d_desc = nullptr; s->d_cnt = nullptr; s->d_exptab = nullptr;
    s->w = s->h = 0; s->cand_cap = s->raw_cap = s->kp_cap = 0;
}
// the three record lists grow on demand (a 1080p frame has ~13000 keypoints before retainBest, noise images far more per pixel)
static bool sift_grow(SiftWs* s, int cand, int raw, int kp)
{
    if (cand > s->cand_cap) {
        (void)hipFree(s->d_cand); s->d_cand = nullptr; s->cand_cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&s->d_cand), sizeof(SiftCand) * (size_t)cand) != hipSuccess) return false;
        s->cand_cap = cand;
    }
    if (raw > s->raw_cap) {
        (void)hipFree(s->d_raw); s->d_raw = nullptr; s->raw_cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&s->d_raw), sizeof(uvo_keypoint) * (size_t)raw) != hipSuccess) return false;
        s->raw_cap = raw;
    }
    if (kp > s->kp_cap) {
        (void)hipFree(s->d_kps); (void)hipFree(s->d_desc); s->d_kps = nullptr; s->d_desc = nullptr; s->kp_cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&s->d_kps), sizeof(uvo_keypoint) * (size_t)kp) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&s->d_desc), sizeof(float) * 128 * (size_t)kp) != hipSuccess) return false;
        s->kp_cap = kp;
    }
    return true;
}
void sift_ws_free(Ctx* c)
{
    SiftWs* s = static_cast<SiftWs*>(c->sift_ws);
    if (!s) return;
    sift_ws_release(s);
    delete s;
    c->sift_ws = nullptr;
}

// ------------------------------------------------------------------------------------------ device helpers
__device__ __forceinline__ int reflect101(int p, int n) { if (n == 1) return 0; while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; } return p; }

// hal::exp32f, scalar path: table of 2^(i/64) * A0 from the host, polynomial in double
__device__ __forceinline__ float sift_exp32f(float x, const float* __restrict__ tab)
{
    const double exp_prescale = 1.4426950408889634073599246810019 * 64, exp_postscale = 1. / 64, exp_max_val = 3000. * 64;
    const double A0 = .9670371139572337719125840413672004409288e-2;
    const float A4 = (float)(1.000000000000002438532970795181890933776 / A0), A3 = (float)(.6931471805521448196800669615864773144641 / A0),
                A2 = (float)(.2402265109513301490103372422686535526573 / A0), A1 = (float)(.5550339366753125211915322047004666939128e-1 / A0);
    double x0 = (double)x * exp_prescale;
    if (x0 < -exp_max_val) x0 = -exp_max_val;
    if (x0 > exp_max_val) x0 = exp_max_val;
    const int val0 = cv_round_d(x0);
    int t = (val0 >> 6) + 127;
    t = !(t & ~255) ? t : (t < 0 ? 0 : 255);
    const float bf = __int_as_float(t << 23);
    x0 = (x0 - val0) * exp_postscale;
    return (float)((double)bf * (double)tab[val0 & 63] * ((((x0 + A1) * x0 + A2) * x0 + A3) * x0 + A4));
}
__device__ __forceinline__ float sift_exp2f_det(float x)            // 2^x: the Taylor series of e^(frac ln 2) in double
{
    const double xd = (double)x, fl = floor(xd), fr = (xd - fl) * 0.69314718055994530942;
    double term = 1, sum = 1;
    for (int k = 1; k <= 24; k++) { term = term * fr / k; sum += term; }
    return (float)ldexp(sum, (int)fl);
}
__device__ __forceinline__ float sift_atan2_deg(float y, float x)     // cv::fastAtan2 (as surf.hip's fast_atan2_deg)
{
    const float sc = (float)(180 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * sc, p3 = -0.3258083974640975f * sc, p5 = 0.1555786518463281f * sc, p7 = -0.04432655554792128f * sc;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)DBL_EPSILON); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)DBL_EPSILON); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}
__device__ __forceinline__ void sift_solve3f(const float a[3][3], const float b[3], float x[3])     // Matx33f::solve(b, DECOMP_LU): Cramer in float
{
    float d = (float)(double)(a[0][0]*(a[1][1]*a[2][2] - a[2][1]*a[1][2]) - a[0][1]*(a[1][0]*a[2][2] - a[2][0]*a[1][2]) + a[0][2]*(a[1][0]*a[2][1] - a[2][0]*a[1][1]));
    if (d == 0) { x[0] = x[1] = x[2] = 0; return; }
    d = 1/d;
    x[0] = d*(b[0]*(a[1][1]*a[2][2] - a[1][2]*a[2][1]) - a[0][1]*(b[1]*a[2][2] - a[1][2]*b[2]) + a[0][2]*(b[1]*a[2][1] - a[1][1]*b[2]));
    x[1] = d*(a[0][0]*(b[1]*a[2][2] - a[1][2]*b[2]) - b[0]*(a[1][0]*a[2][2] - a[1][2]*a[2][0]) + a[0][2]*(a[1][0]*b[2] - b[1]*a[2][0]));
    x[2] = d*(a[0][0]*(a[1][1]*b[2] - b[1]*a[2][1]) - a[0][1]*(a[1][0]*b[2] - b[1]*a[2][0]) + b[0]*(a[1][0]*a[2][1] - a[1][1]*a[2][0]));
}

// ------------------------------------------------------------------------------------------ pyramid kernels
// resize(u8 -> float, 2w x 2h, INTER_LINEAR): the horizontal two-tap values of the two source rows, then the vertical combination
__device__ __forceinline__ void lin_coef(int d, int ssize, int* s0, float* a0, float* a1)
{
    float f = (float)((d + 0.5) * 0.5 - 0.5);
    int s = cv_floor_d(f);
    f -= s;
    if (s < 0) { f = 0; s = 0; }
    if (s + 1 >= ssize) { f = 0; s = ssize - 1; }
    *s0 = s; *a0 = 1.f - f; *a1 = f;
}
__global__ __launch_bounds__(256) void k_sift_resize2x(const uint8_t* __restrict__ img, int w, int h, float* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= 2 * w) return;
    int sx, sy; float a0, a1, b0, b1;
    lin_coef(x, w, &sx, &a0, &a1);
    lin_coef(y, h, &sy, &b0, &b1);
    const int sx1 = sx + 1 < w ? sx + 1 : sx, sy1 = sy + 1 < h ? sy + 1 : sy;
    const float h0 = (float)img[(size_t)sy * w + sx] * a0 + (float)img[(size_t)sy * w + sx1] * a1;
    const float h1 = (float)img[(size_t)sy1 * w + sx] * a0 + (float)img[(size_t)sy1 * w + sx1] * a1;
    dst[(size_t)y * (2 * w) + x] = h0 * b0 + h1 * b1;
}
struct SiftTaps { float k[kSiftMaxTaps]; int r; };
// SymmRowFilter / SymmColumnFilter: s = k0 x0 + sum_i ki (x+i + x-i), BORDER_REFLECT_101
__global__ __launch_bounds__(256) void k_sift_blur(const float* __restrict__ src, float* __restrict__ dst, int w, int h, SiftTaps t, int vertical)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int r = t.r;
    float acc = t.k[r] * src[(size_t)y * w + x];
    if (!vertical) {
        const float* s = src + (size_t)y * w;
        for (int i = 1; i <= r; i++) acc += t.k[r + i] * (s[reflect101(x + i, w)] + s[reflect101(x - i, w)]);
    } else {
        for (int i = 1; i <= r; i++) acc += t.k[r + i] * (src[(size_t)reflect101(y + i, h) * w + x] + src[(size_t)reflect101(y - i, h) * w + x]);
    }
    dst[(size_t)y * w + x] = acc;
}
__global__ __launch_bounds__(256) void k_sift_half(const float* __restrict__ src, int sw, float* __restrict__ dst, int w, int h)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x < w) dst[(size_t)y * w + x] = src[(size_t)(2 * y) * sw + 2 * x];           // INTER_NEAREST
}
__global__ __launch_bounds__(256) void k_sift_dog(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ d, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = b[i] - a[i];
}

// ------------------------------------------------------------------------------------------ extrema
__global__ __launch_bounds__(256) void k_sift_extrema(const float* __restrict__ prev, const float* __restrict__ cur, const float* __restrict__ next,
                                                      int w, int h, int o, int layer, int threshold, SiftCand* cand, int* cnt, int cap)
{
    const int c = blockIdx.x * 256 + threadIdx.x + SIFT_IMG_BORDER, r = blockIdx.y + SIFT_IMG_BORDER;
    if (c >= w - SIFT_IMG_BORDER || r >= h - SIFT_IMG_BORDER) return;
    const float val = cur[(size_t)r * w + c];
    if (!(fabsf(val) > threshold)) return;
    bool ext = true;
    for (int dr = -1; dr <= 1 && ext; dr++)
        for (int dc = -1; dc <= 1 && ext; dc++) {
            const size_t e = (size_t)(r + dr) * w + (c + dc);
            const float a = cur[e], b = prev[e], cc = next[e];
            ext = val > 0 ? (val >= a && val >= b && val >= cc) : (val <= a && val <= b && val <= cc);
        }
    if (!ext) return;
    const int pos = atomicAdd(cnt, 1);
    if (pos < cap) cand[pos] = SiftCand{ o, layer, r, c };
}

struct SiftPyr { const float* gauss[kSiftMaxOctaves * (kSiftMaxLayers + 3)]; const float* dog[kSiftMaxOctaves * (kSiftMaxLayers + 2)]; int ow[kSiftMaxOctaves], oh[kSiftMaxOctaves]; int nL; };

// adjustLocalExtrema + calcOrientationHist + the peak loop of findScaleSpaceExtrema for one candidate
__global__ __launch_bounds__(64) void k_sift_refine(SiftPyr p, const SiftCand* __restrict__ cand, const int* __restrict__ cnt_p, int cand_cap,
                                                    float contrastThreshold, float edgeThreshold, float sigma, const float* __restrict__ exptab,
                                                    uvo_keypoint* raw, int* raw_cnt, int raw_cap)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    const int ncand = min(*cnt_p, cand_cap);
    if (t >= ncand) return;
    const int octv = cand[t].o, nL = p.nL, w = p.ow[octv], h = p.oh[octv];
    int layer = cand[t].layer, r = cand[t].r, c = cand[t].c;
    const float img_scale = 1.f / 255, deriv_scale = img_scale * 0.5f, second_deriv_scale = img_scale, cross_deriv_scale = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0, contr = 0;
    int i = 0;
#define AT(m, rr, cc) (m)[(size_t)(rr) * w + (cc)]
    for (; i < SIFT_MAX_INTERP_STEPS; i++) {
        const float* img = p.dog[octv * (nL + 2) + layer]; const float* prev = p.dog[octv * (nL + 2) + layer - 1]; const float* next = p.dog[octv * (nL + 2) + layer + 1];
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
        sift_solve3f(H, dD, X);
        xi = -X[2]; xr = -X[1]; xc = -X[0];
        if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
        if (fabsf(xi) > (float)(2147483647 / 3) || fabsf(xr) > (float)(2147483647 / 3) || fabsf(xc) > (float)(2147483647 / 3)) return;
        c += cv_round_f(xc); r += cv_round_f(xr); layer += cv_round_f(xi);
        if (layer < 1 || layer > nL || c < SIFT_IMG_BORDER || c >= w - SIFT_IMG_BORDER || r < SIFT_IMG_BORDER || r >= h - SIFT_IMG_BORDER) return;
    }
    if (i >= SIFT_MAX_INTERP_STEPS) return;
    {
        const float* img = p.dog[octv * (nL + 2) + layer]; const float* prev = p.dog[octv * (nL + 2) + layer - 1]; const float* next = p.dog[octv * (nL + 2) + layer + 1];
        const float dD[3] = { (AT(img, r, c + 1) - AT(img, r, c - 1)) * deriv_scale, (AT(img, r + 1, c) - AT(img, r - 1, c)) * deriv_scale,
                              (AT(next, r, c) - AT(prev, r, c)) * deriv_scale };
        const float tt = dD[0] * xc + dD[1] * xr + dD[2] * xi;
        contr = AT(img, r, c) * img_scale + tt * 0.5f;
        if (fabsf(contr) * nL < contrastThreshold) return;
        const float v2 = AT(img, r, c) * 2.f;
        const float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * second_deriv_scale;
        const float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * second_deriv_scale;
        const float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cross_deriv_scale;
        const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        if (det <= 0 || tr * tr * edgeThreshold >= (edgeThreshold + 1) * (edgeThreshold + 1) * det) return;
    }
    uvo_keypoint kpt;
    kpt.x = (c + xc) * (1 << octv);
    kpt.y = (r + xr) * (1 << octv);
    kpt.octave = octv + (layer << 8) + (cv_round_d((xi + 0.5) * 255) << 16);
    kpt.size = sigma * sift_exp2f_det((layer + xi) / nL) * (1 << octv) * 2;
    kpt.response = fabsf(contr);
    kpt.class_id = -1;
    kpt.angle = -1;
    // calcOrientationHist on the Gaussian layer the extremum ended in
    const float scl_octv = kpt.size * 0.5f / (1 << octv);
    const int radius = cv_round_f(4.5f * scl_octv), n = SIFT_ORI_HIST_BINS;
    const float osigma = 1.5f * scl_octv, expf_scale = -1.f / (2.f * osigma * osigma);
    const float* g = p.gauss[octv * (nL + 3) + layer];
    float temphist[SIFT_ORI_HIST_BINS + 4];
    float* th = temphist + 2;
    for (int k = 0; k < n; k++) th[k] = 0.f;
    for (int ii = -radius; ii <= radius; ii++) {
        const int y = r + ii;
        if (y <= 0 || y >= h - 1) continue;
        for (int jj = -radius; jj <= radius; jj++) {
            const int x = c + jj;
            if (x <= 0 || x >= w - 1) continue;
            const float dx = AT(g, y, x + 1) - AT(g, y, x - 1), dy = AT(g, y - 1, x) - AT(g, y + 1, x);
            const float wgt = sift_exp32f((ii * ii + jj * jj) * expf_scale, exptab);
            const float ori = sift_atan2_deg(dy, dx), mag = sqrtf(dx * dx + dy * dy);
            int bin = cv_round_f((n / 360.f) * ori);
            if (bin >= n) bin -= n;
            if (bin < 0) bin += n;
            th[bin] += wgt * mag;
        }
    }
#undef AT
    th[-1] = th[n - 1]; th[-2] = th[n - 2]; th[n] = th[0]; th[n + 1] = th[1];
    float hist[SIFT_ORI_HIST_BINS];
    for (int k = 0; k < n; k++) hist[k] = (th[k - 2] + th[k + 2]) * (1.f / 16.f) + (th[k - 1] + th[k + 1]) * (4.f / 16.f) + th[k] * (6.f / 16.f);
    float omax = hist[0];
    for (int k = 1; k < n; k++) omax = omax > hist[k] ? omax : hist[k];
    const float mag_thr = omax * 0.8f;
    for (int j = 0; j < n; j++) {
        const int l = j > 0 ? j - 1 : n - 1, r2 = j < n - 1 ? j + 1 : 0;
        if (hist[j] > hist[l] && hist[j] > hist[r2] && hist[j] >= mag_thr) {
            float bin = j + 0.5f * (hist[l] - hist[r2]) / (hist[l] - 2 * hist[j] + hist[r2]);
            bin = bin < 0 ? n + bin : (bin >= n ? bin - n : bin);
            kpt.angle = 360.f - (360.f / n) * bin;
            if (fabsf(kpt.angle - 360.f) < FLT_EPSILON) kpt.angle = 0.f;
            const int pos = atomicAdd(raw_cnt, 1);
            if (pos < raw_cap) raw[pos] = kpt;
        }
    }
}

// calcSIFTDescriptor for one keypoint per thread: the votes are float additions into shared bins in sample order, so the window
// is walked by one thread; the 6 x 6 x 10 histogram lives in LDS (one column per thread of the block)
static const int kSiftDescThreads = 32, kSiftHist = 6 * 6 * 10;
__global__ __launch_bounds__(kSiftDescThreads) void k_sift_descriptor(SiftPyr p, const uvo_keypoint* __restrict__ kps, int nk, const float* __restrict__ exptab,
                                                                     float* __restrict__ desc)
{
    __shared__ float s_hist[kSiftHist * kSiftDescThreads];
    const int k = blockIdx.x * kSiftDescThreads + threadIdx.x;
    if (k >= nk) return;
    float* hist = s_hist + threadIdx.x;
#define HI(i) hist[(i) * kSiftDescThreads]
    const uvo_keypoint kp = kps[k];
    const int d = 4, n = 8, nL = p.nL;
    int octave = kp.octave & 255; const int layer = (kp.octave >> 8) & 255;
    octave = octave < 128 ? octave : (-128 | octave);
    const float scale = octave >= 0 ? 1.f / (1 << octave) : (float)(1 << -octave);
    const float size = kp.size * scale;
    const int oi = octave + 1;                                        // octave - firstOctave
    const float* img = p.gauss[oi * (nL + 3) + layer];
    const int cols = p.ow[oi], rows = p.oh[oi];
    float ori = 360.f - kp.angle;
    if (fabsf(ori - 360.f) < FLT_EPSILON) ori = 0.f;
    const float ptx = kp.x * scale, pty = kp.y * scale, scl = size * 0.5f;
    const int px = cv_round_f(ptx), py = cv_round_f(pty);
    double sd, cd;
    det_sincos((double)(ori * (float)(3.14159265358979323846 / 180)), &sd, &cd);
    float cos_t = (float)cd, sin_t = (float)sd;
    const float bins_per_rad = n / 360.f, exp_scale = -1.f / (d * d * 0.5f), hist_width = 3.f * scl;
    int radius = cv_round_f(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
    const int rmax = (int)sqrt(((double)cols) * cols + ((double)rows) * rows);
    radius = radius < rmax ? radius : rmax;
    cos_t /= hist_width; sin_t /= hist_width;
    for (int e = 0; e < kSiftHist; e++) HI(e) = 0.f;
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            const float c_rot = j * cos_t - i * sin_t, r_rot = j * sin_t + i * cos_t;
            float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
            const int r = py + i, c = px + j;
            if (!(rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < rows - 1 && c > 0 && c < cols - 1)) continue;
            const float dx = img[(size_t)r * cols + c + 1] - img[(size_t)r * cols + c - 1], dy = img[(size_t)(r - 1) * cols + c] - img[(size_t)(r + 1) * cols + c];
            const float wgt = sift_exp32f((c_rot * c_rot + r_rot * r_rot) * exp_scale, exptab);
            float obin = (sift_atan2_deg(dy, dx) - ori) * bins_per_rad;
            const float mag = sqrtf(dx * dx + dy * dy) * wgt;
            const int r0 = cv_floor_d(rbin), c0 = cv_floor_d(cbin);
            int o0 = cv_floor_d(obin);
            rbin -= r0; cbin -= c0; obin -= o0;
            if (o0 < 0) o0 += n;
            if (o0 >= n) o0 -= n;
            const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
            const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11, v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
            const float v_rco111 = v_rc11 * obin, v_rco110 = v_rc11 - v_rco111, v_rco101 = v_rc10 * obin, v_rco100 = v_rc10 - v_rco101;
            const float v_rco011 = v_rc01 * obin, v_rco010 = v_rc01 - v_rco011, v_rco001 = v_rc00 * obin, v_rco000 = v_rc00 - v_rco001;
            const int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0;
            HI(idx) += v_rco000; HI(idx + 1) += v_rco001;
            HI(idx + (n + 2)) += v_rco010; HI(idx + (n + 3)) += v_rco011;
            HI(idx + (d + 2) * (n + 2)) += v_rco100; HI(idx + (d + 2) * (n + 2) + 1) += v_rco101;
            HI(idx + (d + 3) * (n + 2)) += v_rco110; HI(idx + (d + 3) * (n + 2) + 1) += v_rco111;
        }
    float* dst = desc + (size_t)k * 128;
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) {
            const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
            HI(idx) += HI(idx + n);
            HI(idx + 1) += HI(idx + n + 1);
            for (int q = 0; q < n; q++) dst[(i * d + j) * n + q] = HI(idx + q);
        }
#undef HI
    const int len = d * d * n;
    float nrm2 = 0;
    for (int q = 0; q < len; q++) nrm2 += dst[q] * dst[q];
    const float thr = sqrtf(nrm2) * 0.2f;
    nrm2 = 0;
    for (int q = 0; q < len; q++) { const float val = dst[q] < thr ? dst[q] : thr; dst[q] = val; nrm2 += val * val; }
    const float sq = sqrtf(nrm2);
    nrm2 = 512.f / (sq > FLT_EPSILON ? sq : FLT_EPSILON);
    for (int q = 0; q < len; q++) {
        const int v = cv_round_f(dst[q] * nrm2);                       // saturate_cast<uchar>
        dst[q] = (float)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

// ------------------------------------------------------------------------------------------ host
static int sift_gauss_kernel(double sigma, float* k)              // getGaussianKernel(n, sigma, CV_32F); n = cvRound(8 sigma + 1) | 1 for CV_32F images
{
    int n = cv_round_d(sigma * 4 * 2 + 1) | 1;
    if (n > kSiftMaxTaps - 1) n = kSiftMaxTaps - 1;
    const double scale2X = -0.5 / (sigma * sigma);
    double t[kSiftMaxTaps], sum = 0;
    for (int i = 0; i < n; i++) { const double x = i - (n - 1) * 0.5; t[i] = exp(scale2X * x * x); sum += t[i]; }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] = (float)(t[i] * sum);
    return n;
}
static uvo_status sift_blur(Ctx* c, SiftWs* s, const float* src, float* dst, int w, int h, double sigma)
{
    SiftTaps t;
    memset(&t, 0, sizeof(t));
    const int n = sift_gauss_kernel(sigma, t.k);
    t.r = n / 2;
    dim3 grid((w + 255) / 256, h);
    hipLaunchKernelGGL(k_sift_blur, grid, dim3(256), 0, c->stream, src, s->tmp, w, h, t, 0);
    hipLaunchKernelGGL(k_sift_blur, grid, dim3(256), 0, c->stream, static_cast<const float*>(s->tmp), dst, w, h, t, 1);
    UVO_HIP_TRY(c, hipGetLastError());
    return UVO_OK;
}

static bool kp_less(const uvo_keypoint& a, const uvo_keypoint& b)       // KeyPoint_LessThan (features2d keypoint.cpp)
{
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    if (a.size != b.size) return a.size > b.size;
    if (a.angle != b.angle) return a.angle < b.angle;
    if (a.response != b.response) return a.response > b.response;
    if (a.octave != b.octave) return a.octave > b.octave;
    if (a.class_id != b.class_id) return a.class_id > b.class_id;
    return false;
}

uvo_status sift_detect(Ctx* c, const uint8_t* gray, int w, int h, int stride, int mem, int nfeatures, int nL, double contrastThreshold,
                       double edgeThreshold, double sigma, uvo_keypoint* kps, float* desc, int cap, int* n_out)
{
    if (nL < 1 || nL > kSiftMaxLayers || w < 16 || h < 16 || w > c->max_w || h > c->max_h || sigma <= 0.5) { c->err = "uvo_sift_detect: nOctaveLayers 1..8, sigma > 0.5, image within the context's size"; return UVO_INVALID_ARG; }
    if (!c->sift_ws) c->sift_ws = new SiftWs();
    SiftWs* s = static_cast<SiftWs*>(c->sift_ws);
    const int nOct = std::min(kSiftMaxOctaves, std::max(1, cv_round_d(log((double)(2 * std::min(w, h))) / log(2.) - 2) + 1));    // firstOctave = -1
    if (s->w != w || s->h != h || s->nL != nL) {
        UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
        sift_ws_release(s);
        int ow = 2 * w, oh = 2 * h;
        bool ok = hipMalloc(reinterpret_cast<void**>(&s->tmp), sizeof(float) * (size_t)ow * oh) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&s->d_img), (size_t)w * h) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&s->d_cnt), sizeof(int) * 4) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&s->d_exptab), sizeof(float) * 64) == hipSuccess;
        for (int o = 0; o < nOct && ok; o++) {
            s->ow[o] = ow; s->oh[o] = oh;
            for (int i = 0; i < nL + 3 && ok; i++) ok = hipMalloc(reinterpret_cast<void**>(&s->gauss[o * (nL + 3) + i]), sizeof(float) * (size_t)ow * oh) == hipSuccess;
            for (int i = 0; i < nL + 2 && ok; i++) ok = hipMalloc(reinterpret_cast<void**>(&s->dog[o * (nL + 2) + i]), sizeof(float) * (size_t)ow * oh) == hipSuccess;
            ow /= 2; oh /= 2;
            if (ow < 1 || oh < 1) { ok = ok && o + 1 >= nOct; }
        }
        ok = ok && sift_grow(s, 8 * c->cap, 4 * c->cap, c->cap);
        if (!ok) { sift_ws_release(s); c->err = "uvo_sift_detect: out of device memory for the scale-space pyramid"; return UVO_HIP_ERROR; }
        float tab[64];
        for (int i = 0; i < 64; i++) tab[i] = (float)(pow(2.0, (double)i / 64) * .9670371139572337719125840413672004409288e-2);     // hal::exp32f's table
        UVO_HIP_TRY(c, hipMemcpy(s->d_exptab, tab, sizeof(tab), hipMemcpyHostToDevice));
        s->w = w; s->h = h; s->nL = nL; s->nOct = nOct;
    }
    hipStream_t st = c->stream;
    const uint8_t* d_img = gray;
    if (!(mem == UVO_MEM_DEVICE && stride == w)) {
        UVO_HIP_TRY(c, hipMemcpy2DAsync(s->d_img, w, gray, stride, w, h, mem == UVO_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        d_img = s->d_img;
    }
    UVO_HIP_TRY(c, hipMemsetAsync(s->d_cnt, 0, sizeof(int) * 4, st));
    // createInitialImage + buildGaussianPyramid + buildDoGPyramid
    double sig[kSiftMaxLayers + 3];
    sig[0] = sigma;
    {
        const double k = pow(2., 1. / nL);
        for (int i = 1; i < nL + 3; i++) { const double sp = pow(k, (double)(i - 1)) * sigma, stt = sp * k; sig[i] = sqrt(stt * stt - sp * sp); }
    }
    float* base = s->dog[0];                                            // the doubled image before its blur: a buffer that is free until octave 0's differences
    hipLaunchKernelGGL(k_sift_resize2x, dim3((2 * w + 255) / 256, 2 * h), dim3(256), 0, st, d_img, w, h, base);
    const float sd2 = (float)sigma * (float)sigma - 0.5f * 0.5f * 4;
    const float sig_diff = sqrtf(sd2 > 0.01f ? sd2 : 0.01f);
    for (int o = 0; o < nOct; o++) {
        const int ow = s->ow[o], oh = s->oh[o];
        for (int i = 0; i < nL + 3; i++) {
            float* dst = s->gauss[o * (nL + 3) + i];
            if (o == 0 && i == 0) UVO_TRY(sift_blur(c, s, base, dst, ow, oh, (double)sig_diff));
            else if (i == 0) hipLaunchKernelGGL(k_sift_half, dim3((ow + 255) / 256, oh), dim3(256), 0, st, static_cast<const float*>(s->gauss[(o - 1) * (nL + 3) + nL]), s->ow[o - 1], dst, ow, oh);
            else UVO_TRY(sift_blur(c, s, s->gauss[o * (nL + 3) + i - 1], dst, ow, oh, sig[i]));
        }
        const size_t npx = (size_t)ow * oh;
        for (int i = 0; i < nL + 2; i++)
            hipLaunchKernelGGL(k_sift_dog, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, st, static_cast<const float*>(s->gauss[o * (nL + 3) + i]),
                               static_cast<const float*>(s->gauss[o * (nL + 3) + i + 1]), s->dog[o * (nL + 2) + i], npx);
    }
    UVO_HIP_TRY(c, hipGetLastError());
    // findScaleSpaceExtrema
    const int threshold = cv_floor_d(0.5 * contrastThreshold / nL * 255);
    SiftPyr p;
    memset(&p, 0, sizeof(p));
    for (int i = 0; i < nOct * (nL + 3); i++) p.gauss[i] = s->gauss[i];
    for (int i = 0; i < nOct * (nL + 2); i++) p.dog[i] = s->dog[i];
    for (int o = 0; o < nOct; o++) { p.ow[o] = s->ow[o]; p.oh[o] = s->oh[o]; }
    p.nL = nL;
    int cnt[4];
    for (int attempt = 0;; attempt++) {
        for (int o = 0; o < nOct; o++) {
            const int ow = s->ow[o], oh = s->oh[o];
            if (ow <= 2 * SIFT_IMG_BORDER || oh <= 2 * SIFT_IMG_BORDER) continue;
            for (int i = 1; i <= nL; i++)
                hipLaunchKernelGGL(k_sift_extrema, dim3((ow - 2 * SIFT_IMG_BORDER + 255) / 256, oh - 2 * SIFT_IMG_BORDER), dim3(256), 0, st,
                                   static_cast<const float*>(s->dog[o * (nL + 2) + i - 1]), static_cast<const float*>(s->dog[o * (nL + 2) + i]),
                                   static_cast<const float*>(s->dog[o * (nL + 2) + i + 1]), ow, oh, o, i, threshold, s->d_cand, s->d_cnt, s->cand_cap);
        }
        hipLaunchKernelGGL(k_sift_refine, dim3((s->cand_cap + 63) / 64), dim3(64), 0, st, p, static_cast<const SiftCand*>(s->d_cand), static_cast<const int*>(s->d_cnt),
                           s->cand_cap, (float)contrastThreshold, (float)edgeThreshold, (float)sigma, static_cast<const float*>(s->d_exptab), s->d_raw, s->d_cnt + 1, s->raw_cap);
        UVO_HIP_TRY(c, hipGetLastError());
        UVO_HIP_TRY(c, hipMemcpyAsync(cnt, s->d_cnt, sizeof(cnt), hipMemcpyDeviceToHost, st));
        UVO_HIP_TRY(c, hipStreamSynchronize(st));
        if (cnt[0] <= s->cand_cap && cnt[1] <= s->raw_cap) break;
        // a list overflowed: both counters still counted everything they saw, so one more pass with room for it is enough, unless the
        // candidate list was cut (then the keypoint count is a lower bound and a third pass may follow)
        if (attempt >= 3 || !sift_grow(s, cnt[0] + cnt[0] / 4, std::max(cnt[1] + cnt[1] / 4, cnt[0] > s->cand_cap ? cnt[0] : 0), 0)) {
            c->err = "uvo_sift_detect: out of device memory for the extrema lists"; return UVO_HIP_ERROR;
        }
        UVO_HIP_TRY(c, hipMemsetAsync(s->d_cnt, 0, sizeof(int) * 4, st));
    }
    // KeyPointsFilter::removeDuplicatedSorted, retainBest(nfeatures), the scaling back of firstOctave = -1: a few thousand records, on the host
    std::vector<uvo_keypoint> raw((size_t)cnt[1]);
    if (cnt[1]) UVO_HIP_TRY(c, hipMemcpy(raw.data(), s->d_raw, sizeof(uvo_keypoint) * raw.size(), hipMemcpyDeviceToHost));
    std::sort(raw.begin(), raw.end(), kp_less);
    std::vector<uvo_keypoint> fin;
    fin.reserve(raw.size());
    for (const uvo_keypoint& k : raw)
        if (fin.empty() || fin.back().x != k.x || fin.back().y != k.y || fin.back().size != k.size || fin.back().angle != k.angle) fin.push_back(k);
    if (nfeatures > 0 && (int)fin.size() > nfeatures) {
        // retainBest: everything whose response is at least the nfeatures-th largest, kept in sorted order (OpenCV leaves them in
        // nth_element's order, which is implementation-defined: DESIGN.md section 6)
        std::vector<float> resp(fin.size());
        for (size_t i = 0; i < fin.size(); i++) resp[i] = fin[i].response;
        std::nth_element(resp.begin(), resp.begin() + (nfeatures - 1), resp.end(), [](float a, float b) { return a > b; });
        const float amb = resp[(size_t)nfeatures - 1];
        size_t m = 0;
        for (size_t i = 0; i < fin.size(); i++) if (fin[i].response >= amb) fin[m++] = fin[i];
        fin.resize(m);
    }
    for (uvo_keypoint& k : fin) { k.octave = (k.octave & ~255) | ((k.octave + -1) & 255); k.x *= 0.5f; k.y *= 0.5f; k.size *= 0.5f; }
    const int nk = (int)fin.size();
    *n_out = nk;
    if ((kps || desc) && nk > cap) { c->err = "uvo_sift_detect: output capacity too small"; return UVO_CAPACITY; }
    if (desc && !sift_grow(s, 0, 0, nk)) { c->err = "uvo_sift_detect: out of device memory for the descriptors"; return UVO_HIP_ERROR; }
    if (kps && nk) memcpy(kps, fin.data(), sizeof(uvo_keypoint) * (size_t)nk);
    if (desc && nk) {
        UVO_HIP_TRY(c, hipMemcpyAsync(s->d_kps, fin.data(), sizeof(uvo_keypoint) * (size_t)nk, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_sift_descriptor, dim3((nk + kSiftDescThreads - 1) / kSiftDescThreads), dim3(kSiftDescThreads), 0, st, p,
                           static_cast<const uvo_keypoint*>(s->d_kps), nk, static_cast<const float*>(s->d_exptab), s->d_desc);
        UVO_HIP_TRY(c, hipGetLastError());
        UVO_HIP_TRY(c, hipMemcpyAsync(desc, s->d_desc, sizeof(float) * 128 * (size_t)nk, hipMemcpyDeviceToHost, st));
        UVO_HIP_TRY(c, hipStreamSynchronize(st));                      // (fin's upload is done too)
    }
    return UVO_OK;
}

// test hook: one Gaussian (dog = 0) or difference-of-Gaussians (dog = 1) layer of the last uvo_sift_detect, to a host buffer
uvo_status sift_layer(Ctx* c, int octave, int layer, int dog, float* out, int cap_floats, int* ow, int* oh)
{
    SiftWs* s = static_cast<SiftWs*>(c->sift_ws);
    if (!s || !s->w) { c->err = "uvo_sift_layer: no uvo_sift_detect has run on this context"; return UVO_INVALID_ARG; }
    if (octave < 0 || octave >= s->nOct || layer < 0 || layer >= s->nL + (dog ? 2 : 3)) { c->err = "uvo_sift_layer: no such layer"; return UVO_INVALID_ARG; }
    *ow = s->ow[octave]; *oh = s->oh[octave];
    const size_t n = (size_t)s->ow[octave] * s->oh[octave];
    if (!out) return UVO_OK;
    if ((size_t)cap_floats < n) { c->err = "uvo_sift_layer: output capacity too small"; return UVO_CAPACITY; }
    const float* src = dog ? s->dog[octave * (s->nL + 2) + layer] : s->gauss[octave * (s->nL + 3) + layer];
    UVO_HIP_TRY(c, hipMemcpy(out, src, sizeof(float) * n, hipMemcpyDeviceToHost));
    return UVO_OK;
}

}  // namespace uvo
