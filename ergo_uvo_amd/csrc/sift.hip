// sift.hip -- the SIFT branch of detect_features (uvo_libraries/src/VO_utility.cpp:107-112):
//     Ptr<SIFT> detector = SIFT::create(10000, 3, 0.03, 10, 1.6);  detector->detectAndCompute(img, noArray(), keypoints, descriptors);
// SURVEY.md 8(f) N4 (the shipped parameter files select SURF; this is the next detector of the switch).  OpenCV 4.5 features2d:
// createInitialImage (u8 -> float, doubled with INTER_LINEAR, blurred to sigma), buildGaussianPyramid (nOctaveLayers + 3 blurs per
// octave, INTER_NEAREST halving), buildDoGPyramid, findScaleSpaceExtrema (26-neighbour extrema, adjustLocalExtrema,
// calcOrientationHist), KeyPointsFilter (duplicates, retainBest), calcSIFTDescriptor -- in the operation order of their scalar paths
// (parity vs OpenCV itself is UNPINNED: DESIGN.md section 0; cosf / sinf / powf(2, x) are this library's deterministic double series).
//
// Layout on the device (DESIGN.md section 7 has the per-kernel figures and counters):
//   pyramid      float images that stay in HBM (0.5 GB at 1080p).  A blur is ONE launch: a 64 x 32 tile with its reflected border staged
//                in LDS (every load issued before the first store), the row filter (RowFilter: taps left to right) LDS -> LDS, the
//                column filter (SymmColumnFilter: centre, then the pairs) LDS -> HBM, and the difference of Gaussians beside it.
//                The octaves of <= 2048 pixels are built by one workgroup from LDS in one launch.
//   extrema      every octave and layer in one launch (3 x 3 max / min per layer, one list reservation per workgroup); refinement
//                a thread per candidate
//   orientation  a wave per extremum, descriptor a wave per keypoint: the float sums into the histograms are ordered (raster order of
//                the window), everything else about a sample is parallel -- samples are evaluated 64 at a time and their votes
//                applied in order by the lanes that own the bins (orientation) / the shares (descriptor)
//   filter       sort by KeyPoint_LessThan, duplicate removal, retainBest: ranks by counting, on the device
// One host synchronisation (the counts), then the copy of the results.  A standalone operator (uvo_sift_detect): the fused stereo /
// mono steps of this library run on SURF; the reference's loop written against the function surface runs on it (shim).
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
struct SiftSurv { uvo_keypoint kpt; int o, layer, r, c; };
struct SiftPyr { const float* gauss[kSiftMaxOctaves * (kSiftMaxLayers + 3)]; const float* dog[kSiftMaxOctaves * (kSiftMaxLayers + 2)]; int ow[kSiftMaxOctaves], oh[kSiftMaxOctaves]; int nL; };
struct SiftWs {
    int w = 0, h = 0, nL = 0, nOct = 0;
    int ow[kSiftMaxOctaves], oh[kSiftMaxOctaves];
    float* gauss[kSiftMaxOctaves * (kSiftMaxLayers + 3)] = {nullptr};
    float* dog[kSiftMaxOctaves * (kSiftMaxLayers + 2)] = {nullptr};
    float* tmp = nullptr; uint8_t* d_img = nullptr;
    float* d_exptab = nullptr;
    float* d_taps = nullptr; int* d_radii = nullptr; double taps_sigma = 0; int taps_nL = 0;     // the layers' filter taps, for k_sift_tail
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // slot 1 in the synchronous step: the lane's second stream (sift_detect_lane)
    SiftCand* d_cand = nullptr; SiftSurv* d_surv = nullptr; uvo_keypoint* d_raw = nullptr; uvo_keypoint* d_kps = nullptr; float* d_desc = nullptr;
    uvo_keypoint* d_sorted = nullptr; uvo_keypoint* d_kept = nullptr; int* d_ints = nullptr;      // rank, dup, keep, greater: raw_cap each
    int* d_cnt = nullptr;         // [0] candidates, [1] raw keypoints, [2] refined extrema, [3] after duplicate removal, [4] final
    int cand_cap = 0, raw_cap = 0;           // d_kps / d_desc and the filter's arrays hold raw_cap records
};
static void sift_ws_release(SiftWs* s)
{
    for (float*& p : s->gauss) { (void)hipFree(p); p = nullptr; }
    for (float*& p : s->dog) { (void)hipFree(p); p = nullptr; }
    (void)hipFree(s->tmp); (void)hipFree(s->d_img); (void)hipFree(s->d_cand); (void)hipFree(s->d_surv); (void)hipFree(s->d_raw); (void)hipFree(s->d_kps); (void)hipFree(s->d_desc);
    (void)hipFree(s->d_cnt); (void)hipFree(s->d_exptab); (void)hipFree(s->d_taps); (void)hipFree(s->d_radii); s->d_taps = nullptr; s->d_radii = nullptr; s->taps_sigma = 0; (void)hipFree(s->d_sorted); (void)hipFree(s->d_kept); (void)hipFree(s->d_ints);
    s->d_sorted = nullptr; s->d_kept = nullptr; s->d_ints = nullptr;
    s->tmp = nullptr; s->d_img = nullptr; s->d_cand = nullptr; s->d_surv = nullptr; s->d_raw = nullptr; s->d_kps = nullptr; s->d_desc = nullptr; s->d_cnt = nullptr; s->d_exptab = nullptr;
    s->w = s->h = 0; s->cand_cap = s->raw_cap = 0;
}
// the three record lists grow on demand (a 1080p frame has ~13000 keypoints before retainBest, noise images far more per pixel)
static bool sift_grow(SiftWs* s, int cand, int raw)
{
    if (cand > s->cand_cap) {
        (void)hipFree(s->d_cand); (void)hipFree(s->d_surv); s->d_cand = nullptr; s->d_surv = nullptr; s->cand_cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&s->d_cand), sizeof(SiftCand) * (size_t)cand) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&s->d_surv), sizeof(SiftSurv) * (size_t)cand) != hipSuccess) return false;
        s->cand_cap = cand;
    }
    if (raw > s->raw_cap) {
        raw = (raw + 255) & ~255;
        (void)hipFree(s->d_raw); (void)hipFree(s->d_kps); (void)hipFree(s->d_desc); (void)hipFree(s->d_sorted); (void)hipFree(s->d_kept); (void)hipFree(s->d_ints);
        s->d_raw = s->d_kps = s->d_sorted = s->d_kept = nullptr; s->d_desc = nullptr; s->d_ints = nullptr; s->raw_cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&s->d_raw), sizeof(uvo_keypoint) * (size_t)raw) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&s->d_kps), sizeof(uvo_keypoint) * (size_t)raw) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&s->d_sorted), sizeof(uvo_keypoint) * (size_t)raw) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&s->d_kept), sizeof(uvo_keypoint) * (size_t)raw) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&s->d_ints), sizeof(int) * 4 * (size_t)raw) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&s->d_desc), sizeof(float) * 128 * (size_t)raw) != hipSuccess) return false;
        s->raw_cap = raw;
    }
    return true;
}
void sift_ws_free(Ctx* c)
{
    for (int i = 0; i < 2; i++) {
        SiftWs* s = static_cast<SiftWs*>(c->sift_ws[i]);
        if (!s) continue;
        if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
        if (s->ev_join) (void)hipEventDestroy(s->ev_join);
        sift_ws_release(s);
        delete s;
        c->sift_ws[i] = nullptr;
    }
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
// GaussianBlur on floats through the filter engine: kernels wider than 5 taps get the generic RowFilter (s = k[0] x[-r]; s += k[t] x[-r+t],
// left to right) and, being symmetrical, SymmColumnFilter (s = k0 x0; s += ki (x[+i] + x[-i])); BORDER_REFLECT_101
__global__ __launch_bounds__(256) void k_sift_blur(const float* __restrict__ src, float* __restrict__ dst, int w, int h, SiftTaps t, int vertical)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int r = t.r;
    float acc;
    if (!vertical) {                                                  // RowFilter: the taps left to right
        const float* s = src + (size_t)y * w;
        acc = t.k[0] * s[reflect101(x - r, w)];
        for (int i = 1; i <= 2 * r; i++) acc += t.k[i] * s[reflect101(x - r + i, w)];
    } else {                                                          // SymmColumnFilter: centre, then the pairs
        acc = t.k[r] * src[(size_t)y * w + x];
        for (int i = 1; i <= r; i++) acc += t.k[r + i] * (src[(size_t)reflect101(y + i, h) * w + x] + src[(size_t)reflect101(y - i, h) * w + x]);
    }
    dst[(size_t)y * w + x] = acc;
}
// The same blur, both passes in one launch for the radii the default parameters produce (R = 2 .. 16): a 64 x TH tile of outputs per
// workgroup, its input with an R-wide reflected border staged in LDS once, the row filter from LDS into LDS (each lane four
// neighbouring outputs from one register window of 4 + 2R inputs, read as 16-byte vectors), the column filter from LDS (each lane
// eight rows of one column from a window of 8 + 2R), and the difference of Gaussians dst - src written beside dst from the
// centre value that is already in LDS.  Every output is the expression of k_sift_blur (same taps, same order, no contraction): the
// row filter left to right, the column filter centre first and then the pairs.
template <int R, int TH>
__global__ __launch_bounds__(256) void k_sift_blur_tile(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ dog, int w, int h, SiftTaps t)
{
    constexpr int TW = 64, IW = TW + 2 * R, IWP = (IW + 3) & ~3, IH = TH + 2 * R;
    __shared__ __attribute__((aligned(16))) float s_in[IH * IWP];
    __shared__ float s_h[IH * TW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    {   // a wave stages rows wv, wv + 4, ...: every load of the tile is issued before the first LDS store (one memory round trip per
        // workgroup instead of one per row); the reflected column indices do not depend on the row
        constexpr int NROW = (IH + 3) / 4;
        const int ca = reflect101(x0 - R + lane, w), cb = reflect101(x0 - R + 64 + lane, w);
        float va[NROW], vb[NROW];
#pragma unroll
        for (int q = 0; q < NROW; q++) {
            const int iy = wv + 4 * q;
            const float* __restrict__ row = src + (size_t)reflect101(y0 - R + min(iy, IH - 1), h) * w;
            va[q] = row[ca];
            vb[q] = lane < 2 * R ? row[cb] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < NROW; q++) {
            const int iy = wv + 4 * q;
            if (iy < IH) {
                s_in[iy * IWP + lane] = va[q];
                if (lane < 2 * R) s_in[iy * IWP + 64 + lane] = vb[q];
            }
        }
    }
    __syncthreads();
    float k[R + 1];                                                     // k[i] = the tap at distance i from the centre (the kernel is symmetrical)
#pragma unroll
    for (int i = 0; i <= R; i++) k[i] = t.k[R + i];
    {   // rows: 16 lanes x 4 outputs per row, 16 rows per pass
        const int xq = (tid & 15) * 4;
        for (int iy = tid >> 4; iy < IH; iy += 16) {
            float win[4 + 2 * R + 3];
            const float4* __restrict__ src4 = reinterpret_cast<const float4*>(&s_in[iy * IWP + xq]);
#pragma unroll
            for (int q = 0; q < (4 + 2 * R + 3) / 4; q++) { const float4 f = src4[q]; win[4 * q] = f.x; win[4 * q + 1] = f.y; win[4 * q + 2] = f.z; win[4 * q + 3] = f.w; }
            float acc[4];
#pragma unroll
            for (int o = 0; o < 4; o++) {                               // RowFilter: left to right, no pairing
                acc[o] = k[R] * win[o];
#pragma unroll
                for (int i = 1; i <= 2 * R; i++) acc[o] += k[i < R ? R - i : i - R] * win[o + i];
            }
            *reinterpret_cast<float4*>(&s_h[iy * TW + xq]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
    __syncthreads();
    {   // columns: lane = column, 8 rows per thread
        constexpr int RPT = TH / 4;
        const int yq = wv * RPT, x = x0 + lane;
        float win[RPT + 2 * R];
#pragma unroll
        for (int q = 0; q < RPT + 2 * R; q++) win[q] = s_h[(yq + q) * TW + lane];
        if (x < w) {
#pragma unroll
            for (int o = 0; o < RPT; o++) {
                float acc = k[0] * win[o + R];
#pragma unroll
                for (int i = 1; i <= R; i++) acc += k[i] * (win[o + R + i] + win[o + R - i]);
                const int y = y0 + yq + o;
                if (y < h) {
                    dst[(size_t)y * w + x] = acc;
                    if (dog) dog[(size_t)y * w + x] = acc - s_in[(yq + o + R) * IWP + lane + R];
                }
            }
        }
    }
}

// The small octaves (at most 2048 pixels a layer; the arrays hold 4096) in ONE launch by one workgroup: the halving, the nL + 2 blurs (two plain passes
// through a scratch image) and the differences of every such octave, one after the other.  They are 30-odd dependent launches of a
// few microseconds each otherwise, most of it dispatch latency; the arithmetic per pixel is k_sift_blur's.
__global__ __launch_bounds__(1024) void k_sift_tail(SiftPyr p, const float* __restrict__ taps, const int* __restrict__ radii, int o_first, int nOct)
{
    __shared__ float s_a[4096], s_b[4096], s_t[4096];                   // the layer being blurred, its successor, the row-filtered image
    __shared__ float s_k[kSiftMaxTaps];
    const int tid = threadIdx.x, nL = p.nL;
    float* cur = s_a; float* nxt = s_b;
    for (int o = o_first; o < nOct; o++) {
        const int w = p.ow[o], h = p.oh[o], pw = p.ow[o - 1], n = w * h;
        float* g0 = const_cast<float*>(p.gauss[o * (nL + 3)]);
        const float* prev = p.gauss[(o - 1) * (nL + 3) + nL];           // (written by an earlier launch, or by this workgroup before a barrier)
        __syncthreads();
        for (int e = tid; e < n; e += 1024) { const int y = e / w, x = e - y * w; const float v = prev[(size_t)(2 * y) * pw + 2 * x]; cur[e] = v; g0[e] = v; }      // INTER_NEAREST
        for (int i = 1; i < nL + 3; i++) {
            float* dst = const_cast<float*>(p.gauss[o * (nL + 3) + i]);
            float* dg = const_cast<float*>(p.dog[o * (nL + 2) + i - 1]);
            const int r = radii[i];
            if (tid < kSiftMaxTaps) s_k[tid] = taps[i * kSiftMaxTaps + tid];
            __syncthreads();
            for (int e = tid; e < n; e += 1024) {
                const int y = e / w, x = e - y * w;
                const float* row = cur + y * w;
                float acc;
                if (r < w) {                                            // one reflection is enough (uniform): |p|, then 2(w - 1) - p
                    int q = x - r; q = q < 0 ? -q : q;
                    acc = s_k[0] * row[q];
                    for (int t = 1; t <= 2 * r; t++) { q = x - r + t; q = q < 0 ? -q : q; q = q >= w ? 2 * w - 2 - q : q; acc += s_k[t] * row[q]; }
                } else {
                    acc = s_k[0] * row[reflect101(x - r, w)];
                    for (int t = 1; t <= 2 * r; t++) acc += s_k[t] * row[reflect101(x - r + t, w)];
                }
                s_t[e] = acc;
            }
            __syncthreads();
            for (int e = tid; e < n; e += 1024) {
                const int y = e / w, x = e - y * w;
                float acc = s_k[r] * s_t[e];
                if (r < h) {
                    for (int t = 1; t <= r; t++) {
                        int qa = y + t, qb = y - t;
                        qa = qa >= h ? 2 * h - 2 - qa : qa; qb = qb < 0 ? -qb : qb;
                        acc += s_k[r + t] * (s_t[qa * w + x] + s_t[qb * w + x]);
                    }
                } else {
                    for (int t = 1; t <= r; t++) acc += s_k[r + t] * (s_t[reflect101(y + t, h) * w + x] + s_t[reflect101(y - t, h) * w + x]);
                }
                nxt[e] = acc; dst[e] = acc; dg[e] = acc - cur[e];
            }
            __syncthreads();
            float* sw = cur; cur = nxt; nxt = sw;
        }
    }
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
struct SiftTiles { int start[kSiftMaxOctaves + 1]; int tx[kSiftMaxOctaves]; int nOct; };
// findScaleSpaceExtrema's 26-neighbour test, every octave and layer in one launch: a workgroup takes a 64 x 16 tile of one octave and
// walks up its nL + 2 difference layers, staging each (with a one-pixel border) in LDS once; a thread keeps the 3 x 3 maximum, the
// 3 x 3 minimum and the centre of its four pixels for the last three layers in registers, and val >= all 26 neighbours is
// val >= the three maxima (val <= the three minima for a negative val) -- the same decisions as the 26 comparisons.
__global__ __launch_bounds__(256) void k_sift_extrema_all(SiftPyr p, SiftTiles tl, int threshold, SiftCand* cand, int* cnt, int cap)
{
    constexpr int TW = 64, TH = 16, SW = TW + 2, SH = TH + 2, SP = 68;
    __shared__ __attribute__((aligned(16))) float s_t[SH * SP];
    __shared__ unsigned short s_list[TW * TH * kSiftMaxLayers];        // layer << 10 | ly << 6 | lx: every pixel of the tile in every layer at worst
    __shared__ int s_n, s_base;
    if (threadIdx.x == 0) s_n = 0;
    int o = 0;
    while (o + 1 < tl.nOct && (int)blockIdx.x >= tl.start[o + 1]) o++;
    const int tile = blockIdx.x - tl.start[o], w = p.ow[o], h = p.oh[o], nL = p.nL;
    const int x0 = SIFT_IMG_BORDER + (tile % tl.tx[o]) * TW, y0 = SIFT_IMG_BORDER + (tile / tl.tx[o]) * TH;
    const int tid = threadIdx.x, lx = (tid & 15) * 4, ly = tid >> 4;
    float mx[3][4], mn[3][4], ce[3][4];
    constexpr int NST = (SH * SW + 255) / 256;                         // staged values per thread
    int goff[NST], soff[NST];
#pragma unroll
    for (int q = 0; q < NST; q++) {
        const int e = min(tid + q * 256, SH * SW - 1), sy = e / SW, sx = e - sy * SW;
        goff[q] = min(y0 - 1 + sy, h - 1) * w + min(x0 - 1 + sx, w - 1);          // (y0 - 1, x0 - 1 >= 4: inside; an octave has < 2^31 pixels)
        soff[q] = sy * SP + sx;
    }
    float pre[NST];
    {
        const float* __restrict__ d0 = p.dog[o * (nL + 2)];
#pragma unroll
        for (int q = 0; q < NST; q++) pre[q] = d0[goff[q]];
    }
    for (int l = 0; l < nL + 2; l++) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NST; q++) if (tid + q * 256 < SH * SW) s_t[soff[q]] = pre[q];
        if (l + 1 < nL + 2) {                                           // the next layer's loads are in flight while this one is tested
            const float* __restrict__ d1 = p.dog[o * (nL + 2) + l + 1];
#pragma unroll
            for (int q = 0; q < NST; q++) pre[q] = d1[goff[q]];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) { mx[0][q] = mx[1][q]; mn[0][q] = mn[1][q]; ce[0][q] = ce[1][q]; mx[1][q] = mx[2][q]; mn[1][q] = mn[2][q]; ce[1][q] = ce[2][q]; }
        float cmx[6], cmn[6];                                           // column maxima / minima over the three rows, columns lx - 1 .. lx + 4
        float rw[3][6];                                                 // (16-byte + 8-byte reads: a lane's six values of a row are contiguous)
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
            const float4 f4 = *reinterpret_cast<const float4*>(&s_t[(ly + rr) * SP + lx]);
            const float2 f2 = *reinterpret_cast<const float2*>(&s_t[(ly + rr) * SP + lx + 4]);
            rw[rr][0] = f4.x; rw[rr][1] = f4.y; rw[rr][2] = f4.z; rw[rr][3] = f4.w; rw[rr][4] = f2.x; rw[rr][5] = f2.y;
        }
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const float a = rw[0][q], b = rw[1][q], cc = rw[2][q];
            cmx[q] = fmaxf(fmaxf(a, b), cc); cmn[q] = fminf(fminf(a, b), cc);
            if (q >= 1 && q <= 4) ce[2][q - 1] = b;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) { mx[2][q] = fmaxf(fmaxf(cmx[q], cmx[q + 1]), cmx[q + 2]); mn[2][q] = fminf(fminf(cmn[q], cmn[q + 1]), cmn[q + 2]); }
        if (l >= 2) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c = x0 + lx + q, r = y0 + ly;
                const float val = ce[1][q];
                if (c < w - SIFT_IMG_BORDER && r < h - SIFT_IMG_BORDER && fabsf(val) > threshold) {
                    const bool ext = val > 0 ? (val >= mx[0][q] && val >= mx[1][q] && val >= mx[2][q]) : (val <= mn[0][q] && val <= mn[1][q] && val <= mn[2][q]);
                    if (ext) s_list[atomicAdd(&s_n, 1)] = (unsigned short)(((l - 1) << 10) | (ly << 6) | (lx + q));
                }
            }
        }
    }
    __syncthreads();
    const int nloc = s_n;                                               // one reservation in the global list per workgroup (the list is unordered anyway)
    if (tid == 0 && nloc) s_base = atomicAdd(cnt, nloc);
    __syncthreads();
    for (int e = tid; e < nloc; e += 256) {
        const int v = s_list[e], pos = s_base + e;
        if (pos < cap) cand[pos] = SiftCand{ o, v >> 10, y0 + ((v >> 6) & 15), x0 + (v & 63) };
    }
}


// adjustLocalExtrema for one candidate per thread (a handful of dependent 3 x 3 x 3 neighbourhood reads); survivors are appended to a list
__global__ __launch_bounds__(64) void k_sift_refine(SiftPyr p, const SiftCand* __restrict__ cand, const int* __restrict__ cnt_p, int cand_cap,
                                                    float contrastThreshold, float edgeThreshold, float sigma, SiftSurv* surv, int* surv_cnt)
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
#undef AT
    const int pos = atomicAdd(surv_cnt, 1);                            // at most one per candidate: the candidate list's capacity bounds it
    if (pos < cand_cap) surv[pos] = SiftSurv{ kpt, octv, layer, r, c };
}

// calcOrientationHist + the peak loop of findScaleSpaceExtrema, one wave per surviving extremum.  As in the descriptor the bins'
// sums are ordered (raster order of the window): the lanes evaluate 64 samples at a time (gradient, exp, atan2, sqrt -> bin, vote),
// pack them in raster order into LDS, and lane b < 36 then walks the packed votes adding the ones of bin b (and +0.0f, which
// changes nothing, for the others: the votes and the sums are never negative).
static const int kOriNB = 4;
__global__ __launch_bounds__(64) void k_sift_orient(SiftPyr p, const SiftSurv* __restrict__ surv, const int* __restrict__ surv_cnt, int surv_cap,
                                                    const float* __restrict__ exptab, uvo_keypoint* raw, int* raw_cnt, int raw_cap)
{
    __shared__ float s_tab[64];
    __shared__ __attribute__((aligned(8))) float2 s_vote[64 * kOriNB + 8];
    __shared__ float s_th[SIFT_ORI_HIST_BINS + 4], s_h[SIFT_ORI_HIST_BINS];
    const int lane = threadIdx.x, n = SIFT_ORI_HIST_BINS, nL = p.nL;
    const int ns = min(*surv_cnt, surv_cap);
    s_tab[lane] = exptab[lane];
    for (int sidx = blockIdx.x; sidx < ns; sidx += gridDim.x) {
        const SiftSurv sv = surv[sidx];
        uvo_keypoint kpt = sv.kpt;
        const int octv = sv.o, w = p.ow[octv], h = p.oh[octv], r = sv.r, c = sv.c;
        const float scl_octv = kpt.size * 0.5f / (1 << octv);
        const int radius = cv_round_f(4.5f * scl_octv);
        const float osigma = 1.5f * scl_octv, expf_scale = -1.f / (2.f * osigma * osigma);
        const float* __restrict__ g = p.gauss[octv * (nL + 3) + sv.layer];
        const int side = 2 * radius + 1;
        const long long total = (long long)side * side;
        float acc = 0.f;                                               // th[lane], lane < 36
        int wi = lane / side, wj = lane - wi * side;
        __syncthreads();
        // kOriNB x 64 samples per step, a lane taking samples t, t + 64, ...: their gradient loads go out together (one memory round
        // trip and one pair of barriers per kOriNB x 64 samples), the votes are packed batch after batch, i.e. still in raster order
        for (long long base = 0; base < total; base += 64 * kOriNB) {
            float gx1[kOriNB], gx0[kOriNB], gy0[kOriNB], gy1[kOriNB]; int d2[kOriNB]; bool pass[kOriNB];
#pragma unroll
            for (int u = 0; u < kOriNB; u++) {
                const int ii = wi - radius, jj = wj - radius, y = r + ii, x = c + jj;
                pass[u] = wi < side && y > 0 && y < h - 1 && x > 0 && x < w - 1;
                wj += 64;
                while (wj >= side) { wj -= side; wi++; }
                d2[u] = ii * ii + jj * jj;
                const size_t e = pass[u] ? (size_t)y * w + x : (size_t)w + 1;
                gx1[u] = g[e + 1]; gx0[u] = g[e - 1]; gy0[u] = g[e - w]; gy1[u] = g[e + w];
            }
            int filled = 0;
#pragma unroll
            for (int u = 0; u < kOriNB; u++) {
                float vote = 0.f; int bin = 0;
                if (pass[u]) {
                    const float dx = gx1[u] - gx0[u], dy = gy0[u] - gy1[u];
                    const float wgt = sift_exp32f(d2[u] * expf_scale, s_tab);
                    const float ori = sift_atan2_deg(dy, dx), mag = sqrtf(dx * dx + dy * dy);
                    bin = cv_round_f((n / 360.f) * ori);
                    if (bin >= n) bin -= n;
                    if (bin < 0) bin += n;
                    vote = wgt * mag;
                }
                const unsigned long long m = __ballot(pass[u]);
                if (pass[u]) {
                    const int pos = filled + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    s_vote[pos] = make_float2(__int_as_float(bin), vote);
                }
                filled += __popcll(m);
            }
            if (lane < 8) s_vote[filled + lane] = make_float2(__int_as_float(-1), 0.f);      // pad to a multiple of eight: bin -1 is nobody's
            __syncthreads();
            for (int q = 0; q < filled; q += 8) {                       // eight votes per LDS round trip
                float2 bv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) bv[u] = s_vote[q + u];
#pragma unroll
                for (int u = 0; u < 8; u++) acc += (__float_as_int(bv[u].x) == lane) ? bv[u].y : 0.f;
            }
            __syncthreads();
        }
        if (lane < n) s_th[lane + 2] = acc;
        __syncthreads();
        if (lane < 2) { s_th[lane] = s_th[n + lane]; s_th[n + 2 + lane] = s_th[2 + lane]; }      // th[-2], th[-1] = th[n-2], th[n-1]; th[n], th[n+1] = th[0], th[1]
        __syncthreads();
        if (lane < n) {
            const float* th = s_th + 2 + lane;
            s_h[lane] = (th[-2] + th[2]) * (1.f / 16.f) + (th[-1] + th[1]) * (4.f / 16.f) + th[0] * (6.f / 16.f);
        }
        __syncthreads();
        float omax = s_h[0];
        for (int q = 1; q < n; q++) omax = omax > s_h[q] ? omax : s_h[q];
        const float mag_thr = omax * 0.8f;
        if (lane < n) {
            const int j = lane, l = j > 0 ? j - 1 : n - 1, r2 = j < n - 1 ? j + 1 : 0;
            const float hj = s_h[j], hl = s_h[l], hr = s_h[r2];
            if (hj > hl && hj > hr && hj >= mag_thr) {
                float bin = j + 0.5f * (hl - hr) / (hl - 2 * hj + hr);
                bin = bin < 0 ? n + bin : (bin >= n ? bin - n : bin);
                kpt.angle = 360.f - (360.f / n) * bin;
                if (fabsf(kpt.angle - 360.f) < FLT_EPSILON) kpt.angle = 0.f;
                const int pos = atomicAdd(raw_cnt, 1);
                if (pos < raw_cap) raw[pos] = kpt;
            }
        }
    }
}

// calcSIFTDescriptor, one wave per keypoint.  The votes are float additions into shared bins and their order (raster order of the
// window's samples) is part of the result, but nothing else about a sample is: the lanes evaluate 64 samples at a time (rotation,
// window test, gradient, exp / atan2 / sqrt, the eight trilinear shares), the ones inside the window are packed in raster order into
// LDS, and eight lanes then apply one sample's eight shares (eight different bins) per step, sample after sample -- LDS operations of
// one wave are performed in issue order, so every bin sees its additions in the order the one-thread loop makes them.
static const int kSiftHist = 6 * 6 * 10;
#ifndef UVO_SIFT_EXPERIMENT
#define UVO_SIFT_EXPERIMENT 0            // timing experiments only: 1 = no ordered additions, 2 = no gradient / exp / atan2 / sqrt
#endif
__global__ __launch_bounds__(64) void k_sift_descriptor(SiftPyr p, const uvo_keypoint* __restrict__ kps, const int* __restrict__ nk_p, int kp_cap,
                                                        const float* __restrict__ exptab, float* __restrict__ desc)
{
    __shared__ float s_hist[kSiftHist];
    __shared__ __attribute__((aligned(16))) float2 s_rec[8][64 + 4];    // [share][queued sample] = (bin, value): a lane of the ordered part reads ITS share of four samples as two 16-byte vectors
    __shared__ float s_tab[64];
    __shared__ __attribute__((aligned(16))) float s_dst[128];
    __shared__ float s_sq[128];
    __shared__ int s_pend[128];                                        // queued samples (row << 16 | column), raster order
    const int lane = threadIdx.x, nk = min(*nk_p, kp_cap);
    s_tab[lane] = exptab[lane];
    for (int k = blockIdx.x; k < nk; k += gridDim.x) {
    const uvo_keypoint kp = kps[k];
    const int d = 4, n = 8, nL = p.nL;
    int octave = kp.octave & 255; const int layer = (kp.octave >> 8) & 255;
    octave = octave < 128 ? octave : (-128 | octave);
    const float scale = octave >= 0 ? 1.f / (1 << octave) : (float)(1 << -octave);
    const float size = kp.size * scale;
    const int oi = octave + 1;                                        // octave - firstOctave
    const float* __restrict__ img = p.gauss[oi * (nL + 3) + layer];
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
    __syncthreads();                                                  // (the previous keypoint's tail is done with s_hist / s_dst)
    for (int e = lane; e < kSiftHist; e += 64) s_hist[e] = 0.f;
    __syncthreads();
    const int side = 2 * radius + 1;
    const long long total = (long long)side * side;
    int wi = lane / side, wj = lane - wi * side;                      // this lane's sample of the current batch: row wi, column wj of the window
    // Two steps, because about half of the window's square lies outside the rotated 5 x 5 cell area: the cheap part (rotation and
    // window test) runs on every sample and queues the ones inside, in raster order; the expensive part (gradient, exp, atan2, sqrt,
    // shares) runs on 64 queued samples at a time, every lane busy.
    int npend = 0;
    for (long long base = 0; base < total || npend > 0; base += 64) {
        if (base < total) {
            bool pass = false;
            if (wi < side) {
                const int i = wi - radius, j = wj - radius;
                const float c_rot = j * cos_t - i * sin_t, r_rot = j * sin_t + i * cos_t;
                const float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
                const int r = py + i, c = px + j;
                pass = rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < rows - 1 && c > 0 && c < cols - 1;
            }
            const unsigned long long m = __ballot(pass);
            if (pass) s_pend[npend + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (wi << 16) | wj;
            npend += __popcll(m);
            wj += 64;
            while (wj >= side) { wj -= side; wi++; }
            if (npend < 64 && base + 64 < total) continue;
        }
        __syncthreads();
        const int cnt = npend < 64 ? npend : 64;
        if (lane < cnt) {
            const int pk = s_pend[lane], i = (pk >> 16) - radius, j = (pk & 0xFFFF) - radius;
            const float c_rot = j * cos_t - i * sin_t, r_rot = j * sin_t + i * cos_t;
            float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
            const int r = py + i, c = px + j;
#if UVO_SIFT_EXPERIMENT == 2
            float obin = 0.3f * i + 0.2f * j; const float mag = 1.f + c_rot;
#else
            const float dx = img[(size_t)r * cols + c + 1] - img[(size_t)r * cols + c - 1], dy = img[(size_t)(r - 1) * cols + c] - img[(size_t)(r + 1) * cols + c];
            const float wgt = sift_exp32f((c_rot * c_rot + r_rot * r_rot) * exp_scale, s_tab);
            float obin = (sift_atan2_deg(dy, dx) - ori) * bins_per_rad;
            const float mag = sqrtf(dx * dx + dy * dy) * wgt;
#endif
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
            // the eight bins of the shares: +0, +1, +(n+2), +(n+3), +(d+2)(n+2), ...
            const int b0 = idx, b2 = b0 + (n + 2), b4 = b0 + (d + 2) * (n + 2), b6 = b0 + (d + 3) * (n + 2);
            s_rec[0][lane] = make_float2(__int_as_float(4 * b0), v_rco000); s_rec[1][lane] = make_float2(__int_as_float(4 * b0 + 4), v_rco001);
            s_rec[2][lane] = make_float2(__int_as_float(4 * b2), v_rco010); s_rec[3][lane] = make_float2(__int_as_float(4 * b2 + 4), v_rco011);
            s_rec[4][lane] = make_float2(__int_as_float(4 * b4), v_rco100); s_rec[5][lane] = make_float2(__int_as_float(4 * b4 + 4), v_rco101);
            s_rec[6][lane] = make_float2(__int_as_float(4 * b6), v_rco110); s_rec[7][lane] = make_float2(__int_as_float(4 * b6 + 4), v_rco111);
        } else if (lane < cnt + 4) {
            // padding up to a multiple of four samples: +0 into the histogram's last element (cell (5, 5): never read)
#pragma unroll
            for (int cs = 0; cs < 8; cs++) s_rec[cs][lane] = make_float2(__int_as_float((kSiftHist - 1) * 4), 0.f);
        }
        const int rest = npend - cnt;                                  // < 64: what the last cheap step queued beyond the 64 taken now
        const int keep = lane < rest ? s_pend[64 + lane] : 0;
        __syncthreads();
        if (lane < rest) s_pend[lane] = keep;
        npend = rest;
        if (lane < 8 && UVO_SIFT_EXPERIMENT != 1) {
            // The ordered part: lane c applies share c of sample after sample.  A sample's eight bins are distinct, and while the next
            // sample has the same eight (the same cell and orientation bin: runs of a few samples along a window row) nobody else
            // touches them, so the running sums stay in the lanes; when the bins change (for all eight lanes at once) the sums are
            // written back and the new bins read -- LDS operations of a wave are performed in issue order, so a bin written by one
            // lane and read by another at the change is current.  What bounds the kernel is the LDS instruction rate of a CU with 25
            // such waves, hence two 16-byte reads per four samples and a write + a read per change instead of four operations a sample.
            const float4* __restrict__ rp = reinterpret_cast<const float4*>(&s_rec[lane][0]);
            // records carry byte offsets into s_hist; the lanes' bins all change at the same sample (bin = idx + a constant per
            // share), so the test is made once, on lane 0's, by the scalar unit; the sums start on the never-read last element
            const char* hb = reinterpret_cast<const char*>(s_hist);
            int cur = (kSiftHist - 1) * 4, scur = cur; float acc = 0.f;
            auto step = [&](float bf, float v) {
                const int bi = __float_as_int(bf), sbi = __builtin_amdgcn_readfirstlane(bi);
                if (sbi != scur) {
                    *reinterpret_cast<float*>(const_cast<char*>(hb) + cur) = acc;
                    acc = *reinterpret_cast<const float*>(hb + bi);
                    cur = bi; scur = sbi;
                }
                acc += v;
            };
            for (int q = 0; q < cnt; q += 4) {
                const float4 ra = rp[q / 2], rb = rp[q / 2 + 1];
                step(ra.x, ra.y); step(ra.z, ra.w); step(rb.x, rb.y); step(rb.z, rb.w);
            }
            *reinterpret_cast<float*>(const_cast<char*>(hb) + cur) = acc;
        }
        __syncthreads();
    }
    // the wrap of the orientation axis, the 4 x 4 x 8 rows, the two normalisations (sums in element order: one lane)
    if (lane < 16) {
        const int idx = ((lane / d + 1) * (d + 2) + (lane % d + 1)) * (n + 2);
        s_hist[idx] += s_hist[idx + n];
        s_hist[idx + 1] += s_hist[idx + n + 1];
    }
    __syncthreads();
    for (int e = lane; e < 128; e += 64) {
        const int cell = e / n, q = e % n;
        const float val = s_hist[((cell / d + 1) * (d + 2) + (cell % d + 1)) * (n + 2) + q];
        s_dst[e] = val; s_sq[e] = val * val;
    }
    __syncthreads();
    float nrm2 = 0;
    for (int q = 0; q < 128; q++) nrm2 += s_sq[q];                   // every lane the same chain (uniform)
    const float thr = sqrtf(nrm2) * 0.2f;
    __syncthreads();
    for (int e = lane; e < 128; e += 64) {
        const float val = s_dst[e] < thr ? s_dst[e] : thr;
        s_dst[e] = val; s_sq[e] = val * val;
    }
    __syncthreads();
    nrm2 = 0;
    for (int q = 0; q < 128; q++) nrm2 += s_sq[q];
    const float sq = sqrtf(nrm2);
    nrm2 = 512.f / (sq > FLT_EPSILON ? sq : FLT_EPSILON);
    float* dst = desc + (size_t)k * 128;
    for (int e = lane; e < 128; e += 64) {
        const int vv = cv_round_f(s_dst[e] * nrm2);                   // saturate_cast<uchar>
        dst[e] = (float)(vv < 0 ? 0 : (vv > 255 ? 255 : vv));
    }
    }
}

// ------------------------------------------------------------------------------------------ KeyPointsFilter on the device
// removeDuplicatedSorted (sort by KeyPoint_LessThan, drop a keypoint that repeats its predecessor's pt, size and angle), retainBest
// (keep the responses that are at least the nfeatures-th largest) and the scaling back of firstOctave = -1.  A few thousand
// records: ranks by counting (every record against every other, the pairs split over the grid and summed with integer atomics).
__device__ __forceinline__ bool sift_kp_less(const uvo_keypoint& a, const uvo_keypoint& b)      // KeyPoint_LessThan (features2d keypoint.cpp)
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
__global__ __launch_bounds__(256) void k_sift_rank(const uvo_keypoint* __restrict__ raw, const int* __restrict__ cnt_p, int cap, int* rank, int* dup)
{
    __shared__ __attribute__((aligned(16))) float s_key[256];
    __shared__ uvo_keypoint s_kp[256];
    const int n = min(*cnt_p, cap), tid = threadIdx.x, i = blockIdx.x * 256 + tid;
    if ((int)blockIdx.x * 256 >= n) return;
    const int chunk = (n + gridDim.y - 1) / gridDim.y, j0 = blockIdx.y * chunk, j1 = min(n, j0 + chunk);
    uvo_keypoint me = raw[min(i, n - 1)];
    const float kme = me.x;
    int r = 0, d = 0;
    for (int jb = j0; jb < j1; jb += 256) {
        __syncthreads();
        if (jb + tid < j1) { const uvo_keypoint o = raw[jb + tid]; s_kp[tid] = o; s_key[tid] = o.x; }
        else s_key[tid] = __int_as_float(0x7fc00000);                  // NaN: neither below nor equal
        __syncthreads();
        const int m = min(256, j1 - jb);
        int ties = 0;
        int r0 = 0, r1 = 0, r2 = 0, r3 = 0;                              // four chains instead of one
#pragma unroll 4
        for (int q = 0; q < 256; q += 4) {
            const float4 k4 = *reinterpret_cast<const float4*>(&s_key[q]);
            r0 += k4.x < kme; r1 += k4.y < kme; r2 += k4.z < kme; r3 += k4.w < kme;
            ties |= (k4.x == kme) | (k4.y == kme) | (k4.z == kme) | (k4.w == kme);
        }
        r += (r0 + r1) + (r2 + r3);
        if (ties) {                                                     // the same x: other orientations of one extremum, a repeat, (rarely) a neighbour -- or just i itself
            for (int q = 0; q < m; q++) {
                if (s_key[q] != kme || jb + q == i) continue;
                const uvo_keypoint o = s_kp[q];
                const bool before = sift_kp_less(o, me) || (!sift_kp_less(me, o) && jb + q < i);
                r += before;
                if (before && o.size == me.size && o.angle == me.angle) d = 1;
            }
        }
    }
    if (i < n) { if (r) atomicAdd(&rank[i], r); if (d) atomicOr(&dup[i], 1); }
}
// sorted[rank] = raw; then the keypoints that do not repeat their predecessor, packed in order: kept[], their count to cnt[3]
__global__ __launch_bounds__(256) void k_sift_place(const uvo_keypoint* __restrict__ raw, const int* __restrict__ cnt_p, int cap, const int* __restrict__ rank,
                                                    const int* __restrict__ dup, uvo_keypoint* sorted, int* keep)
{
    const int n = min(*cnt_p, cap), i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = rank[i];
    sorted[r] = raw[i]; keep[r] = !dup[i];
}
// order-preserving compaction by one workgroup: 16 waves, a contiguous segment each; a wave walks its segment 64 records at a time
// (coalesced) and places the kept ones with ballot / mbcnt, once to count and once to write
__global__ __launch_bounds__(1024) void k_sift_pack(const uvo_keypoint* __restrict__ src, const int* __restrict__ n_p, int cap, const int* __restrict__ keep,
                                                    const int* __restrict__ greater, int nfeatures, int scale_back, uvo_keypoint* dst, int dst_cap, int* n_out)
{
    __shared__ int s_cnt[16];
    const int n = min(*n_p, cap), tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int seg = ((n + 15) / 16 + 63) & ~63, e0 = wv * seg, e1 = min(n, e0 + seg);
    const bool cut = greater && nfeatures > 0 && n > nfeatures;          // retainBest is a no-op otherwise
    int total = 0;
    for (int e = e0 + lane; e - lane < e1; e += 64) {
        const bool k = e < e1 && (cut ? (greater[e] < nfeatures) : (keep ? keep[e] != 0 : true));
        total += __popcll(__ballot(k));
    }
    if (lane == 0) s_cnt[wv] = total;
    __syncthreads();
    int pos = 0, all = 0;
    for (int q = 0; q < 16; q++) { const int v = s_cnt[q]; pos += q < wv ? v : 0; all += v; }
    for (int e = e0 + lane; e - lane < e1; e += 64) {
        const bool k = e < e1 && (cut ? (greater[e] < nfeatures) : (keep ? keep[e] != 0 : true));
        const unsigned long long m = __ballot(k);
        if (k) {
            uvo_keypoint kp = src[e];
            if (scale_back) { kp.octave = (kp.octave & ~255) | ((kp.octave + -1) & 255); kp.x *= 0.5f; kp.y *= 0.5f; kp.size *= 0.5f; }
            const int at = pos + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (at < dst_cap) dst[at] = kp;
        }
        pos += __popcll(m);
    }
    if (tid == 0) *n_out = all;
}
// greater[i] = the number of kept keypoints with a larger response (retainBest keeps i when that is below nfeatures)
__global__ __launch_bounds__(256) void k_sift_greater(const uvo_keypoint* __restrict__ kept, const int* __restrict__ m_p, int cap, int nfeatures, int* greater)
{
    __shared__ float s_r[256];
    const int m = min(*m_p, cap), tid = threadIdx.x, i = blockIdx.x * 256 + tid;
    if (nfeatures <= 0 || m <= nfeatures || (int)blockIdx.x * 256 >= m) return;
    const int chunk = (m + gridDim.y - 1) / gridDim.y, j0 = blockIdx.y * chunk, j1 = min(m, j0 + chunk);
    const float me = kept[min(i, m - 1)].response;
    int g = 0;
    for (int jb = j0; jb < j1; jb += 256) {
        __syncthreads();
        if (jb + tid < j1) s_r[tid] = kept[jb + tid].response;
        __syncthreads();
        const int mm = min(256, j1 - jb);
        for (int q = 0; q < mm; q++) g += s_r[q] > me;
    }
    if (i < m && g) atomicAdd(&greater[i], g);
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
template <int R>
static void sift_blur_tile_launch(hipStream_t st, const float* src, float* dst, float* dog, int w, int h, const SiftTaps& t)
{
    hipLaunchKernelGGL((k_sift_blur_tile<R, 32>), dim3((w + 63) / 64, (h + 31) / 32), dim3(256), 0, st, src, dst, dog, w, h, t);
}
// dst = GaussianBlur(src, sigma); dog (may be null) = dst - src
static uvo_status sift_blur(Ctx* c, SiftWs* s, hipStream_t st, const float* src, float* dst, float* dog, int w, int h, double sigma)
{
    SiftTaps t;
    memset(&t, 0, sizeof(t));
    const int n = sift_gauss_kernel(sigma, t.k);
    t.r = n / 2;
    switch (t.r) {
#define UVO_SIFT_BLUR_CASE(R) case R: sift_blur_tile_launch<R>(st, src, dst, dog, w, h, t); break;
        UVO_SIFT_BLUR_CASE(2) UVO_SIFT_BLUR_CASE(3) UVO_SIFT_BLUR_CASE(4) UVO_SIFT_BLUR_CASE(5) UVO_SIFT_BLUR_CASE(6) UVO_SIFT_BLUR_CASE(7) UVO_SIFT_BLUR_CASE(8)
        UVO_SIFT_BLUR_CASE(9) UVO_SIFT_BLUR_CASE(10) UVO_SIFT_BLUR_CASE(11) UVO_SIFT_BLUR_CASE(12) UVO_SIFT_BLUR_CASE(13) UVO_SIFT_BLUR_CASE(14) UVO_SIFT_BLUR_CASE(15)
        UVO_SIFT_BLUR_CASE(16)
#undef UVO_SIFT_BLUR_CASE
        default: {                                                    // other radii (non-default sigma / layer counts): the two plain passes
            dim3 grid((w + 255) / 256, h);
            hipLaunchKernelGGL(k_sift_blur, grid, dim3(256), 0, st, src, s->tmp, w, h, t, 0);
            hipLaunchKernelGGL(k_sift_blur, grid, dim3(256), 0, st, static_cast<const float*>(s->tmp), dst, w, h, t, 1);
            const size_t npx = (size_t)w * h;
            if (dog) hipLaunchKernelGGL(k_sift_dog, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, st, src, static_cast<const float*>(dst), dog, npx);
        }
    }
    UVO_HIP_TRY(c, hipGetLastError());
    return UVO_OK;
}

// ---- the detector in three host steps: workspace, pyramid, keypoints.  Everything is queued on c->stream; nothing here waits for the
// device except the (re)allocation of a workspace and a change of the small octaves' taps.
static uvo_status sift_ensure(Ctx* c, int slot, int w, int h, int nL, SiftWs** out)
{
    if (!c->sift_ws[slot]) c->sift_ws[slot] = new SiftWs();
    SiftWs* s = static_cast<SiftWs*>(c->sift_ws[slot]);
    *out = s;
    if (s->w == w && s->h == h && s->nL == nL) return UVO_OK;
    const int nOct = std::min(kSiftMaxOctaves, std::max(1, cv_round_d(log((double)(2 * std::min(w, h))) / log(2.) - 2) + 1));    // firstOctave = -1
    UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    sift_ws_release(s);
    int ow = 2 * w, oh = 2 * h;
    bool ok = hipMalloc(reinterpret_cast<void**>(&s->tmp), sizeof(float) * (size_t)ow * oh) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&s->d_img), (size_t)w * h) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&s->d_cnt), sizeof(int) * 8) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&s->d_exptab), sizeof(float) * 64) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&s->d_taps), sizeof(float) * kSiftMaxTaps * (kSiftMaxLayers + 3)) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&s->d_radii), sizeof(int) * (kSiftMaxLayers + 3)) == hipSuccess;
    for (int o = 0; o < nOct && ok; o++) {
        s->ow[o] = ow; s->oh[o] = oh;
        for (int i = 0; i < nL + 3 && ok; i++) ok = hipMalloc(reinterpret_cast<void**>(&s->gauss[o * (nL + 3) + i]), sizeof(float) * (size_t)ow * oh) == hipSuccess;
        for (int i = 0; i < nL + 2 && ok; i++) ok = hipMalloc(reinterpret_cast<void**>(&s->dog[o * (nL + 2) + i]), sizeof(float) * (size_t)ow * oh) == hipSuccess;
        ow /= 2; oh /= 2;
        if (ow < 1 || oh < 1) { ok = ok && o + 1 >= nOct; }
    }
    ok = ok && sift_grow(s, 8 * c->cap, 4 * c->cap);
    if (!ok) { sift_ws_release(s); c->err = "uvo_sift_detect: out of device memory for the scale-space pyramid"; return UVO_HIP_ERROR; }
    float tab[64];
    for (int i = 0; i < 64; i++) tab[i] = (float)(pow(2.0, (double)i / 64) * .9670371139572337719125840413672004409288e-2);     // hal::exp32f's table
    UVO_HIP_TRY(c, hipMemcpy(s->d_exptab, tab, sizeof(tab), hipMemcpyHostToDevice));
    s->w = w; s->h = h; s->nL = nL; s->nOct = nOct;
    return UVO_OK;
}

// the layers' filter taps in device memory, for k_sift_tail (they change with sigma and the layer count only)
static uvo_status sift_upload_taps(Ctx* c, SiftWs* s, double sigma)
{
    const int nL = s->nL;
    if (s->taps_sigma == sigma && s->taps_nL == nL) return UVO_OK;
    float taps[kSiftMaxTaps * (kSiftMaxLayers + 3)]; int radii[kSiftMaxLayers + 3];
    memset(taps, 0, sizeof(taps)); memset(radii, 0, sizeof(radii));
    const double k = pow(2., 1. / nL);
    for (int i = 1; i < nL + 3; i++) {
        const double sp = pow(k, (double)(i - 1)) * sigma, stt = sp * k;
        radii[i] = sift_gauss_kernel(sqrt(stt * stt - sp * sp), taps + i * kSiftMaxTaps) / 2;
    }
    UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));                   // (a launch of the previous call may still read the old taps)
    UVO_HIP_TRY(c, hipMemcpy(s->d_taps, taps, sizeof(taps), hipMemcpyHostToDevice));
    UVO_HIP_TRY(c, hipMemcpy(s->d_radii, radii, sizeof(radii), hipMemcpyHostToDevice));
    s->taps_sigma = sigma; s->taps_nL = nL;
    return UVO_OK;
}

// createInitialImage + buildGaussianPyramid + buildDoGPyramid of the tight device image d_img
static uvo_status sift_pyramid(Ctx* c, SiftWs* s, hipStream_t st, const uint8_t* d_img, double sigma, SiftPyr* pp)
{
    const int w = s->w, h = s->h, nL = s->nL, nOct = s->nOct;
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
    SiftPyr& p = *pp;
    memset(&p, 0, sizeof(p));
    for (int i = 0; i < nOct * (nL + 3); i++) p.gauss[i] = s->gauss[i];
    for (int i = 0; i < nOct * (nL + 2); i++) p.dog[i] = s->dog[i];
    for (int o = 0; o < nOct; o++) { p.ow[o] = s->ow[o]; p.oh[o] = s->oh[o]; }
    p.nL = nL;
    int o_tail = nOct;                                                  // the first octave of k_sift_tail's range
    for (int o = nOct - 1; o >= 1 && (size_t)s->ow[o] * s->oh[o] <= 2048; o--) o_tail = o;
    if (o_tail < nOct) UVO_TRY(sift_upload_taps(c, s, sigma));
    for (int o = 0; o < o_tail; o++) {
        const int ow = s->ow[o], oh = s->oh[o];
        for (int i = 0; i < nL + 3; i++) {
            float* dst = s->gauss[o * (nL + 3) + i];
            if (o == 0 && i == 0) UVO_TRY(sift_blur(c, s, st, base, dst, nullptr, ow, oh, (double)sig_diff));
            else if (i == 0) hipLaunchKernelGGL(k_sift_half, dim3((ow + 255) / 256, oh), dim3(256), 0, st, static_cast<const float*>(s->gauss[(o - 1) * (nL + 3) + nL]), s->ow[o - 1], dst, ow, oh);
            else UVO_TRY(sift_blur(c, s, st, s->gauss[o * (nL + 3) + i - 1], dst, s->dog[o * (nL + 2) + i - 1], ow, oh, sig[i]));     // + buildDoGPyramid's layer i - 1
        }
    }
    if (o_tail < nOct)
        hipLaunchKernelGGL(k_sift_tail, dim3(1), dim3(1024), 0, st, p, static_cast<const float*>(s->d_taps), static_cast<const int*>(s->d_radii), o_tail, nOct);
    UVO_HIP_TRY(c, hipGetLastError());
    return UVO_OK;
}

// findScaleSpaceExtrema, KeyPointsFilter, calcDescriptors: out_kps / out_desc (device, out_cap rows; out_desc may be null) and the
// count *out_n (device; it may exceed out_cap, nothing is written past it).  s->d_cnt[0..2] keep the list lengths for the caller's
// overflow check (every kernel clamps to the capacities).
static uvo_status sift_keypoints(Ctx* c, SiftWs* s, hipStream_t st, const SiftPyr& p, int nfeatures, double contrastThreshold, double edgeThreshold, double sigma,
                                 uvo_keypoint* out_kps, float* out_desc, int out_cap, int* out_n)
{
    const int nL = s->nL, nOct = s->nOct;
    const int threshold = cv_floor_d(0.5 * contrastThreshold / nL * 255);
    SiftTiles tl;
    memset(&tl, 0, sizeof(tl));
    tl.nOct = nOct;
    for (int o = 0; o < nOct; o++) {
        const int ow = s->ow[o] - 2 * SIFT_IMG_BORDER, oh = s->oh[o] - 2 * SIFT_IMG_BORDER;
        tl.tx[o] = ow > 0 ? (ow + 63) / 64 : 0;
        tl.start[o + 1] = tl.start[o] + (ow > 0 && oh > 0 ? tl.tx[o] * ((oh + 15) / 16) : 0);
    }
    const int rc = s->raw_cap;
    int* d_rank = s->d_ints; int* d_dup = s->d_ints + rc; int* d_keep = s->d_ints + 2 * (size_t)rc; int* d_greater = s->d_ints + 3 * (size_t)rc;
    UVO_HIP_TRY(c, hipMemsetAsync(s->d_cnt, 0, sizeof(int) * 8, st));
    UVO_HIP_TRY(c, hipMemsetAsync(s->d_ints, 0, sizeof(int) * 4 * (size_t)rc, st));
    if (tl.start[nOct] > 0)
        hipLaunchKernelGGL(k_sift_extrema_all, dim3(tl.start[nOct]), dim3(256), 0, st, p, tl, threshold, s->d_cand, s->d_cnt, s->cand_cap);
    hipLaunchKernelGGL(k_sift_refine, dim3((s->cand_cap + 63) / 64), dim3(64), 0, st, p, static_cast<const SiftCand*>(s->d_cand), static_cast<const int*>(s->d_cnt),
                       s->cand_cap, (float)contrastThreshold, (float)edgeThreshold, (float)sigma, s->d_surv, s->d_cnt + 2);
    hipLaunchKernelGGL(k_sift_orient, dim3(8192), dim3(64), 0, st, p, static_cast<const SiftSurv*>(s->d_surv), static_cast<const int*>(s->d_cnt + 2), s->cand_cap,
                       static_cast<const float*>(s->d_exptab), s->d_raw, s->d_cnt + 1, rc);
    // KeyPointsFilter::removeDuplicatedSorted, retainBest(nfeatures), the scaling back of firstOctave = -1
    hipLaunchKernelGGL(k_sift_rank, dim3(rc / 256, 64), dim3(256), 0, st, static_cast<const uvo_keypoint*>(s->d_raw), static_cast<const int*>(s->d_cnt + 1), rc, d_rank, d_dup);
    hipLaunchKernelGGL(k_sift_place, dim3(rc / 256), dim3(256), 0, st, static_cast<const uvo_keypoint*>(s->d_raw), static_cast<const int*>(s->d_cnt + 1), rc,
                       static_cast<const int*>(d_rank), static_cast<const int*>(d_dup), s->d_sorted, d_keep);
    hipLaunchKernelGGL(k_sift_pack, dim3(1), dim3(1024), 0, st, static_cast<const uvo_keypoint*>(s->d_sorted), static_cast<const int*>(s->d_cnt + 1), rc,
                       static_cast<const int*>(d_keep), static_cast<const int*>(nullptr), 0, 0, s->d_kept, rc, s->d_cnt + 3);
    hipLaunchKernelGGL(k_sift_greater, dim3(rc / 256, 64), dim3(256), 0, st, static_cast<const uvo_keypoint*>(s->d_kept), static_cast<const int*>(s->d_cnt + 3), rc, nfeatures, d_greater);
    hipLaunchKernelGGL(k_sift_pack, dim3(1), dim3(1024), 0, st, static_cast<const uvo_keypoint*>(s->d_kept), static_cast<const int*>(s->d_cnt + 3), rc,
                       static_cast<const int*>(nullptr), static_cast<const int*>(d_greater), nfeatures, 1, out_kps, out_cap, out_n);
    if (out_desc)
        hipLaunchKernelGGL(k_sift_descriptor, dim3(16384), dim3(64), 0, st, p, static_cast<const uvo_keypoint*>(out_kps), static_cast<const int*>(out_n), out_cap,
                           static_cast<const float*>(s->d_exptab), out_desc);
    UVO_HIP_TRY(c, hipGetLastError());
    return UVO_OK;
}

uvo_status sift_detect(Ctx* c, const uint8_t* gray, int w, int h, int stride, int mem, int nfeatures, int nL, double contrastThreshold,
                       double edgeThreshold, double sigma, uvo_keypoint* kps, float* desc, int cap, int* n_out)
{
    if (nL < 1 || nL > kSiftMaxLayers || w < 16 || h < 16 || w > c->max_w || h > c->max_h || stride < w || sigma <= 0.5) { c->err = "uvo_sift_detect: nOctaveLayers 1..8, sigma > 0.5, image within the context's size, stride >= width"; return UVO_INVALID_ARG; }
    {   // the widest blur (layer nOctaveLayers + 2) must fit the tap arrays: GaussianBlur takes cvRound(8 sigma_i + 1) | 1 taps, 63 are held
        const double k = pow(2., 1. / nL), sp = pow(k, (double)(nL + 1)) * sigma, stt = sp * k;
        if ((cv_round_d(sqrt(stt * stt - sp * sp) * 8 + 1) | 1) > kSiftMaxTaps - 1) { c->err = "uvo_sift_detect: sigma / nOctaveLayers need a blur of more than 63 taps"; return UVO_INVALID_ARG; }
    }
    SiftWs* s = nullptr;
    UVO_TRY(sift_ensure(c, 0, w, h, nL, &s));
    hipStream_t st = c->stream;
    const uint8_t* d_img = gray;
    if (!(mem == UVO_MEM_DEVICE && stride == w)) {
        UVO_HIP_TRY(c, hipMemcpy2DAsync(s->d_img, w, gray, stride, w, h, mem == UVO_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        d_img = s->d_img;
    }
    SiftPyr p;
    UVO_TRY(sift_pyramid(c, s, st, d_img, sigma, &p));
    int cnt[8];
    for (int attempt = 0;; attempt++) {
        UVO_TRY(sift_keypoints(c, s, st, p, nfeatures, contrastThreshold, edgeThreshold, sigma, s->d_kps, desc ? s->d_desc : nullptr, s->raw_cap, s->d_cnt + 4));
        UVO_HIP_TRY(c, hipMemcpyAsync(cnt, s->d_cnt, sizeof(cnt), hipMemcpyDeviceToHost, st));
        UVO_HIP_TRY(c, hipStreamSynchronize(st));
        if (cnt[0] <= s->cand_cap && cnt[1] <= s->raw_cap) break;
        // a list overflowed (its counter still counted everything it saw; every kernel clamped to the capacity): the extrema stage runs
        // again with room for it, unless the candidate list was cut -- then the keypoint count is a lower bound and another pass may follow
        if (attempt >= 3 || !sift_grow(s, cnt[0] + cnt[0] / 4, std::max(cnt[1] + cnt[1] / 4, cnt[0] > s->cand_cap ? cnt[0] : 0))) {
            c->err = "uvo_sift_detect: out of device memory for the extrema lists"; return UVO_HIP_ERROR;
        }
    }
    const int nk = cnt[4];
    *n_out = nk;
    if ((kps || desc) && nk > cap) { c->err = "uvo_sift_detect: output capacity too small"; return UVO_CAPACITY; }
    if (kps && nk) UVO_HIP_TRY(c, hipMemcpyAsync(kps, s->d_kps, sizeof(uvo_keypoint) * (size_t)nk, hipMemcpyDeviceToHost, st));
    if (desc && nk) UVO_HIP_TRY(c, hipMemcpyAsync(desc, s->d_desc, sizeof(float) * 128 * (size_t)nk, hipMemcpyDeviceToHost, st));
    UVO_HIP_TRY(c, hipStreamSynchronize(st));
    return UVO_OK;
}

// ---- detect_features' SIFT branch inside the fused stereo / mono steps (the context's feature detector is "SIFT"): the images are
// c->img[0 .. nimg-1] (surf_upload), the results land where the SURF detector puts its own -- c->det[i].kps / desc (128 floats per
// row), the counts in d_counts[CN_NL + i] -- and nothing waits for the device: the lists are sized once (16 / 4 times max_kpts); a
// frame that overflows them, or yields more keypoints than max_kpts, is reported through d_counts[CN_CAND0 + i] > max_kpts, the
// signal the SURF path uses for the same condition.
__global__ void k_sift_publish(const int* __restrict__ cnt0, int cand_cap0, int raw_cap0, const int* __restrict__ cnt1, int cand_cap1, int raw_cap1,
                               int* cn, int cap, int nimg, int gate_min_features)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < nimg; i++) {
        const int* cnt = i ? cnt1 : cnt0;
        const bool over = cnt[0] > (i ? cand_cap1 : cand_cap0) || cnt[1] > (i ? raw_cap1 : raw_cap0);
        const int n = cn[CN_NL + i];
        cn[CN_CAND0 + i] = over ? cap + 1 : n;
        cn[CN_NL + i] = n < cap ? n : cap;
    }
    if (gate_min_features >= 0 && nimg == 2)                           // VO:556: both images need >= MIN_NUM_FEATURES keypoints, else no stereo matching
        cn[CN_NQA] = (cn[CN_NL] >= gate_min_features && cn[CN_NR] >= gate_min_features) ? cn[CN_NL] : 0;
}
// the lane's workspaces and lists at their pipeline sizes (allocation synchronises the device: done when a sequence starts, not in it)
uvo_status sift_prepare_lane(Ctx* c, int w, int h, int nimg)
{
    for (int i = 0; i < nimg; i++) {
        SiftWs* s = nullptr;
        UVO_TRY(sift_ensure(c, i, w, h, 3, &s));
        if (s->cand_cap < 16 * c->cap || s->raw_cap < 4 * c->cap) {
            UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (!sift_grow(s, 16 * c->cap, 4 * c->cap)) { c->err = "SIFT: out of device memory for the extrema lists"; return UVO_HIP_ERROR; }
        }
        UVO_TRY(sift_upload_taps(c, s, 1.6));
    }
    return UVO_OK;
}
uvo_status sift_detect_lane(Ctx* c, int nimg, int gate_min_features)
{
    const int w = c->img_w, h = c->img_h;
    if (w < 16 || h < 16) { c->err = "SIFT: image too small"; return UVO_INVALID_ARG; }
    UVO_TRY(sift_prepare_lane(c, w, h, nimg));                          // (a no-op once the lane is primed)
    SiftWs* ws[2] = { static_cast<SiftWs*>(c->sift_ws[0]), static_cast<SiftWs*>(c->sift_ws[1]) };
    // One pair in flight (the synchronous step): the right image's detector runs beside the left one's on the lane's second stream, the
    // latency-bound stretches of one under the chip-filling ones of the other (two events: fork after the uploads, join before the
    // counters are published).  With several pairs in flight the lanes overlap each other and a queue parked on an event wait costs
    // the others, so everything stays on the lane's stream.
    const Ctx* m = c->master ? c->master : c;
    const bool fork = nimg == 2 && m->in_sync_step && c->pnp_stream != nullptr && !c->timing;
    hipStream_t streams[2] = { c->stream, fork ? c->pnp_stream : c->stream };
    if (fork) {
        SiftWs* f = ws[1];
        if (!f->ev_fork) { UVO_HIP_TRY(c, hipEventCreateWithFlags(&f->ev_fork, hipEventDisableTiming)); UVO_HIP_TRY(c, hipEventCreateWithFlags(&f->ev_join, hipEventDisableTiming)); }
        UVO_HIP_TRY(c, hipEventRecord(f->ev_fork, c->stream));
        UVO_HIP_TRY(c, hipStreamWaitEvent(streams[1], f->ev_fork, 0));
    }
    for (int i = 0; i < nimg; i++) {
        SiftPyr p;
        UVO_TRY(sift_pyramid(c, ws[i], streams[i], c->img[i], 1.6, &p));
        UVO_TRY(sift_keypoints(c, ws[i], streams[i], p, 10000, 0.03, 10, 1.6, c->det[i].kps, c->det[i].desc, c->cap, c->d_counts + CN_NL + i));   // VOU:109
    }
    if (fork) {
        UVO_HIP_TRY(c, hipEventRecord(ws[1]->ev_join, streams[1]));
        UVO_HIP_TRY(c, hipStreamWaitEvent(c->stream, ws[1]->ev_join, 0));
    }
    SiftWs* s1 = ws[nimg > 1 ? 1 : 0];
    hipLaunchKernelGGL(k_sift_publish, dim3(1), dim3(64), 0, c->stream, static_cast<const int*>(ws[0]->d_cnt), ws[0]->cand_cap, ws[0]->raw_cap,
                       static_cast<const int*>(s1->d_cnt), s1->cand_cap, s1->raw_cap, c->d_counts, c->cap, nimg, gate_min_features);
    UVO_HIP_TRY(c, hipGetLastError());
    return UVO_OK;
}

// test hook: one Gaussian (dog = 0) or difference-of-Gaussians (dog = 1) layer of the last uvo_sift_detect, to a host buffer
uvo_status sift_layer(Ctx* c, int octave, int layer, int dog, float* out, int cap_floats, int* ow, int* oh)
{
    SiftWs* s = static_cast<SiftWs*>(c->sift_ws[0]);
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
