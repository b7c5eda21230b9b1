// akaze.hip -- the AKAZE branch of detect_features on gfx950 (uvo_libraries/src/VO_utility.cpp:93-98):
//     Ptr<AKAZE> detector = AKAZE::create();  detector->detectAndCompute(img, noArray(), keypoints, descriptors);
// DESCRIPTOR_MLDB, full length (486 bits = 61 bytes), 3 channels, threshold 0.001f, 4 octaves x 4 sublevels, DIFF_PM_G2 -- after
// Alcantarilla, Nuevo, Bartoli, "Fast explicit diffusion for accelerated features in nonlinear scale spaces" (BMVC 2013) in the form
// of OpenCV 4.5's features2d/src/kaze/AKAZEFeatures.cpp as far as it can be recalled (PARITY UNPINNED; the float operation order is
// the scalar one of imgproc's separable filters).
//
// Everything that touches pixels runs on the device, one launch per pass (HBM-bound stencils over float planes, coalesced rows):
//   k_ak_u8_to_f32                 the image in [0, 1]
//   k_ak_blur_rows / _cols         GaussianBlur, BORDER_REPLICATE (9 taps for the base level, 5 for the smoothed copies)
//   k_ak_scharr                    Scharr 3 x 3 (for the conductance), BORDER_REFLECT_101
//   k_ak_gradmax / _gradhist       compute_kcontrast: maximum and 300-bin histogram of the gradient magnitude
//   k_ak_pm_g2                     Perona-Malik g2 conductance
//   k_ak_nld_step                  one FED step Lt + step * div(c grad Lt), five-point star, one-sided at the border
//   k_ak_half / k_ak_area          resize INTER_AREA to the next octave (exact halving, or the general table form for odd sizes)
//   k_ak_sep_deriv                 the stretched Scharr pair of compute_derivative_kernels (taps at +-sigma_size)
//   k_ak_det                       determinant of the Hessian x sigma_size^4
//   k_ak_candidates                strict 3 x 3 maxima above the threshold inside the level's border, with their neighbourhood
//   k_ak_orientation               Compute_Main_Orientation, one wave per keypoint (samples, counting sort and window sums in LDS)
//   k_ak_mldb                      the M-LDB descriptor, 32 lanes per keypoint (one per grid cell of the 2x2 + 3x3 + 4x4 grids)
// What OpenCV does SEQUENTIALLY -- the row-major scan that suppresses weaker maxima within sigma_size of a stronger one, the two
// sweeps across neighbouring scales (FindKeypointsSameScale, Find_Scale_Space_Extrema), the 2 x 2 sub-pixel solve -- runs on the host
// over the candidate list (a few thousand 40-byte records; the planes stay on the device), like the RANSAC scans of the pose stages.
#include "uvo_ctx.h"
#include "uvo_math.h"
#include <float.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>

namespace uvo {

double now_us();                                                  // pose.hip / ctx.hip: steady clock, microseconds
static const int kAkMaxLevels = 16, kAkDescBytes = 61, kAkCandCap = 1 << 18;
struct AkLevel { int w, h, octave, sublevel, sigma_size, border, nsteps; float esigma, etime, octave_ratio; float tau[64]; };
struct AkCand { int x, y; float v[9]; int pad; };             // a strict maximum and its 3 x 3 neighbourhood (row-major, v[4] = the maximum); pad = its evolution level
struct AkTab { int si; float alpha; };

struct AkazeWs {
    int w = 0, h = 0, n = 0, cap = 0;
    AkLevel lv[kAkMaxLevels];
    float* Lt[kAkMaxLevels] = {nullptr}; float* Lsmooth[kAkMaxLevels] = {nullptr}; float* Lx[kAkMaxLevels] = {nullptr};
    float* Ly[kAkMaxLevels] = {nullptr}; float* Ldet[kAkMaxLevels] = {nullptr};
    float* img = nullptr; float* s[5] = {nullptr};            // full-size scratch planes
    uint8_t* img8 = nullptr;
    int* hist = nullptr;                                      // [0] max bits, [1..300] histogram
    AkCand* cand = nullptr; int* cand_n = nullptr;            // candidates of one level (device), count
    AkCand* h_cand = nullptr; int* h_hist = nullptr;          // pinned
    uvo_keypoint* d_kps = nullptr; uint8_t* d_desc = nullptr; // outputs (cap)
    AkTab* xtab[kAkMaxLevels] = {nullptr}; AkTab* ytab[kAkMaxLevels] = {nullptr}; int* xofs[kAkMaxLevels] = {nullptr}; int* yofs[kAkMaxLevels] = {nullptr};   // general INTER_AREA tables of the octave changes that are not exact halvings (built with the workspace)
    float* d_kc = nullptr; float* h_kc = nullptr;             // compute_kcontrast's result: pinned source, device copy the conductance kernels read
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; // the launch chain from the first evolution step to the last level's candidates, captured once per image size
    std::vector<uint8_t> mask[kAkMaxLevels]; std::vector<float> val[kAkMaxLevels];               // host: keypoint mask and Ldet at candidates
    std::vector<std::vector<AkCand>> cands;
    std::vector<uint64_t> keys;                               // host: sort keys of the candidate list
};

// ---- fed.cpp: fed_tau_by_process_time(T, 1, 0.25f, true, tau) ----
static bool fed_is_prime(int number)
{
    if (number <= 1) return false;
    if (number == 1 || number == 2 || number == 3 || number == 5 || number == 7) return true;
    if ((number % 2) == 0 || (number % 3) == 0 || (number % 5) == 0 || (number % 7) == 0) return false;
    int upperLimit = (int)sqrt(1.0f + number), divisor = 11;
    while (divisor <= upperLimit) { if (number % divisor == 0) return false; divisor += 2; }
    return true;
}
static int fed_tau(float T, float tau_max, float* tau)
{
    const float t = T / 1.0f;
    const int n = cv_ceil_d((double)(sqrtf(3.0f * t / tau_max + 0.25f) - 0.5f - 1.0e-8f));
    const float scale = 3.0f * t / (tau_max * (float)(n * (n + 1)));
    if (n <= 0) return 0;
    float tauh[256];
    const float c = 1.0f / (4.0f * (float)n + 2.0f), d = scale * tau_max / 2.0f;
    for (int k = 0; k < n; ++k) { const float hh = cosf((float)3.14159265358979323846 * (2.0f * (float)k + 1.0f) * c); tauh[k] = d / (hh * hh); }
    const int kappa = n / 2;
    int prime = n + 1;
    while (!fed_is_prime(prime)) prime++;
    for (int k = 0, l = 0; l < n; ++k, ++l) {
        int index = 0;
        while ((index = ((k + 1) * kappa) % prime - 1) >= n) k++;
        tau[l] = tauh[index];
    }
    return n;
}
// Allocate_Memory_Evolution with AKAZE::create()'s options
static int akaze_plan(int img_w, int img_h, AkLevel* L)
{
    const int omax = 4, nsublevels = 4;
    const float soffset = 1.6f, derivative_factor = 1.5f, smax = 10.0f * sqrtf(2.0f);
    int n = 0;
    for (int i = 0, power = 1; i <= omax - 1; i++, power *= 2) {
        const float rfactor = 1.0f / power;
        const int level_height = (int)(img_h * rfactor), level_width = (int)(img_w * rfactor);
        if ((level_width < 80 || level_height < 40) && i != 0) break;
        for (int j = 0; j < nsublevels; j++) {
            AkLevel* s = &L[n++];
            s->w = level_width; s->h = level_height;
            s->esigma = soffset * powf(2.f, (float)(j) / (float)(nsublevels) + i);
            s->sigma_size = cv_round_f(s->esigma * derivative_factor / power);
            s->etime = 0.5f * (s->esigma * s->esigma);
            s->octave = i; s->sublevel = j; s->octave_ratio = (float)power;
            s->border = cv_round_f(smax * s->sigma_size) + 1;
            s->nsteps = 0;
        }
    }
    for (int i = 1; i < n; i++) L[i].nsteps = fed_tau(L[i].etime - L[i - 1].etime, 0.25f, L[i].tau);
    return n;
}

// ------------------------------------------------------------------------------------------ kernels
__device__ __forceinline__ int ak_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int ak_reflect101(int p, int n) { if (n == 1) return 0; while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; } return p; }

__global__ __launch_bounds__(256) void k_ak_u8_to_f32(const uint8_t* __restrict__ src, int stride, int w, int h, float* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x < w) dst[(size_t)y * w + x] = (float)((double)src[(size_t)y * stride + x] * (1.0 / 255.0));        // convertTo(CV_32F, 1 / 255.)
}
struct AkKernel { float k[16]; int ksize; };
// GaussianBlur rows, BORDER_REPLICATE: 5 taps -> SymmRowSmallFilter (centre, then the mirrored pairs), wider -> the generic RowFilter (left to right)
__global__ __launch_bounds__(256) void k_ak_blur_rows(const float* __restrict__ src, int w, int h, AkKernel kk, float* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const float* s = src + (size_t)y * w;
    const int r = kk.ksize / 2;
    float acc;
    if (kk.ksize <= 5) {
        acc = kk.k[r] * s[x];
        for (int i = 1; i <= r; i++) acc += kk.k[r + i] * (s[ak_clamp(x - i, 0, w - 1)] + s[ak_clamp(x + i, 0, w - 1)]);
    } else {
        acc = kk.k[0] * s[ak_clamp(x - r, 0, w - 1)];
        for (int t = 1; t < kk.ksize; t++) acc += kk.k[t] * s[ak_clamp(x - r + t, 0, w - 1)];
    }
    dst[(size_t)y * w + x] = acc;
}
// columns: SymmColumnFilter (centre, then pairs outward)
__global__ __launch_bounds__(256) void k_ak_blur_cols(const float* __restrict__ src, int w, int h, AkKernel kk, float* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int r = kk.ksize / 2;
    float acc = kk.k[r] * src[(size_t)y * w + x];
    for (int i = 1; i <= r; i++) acc += kk.k[r + i] * (src[(size_t)ak_clamp(y + i, 0, h - 1) * w + x] + src[(size_t)ak_clamp(y - i, 0, h - 1) * w + x]);
    dst[(size_t)y * w + x] = acc;
}
// Scharr(src, dst, CV_32F, dx, dy, 1, 0, BORDER_DEFAULT): [-1 0 1] along the derivative, [3 10 3] across it; the row pass of the three
// rows a column pass needs is evaluated in place (the same operations as a materialised row pass)
__global__ __launch_bounds__(256) void k_ak_scharr(const float* __restrict__ src, int w, int h, int xorder, float* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int xm = ak_reflect101(x - 1, w), xp = ak_reflect101(x + 1, w);
    float t[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const float* s = src + (size_t)ak_reflect101(y + q - 1, h) * w;
        const float a = s[xm], b = s[x], c = s[xp];
        t[q] = xorder ? c - a : b * 10.f + (a + c) * 3.f;
    }
    dst[(size_t)y * w + x] = xorder ? (t[0] + t[2]) * 3.f + t[1] * 10.f : t[2] - t[0];
}
// compute_derivative_kernels + sepFilter2D: taps at -r, 0, +r (r = sigma_size; 3 + 2 (r - 1) taps of which three are not zero)
__global__ __launch_bounds__(256) void k_ak_sep_deriv(const float* __restrict__ src, int w, int h, int xorder, int r, float* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int ksize = 3 + 2 * (r - 1);
    const float ww = 10.0f / 3.0f, nrm = 1.0f / (2.0f * r * (ww + 2.0f)), wn = ww * nrm;
    const int xm = ak_reflect101(x - r, w), xp = ak_reflect101(x + r, w);
    float t[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const float* s = src + (size_t)ak_reflect101(y + (q - 1) * r, h) * w;
        const float a = s[xm], b = s[x], c = s[xp];
        float v;
        if (xorder) v = c - a;
        else if (ksize <= 5) v = wn * b + nrm * (a + c);
        else { v = nrm * a; v += wn * b; v += nrm * c; }
        t[q] = v;
    }
    dst[(size_t)y * w + x] = xorder ? wn * t[1] + nrm * (t[2] + t[0]) : t[2] - t[0];
}
// compute_kcontrast, pass 1: the largest gradient magnitude of the interior (non-negative floats order as their bit patterns)
__global__ __launch_bounds__(256) void k_ak_gradmax(const float* __restrict__ lx, const float* __restrict__ ly, int w, int h, int* __restrict__ hist)
{
    __shared__ float s_m[4];
    const int x = 1 + blockIdx.x * 256 + threadIdx.x;
    float d = 0.f;
    if (x < w - 1)
        for (int y = 1 + blockIdx.y * 16; y < min(h - 1, 1 + (blockIdx.y + 1) * 16); y++) {
            const float a = lx[(size_t)y * w + x], b = ly[(size_t)y * w + x];
            d = fmaxf(d, sqrtf(a * a + b * b));
        }
    for (int o = 32; o > 0; o >>= 1) d = fmaxf(d, __shfl_down(d, o));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) { d = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3])); if (d > 0.f) atomicMax(&hist[0], __float_as_int(d)); }      // (one atomic per workgroup: 32 000 waves on one address took 206 us)
}
// pass 2: bins (int)(modg * ((nbins - 1) / hmax))
__global__ __launch_bounds__(256) void k_ak_gradhist(const float* __restrict__ lx, const float* __restrict__ ly, int w, int h, int nbins, int* __restrict__ hist)
{
    __shared__ int s_h[512];
    for (int i = threadIdx.x; i < nbins; i += 256) s_h[i] = 0;
    __syncthreads();
    const float hmax = __int_as_float(hist[0]);
    const float sc = (nbins - 1) / hmax;
    const int y = 1 + blockIdx.y;
    for (int x = 1 + blockIdx.x * 1024 + threadIdx.x; x < min(w - 1, 1 + (blockIdx.x + 1) * 1024); x += 256) {
        const float a = lx[(size_t)y * w + x], b = ly[(size_t)y * w + x];
        const float d = sqrtf(a * a + b * b);
        atomicAdd(&s_h[(int)(d * sc)], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbins; i += 256) if (s_h[i]) atomicAdd(&hist[1 + i], s_h[i]);
}
// (kcontrast of the frame from device memory -- the launch is part of a captured graph --, x 0.75f once per octave change as
// Create_Nonlinear_Scale_Space does)
__global__ __launch_bounds__(256) void k_ak_pm_g2(const float* __restrict__ lx, const float* __restrict__ ly, int n, const float* __restrict__ kc, int octave, float* __restrict__ dst)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float k = *kc;
    for (int o = 0; o < octave; o++) k *= 0.75f;
    const float k2inv = 1.0f / (k * k);
    dst[i] = 1.0f / (1.0f + ((lx[i] * lx[i] + ly[i] * ly[i]) * k2inv));
}
// non_linear_diffusion_step + add: out = Lt + step * div(c grad Lt).  Interior: the five-point star; image border: the one-sided stencil;
// the four corners get a zero step.
__global__ __launch_bounds__(256) void k_ak_nld_step(const float* __restrict__ lt, const float* __restrict__ lf, int w, int h, float step_size, float* __restrict__ out)
{
    const int j = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    if (j >= w) return;
    const float* lt_c = lt + (size_t)row * w; const float* lf_c = lf + (size_t)row * w;
    const float c = lt_c[j], f = lf_c[j];
    float step_r;
    const bool top = row == 0, bot = row == h - 1, left = j == 0, right = j == w - 1;
    if ((top || bot) && (left || right)) step_r = 0.0f;
    else if (top || bot) {
        const float* lt_o = top ? lt_c + w : lt_c - w; const float* lf_o = top ? lf_c + w : lf_c - w;
        step_r = (f + lf_c[j + 1]) * (lt_c[j + 1] - c) + (f + lf_c[j - 1]) * (lt_c[j - 1] - c) + (f + lf_o[j]) * (lt_o[j] - c);
    } else if (left) step_r = (f + lf_c[1]) * (lt_c[1] - c) + (f + lf_c[w]) * (lt_c[w] - c) + (f + (lf_c - w)[0]) * ((lt_c - w)[0] - c);
    else if (right) step_r = (f + lf_c[j - 1]) * (lt_c[j - 1] - c) + (f + lf_c[j + w]) * (lt_c[j + w] - c) + (f + lf_c[j - w]) * (lt_c[j - w] - c);
    else step_r = (f + lf_c[j + 1]) * (lt_c[j + 1] - c) + (f + lf_c[j - 1]) * (lt_c[j - 1] - c) + (f + lf_c[j + w]) * (lt_c[j + w] - c) + (f + lf_c[j - w]) * (lt_c[j - w] - c);
    out[(size_t)row * w + j] = c + step_r * step_size;
}
// resize INTER_AREA by exactly two (resizeAreaFast_): (S00 + S01 + S10 + S11) * 0.25f
__global__ __launch_bounds__(256) void k_ak_half(const float* __restrict__ src, int sw, int dw, int dh, float* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    const float* S = src + (size_t)(2 * y) * sw + 2 * x;
    float sum = 0;
    sum += S[0]; sum += S[1]; sum += S[sw]; sum += S[sw + 1];
    dst[(size_t)y * dw + x] = sum * 0.25f;
}
// the general form (resizeArea_): per destination pixel the rows of its y entries in table order, each the sum of its x entries in table order
__global__ __launch_bounds__(256) void k_ak_area(const float* __restrict__ src, int sw, int dw, int dh, const AkTab* __restrict__ xtab, const int* __restrict__ xofs,
                                                 const AkTab* __restrict__ ytab, const int* __restrict__ yofs, float* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    float sum = 0.f;
    for (int j = yofs[y]; j < yofs[y + 1]; j++) {
        const float* S = src + (size_t)ytab[j].si * sw;
        float buf = 0.f;
        for (int k = xofs[x]; k < xofs[x + 1]; k++) buf += S[xtab[k].si] * xtab[k].alpha;
        if (j == yofs[y]) sum = ytab[j].alpha * buf; else sum += ytab[j].alpha * buf;
    }
    dst[(size_t)y * dw + x] = sum;
}
__global__ __launch_bounds__(256) void k_ak_det(const float* __restrict__ lxx, const float* __restrict__ lxy, const float* __restrict__ lyy, int n, float sig4, float* __restrict__ dst)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (lxx[i] * lyy[i] - lxy[i] * lxy[i]) * sig4;
}
// FindKeypointsSameScale's parallel half: strict maxima of the 3 x 3 neighbourhood above the threshold inside [border, size - border)
__global__ __launch_bounds__(256) void k_ak_candidates(const float* __restrict__ ldet, int w, int h, int border, float thr, int level, AkCand* __restrict__ out, int* __restrict__ count, int cap)
{
    const int x = border + blockIdx.x * 256 + threadIdx.x, y = border + blockIdx.y;
    if (x >= w - border || y >= h - border) return;
    const float* curr = ldet + (size_t)y * w; const float* prev = curr - w; const float* next = curr + w;
    const float value = curr[x];
    if (value <= thr) return;
    if (value <= curr[x - 1] || value <= curr[x + 1]) return;
    if (value <= prev[x - 1] || value <= prev[x] || value <= prev[x + 1]) return;
    if (value <= next[x - 1] || value <= next[x] || value <= next[x + 1]) return;
    const int slot = atomicAdd(count, 1);
    if (slot >= cap) return;
    AkCand c;
    c.x = x; c.y = y; c.pad = level;
    c.v[0] = prev[x - 1]; c.v[1] = prev[x]; c.v[2] = prev[x + 1]; c.v[3] = curr[x - 1]; c.v[4] = value; c.v[5] = curr[x + 1];
    c.v[6] = next[x - 1]; c.v[7] = next[x]; c.v[8] = next[x + 1];
    out[slot] = c;
}

__device__ __forceinline__ float ak_atan2_deg(float y, float x)      // cv::fastAtan2 (as surf.hip's fast_atan2_deg)
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
__device__ const float kGauss25[7][7] = {
    { 0.02546481f, 0.02350698f, 0.01849125f, 0.01239505f, 0.00708017f, 0.00344629f, 0.00142946f },
    { 0.02350698f, 0.02169968f, 0.01706957f, 0.01144208f, 0.00653582f, 0.00318132f, 0.00131956f },
    { 0.01849125f, 0.01706957f, 0.01342740f, 0.00900066f, 0.00514126f, 0.00250252f, 0.00103800f },
    { 0.01239505f, 0.01144208f, 0.00900066f, 0.00603332f, 0.00344629f, 0.00167749f, 0.00069579f },
    { 0.00708017f, 0.00653582f, 0.00514126f, 0.00344629f, 0.00196855f, 0.00095820f, 0.00039744f },
    { 0.00344629f, 0.00318132f, 0.00250252f, 0.00167749f, 0.00095820f, 0.00046640f, 0.00019346f },
    { 0.00142946f, 0.00131956f, 0.00103800f, 0.00069579f, 0.00039744f, 0.00019346f, 0.00008024f } };
struct AkPlanes { const float* Lt[kAkMaxLevels]; const float* Lx[kAkMaxLevels]; const float* Ly[kAkMaxLevels]; int w[kAkMaxLevels], h[kAkMaxLevels]; float ratio[kAkMaxLevels]; };
// Compute_Main_Orientation: 109 Gaussian-weighted derivative samples within 6 scale units, sorted into 42 angular slices (counting
// sort), the 7-slice window with the largest summed vector.  ONE WAVE PER KEYPOINT, everything in LDS (a thread per keypoint kept its
// 109-entry arrays in scratch memory: 3.0 ms for 17 850 keypoints at 1080p): lanes take the samples two at a time; OpenCV's counting
// sort `ang_order[--slice[bin[i]]] = i` puts the members of a slice in DESCENDING sample order, so a sample's place is its slice's start
// plus the number of later samples of the same slice; lane sn sums window sn in that order (a sequential float sum each -- the windows
// are independent); "the first window with the largest norm" is an arg-max with ties to the lower window.  (Windows the reference
// skips because neither end moved hold the previous window's samples, hence its sums: they can never win the strict comparison.)
struct AkOriTab { signed char i[109], j[109]; };
constexpr AkOriTab ak_make_ori_tab()
{
    AkOriTab t{};
    int k = 0;
    for (int i = -6; i <= 6; ++i) for (int j = -6; j <= 6; ++j) if (i * i + j * j < 36) { t.i[k] = (signed char)i; t.j[k] = (signed char)j; ++k; }
    return t;
}
__device__ const AkOriTab kAkOriTab = ak_make_ori_tab();
__global__ __launch_bounds__(256) void k_ak_orientation(AkPlanes pl, uvo_keypoint* __restrict__ kps, int n)
{
    constexpr int ang_size = 109, slices = 42, win = 7;
    __shared__ float sX[4][ang_size + 3], sY[4][ang_size + 3];
    __shared__ unsigned char sBin[4][ang_size + 3], sOrd[4][ang_size + 3];
    __shared__ int sStart[4][slices + 2];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q_raw = blockIdx.x * 4 + wv;
    const bool live = q_raw < n;
    const int q = live ? q_raw : n - 1;                             // (a ragged tail recomputes the last keypoint and writes nothing)
    const uvo_keypoint kpt = kps[q];
    const int lv = kpt.class_id;
    const float* Lx = pl.Lx[lv]; const float* Ly = pl.Ly[lv];
    const int w = pl.w[lv], h = pl.h[lv];
    const float ratio = pl.ratio[lv];
    const int scale = cv_round_f(0.5f * kpt.size / ratio), x0 = cv_round_f(kpt.x / ratio), y0 = cv_round_f(kpt.y / ratio);
    const float ang_step = (float)(2.0 * 3.14159265358979323846 / slices);
    for (int k = lane; k < ang_size; k += 64) {
        const int i = kAkOriTab.i[k], j = kAkOriTab.j[k];
        const float wgt = kGauss25[i < 0 ? -i : i][j < 0 ? -j : j];
        const int y = ak_clamp(y0 + i * scale, 0, h - 1), x = ak_clamp(x0 + j * scale, 0, w - 1);
        const float rx = wgt * Lx[(size_t)y * w + x], ry = wgt * Ly[(size_t)y * w + x];
        const float ang = ak_atan2_deg(ry, rx) * (float)(3.14159265358979323846 / 180.0);
        int b = (int)(ang / ang_step);
        if (b < 0 || b >= slices) b = 0;
        sX[wv][k] = rx; sY[wv][k] = ry; sBin[wv][k] = (unsigned char)b;
    }
    __syncthreads();
    if (lane <= slices) {                                           // start of slice `lane` = the samples in lower slices; sStart[42] = 109
        int c = 0;
        for (int k = 0; k < ang_size; k++) c += sBin[wv][k] < lane;
        sStart[wv][lane] = c;
    }
    __syncthreads();
    for (int k = lane; k < ang_size; k += 64) {
        const int b = sBin[wv][k];
        int later = 0;
        for (int k2 = k + 1; k2 < ang_size; k2++) later += sBin[wv][k2] == b;
        sOrd[wv][sStart[wv][b] + later] = (unsigned char)k;
    }
    __syncthreads();
    float sumX = 0.0f, sumY = 0.0f, norm = -1.0f;
    if (lane < slices) {
        const int* st = sStart[wv];
        if (lane <= slices - win) { for (int i = st[lane]; i < st[lane + win]; i++) { const int idx = sOrd[wv][i]; sumX += sX[wv][idx]; sumY += sY[wv][idx]; } }
        else {
            const int remain = lane + win - slices;
            for (int i = st[lane]; i < st[slices]; i++) { const int idx = sOrd[wv][i]; sumX += sX[wv][idx]; sumY += sY[wv][idx]; }
            for (int i = st[0]; i < st[remain]; i++) { const int idx = sOrd[wv][i]; sumX += sX[wv][idx]; sumY += sY[wv][idx]; }
        }
        norm = sumX * sumX + sumY * sumY;
    }
    int best = lane;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float on = __shfl_xor(norm, d); const int ob = __shfl_xor(best, d);
        if (on > norm || (on == norm && ob < best)) { norm = on; best = ob; }
    }
    const float maxX = __shfl(sumX, best), maxY = __shfl(sumY, best);
    if (live && lane == 0) kps[q].angle = ak_atan2_deg(maxY, maxX);
}
// MLDB_Full_Descriptor_Invoker: grids of 2 x 2, 3 x 3 and 4 x 4 cells over the rotated 20 x 20-sample pattern (cell sides 10, 7, 5 samples),
// per cell the means of Lt and of the rotated derivatives; every pair of cells of a grid compared channel by channel: 3 (6 + 36 + 120) =
// 486 bits.  Lane c of the keypoint's 32 owns cell c (0..3 | 4..12 | 13..28) and sums its samples in OpenCV's order; lane 0 packs.
__global__ __launch_bounds__(256) void k_ak_mldb(AkPlanes pl, const uvo_keypoint* __restrict__ kps, int n, uint8_t* __restrict__ desc)
{
    __shared__ int s_val[8][29 * 3];
    const int g = threadIdx.x >> 5, lane = threadIdx.x & 31, q = blockIdx.x * 8 + g;
    if (q < n && lane < 29) {
        const uvo_keypoint kpt = kps[q];
        const int lv = kpt.class_id;
        const float* Lt = pl.Lt[lv]; const float* Lx = pl.Lx[lv]; const float* Ly = pl.Ly[lv];
        const int w = pl.w[lv], h = pl.h[lv];
        const float ratio = pl.ratio[lv];
        const float scale = (float)cv_round_f(0.5f * kpt.size / ratio);
        const float xf = kpt.x / ratio, yf = kpt.y / ratio;
        const float angle = (kpt.angle * (float)3.14159265358979323846) / 180.f;
        double sd, cd;
        det_sincos((double)angle, &sd, &cd);
        const float co = (float)cd, si = (float)sd;
        const int grid = lane < 4 ? 0 : (lane < 13 ? 1 : 2), cell = lane - (grid == 0 ? 0 : (grid == 1 ? 4 : 13));
        const int side = grid + 2, sample_step = grid == 0 ? 10 : (grid == 1 ? 7 : 5);          // ceil(10 * {1, 2/3, 1/2})
        const int i0 = -10 + (cell / side) * sample_step, j0 = -10 + (cell % side) * sample_step;
        float di = 0.0f, dx = 0.0f, dy = 0.0f;
        int nsamples = 0;
        for (int k = i0; k < i0 + sample_step; k++)
            for (int l = j0; l < j0 + sample_step; l++) {
                const float sample_y = yf + (l * co * scale + k * si * scale);
                const float sample_x = xf + (-l * si * scale + k * co * scale);
                const int y1 = ak_clamp(cv_round_f(sample_y), 0, h - 1), x1 = ak_clamp(cv_round_f(sample_x), 0, w - 1);
                const float ri = Lt[(size_t)y1 * w + x1];
                di += ri;
                const float rx = Lx[(size_t)y1 * w + x1], ry = Ly[(size_t)y1 * w + x1];
                const float rry = rx * co + ry * si, rrx = -rx * si + ry * co;
                dx += rrx; dy += rry;
                nsamples++;
            }
        di /= nsamples; dx /= nsamples; dy /= nsamples;
        const int a = __float_as_int(di), b = __float_as_int(dx), c = __float_as_int(dy);
        s_val[g][lane * 3] = a ^ ((a < 0) ? 0x7fffffff : 0);                                   // CV_TOGGLE_FLT
        s_val[g][lane * 3 + 1] = b ^ ((b < 0) ? 0x7fffffff : 0);
        s_val[g][lane * 3 + 2] = c ^ ((c < 0) ? 0x7fffffff : 0);
    }
    __syncthreads();
    if (q < n && lane == 0) {
        uint8_t d[kAkDescBytes];
        for (int i = 0; i < kAkDescBytes; i++) d[i] = 0;
        int dpos = 0;
        const int first[3] = { 0, 4, 13 };
        for (int lvl = 0; lvl < 3; lvl++) {
            const int val_count = (lvl + 2) * (lvl + 2);
            const int* iv = &s_val[g][first[lvl] * 3];
            for (int pos = 0; pos < 3; pos++)
                for (int i = 0; i < val_count; i++) {
                    const int ival = iv[3 * i + pos];
                    for (int j = i + 1; j < val_count; j++) { d[dpos >> 3] |= (uint8_t)((ival > iv[3 * j + pos]) << (dpos & 7)); dpos++; }
                }
        }
        for (int i = 0; i < kAkDescBytes; i++) desc[(size_t)q * kAkDescBytes + i] = d[i];
    }
}

// ------------------------------------------------------------------------------------------ host
void akaze_ws_free(Ctx* c)
{
    AkazeWs* s = static_cast<AkazeWs*>(c->akaze_ws);
    if (!s) return;
    for (int i = 0; i < kAkMaxLevels; i++) { (void)hipFree(s->Lt[i]); (void)hipFree(s->Lsmooth[i]); (void)hipFree(s->Lx[i]); (void)hipFree(s->Ly[i]); (void)hipFree(s->Ldet[i]); }
    for (float* p : s->s) (void)hipFree(p);
    (void)hipFree(s->img); (void)hipFree(s->img8); (void)hipFree(s->hist); (void)hipFree(s->cand); (void)hipFree(s->cand_n); (void)hipFree(s->d_kps); (void)hipFree(s->d_desc);
    for (int i = 0; i < kAkMaxLevels; i++) { (void)hipFree(s->xtab[i]); (void)hipFree(s->ytab[i]); (void)hipFree(s->xofs[i]); (void)hipFree(s->yofs[i]); }
    if (s->exec) (void)hipGraphExecDestroy(s->exec);
    if (s->graph) (void)hipGraphDestroy(s->graph);
    (void)hipFree(s->d_kc); (void)hipHostFree(s->h_kc);
    (void)hipHostFree(s->h_cand); (void)hipHostFree(s->h_hist);
    delete s;
    c->akaze_ws = nullptr;
}
static void area_tab(int ssize, int dsize, double scale, std::vector<AkTab>* tab, std::vector<int>* ofs);
static AkazeWs* akaze_ws(Ctx* c, int w, int h)
{
    AkazeWs* s = static_cast<AkazeWs*>(c->akaze_ws);
    if (s && s->w == w && s->h == h) return s;
    akaze_ws_free(c);
    s = new AkazeWs();
    c->akaze_ws = s;
    s->w = w; s->h = h; s->cap = c->cap;
    s->n = akaze_plan(w, h, s->lv);
    bool ok = true;
    auto alloc = [&](float** p, size_t n) { ok = ok && hipMalloc(reinterpret_cast<void**>(p), sizeof(float) * n) == hipSuccess; };
    const size_t npx = (size_t)w * h;
    for (int i = 0; i < s->n; i++) {
        const size_t n = (size_t)s->lv[i].w * s->lv[i].h;
        alloc(&s->Lt[i], n); alloc(&s->Lsmooth[i], n); alloc(&s->Lx[i], n); alloc(&s->Ly[i], n); alloc(&s->Ldet[i], n);
        s->mask[i].resize(n); s->val[i].resize(n);
    }
    alloc(&s->img, npx);
    for (float*& p : s->s) alloc(&p, npx);
    ok = ok && hipMalloc(reinterpret_cast<void**>(&s->img8), npx) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&s->hist), sizeof(int) * 512) == hipSuccess &&
         hipMalloc(reinterpret_cast<void**>(&s->cand), sizeof(AkCand) * kAkCandCap) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&s->cand_n), sizeof(int) * kAkMaxLevels) == hipSuccess &&
         hipMalloc(reinterpret_cast<void**>(&s->d_kps), sizeof(uvo_keypoint) * (size_t)s->cap) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&s->d_desc), (size_t)kAkDescBytes * s->cap) == hipSuccess &&
         hipMalloc(reinterpret_cast<void**>(&s->d_kc), sizeof(float)) == hipSuccess && hipHostMalloc(reinterpret_cast<void**>(&s->h_kc), sizeof(float)) == hipSuccess &&
         hipHostMalloc(reinterpret_cast<void**>(&s->h_cand), sizeof(AkCand) * kAkCandCap) == hipSuccess && hipHostMalloc(reinterpret_cast<void**>(&s->h_hist), sizeof(int) * 512) == hipSuccess;
    // resize(Lt[i - 1], Lt[i], INTER_AREA) at an octave change: exact halving has a kernel of its own; other size ratios (odd sizes) take
    // resizeArea_'s tables, which depend on the sizes alone
    for (int i = 1; i < s->n && ok; i++) {
        if (!(s->lv[i].octave > s->lv[i - 1].octave)) continue;
        const int lw = s->lv[i].w, lh = s->lv[i].h, sw = s->lv[i - 1].w, sh = s->lv[i - 1].h;
        const double scale_x = 1. / ((double)lw / sw), scale_y = 1. / ((double)lh / sh);
        const int isx = cv_round_d(scale_x), isy = cv_round_d(scale_y);
        if (fabs(scale_x - isx) < DBL_EPSILON && fabs(scale_y - isy) < DBL_EPSILON && isx == 2 && isy == 2) continue;
        std::vector<AkTab> xt, yt; std::vector<int> xo, yo;
        area_tab(sw, lw, scale_x, &xt, &xo); area_tab(sh, lh, scale_y, &yt, &yo);
        ok = hipMalloc(reinterpret_cast<void**>(&s->xtab[i]), sizeof(AkTab) * (xt.size() + 1)) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&s->ytab[i]), sizeof(AkTab) * (yt.size() + 1)) == hipSuccess &&
             hipMalloc(reinterpret_cast<void**>(&s->xofs[i]), sizeof(int) * xo.size()) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&s->yofs[i]), sizeof(int) * yo.size()) == hipSuccess &&
             hipMemcpy(s->xtab[i], xt.data(), sizeof(AkTab) * xt.size(), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(s->ytab[i], yt.data(), sizeof(AkTab) * yt.size(), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(s->xofs[i], xo.data(), sizeof(int) * xo.size(), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(s->yofs[i], yo.data(), sizeof(int) * yo.size(), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) { akaze_ws_free(c); return nullptr; }
    s->cands.resize(kAkMaxLevels);
    return s;
}
static void gauss_kernel(int n, double sigma, AkKernel* kk)       // getGaussianKernel(n, sigma, CV_32F): double, normalised, cast
{
    double t[16], sum = 0;
    const double scale2X = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; i++) { const double x = i - (n - 1) * 0.5; t[i] = exp(scale2X * x * x); sum += t[i]; }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) kk->k[i] = (float)(t[i] * sum);
    kk->ksize = n;
}
// computeResizeAreaTab (resize.cpp) of one axis -> (si, alpha) entries grouped by destination index, ofs[d] .. ofs[d + 1]
static void area_tab(int ssize, int dsize, double scale, std::vector<AkTab>* tab, std::vector<int>* ofs)
{
    tab->clear(); ofs->assign(dsize + 1, 0);
    for (int dx = 0; dx < dsize; dx++) {
        (*ofs)[dx] = (int)tab->size();
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cellWidth = std::min(scale, ssize - fsx1);
        int sx1 = cv_ceil_d(fsx1), sx2 = cv_floor_d(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) tab->push_back(AkTab{ sx1 - 1, (float)((sx1 - fsx1) / cellWidth) });
        for (int sx = sx1; sx < sx2; sx++) tab->push_back(AkTab{ sx, (float)(1.0 / cellWidth) });
        if (fsx2 - sx2 > 1e-3) tab->push_back(AkTab{ sx2, (float)(std::min(std::min(fsx2 - sx2, 1.), cellWidth) / cellWidth) });
    }
    (*ofs)[dsize] = (int)tab->size();
}
// find_neighbor_point (AKAZEFeatures.cpp): a keypoint of `mask` within search_radius (L2) of (x, y), scanning the square window row-major
static bool find_neighbor_point(int x, int y, const std::vector<uint8_t>& mask, int cols, int rows, int search_radius, int* idx)
{
    // (the marks are sparse: eight mask bytes are tested at a time and only a non-zero word is walked byte by byte, in the same order)
    const int j0 = std::max(x - search_radius, 0), j1 = std::min(x + search_radius, cols), r2 = search_radius * search_radius;
    for (int i = std::max(y - search_radius, 0); i < std::min(y + search_radius, rows); ++i) {
        const uint8_t* curr = mask.data() + (size_t)i * cols;
        const int dy = i - y;
        for (int j = j0; j < j1;) {
            if (j1 - j >= 8) { uint64_t wd; memcpy(&wd, curr + j, 8); if (wd == 0) { j += 8; continue; } }
            if (curr[j] != 0) { const int dx = j - x; if (dx * dx + dy * dy <= r2) { *idx = i * cols + j; return true; } }
            ++j;
        }
    }
    return false;
}

uvo_status akaze_detect(Ctx* c, const uint8_t* gray, int w, int h, int stride, int mem, uvo_keypoint* kps, uint8_t* desc, int cap, int* n_out)
{
    *n_out = 0;
    if (w < 16 || h < 16 || w > c->max_w || h > c->max_h || stride < w) { c->err = "uvo_akaze_detect: image size outside the context's limits"; return UVO_INVALID_ARG; }
    AkazeWs* s = akaze_ws(c, w, h);
    if (!s) { c->err = "AKAZE workspace allocation failed"; return UVO_HIP_ERROR; }
    hipStream_t st = c->stream;
    const dim3 blk(256);
    auto grid = [](int ww, int hh) { return dim3((ww + 255) / 256, hh); };
    static const bool dbg = getenv("UVO_DBG_PHASE") != nullptr;                                            // host wall clock of the call's phases
    double tph[8] = {0}; int nph = 0;
    auto stamp = [&]() { if (dbg && nph < 8) tph[nph++] = uvo::now_us(); };
    stamp();
    // ---- Create_Nonlinear_Scale_Space ----
    const uint8_t* d_img8 = gray;
    int d_stride = stride;
    if (mem != UVO_MEM_DEVICE) {
        UVO_HIP_TRY(c, hipMemcpy2DAsync(s->img8, w, gray, stride, w, h, hipMemcpyHostToDevice, st));
        d_img8 = s->img8; d_stride = w;
    }
    hipLaunchKernelGGL(k_ak_u8_to_f32, grid(w, h), blk, 0, st, d_img8, d_stride, w, h, s->img);
    AkKernel k9, k5;
    { int ks = cv_ceil_d((double)(2.0f * (1.0f + (1.6f - 0.8f) / (0.3f)))); ks |= 1; gauss_kernel(ks, (double)1.6f, &k9); }
    gauss_kernel(5, (double)1.0f, &k5);
    // last frame's keypoint marks (the masks are whole planes: clearing the few thousand marked entries, not 11 MB of planes per frame)
    for (int i = 0; i < s->n; i++) { for (const AkCand& cd : s->cands[i]) s->mask[i][(size_t)cd.y * s->lv[i].w + cd.x] = 0; s->cands[i].clear(); }
    hipLaunchKernelGGL(k_ak_blur_rows, grid(w, h), blk, 0, st, s->img, w, h, k9, s->s[0]);
    hipLaunchKernelGGL(k_ak_blur_cols, grid(w, h), blk, 0, st, s->s[0], w, h, k9, s->Lsmooth[0]);
    UVO_HIP_TRY(c, hipMemcpyAsync(s->Lt[0], s->Lsmooth[0], sizeof(float) * (size_t)w * h, hipMemcpyDeviceToDevice, st));
    float kcontrast = 0.f;
    if (s->n > 1) {                                                  // compute_kcontrast on the image blurred with sigma 1
        hipLaunchKernelGGL(k_ak_blur_rows, grid(w, h), blk, 0, st, s->img, w, h, k5, s->s[0]);
        hipLaunchKernelGGL(k_ak_blur_cols, grid(w, h), blk, 0, st, s->s[0], w, h, k5, s->s[1]);
        hipLaunchKernelGGL(k_ak_scharr, grid(w, h), blk, 0, st, s->s[1], w, h, 1, s->s[2]);
        hipLaunchKernelGGL(k_ak_scharr, grid(w, h), blk, 0, st, s->s[1], w, h, 0, s->s[3]);
        const int nbins = 300;
        UVO_HIP_TRY(c, hipMemsetAsync(s->hist, 0, sizeof(int) * 512, st));
        if (w > 2 && h > 2) {
            hipLaunchKernelGGL(k_ak_gradmax, dim3((w - 2 + 255) / 256, (h - 2 + 15) / 16), blk, 0, st, s->s[2], s->s[3], w, h, s->hist);
            hipLaunchKernelGGL(k_ak_gradhist, dim3((w - 2 + 1023) / 1024, h - 2), blk, 0, st, s->s[2], s->s[3], w, h, nbins, s->hist);
        }
        UVO_HIP_TRY(c, hipMemcpyAsync(s->h_hist, s->hist, sizeof(int) * 512, hipMemcpyDeviceToHost, st));
        UVO_HIP_TRY(c, hipStreamSynchronize(st));
        float hmax; memcpy(&hmax, &s->h_hist[0], 4);
        const int* hist = s->h_hist + 1;
        const int total = (w - 2) * (h - 2);
        kcontrast = 0.03f;
        if (total > 0 && hmax != 0.0f) {
            const int nthreshold = (int)((total - hist[0]) * 0.7f);
            int nelements = 0;
            for (int k = 1; k < nbins; k++) {
                if (nelements >= nthreshold) { kcontrast = (float)hmax * k / nbins; break; }
                nelements = nelements + hist[k];
            }
        }
    }
    stamp();
    *s->h_kc = kcontrast;
    UVO_HIP_TRY(c, hipMemcpyAsync(s->d_kc, s->h_kc, sizeof(float), hipMemcpyHostToDevice, st));
    // ---- the evolution (levels 1 ..), Compute_Determinant_Hessian_Response and the candidates of every level: ~600 small launches (166 FED
    // steps, 80 derivative passes at 1080p) whose arguments depend on the image SIZE only -- captured once per workspace as a HIP graph
    // and replayed with one call per frame.  Measured: the chain is bound by the DEVICE (600 kernels of 3-6 us back to back: 3.2 ms at
    // 1080p either way; whole call 1.46 ms as a graph against 1.56 ms launch by launch at 640 x 360, level at 1080p); what the graph
    // buys is the host thread, idle instead of issuing launches for 3 ms. ----
    auto record = [&]() -> uvo_status {
        for (int i = 1; i < s->n; i++) {
            const AkLevel& lv = s->lv[i];
            const int lw = lv.w, lh = lv.h;
            if (lv.octave > s->lv[i - 1].octave) {
                const int sw = s->lv[i - 1].w;
                if (!s->xtab[i]) hipLaunchKernelGGL(k_ak_half, grid(lw, lh), blk, 0, st, s->Lt[i - 1], sw, lw, lh, s->Lt[i]);
                else hipLaunchKernelGGL(k_ak_area, grid(lw, lh), blk, 0, st, s->Lt[i - 1], sw, lw, lh, s->xtab[i], s->xofs[i], s->ytab[i], s->yofs[i], s->Lt[i]);
            } else UVO_HIP_TRY(c, hipMemcpyAsync(s->Lt[i], s->Lt[i - 1], sizeof(float) * (size_t)lw * lh, hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(k_ak_blur_rows, grid(lw, lh), blk, 0, st, s->Lt[i], lw, lh, k5, s->s[0]);
            hipLaunchKernelGGL(k_ak_blur_cols, grid(lw, lh), blk, 0, st, s->s[0], lw, lh, k5, s->Lsmooth[i]);
            hipLaunchKernelGGL(k_ak_scharr, grid(lw, lh), blk, 0, st, s->Lsmooth[i], lw, lh, 1, s->s[0]);
            hipLaunchKernelGGL(k_ak_scharr, grid(lw, lh), blk, 0, st, s->Lsmooth[i], lw, lh, 0, s->s[1]);
            hipLaunchKernelGGL(k_ak_pm_g2, dim3((lw * lh + 255) / 256), blk, 0, st, s->s[0], s->s[1], lw * lh, s->d_kc, lv.octave, s->s[2]);
            // Fast Explicit Diffusion: the cycle's steps, ping-pong between Lt[i] and a scratch plane
            float* cur = s->Lt[i]; float* nxt = s->s[3];
            for (int j = 0; j < lv.nsteps; j++) {
                hipLaunchKernelGGL(k_ak_nld_step, grid(lw, lh), blk, 0, st, cur, s->s[2], lw, lh, lv.tau[j] * 0.5f, nxt);
                std::swap(cur, nxt);
            }
            if (cur != s->Lt[i]) UVO_HIP_TRY(c, hipMemcpyAsync(s->Lt[i], cur, sizeof(float) * (size_t)lw * lh, hipMemcpyDeviceToDevice, st));
        }
        UVO_HIP_TRY(c, hipMemsetAsync(s->cand_n, 0, sizeof(int) * kAkMaxLevels, st));
        for (int i = 0; i < s->n; i++) {
            const AkLevel& lv = s->lv[i];
            const int lw = lv.w, lh = lv.h, r = lv.sigma_size;
            hipLaunchKernelGGL(k_ak_sep_deriv, grid(lw, lh), blk, 0, st, s->Lsmooth[i], lw, lh, 1, r, s->Lx[i]);
            hipLaunchKernelGGL(k_ak_sep_deriv, grid(lw, lh), blk, 0, st, s->Lx[i], lw, lh, 1, r, s->s[0]);        // Lxx
            hipLaunchKernelGGL(k_ak_sep_deriv, grid(lw, lh), blk, 0, st, s->Lx[i], lw, lh, 0, r, s->s[1]);        // Lxy
            hipLaunchKernelGGL(k_ak_sep_deriv, grid(lw, lh), blk, 0, st, s->Lsmooth[i], lw, lh, 0, r, s->Ly[i]);
            hipLaunchKernelGGL(k_ak_sep_deriv, grid(lw, lh), blk, 0, st, s->Ly[i], lw, lh, 0, r, s->s[2]);        // Lyy
            hipLaunchKernelGGL(k_ak_det, dim3((lw * lh + 255) / 256), blk, 0, st, s->s[0], s->s[1], s->s[2], lw * lh, (float)(r * r * r * r), s->Ldet[i]);
            if (lv.border + 1 >= lh || lw - 2 * lv.border <= 0 || lh - 2 * lv.border <= 0) continue;              // "if border is too big we shouldn't search any keypoints"
            // every level appends to one list (the record carries its level); cand_n[0] counts them all
            hipLaunchKernelGGL(k_ak_candidates, grid(lw - 2 * lv.border, lh - 2 * lv.border), blk, 0, st, s->Ldet[i], lw, lh, lv.border, 0.001f, i, s->cand, s->cand_n, kAkCandCap);
        }
        return UVO_OK;
    };
    static const bool use_graph = !(getenv("UVO_AKAZE_GRAPH") && atoi(getenv("UVO_AKAZE_GRAPH")) == 0);      // UVO_AKAZE_GRAPH=0: launch by launch (measurement)
    if (use_graph) {
        if (!s->exec) {
            UVO_HIP_TRY(c, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            const uvo_status rs = record();
            hipGraph_t g = nullptr;
            const hipError_t ee = hipStreamEndCapture(st, &g);
            if (rs != UVO_OK || ee != hipSuccess || !g) { if (g) (void)hipGraphDestroy(g); if (rs == UVO_OK) c->err = std::string("AKAZE: graph capture: ") + hipGetErrorString(ee); return rs != UVO_OK ? rs : UVO_HIP_ERROR; }
            s->graph = g;
            UVO_HIP_TRY(c, hipGraphInstantiate(&s->exec, s->graph, nullptr, nullptr, 0));
        }
        UVO_HIP_TRY(c, hipGraphLaunch(s->exec, st));
    } else UVO_TRY(record());
    int n_cand = 0;
    UVO_HIP_TRY(c, hipMemcpyAsync(s->h_hist, s->cand_n, sizeof(int), hipMemcpyDeviceToHost, st));
    UVO_HIP_TRY(c, hipStreamSynchronize(st));
    n_cand = s->h_hist[0];
    if (n_cand > kAkCandCap) { c->err = "AKAZE: more local maxima than the candidate list holds"; return UVO_CAPACITY; }
    if (n_cand) {
        UVO_HIP_TRY(c, hipMemcpyAsync(s->h_cand, s->cand, sizeof(AkCand) * n_cand, hipMemcpyDeviceToHost, st));
        UVO_HIP_TRY(c, hipStreamSynchronize(st));
        // level by level, row-major inside a level -- the order the scans visit them: one sort of 64-bit keys (level, y, x, list position)
        std::vector<uint64_t>& keys = s->keys;
        keys.resize((size_t)n_cand);
        for (int q = 0; q < n_cand; q++) { const AkCand& cd = s->h_cand[q]; keys[(size_t)q] = ((uint64_t)cd.pad << 58) | ((uint64_t)cd.y << 40) | ((uint64_t)cd.x << 22) | (uint64_t)q; }
        std::sort(keys.begin(), keys.end());
        for (int q = 0; q < n_cand; q++) { const AkCand& cd = s->h_cand[keys[(size_t)q] & 0x3fffffu]; s->cands[cd.pad].push_back(cd); }
    }
    UVO_HIP_TRY(c, hipGetLastError());
    stamp();
    // ---- FindKeypointsSameScale's sequential half, Find_Scale_Space_Extrema's two sweeps (host, candidate lists) ----
    for (int i = 0; i < s->n; i++) {
        const AkLevel& lv = s->lv[i];
        for (const AkCand& cd : s->cands[i]) {
            const float value = cd.v[4];
            s->val[i][(size_t)cd.y * lv.w + cd.x] = value;
            int idx = 0;
            if (find_neighbor_point(cd.x, cd.y, s->mask[i], lv.w, lv.h, lv.sigma_size, &idx)) {
                if (value > s->val[i][idx]) s->mask[i][idx] = 0;
                else continue;
            }
            s->mask[i][(size_t)cd.y * lv.w + cd.x] = 1;
        }
    }
    for (int i = 1; i < s->n; i++) {
        const AkLevel& lv = s->lv[i]; const AkLevel& lp = s->lv[i - 1];
        const int diff_ratio = (int)lv.octave_ratio / (int)lp.octave_ratio, search_radius = lv.sigma_size * diff_ratio;
        for (const AkCand& cd : s->cands[i]) {
            const size_t j = (size_t)cd.y * lv.w + cd.x;
            if (s->mask[i][j] == 0) continue;
            int idx = 0;
            if (find_neighbor_point(cd.x * diff_ratio, cd.y * diff_ratio, s->mask[i - 1], lp.w, lp.h, search_radius, &idx))
                if (s->val[i][j] > s->val[i - 1][idx]) s->mask[i - 1][idx] = 0;
        }
    }
    for (int i = s->n - 2; i >= 0; i--) {
        const AkLevel& lv = s->lv[i]; const AkLevel& ln = s->lv[i + 1];
        const int diff_ratio = (int)ln.octave_ratio / (int)lv.octave_ratio, search_radius = ln.sigma_size;
        for (const AkCand& cd : s->cands[i]) {
            const size_t j = (size_t)cd.y * lv.w + cd.x;
            if (s->mask[i][j] == 0) continue;
            int idx = 0;
            if (find_neighbor_point(cd.x / diff_ratio, cd.y / diff_ratio, s->mask[i + 1], ln.w, ln.h, search_radius, &idx))
                if (s->val[i][j] > s->val[i + 1][idx]) s->mask[i + 1][idx] = 0;
        }
    }
    // ---- Do_Subpixel_Refinement (host: a 2 x 2 solve per keypoint on the neighbourhood the candidate record carries) ----
    std::vector<uvo_keypoint> out;
    for (int i = 0; i < s->n; i++) {
        const AkLevel& e = s->lv[i];
        const float ratio = e.octave_ratio;
        for (const AkCand& cd : s->cands[i]) {
            if (s->mask[i][(size_t)cd.y * e.w + cd.x] == 0) continue;
            uvo_keypoint k;
            k.x = cd.x * e.octave_ratio; k.y = cd.y * e.octave_ratio;
            k.size = e.esigma * 1.5f;
            k.angle = -1; k.response = cd.v[4]; k.octave = e.octave; k.class_id = i;
            const float* v = cd.v;                                   // v[3 * (dy + 1) + (dx + 1)]
            const float Dx = 0.5f * (v[5] - v[3]);
            const float Dy = 0.5f * (v[7] - v[1]);
            const float Dxx = v[5] + v[3] - 2.0f * v[4];
            const float Dyy = v[7] + v[1] - 2.0f * v[4];
            const float Dxy = 0.25f * (v[8] + v[0] - v[2] - v[6]);
            float dx = 0.0f, dy = 0.0f;                              // solve(Matx22f, Vec2f, DECOMP_LU): Cramer in float
            const float det = Dxx * Dyy - Dxy * Dxy;
            if (det != 0) { const float d = 1 / det; dx = d * ((-Dx) * Dyy - (-Dy) * Dxy); dy = d * ((-Dy) * Dxx - (-Dx) * Dxy); }
            if (fabsf(dx) > 1.0f || fabsf(dy) > 1.0f) continue;
            k.x += dx * ratio + .5f * (ratio - 1.f);
            k.y += dy * ratio + .5f * (ratio - 1.f);
            k.angle = 0.0f;
            k.size *= 2.0f;
            out.push_back(k);
        }
    }
    const int n = (int)out.size();
    *n_out = n;
    stamp();
    if (n > s->cap) { c->err = "AKAZE found more keypoints than the context's max_kpts"; return UVO_CAPACITY; }
    if ((kps || desc) && n > cap) { c->err = "uvo_akaze_detect: output capacity too small"; return UVO_CAPACITY; }
    if (n == 0) return UVO_OK;
    // ---- Compute_Keypoints_Orientation, MLDB descriptors ----
    AkPlanes pl;
    memset(&pl, 0, sizeof(pl));
    for (int i = 0; i < s->n; i++) { pl.Lt[i] = s->Lt[i]; pl.Lx[i] = s->Lx[i]; pl.Ly[i] = s->Ly[i]; pl.w[i] = s->lv[i].w; pl.h[i] = s->lv[i].h; pl.ratio[i] = s->lv[i].octave_ratio; }
    UVO_HIP_TRY(c, hipMemcpyAsync(s->d_kps, out.data(), sizeof(uvo_keypoint) * n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_ak_orientation, dim3((n + 3) / 4), dim3(256), 0, st, pl, s->d_kps, n);
    hipLaunchKernelGGL(k_ak_mldb, dim3((n + 7) / 8), dim3(256), 0, st, pl, s->d_kps, n, s->d_desc);
    UVO_HIP_TRY(c, hipGetLastError());
    if (kps) UVO_HIP_TRY(c, hipMemcpyAsync(kps, s->d_kps, sizeof(uvo_keypoint) * n, hipMemcpyDeviceToHost, st));
    if (desc) UVO_HIP_TRY(c, hipMemcpyAsync(desc, s->d_desc, (size_t)kAkDescBytes * n, hipMemcpyDeviceToHost, st));
    UVO_HIP_TRY(c, hipStreamSynchronize(st));
    stamp();
    if (dbg && nph == 5) fprintf(stderr, "[uvo] akaze phases (ms): contrast %.3f | evolution + responses + candidates (%d) %.3f | host suppression + refinement %.3f | orientation + M-LDB + copies (%d keypoints) %.3f\n",
                                 (tph[1] - tph[0]) * 1e-3, n_cand, (tph[2] - tph[1]) * 1e-3, (tph[3] - tph[2]) * 1e-3, n, (tph[4] - tph[3]) * 1e-3);
    return UVO_OK;
}
// intermediates for the parity tests: what = 0 Lt, 1 Lsmooth, 2 Lx, 3 Ly, 4 Ldet of `level` after the last akaze_detect
uvo_status akaze_plane(Ctx* c, int level, int what, float* out, int cap_floats, int* ow, int* oh)
{
    AkazeWs* s = static_cast<AkazeWs*>(c->akaze_ws);
    if (!s || level < 0 || level >= s->n || what < 0 || what > 4) { c->err = "uvo_akaze_plane: no such plane (run uvo_akaze_detect first)"; return UVO_INVALID_ARG; }
    const float* src = what == 0 ? s->Lt[level] : what == 1 ? s->Lsmooth[level] : what == 2 ? s->Lx[level] : what == 3 ? s->Ly[level] : s->Ldet[level];
    *ow = s->lv[level].w; *oh = s->lv[level].h;
    const size_t n = (size_t)*ow * *oh;
    if ((size_t)cap_floats < n) { c->err = "uvo_akaze_plane: output capacity too small"; return UVO_CAPACITY; }
    UVO_HIP_TRY(c, hipMemcpy(out, src, sizeof(float) * n, hipMemcpyDeviceToHost));
    return UVO_OK;
}

}  // namespace uvo
