// orb.hip -- the ORB branch of detect_features on gfx950 (uvo_libraries/src/VO_utility.cpp:100-105):
//     Ptr<ORB> detector = ORB::create(10000, 1.2, 8, 31, 0, 2, ORB::HARRIS_SCORE, 31, 10);
//     detector->detectAndCompute(img, noArray(), keypoints, descriptors);
// oriented FAST + rotated BRIEF (Rublee, Rabaud, Konolige, Bradski, ICCV 2011) in the form of OpenCV 4.5's features2d/src/orb.cpp as far
// as it can be recalled (PARITY UNPINNED).  All of it is integer or byte work on small images -- HBM / L1 bound, no matrix shapes:
//   k_orb_resize       the pyramid: level l = resize(level l - 1, INTER_LINEAR_EXACT): 8.8 fixed-point weights per axis from tables
//                      computed once per image size on the host, two roundings (16-bit row result, then >> 16)
//   k_orb_fast_score   FAST-9/16: per pixel the largest threshold at which nine contiguous circle pixels are all darker / all brighter
//                      (sliding minima and maxima over the ring by doubling: windows of 2, 4, 8, 9), zero where that is below the
//                      threshold -- what FAST_t<16> + cornerScore<16> leave in their score rows
//   k_orb_nms_rows     strict 3 x 3 maxima of the score map inside the edge margin, one wave per image row, in ROW-MAJOR order
//                      (count pass, k_orb_row_scan over every level's rows, write pass): the order retainBest starts from
//   k_orb_harris       HarrisResponses (7 x 7 block of Sobel products, integer sums) for the keypoints the FAST ranking kept
//   k_orb_keypoints    ICAngles (integer moments over the disc of radius 15, 32 lanes per keypoint, fastAtan2) and the KeyPoint fields
//   k_orb_blur         GaussianBlur 7 x 7, sigma 2, as the 8-bit filter engine runs it on a submatrix: integer taps [18 34 49 55 49 34 18],
//                      rows then columns in 32 bits through an LDS tile, one rounding (+ 2^15) >> 16, BORDER_REFLECT_101
//   k_orb_describe     rBRIEF: 32 lanes per keypoint (one per descriptor byte), the sampling table in LDS, rotated by the keypoint's
//                      angle, coordinates rounded half to even
// KeyPointsFilter::retainBest runs twice per level (on the FAST scores for twice the level's share, on the Harris responses for the
// share).  Its surviving SET is a threshold on the n-th largest response, but the ORDER it leaves -- which is the order of the output
// keypoints -- is that of libstdc++'s std::nth_element + std::partition, a sequential algorithm: the host replays it on the response
// arrays (4 bytes per keypoint down, 4 bytes per survivor up; the images never leave the device), like the RANSAC scans of the pose
// stages.
// THE SAMPLING TABLE IS AN INPUT (uvo_orb_set_pattern): OpenCV's bit_pattern_31_ (1024 integers learned offline) cannot be restated
// and the reference holds no copy; without a table the detector returns keypoints and refuses descriptors.
#include "uvo_ctx.h"
#include "uvo_math.h"
#include <float.h>
#include <string.h>
#include <math.h>
#include <vector>

namespace uvo {

static const int kOrbMaxLevels = 16, kOrbDescBytes = 32;
struct OrbParams { int nfeatures = 10000; float scaleFactor = 1.2f; int nlevels = 8, edgeThreshold = 31, patchSize = 31, fastThreshold = 10; };
struct OrbLevels {                                                  // by value into the kernels
    int n;
    int w[kOrbMaxLevels], h[kOrbMaxLevels], stride[kOrbMaxLevels], row0[kOrbMaxLevels + 1];
    const uint8_t* img[kOrbMaxLevels]; uint8_t* blur[kOrbMaxLevels]; uint8_t* score[kOrbMaxLevels];
    float scale[kOrbMaxLevels];
};
struct OrbPrefix { int n; int start[kOrbMaxLevels + 1]; };          // list positions at which each level's keypoints start
struct OrbWs {
    OrbParams p;
    bool has_pattern = false;
    int8_t* d_pattern = nullptr;                                    // 512 x (x, y)
    // sized by (w, h, p):
    int w = 0, h = 0, cap = 0, border = 0, total_rows = 0, fast_cap = 0;
    int want[kOrbMaxLevels] = {0};
    int umax[40] = {0};
    OrbLevels L;
    uint8_t* d_pix = nullptr;                                       // every level's image, blurred copy and score map
    uint16_t* d_tab = nullptr;                                      // resize tables: per level l >= 1: xofs[w], xc1[w], yofs[h], yc1[h]
    size_t tab_off[kOrbMaxLevels] = {0};
    int* d_rows = nullptr;                                          // [total_rows] counts, [total_rows + 1] offsets, [kOrbMaxLevels + 1] level starts
    uint32_t* d_pos = nullptr; float* d_score = nullptr;            // FAST keypoints in row-major order, level after level
    int* d_sel = nullptr; float* d_resp = nullptr; int* d_fin = nullptr;
    uvo_keypoint* d_kps = nullptr; uint8_t* d_desc = nullptr;
    int* h_int = nullptr; float* h_f = nullptr;                     // pinned
};

// ------------------------------------------------------------------------------------------------ kernels
__global__ __launch_bounds__(256) void k_orb_resize(const uint8_t* __restrict__ src, int sw, int sh, int sstride, uint8_t* __restrict__ dst, int dw, int dh,
                                                    const uint16_t* __restrict__ xofs, const uint16_t* __restrict__ xc1, const uint16_t* __restrict__ yofs, const uint16_t* __restrict__ yc1)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    const int xo = xofs[x], xo1 = min(xo + 1, sw - 1), yo = yofs[y], yo1 = min(yo + 1, sh - 1);
    const uint32_t cx = xc1[x], cy = yc1[y];
    const uint8_t* r0 = src + (size_t)yo * sstride;
    const uint8_t* r1 = src + (size_t)yo1 * sstride;
    const uint32_t h0 = (256u - cx) * r0[xo] + cx * r0[xo1];       // hlineResize: 8.8
    const uint32_t h1 = (256u - cx) * r1[xo] + cx * r1[xo1];
    const uint32_t v = (h0 * (256u - cy) + h1 * cy + 32768u) >> 16; // vlineResize: 16.16, one rounding
    dst[(size_t)y * dw + x] = (uint8_t)min(v, 255u);
}

// FAST-9/16.  ring[k] = v - p_k around the Bresenham circle of radius 3 (fast.cpp makeOffsets); A = max over the 16 arcs of nine of
// min(ring), B = the same for p_k - v; a corner at threshold t iff max(A, B) > t, its score max(A, B) - 1 (cornerScore<16>).
__global__ __launch_bounds__(256) void k_orb_fast_score(const uint8_t* __restrict__ img, int w, int h, int stride, int threshold, uint8_t* __restrict__ score)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    int out = 0;
    if (x >= 3 && x < w - 3 && y >= 3 && y < h - 3) {
        const uint8_t* p = img + (size_t)y * stride + x;
        const int v = p[0];
        int d[16];
        d[0] = v - p[3 * stride];       d[1] = v - p[3 * stride + 1];   d[2] = v - p[2 * stride + 2];   d[3] = v - p[stride + 3];
        d[4] = v - p[3];                d[5] = v - p[-stride + 3];      d[6] = v - p[-2 * stride + 2];  d[7] = v - p[-3 * stride + 1];
        d[8] = v - p[-3 * stride];      d[9] = v - p[-3 * stride - 1];  d[10] = v - p[-2 * stride - 2]; d[11] = v - p[-stride - 3];
        d[12] = v - p[-3];              d[13] = v - p[stride - 3];      d[14] = v - p[2 * stride - 2];  d[15] = v - p[3 * stride - 1];
        int mn[16], mx[16], t0[16], t1[16];
#pragma unroll
        for (int i = 0; i < 16; i++) { t0[i] = min(d[i], d[(i + 1) & 15]); t1[i] = max(d[i], d[(i + 1) & 15]); }          // windows of 2
#pragma unroll
        for (int i = 0; i < 16; i++) { mn[i] = min(t0[i], t0[(i + 2) & 15]); mx[i] = max(t1[i], t1[(i + 2) & 15]); }      // 4
#pragma unroll
        for (int i = 0; i < 16; i++) { t0[i] = min(mn[i], mn[(i + 4) & 15]); t1[i] = max(mx[i], mx[(i + 4) & 15]); }      // 8
        int A = -256, B = -256;
#pragma unroll
        for (int i = 0; i < 16; i++) { A = max(A, min(t0[i], d[(i + 8) & 15])); B = max(B, -max(t1[i], d[(i + 8) & 15])); } // 9
        const int m = max(A, B);
        if (m > threshold) out = m - 1;
    }
    score[(size_t)y * w + x] = (uint8_t)out;
}

// One wave per image row of every level: the strict 3 x 3 maxima of the score map inside the margin (FAST's own 3 pixels and
// KeyPointsFilter::runByImageBorder's edgeThreshold), left to right.  WRITE = false: the row's count; true: positions and scores at
// the row's offset.
template <bool WRITE>
__global__ __launch_bounds__(256) void k_orb_nms_rows(OrbLevels L, int margin, int* __restrict__ rowcount, const int* __restrict__ rowoff,
                                                      uint32_t* __restrict__ pos, float* __restrict__ resp)
{
    const int grow = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (grow >= L.row0[L.n]) return;
    int l = 0;
    while (grow >= L.row0[l + 1]) l++;
    const int y = grow - L.row0[l], w = L.w[l], h = L.h[l];
    int running = 0;
    if (w > 2 * margin && h > 2 * margin && y >= margin && y < h - margin) {
        const uint8_t* s = L.score[l] + (size_t)y * w;
        const int base = WRITE ? rowoff[grow] : 0;
        for (int x0 = margin; x0 < w - margin; x0 += 64) {
            const int x = x0 + lane;
            bool keep = false;
            int sc = 0;
            if (x < w - margin) {
                sc = s[x];
                if (sc) keep = sc > s[x - 1] && sc > s[x + 1] && sc > s[x - w - 1] && sc > s[x - w] && sc > s[x - w + 1] && sc > s[x + w - 1] && sc > s[x + w] && sc > s[x + w + 1];
            }
            const unsigned long long b = __ballot(keep);
            if (WRITE && keep) {
                const int at = base + running + __popcll(b & ((1ull << lane) - 1ull));
                pos[at] = (uint32_t)x | ((uint32_t)y << 16);
                resp[at] = (float)sc;
            }
            running += __popcll(b);
        }
    }
    if (!WRITE && lane == 0) rowcount[grow] = running;
}
// exclusive scan of the row counts (one workgroup); off[n] = the total; lvl[l] = the offset at which level l starts
__global__ __launch_bounds__(1024) void k_orb_row_scan(const int* __restrict__ cnt, int n, int* __restrict__ off, OrbLevels L, int* __restrict__ lvl)
{
    __shared__ int part[1024];
    const int t = threadIdx.x, per = (n + 1023) / 1024, b = t * per, e = min(b + per, n);
    int s = 0;
    for (int i = b; i < e; i++) s += cnt[i];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { const int v = t >= d ? part[t - d] : 0; __syncthreads(); part[t] += v; __syncthreads(); }
    int run = part[t] - s;
    for (int i = b; i < e; i++) { off[i] = run; run += cnt[i]; }
    if (t == 1023) off[n] = part[1023];
    __syncthreads();
    if (t <= L.n) lvl[t] = t == L.n ? part[1023] : off[L.row0[t]];
}

__device__ __forceinline__ int orb_level_of(const OrbPrefix& P, int j) { int l = 0; while (l + 1 < P.n && j >= P.start[l + 1]) l++; return l; }

// HarrisResponses(pyramid, layers, keypoints, 7, 0.04f) for list entry j (the FAST keypoint sel[j])
__global__ __launch_bounds__(256) void k_orb_harris(OrbLevels L, OrbPrefix P, const int* __restrict__ sel, int n, const uint32_t* __restrict__ pos, float* __restrict__ resp)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int l = orb_level_of(P, j), step = L.stride[l];
    const uint32_t pp = pos[sel[j]];
    const uint8_t* c0 = L.img[l] + (size_t)(pp >> 16) * step + (pp & 0xffffu);
    int a = 0, b = 0, c = 0;
    for (int i = -3; i <= 3; i++) {
        const uint8_t* rm = c0 + (i - 1) * step; const uint8_t* r0 = c0 + i * step; const uint8_t* rp = c0 + (i + 1) * step;
#pragma unroll
        for (int q = -3; q <= 3; q++) {
            const int Ix = ((int)r0[q + 1] - (int)r0[q - 1]) * 2 + ((int)rm[q + 1] - (int)rm[q - 1]) + ((int)rp[q + 1] - (int)rp[q - 1]);
            const int Iy = ((int)rp[q] - (int)rm[q]) * 2 + ((int)rp[q - 1] - (int)rm[q - 1]) + ((int)rp[q + 1] - (int)rm[q + 1]);
            a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
        }
    }
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    resp[j] = ((float)a * (float)b - (float)c * (float)c - 0.04f * ((float)a + (float)b) * ((float)a + (float)b)) * scale_sq_sq;
}

__device__ __forceinline__ float orb_atan2_deg(float y, float x)     // cv::fastAtan2 (as surf.hip's fast_atan2_deg)
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
struct OrbUmax { int u[40]; };
// The final keypoints: entry i is Harris-list entry fin[i]; ICAngles over the disc of half_k (32 lanes per keypoint, lane = column),
// pt *= the level's scale, size = patchSize * scale, octave = the level.
__global__ __launch_bounds__(256) void k_orb_keypoints(OrbLevels L, OrbPrefix P, OrbUmax um, int half_k, int patch, const int* __restrict__ fin, int n,
                                                       const int* __restrict__ sel, const uint32_t* __restrict__ pos, const float* __restrict__ resp, uvo_keypoint* __restrict__ kps)
{
    const int i = blockIdx.x * 8 + (threadIdx.x >> 5), lane = threadIdx.x & 31;
    if (i >= n) return;                                             // (a whole 32-lane group leaves together)
    const int j = fin[i], l = orb_level_of(P, j), step = L.stride[l];
    const uint32_t pp = pos[sel[j]];
    const int px = (int)(pp & 0xffffu), py = (int)(pp >> 16);
    const uint8_t* c0 = L.img[l] + (size_t)py * step + px;
    int m01 = 0, m10 = 0;
    for (int u = lane - half_k; u <= half_k; u += 32) {
        const int au = u < 0 ? -u : u;
        for (int v = -half_k; v <= half_k; v++) {
            if (au > um.u[v < 0 ? -v : v]) continue;
            const int val = c0[v * step + u];
            m10 += u * val; m01 += v * val;
        }
    }
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) { m01 += __shfl_xor(m01, d, 32); m10 += __shfl_xor(m10, d, 32); }
    if (lane == 0) {
        const float s = L.scale[l];
        uvo_keypoint k;
        k.x = (float)px * s; k.y = (float)py * s;
        k.size = (float)patch * s;
        k.angle = orb_atan2_deg((float)m01, (float)m10);
        k.response = resp[j];
        k.octave = l; k.class_id = -1;
        kps[i] = k;
    }
}

struct OrbTaps { int k[7]; };
__device__ __forceinline__ int orb_reflect101(int p, int n) { while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p; return p; }
__global__ __launch_bounds__(256) void k_orb_blur(const uint8_t* __restrict__ src, int w, int h, int stride, OrbTaps kk, uint8_t* __restrict__ dst)
{
    __shared__ uint8_t tile[22][72];
    __shared__ int rowf[22][64];
    const int bx = blockIdx.x * 64, by = blockIdx.y * 16, t = threadIdx.x;
    for (int idx = t; idx < 22 * 70; idx += 256) {
        const int ty = idx / 70, tx = idx - ty * 70;
        tile[ty][tx] = src[(size_t)orb_reflect101(by + ty - 3, h) * stride + orb_reflect101(bx + tx - 3, w)];
    }
    __syncthreads();
    for (int idx = t; idx < 22 * 64; idx += 256) {
        const int r = idx >> 6, x = idx & 63;
        int a = 0;
#pragma unroll
        for (int q = 0; q < 7; q++) a += kk.k[q] * tile[r][x + q];
        rowf[r][x] = a;
    }
    __syncthreads();
    for (int idx = t; idx < 16 * 64; idx += 256) {
        const int y = idx >> 6, x = idx & 63;
        if (bx + x >= w || by + y >= h) continue;
        int a = 0;
#pragma unroll
        for (int q = 0; q < 7; q++) a += kk.k[q] * rowf[y + q][x];
        a = (a + (1 << 15)) >> 16;
        dst[(size_t)(by + y) * w + bx + x] = (uint8_t)min(max(a, 0), 255);
    }
}

// computeOrbDescriptors, WTA_K 2: lane = descriptor byte; bit b = I(p[16 byte + 2 b]) < I(p[16 byte + 2 b + 1]) on the blurred level
__global__ __launch_bounds__(256) void k_orb_describe(OrbLevels L, const int8_t* __restrict__ pattern, const uvo_keypoint* __restrict__ kps, int n, uint8_t* __restrict__ desc)
{
    __shared__ int pat32[256];                                      // the table's 1024 signed bytes, copied a word per thread
    pat32[threadIdx.x] = reinterpret_cast<const int*>(pattern)[threadIdx.x];
    const int8_t* pat = reinterpret_cast<const int8_t*>(pat32);
    __syncthreads();
    const int i = blockIdx.x * 8 + (threadIdx.x >> 5), lane = threadIdx.x & 31;
    if (i >= n) return;
    const uvo_keypoint k = kps[i];
    const int l = k.octave, step = L.w[l];
    const float s = 1.f / L.scale[l];
    float angle = k.angle;
    angle *= (float)(3.1415926535897932384626433832795 / 180.f);
    double sd, cd;
    det_sincos((double)angle, &sd, &cd);                            // (OpenCV: libm's cosf / sinf -- stated departure, as the SIFT and AKAZE branches)
    const float a = (float)cd, b = (float)sd;
    const uint8_t* c0 = L.blur[l] + (size_t)cv_round_f(k.y * s) * step + cv_round_f(k.x * s);
    int val = 0;
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
        int t[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int q = 2 * (16 * lane + 2 * bit + e);
            const float fx = (float)pat[q], fy = (float)pat[q + 1];
            const float x = fx * a - fy * b, y = fx * b + fy * a;
            t[e] = c0[cv_round_f(y) * step + cv_round_f(x)];
        }
        val |= (t[0] < t[1]) << bit;
    }
    desc[(size_t)i * kOrbDescBytes + lane] = (uint8_t)val;
}

// ------------------------------------------------------------------------------------------------ host
// KeyPointsFilter::retainBest's order: std::nth_element(first, first + n - 1, last, response greater) followed by
// std::partition(first + n, last, response >= boundary) as libstdc++ implements them (introselect: median of three to the front,
// unguarded Hoare partition, heap select after 2 log2(n) bad splits, insertion sort of the last three; the bidirectional partition).
namespace {
struct RbItem { float r; int i; };
inline bool rb_gt(const RbItem& a, const RbItem& b) { return a.r > b.r; }
void rb_median_to_first(RbItem* res, RbItem* a, RbItem* b, RbItem* c)
{
    if (rb_gt(*a, *b)) { if (rb_gt(*b, *c)) std::swap(*res, *b); else if (rb_gt(*a, *c)) std::swap(*res, *c); else std::swap(*res, *a); }
    else if (rb_gt(*a, *c)) std::swap(*res, *a);
    else if (rb_gt(*b, *c)) std::swap(*res, *c);
    else std::swap(*res, *b);
}
RbItem* rb_hoare(RbItem* first, RbItem* last, const RbItem* pivot)
{
    for (;;) {
        while (rb_gt(*first, *pivot)) ++first;
        --last;
        while (rb_gt(*pivot, *last)) --last;
        if (!(first < last)) return first;
        std::swap(*first, *last);
        ++first;
    }
}
void rb_sift(RbItem* first, ptrdiff_t hole, ptrdiff_t len, RbItem value)      // __adjust_heap + __push_heap
{
    const ptrdiff_t top = hole;
    ptrdiff_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (rb_gt(first[child], first[child - 1])) child--;
        first[hole] = first[child]; hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); first[hole] = first[child - 1]; hole = child - 1; }
    ptrdiff_t parent = (hole - 1) / 2;
    while (hole > top && rb_gt(first[parent], value)) { first[hole] = first[parent]; hole = parent; parent = (hole - 1) / 2; }
    first[hole] = value;
}
void rb_heap_select(RbItem* first, RbItem* middle, RbItem* last)
{
    const ptrdiff_t len = middle - first;
    if (len >= 2) for (ptrdiff_t parent = (len - 2) / 2;; parent--) { rb_sift(first, parent, len, first[parent]); if (parent == 0) break; }
    for (RbItem* i = middle; i < last; ++i)
        if (rb_gt(*i, *first)) { const RbItem v = *i; *i = *first; rb_sift(first, 0, len, v); }
}
void rb_nth(RbItem* first, RbItem* nth, RbItem* last)
{
    if (first == last || nth == last) return;
    int depth = 0;
    for (ptrdiff_t n = last - first; n > 1; n >>= 1) depth += 2;
    while (last - first > 3) {
        if (depth == 0) { rb_heap_select(first, nth + 1, last); std::swap(*first, *nth); return; }
        --depth;
        rb_median_to_first(first, first + 1, first + (last - first) / 2, last - 1);
        RbItem* cut = rb_hoare(first + 1, last, first);
        if (cut <= nth) first = cut; else last = cut;
    }
    for (RbItem* i = first + 1; i < last; ++i) {                    // __insertion_sort
        const RbItem v = *i;
        if (rb_gt(v, *first)) { for (RbItem* q = i; q > first; --q) *q = *(q - 1); *first = v; }
        else { RbItem* q = i; while (rb_gt(v, *(q - 1))) { *q = *(q - 1); --q; } *q = v; }
    }
}
// responses r[0 .. n) -> the surviving old indices in retainBest's order, appended to `out` with `base` added
void retain_best_order(const float* r, int n, int n_points, int base, std::vector<RbItem>* tmp, std::vector<int>* out)
{
    if (!(n_points >= 0 && n > n_points)) { for (int i = 0; i < n; i++) out->push_back(base + i); return; }
    if (n_points == 0) return;
    tmp->resize(n);
    RbItem* v = tmp->data();
    for (int i = 0; i < n; i++) { v[i].r = r[i]; v[i].i = i; }
    rb_nth(v, v + n_points - 1, v + n);
    const float amb = v[n_points - 1].r;
    RbItem *first = v + n_points, *last = v + n;
    for (;;) {
        while (first != last && first->r >= amb) ++first;
        if (first == last) break;
        --last;
        while (first != last && !(last->r >= amb)) --last;
        if (first == last) break;
        std::swap(*first, *last);
        ++first;
    }
    for (RbItem* q = v; q < first; ++q) out->push_back(base + q->i);
}

// resize.cpp interpolationLinear<uint8_t>::getCoeffs over one axis: source offset and the 8.8 weight of the right / lower neighbour.  An
// index left of the first (right of the last) sample centre takes that sample alone: offset at the end, weight 0.
void linear_exact_table(int ssize, int dsize, uint16_t* ofs, uint16_t* c1)
{
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1.0 / inv_scale;
    for (int val = 0; val < dsize; val++) {
        const double fval = scale * ((double)val + 0.5) - 0.5;
        const int ival = cv_floor_d(fval);
        ofs[val] = 0; c1[val] = 0;
        if (ival >= 0 && ssize > 1) {
            if (ival < ssize - 1) { ofs[val] = (uint16_t)ival; c1[val] = (uint16_t)cv_round_d((fval - (double)ival) * 256.0); }
            else ofs[val] = (uint16_t)(ssize - 1);
        }
    }
}
}  // namespace

static void orb_free_sized(OrbWs* s)
{
    (void)hipFree(s->d_pix); (void)hipFree(s->d_tab); (void)hipFree(s->d_rows); (void)hipFree(s->d_pos); (void)hipFree(s->d_score); (void)hipFree(s->d_sel);
    (void)hipFree(s->d_resp); (void)hipFree(s->d_fin); (void)hipFree(s->d_kps); (void)hipFree(s->d_desc); (void)hipHostFree(s->h_int); (void)hipHostFree(s->h_f);
    s->d_pix = nullptr; s->d_tab = nullptr; s->d_rows = nullptr; s->d_pos = nullptr; s->d_score = nullptr; s->d_sel = nullptr; s->d_resp = nullptr; s->d_fin = nullptr;
    s->d_kps = nullptr; s->d_desc = nullptr; s->h_int = nullptr; s->h_f = nullptr;
    s->w = s->h = 0;
}
void orb_ws_free(Ctx* c)
{
    OrbWs* s = static_cast<OrbWs*>(c->orb_ws);
    if (!s) return;
    orb_free_sized(s);
    (void)hipFree(s->d_pattern);
    delete s;
    c->orb_ws = nullptr;
}
static OrbWs* orb_state(Ctx* c)
{
    if (!c->orb_ws) c->orb_ws = new OrbWs();
    return static_cast<OrbWs*>(c->orb_ws);
}
// ORB_Impl::detectAndCompute's level geometry and computeKeyPoints' shares (orb.cpp), tables and buffers for one image size
static uvo_status orb_plan(Ctx* c, OrbWs* s, int w, int h)
{
    if (s->w == w && s->h == h && s->d_pix) return UVO_OK;
    orb_free_sized(s);
    const OrbParams& p = s->p;
    const double scaleFactor = (double)p.scaleFactor;               // ORB::create takes a float, ORB_Impl keeps a double
    const int half = p.patchSize / 2;
    s->border = std::max(p.edgeThreshold, std::max(cv_ceil_d(half * sqrt(2.0)), 9 / 2)) + 1;
    OrbLevels& L = s->L;
    memset(&L, 0, sizeof(L));
    L.n = p.nlevels;
    size_t px = 0;
    int rows = 0;
    for (int l = 0; l < L.n; l++) {
        const float sc = (float)pow(scaleFactor, (double)l);        // getScale(level, firstLevel = 0, scaleFactor)
        const float inv = 1.0f / sc;
        L.scale[l] = sc;
        L.w[l] = cv_round_f((float)w * inv); L.h[l] = cv_round_f((float)h * inv);
        if (L.w[l] < 2 || L.h[l] < 2) { c->err = "uvo_orb_detect: the image is too small for this many pyramid levels"; return UVO_INVALID_ARG; }
        L.stride[l] = L.w[l];
        L.row0[l] = rows; rows += L.h[l];
        px += (size_t)L.w[l] * L.h[l];
    }
    L.row0[L.n] = rows;
    s->total_rows = rows;
    const float factor = (float)(1.0 / scaleFactor);
    float ndesired = (float)p.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)L.n));
    int sum = 0;
    for (int l = 0; l < L.n - 1; l++) { s->want[l] = cv_round_f(ndesired); sum += s->want[l]; ndesired *= factor; }
    s->want[L.n - 1] = std::max(p.nfeatures - sum, 0);
    {   // the disc's row ends (computeKeyPoints: umax)
        const int vmax = cv_floor_d(half * sqrtf(2.f) / 2 + 1), vmin = cv_ceil_d(half * sqrtf(2.f) / 2);
        memset(s->umax, 0, sizeof(s->umax));
        for (int v = 0; v <= vmax; ++v) s->umax[v] = cv_round_d(sqrt((double)half * half - v * v));
        for (int v = half, v0 = 0; v >= vmin; --v) { while (s->umax[v0] == s->umax[v0 + 1]) ++v0; s->umax[v] = v0; ++v0; }
    }
    s->fast_cap = (int)std::min<size_t>(px / 4 + 64, (size_t)1 << 24);      // strict 3 x 3 maxima: at most one per 2 x 2 block
    s->cap = std::max(c->cap, p.nfeatures + p.nfeatures / 4 + 1024);     // retainBest bounds the output by nfeatures, ties at the per-level cuts aside
    bool ok = hipMalloc(reinterpret_cast<void**>(&s->d_pix), 3 * px + 64) == hipSuccess;
    // resize tables
    std::vector<uint16_t> tab;
    for (int l = 1; l < L.n; l++) {
        s->tab_off[l] = tab.size();
        tab.resize(tab.size() + 2 * (size_t)(L.w[l] + L.h[l]));
        uint16_t* t = tab.data() + s->tab_off[l];
        linear_exact_table(L.w[l - 1], L.w[l], t, t + L.w[l]);
        linear_exact_table(L.h[l - 1], L.h[l], t + 2 * L.w[l], t + 2 * L.w[l] + L.h[l]);
    }
    ok = ok && hipMalloc(reinterpret_cast<void**>(&s->d_tab), sizeof(uint16_t) * (tab.size() + 8)) == hipSuccess &&
         hipMalloc(reinterpret_cast<void**>(&s->d_rows), sizeof(int) * (2 * (size_t)rows + 2 + kOrbMaxLevels + 1)) == hipSuccess &&
         hipMalloc(reinterpret_cast<void**>(&s->d_pos), sizeof(uint32_t) * (size_t)s->fast_cap) == hipSuccess &&
         hipMalloc(reinterpret_cast<void**>(&s->d_score), sizeof(float) * (size_t)s->fast_cap) == hipSuccess &&
         hipMalloc(reinterpret_cast<void**>(&s->d_sel), sizeof(int) * (size_t)s->fast_cap) == hipSuccess &&
         hipMalloc(reinterpret_cast<void**>(&s->d_resp), sizeof(float) * (size_t)s->fast_cap) == hipSuccess &&
         hipMalloc(reinterpret_cast<void**>(&s->d_fin), sizeof(int) * (size_t)s->fast_cap) == hipSuccess &&
         hipMalloc(reinterpret_cast<void**>(&s->d_kps), sizeof(uvo_keypoint) * (size_t)s->cap) == hipSuccess &&
         hipMalloc(reinterpret_cast<void**>(&s->d_desc), (size_t)kOrbDescBytes * s->cap) == hipSuccess &&
         hipHostMalloc(reinterpret_cast<void**>(&s->h_int), sizeof(int) * ((size_t)s->fast_cap + 64)) == hipSuccess &&
         hipHostMalloc(reinterpret_cast<void**>(&s->h_f), sizeof(float) * (size_t)s->fast_cap) == hipSuccess;
    if (ok && !tab.empty()) ok = hipMemcpy(s->d_tab, tab.data(), sizeof(uint16_t) * tab.size(), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { orb_free_sized(s); c->err = "ORB workspace allocation failed"; return UVO_HIP_ERROR; }
    uint8_t* q = s->d_pix;
    for (int l = 0; l < L.n; l++) { const size_t n = (size_t)L.w[l] * L.h[l]; L.img[l] = q; L.blur[l] = q + px; L.score[l] = q + 2 * px; q += n; }
    s->w = w; s->h = h;
    return UVO_OK;
}

uvo_status orb_configure(Ctx* c, int nfeatures, float scaleFactor, int nlevels, int edgeThreshold, int patchSize, int fastThreshold)
{
    OrbWs* s = orb_state(c);
    const int reach = cv_ceil_d((patchSize / 2) * sqrt(2.0)) + 1;   // the rotated table's reach; OpenCV pads its levels by this much, this build keeps keypoints that far inside
    if (nfeatures < 1 || nlevels < 1 || nlevels > kOrbMaxLevels || !(scaleFactor > 1.0f) || patchSize < 5 || patchSize > 63 || fastThreshold < 1 || fastThreshold > 254 ||
        edgeThreshold < reach || edgeThreshold < 5) {
        c->err = "uvo_orb_configure: nfeatures >= 1, scaleFactor > 1, nlevels 1..16, patchSize 5..63, fastThreshold 1..254, edgeThreshold >= ceil(patchSize / 2 * sqrt 2) + 1";
        return UVO_INVALID_ARG;
    }
    if (patchSize != s->p.patchSize) s->has_pattern = false;        // a table belongs to its patch size
    s->p.nfeatures = nfeatures; s->p.scaleFactor = scaleFactor; s->p.nlevels = nlevels; s->p.edgeThreshold = edgeThreshold; s->p.patchSize = patchSize; s->p.fastThreshold = fastThreshold;
    orb_free_sized(s);
    return UVO_OK;
}
uvo_status orb_set_pattern(Ctx* c, const int* pattern)
{
    OrbWs* s = orb_state(c);
    if (!pattern) { s->has_pattern = false; return UVO_OK; }
    int8_t p8[1024];
    const int half = s->p.patchSize / 2;
    for (int i = 0; i < 1024; i++) {
        if (pattern[i] < -half || pattern[i] > half) { c->err = "uvo_orb_set_pattern: a coordinate lies outside the patch (|x|, |y| <= patchSize / 2)"; return UVO_INVALID_ARG; }
        p8[i] = (int8_t)pattern[i];
    }
    if (!s->d_pattern) UVO_HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&s->d_pattern), 1024));
    UVO_HIP_TRY(c, hipMemcpy(s->d_pattern, p8, 1024, hipMemcpyHostToDevice));
    s->has_pattern = true;
    return UVO_OK;
}

uvo_status orb_detect(Ctx* c, const uint8_t* gray, int w, int h, int stride, int mem, uvo_keypoint* kps, uint8_t* desc, int cap, int* n_out)
{
    *n_out = 0;
    if (w < 16 || h < 16 || w > c->max_w || h > c->max_h || stride < w || w > 65535 || h > 65535) { c->err = "uvo_orb_detect: image size outside the context's limits"; return UVO_INVALID_ARG; }
    OrbWs* s = orb_state(c);
    if (desc && !s->has_pattern) {
        c->err = "uvo_orb_detect: descriptors need the sampling table -- OpenCV's bit_pattern_31_ (orb.cpp), 256 x (x0, y0, x1, y1) -- through uvo_orb_set_pattern; pass desc = NULL for keypoints only";
        return UVO_INVALID_ARG;
    }
    UVO_TRY(orb_plan(c, s, w, h));
    hipStream_t st = c->stream;
    OrbLevels L = s->L;
    const OrbParams& p = s->p;
    // ---- the pyramid ----
    if (mem == UVO_MEM_DEVICE) { L.img[0] = gray; L.stride[0] = stride; }
    else UVO_HIP_TRY(c, hipMemcpy2DAsync(const_cast<uint8_t*>(L.img[0]), w, gray, stride, w, h, hipMemcpyHostToDevice, st));
    for (int l = 1; l < L.n; l++) {
        const uint16_t* t = s->d_tab + s->tab_off[l];
        hipLaunchKernelGGL(k_orb_resize, dim3((L.w[l] + 255) / 256, L.h[l]), dim3(256), 0, st, L.img[l - 1], L.w[l - 1], L.h[l - 1], L.stride[l - 1],
                           const_cast<uint8_t*>(L.img[l]), L.w[l], L.h[l], t, t + L.w[l], t + 2 * L.w[l], t + 2 * L.w[l] + L.h[l]);
    }
    // ---- FAST on every level, maxima in row-major order ----
    for (int l = 0; l < L.n; l++)
        hipLaunchKernelGGL(k_orb_fast_score, dim3((L.w[l] + 63) / 64, (L.h[l] + 3) / 4), dim3(256), 0, st, L.img[l], L.w[l], L.h[l], L.stride[l], p.fastThreshold, L.score[l]);
    const int margin = std::max(p.edgeThreshold, 3), R = s->total_rows;
    int* d_cnt = s->d_rows; int* d_off = s->d_rows + R; int* d_lvl = s->d_rows + 2 * R + 1;
    hipLaunchKernelGGL(k_orb_nms_rows<false>, dim3((R + 3) / 4), dim3(256), 0, st, L, margin, d_cnt, d_off, s->d_pos, s->d_score);
    hipLaunchKernelGGL(k_orb_row_scan, dim3(1), dim3(1024), 0, st, d_cnt, R, d_off, L, d_lvl);
    UVO_HIP_TRY(c, hipMemcpyAsync(s->h_int, d_lvl, sizeof(int) * (L.n + 1), hipMemcpyDeviceToHost, st));
    UVO_HIP_TRY(c, hipStreamSynchronize(st));
    int lvl[kOrbMaxLevels + 1];
    memcpy(lvl, s->h_int, sizeof(int) * (L.n + 1));
    const int n_fast = lvl[L.n];
    if (n_fast > s->fast_cap) { c->err = "ORB: more FAST corners than the list holds"; return UVO_CAPACITY; }
    std::vector<int> sel;
    OrbPrefix P1; memset(&P1, 0, sizeof(P1)); P1.n = L.n;
    std::vector<RbItem> tmp;
    if (n_fast > 0) {
        hipLaunchKernelGGL(k_orb_nms_rows<true>, dim3((R + 3) / 4), dim3(256), 0, st, L, margin, d_cnt, d_off, s->d_pos, s->d_score);
        UVO_HIP_TRY(c, hipMemcpyAsync(s->h_f, s->d_score, sizeof(float) * n_fast, hipMemcpyDeviceToHost, st));
        UVO_HIP_TRY(c, hipStreamSynchronize(st));
        // retainBest(keypoints, 2 * featuresNum) on the FAST scores (HARRIS_SCORE keeps twice the share for the second ranking)
        for (int l = 0; l < L.n; l++) { P1.start[l] = (int)sel.size(); retain_best_order(s->h_f + lvl[l], lvl[l + 1] - lvl[l], 2 * s->want[l], lvl[l], &tmp, &sel); }
    }
    P1.start[L.n] = (int)sel.size();
    const int n1 = (int)sel.size();
    std::vector<int> fin;
    if (n1 > 0) {
        UVO_HIP_TRY(c, hipMemcpyAsync(s->d_sel, sel.data(), sizeof(int) * n1, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_orb_harris, dim3((n1 + 255) / 256), dim3(256), 0, st, L, P1, s->d_sel, n1, s->d_pos, s->d_resp);
        UVO_HIP_TRY(c, hipMemcpyAsync(s->h_f, s->d_resp, sizeof(float) * n1, hipMemcpyDeviceToHost, st));
        UVO_HIP_TRY(c, hipStreamSynchronize(st));
        // retainBest(keypoints, featuresNum) on the Harris responses, level by level
        for (int l = 0; l < L.n; l++) retain_best_order(s->h_f + P1.start[l], P1.start[l + 1] - P1.start[l], s->want[l], P1.start[l], &tmp, &fin);
    }
    UVO_HIP_TRY(c, hipGetLastError());
    const int n = (int)fin.size();
    *n_out = n;
    if (n > s->cap) { c->err = "ORB: more keypoints tie at the per-level cuts than the output list has room for"; return UVO_CAPACITY; }
    if ((kps || desc) && n > cap) { c->err = "uvo_orb_detect: output capacity too small"; return UVO_CAPACITY; }
    if (n == 0) return UVO_OK;
    // ---- ICAngles + the KeyPoint fields; blur; descriptors ----
    OrbUmax um; memcpy(um.u, s->umax, sizeof(um.u));
    UVO_HIP_TRY(c, hipMemcpyAsync(s->d_fin, fin.data(), sizeof(int) * n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_orb_keypoints, dim3((n + 7) / 8), dim3(256), 0, st, L, P1, um, p.patchSize / 2, p.patchSize, s->d_fin, n, s->d_sel, s->d_pos, s->d_resp, s->d_kps);
    if (desc) {
        OrbTaps kk;
        {   // getGaussianKernel(7, 2, CV_32F) x 2^8 rounded: the 8-bit filter engine's integer taps
            double t[7], sum = 0;
            for (int i = 0; i < 7; i++) { const double x = i - 3; t[i] = exp(-0.5 / (2.0 * 2.0) * x * x); sum += t[i]; }
            for (int i = 0; i < 7; i++) kk.k[i] = cv_round_f((float)(t[i] * (1. / sum)) * 256.f);
        }
        for (int l = 0; l < L.n; l++)
            hipLaunchKernelGGL(k_orb_blur, dim3((L.w[l] + 63) / 64, (L.h[l] + 15) / 16), dim3(256), 0, st, L.img[l], L.w[l], L.h[l], L.stride[l], kk, L.blur[l]);
        hipLaunchKernelGGL(k_orb_describe, dim3((n + 7) / 8), dim3(256), 0, st, L, s->d_pattern, s->d_kps, n, s->d_desc);
    }
    UVO_HIP_TRY(c, hipGetLastError());
    if (kps) UVO_HIP_TRY(c, hipMemcpyAsync(kps, s->d_kps, sizeof(uvo_keypoint) * n, hipMemcpyDeviceToHost, st));
    if (desc) UVO_HIP_TRY(c, hipMemcpyAsync(desc, s->d_desc, (size_t)kOrbDescBytes * n, hipMemcpyDeviceToHost, st));
    UVO_HIP_TRY(c, hipStreamSynchronize(st));
    return UVO_OK;
}
// intermediates for the parity tests: what = 0 the level's image, 1 its blurred copy (after a detect with descriptors), 2 its FAST score map
uvo_status orb_level_plane(Ctx* c, int level, int what, uint8_t* out, int cap_bytes, int* ow, int* oh)
{
    OrbWs* s = static_cast<OrbWs*>(c->orb_ws);
    if (!s || !s->d_pix || level < 0 || level >= s->L.n || what < 0 || what > 2 || (what == 0 && level == 0)) {
        c->err = "uvo_orb_plane: no such plane (run uvo_orb_detect first; level 0's image is the caller's)"; return UVO_INVALID_ARG;
    }
    *ow = s->L.w[level]; *oh = s->L.h[level];
    const size_t n = (size_t)*ow * *oh;
    if ((size_t)cap_bytes < n) { c->err = "uvo_orb_plane: output capacity too small"; return UVO_CAPACITY; }
    const uint8_t* src = what == 0 ? s->L.img[level] : what == 1 ? s->L.blur[level] : s->L.score[level];
    UVO_HIP_TRY(c, hipMemcpy(out, src, n, hipMemcpyDeviceToHost));
    return UVO_OK;
}

}  // namespace uvo
