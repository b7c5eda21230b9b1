/* o_orb.c -- TEST INFRASTRUCTURE (CPU oracle; never linked into the product).
 *
 * The ORB branch of detect_features (uvo_libraries/src/VO_utility.cpp:100-105):
 *     Ptr<ORB> detector = ORB::create(10000, 1.2, 8, 31, 0, 2, ORB::HARRIS_SCORE, 31, 10);
 *     detector->detectAndCompute(img, noArray(), keypoints, descriptors);
 * Restated from memory of OpenCV 4.5 features2d/src/orb.cpp (ORB_Impl::detectAndCompute, computeKeyPoints, HarrisResponses, ICAngles,
 * computeOrbDescriptors, makeRandomPattern), fast.cpp / fast_score.cpp (FAST_t<16>, cornerScore<16>), keypoint.cpp
 * (KeyPointsFilter::runByImageBorder, retainBest), imgproc resize.cpp (INTER_LINEAR_EXACT: resize_bitExact, interpolationLinear,
 * ufixedpoint16), smooth / filter (GaussianBlur of a CV_8U SUBMATRIX without BORDER_ISOLATED: the separable filter engine's 8-bit
 * fixed-point kernels), core mathfuncs (fastAtan2), and of libstdc++'s std::nth_element / std::partition, whose element order
 * retainBest exposes.  Published method: E. Rublee, V. Rabaud, K. Konolige, G. Bradski, "ORB: an efficient alternative to SIFT or
 * SURF", ICCV 2011; FAST: E. Rosten, T. Drummond, ECCV 2006.
 *
 * THE DESCRIPTOR'S SAMPLING PATTERN IS AN INPUT.  With patchSize == 31 OpenCV reads its 256 test pairs from `bit_pattern_31_`, a table
 * of 1024 integers learned offline (orb.cpp); it cannot be restated from memory and the reference holds no copy.  Every function here
 * takes the table from the caller in OpenCV's own layout (x0, y0, x1, y1 per bit).  For patch sizes other than 31 OpenCV draws the
 * pairs itself (makeRandomPattern, cv::RNG(0x34985739)); orc_orb_random_pattern restates that generator and the tests use ITS
 * output as the table, so every code path is exercised -- with the learned table the same code gives OpenCV's descriptors.
 *
 * PARITY UNPINNED, confidence MEDIUM: OpenCV is absent here and the reference holds no ORB vectors.  What this file pins is the HIP
 * implementation (ergo_uvo_amd/csrc/orb.hip) to one fixed operation order, and tests/test_oracle_orb_kat.py pins this file to the
 * closed forms of its parts.  Stated departure shared with the HIP path: cos / sin of the keypoint angle are orc_sincos rounded to
 * float (as the SIFT and AKAZE branches; OpenCV calls libm's cosf / sinf, which are not correctly rounded). */
#include "uvo_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORB_MAX_LEVELS 16
#define HARRIS_K 0.04f

void orc_gaussian_kernel_f32(int n, double sigma, float* out);      /* o_surf.c: getGaussianKernel(n, sigma, CV_32F) */
static int reflect101(int p, int n) { if (n == 1) return 0; while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; } return p; }

/* ---- orb.cpp: getScale, the level sizes, the features wanted per level ---- */
static float orb_get_scale(int level, int firstLevel, double scaleFactor) { return (float)pow(scaleFactor, (double)(level - firstLevel)); }

/* out[level] = { width, height, nfeatures wanted }; scale[level]; returns the border (max(edgeThreshold, ceil(half patch * sqrt 2), 4) + 1) */
int orc_orb_levels(int img_w, int img_h, int nfeatures, float scaleFactor_f, int nlevels, int edgeThreshold, int firstLevel, int patchSize,
                   int* out /* [nlevels][3] */, float* scale /* [nlevels] */)
{
    const double scaleFactor = (double)scaleFactor_f;               /* ORB::create takes a float, ORB_Impl keeps a double */
    const int halfPatchSize = patchSize / 2;
    const int descPatchSize = orc_cvCeil(halfPatchSize * sqrt(2.0));
    int border = edgeThreshold > descPatchSize ? edgeThreshold : descPatchSize;
    if (border < 9 / 2) border = 9 / 2;                             /* HARRIS_BLOCK_SIZE / 2 */
    border += 1;
    for (int l = 0; l < nlevels; l++) {
        const float sc = orb_get_scale(l, firstLevel, scaleFactor);
        const float inv = 1.0f / sc;
        scale[l] = sc;
        out[3 * l + 0] = orc_cvRoundf((float)img_w * inv);
        out[3 * l + 1] = orc_cvRoundf((float)img_h * inv);
    }
    /* computeKeyPoints */
    const float factor = (float)(1.0 / scaleFactor);
    float ndesired = (float)nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        out[3 * l + 2] = orc_cvRoundf(ndesired);
        sum += out[3 * l + 2];
        ndesired *= factor;
    }
    out[3 * (nlevels - 1) + 2] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    return border;
}

/* ---- resize.cpp: INTER_LINEAR_EXACT for CV_8UC1 (resize_bitExact<uint8_t, interpolationLinear<uint8_t>>) ---- */
/* per destination index: source offset and the two 8.8 fixed-point weights; [dst_min, dst_max) is the interpolated range, indices
 * below it take the first source sample and indices from dst_max on the last */
static void linear_exact_coeffs(int ssize, int dsize, int* ofs, uint16_t* c0, uint16_t* c1, int* dst_min, int* dst_max)
{
    const double inv_scale = (double)dsize / ssize;                 /* resize(): inv_scale_x = (double)dsize.width / ssize.width */
    const double scale = 1.0 / inv_scale;                           /* softdouble: IEEE double, no fused operations */
    int minofst = 0, maxofst = dsize;
    for (int val = 0; val < dsize; val++) {
        const double fval = scale * ((double)val + 0.5) - 0.5;
        const int ival = orc_cvFloor(fval);
        ofs[val] = 0; c0[val] = 256; c1[val] = 0;
        if (ival >= 0 && ssize > 1) {
            if (ival < ssize - 1) {
                ofs[val] = ival;
                c1[val] = (uint16_t)orc_cvRound((fval - (double)ival) * 256.0);
                c0[val] = (uint16_t)(256 > c1[val] ? 256 - c1[val] : 0);
            } else {
                ofs[val] = ssize - 1;
                if (val < maxofst) maxofst = val;
            }
        } else if (val + 1 > minofst) minofst = val + 1;
    }
    *dst_min = minofst; *dst_max = maxofst;
}
void orc_resize_linear_exact_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride)
{
    int* xo = (int*)malloc(sizeof(int) * (size_t)(dw + dh));
    int* yo = xo + dw;
    uint16_t* xc0 = (uint16_t*)malloc(sizeof(uint16_t) * 2 * (size_t)(dw + dh));
    uint16_t *xc1 = xc0 + dw, *yc0 = xc1 + dw, *yc1 = yc0 + dh;
    int xmin, xmax, ymin, ymax;
    linear_exact_coeffs(sw, dw, xo, xc0, xc1, &xmin, &xmax);
    linear_exact_coeffs(sh, dh, yo, yc0, yc1, &ymin, &ymax);
    uint16_t* l0 = (uint16_t*)malloc(sizeof(uint16_t) * 2 * (size_t)dw);
    uint16_t* l1 = l0 + dw;
    for (int dy = 0; dy < dh; dy++) {
        /* hlineResize of the one or two source rows this destination row reads (8.8 fixed point) */
        const int two = dy >= ymin && dy < ymax;
        const int r0 = dy < ymin ? 0 : (dy >= ymax ? sh - 1 : yo[dy]);
        for (int k = 0; k <= two; k++) {
            const uint8_t* s = src + (size_t)(r0 + k) * sstride;
            uint16_t* l = k ? l1 : l0;
            for (int dx = 0; dx < dw; dx++) {
                if (dx < xmin) l[dx] = (uint16_t)(s[0] << 8);
                else if (dx >= xmax) l[dx] = (uint16_t)(s[sw - 1] << 8);
                else l[dx] = (uint16_t)(xc0[dx] * s[xo[dx]] + xc1[dx] * s[xo[dx] + 1]);
            }
        }
        uint8_t* d = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++) {
            if (!two) d[dx] = (uint8_t)((l0[dx] + 128) >> 8);                                          /* vlineSet */
            else {
                const uint32_t v = (uint32_t)l0[dx] * yc0[dy] + (uint32_t)l1[dx] * yc1[dy];           /* vlineResize, 16.16 */
                const uint32_t r = (v + 32768u) >> 16;
                d[dx] = (uint8_t)(r > 255 ? 255 : r);
            }
        }
    }
    free(l0); free(xc0); free(xo);
}

/* ---- the pyramid (ORB_Impl::detectAndCompute): level 0 = the image, level l = INTER_LINEAR_EXACT of level l - 1, each with a border of
 * `border` pixels of BORDER_REFLECT_101.  OpenCV packs the levels into one buffer; nothing reads across levels, so each level is
 * kept as an image of its own here. ---- */
typedef struct { int w, h, stride; uint8_t* buf; uint8_t* px; } orb_level;      /* px = buf + border * stride + border */
static void level_alloc(orb_level* L, int w, int h, int border)
{
    L->w = w; L->h = h; L->stride = w + 2 * border;
    L->buf = (uint8_t*)calloc((size_t)L->stride * (h + 2 * border), 1);
    L->px = L->buf + (size_t)border * L->stride + border;
}
static void level_make_border(orb_level* L, int border)
{
    for (int y = -border; y < L->h + border; y++) {
        const uint8_t* s = L->px + (ptrdiff_t)reflect101(y, L->h) * L->stride;
        uint8_t* d = L->px + (ptrdiff_t)y * L->stride;
        for (int x = -border; x < L->w + border; x++)
            if (y < 0 || y >= L->h || x < 0 || x >= L->w) d[x] = s[reflect101(x, L->w)];
    }
}
static void build_pyramid(const uint8_t* img, int w, int h, int stride, int nlevels, const int* lv, int border, orb_level* P)
{
    for (int l = 0; l < nlevels; l++) {
        level_alloc(&P[l], lv[3 * l], lv[3 * l + 1], border);
        if (l == 0) for (int y = 0; y < h; y++) memcpy(P[0].px + (size_t)y * P[0].stride, img + (size_t)y * stride, (size_t)w);
        else orc_resize_linear_exact_u8(P[l - 1].px, P[l - 1].w, P[l - 1].h, P[l - 1].stride, P[l].px, P[l].w, P[l].h, P[l].stride);
        level_make_border(&P[l], border);
    }
}

/* ---- fast.cpp: FAST_t<16> with non-maximum suppression; fast_score.cpp: cornerScore<16> ---- */
static const int fast_off[16][2] = { {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3}, {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3} };
/* 0 when (x, y) is no corner at `threshold` (no 9 contiguous circle pixels all darker than v - threshold or all brighter than
 * v + threshold), else the corner score: the largest threshold at which it still is one */
static int fast_score_at(const uint8_t* p, int stride, int threshold)
{
    const int v = p[0];
    int d[25];
    for (int k = 0; k < 25; k++) d[k] = v - p[fast_off[k & 15][1] * stride + fast_off[k & 15][0]];
    int best = 0, is_corner = 0;
    for (int s = 0; s < 16; s++) {
        int mn = d[s], mx = d[s];
        for (int k = 1; k < 9; k++) { if (d[s + k] < mn) mn = d[s + k]; if (d[s + k] > mx) mx = d[s + k]; }
        if (mn > threshold || -mx > threshold) is_corner = 1;
        if (mn > best) best = mn;
        if (-mx > best) best = -mx;
    }
    if (!is_corner) return 0;
    if (best < threshold) best = threshold;                         /* cornerScore starts from a0 = threshold */
    return best - 1;
}
/* the score map of FAST (0 = no corner), rows / columns 3 .. size - 4 only, as FAST_t's loops */
void orc_fast_scores(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score /* w * h */)
{
    memset(score, 0, (size_t)w * h);
    if (threshold < 0) threshold = 0;
    if (threshold > 255) threshold = 255;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) score[(size_t)y * w + x] = (uint8_t)fast_score_at(img + (size_t)y * stride + x, stride, threshold);
}
/* FAST(img, keypoints, threshold, true): corners whose score is strictly above their 8 neighbours', in row-major order */
int orc_fast_detect(const uint8_t* img, int w, int h, int stride, int threshold, orc_keypoint* kps, int cap)
{
    uint8_t* sc = (uint8_t*)malloc((size_t)w * h);
    orc_fast_scores(img, w, h, stride, threshold, sc);
    int n = 0;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const int s = sc[(size_t)y * w + x];
            if (!s) continue;
            const uint8_t *a = sc + (size_t)(y - 1) * w + x, *b = a + w, *c = b + w;
            if (s > b[-1] && s > b[1] && s > a[-1] && s > a[0] && s > a[1] && s > c[-1] && s > c[0] && s > c[1]) {
                if (n < cap) { orc_keypoint k = { (float)x, (float)y, 7.f, -1.f, (float)s, 0, -1 }; kps[n] = k; }
                n++;
            }
        }
    free(sc);
    return n <= cap ? n : -n;
}

/* ---- keypoint.cpp: KeyPointsFilter::retainBest -- libstdc++'s std::nth_element (introselect) with KeypointResponseGreater, then
 * std::partition of the tail by response >= the boundary response.  The surviving SET is every keypoint whose response is at least
 * the n-th largest; their ORDER is whatever those two algorithms leave, restated here step for step (bits/stl_algo.h, stl_heap.h:
 * unchanged between GCC 5 and 13). ---- */
typedef struct { float r; int i; } rb_item;
#define RB_GT(a, b) ((a).r > (b).r)
static void rb_swap(rb_item* a, rb_item* b) { rb_item t = *a; *a = *b; *b = t; }
static void rb_move_median_to_first(rb_item* result, rb_item* a, rb_item* b, rb_item* c)
{
    if (RB_GT(*a, *b)) {
        if (RB_GT(*b, *c)) rb_swap(result, b);
        else if (RB_GT(*a, *c)) rb_swap(result, c);
        else rb_swap(result, a);
    } else if (RB_GT(*a, *c)) rb_swap(result, a);
    else if (RB_GT(*b, *c)) rb_swap(result, c);
    else rb_swap(result, b);
}
static rb_item* rb_unguarded_partition(rb_item* first, rb_item* last, rb_item* pivot)
{
    for (;;) {
        while (RB_GT(*first, *pivot)) ++first;
        --last;
        while (RB_GT(*pivot, *last)) --last;
        if (!(first < last)) return first;
        rb_swap(first, last);
        ++first;
    }
}
static void rb_push_heap(rb_item* first, ptrdiff_t hole, ptrdiff_t top, rb_item value)
{
    ptrdiff_t parent = (hole - 1) / 2;
    while (hole > top && RB_GT(first[parent], value)) { first[hole] = first[parent]; hole = parent; parent = (hole - 1) / 2; }
    first[hole] = value;
}
static void rb_adjust_heap(rb_item* first, ptrdiff_t hole, ptrdiff_t len, rb_item value)
{
    const ptrdiff_t top = hole;
    ptrdiff_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (RB_GT(first[child], first[child - 1])) child--;
        first[hole] = first[child]; hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); first[hole] = first[child - 1]; hole = child - 1; }
    rb_push_heap(first, hole, top, value);
}
static void rb_heap_select(rb_item* first, rb_item* middle, rb_item* last)
{
    const ptrdiff_t len = middle - first;
    if (len >= 2) for (ptrdiff_t parent = (len - 2) / 2;; parent--) { rb_adjust_heap(first, parent, len, first[parent]); if (parent == 0) break; }
    for (rb_item* i = middle; i < last; ++i)
        if (RB_GT(*i, *first)) { rb_item v = *i; *i = *first; rb_adjust_heap(first, 0, len, v); }
}
static void rb_insertion_sort(rb_item* first, rb_item* last)
{
    if (first == last) return;
    for (rb_item* i = first + 1; i != last; ++i) {
        rb_item v = *i;
        if (RB_GT(v, *first)) { memmove(first + 1, first, (size_t)(i - first) * sizeof(rb_item)); *first = v; }
        else { rb_item* l = i; rb_item* nx = i - 1; while (RB_GT(v, *nx)) { *l = *nx; l = nx; --nx; } *l = v; }
    }
}
static void rb_nth_element(rb_item* first, rb_item* nth, rb_item* last)
{
    if (first == last || nth == last) return;
    int lg = 0; for (ptrdiff_t n = last - first; n > 1; n >>= 1) lg++;
    int depth = 2 * lg;
    while (last - first > 3) {
        if (depth == 0) { rb_heap_select(first, nth + 1, last); rb_swap(first, nth); return; }
        --depth;
        rb_item* mid = first + (last - first) / 2;
        rb_move_median_to_first(first, first + 1, mid, last - 1);
        rb_item* cut = rb_unguarded_partition(first + 1, last, first);
        if (cut <= nth) first = cut; else last = cut;
    }
    rb_insertion_sort(first, last);
}
/* responses[n] in their current order -> perm[new position] = old index; returns the new count */
int orc_retain_best(const float* responses, int n, int n_points, int* perm)
{
    if (!(n_points >= 0 && n > n_points)) { for (int i = 0; i < n; i++) perm[i] = i; return n; }
    if (n_points == 0) return 0;
    rb_item* v = (rb_item*)malloc(sizeof(rb_item) * (size_t)n);
    for (int i = 0; i < n; i++) { v[i].r = responses[i]; v[i].i = i; }
    rb_nth_element(v, v + n_points - 1, v + n);
    const float amb = v[n_points - 1].r;
    rb_item *first = v + n_points, *last = v + n;                   /* std::partition(first, last, response >= amb), bidirectional form */
    for (;;) {
        for (;;) { if (first == last) goto done; else if (first->r >= amb) ++first; else break; }
        --last;
        for (;;) { if (first == last) goto done; else if (!(last->r >= amb)) --last; else break; }
        rb_swap(first, last);
        ++first;
    }
done:;
    const int m = (int)(first - v);
    for (int i = 0; i < m; i++) perm[i] = v[i].i;
    free(v);
    return m;
}
static int retain_best_kps(orc_keypoint* k, int n, int n_points)
{
    float* r = (float*)malloc(sizeof(float) * (size_t)(n + 1));
    int* perm = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    for (int i = 0; i < n; i++) r[i] = k[i].response;
    const int m = orc_retain_best(r, n, n_points, perm);
    orc_keypoint* t = (orc_keypoint*)malloc(sizeof(orc_keypoint) * (size_t)(m + 1));
    for (int i = 0; i < m; i++) t[i] = k[perm[i]];
    memcpy(k, t, sizeof(orc_keypoint) * (size_t)m);
    free(t); free(perm); free(r);
    return m;
}

/* ---- orb.cpp: HarrisResponses(blockSize 7, k 0.04) at an integer level position ---- */
float orc_orb_harris(const uint8_t* center, int step)
{
    const int blockSize = 7, r = blockSize / 2;
    const float scale = 1.f / ((1 << 2) * blockSize * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    int a = 0, b = 0, c = 0;
    for (int i = -r; i <= r; i++)
        for (int j = -r; j <= r; j++) {
            const uint8_t* p = center + i * step + j;
            const int Ix = (p[1] - p[-1]) * 2 + (p[-step + 1] - p[-step - 1]) + (p[step + 1] - p[step - 1]);
            const int Iy = (p[step] - p[-step]) * 2 + (p[step - 1] - p[-step - 1]) + (p[step + 1] - p[-step + 1]);
            a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
        }
    return ((float)a * (float)b - (float)c * (float)c - HARRIS_K * ((float)a + (float)b) * ((float)a + (float)b)) * scale_sq_sq;
}

float orc_fast_atan2(float y, float x);                             /* o_surf.c: cv::fastAtan2 (scalar atan_f32), degrees */
/* ---- orb.cpp: the circular patch's row ends (computeKeyPoints) and ICAngles ---- */
void orc_orb_umax(int halfPatchSize, int* umax /* halfPatchSize + 2 */)
{
    const int vmax = orc_cvFloor(halfPatchSize * sqrtf(2.f) / 2 + 1);
    const int vmin = orc_cvCeil(halfPatchSize * sqrtf(2.f) / 2);
    for (int v = 0; v <= halfPatchSize + 1; v++) umax[v] = 0;
    for (int v = 0; v <= vmax; ++v) umax[v] = orc_cvRound(sqrt((double)halfPatchSize * halfPatchSize - v * v));
    for (int v = halfPatchSize, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}
float orc_orb_ic_angle(const uint8_t* center, int step, const int* umax, int half_k)
{
    int m_01 = 0, m_10 = 0;
    for (int u = -half_k; u <= half_k; ++u) m_10 += u * center[u];
    for (int v = 1; v <= half_k; ++v) {
        int v_sum = 0;
        const int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            const int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* ---- GaussianBlur(level, level, Size(7, 7), 2, 2, BORDER_REFLECT_101) on a CV_8U submatrix without BORDER_ISOLATED: sepFilter2D through the
 * filter engine's 8-bit path -- the float kernel times 256 rounded to integers for rows and columns (createSeparableLinearFilter:
 * bdepth CV_32S, bits 8 + 8), result (sum + 2^15) >> 16 saturated.  The taps outside the level are the level's reflect-101 border,
 * still unblurred (the blur writes inside the level only).  In place. ---- */
void orc_orb_blur_kernel(int* k7)
{
    float g[7];
    orc_gaussian_kernel_f32(7, 2.0, g);
    for (int i = 0; i < 7; i++) k7[i] = orc_cvRoundf(g[i] * 256.f);
}
static void level_blur(orb_level* L)
{
    int k[7];
    orc_orb_blur_kernel(k);
    const int w = L->w, h = L->h;
    int* rows = (int*)malloc(sizeof(int) * (size_t)w * (h + 6));
    for (int y = -3; y < h + 3; y++) {
        const uint8_t* s = L->px + (ptrdiff_t)y * L->stride;
        int* r = rows + (size_t)(y + 3) * w;
        for (int x = 0; x < w; x++) { int a = 0; for (int t = -3; t <= 3; t++) a += k[t + 3] * s[x + t]; r[x] = a; }
    }
    for (int y = 0; y < h; y++) {
        uint8_t* d = L->px + (ptrdiff_t)y * L->stride;
        for (int x = 0; x < w; x++) {
            int a = 0;
            for (int t = 0; t < 7; t++) a += k[t] * rows[(size_t)(y + t) * w + x];
            a = (a + (1 << 15)) >> 16;
            d[x] = (uint8_t)(a < 0 ? 0 : (a > 255 ? 255 : a));
        }
    }
    free(rows);
}
void orc_orb_blur_u8(const uint8_t* src, int w, int h, uint8_t* dst)
{
    orb_level L;
    level_alloc(&L, w, h, 3);
    for (int y = 0; y < h; y++) memcpy(L.px + (size_t)y * L.stride, src + (size_t)y * w, (size_t)w);
    level_make_border(&L, 3);
    level_blur(&L);
    for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * w, L.px + (size_t)y * L.stride, (size_t)w);
    free(L.buf);
}

/* ---- orb.cpp: makeRandomPattern (the table OpenCV draws for patch sizes other than 31) ---- */
void orc_orb_random_pattern(int patchSize, int* pattern /* npoints * 2 */, int npoints)
{
    orc_rng rng; orc_rng_init(&rng, 0x34985739);
    for (int i = 0; i < npoints; i++) {
        pattern[2 * i] = orc_rng_uniform(&rng, -patchSize / 2, patchSize / 2 + 1);
        pattern[2 * i + 1] = orc_rng_uniform(&rng, -patchSize / 2, patchSize / 2 + 1);
    }
}
/* ---- orb.cpp: computeOrbDescriptors, WTA_K == 2: 32 bytes, bit b of byte i = [ I(p_{16 i + 2 b}) < I(p_{16 i + 2 b + 1}) ], the pattern
 * rotated by the keypoint's angle, coordinates rounded half to even ---- */
void orc_orb_describe(const uint8_t* center, int step, float angle_deg, const int* pattern, uint8_t* desc)
{
    float angle = angle_deg;
    angle *= (float)(3.1415926535897932384626433832795 / 180.f);
    double sd, cd;
    orc_sincos((double)angle, &sd, &cd);
    const float a = (float)cd, b = (float)sd;
    for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int bit = 0; bit < 8; bit++) {
            int t[2];
            for (int e = 0; e < 2; e++) {
                const int* p = pattern + 2 * (16 * i + 2 * bit + e);
                const float x = (float)p[0] * a - (float)p[1] * b;
                const float y = (float)p[0] * b + (float)p[1] * a;
                t[e] = center[orc_cvRoundf(y) * step + orc_cvRoundf(x)];
            }
            val |= (t[0] < t[1]) << bit;
        }
        desc[i] = (uint8_t)val;
    }
}

/* one pyramid level as OpenCV holds it before (blurred = 0) or after (1) the descriptor stage's GaussianBlur, for the tests */
int orc_orb_level_image(const uint8_t* img, int w, int h, int stride, float scaleFactor, int nlevels, int level, int blurred, uint8_t* out, int cap, int* ow, int* oh)
{
    int lv[3 * ORB_MAX_LEVELS]; float sc[ORB_MAX_LEVELS];
    if (nlevels < 1 || nlevels > ORB_MAX_LEVELS || level < 0 || level >= nlevels) return -1;
    const int border = orc_orb_levels(w, h, 500, scaleFactor, nlevels, 31, 0, 31, lv, sc);
    orb_level P[ORB_MAX_LEVELS];
    build_pyramid(img, w, h, stride, nlevels, lv, border, P);
    if (blurred) level_blur(&P[level]);
    *ow = P[level].w; *oh = P[level].h;
    int ok = P[level].w * P[level].h <= cap;
    if (ok) for (int y = 0; y < P[level].h; y++) memcpy(out + (size_t)y * P[level].w, P[level].px + (size_t)y * P[level].stride, (size_t)P[level].w);
    for (int l = 0; l < nlevels; l++) free(P[l].buf);
    return ok ? 0 : -2;
}

/* ---- ORB_Impl::detectAndCompute(img, noArray(), keypoints, descriptors) for WTA_K 2, HARRIS_SCORE (VOU:103-104).  `pattern`: 1024 ints in
 * OpenCV's bit_pattern_31_ layout, or NULL for keypoints only.  Returns the count, or -(count) if cap is too small. ---- */
int orc_orb_detect_and_compute(const uint8_t* img, int w, int h, int stride, int nfeatures, float scaleFactor, int nlevels, int edgeThreshold,
                               int firstLevel, int patchSize, int fastThreshold, const int* pattern, orc_keypoint* kps, uint8_t* desc, int cap)
{
    if (nlevels < 1 || nlevels > ORB_MAX_LEVELS || firstLevel != 0 || patchSize < 2) return 0;
    int lv[3 * ORB_MAX_LEVELS]; float sc[ORB_MAX_LEVELS];
    const int border = orc_orb_levels(w, h, nfeatures, scaleFactor, nlevels, edgeThreshold, firstLevel, patchSize, lv, sc);
    orb_level P[ORB_MAX_LEVELS];
    build_pyramid(img, w, h, stride, nlevels, lv, border, P);
    const int half = patchSize / 2;
    int* umax = (int*)calloc((size_t)half + 2, sizeof(int));
    orc_orb_umax(half, umax);

    int total = 0, n_all = 0, counters[ORB_MAX_LEVELS];
    size_t all_cap = 1024;
    orc_keypoint* all = (orc_keypoint*)malloc(sizeof(orc_keypoint) * all_cap);
    for (int l = 0; l < nlevels; l++) {
        const int lw = P[l].w, lh = P[l].h, want = lv[3 * l + 2];
        int fcap = lw * lh / 4 + 16;
        orc_keypoint* k = (orc_keypoint*)malloc(sizeof(orc_keypoint) * (size_t)fcap);
        int n = orc_fast_detect(P[l].px, lw, lh, P[l].stride, fastThreshold, k, fcap);
        if (n < 0) n = 0;                                           /* cannot happen: at most a quarter of the pixels survive the 3 x 3 maxima */
        /* KeyPointsFilter::runByImageBorder(keypoints, img.size(), edgeThreshold) */
        if (edgeThreshold > 0) {
            if (lh <= edgeThreshold * 2 || lw <= edgeThreshold * 2) n = 0;
            else { int m = 0; for (int i = 0; i < n; i++) if (k[i].x >= edgeThreshold && k[i].x < lw - edgeThreshold && k[i].y >= edgeThreshold && k[i].y < lh - edgeThreshold) k[m++] = k[i]; n = m; }
        }
        n = retain_best_kps(k, n, 2 * want);                        /* HARRIS_SCORE: twice the level's share survives the FAST ranking */
        counters[l] = n;
        for (int i = 0; i < n; i++) { k[i].octave = l; k[i].size = (float)patchSize * sc[l]; }
        if ((size_t)(n_all + n) > all_cap) { while ((size_t)(n_all + n) > all_cap) all_cap *= 2; all = (orc_keypoint*)realloc(all, sizeof(orc_keypoint) * all_cap); }
        memcpy(all + n_all, k, sizeof(orc_keypoint) * (size_t)n);
        n_all += n;
        free(k);
    }
    /* HarrisResponses on everything kept, then retainBest per level to the level's share */
    for (int i = 0; i < n_all; i++) {
        const orb_level* L = &P[all[i].octave];
        all[i].response = orc_orb_harris(L->px + (ptrdiff_t)orc_cvRoundf(all[i].y) * L->stride + orc_cvRoundf(all[i].x), L->stride);
    }
    orc_keypoint* fin = (orc_keypoint*)malloc(sizeof(orc_keypoint) * (size_t)(n_all + 1));
    for (int l = 0, off = 0; l < nlevels; l++) {
        const int n = retain_best_kps(all + off, counters[l], lv[3 * l + 2]);
        memcpy(fin + total, all + off, sizeof(orc_keypoint) * (size_t)n);
        total += n; off += counters[l];
    }
    free(all);
    /* ICAngles, then pt *= the level's scale */
    for (int i = 0; i < total; i++) {
        const orb_level* L = &P[fin[i].octave];
        fin[i].angle = orc_orb_ic_angle(L->px + (ptrdiff_t)orc_cvRoundf(fin[i].y) * L->stride + orc_cvRoundf(fin[i].x), L->stride, umax, half);
    }
    for (int i = 0; i < total; i++) { const float s = sc[fin[i].octave]; fin[i].x *= s; fin[i].y *= s; }
    const int ok = total <= cap;
    if (ok) memcpy(kps, fin, sizeof(orc_keypoint) * (size_t)total);
    if (ok && desc && pattern && total > 0) {
        for (int l = 0; l < nlevels; l++) level_blur(&P[l]);
        for (int i = 0; i < total; i++) {
            const orb_level* L = &P[fin[i].octave];
            const float s = 1.f / sc[fin[i].octave];
            orc_orb_describe(L->px + (ptrdiff_t)orc_cvRoundf(fin[i].y * s) * L->stride + orc_cvRoundf(fin[i].x * s), L->stride, fin[i].angle, pattern, desc + 32 * (size_t)i);
        }
    }
    free(fin); free(umax);
    for (int l = 0; l < nlevels; l++) free(P[l].buf);
    return ok ? total : -total;
}
