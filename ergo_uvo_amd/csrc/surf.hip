// surf.hip -- upright SURF-64 detect + describe on gfx950 (replaces detect_features' SURF branch,
// VO_utility.cpp:114-119 -> OpenCV xfeatures2d::SURF::detectAndCompute).
//
// Pipeline per image (both images of a stereo pair go through every launch together, blockIdx.z):
//   integral_rows / integral_colsum / integral_colfinal : u8 -> s32 integral image (HBM-bound), row-major and
//                             de-interleaved by (row & 3, col & 3) for the coarse octaves
//   hessian_nms_c<O> (O = 0, 1) / hessian_nms_p<O> (O = 2, 3) : per octave, fused box-filter Hessian (5 layers) +
//                             3x3x3 NMS + quadratic interpolation; box patterns compile-time; the integral tile is
//                             staged in LDS (octaves 0-1) or read from the planes (2-3); det planes never leave LDS;
//                             candidates are appended with one atomic per keypoint
//   rank_partial / rank_scatter : deterministic ordering (OpenCV's KeypointGreater) by counting rank
//   descriptor64_small / _big (+ _big_tabs, _big_finish) : INTER_AREA window resample + Haar gradients + 4x4x4 sums
// All float arithmetic keeps OpenCV's operation order (int box sum * float weight accumulated in
// double; no FMA contraction) so results are bit-identical to the CPU restatement.
#include "uvo_ctx.h"
#include <type_traits>
#include <string.h>
#include "uvo_math.h"
#include <algorithm>
#include <utility>
#include <vector>
#include <string>

namespace uvo {

// ------------------------------------------------------------------------------------------
// integral image
// ------------------------------------------------------------------------------------------
// planes[im]: the integral image de-interleaved by (row & 3, column & 3): 16 planes of ph x pw, plane (ry, rx) holds
// S[4i + ry][4j + rx] at [i][j].  The step-4 / step-8 sample walks of octaves 2 and 3 become unit / two-element
// strides in them (coalesced), where the row-major image gives one useful word per 16 or 32 bytes.
struct ImgPair { const uint8_t* img[2]; int32_t* sum[2]; int32_t* planes[2]; int pw, pstride; int* cand_n; int* big_n; int* surv_n; };
struct AreaTab;
struct CandOut { uvo_keypoint* cand[2]; int* count; int cap; };
// NMS survivors of a frame: one shared list, a workgroup appends its tile's survivors behind ONE atomic.  (Round 4 also built per-tile
// slots without any atomic: the detection launch gained 0.5 us per tile, k_hessian_finish lost 14 us walking a sparse, unevenly filled
// structure -- tiles in textured regions hold several times the average -- and the list came back.)
struct SurvOut { Survivor* list; int* count; int cap; };
static_assert(sizeof(Survivor) == 128, "a survivor record is 32 words");
struct SortArgs { const uvo_keypoint* cand[2]; const int* cand_n; uvo_keypoint* out[2]; int* out_n[2]; int* rank; int cap;
                  int4* big_par; int* big_n; int* gate_nqa; int gate_min_features; };
struct DescArgs { const uint8_t* img[2]; uvo_keypoint* kps[2]; float* desc[2]; const int* n[2]; const float* DW;
                  const int4* big_par; const int* big_n; const int* big_large; int cap;
                  const AreaTab* tabs; const int* iscale;        // [kMaxWin + 1][21] resize tables, [kMaxWin + 1] integer scale (0: general path)
                  int extended;                                  // SURF_EXTENDED: 128 elements per descriptor row (8 sums per cell)
                  const int32_t* sum[2]; const float* ori_w; int* ori_drop; };   // orientation assignment (SURF_UPRIGHT = false): integral images, sample weights
// Everything of the detector that belongs to ONE pipeline lane (one stereo pair: two images).  Every kernel of the stage takes a
// LanePair and indexes it with the high bit of its image index -- blockIdx.y (or .z / .x where the kernel counts images there)
// runs over 2 * nimg in a two-pair launch: the detector stages of two consecutive pairs of the stream, on two lanes' buffers, in
// ONE launch each (uvo_stereo_submit, batch mode: the thin kernels -- scans, rank sort, finish passes -- take as long for four
// images as for two).  A single pair is a LanePair whose entries are the same.
struct LaneArgs { ImgPair ip; int32_t* part; SurvOut sv; CandOut out; SortArgs sa; DescArgs da; uint8_t* patch; const int4* big_in; int4* big_out; int* big_large; };
struct LanePair { LaneArgs a[2]; };
#define UVO_LANE_IM(idx) ((int)(idx) & 1)
#define UVO_LANE_OF(lp, idx) ((lp).a[(int)(idx) >> 1])
// The integral image in three launches that move 4 + 4 + 33 MB for a 1080p pair (image twice, result once):
//   k_integral_strip_sums   a workgroup per strip of 8 image rows: row prefixes in registers, their column sums over the strip
//   k_integral_strip_scan   exclusive scan of those sums over the strips (a 135 x 1921 matrix at 1080p)
//   k_integral_strip_final  the row prefixes again, running column sums from the scanned offsets, results to the row-major
//                           image (through LDS, so that rows are written as consecutive words) and to the 16 planes
// Integer arithmetic: any order of additions gives cv::integral's CV_32S values.
static const int kStripRows = 8;
static const int kStripCols = 2048;           // columns of the sum image per pass of a workgroup: 8 per thread

// Row prefixes of the strip's rows at this thread's eight columns c0 .. c0 + 7 of the sum image: E[r] = sum of the pixels of row
// y0 + r left of column c0 (pixel x contributes to sum columns > x); lo/hi = the eight pixels c0 .. c0 + 7 of that row, packed.
// carry[r] (uniform) = the row's total over the previous passes, updated for the next one.
__device__ __forceinline__ void strip_row_prefix(const uint8_t* img, int w, int h, int y0, int c0, int (*wt)[4], int (&carry)[kStripRows],
                                                 unsigned (&lo)[kStripRows], unsigned (&hi)[kStripRows], int (&E)[kStripRows])
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool vec8 = (w & 7) == 0;
#pragma unroll
    for (int r = 0; r < kStripRows; r++) {
        const int y = y0 + r;
        lo[r] = hi[r] = 0;
        if (y < h && c0 < w) {
            const uint8_t* src = img + (size_t)y * w + c0;
            if (vec8) { const uint2 v = *reinterpret_cast<const uint2*>(src); lo[r] = v.x; hi[r] = v.y; }     // w % 8 == 0: rows and c0 are 8-byte aligned
            else {
                for (int k = 0; k < 8; k++) if (c0 + k < w) { if (k < 4) lo[r] |= (unsigned)src[k] << (8 * k); else hi[r] |= (unsigned)src[k] << (8 * (k - 4)); }
            }
        }
    }
    int local[kStripRows], inc[kStripRows];
#pragma unroll
    for (int r = 0; r < kStripRows; r++) inc[r] = local[r] = (int)__builtin_amdgcn_sad_u8(lo[r], 0u, __builtin_amdgcn_sad_u8(hi[r], 0u, 0u));
    // inclusive scan over the wave in six DPP steps: within the rows of 16 lanes, then the rows' totals carried across
#pragma unroll
    for (int r = 0; r < kStripRows; r++) {
        int v = inc[r];
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
        v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
        v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
        v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
        inc[r] = v;
    }
    if (lane == 63) {
#pragma unroll
        for (int r = 0; r < kStripRows; r++) wt[r][wv] = inc[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kStripRows; r++) {
        const int t0 = wt[r][0], t1 = wt[r][1], t2 = wt[r][2], t3 = wt[r][3];
        const int base = carry[r] + (wv > 0 ? t0 : 0) + (wv > 1 ? t1 : 0) + (wv > 2 ? t2 : 0);
        E[r] = base + inc[r] - local[r];
        carry[r] += t0 + t1 + t2 + t3;
    }
}
__device__ __forceinline__ int packed_px(unsigned lo, unsigned hi, int k) { return (int)(((k < 4 ? lo : hi) >> (8 * (k & 3))) & 255u); }

__global__ __launch_bounds__(256) void k_integral_strip_sums(LanePair lp, int w, int h, int cstride, int nstrip)
{
    const LaneArgs& LA = UVO_LANE_OF(lp, blockIdx.y); const ImgPair& ip = LA.ip; int32_t* part = LA.part;
    const int strip = blockIdx.x, im = UVO_LANE_IM(blockIdx.y), tid = threadIdx.x;
    __shared__ int wt[kStripRows][4];
    int carry[kStripRows];
#pragma unroll
    for (int r = 0; r < kStripRows; r++) carry[r] = 0;
    int32_t* dst = part + ((size_t)im * nstrip + strip) * cstride;
    for (int cb = 0; cb <= w; cb += kStripCols) {
        unsigned lo[kStripRows], hi[kStripRows]; int E[kStripRows];
        strip_row_prefix(ip.img[im], w, h, strip * kStripRows, cb + 8 * tid, wt, carry, lo, hi, E);
        int cs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < kStripRows; r++) {
            int run = E[r];
#pragma unroll
            for (int k = 0; k < 8; k++) { cs[k] += run; run += packed_px(lo[r], hi[r], k); }
        }
        int4* d4 = reinterpret_cast<int4*>(dst + cb + 8 * tid);                      // cstride is a multiple of kStripCols: aligned
        d4[0] = make_int4(cs[0], cs[1], cs[2], cs[3]); d4[1] = make_int4(cs[4], cs[5], cs[6], cs[7]);
        __syncthreads();                                                                // wt is rewritten by the next pass
    }
    if (strip == 0 && tid == 0) { ip.cand_n[im] = 0; ip.big_n[im] = 0; if (im == 0) { *ip.surv_n = 0; ip.surv_n[CN_ORI_DROP - CN_SURV] = 0; } }   // the frame's candidate / large-window / survivor counters start at zero
}

// part[im][s][x] <- sum of part[im][0 .. s-1][x]: 32 columns x 8 groups of strips per workgroup
__global__ __launch_bounds__(256) void k_integral_strip_scan(LanePair lp, int cstride, int nstrip, int w)
{
    int32_t* part = UVO_LANE_OF(lp, blockIdx.y).part;
    const int tid = threadIdx.x, xl = tid & 31, g = tid >> 5, x = blockIdx.x * 32 + xl, im = UVO_LANE_IM(blockIdx.y);
    __shared__ int gsum[8][32];
    const int per = (nstrip + 7) / 8, s0 = g * per, s1 = min(nstrip, s0 + per);
    int32_t* p = part + (size_t)im * nstrip * cstride + x;
    const bool live = x <= w;
    int tot = 0;
#pragma unroll 8
    for (int sI = s0; sI < s1; sI++) tot += live ? p[(size_t)sI * cstride] : 0;
    gsum[g][xl] = tot;
    __syncthreads();
    int acc = 0;
    for (int k = 0; k < g; k++) acc += gsum[k][xl];
    if (!live) return;
#pragma unroll 8
    for (int sI = s0; sI < s1; sI++) { const int v = p[(size_t)sI * cstride]; p[(size_t)sI * cstride] = acc; acc += v; }
}

__global__ __launch_bounds__(256) void k_integral_strip_final(LanePair lp, int w, int h, int cstride, int nstrip)
{
    const LaneArgs& LA = UVO_LANE_OF(lp, blockIdx.y); const ImgPair& ip = LA.ip; const int32_t* part = LA.part;
    const int strip = blockIdx.x, im = UVO_LANE_IM(blockIdx.y), tid = threadIdx.x;
    const int sw = w + 1, y0 = strip * kStripRows;
    __shared__ int wt[kStripRows][4];
    __shared__ __align__(16) int rowbuf[kStripRows / 2][kStripCols];
    int carry[kStripRows];
#pragma unroll
    for (int r = 0; r < kStripRows; r++) carry[r] = 0;
    const int32_t* src = part + ((size_t)im * nstrip + strip) * cstride;
    int32_t* sum = ip.sum[im];
    int32_t* planes = ip.planes[im];
    for (int cb = 0; cb <= w; cb += kStripCols) {
        const int c0 = cb + 8 * tid;
        const int4* s4 = reinterpret_cast<const int4*>(src + c0);
        const int4 a0 = s4[0], a1 = s4[1];                                               // S[y0][c0 .. c0 + 7]
        unsigned lo[kStripRows], hi[kStripRows]; int E[kStripRows];
        strip_row_prefix(ip.img[im], w, h, y0, c0, wt, carry, lo, hi, E);
        int acc[8] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w };
#pragma unroll
        for (int half = 0; half < 2; half++) {
#pragma unroll
            for (int rr = 0; rr < kStripRows / 2; rr++) {
                const int r = half * (kStripRows / 2) + rr, ys = y0 + r + 1;               // row of the sum image
                int run = E[r];
#pragma unroll
                for (int k = 0; k < 8; k++) { acc[k] += run; run += packed_px(lo[r], hi[r], k); }
                int4* rb = reinterpret_cast<int4*>(&rowbuf[rr][8 * tid]);
                rb[0] = make_int4(acc[0], acc[1], acc[2], acc[3]); rb[1] = make_int4(acc[4], acc[5], acc[6], acc[7]);
                if (((r + 1) & 3) != 2 && ys <= h && c0 <= w) {        // (y0 is a multiple of 8: ys & 3 == (r + 1) & 3; planes of rows / columns = 2 (mod 4) are never read: plane_written)
                    // planes: sum column x -> plane (ys & 3, x & 3) at [ys >> 2][x >> 2]; this thread's columns give two adjacent entries per plane
                    int32_t* pl = planes + (size_t)((ys & 3) * 4) * ip.pstride + (size_t)(ys >> 2) * ip.pw + (c0 >> 2);
                    if (c0 + 7 <= w) {
#pragma unroll
                        for (int rx = 0; rx < 4; rx++) if (rx != 2) *reinterpret_cast<int2*>(pl + (size_t)rx * ip.pstride) = make_int2(acc[rx], acc[rx + 4]);   // pw, pstride even: 8-byte aligned
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; k++) if ((k & 3) != 2 && c0 + k <= w) pl[(size_t)(k & 3) * ip.pstride + (k >> 2)] = acc[k];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < kStripRows / 2; rr++) {
                const int ys = y0 + half * (kStripRows / 2) + rr + 1;
                if (ys <= h) {
                    int32_t* drow = sum + (size_t)ys * sw + cb;
#pragma unroll
                    for (int j = 0; j < 8; j++) { const int xs = tid + 256 * j; if (cb + xs <= w) drow[xs] = rowbuf[rr][xs]; }
                }
            }
            __syncthreads();
        }
    }
    if (strip == 0) for (int x = tid; x < sw; x += 256) sum[x] = 0;
}

// ------------------------------------------------------------------------------------------
// Hessian layers
// ------------------------------------------------------------------------------------------
struct LayerPat {
    int dx1[10], dy1[10], dx2[10], dy2[10];   // boxes: 0-2 Dx, 3-5 Dy, 6-9 Dxy
    float w[10];
    int size, margin, samples_i, samples_j;   // samples = 0 when the layer does not fit the image
};
struct OctavePat {
    LayerPat L[5];
    int step, rows, cols;
    int lo, hi;                // sum-coordinate extent of all boxes relative to plane_coord*step
    int nms_margin[3];         // for middle layers 1..3
    int octave;
};

// resizeHaarPattern (surf.cpp): coordinates cvRound(ratio*c), weight w/((float)(dx2-dx1)*(dy2-dy1))
// What a detection tile needs of its octave's pattern at run time (the box geometry is compile-time there): by value in the merged
// launch's arguments.  Read from the pattern table in global memory instead, every field was a scalar load at its first use with an
// s_waitcnt lgkmcnt(0) behind it -- 117 of them scattered over the kernel, each also draining the LDS reads in flight.
struct OctDims { struct { int samples_i, samples_j; } L[5]; int rows, cols, nms_margin[3], octave; };
struct HessDims { OctDims o[4]; };
static void make_pattern(int size, LayerPat* lp)
{
    static const int dx_s[3][5]  = { {0, 2, 3, 7, 1}, {3, 2, 6, 7, -2}, {6, 2, 9, 7, 1} };
    static const int dy_s[3][5]  = { {2, 0, 7, 3, 1}, {2, 3, 7, 6, -2}, {2, 6, 7, 9, 1} };
    static const int dxy_s[4][5] = { {1, 1, 4, 4, 1}, {5, 1, 8, 4, -1}, {1, 5, 4, 8, -1}, {5, 5, 8, 8, 1} };
    float ratio = (float)size / 9;
    for (int k = 0; k < 10; k++) {
        const int* s = k < 3 ? dx_s[k] : k < 6 ? dy_s[k - 3] : dxy_s[k - 6];
        int dx1 = cv_round_f(ratio * s[0]), dy1 = cv_round_f(ratio * s[1]);
        int dx2 = cv_round_f(ratio * s[2]), dy2 = cv_round_f(ratio * s[3]);
        lp->dx1[k] = dx1; lp->dy1[k] = dy1; lp->dx2[k] = dx2; lp->dy2[k] = dy2;
        lp->w[k] = s[4] / ((float)(dx2 - dx1) * (dy2 - dy1));
    }
    lp->size = size;
}

static void make_octave(int octave, int nOctaveLayers, int w, int h, OctavePat* op)
{
    int step = 1 << octave;
    op->step = step; op->rows = h / step; op->cols = w / step; op->octave = octave;
    op->lo = 0; op->hi = 0;
    for (int l = 0; l < 5; l++) {
        int size = (9 + 6 * l) << octave;
        LayerPat* lp = &op->L[l];
        make_pattern(size, lp);
        lp->margin = (size / 2) / step;
        bool fits = l < nOctaveLayers + 2 && size <= h && size <= w;
        lp->samples_i = fits ? 1 + (h - size) / step : 0;
        lp->samples_j = fits ? 1 + (w - size) / step : 0;
        int lo = -lp->margin * step, hi = lo + size;
        if (lo < op->lo) op->lo = lo;
        if (hi > op->hi) op->hi = hi;
    }
    for (int m = 0; m < 3; m++) op->nms_margin[m] = (op->L[m + 2].size / 2) / step + 1;
}

// calcHaarPattern x3 -> dx, dy, dxy for the template whose top-left integral sample is S(0,0)
template <class SumAt>
__device__ __forceinline__ void haar_response(const LayerPat& lp, SumAt S, float* pdx, float* pdy, float* pdxy)
{
    double d = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int v = S(lp.dy1[k], lp.dx1[k]) + S(lp.dy2[k], lp.dx2[k]) - S(lp.dy2[k], lp.dx1[k]) - S(lp.dy1[k], lp.dx2[k]);
        d += (float)v * lp.w[k];
    }
    *pdx = (float)d;
    d = 0;
#pragma unroll
    for (int k = 3; k < 6; k++) {
        int v = S(lp.dy1[k], lp.dx1[k]) + S(lp.dy2[k], lp.dx2[k]) - S(lp.dy2[k], lp.dx1[k]) - S(lp.dy1[k], lp.dx2[k]);
        d += (float)v * lp.w[k];
    }
    *pdy = (float)d;
    if (!pdxy) return;                     // the Laplacian sign needs dx + dy only
    d = 0;
#pragma unroll
    for (int k = 6; k < 10; k++) {
        int v = S(lp.dy1[k], lp.dx1[k]) + S(lp.dy2[k], lp.dx2[k]) - S(lp.dy2[k], lp.dx1[k]) - S(lp.dy1[k], lp.dx2[k]);
        d += (float)v * lp.w[k];
    }
    *pdxy = (float)d;
}

// Matx33f::solve(b, DECOMP_LU) == Cramer's rule in float (Matx_FastSolveOp<float,3,3,1>)
__device__ __forceinline__ void solve3f(const float a[3][3], const float b[3], float x[3])
{
    float d = (float)(double)(a[0][0]*(a[1][1]*a[2][2] - a[2][1]*a[1][2]) -
                              a[0][1]*(a[1][0]*a[2][2] - a[2][0]*a[1][2]) +
                              a[0][2]*(a[1][0]*a[2][1] - a[2][0]*a[1][1]));
    if (d == 0) { x[0] = x[1] = x[2] = 0; return; }
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
}


// findMaximaInLayer's tail for one 3x3x3 maximum: centre, interpolateKeypoint, SURFInvoker's size check
// (the caller appends).  `trace` is dx + dy of the centre sample.
template <int STEP>
__device__ __forceinline__ bool make_keypoint(float N9[3][9], float val0, float trace, int i, int j, int size, int ds, int octave,
                                              int w, int h, uvo_keypoint* pkp)
{
    int sum_i = STEP * (i - (size / 2) / STEP);
    int sum_j = STEP * (j - (size / 2) / STEP);
    float center_i = sum_i + (size - 1) * 0.5f;
    float center_j = sum_j + (size - 1) * 0.5f;
    float bb[3] = { -(N9[1][5]-N9[1][3])/2, -(N9[1][7]-N9[1][1])/2, -(N9[2][4]-N9[0][4])/2 };
    float A[3][3] = {
        { N9[1][3]-2*N9[1][4]+N9[1][5], (N9[1][8]-N9[1][6]-N9[1][2]+N9[1][0])/4, (N9[2][5]-N9[2][3]-N9[0][5]+N9[0][3])/4 },
        { (N9[1][8]-N9[1][6]-N9[1][2]+N9[1][0])/4, N9[1][1]-2*N9[1][4]+N9[1][7], (N9[2][7]-N9[2][1]-N9[0][7]+N9[0][1])/4 },
        { (N9[2][5]-N9[2][3]-N9[0][5]+N9[0][3])/4, (N9[2][7]-N9[2][1]-N9[0][7]+N9[0][1])/4, N9[0][4]-2*N9[1][4]+N9[2][4] } };
    float x[3];
    solve3f(A, bb, x);
    bool ok = (x[0] != 0 || x[1] != 0 || x[2] != 0) &&
              fabsf(x[0]) <= 1 && fabsf(x[1]) <= 1 && fabsf(x[2]) <= 1;
    if (!ok) return false;
    uvo_keypoint kp;
    kp.x = center_j + x[0] * STEP;
    kp.y = center_i + x[1] * STEP;
    kp.size = (float)cv_round_f((float)size + x[2] * ds);
    kp.angle = 360.f - 90.f;            // upright: descriptor_dir
    kp.response = val0;
    kp.octave = octave;
    kp.class_id = (trace > 0) - (trace < 0);
    // SURFInvoker: keypoints whose gradient wavelet exceeds the integral image are dropped
    float s = kp.size * 1.2f / 9.0f;
    int grad_wav_size = 2 * cv_round_f(2 * s);
    if (h + 1 < grad_wav_size || w + 1 < grad_wav_size) return false;
    *pkp = kp;
    return true;
}

// ------------------------------------------------------------------------------------------
// Octaves 0 and 1: one workgroup computes TW x TH plane samples (1-sample halo, (TW-2) x (TH-2) NMS outputs) of the three
// middle layers from an integral tile in LDS (the outer layers 0 and 4 are evaluated lazily, see nms_survivors /
// k_hessian_finish below), with the box patterns, weights and every LDS offset
// resolved at compile time (sizes (9+6l)<<O are fixed by the octave), so one box corner is one
// ds_read_b32 with an immediate offset and corners shared between the boxes of a filter are read
// once (32 reads per sample instead of 40, no address arithmetic).  For STEP > 1 the integral tile
// is stored de-interleaved by column residue, which turns the stride-STEP sample walk into
// unit-stride (conflict-free) LDS reads.
// ------------------------------------------------------------------------------------------
static const float kDetBelow = -3.0e38f;      // sdet marker: determinant not evaluated, known to be <= the threshold
constexpr int cround_pos(float v)      // cvRound for v >= 0 (round half to even), usable in constant expressions
{
    int i = (int)v;
    float f = v - (float)i;
    return f > 0.5f ? i + 1 : (f < 0.5f ? i : ((i & 1) ? i + 1 : i));
}
template <int SIZE>
struct LayerC {
    static constexpr float ratio = (float)SIZE / 9;
    static constexpr int r(int c) { return cround_pos(ratio * (float)c); }
    static constexpr float wt(int wgt, int x1, int y1, int x2, int y2)
    {
        return (float)wgt / ((float)(r(x2) - r(x1)) * (float)(r(y2) - r(y1)));
    }
};
template <int O>
struct OctC {
    static constexpr int STEP = 1 << O;
    static constexpr int size(int l) { return (9 + 6 * l) << O; }
    static constexpr int margin(int l) { return (size(l) / 2) / STEP; }
    // extent of the integral tile around a sample: the largest layer a detection workgroup evaluates is layer 3 (the outer
    // layers 0 and 4 are evaluated by k_hessian_finish from the global integral), and it reaches furthest both ways
    static constexpr int LO = -margin(3) * STEP;
    static constexpr int HI = -margin(3) * STEP + size(3);
};

// Integral tile of a TW-sample-wide workgroup: TWs columns, padded to whole quads (TWq), PW words per column-residue plane.
template <int O, int TW>
struct OctTile {
    static constexpr int TWs = (TW - 1) * OctC<O>::STEP + (OctC<O>::HI - OctC<O>::LO) + 1;
    static constexpr int TWq = (TWs + 3) & ~3;
    static constexpr int PW = TWq / OctC<O>::STEP;
};
struct __attribute__((packed, aligned(4))) SumQuad { int32_t a, b, c, d; };     // four integral columns, 4-byte aligned

// dx, dy, dxy -> det of one sample; SV(dy, dx) fetches the integral value at compile-time offset (dy, dx) from the
// template's top-left corner.  Corners shared between the boxes of a filter are read once (32 reads per sample).
// `skip_thr` (in scope at the expansion): when dx*dy <= skip_thr the determinant cannot exceed the threshold either
// (det = fl(dx*dy - fl(0.81f*dxy*dxy)) <= dx*dy), so Dxy's 16 corners are not read and the sample gets kDetBelow: it can
// neither be a maximum nor beat one, and k_hessian_finish evaluates it exactly if it ends up in a keypoint's neighbourhood.
// Arithmetic of one box, restated for the instruction mix of gfx950 (tools/probe/issue_rate_probe.hip: v_cvt_f32_i32 7.9 cycles per
// wave-instruction, v_cvt_f64_f32 8.6, v_cvt_f32_f64 8.4 against 4.8-5.4 for adds, subtractions, v_sad_u32 and v_pk_fma_f32 with its
// two products) -- OpenCV's value `(float)v * w` for the box sum v, bit for bit:
//   * v is a sum of pixels, 0 <= v < 2^23 (the largest box, 88 x 147 pixels of 255, is 3.3e6), so the float whose BITS are
//     v + 0x4B000000 is exactly 2^23 + v, and fma(2^23 + v, w, -(2^23 w)) rounds the exact product v w once: fl(v w) = (float)v * w
//     (2^23 w is a power-of-two multiple of w, exact; a zero box gives +0 either way: w0 > 0 leads every sum below).  No int -> float
//     conversion, and two boxes of equal weight share one v_pk_fma_f32.
//   * the four corners of a box are two differences along the direction the boxes of a filter share an edge in: for Dx
//     g_k = S(y7, x_k) - S(y2, x_k) >= 0 at the four x_k, and box_j + 2^23-bits = (g_{k+1} + (j+1) M) - (g_k + j M) with
//     M = 0x4B000000 folded into v_sad_u32's addend (|a - b| + c; the integral is monotone, so |a - b| = a - b): seven integer
//     instructions per filter instead of nine plus a constant add, twelve for Dxy.
//   * the double accumulation `d = 0; d += p0; ...` starts from p0 itself (0 + p0 = p0 exactly, p0 >= +0).
// The sums of three or four such floats in double are exact (their exponents lie within 24 binades), so the order is immaterial to
// the result; it is kept anyway.
typedef float uvo_v2f __attribute__((ext_vector_type(2)));
static constexpr unsigned kBoxMagic = 0x4B000000u;           // bits of 2^23
__device__ __forceinline__ unsigned sad_u32(int a, int b, unsigned c)      // |a - b| + c (mod 2^32), a, b >= 0
{
    unsigned r;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
#define UVO_BOX_FMA2(xa, xb, wgt) __builtin_elementwise_fma(uvo_v2f{__uint_as_float(xa), __uint_as_float(xb)}, uvo_v2f{(wgt), (wgt)}, uvo_v2f{-8388608.0f * (wgt), -8388608.0f * (wgt)})
// threadIdx.x behind an opaque (empty) volatile asm: what a tile's code derives from it stays inside the merged launch's tile loop instead of
// being hoisted out of it as loop-invariant (all three kinds' tile-invariant addresses live at once: 111-156 VGPRs instead of 58)
__device__ __forceinline__ int lane_tid() { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; }
// first half: dx * dy of one sample and layer (Dx and Dy: sixteen corners)
#define UVO_HESSIAN_PP(SV, pp_out)                                                                        \
    {                                                                                                     \
        /* Dx: boxes (0,2,3,7,+1) (3,2,6,7,-2) (6,2,9,7,+1): column differences between rows c2 and c7 */ \
        const unsigned gx0 = (unsigned)(SV(c7, c0) - SV(c2, c0)), gx3 = sad_u32(SV(c7, c3), SV(c2, c3), kBoxMagic),                       \
                       gx6 = sad_u32(SV(c7, c6), SV(c2, c6), 2u * kBoxMagic), gx9 = sad_u32(SV(c7, c9), SV(c2, c9), 3u * kBoxMagic);     \
        /* Dy: boxes (2,0,7,3,+1) (2,3,7,6,-2) (2,6,7,9,+1): row differences between columns c2 and c7 */ \
        const unsigned gy0 = (unsigned)(SV(c0, c7) - SV(c0, c2)), gy3 = sad_u32(SV(c3, c7), SV(c3, c2), kBoxMagic),                       \
                       gy6 = sad_u32(SV(c6, c7), SV(c6, c2), 2u * kBoxMagic), gy9 = sad_u32(SV(c9, c7), SV(c9, c2), 3u * kBoxMagic);     \
        static_assert(wx0 == wx2 && wy0 == wy2 && wx0 == wy0 && wx1 == wy1, "the outer boxes of Dx and Dy share one weight");            \
        const uvo_v2f px02 = UVO_BOX_FMA2(gx3 - gx0, gx9 - gx6, wx0), py02 = UVO_BOX_FMA2(gy3 - gy0, gy9 - gy6, wy0);                     \
        const uvo_v2f pxy1 = UVO_BOX_FMA2(gx6 - gx3, gy6 - gy3, wx1);                                                                     \
        double d = (double)px02.x;                                                                        \
        d += (double)pxy1.x;                                                                              \
        d += (double)px02.y;                                                                              \
        const float dx = (float)d;                                                                        \
        d = (double)py02.x;                                                                               \
        d += (double)pxy1.y;                                                                              \
        d += (double)py02.y;                                                                              \
        const float dy = (float)d;                                                                        \
        pp_out = dx * dy;                                                                                 \
    }
// second half: Dxy (sixteen more corners) and the determinant, for a sample whose dx * dy can still beat the threshold
#define UVO_HESSIAN_DXY(SV, pp_in, det)                                                                   \
    {                                                                                                     \
        /* Dxy: boxes (1,1,4,4,+1) (5,1,8,4,-1) (1,5,4,8,-1) (5,5,8,8,+1): row differences c1..c4 and c5..c8 at the four columns */      \
        const unsigned u1 = (unsigned)(SV(c4, c1) - SV(c1, c1)), u4 = sad_u32(SV(c4, c4), SV(c1, c4), kBoxMagic);                         \
        const unsigned u5 = (unsigned)(SV(c4, c5) - SV(c1, c5)), u8 = sad_u32(SV(c4, c8), SV(c1, c8), kBoxMagic);                         \
        const unsigned t1 = (unsigned)(SV(c8, c1) - SV(c5, c1)), t4 = sad_u32(SV(c8, c4), SV(c5, c4), kBoxMagic);                         \
        const unsigned t5 = (unsigned)(SV(c8, c5) - SV(c5, c5)), t8 = sad_u32(SV(c8, c8), SV(c5, c8), kBoxMagic);                         \
        static_assert(wd0 == wd3 && wd1 == wd2 && wd1 == -wd0, "the four boxes of Dxy share one weight up to sign");                      \
        const uvo_v2f pd03 = UVO_BOX_FMA2(u4 - u1, t8 - t5, wd0), pd12 = UVO_BOX_FMA2(u8 - u5, t4 - t1, wd1);                             \
        double d = (double)pd03.x;                                                                        \
        d += (double)pd12.x;                                                                              \
        d += (double)pd12.y;                                                                              \
        d += (double)pd03.y;                                                                              \
        const float dxy = (float)d;                                                                       \
        det = (pp_in) - 0.81f * dxy * dxy;                                                                \
    }
// (A variant that took Dy only where |dx| * 512 > threshold -- a wave whose 64 samples all fall under that bound skips Dy's eight
// corners -- was measured and removed: on the bench's scene no wave qualifies and the split cost 0.9 us of the launch)
#define UVO_HESSIAN_DET(SV, det)                                                                          \
    {                                                                                                     \
        float pp_;                                                                                        \
        UVO_HESSIAN_PP(SV, pp_)                                                                           \
        if (!(pp_ > skip_thr)) det = kDetBelow; else UVO_HESSIAN_DXY(SV, pp_, det)                        \
    }
#define UVO_HESSIAN_CONSTS(LC)                                                                            \
    constexpr int c0 = LC::r(0), c1 = LC::r(1), c2 = LC::r(2), c3 = LC::r(3), c4 = LC::r(4), c5 = LC::r(5), \
                  c6 = LC::r(6), c7 = LC::r(7), c8 = LC::r(8), c9 = LC::r(9);                             \
    constexpr float wx0 = LC::wt(1, 0, 2, 3, 7), wx1 = LC::wt(-2, 3, 2, 6, 7), wx2 = LC::wt(1, 6, 2, 9, 7); \
    constexpr float wy0 = LC::wt(1, 2, 0, 7, 3), wy1 = LC::wt(-2, 2, 3, 7, 6), wy2 = LC::wt(1, 2, 6, 7, 9); \
    constexpr float wd0 = LC::wt(1, 1, 1, 4, 4), wd1 = LC::wt(-1, 5, 1, 8, 4), wd2 = LC::wt(-1, 1, 5, 4, 8), wd3 = LC::wt(1, 5, 5, 8, 8); \
    (void)c0; (void)c9

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (N > 0) { static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

// A corner read from the tile.  As plain loads the compiler pairs corners of one tile row into ds_read2_b32 -- whose two offsets reach
// 1 KB, three tile rows -- behind a v_add_u32 that forms the pair's base: eight address additions per determinant on the unit that
// bounds the kernel.  Volatile LDS loads are not paired (each a ds_read_b32 with its 16-bit offset from the sample's one base register):
// 9 % fewer vector instructions in the octave-0 loop, twice the LDS instructions, 64.5 against 65.0 us -- measured, not kept.
#define UVO_SV_LOAD(base, off) ((base)[off])
// det plane of layer L (1..3) of a workgroup's TW x TH samples from the integral tile in LDS; 0 where the template does not fit
template <int O, int L, int TW, int TH, int NT, class OP>
__device__ __forceinline__ void det_layer_c(const int32_t* __restrict__ stile, float* __restrict__ sdet, const OP& op,
                                            int px0, int py0, float skip_thr)
{
    using OC = OctC<O>;
    constexpr int STEP = OC::STEP, SIZE = OC::size(L);
    using LC = LayerC<SIZE>;
    constexpr int PW = OctTile<O, TW>::PW;
    constexpr int OFFL = -OC::margin(L) * STEP - OC::LO;          // tile offset of this layer's template origin
    static_assert(OFFL >= 0, "layer origin outside the tile");
    // tile index of corner (dy, dx) relative to base = &stile[ry*STEP*STEP*PW + rx]
#define SV(dy, dx) UVO_SV_LOAD(base, ((OFFL + (dy)) * STEP + ((OFFL + (dx)) % STEP)) * PW + (OFFL + (dx)) / STEP)
    UVO_HESSIAN_CONSTS(LC);
    const auto& lp = op.L[L];
    const int tid = lane_tid();
    // a compile-time trip count (the last pass is guarded): the box arithmetic holds inline assembly, which the compiler treats as
    // convergent and will not unroll behind a run-time remainder
#pragma unroll
    for (int it = 0; it < (TW * TH + NT - 1) / NT; it++) {
        const int sidx = tid + it * NT;
        if ((it + 1) * NT > TW * TH && sidx >= TW * TH) break;
        const int ry = sidx / TW, rx = sidx - ry * TW;
        const int oi = py0 + ry - OC::margin(L), oj = px0 + rx - OC::margin(L);
        float det = 0.f;
        if (oi >= 0 && oi < lp.samples_i && oj >= 0 && oj < lp.samples_j) {
            const int32_t* base = stile + ry * (STEP * STEP * PW) + rx;
            UVO_HESSIAN_DET(SV, det)
        }
        sdet[((L - 1) * TH + ry) * TW + rx] = det;
    }
#undef SV
}

// The box corners of octaves 2 and 3 (filters 27 .. 99 and 51 .. 195 pixels at steps 4 and 8, layers 1 - 3) only ever fall on rows and
// columns of the integral with index = 0, 1 or 3 (mod 4): nine of the sixteen de-interleaved planes are read, and k_integral_strip_final
// writes only those (7 MB less per pair).  Every plane read goes through this check at compile time.
template <int DY, int DX> __device__ __forceinline__ constexpr int plane_written()
{
    static_assert((DY & 3) != 2 && (DX & 3) != 2, "this corner lies in a plane that k_integral_strip_final does not write");
    return 0;
}
typedef __amdgpu_buffer_rsrc_t ImgRsrc;
__device__ __forceinline__ ImgRsrc img_rsrc(const uint8_t* img, int bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, bytes, 0x00020000);      // raw buffer, dword data format (gfx9 family)
}

// The same layer from the de-interleaved planes in global memory (octaves 2 and 3: the integral tile of one
// workgroup would not fit LDS).  Sample (oi, oj) has its top-left at pixel (STEP*oi, STEP*oj); corner (dy, dx) is in
// plane (dy & 3, dx & 3) at [STEP/4*oi + dy/4][STEP/4*oj + dx/4]: plane and offsets are compile-time, lanes along oj.
template <int O, int L, int TW, int TH, int NT, class OP>
__device__ __forceinline__ void det_layer_p(const int32_t* __restrict__ planes, int pw, int pstride, float* __restrict__ sdet,
                                            const OP& op, int px0, int py0, float skip_thr)
{
    using OC = OctC<O>;
    constexpr int STEP = OC::STEP, SIZE = OC::size(L), Q = STEP / 4;
    static_assert(STEP % 4 == 0, "plane variant needs a step that is a multiple of 4");
    using LC = LayerC<SIZE>;
    // The planes are read through a buffer resource: a corner's address is the lane's sample offset (one VGPR) + a scalar offset
    // (plane and row of the corner: SALU arithmetic on pw / pstride) + the column as the instruction's immediate, instead of a
    // 64-bit flat address (a VGPR pair and a v_lshl_add_u64) per corner.
    const ImgRsrc rs = img_rsrc(reinterpret_cast<const uint8_t*>(planes), 16 * pstride * 4);
#define SVP(dy, dx) (int)__builtin_amdgcn_raw_buffer_load_b32(rs, voff + 4 * ((dx) >> 2) + plane_written<(dy), (dx)>(), 4 * ((((dy) & 3) * 4 + ((dx) & 3)) * pstride + ((dy) >> 2) * pw), 0)
    UVO_HESSIAN_CONSTS(LC);
    const auto& lp = op.L[L];
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < (TW * TH + NT - 1) / NT; it++) {
        const int sidx = tid + it * NT;
        if ((it + 1) * NT > TW * TH && sidx >= TW * TH) break;
        const int ry = sidx / TW, rx = sidx - ry * TW;
        const int oi = py0 + ry - OC::margin(L), oj = px0 + rx - OC::margin(L);
        float det = 0.f;
        if (oi >= 0 && oi < lp.samples_i && oj >= 0 && oj < lp.samples_j) {
            const int voff = 4 * ((oi * Q) * pw + oj * Q);
            UVO_HESSIAN_DET(SVP, det)
        }
        sdet[((L - 1) * TH + ry) * TW + rx] = det;
    }
#undef SVP
}

// The three middle layers of a plane-octave sample in TWO memory round trips instead of six (the merged launch): first the sixteen
// Dx / Dy corners of all three layers -- 48 independent buffer loads per lane, issued together -- then, for the layers whose dx * dy
// can still beat the threshold, their Dxy corners together.  A plane tile used to live ~13 us in its workgroup slot, almost all of it
// waiting for six dependent batches of loads.  The loads are unconditional (an out-of-range sample's address is out of the buffer's
// range: the resource returns 0) and the in-range test selects afterwards.
template <int O, int L>
__device__ __forceinline__ float plane_pp(ImgRsrc rs, int voff, int pw, int pstride)
{
    using LC = LayerC<OctC<O>::size(L)>;
#define SVP(dy, dx) (int)__builtin_amdgcn_raw_buffer_load_b32(rs, voff + 4 * ((dx) >> 2) + plane_written<(dy), (dx)>(), 4 * ((((dy) & 3) * 4 + ((dx) & 3)) * pstride + ((dy) >> 2) * pw), 0)
    UVO_HESSIAN_CONSTS(LC);
    float pp;
    UVO_HESSIAN_PP(SVP, pp)
    return pp;
}
template <int O, int L>
__device__ __forceinline__ float plane_det(ImgRsrc rs, int voff, int pw, int pstride, float pp)
{
    using LC = LayerC<OctC<O>::size(L)>;
    UVO_HESSIAN_CONSTS(LC);
    float det;
    UVO_HESSIAN_DXY(SVP, pp, det)
#undef SVP
    return det;
}
template <int O, int TW, int TH, int NT, class OP>
__device__ __forceinline__ void det_layers_p3(const int32_t* __restrict__ planes, int pw, int pstride, float* __restrict__ sdet,
                                              const OP& op, int px0, int py0, float skip_thr)
{
    using OC = OctC<O>;
    constexpr int Q = OC::STEP / 4;
    const ImgRsrc rs = img_rsrc(reinterpret_cast<const uint8_t*>(planes), 16 * pstride * 4);
    const int tid = lane_tid();
#pragma unroll
    for (int it = 0; it < (TW * TH + NT - 1) / NT; it++) {
        const int sidx = tid + it * NT;
        if ((it + 1) * NT > TW * TH && sidx >= TW * TH) break;
        const int ry = sidx / TW, rx = sidx - ry * TW;
        bool ok[3]; int voff[3]; float pp[3], det[3];
#pragma unroll
        for (int l = 0; l < 3; l++) {
            const int m = l == 0 ? OC::margin(1) : l == 1 ? OC::margin(2) : OC::margin(3);
            const auto& lp = op.L[l + 1];
            const int oi = py0 + ry - m, oj = px0 + rx - m;
            ok[l] = oi >= 0 && oi < lp.samples_i && oj >= 0 && oj < lp.samples_j;
            voff[l] = ok[l] ? 4 * ((oi * Q) * pw + oj * Q) : 0x7FFFFFF0;      // past the buffer: its loads return 0
        }
        pp[0] = plane_pp<O, 1>(rs, voff[0], pw, pstride);
        pp[1] = plane_pp<O, 2>(rs, voff[1], pw, pstride);
        pp[2] = plane_pp<O, 3>(rs, voff[2], pw, pstride);
        const bool n0 = ok[0] && pp[0] > skip_thr, n1 = ok[1] && pp[1] > skip_thr, n2 = ok[2] && pp[2] > skip_thr;
        det[0] = ok[0] ? kDetBelow : 0.f; det[1] = ok[1] ? kDetBelow : 0.f; det[2] = ok[2] ? kDetBelow : 0.f;
        if (n0 || n1 || n2) {
            // (a layer that does not need its Dxy reads past the buffer instead: one instruction stream, no divergence between the layers)
            const float d0 = plane_det<O, 1>(rs, n0 ? voff[0] : 0x7FFFFFF0, pw, pstride, pp[0]);
            const float d1 = plane_det<O, 2>(rs, n1 ? voff[1] : 0x7FFFFFF0, pw, pstride, pp[1]);
            const float d2 = plane_det<O, 3>(rs, n2 ? voff[2] : 0x7FFFFFF0, pw, pstride, pp[2]);
            if (n0) det[0] = d0;
            if (n1) det[1] = d1;
            if (n2) det[2] = d2;
        }
#pragma unroll
        for (int l = 0; l < 3; l++) sdet[(l * TH + ry) * TW + rx] = det[l];
    }
}

// ------------------------------------------------------------------------------------------
// Non-maximum suppression with lazy outer layers, in two kernels.  findMaximaInLayer needs, for each middle layer
// L = 1..3, the 3 x 3 x 3 neighbourhood in layers L-1, L, L+1.  Layers 1..3 are centres and are computed for every
// sample (`sdet`, three planes); layers 0 and 4 only ever appear as neighbours, so they are evaluated just around the
// samples that already beat their threshold, their eight in-layer neighbours and the adjacent layers held in sdet:
// a few thousand per frame instead of 2/5 of all determinants.  The detection kernels append those survivors, with the
// neighbourhood values they have, to a global list; k_hessian_finish (every octave in one launch) evaluates the missing
// nine determinants, makes the last comparison and emits the keypoint.  Keeping the interpolation / Laplacian-sign code
// out of the detection kernels also keeps them at ~60 VGPRs.  The comparisons made are the same ones, so the keypoints
// are the same.
// ------------------------------------------------------------------------------------------

// LDS of nms_survivors: the workgroup's survivor list, its length and its base in the global list.  Survivors are strict 3 x 3
// maxima of their own layer, so no two are adjacent: at most ceil((TW-2)/2) x ceil((TH-2)/2) per layer, in three layers.
template <int TW, int TH>
struct NmsLds { static constexpr int kList = 3 * ((TW - 1) / 2) * ((TH - 1) / 2), kWords = kList + 2; };

// UVO_HESS_STAMPS=<file> (measurement): wall-clock stamps (100 MHz) of every workgroup's phases in the last detection launch --
// start, integral tile in LDS, box sums done, end -- with the tile kind, written as CSV when the context goes
// The stamps exist in the measurement build only (make STAMPS=1 -> lib_ab/libuvo_hip_stamps.so, tools/probe/gpu.sh stamps): reading the
// pointer costs a scalar load and an s_waitcnt lgkmcnt(0) per stamp, which also drains the LDS reads in flight -- the descriptor launch
// measured 66 us with its stamps compiled in and switched off, 58 us without them.
#ifndef UVO_STAMPS
#define UVO_STAMPS 0
#endif
__device__ long long* g_hess_stamps = nullptr;
__device__ __forceinline__ void hess_stamp(int k, long long v = -1)
{
#if UVO_STAMPS
    long long* st = g_hess_stamps;
    if (st && threadIdx.x == 0) st[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + k] = v >= 0 ? v : (long long)wall_clock64();
#endif
}
// the survivor count of a tile starts at zero: called before the barrier that precedes nms_survivors (one barrier less per tile: under
// three workgroups per CU a barrier costs the wait for the slowest of eight time-sliced waves)
template <int TW, int TH>
__device__ __forceinline__ void nms_zero(unsigned* s_list) { if (lane_tid() == 0) reinterpret_cast<int*>(s_list + NmsLds<TW, TH>::kList)[0] = 0; }
template <int TW, int TH, int NT, class OP>
__device__ __forceinline__ void nms_survivors(const float* __restrict__ sdet, unsigned* __restrict__ s_list, const OP& op, float thr,
                                              int px0, int py0, int im, const SurvOut& sv)
{
    const int tid = lane_tid();
    int* s_n = reinterpret_cast<int*>(s_list + NmsLds<TW, TH>::kList);       // [0] count (zeroed by the caller before its last barrier: nms_zero), [1] base
    // Strict 3 x 3 x 3 maxima by separable maxima instead of up to 26 comparisons per candidate behind a divergent branch (in-kernel
    // stamps: the scan was 1.9 of an octave-0 tile's 11 us, the wave executing the whole neighbourhood test for every sample position
    // at which ANY of its lanes beat the threshold).  Thread tid owns column tid % TW (TW a power of two) and ITER consecutive rows:
    // it reads its column's ITER + 2 values of every layer once (LDS, immediate offsets), takes the vertical 3-maxima, and gets its
    // left and right neighbours' from the adjacent lanes (DPP wave shifts: adjacent lanes are adjacent columns).  val0 is a strict
    // maximum of its layer iff it exceeds max(above, below, left column's 3-max, right column's 3-max), and of an adjacent layer
    // iff it exceeds max(own, left, right column's 3-max) there: the same comparisons, so the same survivors.
    static_assert((TW & (TW - 1)) == 0 && NT % TW == 0 && TW <= 64, "tile width: a power of two that divides the workgroup, a row within a wave");
    constexpr int G = NT / TW, ITER = (TH - 2 + G - 1) / G;
    const int rx = tid & (TW - 1), g = tid / TW;
    const int j = px0 + rx;
    float col[3][ITER + 2];
#pragma unroll
    for (int l = 0; l < 3; l++)
#pragma unroll
        for (int k = 0; k < ITER + 2; k++) col[l][k] = sdet[(l * TH + min(g * ITER + k, TH - 1)) * TW + rx];
    // A wave none of whose centre values beats the threshold has no survivor to find: it skips the maxima altogether (wave-uniform
    // branch; 67.6 -> 64.8 us at C3.  Deciding layer by layer -- a layer's tests only where it has a candidate, its 3-maxima only where
    // a neighbouring layer's tests need them -- measured the same, 64.9.)
    float cmax = col[0][1];
#pragma unroll
    for (int l = 0; l < 3; l++)
#pragma unroll
        for (int k = 0; k < ITER; k++) cmax = fmaxf(cmax, col[l][k + 1]);
    if (__any(cmax > thr)) {
    float m3[3][ITER], m3l[3][ITER], m3r[3][ITER];
#pragma unroll
    for (int l = 0; l < 3; l++)
#pragma unroll
        for (int k = 0; k < ITER; k++) {
            const float v = fmaxf(fmaxf(col[l][k], col[l][k + 1]), col[l][k + 2]);
            m3[l][k] = v;
            // value of lane - 1 / lane + 1 (wave_shr:1 / wave_shl:1; a lane without a source keeps its own value: halo columns only)
            m3l[l][k] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xF, 0xF, false));
            m3r[l][k] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130, 0xF, 0xF, false));
        }
#pragma unroll
    for (int l = 0; l < 3; l++) {
        const int L = l + 1;
        const int m = op.nms_margin[l];
        const bool col_ok = op.L[L].samples_i != 0 && op.L[L + 1].samples_i != 0 && rx >= 1 && rx <= TW - 2 && j >= m && j < op.cols - m;
#pragma unroll
        for (int k = 0; k < ITER; k++) {
            const int ry = g * ITER + k + 1, i = py0 + ry;
            const float val0 = col[l][k + 1];
            float nb = fmaxf(fmaxf(col[l][k], col[l][k + 2]), fmaxf(m3l[l][k], m3r[l][k]));           // the eight in-layer neighbours
            if (L <= 2) nb = fmaxf(nb, fmaxf(m3[l + 1][k], fmaxf(m3l[l + 1][k], m3r[l + 1][k])));      // the nine of the layer above
            if (L >= 2) nb = fmaxf(nb, fmaxf(m3[l - 1][k], fmaxf(m3l[l - 1][k], m3r[l - 1][k])));      // the nine of the layer below
            const bool is_max = col_ok && ry <= TH - 2 && i >= m && i < op.rows - m && val0 > thr && val0 > nb;
            if (is_max) s_list[atomicAdd(&s_n[0], 1)] = (unsigned)L | ((unsigned)ry << 8) | ((unsigned)rx << 16);
        }
    }
    }
    hess_stamp(5);
    __syncthreads();
    hess_stamp(6);
    const int nloc = s_n[0];
    // one global atomic per workgroup (a device-scope increment of one address costs ~7 ns chip-wide, and its wave a round trip)
    if (nloc == 0) return;
    if (tid == 0) s_n[1] = atomicAdd(sv.count, nloc);
    __syncthreads();
    hess_stamp(7);
    const int base = s_n[1];
    // a record is 32 words (im, octave, L, i, j, 27 neighbourhood values): 32 lanes write one word each (one thread per survivor
    // writing its 128 bytes took ~1 us of a tile's life)
    for (int e = tid; e < nloc * 32; e += NT) {
        const int t = e >> 5, wd = e & 31;
        if (base + t >= sv.cap) continue;             // k_hessian_finish turns the overflow into the capacity error
        Survivor* r = sv.list + base + t;
        const unsigned ent = s_list[t];
        const int L = ent & 255u, ry = (ent >> 8) & 255u, rx = ent >> 16;
        int val;
        if (wd < 5) val = wd == 0 ? im : wd == 1 ? op.octave : wd == 2 ? L : wd == 3 ? py0 + ry : px0 + rx;
        else {
            const int q = wd - 5, row = q / 9, nb9 = q - row * 9;                  // row 0: layer L-1, 1: L, 2: L+1
            const float* d = sdet + ((L - 2 + row) * TH + ry + nb9 / 3 - 1) * TW + rx + nb9 % 3 - 1;
            const bool have = (row == 0 && L >= 2) || row == 1 || (row == 2 && L <= 2);
            val = have ? __float_as_int(*d) : 0;
        }
        reinterpret_cast<int*>(r)[wd] = val;
    }
}

// Sixteen lanes per survivor, 64 survivors per workgroup: lanes 0..8 evaluate the nine determinants of the outer layer (0
// below layer 1, 4 above layer 3) from the integral image with the run-time box patterns (calcLayerDetAndTrace's
// arithmetic, as k_hessian_layer_debug); lane 0 makes the comparison against them and runs findMaximaInLayer's tail.  The
// workgroup's keypoints are collected in LDS and appended with one atomic per image.
static const int kFinishPerWg = 16;
__global__ __launch_bounds__(256) void k_hessian_finish(LanePair lp, const OctavePat* __restrict__ ops, int w, int h)
{
    const LaneArgs& LA = lp.a[blockIdx.y]; const SurvOut& sv = LA.sv; const ImgPair& ip = LA.ip; const CandOut& out = LA.out;
    __shared__ uvo_keypoint s_kp[2][kFinishPerWg];
    __shared__ int s_cnt[2], s_base[2];
    const int grp = threadIdx.x >> 4, k = threadIdx.x & 15;
    const int n = min(*sv.count, sv.cap);
    if (blockIdx.x == 0 && threadIdx.x == 0 && *sv.count > sv.cap) atomicAdd(&out.count[0], out.cap + 1);     // more survivors than the list holds: reported as the keypoint capacity error
    if (blockIdx.x * kFinishPerWg >= n) return;
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int sw = w + 1;
#pragma unroll 1
    for (int rd = 0; rd < kFinishPerWg / 16; rd++) {
        const int t = blockIdx.x * kFinishPerWg + rd * 16 + grp;
        if (t >= n) continue;                            // whole groups skip together
        const Survivor* r = sv.list + t;
        const int im = r->im, L = r->L, i = r->i, j = r->j;
        const OctavePat& op = ops[r->octave];
        const int step = op.step;
        const int32_t* __restrict__ gsum = ip.sum[im];
        // Row r of the neighbourhood is layer L-1+r.  The outer row (layer 0 below L = 1, layer 4 above L = 3) was never evaluated;
        // the other rows come from the detection kernel and may hold kDetBelow where dx*dy <= threshold made the determinant
        // irrelevant to the comparisons -- its exact value still enters the interpolation, so those are evaluated here too.
        auto exact_det = [&](const LayerPat& lo, int ii, int jj) {
            const int oi = ii - lo.margin, oj = jj - lo.margin;
            float d = 0.f;
            if (oi >= 0 && oi < lo.samples_i && oj >= 0 && oj < lo.samples_j) {
                const int32_t* o = gsum + (size_t)(oi * step) * sw + oj * step;
                float dx, dy, dxy;
                haar_response(lo, [&](int yy, int xx) { return o[(size_t)yy * sw + xx]; }, &dx, &dy, &dxy);
                d = dx * dy - 0.81f * dxy * dxy;
            }
            return d;
        };
        // the 27 entries are spread over the group's 16 lanes: lane k takes entries k and k + 16 (entry e = 9 * row + neighbour)
        float v2[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int e = k + 16 * q;
            v2[q] = r->n9[e < 27 ? e : 0];
        }
#pragma unroll 1
        for (int q = 0; q < 2; q++) {
            const int e = k + 16 * q;
            if (e >= 27) continue;
            const int row = e / 9, nb = e - row * 9;
            const bool outer = (L == 1 && row == 0) || (L == 3 && row == 2);
            if (outer || v2[q] == kDetBelow) v2[q] = exact_det(op.L[L - 1 + row], i + nb / 3 - 1, j + nb % 3 - 1);
        }
        const int g0 = (threadIdx.x & 63) & ~15;          // first lane of this group within the wave
        float N9[3][9];
#pragma unroll
        for (int e = 0; e < 27; e++) N9[e / 9][e % 9] = __shfl(v2[e / 16], g0 + (e & 15));
        if (k != 0) continue;
        const float val0 = N9[1][4];
        if (L != 2) {
            bool is_max = true;
#pragma unroll
            for (int b = 0; b < 9; b++) is_max = is_max && (val0 > (L == 1 ? N9[0][b] : N9[2][b]));
            if (!is_max) continue;
        }
        const LayerPat& lp = op.L[L];
        float dx, dy;
        {
            const int32_t* o = gsum + (size_t)((i - lp.margin) * step) * sw + (j - lp.margin) * step;
            haar_response(lp, [&](int yy, int xx) { return o[(size_t)yy * sw + xx]; }, &dx, &dy, nullptr);
        }
        const int ds = lp.size - op.L[L - 1].size;
        uvo_keypoint kp;
        bool ok;
        switch (step) {
        case 1:  ok = make_keypoint<1>(N9, val0, dx + dy, i, j, lp.size, ds, op.octave, w, h, &kp); break;
        case 2:  ok = make_keypoint<2>(N9, val0, dx + dy, i, j, lp.size, ds, op.octave, w, h, &kp); break;
        case 4:  ok = make_keypoint<4>(N9, val0, dx + dy, i, j, lp.size, ds, op.octave, w, h, &kp); break;
        default: ok = make_keypoint<8>(N9, val0, dx + dy, i, j, lp.size, ds, op.octave, w, h, &kp); break;
        }
        if (ok) s_kp[im][atomicAdd(&s_cnt[im], 1)] = kp;
    }
    __syncthreads();
    if (threadIdx.x < 2 && s_cnt[threadIdx.x] > 0) s_base[threadIdx.x] = atomicAdd(&out.count[threadIdx.x], s_cnt[threadIdx.x]);
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * kFinishPerWg; e += 256) {
        const int im = e / kFinishPerWg, q = e - im * kFinishPerWg;
        if (q < s_cnt[im] && s_base[im] + q < out.cap) out.cand[im][s_base[im] + q] = s_kp[im][q];
    }
}

// quads a thread has in flight during the tile fill: the whole tile in one round trip when that takes at most six per thread
#define UVO_FILL(quads, nt) (((quads) + (nt) - 1) / (nt) <= 6 ? ((quads) + (nt) - 1) / (nt) : 4)
// One tile of octave 0 or 1 (tile bx, by of image im): the body of k_hessian_nms_c and of the octave-0 blocks of k_hessian_nms_c0_p23
template <int O, int TW, int TH, int NT, class OP>
__device__ __forceinline__ void hessian_nms_c_tile(const ImgPair& ip, int w, int h, const OP& op, float thr, const SurvOut& sv,
                                                   int bx, int by, int im, unsigned char* smem)
{
    using OC = OctC<O>;
    constexpr int STEP = OC::STEP;
    constexpr int TWs = OctTile<O, TW>::TWs;
    constexpr int THs = (TH - 1) * STEP + (OC::HI - OC::LO) + 1;
    constexpr int PW = OctTile<O, TW>::PW;
    const int tid = lane_tid();
    const int sw = w + 1;
    const int32_t* __restrict__ gsum = ip.sum[im];
    float* sdet = reinterpret_cast<float*>(smem);                    // [3][TH][TW]: layers 1..3
    int32_t* stile = reinterpret_cast<int32_t*>(smem + sizeof(float) * 3 * TH * TW);   // [THs][STEP][PW]

    const int px0 = bx * (TW - 2) - 1, py0 = by * (TH - 2) - 1;
    const int sx0 = px0 * STEP + OC::LO, sy0 = py0 * STEP + OC::LO;
    // Tile fill: whole quads (global_load_dwordx4 at 4-byte alignment, ds_write_b128 / 2 x ds_write_b64), kFill of them in
    // flight per thread before the first LDS store.  Rows are clamped to the image; columns are not: every corner a valid
    // sample reads lies inside the image (samples_i/j are defined that way), out-of-image elements are only ever read by
    // masked samples, and the integral buffers carry kSumPad ints of slack on both sides, so the overshoot of the first
    // and last rows stays inside the allocation.
    constexpr int QPR = OctTile<O, TW>::TWq / 4, kQuads = THs * QPR, kFill = UVO_FILL(kQuads, NT), kQ = NT / QPR, kR = NT % QPR;
    static_assert(STEP == 1 || STEP == 2, "tile fill handles octaves 0 and 1");
    static_assert(OctTile<O, TW>::TWq < kSumPad && STEP - OC::LO < kSumPad, "integral slack too small for the tile overshoot");
    int fidx = tid, fty = tid / QPR, fq = tid - fty * QPR;           // quad, tile row, quad in row: advanced by NT per step
#pragma unroll 1
    for (int i0 = 0; i0 < kQuads; i0 += NT * kFill) {
        SumQuad v[kFill];
        int dst[kFill];
#pragma unroll
        for (int u = 0; u < kFill; u++) {
            const bool in = fidx < kQuads;
            const int gy = min(max(sy0 + fty, 0), h), gx = sx0 + 4 * fq;
            v[u] = *reinterpret_cast<const SumQuad*>(gsum + (in ? gy * sw + gx : 0));
            dst[u] = in ? fty * STEP * PW + (4 / STEP) * fq : -1;
            fidx += NT; fq += kR; fty += kQ;
            if (fq >= QPR) { fq -= QPR; fty++; }
        }
#pragma unroll
        for (int u = 0; u < kFill; u++) {
            if (dst[u] < 0) continue;
            if constexpr (STEP == 1) {
                *reinterpret_cast<int4*>(stile + dst[u]) = make_int4(v[u].a, v[u].b, v[u].c, v[u].d);
            } else {
                *reinterpret_cast<int2*>(stile + dst[u]) = make_int2(v[u].a, v[u].c);          // even columns
                *reinterpret_cast<int2*>(stile + dst[u] + PW) = make_int2(v[u].b, v[u].d);     // odd columns
            }
        }
    }
    nms_zero<TW, TH>(reinterpret_cast<unsigned*>(stile + THs * STEP * PW));
    __syncthreads();
    hess_stamp(1);
    det_layer_c<O, 1, TW, TH, NT>(stile, sdet, op, px0, py0, thr);
    det_layer_c<O, 2, TW, TH, NT>(stile, sdet, op, px0, py0, thr);
    det_layer_c<O, 3, TW, TH, NT>(stile, sdet, op, px0, py0, thr);
    __syncthreads();
    hess_stamp(2);
    nms_survivors<TW, TH, NT>(sdet, reinterpret_cast<unsigned*>(stile + THs * STEP * PW), op, thr, px0, py0, im, sv);
}
template <int O, int TW, int TH, int NT>
__global__ __launch_bounds__(NT) void k_hessian_nms_c(LanePair lp, int w, int h, OctavePat op, float thr)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const LaneArgs& LA = UVO_LANE_OF(lp, blockIdx.z);
    hessian_nms_c_tile<O, TW, TH, NT>(LA.ip, w, h, op, thr, LA.sv, blockIdx.x, blockIdx.y, UVO_LANE_IM(blockIdx.z), smem);
}

// Octaves 2 and 3: det layers from the de-interleaved planes, det planes in LDS, survivors as above.
template <int O, int TW, int TH, int NT>
__global__ __launch_bounds__(NT) void k_hessian_nms_p(LanePair lp, int w, int h, OctavePat op, float thr)
{
    __shared__ float sdet[3 * TH * TW];
    __shared__ unsigned s_list[NmsLds<TW, TH>::kWords];
    const LaneArgs& LA = UVO_LANE_OF(lp, blockIdx.z); const ImgPair& ip = LA.ip; const SurvOut& sv = LA.sv;
    const int im = UVO_LANE_IM(blockIdx.z);
    const int px0 = blockIdx.x * (TW - 2) - 1, py0 = blockIdx.y * (TH - 2) - 1;
    det_layer_p<O, 1, TW, TH, NT>(ip.planes[im], ip.pw, ip.pstride, sdet, op, px0, py0, thr);
    det_layer_p<O, 2, TW, TH, NT>(ip.planes[im], ip.pw, ip.pstride, sdet, op, px0, py0, thr);
    det_layer_p<O, 3, TW, TH, NT>(ip.planes[im], ip.pw, ip.pstride, sdet, op, px0, py0, thr);
    nms_zero<TW, TH>(s_list);
    __syncthreads();
    nms_survivors<TW, TH, NT>(sdet, s_list, op, thr, px0, py0, im, sv);
}

// Octaves 2 and 3 in one launch (block b walks octave 2's tiles, then octave 3's): both are latency-bound on plane reads and
// neither fills the chip (5120 and 2560 waves); side by side they take about what octave 2 took alone.
static const int kP23Threads = 512;
struct P23Grid { int nbx2, nb2, nbx3, nb3; };
template <class OP>
__device__ __forceinline__ void hessian_nms_p23_tile(const ImgPair& ip, const OP& op2, const OP& op3, float thr, const SurvOut& sv,
                                                     bool oct3, int bx, int by, int im, float* sdet, unsigned* s_list)
{
    constexpr int TW2 = 32, TH2 = 16, TW3 = 16, TH3 = 16;
    static_assert(TW3 * TH3 <= TW2 * TH2 && NmsLds<TW3, TH3>::kWords <= NmsLds<TW2, TH2>::kWords, "octave 3 reuses octave 2's LDS");
    if (!oct3) {
        const int px0 = bx * (TW2 - 2) - 1, py0 = by * (TH2 - 2) - 1;
        det_layers_p3<2, TW2, TH2, kP23Threads>(ip.planes[im], ip.pw, ip.pstride, sdet, op2, px0, py0, thr);
        nms_zero<TW2, TH2>(s_list);
        __syncthreads();
        hess_stamp(2);
        nms_survivors<TW2, TH2, kP23Threads>(sdet, s_list, op2, thr, px0, py0, im, sv);
    } else {
        const int px0 = bx * (TW3 - 2) - 1, py0 = by * (TH3 - 2) - 1;
        det_layers_p3<3, TW3, TH3, kP23Threads>(ip.planes[im], ip.pw, ip.pstride, sdet, op3, px0, py0, thr);
        nms_zero<TW3, TH3>(s_list);
        __syncthreads();
        hess_stamp(2);
        nms_survivors<TW3, TH3, kP23Threads>(sdet, s_list, op3, thr, px0, py0, im, sv);
    }
}
static const int kP23SdetFloats = 3 * 16 * 32;
// All four octaves in ONE launch.  Octave 0 is VALU-bound (79 % busy on its own), octave 1 waits on its L2 -> LDS tile fill
// (VALU 46 %), octaves 2 + 3 on plane reads (7-21 %): their tiles are dealt out evenly over the launch (`order[b]`: kind and tile
// index of block b, built on the host so that every kind is spread uniformly), so that every CU holds all kinds and the memory waits
// of one hide behind the box sums of another.  Every block reserves the largest tile's LDS (octave 1 with 20 sample rows: 52.6 KB,
// three blocks per CU -- the octave-0 / plane mix runs as fast with three as with four, 63 us, but not with two, 81 us).  The
// patterns come from the device copy of the table (four of them exceed the 4 KB of kernel arguments).
static const int kO1TileRows = 18;             // sample rows of an octave-1 tile in the merged launch
struct HessGrid { int nbx0, nb0, nbx1, nb1; P23Grid g; };          // tiles per kind
// (Measured and not kept, round 4: two to four consecutive tiles of a kind per workgroup, to save the ~1.9 us a workgroup slot's
// turn-over costs per tile -- 93 / 103 / 105 us against 84: a slot bound to a fixed run of tiles loses the balance that "whichever
// slot frees up takes the next tile" gives; the same lesson as the resident-workgroup queues)
template <int TW, int TH>
__global__ __launch_bounds__(kP23Threads) void k_hessian_nms_all(LanePair lp, int w, int h, HessDims hd, float thr,
                                                                 const uint32_t* __restrict__ order)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const LaneArgs& LA = UVO_LANE_OF(lp, blockIdx.y); const ImgPair& ip = LA.ip;
    // the survivor list's address, counter and capacity are read here, with the other arguments, rather than by scalar loads of their own
    // behind the tile's last barriers
    SurvOut sv = LA.sv;
    asm volatile("" : "+s"(sv.list), "+s"(sv.count), "+s"(sv.cap));
    const int im = UVO_LANE_IM(blockIdx.y);
    const unsigned e = order[blockIdx.x];
    const int kind = e >> 30, bx = e & 0x3FFF, by = (e >> 14) & 0x3FFF;
    hess_stamp(0); hess_stamp(4, kind);
    if (kind == 0) hessian_nms_c_tile<0, TW, TH, kP23Threads>(ip, w, h, hd.o[0], thr, sv, bx, by, im, smem);
    else if (kind == 1) hessian_nms_c_tile<1, 32, kO1TileRows, kP23Threads>(ip, w, h, hd.o[1], thr, sv, bx, by, im, smem);
    else {
        float* sdet = reinterpret_cast<float*>(smem);
        hessian_nms_p23_tile(ip, hd.o[2], hd.o[3], thr, sv, kind == 3, bx, by, im, sdet, reinterpret_cast<unsigned*>(sdet + kP23SdetFloats));
    }
    hess_stamp(3);
}

// debug / parity hook: one det+trace layer written to global planes (rows x cols)
// Measurement only (tools/probe/gpu.sh sens): a kernel of known duration in the middle of stage A -- one wave ("thin") or three
// workgroups per CU holding a detection tile's LDS ("fat") -- to read off how the pipeline's cadence follows either kind of time.
__global__ __launch_bounds__(256) void k_probe_hold(int ticks)
{
    extern __shared__ int probe_lds[];
    if (threadIdx.x == 0) probe_lds[0] = ticks;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

__global__ void k_hessian_layer_debug(const int32_t* gsum, int w, int h, LayerPat lp, int step, int rows, int cols,
                                      float* det, float* trace)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= lp.samples_j || i >= lp.samples_i) return;
    const int sw = w + 1;
    const int32_t* o = gsum + (size_t)(i * step) * sw + j * step;
    float dx, dy, dxy;
    haar_response(lp, [&](int yy, int xx) { return o[(size_t)yy * sw + xx]; }, &dx, &dy, &dxy);
    size_t pos = (size_t)(i + lp.margin) * cols + (j + lp.margin);
    det[pos] = dx * dy - 0.81f * dxy * dxy;
    trace[pos] = dx + dy;
}

// ------------------------------------------------------------------------------------------
// deterministic ordering: rank = number of keypoints that sort before this one under OpenCV's
// KeypointGreater (response desc, size desc, octave desc, y desc, x asc).  The five fields are
// packed into order-preserving integer keys so a comparison is three 64-bit compares; ranks are
// accumulated over (i-block, j-chunk) tiles with integer atomics (order-independent), then the
// records are scattered to their rank.  Atomic-append order never reaches the output.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ord_f32(float f)
{
    uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
struct SortKey { unsigned long long k1, k2, k3; };
__device__ __forceinline__ SortKey make_sort_key(const uvo_keypoint& kp)
{
    SortKey k;
    k.k1 = ((unsigned long long)ord_f32(kp.response) << 32) | ord_f32(kp.size);
    k.k2 = ((unsigned long long)(uint32_t)kp.octave << 32) | ord_f32(kp.y);
    k.k3 = ((unsigned long long)(~ord_f32(kp.x)) << 32) | (uint32_t)(1 - kp.class_id);
    return k;
}

static const int kSmallWin = 128;       // descriptor windows up to this size use the small-LDS kernel
static const int kMaxWin = 740;         // (int)(21 * 264 * 1.2f / 9) = 739: the window of the largest octave-3 keypoint
// Large windows are listed in append order by k_rank_scatter and then sorted by descending window size (a (keypoint,
// column) task of descriptor64_big costs about ceil(win/256) lane passes x ceil((win/21 + 2)/16) tap batches, 1..9 units),
// which the task dealing of that kernel relies on.  Counting sort, one workgroup per image.
static const int kBigBins = 1024;
static const int kTripleWin = 246;     // windows up to this size: three destination columns per task (3 x 246 floats share the 740-float row buffer)
__global__ __launch_bounds__(1024) void k_big_sort(LanePair lp, int cap, const int* __restrict__ iscale_tab)
{
    const LaneArgs& LA = UVO_LANE_OF(lp, blockIdx.x);
    const int4* __restrict__ in = LA.big_in; int4* __restrict__ out = LA.big_out; const int* __restrict__ big_n = LA.da.big_n; int* __restrict__ big_large = LA.big_large;
    const int im = UVO_LANE_IM(blockIdx.x), tid = threadIdx.x;
    const int n = min(big_n[im], cap);
    __shared__ int hist[kBigBins], scan[kBigBins];
    hist[tid] = 0;
    __syncthreads();
    for (int e = tid; e < n; e += 1024) atomicAdd(&hist[min(kBigBins - 1, max(0, kBigBins - 1 - in[im * cap + e].y))], 1);
    __syncthreads();
    // exclusive scan of the 1024 bins (Hillis-Steele on the inclusive sums)
    int v = hist[tid];
    scan[tid] = v;
    __syncthreads();
    for (int off = 1; off < kBigBins; off <<= 1) {
        const int add = tid >= off ? scan[tid - off] : 0;
        __syncthreads();
        scan[tid] += add;
        __syncthreads();
    }
    hist[tid] = scan[tid] - v;                     // first output position of the bin
    if (tid == kBigBins - 1 - kTripleWin) big_large[im] = scan[tid] - v;     // entries before this bin are wider than kTripleWin
    __syncthreads();
    for (int e = tid; e < n; e += 1024) {
        int4 par = in[im * cap + e];
        const int pos = atomicAdd(&hist[min(kBigBins - 1, max(0, kBigBins - 1 - par.y))], 1);
        // resizeAreaFast_ applies when the scale is an integer to within DBL_EPSILON (tabulated per window size):
        // .y = win_size | (iscale << 16), iscale = 0 for the general path
        par.y |= iscale_tab[min(par.y, kMaxWin)] << 16;
        out[im * cap + pos] = par;
    }
}
static const int kSortChunk = 128;     // compared-against keypoints per workgroup: small, so that ~600 workgroups share the work

__global__ __launch_bounds__(256) void k_rank_partial(LanePair lp)
{
    const SortArgs& a = UVO_LANE_OF(lp, blockIdx.y).sa;
    const int im = UVO_LANE_IM(blockIdx.y), tid = threadIdx.x;
    const int n = min(a.cand_n[im], a.cap);
    __shared__ SortKey tile[kSortChunk];
    __shared__ unsigned long long tile_k1[kSortChunk];
    // the count lives on the device: a fixed grid walks the (256 keypoints) x (kSortChunk compared-against) tiles that exist
    // (a cap x cap grid is 4096 workgroups of which ~550 find work, and dispatching the empty ones costs more than the work)
    const int nib = (n + 255) / 256, ntiles = nib * ((n + kSortChunk - 1) / kSortChunk);
    for (int t_ = blockIdx.x; t_ < ntiles; t_ += gridDim.x) {
        const int i0 = (t_ % nib) * 256, j0 = (t_ / nib) * kSortChunk;
        const int cnt = min(kSortChunk, n - j0);
        const int me = i0 + tid;
        uvo_keypoint mykp = {};
        if (me < n) mykp = a.cand[im][me];                   // (in flight together with the tile's records: one memory round trip, not two)
        for (int t = tid; t < cnt; t += 256) { tile[t] = make_sort_key(a.cand[im][j0 + t]); tile_k1[t] = tile[t].k1; }
        __syncthreads();
        if (me < n) {
            const SortKey mine = make_sort_key(mykp);
            int rank = 0;
            // the first key (response, size) decides almost every comparison: one 64-bit compare per pair on a dense array of
            // first keys, the full lexicographic comparison only on the rare equal ones
#pragma unroll 16
            for (int k = 0; k < cnt; k++) {
                const unsigned long long o1 = tile_k1[k];
                rank += o1 > mine.k1 ? 1 : 0;
                if (o1 == mine.k1) {
                    const SortKey o = tile[k];
                    rank += (o.k2 > mine.k2 || (o.k2 == mine.k2 && (o.k3 > mine.k3 || (o.k3 == mine.k3 && j0 + k < me)))) ? 1 : 0;
                }
            }
            atomicAdd(&a.rank[im * a.cap + me], rank);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_rank_scatter(LanePair lp)
{
    const SortArgs& a = UVO_LANE_OF(lp, blockIdx.y).sa;
    const int im = UVO_LANE_IM(blockIdx.y);
    const int n = min(a.cand_n[im], a.cap);
    const int me = blockIdx.x * 256 + threadIdx.x;
    bool big = false;
    int4 par = make_int4(0, 0, 0, 0);
    if (me < n) {
        int r = a.rank[im * a.cap + me];
        a.rank[im * a.cap + me] = 0;                 // ready for the next frame
        const uvo_keypoint kp = a.cand[im][me];
        a.out[im][r] = kp;
        // keypoints whose descriptor window exceeds kSmallWin go to the large-window descriptor launch, with the window
        // geometry every task of theirs needs (SURFInvoker: win_size, win_offset, start_x/start_y)
        const float sc = kp.size * 1.2f / 9.0f;
        const int win_size = (int)((20 + 1) * sc);
        if (win_size > kSmallWin) {
            const float win_offset = -(float)(win_size - 1) / 2;
            big = true;
            par = make_int4(r, win_size, cv_round_f(kp.x + win_offset), cv_round_f(kp.y - win_offset));
        }
    }
    {   // one atomic per wave
        const int lane = threadIdx.x & 63;
        const unsigned long long m = __ballot(big);
        if (m != 0) {
            int base = 0;
            const int leader = __ffsll((long long)m) - 1;
            if (lane == leader) base = atomicAdd(&a.big_n[im], __popcll(m));
            base = __shfl(base, leader);
            if (big) a.big_par[im * a.cap + base + __popcll(m & ((1ull << lane) - 1))] = par;
        }
    }
    if (me == 0) *a.out_n[im] = n;
    if (me == 0 && im == 0 && a.gate_nqa) {          // VO:556: both images need >= MIN_NUM_FEATURES keypoints, else no stereo matching
        const int nL = n, nR = min(a.cand_n[1], a.cap);
        *a.gate_nqa = (nL >= a.gate_min_features && nR >= a.gate_min_features) ? nL : 0;
    }
}

// ------------------------------------------------------------------------------------------
// descriptor: one workgroup per keypoint
// ------------------------------------------------------------------------------------------
struct AreaTab { int sx1, sx2; float a_first, a_mid, a_last; bool has_first, has_last; };

// computeResizeAreaTab (resize.cpp) for one destination index.  The tables depend on the window size only, so they are built
// once per context on the host for every possible size (surf_build_area_tables; plain IEEE double arithmetic, the same
// values a device evaluation gives) and the kernels look them up.
__host__ __device__ __forceinline__ AreaTab area_tab(int dx, int ssize, double scale)
{
    AreaTab t;
    double fsx1 = dx * scale;
    double fsx2 = fsx1 + scale;
    double cellWidth = scale < ssize - fsx1 ? scale : ssize - fsx1;
    int sx1 = cv_ceil_d(fsx1), sx2 = cv_floor_d(fsx2);
    sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
    sx1 = sx1 < sx2 ? sx1 : sx2;
    t.sx1 = sx1; t.sx2 = sx2;
    t.has_first = sx1 - fsx1 > 1e-3;
    t.a_first = (float)((sx1 - fsx1) / cellWidth);
    t.a_mid = (float)(1.0 / cellWidth);
    t.has_last = fsx2 - sx2 > 1e-3;
    double a = fsx2 - sx2; if (a > 1.) a = 1.; if (a > cellWidth) a = cellWidth;
    t.a_last = (float)(a / cellWidth);
    return t;
}

__device__ __forceinline__ uint8_t sat_u8(float v) { int iv = cv_round_f(v); return (uint8_t)(iv < 0 ? 0 : iv > 255 ? 255 : iv); }


// PATCH (21 x 21, shared) -> gradients, 4x4 cells of 4 (extended: 8) sums, normalisation -> row k of a.desc[im]; 256 threads,
// PATCH already synchronised
__device__ __forceinline__ void describe_tail(const DescArgs& a, int im, int k, const int (*PATCH)[21])
{
    const int tid = threadIdx.x;
    __shared__ float DX[20][20], DY[20][20];
    __shared__ float vec[128];
    __shared__ float s_scale;
    const int dsize = a.extended ? 128 : 64;
    for (int o = tid; o < 400; o += 256) {
        int i = o / 20, j = o - i * 20;
        float dw = a.DW[o];
        float vx = (PATCH[i][j+1] - PATCH[i][j] + PATCH[i+1][j+1] - PATCH[i+1][j]) * dw;
        float vy = (PATCH[i+1][j] - PATCH[i][j] + PATCH[i+1][j+1] - PATCH[i][j+1]) * dw;
        DX[i][j] = vx; DY[i][j] = vy;
    }
    __syncthreads();
    if (tid < dsize) {
        float acc = 0.f;
        if (!a.extended) {
            int cell = tid >> 2, comp = tid & 3, ci = cell >> 2, cj = cell & 3;
            for (int y = ci * 5; y < ci * 5 + 5; y++)
                for (int x = cj * 5; x < cj * 5 + 5; x++) {
                    float t = (comp & 1) ? DY[y][x] : DX[y][x];
                    acc += (comp & 2) ? (float)fabs(t) : t;
                }
        } else {
            // surf.cpp, extended: (sum tx, sum |tx|) over ty >= 0 and over ty < 0, then (sum ty, sum |ty|) over tx >= 0 and over tx < 0
            int cell = tid >> 3, comp = tid & 7, ci = cell >> 2, cj = cell & 3;
            for (int y = ci * 5; y < ci * 5 + 5; y++)
                for (int x = cj * 5; x < cj * 5 + 5; x++) {
                    const float tx = DX[y][x], ty = DY[y][x];
                    const float val = comp < 4 ? tx : ty, sel = comp < 4 ? ty : tx;
                    const bool take = (comp & 2) ? sel < 0 : sel >= 0;
                    if (take) acc += (comp & 1) ? (float)fabs(val) : val;
                }
        }
        vec[tid] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        double square_mag = 0;
        for (int kk = 0; kk < dsize; kk++) square_mag += vec[kk] * vec[kk];
        s_scale = (float)(1. / (sqrt(square_mag) + FLT_EPSILON));
    }
    __syncthreads();
    if (tid < dsize) a.desc[im][(size_t)k * dsize + tid] = vec[tid] * s_scale;
}

// ------------------------------------------------------------------------------------------
// The horizontal pass of resizeArea_ for ONE destination column of ONE window, by one wave (round 3).
//   buf[i] = sum over the column's taps c of WIN[i][c] * alpha_c,  WIN[i][c] = img(clamp(start_y - c), clamp(start_x + i)),
// accumulated in OpenCV's tap order.  Everything about the task is the same for every lane, so it lives in scalar registers:
// the image is read through a buffer resource whose row offset is the SGPR `soffset` of the load (no per-tap address VALU, no
// 64-bit adds: sub, max, min, mul on the scalar unit), the first and the last tap are peeled (their weights differ), the middle
// taps all carry a_mid and run in batches of eight loads in flight, and the tail of the last batch is branched over rather than
// loaded and masked.  Round 2's loop spent 16.7 VALU and 22 SALU instructions per useful tap (address add, two scalar selects
// per tap for the weight, sixteen loads per batch whatever the tap count): the scalar unit -- one per CU -- was 73 % busy.
// A lane owns PX adjacent pixels of each of Q chunks of 64 * PX columns: PX = 4 (aligned dword loads; Q = 1..3 covers the 739-pixel
// windows), PX = 2 and PX = 1 for windows up to 127 / 64 pixels so that small windows do not idle three quarters of the lanes.
// ------------------------------------------------------------------------------------------
static const int kTapBatch = 16;           // image rows a lane of descriptor64_big has in flight
// UVO_DESC_STAMPS=<file> (measurement): wall-clock stamps (100 MHz) of the phases of every small-window workgroup of the last descriptor launch
__device__ long long* g_desc_stamps = nullptr;
__device__ __forceinline__ void desc_stamp(int im, int k, int slot, long long v = -1)
{
#if UVO_STAMPS
    long long* st = g_desc_stamps;
    if (st && threadIdx.x == 0) st[((size_t)im * 16384 + (k & 16383)) * 8 + slot] = v >= 0 ? v : (long long)wall_clock64();
#endif
}
// (Round 4, measured and removed: staging a small window's bytes in LDS first -- every load of the workgroup in flight at once -- and
// running the horizontal pass from there.  Stamps: table + barrier 1.2 us and horizontal pass 5.3 us became staging 4.0 us + LDS pass
// 3.9 us; the small-window half of the launch alone 30.0 -> 34.3 us.)
static const int kPatchStride = 448;       // bytes of patch scratch per keypoint (441 used)
__device__ __forceinline__ int sgpr_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
// a < b ? x : y on scalar registers (the compiler turns the C expression into VALU selects on copies of the operands)
__device__ __forceinline__ int ssel_lt(int a, int b, int x, int y)
{
    int r;
    asm volatile("s_cmp_lt_i32 %1, %2\n\ts_cselect_b32 %0, %3, %4" : "=s"(r) : "s"(a), "s"(b), "s"(x), "s"(y) : "scc");
    return r;
}
__device__ __forceinline__ float sgpr_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

struct ColTask { int y0, n, h1, w; float a_first, a_mid, a_last; };      // tap t reads row clamp(y0 - t, 0, h1); n >= 1 taps

template <int PX> __device__ __forceinline__ unsigned px_load(ImgRsrc rs, int voff, int soff);
template <> __device__ __forceinline__ unsigned px_load<1>(ImgRsrc rs, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b8(rs, voff, soff, 0); }
template <> __device__ __forceinline__ unsigned px_load<2>(ImgRsrc rs, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, 0); }
template <> __device__ __forceinline__ unsigned px_load<4>(ImgRsrc rs, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0); }

template <int PX, int Q>
struct AreaCol {
    ImgRsrc rs; int w, h1;
    int x[Q], voff[Q];
    float acc[Q][PX];
    // The window's in-image columns are [xlo, xhi) = [max(start_x, 0), min(start_x + win, w)); lane l owns the pixels
    // xa + PX * (l + 64 q) .., xa = xlo rounded down to PX
    __device__ __forceinline__ AreaCol(ImgRsrc rs_, int w_, int h_, int lane, int xa) : rs(rs_), w(w_), h1(h_ - 1)
    {
#pragma unroll
        for (int q = 0; q < Q; q++) { x[q] = xa + PX * (lane + 64 * q); voff[q] = x[q] < w - PX ? x[q] : w - PX; }      // loads stay inside the row whatever the lane
    }
    __device__ __forceinline__ void load(const ColTask& t, int tap, unsigned (&v)[Q]) const
    {
        int y = t.y0 - tap; y = y > 0 ? y : 0; y = y < h1 ? y : h1;
        const int soff = y * w;
#pragma unroll
        for (int q = 0; q < Q; q++) v[q] = px_load<PX>(rs, voff[q], soff);
    }
    // (float)pixel * alpha.  The conversion-free form the detector uses -- fma(bits(pixel | 0x4B000000), alpha, -(2^23 alpha)), v_or_b32_sdwa +
    // v_pk_fma_f32 -- was measured here and removed: it holds every loaded word's four floats at once, 37 registers spilled under this
    // kernel's 64, 90.5 us against 57.2.
    static __device__ __forceinline__ float px_times(unsigned v, int p, float alpha) { return (float)(int)((v >> (8 * p)) & 255u) * alpha; }
    __device__ __forceinline__ void first(const unsigned (&v)[Q], float alpha)      // 0.f + v * a == v * a
    {
#pragma unroll
        for (int q = 0; q < Q; q++)
#pragma unroll
            for (int p = 0; p < PX; p++) acc[q][p] = px_times(v[q], p, alpha);
    }
    __device__ __forceinline__ void accum(const unsigned (&v)[Q], float alpha)
    {
#pragma unroll
        for (int q = 0; q < Q; q++)
#pragma unroll
            for (int p = 0; p < PX; p++) acc[q][p] += px_times(v[q], p, alpha);
    }
    // exactly N taps, everything static: N loads go out, then N sums (Q = 1: the columns of windows up to ~250 pixels)
    template <int N> __device__ __forceinline__ void taps_static(const ColTask& t)
    {
        unsigned v[N][Q];
#pragma unroll
        for (int i = 0; i < N; i++) load(t, i, v[i]);
        first(v[0], t.a_first);
#pragma unroll
        for (int i = 1; i < N - 1; i++) accum(v[i], t.a_mid);
        if (N > 1) accum(v[N - 1], t.a_last);
    }
    // any tap count: first and last tap peeled, the middle taps in batches of B loads per chunk; a batch is always loaded whole
    // (a tap index past the column reads the last tap's row again -- it is in flight already) and a tap past the column
    // gets weight +0, which leaves the non-negative sums unchanged: no branch, no conditionally defined register
    __device__ __forceinline__ void taps_any(const ColTask& t)
    {
        constexpr int B = Q == 1 ? 8 : (Q == 2 ? 7 : 4);
        const int last = t.n - 1;
        unsigned v0[Q], vl[Q], v[B][Q];
        load(t, 0, v0);
        load(t, last > 0 ? last : 0, vl);
        first(v0, t.n > 0 ? t.a_first : 0.f);
        for (int tap = 1; tap < last; tap += B) {
#pragma unroll
            for (int i = 0; i < B; i++) load(t, tap + i < last ? tap + i : last, v[i]);
#pragma unroll
            for (int i = 0; i < B; i++) accum(v[i], tap + i < last ? t.a_mid : 0.f);
        }
        accum(vl, last > 0 ? t.a_last : 0.f);
    }
    // one column, the variant by its tap count
    __device__ __forceinline__ void taps(const ColTask& t)
    {
        if (Q == 1) {
            switch (t.n) {
            case 1: taps_static<1>(t); break;   case 2: taps_static<2>(t); break;   case 3: taps_static<3>(t); break;
            case 4: taps_static<4>(t); break;   case 5: taps_static<5>(t); break;   case 6: taps_static<6>(t); break;
            case 7: taps_static<7>(t); break;   case 8: taps_static<8>(t); break;   case 9: taps_static<9>(t); break;
            case 10: taps_static<10>(t); break; case 11: taps_static<11>(t); break; case 12: taps_static<12>(t); break;
            case 13: taps_static<13>(t); break; case 14: taps_static<14>(t); break;
            default: taps_any(t); break;
            }
        } else if (Q == 2 && PX == 1) {                 // the small-window part: up to 128 columns a pixel per lane, up to 8 taps
            switch (t.n) {
            case 1: taps_static<1>(t); break;   case 2: taps_static<2>(t); break;   case 3: taps_static<3>(t); break;
            case 4: taps_static<4>(t); break;   case 5: taps_static<5>(t); break;   case 6: taps_static<6>(t); break;
            case 7: taps_static<7>(t); break;   case 8: taps_static<8>(t); break;
            default: taps_any(t); break;
            }
        } else taps_any(t);
    }
    // The sums land at row[(x - start_x) + sh], sh = start_x & 3, which makes every lane's PX floats one aligned LDS store; window
    // rows left / right of the image replicate the border column's sum (WIN clamps x).  The caller reads row[i + sh] for i in
    // [0, win).  `row` holds at least win + 8 floats.
    __device__ __forceinline__ void store(int lane, int start_x, int win_size, float* __restrict__ row) const
    {
        const int xlo = start_x > 0 ? start_x : 0, xhi = start_x + win_size < w ? start_x + win_size : w;
        const int sh = start_x & 3;
        float* dst = row + sh - start_x;
#pragma unroll
        for (int q = 0; q < Q; q++) {
            if (x[q] < xhi) {
                if (PX == 4) *reinterpret_cast<float4*>(dst + x[q]) = make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
                else if (PX == 2) *reinterpret_cast<float2*>(dst + x[q]) = make_float2(acc[q][0], acc[q][1]);
                else dst[x[q]] = acc[q][0];
            }
        }
        const int ilo = xlo - start_x, ihi = xhi - start_x;
        if (ilo > 0) { const float b = row[sh + ilo]; for (int i = lane; i < ilo; i += 64) row[sh + i] = b; }
        if (ihi < win_size) { const float b = row[sh + ihi - 1]; for (int i = ihi + lane; i < win_size; i += 64) row[sh + i] = b; }
    }
};

template <int PX, int Q>
__device__ __forceinline__ void area_column(ImgRsrc rs, int lane, int start_x, int win_size, const ColTask& t, float* __restrict__ row)
{
    const int xlo = start_x > 0 ? start_x : 0;
    AreaCol<PX, Q> c(rs, t.w, t.h1 + 1, lane, xlo & ~(PX - 1));
    c.taps(t);
    c.store(lane, start_x, win_size, row);
}
// the column's entry of the window's resize table, held one entry per lane (entry lane % 21), as scalars; and its tap range
__device__ __forceinline__ ColTask col_task(const AreaTab& ty, int dx, int start_y, int w, int h)
{
    const int sx1 = __builtin_amdgcn_readlane(ty.sx1, dx), sx2 = __builtin_amdgcn_readlane(ty.sx2, dx);
    const float a_first = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ty.a_first), dx));
    const float a_mid = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ty.a_mid), dx));
    const float a_last = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ty.a_last), dx));
    const bool has_first = __builtin_amdgcn_readlane((int)ty.has_first, dx) != 0, has_last = __builtin_amdgcn_readlane((int)ty.has_last, dx) != 0;
    const int c_begin = has_first ? sx1 - 1 : sx1, c_end = has_last ? sx2 + 1 : sx2;
    // resizeArea_'s table entry of tap cc: cc < sx1 ? a_first : (cc < sx2 ? a_mid : a_last), for the first and the last tap; selected
    // on the bit patterns in scalar registers (written as a lambda over the floats the compiler selected between ADDRESSES of
    // stack copies and loaded the winner back from scratch: two dependent memory round trips per column)
    const int bf = sgpr_i(__float_as_int(a_first)), bm = sgpr_i(__float_as_int(a_mid)), bl = sgpr_i(__float_as_int(a_last));
    ColTask t;
    t.y0 = start_y - c_begin; t.n = c_end - c_begin; t.h1 = h - 1; t.w = w;
    t.a_first = __int_as_float(ssel_lt(c_begin, sx1, bf, ssel_lt(c_begin, sx2, bm, bl)));
    t.a_mid = a_mid;
    t.a_last = __int_as_float(ssel_lt(c_end - 1, sx1, bf, ssel_lt(c_end - 1, sx2, bm, bl)));
    return t;
}

// One workgroup per keypoint.  PATCH = cv::resize(WIN, 21x21, INTER_AREA) with
// WIN[i][j] = img(clamp(start_y - j), clamp(start_x + i)) is evaluated separably, exactly as
// resizeArea_ does: buf[i][dx] = sum_j WIN[i][j]*alpha_j (lanes run along i = image x, coalesced),
// then PATCH[dy][dx] = sum_i beta_i*buf[i][dx].  Windows up to kSmallWin = 128 samples (10.5 KB of LDS); larger
// ones are skipped here and handled by descriptor64_big.
__device__ __forceinline__ void describe_keypoint(const DescArgs& a, int w, int h, int k, int im)
{
    const int tid = threadIdx.x;
    desc_stamp(im, k, 0);
    const uvo_keypoint kp = a.kps[im][k];
    const float size = kp.size;
    const float s = size * 1.2f / 9.0f;
    const int win_size = (int)((20 + 1) * s);
    if (win_size > kSmallWin) return;                  // large windows: descriptor64_big
    desc_stamp(im, k, 1); desc_stamp(im, k, 6, win_size);
    extern __shared__ __align__(16) unsigned char smem_desc[];
    float* buf = reinterpret_cast<float*>(smem_desc);                 // [21][bp]
    const int bp = (win_size + 3) | 1;                                // odd pitch: the vertical pass walks 21 columns bank-conflict-free (+3: area_column's alignment shift)
    __shared__ AreaTab tab[21];
    __shared__ int PATCH[21][21];
    const uint8_t* __restrict__ img = a.img[im];
    const float win_offset = -(float)(win_size - 1) / 2;
    const int start_x = cv_round_f(kp.x + win_offset);
    const int start_y = cv_round_f(kp.y - win_offset);
    // (computed here rather than looked up in a.tabs / a.iscale: a second dependent fetch costs this latency-bound kernel more)
    const double inv_scale = (double)21 / win_size;
    const double scale = 1. / inv_scale;
    const int iscale = cv_round_d(scale);
    const bool area_fast = fabs(scale - iscale) < DBL_EPSILON;

    if (area_fast) {
        // resizeAreaFast_: integer block sums; 2x2 -> (sum+2)>>2, else saturate(sum * (1.f/area))
        for (int o = tid; o < 441; o += 256) {
            int dy = o / 21, dx = o - dy * 21;
            int sum = 0;
            for (int sx = 0; sx < iscale; sx++) {
                int y = start_y - (dx * iscale + sx); y = y > 0 ? y : 0; y = y < h - 1 ? y : h - 1;
                for (int sy = 0; sy < iscale; sy++) {
                    int x = start_x + dy * iscale + sy; x = x > 0 ? x : 0; x = x < w - 1 ? x : w - 1;
                    sum += img[(size_t)y * w + x];
                }
            }
            int result;
            if (iscale == 2) result = (sum + 2) >> 2;
            else { float sc = 1.f / (iscale * iscale); result = sat_u8(sum * sc); }
            PATCH[dy][dx] = result;
        }
    } else {
        if (tid < 21) tab[tid] = area_tab(tid, win_size, scale);
        __syncthreads();
        desc_stamp(im, k, 2);
        // horizontal pass of resizeArea_ (over WIN columns j = image rows), lanes along WIN rows i = image x: wave wv takes the
        // destination columns dx = wv, wv + 4, .. through the shared column core (round 3): a column's taps are the same for
        // every lane, so row clamps, offsets and weights are scalar work and a lane's pixel costs cvt + mul + add.  (Round 2 ran
        // a column per half-wave with per-lane tap parameters: ~33 VALU instructions per load.)
        {
            const int wv = sgpr_i(tid >> 6), lane = tid & 63;
            const AreaTab ty = tab[lane % 21];
            const int xlo = start_x > 0 ? start_x : 0, xhi = start_x + win_size < w ? start_x + win_size : w;
            const ImgRsrc rs = img_rsrc(img, w * h);
            const int sx = sgpr_i(start_x), sy = sgpr_i(start_y), ws = sgpr_i(win_size);
            if (xhi <= xlo) { for (int it = tid; it < 21 * bp; it += 256) buf[it] = 0.f; }        // cannot happen for a keypoint inside the image
            else if (xhi - xlo <= 64) {
                AreaCol<1, 1> c(rs, w, h, lane, xlo);
                for (int dx = wv; dx < 21; dx += 4) { const ColTask ct = col_task(ty, dx, sy, w, h); c.taps(ct); c.store(lane, sx, ws, buf + dx * bp); }
            } else {
                AreaCol<1, 2> c(rs, w, h, lane, xlo);
                for (int dx = wv; dx < 21; dx += 4) { const ColTask ct = col_task(ty, dx, sy, w, h); c.taps(ct); c.store(lane, sx, ws, buf + dx * bp); }
            }
        }
        __syncthreads();
        desc_stamp(im, k, 3);
        // vertical pass (over WIN rows i)
        for (int o = tid; o < 441; o += 256) {
            int dy = o / 21, dx = o - dy * 21;
            const AreaTab ty = tab[dy];
            const float* col = buf + dx * bp + (start_x & 3);
            float sum = 0.f;
            if (ty.has_first) sum += ty.a_first * col[ty.sx1 - 1];                    // the loop of resizeArea_'s table, split by weight
            for (int r = ty.sx1; r < ty.sx2; r++) sum += ty.a_mid * col[r];
            if (ty.has_last) sum += ty.a_last * col[ty.sx2];
            PATCH[dy][dx] = sat_u8(sum);
        }
    }
    __syncthreads();
    desc_stamp(im, k, 4); if (area_fast) desc_stamp(im, k, 7, 2);
    describe_tail(a, im, k, PATCH);
    desc_stamp(im, k, 5);
}

// ------------------------------------------------------------------------------------------
// SURF_UPRIGHT = false (the shipped configurations are upright): orientation assignment and the rotated
// sampling window of SURFInvoker, surf.cpp.
// ------------------------------------------------------------------------------------------
static const int kOriRadius = 6, kOriWin = 60, kOriInc = 5, kOriSamples = 113;       // points (i, j), |i|, |j| <= 6, i*i + j*j <= 36
// cv::fastAtan2 (degrees), which is also what cv::phase(X, Y, angle, true) evaluates per element
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
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
// resizeHaarPattern + calcHaarPattern for the two 2-box gradient wavelets (dx_s = {0,0,2,4,-1},{2,0,4,4,1}; dy_s = {0,0,4,2,1},{0,2,4,4,-1})
// of size `gws` at integral position p (row pitch sw): int box sum * float weight accumulated in double, as the detector does
__device__ __forceinline__ void ori_gradients(const int32_t* __restrict__ p, int sw, int gws, float* vx, float* vy)
{
    const float ratio = (float)gws / 4;
    const int c0 = cv_round_f(ratio * 0), c2 = cv_round_f(ratio * 2), c4 = cv_round_f(ratio * 4);
    auto box = [&](int x1, int y1, int x2, int y2) { return p[y1 * sw + x1] + p[y2 * sw + x2] - p[y2 * sw + x1] - p[y1 * sw + x2]; };
    {
        const float w0 = -1 / ((float)(c2 - c0) * (c4 - c0)), w1 = 1 / ((float)(c4 - c2) * (c4 - c0));
        double d = 0;
        d += box(c0, c0, c2, c4) * w0; d += box(c2, c0, c4, c4) * w1;
        *vx = (float)d;
    }
    {
        const float w0 = 1 / ((float)(c4 - c0) * (c2 - c0)), w1 = -1 / ((float)(c4 - c0) * (c4 - c2));
        double d = 0;
        d += box(c0, c0, c4, c2) * w0; d += box(c0, c2, c4, c4) * w1;
        *vy = (float)d;
    }
}

// One wave per keypoint: the 113 gradient samples (two per lane), then the 72 sliding windows -- lane d sums the samples of
// window 5d (and 5(d + 64)) in sample order, as the sequential loop does -- and the first largest window gives the direction.
__global__ __launch_bounds__(256) void k_surf_orientation(DescArgs a, int w, int h)
{
    const int im = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = *a.n[im];
    const int k = blockIdx.x * 4 + wv;
    __shared__ float sX[4][128], sY[4][128]; __shared__ int sA[4][128];        // the valid samples, compacted in sample order
    if (k >= n) return;
    uvo_keypoint kp = a.kps[im][k];
    const float s = kp.size * 1.2f / 9.0f;
    const int gws = 2 * cv_round_f(2 * s);
    const int sum_rows = h + 1, sw = w + 1;
    const int32_t* __restrict__ sum = a.sum[im];
    // sample kk is the kk-th (i, j) of the raster walk over [-6, 6]^2 with i*i + j*j <= 36: the 169 cells in three lane passes,
    // the valid ones compacted in that order with ballots
    int cnt_before = 0;
    for (int pass = 0; pass < 3; pass++) {
        const int cell = pass * 64 + lane;
        bool inside = false, valid = false; float vx = 0, vy = 0;
        if (cell < 169) {
            const int i = cell / 13 - kOriRadius, j = cell % 13 - kOriRadius;
            inside = i * i + j * j <= kOriRadius * kOriRadius;
            if (inside) {
                const int x = cv_round_f(kp.x + i * s - (float)(gws - 1) / 2), y = cv_round_f(kp.y + j * s - (float)(gws - 1) / 2);
                if (!(y < 0 || y >= sum_rows - gws || x < 0 || x >= sw - gws)) {
                    valid = true;
                    ori_gradients(sum + (size_t)y * sw + x, sw, gws, &vx, &vy);
                    const float wgt = a.ori_w[(i + kOriRadius) * 13 + (j + kOriRadius)];
                    vx *= wgt; vy *= wgt;
                }
            }
        }
        const unsigned long long m = __ballot(valid);
        if (valid) {
            const int pos = cnt_before + __popcll(m & ((1ull << lane) - 1ull));
            sX[wv][pos] = vx; sY[wv][pos] = vy; sA[wv][pos] = cv_round_f(fast_atan2_deg(vy, vx));
        }
        cnt_before += __popcll(m);
    }
    const int nangle = cnt_before;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (nangle == 0) { if (lane == 0) atomicAdd(a.ori_drop, 1); return; }      // OpenCV drops the keypoint: reported, never silently kept
    float best_mod = 0.f, bx = 0.f, by = 0.f; int best_i = 1 << 30;
    for (int d = lane; d < 360 / kOriInc; d += 64) {
        const int i = d * kOriInc;
        float sumx = 0.f, sumy = 0.f;
        for (int j = 0; j < nangle; j++) {
            const int dd = abs(sA[wv][j] - i);
            if (dd < kOriWin / 2 || dd > 360 - kOriWin / 2) { sumx += sX[wv][j]; sumy += sY[wv][j]; }
        }
        const float mod = sumx * sumx + sumy * sumy;
        if (mod > best_mod) { best_mod = mod; bx = sumx; by = sumy; best_i = i; }      // per lane: its windows in increasing order, strict >
    }
    // the sequential scan keeps the FIRST window with the largest modulus (strict >; a zero modulus never wins)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(best_mod, o), ox = __shfl_xor(bx, o), oy = __shfl_xor(by, o); const int oi = __shfl_xor(best_i, o);
        if (om > best_mod || (om == best_mod && oi < best_i)) { best_mod = om; bx = ox; by = oy; best_i = oi; }
    }
    if (lane == 0) a.kps[im][k].angle = fast_atan2_deg(-by, bx);
}

// The descriptor of one keypoint with a rotated sampling window, one workgroup per keypoint.  WIN[i][j] is the bilinear sample
// at start + i (sin, cos) + j (cos, -sin) with OpenCV's running sums (a float accumulated per row i, a double per column j), so
// thread i walks its row j = 0 .. win-1 and feeds cv::resize(INTER_AREA)'s horizontal pass on the fly; the vertical pass and the
// tail are the upright kernels'.  LDS: 21 x (win | 1) floats (dynamic) + the per-row start positions.
__global__ __launch_bounds__(256) void k_descriptor_rot(DescArgs a, int w, int h)
{
    const int im = blockIdx.y, tid = threadIdx.x;
    const int n = *a.n[im];
    extern __shared__ __align__(16) unsigned char smem_rot[];
    float* buf = reinterpret_cast<float*>(smem_rot);                    // [21][bp]
    __shared__ float s_sx[kMaxWin + 1], s_sy[kMaxWin + 1];
    __shared__ AreaTab tab[21];
    __shared__ int PATCH[21][21];
    const uint8_t* __restrict__ img = a.img[im];
    for (int k = blockIdx.x; k < n; k += gridDim.x) {
        const uvo_keypoint kp = a.kps[im][k];
        const float s = kp.size * 1.2f / 9.0f;
        const int win_size = (int)((20 + 1) * s);
        const int bp = win_size | 1;
        const int iscale = a.iscale[min(win_size, kMaxWin)];
        if (tid < 21) tab[tid] = a.tabs[min(win_size, kMaxWin) * 21 + tid];
        float descriptor_dir = kp.angle;
        descriptor_dir *= (float)(3.14159265358979323846 / 180);
        double sd, cd;
        det_sincos((double)descriptor_dir, &sd, &cd);                   // OpenCV: std::sin / std::cos of a float; here the double series rounded to float (glibc sinf/cosf round the same way but for rare ties)
        const float sin_dir = -(float)sd, cos_dir = (float)cd;
        if (tid == 0) {                                                 // start_x += sin_dir; start_y += cos_dir, row after row (float)
            const float win_offset = -(float)(win_size - 1) / 2;
            float start_x = kp.x + win_offset * cos_dir + win_offset * sin_dir;
            float start_y = kp.y - win_offset * sin_dir + win_offset * cos_dir;
            for (int i = 0; i < win_size; i++, start_x += sin_dir, start_y += cos_dir) { s_sx[i] = start_x; s_sy[i] = start_y; }
        }
        __syncthreads();
        const int ncols1 = w - 1, nrows1 = h - 1;
        for (int i = tid; i < win_size; i += 256) {
            double pixel_x = s_sx[i], pixel_y = s_sy[i];
            int dx = 0;
            float b = 0.f;                                              // running sum of destination column dx
            int isum = 0;
            AreaTab tx = tab[0];
            for (int j = 0; j < win_size; j++, pixel_x += cos_dir, pixel_y -= sin_dir) {
                const int ix = cv_floor_d(pixel_x), iy = cv_floor_d(pixel_y);
                int v;
                if ((unsigned)ix < (unsigned)ncols1 && (unsigned)iy < (unsigned)nrows1) {
                    const float fa = (float)(pixel_x - ix), fb = (float)(pixel_y - iy);
                    const uint8_t* ip = img + (size_t)iy * w + ix;
                    v = cv_round_f(ip[0] * (1.f - fa) * (1.f - fb) + ip[1] * fa * (1.f - fb) + ip[w] * (1.f - fa) * fb + ip[w + 1] * fa * fb) & 255;
                } else {
                    int x = cv_round_d(pixel_x), y = cv_round_d(pixel_y);
                    x = x > 0 ? x : 0; x = x < ncols1 ? x : ncols1; y = y > 0 ? y : 0; y = y < nrows1 ? y : nrows1;
                    v = img[(size_t)y * w + x];
                }
                if (iscale) {                                           // resizeAreaFast_: integer block sums along j
                    isum += v;
                    if ((j + 1) % iscale == 0) { if (dx < 21) reinterpret_cast<int*>(buf)[dx * bp + i] = isum; isum = 0; dx++; }
                } else {
                    // source column j belongs to destination column dx as its (partial) first tap, a middle tap, or its (partial)
                    // last tap -- and then also to column dx + 1 as that one's first tap
                    while (dx < 21) {
                        const int c_begin = tx.has_first ? tx.sx1 - 1 : tx.sx1, c_end = tx.has_last ? tx.sx2 + 1 : tx.sx2;
                        if (j < c_begin) break;
                        if (j < c_end) {
                            const float alpha = j < tx.sx1 ? tx.a_first : (j < tx.sx2 ? tx.a_mid : tx.a_last);
                            b += v * alpha;
                        }
                        if (j + 1 >= c_end) {                           // column dx is complete after this source column
                            buf[dx * bp + i] = b; b = 0.f; dx++;
                            if (dx < 21) tx = tab[dx];
                            continue;                                   // the same source column may open the next destination column
                        }
                        break;
                    }
                }
            }
        }
        __syncthreads();
        if (iscale) {
            for (int o = tid; o < 441; o += 256) {
                const int dy = o / 21, dx = o - dy * 21;
                int sum = 0;
                for (int sy = 0; sy < iscale; sy++) sum += reinterpret_cast<const int*>(buf)[dx * bp + dy * iscale + sy];
                int result;
                if (iscale == 2) result = (sum + 2) >> 2;
                else { float sc = 1.f / (iscale * iscale); result = sat_u8(sum * sc); }
                PATCH[dy][dx] = result;
            }
        } else {
            for (int o = tid; o < 441; o += 256) {
                const int dy = o / 21, dx = o - dy * 21;
                const AreaTab ty = tab[dy];
                const float* col = buf + dx * bp;
                float sum = 0.f;
                if (ty.has_first) sum += ty.a_first * col[ty.sx1 - 1];
                for (int r = ty.sx1; r < ty.sx2; r++) sum += ty.a_mid * col[r];
                if (ty.has_last) sum += ty.a_last * col[ty.sx2];
                PATCH[dy][dx] = sat_u8(sum);
            }
        }
        __syncthreads();
        describe_tail(a, im, k, PATCH);
        __syncthreads();
    }
}


// small windows: one workgroup per keypoint (block bx of nbx of the launch's small-window part)
__device__ __forceinline__ void descriptor64_small(const DescArgs& a, int w, int h, int bx, int nbx, int im)
{
    const int n = *a.n[im];
    // one workgroup per keypoint when the grid has max_kpts of them (measured faster than a smaller grid walking the list:
    // the hardware hands the next keypoint to whichever CU frees up); the loop covers smaller grids
    for (int k = bx; k < n; k += nbx) {
        describe_keypoint(a, w, h, k, im);
        __syncthreads();                     // the LDS buffers are reused by the next keypoint
    }
}
// Large windows (up to 739 samples: the keypoints of octaves 2 and 3).  One keypoint is 21 independent tasks, one per
// destination column dx of the area resize: a task needs only the ~win/21 image rows of that column's taps, computes
// buf[i][dx] for every window row i exactly as above, and finishes the 21 outputs PATCH[dy][dx] of its column.  A
// fixed grid walks the (keypoint, dx) tasks of the list built by k_rank_scatter, so the largest window is spread over
// 21 workgroups instead of serialising one; k_descriptor64_big_finish turns the patches into descriptors.
// One task per WAVE (no workgroup barriers: four independent waves per workgroup, 8192 waves resident).  A task is one
// destination column of a keypoint whose window is wider than kTripleWin, or three adjacent columns of a narrower one (their
// horizontal passes run one after the other into three thirds of the row buffer, then one vertical pass finishes the 63
// outputs, a lane each): the per-task work that does not depend on the window -- parameters, tables, the vertical pass,
// its 21-of-64 lanes -- is shared by three columns where the LDS allows it.
static const int kBigRow = 768;            // floats of row buffer per wave of the large-window part (a 739-pixel window + the 8 floats of slack area_column asks for; 3 x 256 for three-column tasks)
__device__ __forceinline__ void descriptor64_big(const DescArgs& a, int w, int h, uint8_t* __restrict__ patch, int bx, int nbx, int im)
{
    const int lane = threadIdx.x & 63, wv = sgpr_i(threadIdx.x >> 6);
    const int nb = a.big_n[im], nl = min(a.big_large[im], nb);        // the sorted list: nl wide windows first
    extern __shared__ __align__(16) unsigned char smem_desc[];        // shared with the small-window part: 4 x kBigRow floats here
    float* bufrow0 = reinterpret_cast<float*>(smem_desc) + wv * kBigRow;
    const uint8_t* __restrict__ img = a.img[im];
    const ImgRsrc rs = img_rsrc(img, w * h);
    // Tasks run down the size-sorted list (21 per wide keypoint, then 7 per narrower one) and are dealt to the waves in
    // rounds of alternating direction (round r hands task r*NW + p to wave p, or to wave NW-1-p when r is odd): costs span
    // 1..9 units, and this keeps the per-wave totals within about one task of each other, where a plain stride left the
    // waves that drew the giants running long after the rest.  (A shared atomic cursor does not work here: ~55k
    // device-scope increments of one address from 8 XCDs serialise at ~7 ns each.)
    const int nt1 = nl * 21, ntask = nt1 + (nb - nl) * 7, NW = nbx * 4, wid = bx * 4 + wv;
    auto task_of = [&](int r) { return r * NW + ((r & 1) ? NW - 1 - wid : wid); };
    auto entry_of = [&](int t) { return t < nt1 ? t / 21 : nl + (t - nt1) / 7; };
    // A task starts with two dependent fetches -- its keypoint's parameters, then that window size's resize table -- and a wave
    // runs only a few tasks, one after the other: both are fetched one task ahead (lane l keeps entry l % 21 of the table; the
    // column's entry is read out of lane dx with v_readlane), so a task begins with everything in registers.
    const int tl = lane % 21;
    int4 par_next = make_int4(0, 0, 0, 0);
    AreaTab tab_next = a.tabs[tl];
    if (task_of(0) < ntask) { par_next = a.big_par[im * a.cap + entry_of(task_of(0))]; tab_next = a.tabs[min(par_next.y & 0xFFFF, kMaxWin) * 21 + tl]; }
    for (int r = 0; r * NW < ntask; r++) {
        const int t = task_of(r);
        const int4 par = par_next;                            // (sorted index, win_size, start_x, start_y) from k_rank_scatter
        const AreaTab ty = tab_next;                          // entry lane % 21 of this task's resize table
        if (task_of(r + 1) < ntask) {                         // the next task's, in flight meanwhile
            par_next = a.big_par[im * a.cap + entry_of(task_of(r + 1))];
            tab_next = a.tabs[min(par_next.y & 0xFFFF, kMaxWin) * 21 + tl];
        }
        if (t >= ntask) continue;
        const int e = entry_of(t);
        const bool st_on = UVO_STAMPS && g_desc_stamps != nullptr && lane == 0;
        if (st_on) { long long* sp = g_desc_stamps + ((size_t)(2 + im) * 16384 + (t & 16383)) * 8; sp[0] = wall_clock64(); sp[6] = par.y & 0xFFFF; sp[7] = (t < nt1 ? 1 : 3) + 10 * (par.y >> 16 != 0); }
        const int ncols = t < nt1 ? 1 : 3, dx0 = t < nt1 ? t - e * 21 : 3 * ((t - nt1) - (e - nl) * 7);
        const int bstride = 256;                              // floats between the column buffers of a three-column task (kTripleWin + 8 <= 256)
        // the task is the same for every lane: scalar registers, so that row clamps, row addresses and tap weights are SALU work
        const int win_size = sgpr_i(par.y) & 0xFFFF, start_x = sgpr_i(par.z), start_y = sgpr_i(par.w);
        const int iscale = sgpr_i(par.y) >> 16;               // non-zero: resizeAreaFast_ with this integer scale (k_big_sort)
        const bool area_fast = iscale != 0;
        uint8_t* out = patch + ((size_t)im * a.cap + e) * kPatchStride;
        // rows 4-byte aligned: aligned 32-bit loads cover four columns at a time over the window's in-image columns
        // [xlo, xhi); window columns left or right of the image replicate the border column (WIN clamps x), so their
        // sums are copies of the first / last in-image column's
        const int xlo = start_x > 0 ? start_x : 0, xhi = start_x + win_size < w ? start_x + win_size : w;
        const bool vec_ok = (w & 3) == 0 && xlo < xhi;
        const int xa = xlo & ~3, ilo = xlo - start_x, ihi = xhi - start_x;
        if (area_fast) {
            // resizeAreaFast_: integer block sums (any order); column c = dy*iscale + sy of the window
            for (int d = 0; d < ncols; d++) {
                const int dx = dx0 + d;
                int* colsum = reinterpret_cast<int*>(bufrow0 + d * bstride);
                if (vec_ok) {
                    for (int x4 = xa + 4 * lane; x4 < xhi; x4 += 256) {
                        int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
                        for (int sx = 0; sx < iscale; sx++) {
                            int y = start_y - (dx * iscale + sx); y = y > 0 ? y : 0; y = y < h - 1 ? y : h - 1;
                            const unsigned v = *reinterpret_cast<const unsigned*>(img + ((unsigned)(y * w) + (unsigned)x4));
                            s0 += v & 255u; s1 += (v >> 8) & 255u; s2 += (v >> 16) & 255u; s3 += v >> 24;
                        }
                        const int c = x4 - start_x;
                        if (c >= ilo) colsum[c] = s0;
                        if (c + 1 >= ilo && c + 1 < ihi) colsum[c + 1] = s1;
                        if (c + 2 >= ilo && c + 2 < ihi) colsum[c + 2] = s2;
                        if (c + 3 >= ilo && c + 3 < ihi) colsum[c + 3] = s3;
                    }
                    if (ilo > 0) { const int v = colsum[ilo]; for (int c = lane; c < ilo; c += 64) colsum[c] = v; }
                    if (ihi < win_size) { const int v = colsum[ihi - 1]; for (int c = ihi + lane; c < win_size; c += 64) colsum[c] = v; }
                } else
                for (int c = lane; c < win_size; c += 64) {
                    int x = start_x + c; x = x > 0 ? x : 0; x = x < w - 1 ? x : w - 1;
                    int sum = 0;
                    for (int sx = 0; sx < iscale; sx++) {
                        int y = start_y - (dx * iscale + sx); y = y > 0 ? y : 0; y = y < h - 1 ? y : h - 1;
                        sum += img[(size_t)y * w + x];
                    }
                    colsum[c] = sum;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < 21 * ncols) {
                const int d = lane / 21, dy = lane - d * 21;
                const int* colsum = reinterpret_cast<const int*>(bufrow0 + d * bstride);
                int sum = 0;
                for (int sy = 0; sy < iscale; sy++) sum += colsum[dy * iscale + sy];
                int result;
                if (iscale == 2) result = (sum + 2) >> 2;
                else { float sc = 1.f / (iscale * iscale); result = sat_u8(sum * sc); }
                out[dy * 21 + dx0 + d] = (uint8_t)result;
            }
        } else {
            for (int d = 0; d < ncols; d++) {
                const int dx = dx0 + d;
                float* bufrow = bufrow0 + d * bstride;
                if (vec_ok) {
                    // rows 4-byte aligned: the shared column core (aligned dword loads, four pixels per lane and chunk of 256 columns)
                    const ColTask ct = col_task(ty, dx, start_y, w, h);
                    const int span = xhi - xa;
                    if (span <= 256) area_column<4, 1>(rs, lane, start_x, win_size, ct, bufrow);
                    else if (span <= 512) area_column<4, 2>(rs, lane, start_x, win_size, ct, bufrow);
                    else area_column<4, 3>(rs, lane, start_x, win_size, ct, bufrow);
                    continue;
                }
                AreaTab tx;                                    // the column's entry, the same for every lane: scalar registers
                tx.sx1 = __builtin_amdgcn_readlane(ty.sx1, dx); tx.sx2 = __builtin_amdgcn_readlane(ty.sx2, dx);
                tx.a_first = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ty.a_first), dx));
                tx.a_mid = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ty.a_mid), dx));
                tx.a_last = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ty.a_last), dx));
                tx.has_first = __builtin_amdgcn_readlane((int)ty.has_first, dx) != 0; tx.has_last = __builtin_amdgcn_readlane((int)ty.has_last, dx) != 0;
                const int c_begin = tx.has_first ? tx.sx1 - 1 : tx.sx1;
                const int c_end = tx.has_last ? tx.sx2 + 1 : tx.sx2;
                for (int i = lane; i < win_size; i += 64) {    // image rows not a multiple of four bytes: a pixel per lane and tap
                    int x = start_x + i; x = x > 0 ? x : 0; x = x < w - 1 ? x : w - 1;
                    float b = 0.f;
                    // taps fetched eight at a time (independent loads in flight), accumulated in order
                    for (int c0 = c_begin; c0 < c_end; c0 += 8) {
                        int v[8];
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            int y = start_y - (c0 + q); y = y > 0 ? y : 0; y = y < h - 1 ? y : h - 1;
                            v[q] = img[(size_t)y * w + x];
                        }
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            int cc = c0 + q;
                            float alpha = cc < tx.sx1 ? tx.a_first : (cc < tx.sx2 ? tx.a_mid : tx.a_last);
                            if (cc < c_end) b += v[q] * alpha;
                        }
                    }
                    bufrow[i] = b;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (st_on) g_desc_stamps[((size_t)(2 + im) * 16384 + (t & 16383)) * 8 + 1] = wall_clock64();
            if (lane < 21 * ncols) {                        // vertical passes of the task's columns, 21 lanes each
                const int d = lane / 21, dy = lane - d * 21;
                const float* bufrow = bufrow0 + d * bstride + (vec_ok ? (start_x & 3) : 0);      // area_column's alignment shift
                float sum = 0.f;
                if (ty.has_first) sum += ty.a_first * bufrow[ty.sx1 - 1];            // the loop of resizeArea_'s table, split by weight
                for (int r = ty.sx1; r < ty.sx2; r++) sum += ty.a_mid * bufrow[r];
                if (ty.has_last) sum += ty.a_last * bufrow[ty.sx2];
                out[dy * 21 + dx0 + d] = sat_u8(sum);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (st_on) g_desc_stamps[((size_t)(2 + im) * 16384 + (t & 16383)) * 8 + 2] = wall_clock64();
    }
}
// Both window classes in one launch: blocks [0, nbig) are the persistent waves of the large-window tasks, the rest take one
// small-window keypoint each.  The large-window part stalls on its per-tap dependency chains, the small-window part on its
// barriers; resident together they keep the VALU busier than one after the other (and a launch is saved).
// (8 waves per SIMD: the compiler would settle at 66 VGPRs = 7 waves)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_descriptor64(LanePair lp, int w, int h, int nbig, int part, int nim, int order)
{
    // The launch is one row of workgroups; `order` says who comes first (the hardware starts workgroups in index order, and with
    // lifetimes of 4 to 14 us that order decides what is resident together): 0: image by image, each image's large-window waves before
    // its small-window workgroups (rounds 2-4); 1: every image's large-window waves, then the small-window workgroups image by image;
    // 2: the same with the small-window workgroups of the images alternating
    const int nsmall = (int)gridDim.x / nim - nbig;
    int id = blockIdx.x, yy, bx; bool big;
    if (order == 0) { yy = id / (nbig + nsmall); bx = id - yy * (nbig + nsmall); big = bx < nbig; if (!big) bx -= nbig; }
    else if (id < nim * nbig) { yy = id / nbig; bx = id - yy * nbig; big = true; }
    else { id -= nim * nbig; big = false; if (order == 1) { yy = id / nsmall; bx = id - yy * nsmall; } else { yy = id % nim; bx = id / nim; } }
    const LaneArgs& LA = UVO_LANE_OF(lp, yy); const DescArgs& a = LA.da; uint8_t* __restrict__ patch = LA.patch;
    // part (UVO_DESC_PART, measurement only -- the other class of keypoints gets no descriptor): 1 = large windows only, 2 = small only
    if (big) { if (part != 2) descriptor64_big(a, w, h, patch, bx, nbig, UVO_LANE_IM(yy)); }
    else if (part != 1) descriptor64_small(a, w, h, bx, nsmall, UVO_LANE_IM(yy));
}
__global__ __launch_bounds__(256) void k_descriptor64_big_finish(LanePair lp)
{
    const LaneArgs& LA = UVO_LANE_OF(lp, blockIdx.y); const DescArgs& a = LA.da; const uint8_t* __restrict__ patch = LA.patch;
    const int im = UVO_LANE_IM(blockIdx.y), tid = threadIdx.x;
    const int nb = a.big_n[im];
    __shared__ int PATCH[21][21];
    for (int e = blockIdx.x; e < nb; e += gridDim.x) {
        const uint8_t* src = patch + ((size_t)im * a.cap + e) * kPatchStride;
        for (int o = tid; o < 441; o += 256) PATCH[o / 21][o % 21] = src[o];
        __syncthreads();
        describe_tail(a, im, a.big_par[im * a.cap + e].x, PATCH);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// resize tables and integer-scale flags of every window size, once per context
uvo_status surf_build_area_tables(Ctx* c)
{
    std::vector<AreaTab> tabs((size_t)(kMaxWin + 1) * 21);
    std::vector<int> isc(kMaxWin + 1, 0);
    memset(tabs.data(), 0, tabs.size() * sizeof(AreaTab));
    for (int win = 1; win <= kMaxWin; win++) {
        const double scale = 1. / ((double)21 / win);              // cv::resize: inv_scale_x = dsize.width / ssize.width, scale_x = 1. / inv_scale_x
        for (int d = 0; d < 21; d++) tabs[(size_t)win * 21 + d] = area_tab(d, win, scale);
        const int iscale = cv_round_d(scale);
        if (fabs(scale - iscale) < DBL_EPSILON) isc[win] = iscale;
    }
    UVO_HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_area_tabs), tabs.size() * sizeof(AreaTab)));
    UVO_HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_area_iscale), isc.size() * sizeof(int)));
    UVO_HIP_TRY(c, hipMemcpy(c->d_area_tabs, tabs.data(), tabs.size() * sizeof(AreaTab), hipMemcpyHostToDevice));
    UVO_HIP_TRY(c, hipMemcpy(c->d_area_iscale, isc.data(), isc.size() * sizeof(int), hipMemcpyHostToDevice));
    // SURFInvoker ctor: getGaussianKernel(2 * ORI_RADIUS + 1, SURF_ORI_SIGMA = 2.5, CV_32F), weights G[i] * G[j] of the orientation samples
    {
        const int N = 2 * kOriRadius + 1;
        double t[N], sum = 0; float G[N], W[N * N];
        const double sigma = 2.5f, scale2X = -0.5 / (sigma * sigma);
        for (int i = 0; i < N; i++) { const double x = i - (N - 1) * 0.5; t[i] = exp(scale2X * x * x); sum += t[i]; }
        sum = 1. / sum;
        for (int i = 0; i < N; i++) G[i] = (float)(t[i] * sum);
        for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) W[i * N + j] = G[i] * G[j];
        UVO_HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_ori_w), sizeof(W)));
        UVO_HIP_TRY(c, hipMemcpy(c->d_ori_w, W, sizeof(W), hipMemcpyHostToDevice));
    }
    return UVO_OK;
}

uvo_status surf_upload(Ctx* c, int slot, const uint8_t* gray, int w, int h, int stride, int mem)
{
    if (!gray || w <= 0 || h <= 0 || w > c->max_w || h > c->max_h || stride < w || slot < 0 || slot > 1) {
        c->err = "surf_upload: bad image geometry (w/h exceed the context's max_w/max_h?)";
        return UVO_INVALID_ARG;
    }
    if (slot == 1 && (w != c->img_w || h != c->img_h)) { c->err = "left/right image sizes differ"; return UVO_INVALID_ARG; }
    c->img_w = w; c->img_h = h;
    // A tight, 16-byte aligned image that is already in device memory is read in place: the caller keeps it valid and unmodified
    // until the pair is collected (include/uvo_hip.h, "memory"), which is what the copy needed too -- it is queued, not done, when
    // submit returns.  Saves two 2 MB device-to-device copies and two launches per pair.
    if (mem == UVO_MEM_DEVICE && stride == w && (reinterpret_cast<uintptr_t>(gray) & 15u) == 0) { c->img[slot] = gray; return UVO_OK; }
    UVO_HIP_TRY(c, hipMemcpy2DAsync(c->d_img[slot], w, gray, stride, w, h,
                                    mem == UVO_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    c->img[slot] = c->d_img[slot];
    return UVO_OK;
}

// what the detector's kernels need of lane c (see LaneArgs); gate_min_features >= 0: the rank scatter also evaluates VO:556
static LaneArgs lane_args(Ctx* c, int nimg, int gate_min_features)
{
    LaneArgs a;
    a.ip = ImgPair{ { c->img[0], c->img[1] }, { c->d_sum[0], c->d_sum[1] }, { c->d_planes[0], c->d_planes[1] }, c->plane_pw, c->plane_stride, c->d_cand_n, c->d_big_n, c->d_counts + CN_SURV };
    a.part = c->d_colpart;
    a.sv = SurvOut{ c->d_surv, c->d_counts + CN_SURV, c->surv_cap };
    a.out = CandOut{ { c->d_cand[0], c->d_cand[1] }, c->d_cand_n, c->cap };
    a.sa = SortArgs{ { c->d_cand[0], c->d_cand[1] }, c->d_cand_n, { c->det[0].kps, c->det[1].kps }, { c->det[0].n, c->det[1].n }, c->d_rank, c->cap, c->d_big_par, c->d_big_n,
                     gate_min_features >= 0 && nimg == 2 ? c->d_counts + CN_NQA : nullptr, gate_min_features };
    a.da = DescArgs{ { c->img[0], c->img[1] }, { c->det[0].kps, c->det[1].kps }, { c->det[0].desc, c->det[1].desc }, { c->det[0].n, c->det[1].n }, c->d_DW,
                     c->d_big_par + (size_t)2 * c->cap, c->d_big_n, c->d_counts + CN_BIGL0, c->cap,      // the sorted list
                     c->d_area_tabs, c->d_area_iscale, c->p.SURF_EXTENDED ? 1 : 0, { c->d_sum[0], c->d_sum[1] }, c->d_ori_w, c->d_counts + CN_ORI_DROP };
    a.patch = c->d_big_patch;
    a.big_in = c->d_big_par; a.big_out = c->d_big_par + (size_t)2 * c->cap; a.big_large = c->d_counts + CN_BIGL0;
    return a;
}
// one lane, or the two lanes of a two-pair launch (c2; both hold image pairs of the same size, kernels go to c's stream)
static LanePair lane_pair(Ctx* c, Ctx* c2, int nimg, int gate_min_features)
{
    LanePair lp;
    lp.a[0] = lane_args(c, nimg, gate_min_features);
    lp.a[1] = c2 ? lane_args(c2, nimg, gate_min_features) : lp.a[0];
    return lp;
}

static uvo_status surf_integral_lanes(Ctx* c, const LanePair& lp, int nim)       // nim: images over all lanes of the launch
{
    const int w = c->img_w, h = c->img_h, sw = w + 1;
    const int nstrip = (h + kStripRows - 1) / kStripRows, cstride = c->colpart_stride;
    StageTimer t(c, ST_INTEGRAL);
    hipLaunchKernelGGL(k_integral_strip_sums, dim3(nstrip, nim), dim3(256), 0, c->stream, lp, w, h, cstride, nstrip);
    hipLaunchKernelGGL(k_integral_strip_scan, dim3((sw + 31) / 32, nim), dim3(256), 0, c->stream, lp, cstride, nstrip, w);
    hipLaunchKernelGGL(k_integral_strip_final, dim3(nstrip, nim), dim3(256), 0, c->stream, lp, w, h, cstride, nstrip);
    UVO_HIP_TRY(c, hipGetLastError());
    return UVO_OK;
}
uvo_status surf_integral(Ctx* c, int nimg) { return surf_integral_lanes(c, lane_pair(c, nullptr, nimg, -1), nimg); }

template <int O, int TW, int TH, int NT>
static hipError_t launch_hessian_p(Ctx* c, const LanePair& lp, int nim, const OctavePat& op, float thr)
{
    const int w = c->img_w, h = c->img_h;
    dim3 grid((op.cols + TW - 3) / (TW - 2), (op.rows + TH - 3) / (TH - 2), nim);
    hipLaunchKernelGGL((k_hessian_nms_p<O, TW, TH, NT>), grid, dim3(NT), 0, c->stream, lp, w, h, op, thr);
    return hipGetLastError();
}

template <int O, int TW, int TH, int NT>
static hipError_t launch_hessian_c(Ctx* c, const LanePair& lp, int nim, const OctavePat& op, float thr)
{
    using OC = OctC<O>;
    constexpr int STEP = OC::STEP;
    constexpr int THs = (TH - 1) * STEP + (OC::HI - OC::LO) + 1;
    constexpr int PW = OctTile<O, TW>::PW;
    const int w = c->img_w, h = c->img_h;
    const size_t lds = sizeof(float) * 3 * TW * TH + sizeof(int32_t) * (size_t)THs * STEP * PW + sizeof(unsigned) * NmsLds<TW, TH>::kWords;
    dim3 grid((op.cols + TW - 3) / (TW - 2), (op.rows + TH - 3) / (TH - 2), nim);
    auto kern = k_hessian_nms_c<O, TW, TH, NT>;
    static bool attr_dev[64] = {false};                       // once per device (a process may hold contexts on several)
    bool& attr_set = attr_dev[c->device & 63];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, c->stream, lp, w, h, op, thr);
    return hipGetLastError();
}

// Geometry of the merged detection launch (all four octaves in one grid) for a w x h image
struct MergedGrid { int nbx0, nb0, nbx1, nb1, nb23, total; P23Grid g; };
static const int kMergedTW0 = 64, kMergedTH0 = 32, kMergedTW1 = 32;
static MergedGrid merged_grid(const OctavePat* ops)
{
    constexpr int TW0 = kMergedTW0, TH0 = kMergedTH0, TW1 = kMergedTW1, TH1 = kO1TileRows;
    MergedGrid m;
    m.nbx0 = (ops[0].cols + TW0 - 3) / (TW0 - 2); m.nb0 = m.nbx0 * ((ops[0].rows + TH0 - 3) / (TH0 - 2));
    m.nbx1 = (ops[1].cols + TW1 - 3) / (TW1 - 2); m.nb1 = m.nbx1 * ((ops[1].rows + TH1 - 3) / (TH1 - 2));
    m.g.nbx2 = (ops[2].cols + 32 - 3) / (32 - 2); m.g.nb2 = m.g.nbx2 * ((ops[2].rows + 16 - 3) / (16 - 2));
    m.g.nbx3 = (ops[3].cols + 16 - 3) / (16 - 2); m.g.nb3 = m.g.nbx3 * ((ops[3].rows + 16 - 3) / (16 - 2));
    m.nb23 = m.g.nb2 + m.g.nb3; m.total = m.nb0 + m.nb1 + m.nb23;
    return m;
}
static bool merged_launch(const Ctx* c)
{
    static const bool split_env = getenv("UVO_HESSIAN_SPLIT") != nullptr;       // diagnostic: one launch per octave
    return c->p.SURF_OCTAVES_NUMBER == 4 && !split_env;
}

// Everything the detector keeps per image SIZE (and octave count) on this lane: the octave patterns for k_hessian_finish and the
// block order of the merged detection launch.  Idempotent; surf_detect calls it for the image at hand, and the pipeline calls it
// for every lane as soon as the stream's image size is known, so that a lane's first pair does not pay two allocations, two
// uploads and four host syncs inside somebody's timed loop.
uvo_status surf_prepare(Ctx* c, int w, int h)
{
    if (c->p.SURF_OCTAVES_NUMBER < 1 || c->p.SURF_OCTAVES_NUMBER > 4 || c->p.SURF_OCTAVES_LAYERS != 3) {
        c->err = "SURF: supported nOctaves 1..4, nOctaveLayers 3";
        return UVO_INVALID_ARG;
    }
    OctavePat ops[4];
    memset(ops, 0, sizeof(ops));
    for (int o = 0; o < c->p.SURF_OCTAVES_NUMBER; o++) make_octave(o, c->p.SURF_OCTAVES_LAYERS, w, h, &ops[o]);
    if (!c->d_octpat) UVO_HIP_TRY(c, hipMalloc(&c->d_octpat, sizeof(ops)));
    if (c->h_octpat.size() != sizeof(ops) || memcmp(c->h_octpat.data(), ops, sizeof(ops)) != 0) {      // new image size or octave count
        UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));          // nothing may still read the old table or the staging copy
        c->h_octpat.assign(reinterpret_cast<const unsigned char*>(ops), reinterpret_cast<const unsigned char*>(ops) + sizeof(ops));
        UVO_HIP_TRY(c, hipMemcpyAsync(c->d_octpat, c->h_octpat.data(), sizeof(ops), hipMemcpyHostToDevice, c->stream));
        UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    if (!merged_launch(c)) return UVO_OK;
    const MergedGrid m = merged_grid(ops);
    const int nb0 = m.nb0, nb1 = m.nb1, nb23 = m.nb23, total = m.total;
    if (nb0 > 0x3FFF || nb1 > 0x3FFF || nb23 > 0x3FFF) { c->err = "SURF: image too large for the merged detection launch's tile table"; return UVO_INVALID_ARG; }
    const int ng[3] = { nb0, nb1, nb23 }, ngroups = total;
    if ((int)c->h_hess_order.size() != ngroups || c->hess_order_key[0] != nb0 || c->hess_order_key[1] != nb1 || c->hess_order_key[2] != nb23) {
        // block b's kind and tile: the i-th tile of a kind with n tiles sits at position (i + 1/2) / n of the launch
        // (an entry is the block's octave and its tile's column and row, 2 + 14 + 14 bits: one scalar load at the top of the kernel;
        // as 16-bit entries -- kind and tile index -- it took a vector load, a wait and two integer divisions before a tile knew its place)
        std::vector<std::pair<double, uint32_t>> pos;
        pos.reserve((size_t)ngroups);
        for (int k = 0; k < 3; k++) for (int i = 0; i < ng[k]; i++) {
            uint32_t e;
            if (k == 0) e = (0u << 30) | ((uint32_t)(i / m.nbx0) << 14) | (uint32_t)(i % m.nbx0);
            else if (k == 1) e = (1u << 30) | ((uint32_t)(i / m.nbx1) << 14) | (uint32_t)(i % m.nbx1);
            else if (i < m.g.nb2) e = (2u << 30) | ((uint32_t)(i / m.g.nbx2) << 14) | (uint32_t)(i % m.g.nbx2);
            else e = (3u << 30) | ((uint32_t)((i - m.g.nb2) / m.g.nbx3) << 14) | (uint32_t)((i - m.g.nb2) % m.g.nbx3);
            pos.emplace_back((i + 0.5) / ng[k] + k * 1e-9, e);
        }
        std::sort(pos.begin(), pos.end());
        c->h_hess_order.resize((size_t)ngroups);
        for (int b = 0; b < ngroups; b++) c->h_hess_order[(size_t)b] = pos[(size_t)b].second;
        UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));               // nothing may still read the old table
        if (c->d_hess_order_cap < (size_t)ngroups) {
            if (c->d_hess_order) (void)hipFree(c->d_hess_order);
            c->d_hess_order = nullptr; c->d_hess_order_cap = 0;
            UVO_HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_hess_order), sizeof(uint32_t) * (size_t)ngroups));
            c->d_hess_order_cap = (size_t)ngroups;
        }
        UVO_HIP_TRY(c, hipMemcpyAsync(c->d_hess_order, c->h_hess_order.data(), sizeof(uint32_t) * (size_t)ngroups, hipMemcpyHostToDevice, c->stream));
        UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->hess_order_key[0] = nb0; c->hess_order_key[1] = nb1; c->hess_order_key[2] = nb23;
    }
    return UVO_OK;
}

// c2 != nullptr: the detector stages of TWO lanes (two consecutive stereo pairs of one stream, same image size) in one launch
// each, queued on c's stream; results land in each lane's own buffers exactly as two calls would leave them.
uvo_status surf_detect_lanes(Ctx* c, Ctx* c2, int nimg, int gate_min_features)
{
    const int w = c->img_w, h = c->img_h;
    if (c2 && (nimg != 2 || c2->img_w != w || c2->img_h != h || !c->p.SURF_UPRIGHT)) { c->err = "two-pair launch: upright SURF on two image pairs of one size"; return UVO_INVALID_ARG; }
    UVO_TRY(surf_prepare(c, w, h));
    if (c2) { const uvo_status st2 = surf_prepare(c2, w, h); if (st2 != UVO_OK) { c->err = c2->err; return st2; } }
    const LanePair lp = lane_pair(c, c2, nimg, gate_min_features);
    const int nim = c2 ? 2 * nimg : nimg, nlanes = c2 ? 2 : 1;
    UVO_TRY(surf_integral_lanes(c, lp, nim));
    const float thr = (float)c->p.SURF_MIN_HESSIAN;
    {
        OctavePat ops[4];
        memset(ops, 0, sizeof(ops));
        for (int o = 0; o < c->p.SURF_OCTAVES_NUMBER; o++) make_octave(o, c->p.SURF_OCTAVES_LAYERS, w, h, &ops[o]);
        // four octaves (the shipped configuration): one launch for all of them
        const bool merged = merged_launch(c);
        if (merged) {
            constexpr int TW0 = kMergedTW0, TH0 = kMergedTH0, TW1 = kMergedTW1, TH1 = kO1TileRows;
            constexpr int THs0 = (TH0 - 1) * OctC<0>::STEP + (OctC<0>::HI - OctC<0>::LO) + 1, THs1 = (TH1 - 1) * OctC<1>::STEP + (OctC<1>::HI - OctC<1>::LO) + 1;
            constexpr size_t lds0 = sizeof(float) * 3 * TW0 * TH0 + sizeof(int32_t) * (size_t)THs0 * OctC<0>::STEP * OctTile<0, TW0>::PW + sizeof(unsigned) * NmsLds<TW0, TH0>::kWords;
            constexpr size_t lds1 = sizeof(float) * 3 * TW1 * TH1 + sizeof(int32_t) * (size_t)THs1 * OctC<1>::STEP * OctTile<1, TW1>::PW + sizeof(unsigned) * NmsLds<TW1, TH1>::kWords;
            constexpr size_t lds = lds0 > lds1 ? lds0 : lds1;
            static_assert(lds <= 54600, "three blocks of the merged detection launch must fit a CU's 160 KB of LDS");
            const MergedGrid mg = merged_grid(ops);
            const HessGrid hg = { mg.nbx0, mg.nb0, mg.nbx1, mg.nb1, mg.g };
            const int total = mg.total;
            auto kern = k_hessian_nms_all<TW0, TH0>;
            static bool attr_dev[64] = {false};
            bool& attr_set = attr_dev[c->device & 63];
            if (!attr_set) { UVO_HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr_set = true; }
            StageTimer t(c, ST_HESSIAN_O0);
            static const size_t lds_pad = getenv("UVO_HESS_LDS") ? (size_t)atoi(getenv("UVO_HESS_LDS")) : 0;      // measurement: fewer blocks per CU
            const size_t lds_launch = lds_pad > lds ? lds_pad : lds;
            if (lds_pad > lds) { static bool once = false; if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_launch); once = true; } }
            static const char* stamps_path = UVO_STAMPS ? getenv("UVO_HESS_STAMPS") : nullptr;
            static long long* d_stamps = nullptr; static size_t stamps_n = 0;
            if (stamps_path && c->lane_id == 0 && !c->master) {
                if (!d_stamps) { stamps_n = (size_t)total * 4 * 8; (void)hipMalloc(reinterpret_cast<void**>(&d_stamps), sizeof(long long) * stamps_n); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_hess_stamps), &d_stamps, sizeof(d_stamps)); }
                (void)hipMemsetAsync(d_stamps, 0, sizeof(long long) * stamps_n, c->stream);
            }
            HessDims hd;
            for (int o = 0; o < 4; o++) {
                OctDims& d = hd.o[o];
                for (int l = 0; l < 5; l++) { d.L[l].samples_i = ops[o].L[l].samples_i; d.L[l].samples_j = ops[o].L[l].samples_j; }
                d.rows = ops[o].rows; d.cols = ops[o].cols; d.octave = ops[o].octave;
                for (int l = 0; l < 3; l++) d.nms_margin[l] = ops[o].nms_margin[l];
            }
            Ctx::TraceRec* trh = (c->trace_on && c->trace_cur >= 0) ? &c->trace[c->trace_cur] : nullptr;      // uvo_trace_row::dev_ms[6], [7]
            if (trh) (void)hipEventRecord(trh->ev[6], c->stream);
            hipLaunchKernelGGL(kern, dim3(total, nim), dim3(kP23Threads), lds_launch, c->stream, lp, w, h, hd, thr, c->d_hess_order);
            if (trh) { (void)hipEventRecord(trh->ev[7], c->stream); trh->det_marked = true; }
            if (stamps_path && d_stamps && c->lane_id == 0 && !c->master) {        // (measurement: synchronous, every launch rewrites the file)
                std::vector<long long> hs(stamps_n);
                (void)hipStreamSynchronize(c->stream);
                (void)hipMemcpy(hs.data(), d_stamps, sizeof(long long) * stamps_n, hipMemcpyDeviceToHost);
                if (FILE* f = fopen(stamps_path, "w")) {
                    fprintf(f, "block,im,kind,t_start,t_filled,t_det,t_end,t_scan,t_scanbar,t_atomic\n");
                    for (int y = 0; y < nim; y++) for (int b = 0; b < total; b++) { const long long* r = hs.data() + ((size_t)y * total + b) * 8; fprintf(f, "%d,%d,%lld,%lld,%lld,%lld,%lld,%lld,%lld,%lld\n", b, y, r[4], r[0], r[1], r[2], r[3], r[5], r[6], r[7]); }
                    fclose(f);
                }
            }
            UVO_HIP_TRY(c, hipGetLastError());
        }
        for (int o = 0; o < c->p.SURF_OCTAVES_NUMBER && !merged; o++) {
            StageTimer t(c, ST_HESSIAN_O0 + o);
            const OctavePat& op = ops[o];
            hipError_t e;
            // tile shapes chosen by pipelined throughput (tools/probe/ab.sh): octave 0 64 x 24 samples = 40 KB of LDS (18 KB det planes,
            // 19 KB integral tile, 3 KB survivor list), octave 1 32 x 24 = 59 KB (9 KB + 48 KB: the 54-pixel templates make the halo
            // most of the tile)
            if (o == 0)      e = launch_hessian_c<0, 64, 24, 512>(c, lp, nim, op, thr);
            else if (o == 1) e = launch_hessian_c<1, 32, 24, 512>(c, lp, nim, op, thr);
            else if (o == 2) e = launch_hessian_p<2, 32, 16, 512>(c, lp, nim, op, thr);
            else             e = launch_hessian_p<3, 16, 16, 256>(c, lp, nim, op, thr);
            UVO_HIP_TRY(c, e);
        }
        {   // the survivors of every octave: outer-layer determinants, last comparison, keypoints
            StageTimer t(c, ST_HESSIAN_O0 + c->p.SURF_OCTAVES_NUMBER - 1);
            hipLaunchKernelGGL(k_hessian_finish, dim3((c->surv_cap + kFinishPerWg - 1) / kFinishPerWg, nlanes), dim3(256), 0, c->stream, lp, static_cast<const OctavePat*>(c->d_octpat), w, h);
            UVO_HIP_TRY(c, hipGetLastError());
        }
        static const int probe_thin = getenv("UVO_PROBE_THIN_US") ? atoi(getenv("UVO_PROBE_THIN_US")) : 0, probe_fat = getenv("UVO_PROBE_FAT_US") ? atoi(getenv("UVO_PROBE_FAT_US")) : 0;
        if (probe_thin > 0) hipLaunchKernelGGL(k_probe_hold, dim3(1), dim3(64), 64, c->stream, probe_thin * 100);           // wall_clock64: 100 MHz
        if (probe_fat > 0) {
            static bool once = false;
            if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe_hold), hipFuncAttributeMaxDynamicSharedMemorySize, 53000); once = true; }
            hipLaunchKernelGGL(k_probe_hold, dim3(768), dim3(256), 53000, c->stream, probe_fat * 100);
        }
    }
    {
        StageTimer t(c, ST_SORT);
        const int tiles_max = ((c->cap + 255) / 256) * ((c->cap + kSortChunk - 1) / kSortChunk);
        hipLaunchKernelGGL(k_rank_partial, dim3(tiles_max < 512 ? tiles_max : 512, nim), dim3(256), 0, c->stream, lp);
        hipLaunchKernelGGL(k_rank_scatter, dim3((c->cap + 255) / 256, nim), dim3(256), 0, c->stream, lp);
        UVO_HIP_TRY(c, hipGetLastError());
    }
    {
        StageTimer t(c, ST_DESCRIPTOR);
        const size_t lds_small = sizeof(float) * 21 * ((kSmallWin + 3) | 1), lds_big = sizeof(float) * 4 * kBigRow;
        hipLaunchKernelGGL(k_big_sort, dim3(nim), dim3(1024), 0, c->stream, lp, c->cap, c->d_area_iscale);
        if (c->p.SURF_UPRIGHT) {
            const int nbig = 1024;                             // 8192 persistent waves for the large-window tasks (512: 80 us, 768..2048: 66-69 us)
            static const int desc_part = getenv("UVO_DESC_PART") ? atoi(getenv("UVO_DESC_PART")) : 0;      // measurement only
            static const char* dstamps_path = UVO_STAMPS ? getenv("UVO_DESC_STAMPS") : nullptr;
            static long long* d_dstamps = nullptr; const size_t dstamps_n = (size_t)4 * 16384 * 8;
            const bool dstamp = dstamps_path && c->lane_id == 0 && !c->master && nim == 2;
            if (dstamp) {
                if (!d_dstamps) { (void)hipMalloc(reinterpret_cast<void**>(&d_dstamps), sizeof(long long) * dstamps_n); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_desc_stamps), &d_dstamps, sizeof(d_dstamps)); }
                (void)hipMemsetAsync(d_dstamps, 0, sizeof(long long) * dstamps_n, c->stream);
            }
            // small-window part: a workgroup per keypoint -- as many as the last pair whose counts reached the host had, plus a quarter
            // (the workgroups walk the list when there are more: any grid gives the same descriptors); max_kpts of them before that
            const Ctx* hm = c->master ? c->master : c;
            static const int fixed_grid = getenv("UVO_DESC_GRID") ? atoi(getenv("UVO_DESC_GRID")) : 0;              // measurement: 0 = by the hint, -1 = max_kpts, n = n
            int nsmall = hm->kp_hint > 0 ? ((hm->kp_hint + hm->kp_hint / 4 + 255) & ~255) : c->cap;
            if (fixed_grid != 0) nsmall = fixed_grid < 0 ? c->cap : fixed_grid;
            nsmall = nsmall < 512 ? 512 : (nsmall > c->cap ? c->cap : nsmall);
            // order 2 + the hint-sized grid: 57.2 us; image by image on max_kpts workgroups (rounds 2-4): 60.5; large-window waves first,
            // small-window workgroups image by image: 59.7; 512 / 2048 large-window workgroups per image: 58.1 / 58.9 (tools/probe/desc_order.sh)
            static const int desc_order = getenv("UVO_DESC_ORDER") ? atoi(getenv("UVO_DESC_ORDER")) : 2;            // measurement
            static const int nbig_env = getenv("UVO_DESC_NBIG") ? atoi(getenv("UVO_DESC_NBIG")) : 0;
            const int nbig_l = nbig_env > 0 ? nbig_env : nbig;
            hipLaunchKernelGGL(k_descriptor64, dim3((nbig_l + nsmall) * nim), dim3(256), lds_small > lds_big ? lds_small : lds_big, c->stream, lp, w, h, nbig_l, desc_part, nim, desc_order);
            if (dstamp && d_dstamps) {                        // (measurement: synchronous, every launch rewrites the file)
                std::vector<long long> hs(dstamps_n);
                (void)hipStreamSynchronize(c->stream);
                (void)hipMemcpy(hs.data(), d_dstamps, sizeof(long long) * dstamps_n, hipMemcpyDeviceToHost);
                if (FILE* f = fopen(dstamps_path, "w")) {
                    fprintf(f, "im,k,t_start,t_kp,t_ready,t_horiz,t_vert,t_end,win,kind\n");
                    for (int y = 0; y < 2; y++) for (int k = 0; k < 16384; k++) { const long long* r = hs.data() + ((size_t)y * 16384 + k) * 8; if (r[0]) fprintf(f, "%d,%d,%lld,%lld,%lld,%lld,%lld,%lld,%lld,%lld\n", y, k, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]); }
                    fclose(f);
                }
                if (FILE* f = fopen((std::string(dstamps_path) + ".big").c_str(), "w")) {
                    fprintf(f, "im,task,t_start,t_horiz,t_end,win,kind\n");
                    for (int y = 0; y < 2; y++) for (int k = 0; k < 16384; k++) { const long long* r = hs.data() + ((size_t)(2 + y) * 16384 + k) * 8; if (r[0] && r[2]) fprintf(f, "%d,%d,%lld,%lld,%lld,%lld,%lld\n", y, k, r[0], r[1], r[2], r[6], r[7]); }
                    fclose(f);
                }
            }
            hipLaunchKernelGGL(k_descriptor64_big_finish, dim3(1024, nim), dim3(256), 0, c->stream, lp);
        } else {
            // orientation assignment, then every descriptor from its rotated window (not the shipped configuration)
            const DescArgs& da = lp.a[0].da;
            const size_t lds_rot = sizeof(float) * 21 * (kMaxWin + 1);
            static bool rot_attr_dev[64] = {false};
            bool& rot_attr = rot_attr_dev[c->device & 63];
            if (!rot_attr) { UVO_HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_descriptor_rot), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rot)); rot_attr = true; }
            hipLaunchKernelGGL(k_surf_orientation, dim3((c->cap + 3) / 4, nimg), dim3(256), 0, c->stream, da, w, h);
            hipLaunchKernelGGL(k_descriptor_rot, dim3(2048, nimg), dim3(256), lds_rot, c->stream, da, w, h);
        }
        UVO_HIP_TRY(c, hipGetLastError());
    }
    return UVO_OK;
}
uvo_status surf_detect(Ctx* c, int nimg, int gate_min_features) { return surf_detect_lanes(c, nullptr, nimg, gate_min_features); }

uvo_status surf_hessian_layer_debug(Ctx* c, int octave, int layer, float* det, float* trace)
{
    const int w = c->img_w, h = c->img_h;
    if (octave < 0 || octave > 3 || layer < 0 || layer > 4 || w <= 0) { c->err = "hessian_layer: bad octave/layer or no image"; return UVO_INVALID_ARG; }
    OctavePat op;
    make_octave(octave, 3, w, h, &op);
    size_t n = (size_t)op.rows * op.cols;
    float *d_det = nullptr, *d_tr = nullptr;
    UVO_HIP_TRY(c, hipMalloc(&d_det, sizeof(float) * n * 2));
    d_tr = d_det + n;
    UVO_HIP_TRY(c, hipMemsetAsync(d_det, 0, sizeof(float) * n * 2, c->stream));
    const LayerPat& lp = op.L[layer];
    if (lp.samples_i > 0) {
        dim3 g((lp.samples_j + 255) / 256, lp.samples_i);
        hipLaunchKernelGGL(k_hessian_layer_debug, g, dim3(256), 0, c->stream, c->d_sum[0], w, h, lp, op.step, op.rows, op.cols, d_det, d_tr);
    }
    UVO_HIP_TRY(c, hipMemcpyAsync(det, d_det, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    UVO_HIP_TRY(c, hipMemcpyAsync(trace, d_tr, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    (void)hipFree(d_det);
    return UVO_OK;
}

}  // namespace uvo
