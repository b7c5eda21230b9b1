// pose.hip -- triangulation, extract_3Dpoints and EPnP PnP-RANSAC on gfx950 (replaces the calls at
// visual_odometry.h:631-648: cv::triangulatePoints, extract_3Dpoints (VO_utility.cpp:188-237),
// cv::solvePnPRansac(SOLVEPNP_EPNP)).
//
// These stages are tiny (<= a few thousand points) and latency-bound, not bandwidth-bound; what
// matters is that the sequential OpenCV semantics survive parallel execution unchanged:
//   * every fp64 solver runs one problem per thread (4x4 / 12x12 Jacobi SVD in LDS, [elem][lane]
//     interleaved) in the reference's operation order;
//   * RANSAC: the host replays cv::RNG to produce the exact 5-point subsets, ALL hypotheses are
//     solved (k_pnp_hyp) and scored (k_pnp_score) in parallel, then the host replays OpenCV's
//     sequential "better than best => shrink niters" scan over the counts to pick the identical
//     winner; the winner's mask is recomputed and the inlier refit runs block-cooperatively with
//     each floating-point sum kept in its sequential order.
#include <dlfcn.h>
#include "uvo_ctx.h"
#include <atomic>
#include <chrono>
#include "uvo_epnp.h"
#include "uvo_epnp_fast.h"
#include <string.h>
#include <stdlib.h>
#include <mutex>

namespace uvo {

// RANSACUpdateNumIters (ptsetreg.cpp); host libm, as the reference
int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters)
{
    p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : cv_round_d(num / denom);
}

struct Mat34 { double v[12]; };
struct Cam { double R[9], t[3], fx, fy, cx, cy; };

// ---------------------------------------------------------------- triangulatePoints
static const int kTriThreads = 64;
// extract_3Dpoints, per point (VOU:188-221): convertPointsFromHomogeneous (float) + mean reprojection error in both views + z > 0
__device__ __forceinline__ void extract3d_point(const float4 p, const uvo_point2f k1, const uvo_point2f k2, const Cam& c1, const Cam& c2, double tol,
                                                double* cam1, int* flag, int i)
{
    float scale = p.w != 0.f ? 1.f / p.w : 1.f;
    double X = (double)(p.x * scale), Y = (double)(p.y * scale), Z = (double)(p.z * scale);
    cam1[3*i] = X; cam1[3*i + 1] = Y; cam1[3*i + 2] = Z;
    double u, v;
    project_point(X, Y, Z, c1.R, c1.t, c1.fx, c1.fy, c1.cx, c1.cy, &u, &v);
    double dx = k1.x - u, dy = k1.y - v;
    double e1 = sqrt(dx * dx + dy * dy);
    project_point(X, Y, Z, c2.R, c2.t, c2.fx, c2.fy, c2.cx, c2.cy, &u, &v);
    dx = k2.x - u; dy = k2.y - v;
    double e2 = sqrt(dx * dx + dy * dy);
    double mean = (e1 + e2) / 2.0;
    flag[i] = ((mean < tol) && (Z > 0)) ? 1 : 0;
}

// FILTER: the loops call triangulatePoints and extract_3Dpoints back to back on the same point pairs (VO:631-632, VO:355-356);
// the per-point part of the latter then runs on the point just triangulated instead of in a launch of its own
// per-lane arguments of the two kernels below: blockIdx.y selects the lane in a two-pair launch (uvo_stereo_submit, batch mode)
struct TriLane { const uvo_point2f* x1; const uvo_point2f* x2; const int* n_p; float4* out; double* cam1; int* flag; };
struct TriLanes { TriLane l[2]; };
struct Ex3Lane { const double* cam1; const int* flag; const uvo_point2f* xc; const int* n_p; int* tmp_idx; double* good_pts; int* good_idx; float* opts;
                 uvo_point2f* ipts; int* counts /* [1] = G */; int* counts_host /* pinned mirror of all the step's counters, or null */;
                 // the stereo loop's tail (k_stereo_tail): point i is row map[i].queryIdx of cam1 / flag / pts4 (the previous pair's set, triangulated
                 // row by row in that pair's own tail) and its current-image point is keypoint map[i].trainIdx of cL; null: point i is row i
                 const uvo_dmatch* map; const uvo_keypoint* cL; const float4* pts4; float4* out_pts4; int* tmp_row; };
struct Ex3Lanes { Ex3Lane l[2]; };
// cv::triangulatePoints for one point pair, by thread threadIdx.x < TT of its workgroup (the 4x4 Jacobi SVD's operands live in
// LDS, one column of `lds` per thread)
static const int kTriLdsDoubles = (16 + 16 + 16 + 4 + 4) * kTriThreads;
template <int TT = kTriThreads>
__device__ __forceinline__ float4 triangulate_point(const Mat34& P1, const Mat34& P2, const uvo_point2f a, const uvo_point2f b, double* lds)
{
    using A = SArr<TT>;
    A Am{lds + threadIdx.x}, At = Am + 16, Vt = Am + 32, W = Am + 48, Wt = Am + 52;
    const double xa = a.x, ya = a.y, xb = b.x, yb = b.y;
    for (int k = 0; k < 4; k++) {
        Am[0*4 + k] = xa * P1.v[2*4 + k] - P1.v[0*4 + k];
        Am[1*4 + k] = ya * P1.v[2*4 + k] - P1.v[1*4 + k];
        Am[2*4 + k] = xb * P2.v[2*4 + k] - P2.v[0*4 + k];
        Am[3*4 + k] = yb * P2.v[2*4 + k] - P2.v[1*4 + k];
    }
    svd_square<4>(Am, At, W, Vt, Wt);
    return make_float4((float)Vt[12], (float)Vt[13], (float)Vt[14], (float)Vt[15]);
}
template <bool FILTER>
__global__ __launch_bounds__(kTriThreads) void k_triangulate(Mat34 P1, Mat34 P2, TriLanes lanes, int n_imm, Cam c1, Cam c2, double tol)
{
    const TriLane& LN = lanes.l[blockIdx.y];
    const uvo_point2f* x1 = LN.x1; const uvo_point2f* x2 = LN.x2; const int* n_p = LN.n_p; float4* out = LN.out; double* cam1 = LN.cam1; int* flag = LN.flag;
    const int n = n_p ? *n_p : n_imm;
    const int i = blockIdx.x * kTriThreads + threadIdx.x;
    __shared__ double lds[kTriLdsDoubles];
    if (i >= n) return;
    const float4 X = triangulate_point(P1, P2, x1[i], x2[i], lds);
    out[i] = X;
    if (FILTER) extract3d_point(X, x1[i], x2[i], c1, c2, tol, cam1, flag, i);
}

// The mono loop's triangulation (visual_odometry.h:355-356) under a second camera that a kernel chose: recoverPose's four candidates
// travel by value, the index of the winner is read from device memory (mono.hip: k_mono_pick), so the launch can be queued before the
// host knows which candidate won.
struct PoseChoice { Mat34 P2[4]; Cam c2[4]; };
__global__ __launch_bounds__(kTriThreads) void k_triangulate_pick(Mat34 P1, PoseChoice pc, const int* __restrict__ best_p, TriLane LN, Cam c1, double tol)
{
    const int n = *LN.n_p;
    const int i = blockIdx.x * kTriThreads + threadIdx.x;
    __shared__ double lds[kTriLdsDoubles];
    if (i >= n) return;
    const int b = *best_p & 3;
    const float4 X = triangulate_point(P1, pc.P2[b], LN.x1[i], LN.x2[i], lds);
    LN.out[i] = X;
    extract3d_point(X, LN.x1[i], LN.x2[i], c1, pc.c2[b], tol, LN.cam1, LN.flag, i);
}

// ---------------------------------------------------------------- extract_3Dpoints
// stage A on caller-provided homogeneous points (uvo_extract_3d_points)
__global__ __launch_bounds__(256) void k_extract3d_a(const float4* pts4, const uvo_point2f* k1, const uvo_point2f* k2,
                                                     Cam c1, Cam c2, double tol, const int* n_p, int n_imm,
                                                     double* cam1, int* flag)
{
    const int n = n_p ? *n_p : n_imm;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    extract3d_point(pts4[i], k1[i], k2[i], c1, c2, tol, cam1, flag, i);
}

// reproject_errors (VOU:632-651) on caller-provided points
__global__ __launch_bounds__(256) void k_reproject_errors(const double* world, const uvo_point2f* img, Cam cm, int n, double* err)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double u, v;
    project_point(world[3*i], world[3*i + 1], world[3*i + 2], cm.R, cm.t, cm.fx, cm.fy, cm.cx, cm.cy, &u, &v);
    double dx = img[i].x - u, dy = img[i].y - v;
    err[i] = sqrt(dx * dx + dy * dy);
}

// block-wide ordered compaction helper: returns the exclusive position of a kept element and
// advances *s_base (shared) by the number kept in this pass.  1024 threads.
__device__ __forceinline__ int block_compact_pos(bool keep, int* wtot, int* s_base)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned long long bal = __ballot(keep);
    int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wtot[wv] = __popcll(bal);
    __syncthreads();
    int off = *s_base;
    for (int k = 0; k < wv; k++) off += wtot[k];
    __syncthreads();
    if (tid == 0) { int t = 0; for (int k = 0; k < (int)(blockDim.x >> 6); k++) t += wtot[k]; *s_base += t; }
    __syncthreads();
    return off + before;
}

// stage B (one workgroup of 1024 threads): first compaction, mean/variance of z in index order, +-3 sigma filter,
// second compaction; also gathers the PnP inputs (float object points, current-image points).
struct Ex3Smem { int wtot[16]; int s_base; int s_need_seq; double s_mean, s_sd3, s_rad, s_sumsq; double red[3][16]; double zbuf[2048], zsq[2048]; };
__device__ __forceinline__ void extract3d_block(const Ex3Lane& LN, int n_imm, int min_pts, int force_seq /* always take the ordered sums (test hook) */, Ex3Smem& sm)
{
    const double* cam1 = LN.cam1; const int* flag = LN.flag; const uvo_point2f* xc = LN.xc; const int* n_p = LN.n_p; int* tmp_idx = LN.tmp_idx;
    double* good_pts = LN.good_pts; int* good_idx = LN.good_idx; float* opts = LN.opts; uvo_point2f* ipts = LN.ipts; int* counts = LN.counts; int* counts_host = LN.counts_host;
    const uvo_dmatch* map = LN.map; int* tmp_row = LN.tmp_row;
    const int n = n_p ? *n_p : n_imm;
    const int tid = threadIdx.x;
    // the last kernel of the loops' device stage: it leaves the counters where the host reads them (no copy to queue behind it)
    // (every thread calls it with the same G: one wave-wide store, i.e. one burst over PCIe instead of twenty single writes)
    auto publish = [&](int G) { if (tid == 0) counts[1] = G; if (counts_host && tid < CN_TOTAL) counts_host[tid] = tid == 1 ? G : counts[tid]; };
    int* wtot = sm.wtot;
    if (tid == 0) sm.s_base = 0;
    __syncthreads();
    const bool enough = n >= min_pts;                       // VOU:203
    for (int base = 0; base < n; base += 1024) {
        int i = base + tid;
        int row = i;
        if (map && i < n) { row = map[i].queryIdx; LN.out_pts4[i] = LN.pts4[row]; }      // the homogeneous points in match order, as triangulatePoints returns them
        bool keep = enough && i < n && flag[row] != 0;
        int pos = block_compact_pos(keep, wtot, &sm.s_base);
        if (keep) { tmp_idx[pos] = i; if (map) tmp_row[pos] = row; }
    }
    __syncthreads();
    const int ngood = sm.s_base;
    if (ngood < min_pts || ngood == 0) {                    // VOU:222
        publish(0);
        return;
    }
    const int* zrow = map ? tmp_row : tmp_idx;              // where the i-th kept point's coordinates are
    // MU:35-56 computes mean and variance from a sum and a sum of squares taken in index order; what leaves this kernel is only WHO
    // passes mean -+ 3 sigma.  Round 3: the two sums are first taken as workgroup tree sums (any order differs from the ordered sums
    // by at most ~2 n u sum|z|, u = 2^-53), the resulting thresholds get a rigorous error radius, and unless some z lies inside that
    // radius of a threshold -- then, or when the variance is not safely positive, the ordered chains below decide -- the set that
    // passes is the ordered sums' set.  (The ordered chains are ~1500 dependent fp64 additions on one lane: 17-24 us of this
    // single-workgroup kernel; UVO_EXTRACT3D_SEQ=1 / the `force_seq` argument always takes them.)
    {
        double ps = 0, pa = 0, pq = 0;
        for (int i = tid; i < ngood; i += 1024) { const double z = cam1[3 * zrow[i] + 2]; ps += z; pa += fabs(z); pq += z * z; }
        for (int o = 32; o > 0; o >>= 1) { ps += __shfl_down(ps, o); pa += __shfl_down(pa, o); pq += __shfl_down(pq, o); }
        if ((tid & 63) == 0) { sm.red[0][tid >> 6] = ps; sm.red[1][tid >> 6] = pa; sm.red[2][tid >> 6] = pq; }
        __syncthreads();
        if (tid == 0) {
            double S = 0, A = 0, Q = 0;
            for (int k = 0; k < 16; k++) { S += sm.red[0][k]; A += sm.red[1][k]; Q += sm.red[2][k]; }
            const double n = (double)ngood, u = 1.1102230246251565e-16;
            const double mean = S / n, variance = Q / n - mean * mean;
            const double sd3 = 3.0 * sqrt(variance);
            // error radii of the ordered-vs-tree difference, generously rounded up (factor 8 in place of the 2 of the textbook bound)
            const double dS = 8 * n * u * A, dQ = 8 * n * u * Q;
            const double dmean = dS / n + 4 * u * fabs(mean);
            const double dvar = dQ / n + 2 * fabs(mean) * dmean + 8 * u * (Q / n + mean * mean);
            bool safe = force_seq == 0 && variance > 0 && variance > 64 * dvar;         // also false for NaN
            double rad = 0;
            if (safe) {
                const double dsd3 = 3.0 * dvar / sqrt(variance) + 4 * u * sd3;          // d sqrt(v) <= dv / (2 sqrt(v - dv)), rounded up
                rad = 4 * (dmean + dsd3 + 4 * u * (fabs(mean) + sd3));
            }
            sm.s_mean = mean; sm.s_sd3 = sd3; sm.s_rad = rad; sm.s_need_seq = safe ? 0 : 1;
        }
        __syncthreads();
        if (!sm.s_need_seq) {
            const double hi = sm.s_mean + sm.s_sd3, lo = sm.s_mean - sm.s_sd3, rad = sm.s_rad;
            bool near = false;
            for (int i = tid; i < ngood; i += 1024) { const double z = cam1[3 * zrow[i] + 2]; near = near || fabs(z - hi) <= rad || fabs(z - lo) <= rad; }
            if (near) sm.s_need_seq = 1;                     // (benign race: every writer stores 1, a late reader skips a scan that could only store 1)
        }
        __syncthreads();
    }
    if (sm.s_need_seq)
    {   // MU:35-56: sum and sum of squares in index order.  z and z*z are staged through LDS in chunks; the two sequential chains run
        // on two different waves (one lane each), so each issues one fp64 add per element instead of sharing a SIMD's issue slots
        double* zbuf = sm.zbuf; double* zsq = sm.zsq;
        double acc = 0.0;                                   // lane 0: the sum; lane 64: the sum of squares
        for (int base = 0; base < ngood; base += 2048) {
            const int cnt = min(2048, ngood - base);
            for (int i = tid; i < cnt; i += 1024) { const double z = cam1[3 * zrow[base + i] + 2]; zbuf[i] = z; zsq[i] = z * z; }
            __syncthreads();
            if (tid == 0 || tid == 64) {
                // the additions are one dependent chain; the LDS reads of the next sixteen operands are issued before the current
                // sixteen are added (a batch that waits for its own reads costs ~23 cycles per element instead of ~9)
                const double* src = tid == 0 ? zbuf : zsq;
                acc = seq_sum_pipelined(acc, cnt, [&](int i) { return src[i]; });
            }
            __syncthreads();
        }
        if (tid == 64) sm.s_sumsq = acc;
        __syncthreads();
        if (tid == 0) {
            const double sum = acc, sumsq = sm.s_sumsq;
            double mean = sum / ngood;
            double variance = (sumsq / ngood) - (mean * mean);
            sm.s_mean = mean; sm.s_sd3 = 3.0 * sqrt(variance);
        }
    }
    __syncthreads();
    if (tid == 0) sm.s_base = 0;
    __syncthreads();
    const double mean = sm.s_mean, sd3 = sm.s_sd3;
    for (int base = 0; base < ngood; base += 1024) {
        int i = base + tid;
        bool keep = false; int src = 0, row = 0; double z = 0;
        if (i < ngood) { src = tmp_idx[i]; row = zrow[i]; z = cam1[3*row + 2]; keep = (z <= mean + sd3) && (z >= mean - sd3); }
        int pos = block_compact_pos(keep, wtot, &sm.s_base);
        if (keep) {
            double X = cam1[3*row], Y = cam1[3*row + 1];
            good_idx[pos] = src;
            good_pts[3*pos] = X; good_pts[3*pos + 1] = Y; good_pts[3*pos + 2] = z;
            opts[3*pos] = (float)X; opts[3*pos + 1] = (float)Y; opts[3*pos + 2] = (float)z;   // solvePnPRansac: opoints -> CV_32F
            if (map) { const uvo_keypoint k = LN.cL[map[src].trainIdx]; ipts[pos] = uvo_point2f{k.x, k.y}; }
            else if (xc) ipts[pos] = xc[src];
        }
    }
    __syncthreads();
    publish(sm.s_base);
}
__global__ __launch_bounds__(1024) void k_extract3d_b(Ex3Lanes lanes, int n_imm, int min_pts, int force_seq)
{
    __shared__ Ex3Smem sm;
    extract3d_block(lanes.l[blockIdx.y], n_imm, min_pts, force_seq, sm);
}

// ---------------------------------------------------------------- the tail of the stereo loop's stage A in ONE launch (round 4)
// VO:569-579 (this pair's "after stereo match" set), VO:631 (triangulatePoints) and VO:632 (extract_3Dpoints) were three launches in a
// row -- gather 4.7 us, triangulation 21.5 us (one dependent fp64 chain per point), extract_3Dpoints 16.4 us (one workgroup) -- and a
// stage A's length decides the pipeline's cadence almost one to one (tools/probe/gpu.sh sens).  Triangulation and the per-point half of
// extract_3Dpoints are functions of ONE row of the previous pair's set (its left and right keypoint and the rig), not of the triangular
// match that selects the row: every pair therefore triangulates the rows of its OWN set here -- for its successor -- while workgroup 0
// runs extract_3Dpoints on the rows of the previous pair's set that this pair's triangular matches select.  The three roles share
// nothing inside the launch (the triangulation reads the keypoints through the stereo matches, not the set being gathered).
struct TailArgs {
    const uvo_dmatch* m_s; const int* cn; const uvo_keypoint* kL; const uvo_keypoint* kR; const float* dL;       // stereo matches, the detector's lists
    uvo_keypoint* okL; uvo_keypoint* okR; float* odL; int dim;                                                    // -> this pair's set
    float4* as_pts4; double* as_cam1; int* as_flag;                                                               // -> its rows, triangulated
    int tri_blocks, p4_blocks, gather_blocks, fused;
    float4* out_pts4;                                                                                             // triangulatePoints' output, match order (fused form)
    Ex3Lane ex;
};
// The launch's workgroups have to find room NEXT TO another pair's chip-filling kernels (the detection launch leaves 4 KB of a CU's
// LDS and most of its registers, the descriptor launch a wave slot and 64 registers per SIMD as its workgroups retire), so they are
// small where it was measured to matter: 256 threads, ~80 registers.  As 1024-thread workgroups with 33 KB the same launch took 61 us
// in the pipeline against 22 us alone.
static const int kTailThreads = 256;
// rows triangulated per workgroup (448 bytes of LDS each for the 4x4 Jacobi SVD's operands).  Measured, launch alone / pipeline:
// 8 rows (3.6 KB, fits beside three detection workgroups): 31.6 us / 4661 pairs/s; 16: 26.0 / 4720; 32: 26.0 / 4731; 64: 23.4 / 4760.
#ifndef UVO_TAIL_TRI
#define UVO_TAIL_TRI 64
#endif
static const int kTailTri = UVO_TAIL_TRI;
static const int kEx3Passes = 32;         // extract3d_rows addresses kEx3Passes x kTailThreads = 8192 triangular matches; larger contexts: k_extract3d_b
// extract_3Dpoints of the stereo loop on ONE small workgroup (round 4).  Thread t owns matches t, t + 256, ..; a match's state between
// the sweeps is one bit (it passed the reprojection test; it passed +-3 sigma), everything else is re-read: three sweeps in chunks of
// eight (four) passes whose loads are all in flight together -- ~14 memory round trips for 3000 matches, where the any-size kernel
// (extract3d_block, 1024 threads, both compactions staged through memory) makes two per pass.  The first compaction is never
// materialised: only its count and the statistics of its members are needed, unless the ordered sums have to decide.
// Same results: the +-3 sigma set is the ordered sums' set by the argument in extract3d_block (any summation order is within the radius).
// (16-byte aligned: a pass's four wave counts are read as one int4)
struct alignas(16) Ex3RowsSmem { int cnt[kEx3Passes][kTailThreads / 64]; int s_need_seq; double s_mean, s_sd3, s_rad, s_sumsq; double red[3][kTailThreads / 64]; double zbuf[128], zsq[128]; };
__device__ __forceinline__ void extract3d_rows(const Ex3Lane& LN, int min_pts, int force_seq, Ex3RowsSmem& sm)
{
    constexpr int NT = kTailThreads, NW = NT / 64, R = kEx3Passes, C = 8;
    static_assert(NW == 4, "one int4 of wave counts per pass");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int n = min(*LN.n_p, R * NT);                      // (the host takes this path only when the context's capacity fits)
    const int passes = (n + NT - 1) / NT;
    const uvo_dmatch* __restrict__ map = LN.map; const double* __restrict__ cam1 = LN.cam1; const int* __restrict__ flag = LN.flag;
    int* counts = LN.counts; int* counts_host = LN.counts_host;
    auto publish = [&](int G) { if (tid == 0) counts[1] = G; if (counts_host && tid < CN_TOTAL) counts_host[tid] = tid == 1 ? G : counts[tid]; };
    // per (pass, wave) counts of the matches whose bit is set in `mask` (bit p: match p * NT + tid)
    auto fill_table = [&](unsigned mask) {
        __syncthreads();                                     // the table's previous use is over
#pragma unroll
        for (int p = 0; p < R; p++) { const unsigned long long bal = __ballot((mask >> p) & 1u); if (lane == 0) sm.cnt[p][wv] = __popcll(bal); }
        __syncthreads();
    };
    // position of this thread's match of pass p in the ordered compaction, given the matches before pass p (`run`, advanced)
    auto position = [&](int p, bool bit, int& run) {
        const int4 c = *reinterpret_cast<const int4*>(sm.cnt[p]);
        const int pos = run + (wv > 0 ? c.x : 0) + (wv > 1 ? c.y : 0) + (wv > 2 ? c.z : 0) + __popcll(__ballot(bit) & below);
        run += c.x + c.y + c.z + c.w;
        return pos;
    };
    // depths of the matches in `mask`, eight passes in flight: f(p, z)
    auto for_depths = [&](unsigned mask, auto f) {
        for (int c0 = 0; c0 < passes; c0 += C) {
            int row[C]; double z[C];
#pragma unroll
            for (int q = 0; q < C; q++) { row[q] = 0; if ((mask >> (c0 + q)) & 1u) row[q] = map[(c0 + q) * NT + tid].queryIdx; }
#pragma unroll
            for (int q = 0; q < C; q++) { z[q] = 0; if ((mask >> (c0 + q)) & 1u) z[q] = cam1[3 * row[q] + 2]; }
#pragma unroll
            for (int q = 0; q < C; q++) if ((mask >> (c0 + q)) & 1u) f(c0 + q, z[q]);
        }
    };
    const bool enough = n >= min_pts;                        // VOU:203
    // sweep 1: who passed the reprojection test (VOU:188-221's flag, set where the row was triangulated), their count and tree sums
    unsigned keep = 0;
    double ps = 0, pa = 0, pq = 0;
    for (int c0 = 0; c0 < passes; c0 += C) {
        int row[C], fl[C]; double z[C];
#pragma unroll
        for (int q = 0; q < C; q++) { const int i = (c0 + q) * NT + tid; row[q] = -1; if (i < n) row[q] = map[i].queryIdx; }
#pragma unroll
        for (int q = 0; q < C; q++) { fl[q] = 0; z[q] = 0; if (row[q] >= 0) { fl[q] = flag[row[q]]; z[q] = cam1[3 * row[q] + 2]; } }
#pragma unroll
        for (int q = 0; q < C; q++) if (enough && fl[q] != 0) { keep |= 1u << (c0 + q); ps += z[q]; pa += fabs(z[q]); pq += z[q] * z[q]; }
    }
    int pc = __popc(keep);
    for (int o = 32; o > 0; o >>= 1) { ps += __shfl_down(ps, o); pa += __shfl_down(pa, o); pq += __shfl_down(pq, o); pc += __shfl_down(pc, o); }
    if (lane == 0) { sm.red[0][wv] = ps; sm.red[1][wv] = pa; sm.red[2][wv] = pq; sm.cnt[0][wv] = pc; }
    __syncthreads();
    int ngood = 0;
    for (int k = 0; k < NW; k++) ngood += sm.cnt[0][k];
    if (ngood < min_pts || ngood == 0) {                     // VOU:222
        publish(0);
        return;
    }
    if (tid == 0) {
        double S = 0, A = 0, Q = 0;
        for (int k = 0; k < NW; k++) { S += sm.red[0][k]; A += sm.red[1][k]; Q += sm.red[2][k]; }
        const double nn = (double)ngood, u = 1.1102230246251565e-16;
        const double mean = S / nn, variance = Q / nn - mean * mean;
        const double sd3 = 3.0 * sqrt(variance);
        const double dS = 8 * nn * u * A, dQ = 8 * nn * u * Q;                        // as in extract3d_block
        const double dmean = dS / nn + 4 * u * fabs(mean);
        const double dvar = dQ / nn + 2 * fabs(mean) * dmean + 8 * u * (Q / nn + mean * mean);
        const bool safe = force_seq == 0 && variance > 0 && variance > 64 * dvar;     // also false for NaN
        double rad = 0;
        if (safe) {
            const double dsd3 = 3.0 * dvar / sqrt(variance) + 4 * u * sd3;
            rad = 4 * (dmean + dsd3 + 4 * u * (fabs(mean) + sd3));
        }
        sm.s_mean = mean; sm.s_sd3 = sd3; sm.s_rad = rad; sm.s_need_seq = safe ? 0 : 1;
    }
    __syncthreads();
    // sweep 2: the +-3 sigma set under the tree sums' thresholds, and whether any depth is too close to one of them to trust it
    unsigned keep2 = 0;
    {
        const double mean = sm.s_mean, sd3 = sm.s_sd3, hi = mean + sd3, lo = mean - sd3, rad = sm.s_rad;
        bool near = false;
        for_depths(keep, [&](int p, double z) { near = near || fabs(z - hi) <= rad || fabs(z - lo) <= rad; if (z <= hi && z >= lo) keep2 |= 1u << p; });
        __syncthreads();                                     // everyone has read thread 0's verdict
        if (near) sm.s_need_seq = 1;
        __syncthreads();
    }
    if (sm.s_need_seq) {
        // MU:35-56: the sum and the sum of squares in index order decide.  The members' depths go to scratch (the output buffer, not
        // yet written) in compaction order and through LDS in chunks; the two chains run on one lane of two different waves.
        double* zs = LN.good_pts;
        fill_table(keep);
        {
            int run = 0;
            for (int p = 0; p < passes; p++) {               // (pass by pass: this path is rare)
                const bool bit = (keep >> p) & 1u;
                const int pos = position(p, bit, run);
                if (bit) zs[pos] = cam1[3 * map[p * NT + tid].queryIdx + 2];
            }
        }
        __syncthreads();
        double acc = 0.0;                                    // thread 0: the sum; thread 64: the sum of squares
        for (int base = 0; base < ngood; base += 128) {
            const int cnt = min(128, ngood - base);
            if (tid < cnt) { const double v = zs[base + tid]; sm.zbuf[tid] = v; sm.zsq[tid] = v * v; }
            __syncthreads();
            if (tid == 0 || tid == 64) {
                const double* src = tid == 0 ? sm.zbuf : sm.zsq;
                acc = seq_sum_pipelined(acc, cnt, [&](int i) { return src[i]; });
            }
            __syncthreads();
        }
        if (tid == 64) sm.s_sumsq = acc;
        __syncthreads();
        if (tid == 0) {
            const double mean = acc / ngood;
            const double variance = (sm.s_sumsq / ngood) - (mean * mean);
            sm.s_mean = mean; sm.s_sd3 = 3.0 * sqrt(variance);
        }
        __syncthreads();
        const double mean = sm.s_mean, sd3 = sm.s_sd3;
        keep2 = 0;
        for_depths(keep, [&](int p, double z) { if ((z <= mean + sd3) && (z >= mean - sd3)) keep2 |= 1u << p; });
    }
    // sweep 3: the second compaction and the PnP inputs, four passes in flight
    fill_table(keep2);
    int run = 0;
    for (int c0 = 0; c0 < passes; c0 += 4) {
        int row[4], train[4]; double X[4], Y[4], Z[4]; float kx[4], ky[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            row[q] = 0; train[q] = 0;
            if ((keep2 >> (c0 + q)) & 1u) { const int2 m = *reinterpret_cast<const int2*>(map + (c0 + q) * NT + tid); row[q] = m.x; train[q] = m.y; }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            X[q] = Y[q] = Z[q] = 0; kx[q] = ky[q] = 0;
            if ((keep2 >> (c0 + q)) & 1u) {
                X[q] = cam1[3 * row[q]]; Y[q] = cam1[3 * row[q] + 1]; Z[q] = cam1[3 * row[q] + 2];
                const uvo_keypoint* k = LN.cL + train[q]; kx[q] = k->x; ky[q] = k->y;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int p = c0 + q;
            if (p >= passes) break;
            const bool bit = (keep2 >> p) & 1u;
            const int pos = position(p, bit, run);
            if (bit) {
                LN.good_idx[pos] = p * NT + tid;
                LN.good_pts[3*pos] = X[q]; LN.good_pts[3*pos + 1] = Y[q]; LN.good_pts[3*pos + 2] = Z[q];
                LN.opts[3*pos] = (float)X[q]; LN.opts[3*pos + 1] = (float)Y[q]; LN.opts[3*pos + 2] = (float)Z[q];     // solvePnPRansac: opoints -> CV_32F
                LN.ipts[pos] = uvo_point2f{kx[q], ky[q]};
            }
        }
    }
    __syncthreads();
    publish(run);
}
__device__ __forceinline__ void tail_gather_rows(const TailArgs& a, int first_row)        // 16 rows per workgroup, sixteen lanes per descriptor row
{
    const int meff = a.cn[CN_MEFF];
    const int row = first_row + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    if (row >= meff) return;
    const int q = a.m_s[row].queryIdx, t = a.m_s[row].trainIdx;
    for (int v = sub; v < a.dim / 4; v += 16) reinterpret_cast<float4*>(a.odL + (size_t)row * a.dim)[v] = reinterpret_cast<const float4*>(a.dL + (size_t)q * a.dim)[v];
    if (sub == 0) { a.okL[row] = a.kL[q]; a.okR[row] = a.kR[t]; }
}
// workgroup 0 (when a.fused): extract_3Dpoints; then tri_blocks workgroups of kTriThreads rows to triangulate, p4_blocks that leave
// triangulatePoints' output in match order for uvo_stereo_get, and the gather of the set
union alignas(16) TailSmem { Ex3RowsSmem e; double tri[(16 + 16 + 16 + 4 + 4) * kTailTri]; };
__global__ __launch_bounds__(kTailThreads) void k_stereo_tail(Mat34 P1, Mat34 P2, Cam c1, Cam c2, double tol, TailArgs a, int min_pts, int force_seq)
{
    __shared__ TailSmem sm;
    int b = blockIdx.x;
    if (a.fused) { if (b == 0) { extract3d_rows(a.ex, min_pts, force_seq, sm.e); return; } b--; }
    // (every role walks its rows with the stride of its share of the grid: the host sizes the shares by the last known keypoint
    // count -- a grid for max_kpts rows is 1500 workgroups of which 400 find work -- and any size gives the same result)
    if (b < a.tri_blocks) {
        if (threadIdx.x >= kTailTri) return;
        const int meff = a.cn[CN_MEFF];
        for (int row = b * kTailTri + threadIdx.x; row < meff; row += a.tri_blocks * kTailTri) {
            const uvo_dmatch m = a.m_s[row];
            const uvo_keypoint l = a.kL[m.queryIdx], r = a.kR[m.trainIdx];
            const uvo_point2f x1 = {l.x, l.y}, x2 = {r.x, r.y};
            const float4 X = triangulate_point<kTailTri>(P1, P2, x1, x2, sm.tri);
            a.as_pts4[row] = X;
            extract3d_point(X, x1, x2, c1, c2, tol, a.as_cam1, a.as_flag, row);
        }
        return;
    }
    b -= a.tri_blocks;
    if (b < a.p4_blocks) {
        const int T = *a.ex.n_p;
        for (int i = b * kTailThreads + threadIdx.x; i < T; i += a.p4_blocks * kTailThreads) a.out_pts4[i] = a.ex.pts4[a.ex.map[i].queryIdx];
        return;
    }
    b -= a.p4_blocks;
    const int meff = a.cn[CN_MEFF];
    for (int first = b * (kTailThreads / 16); first < meff; first += a.gather_blocks * (kTailThreads / 16)) tail_gather_rows(a, first);
}
// the same per-row triangulation for a set that was gathered by other means (the init phase's first set; the second pair of a two-pair launch)
__global__ __launch_bounds__(kTriThreads) void k_as_triangulate(Mat34 P1, Mat34 P2, Cam c1, Cam c2, double tol, const uvo_keypoint* aL, const uvo_keypoint* aR,
                                                                const int* n_p, int n_imm, float4* as_pts4, double* as_cam1, int* as_flag)
{
    __shared__ double lds[kTriLdsDoubles];
    const int n = n_p ? *n_p : n_imm;
    const int row = blockIdx.x * kTriThreads + threadIdx.x;
    if (row >= n) return;
    const uvo_keypoint l = aL[row], r = aR[row];
    const uvo_point2f x1 = {l.x, l.y}, x2 = {r.x, r.y};
    const float4 X = triangulate_point(P1, P2, x1, x2, lds);
    as_pts4[row] = X;
    extract3d_point(X, x1, x2, c1, c2, tol, as_cam1, as_flag, row);
}

// ---------------------------------------------------------------- PnP RANSAC
static const int kHypGroups = 8;         // hypotheses per workgroup (one wave): 8 lanes cooperate on each
static const int kHypPerGroup = EPNP_SMALL + 15 + 10 + 20 + 45 + 15;   // doubles of LDS per hypothesis

__device__ long long g_hyp_clk[16];         // diagnostic: phase stamps of hypothesis 0 (printed when UVO_DBG_PHASE is set)
// The PnP stage of up to kMaxPnpBatch pairs (pipeline lanes) shares every launch: blockIdx.y (or .x for the one-workgroup
// kernels) selects the job.  The kernels are latency-bound and far too small to fill the chip, and the device runs
// only a few kernels at a time, so batching the pairs that are ready divides the stage's share of that budget.
static const int kMaxPnpBatch = 8;
struct PnpJob {
    const float* opts; const uvo_point2f* ipts; const int* subsets; double* models; int* hcount;
    int* inliers; double* ws; int* countsB; double* pose;
    int* ninl_host;                          // pinned host mirrors (hcount, pose and ninl_host are written straight into host memory: no copies to queue)
    int G, nhyp, best;
    int n_best, seq;                         // inliers of the winning hypothesis; refit by the sequential (OpenCV-ordered) kernel
};
struct PnpBatch { int n, cap; double fx, fy, cx, cy; float thr2; int dbg; PnpJob job[kMaxPnpBatch]; };

__global__ __launch_bounds__(64) void k_pnp_hyp(PnpBatch b)
{
    const PnpJob& jb = b.job[blockIdx.y];
    const float* opts = jb.opts; const uvo_point2f* ipts = jb.ipts; const int* subsets = jb.subsets; double* models = jb.models;
    const int nhyp = jb.nhyp;
    const double fx = b.fx, fy = b.fy, cx = b.cx, cy = b.cy;
    if ((int)blockIdx.x * kHypGroups >= nhyp) return;
    extern __shared__ __align__(16) unsigned char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int group = threadIdx.x >> 3, lane = threadIdx.x & 7;
    const int hyp_raw = blockIdx.x * kHypGroups + group;
    const int hyp = hyp_raw < nhyp ? hyp_raw : nhyp - 1;      // surplus groups redo the last one (they must reach every barrier)
    using P = GroupPolicy<kHypGroups>;
    using A = P::Arr;
    A base{lds + group};
    Epnp<P> e;
    e.uc = cx; e.vc = cy; e.fu = fx; e.fv = fy; e.n = 5;
    e.clk = (blockIdx.x == 0 && blockIdx.y == 0 && group == 0) ? g_hyp_clk : nullptr;
    e.s = base; e.pws = base + EPNP_SMALL; e.us = e.pws + 15; e.alphas = e.us + 10; e.pcs = e.alphas + 20; e.tmp = e.pcs + 45;
    const double ifx = 1. / fx, ify = 1. / fy;
    if (lane < 5) {
        const int i = lane, id = subsets[hyp * 5 + i];
        e.pws[3*i] = opts[3*id]; e.pws[3*i + 1] = opts[3*id + 1]; e.pws[3*i + 2] = opts[3*id + 2];
        // undistortPoints with zero distortion, stored CV_32FC2, then epnp::init_points
        double x = (double)(float)((ipts[id].x - cx) * ifx), y = (double)(float)((ipts[id].y - cy) * ify);
        e.us[2*i] = x * fx + cx; e.us[2*i + 1] = y * fy + cy;
    }
    __syncthreads();
    double rvec[3], tvec[3];
    e.compute_pose(rvec, tvec);
    if (lane == 0 && hyp_raw < nhyp) {
        double* m = models + (size_t)hyp * 6;
        m[0] = rvec[0]; m[1] = rvec[1]; m[2] = rvec[2]; m[3] = tvec[0]; m[4] = tvec[1]; m[5] = tvec[2];
    }
}

// PnPRansacCallback::computeError + findInliers for one model: projectPoints (double, stored float),
// err = |ipt - proj|^2 in float, inlier iff err <= (float)(thr*thr)
__device__ __forceinline__ bool pnp_is_inlier(const float* opts, const uvo_point2f* ipts, int i, const double* R, const double* t,
                                              double fx, double fy, double cx, double cy, float thr2)
{
    double u, v;
    project_point((double)opts[3*i], (double)opts[3*i + 1], (double)opts[3*i + 2], R, t, fx, fy, cx, cy, &u, &v);
    float dx = ipts[i].x - (float)u, dy = ipts[i].y - (float)v;
    float err = dx * dx + dy * dy;
    return err <= thr2;
}

__global__ __launch_bounds__(256) void k_pnp_score(PnpBatch b)
{
    const PnpJob& jb = b.job[blockIdx.y];
    const float* opts = jb.opts; const uvo_point2f* ipts = jb.ipts; const double* models = jb.models; int* hcount = jb.hcount;
    const int n = jb.G;
    const double fx = b.fx, fy = b.fy, cx = b.cx, cy = b.cy; const float thr2 = b.thr2;
    const int hyp = blockIdx.x, tid = threadIdx.x;
    if (hyp >= jb.nhyp) return;
    __shared__ double sR[9], st[3];
    __shared__ int s_cnt;
    if (tid == 0) {
        const double* m = models + (size_t)hyp * 6;
        double R[9]; rodrigues_vec2mat(m, R);
        for (int k = 0; k < 9; k++) sR[k] = R[k];
        st[0] = m[3]; st[1] = m[4]; st[2] = m[5];
        s_cnt = 0;
    }
    __syncthreads();
    double R[9], t[3];
    for (int k = 0; k < 9; k++) R[k] = sR[k];
    t[0] = st[0]; t[1] = st[1]; t[2] = st[2];
    int cnt = 0;
    for (int i = tid; i < n; i += 256) cnt += pnp_is_inlier(opts, ipts, i, R, t, fx, fy, cx, cy, thr2) ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if ((tid & 63) == 0) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    if (tid == 0) hcount[hyp] = s_cnt;
}

// winner's mask -> ascending inlier list + refit inputs (double points; undistort in double)
__global__ __launch_bounds__(1024) void k_pnp_mask(PnpBatch b)
{
    const PnpJob& jb = b.job[blockIdx.x];
    const float* opts = jb.opts; const uvo_point2f* ipts = jb.ipts; const int n = jb.G;
    const double* model = jb.models + (size_t)jb.best * 6;
    const double fx = b.fx, fy = b.fy, cx = b.cx, cy = b.cy; const float thr2 = b.thr2;
    int* inliers = jb.inliers; double* pws = jb.ws; double* us = jb.ws + 3 * (size_t)b.cap; int* counts = jb.countsB;   // [0] = n_inliers
    const int tid = threadIdx.x;
    __shared__ int wtot[16];
    __shared__ int s_base;
    double R[9], t[3];
    rodrigues_vec2mat(model, R);
    t[0] = model[3]; t[1] = model[4]; t[2] = model[5];
    if (tid == 0) s_base = 0;
    __syncthreads();
    const double ifx = 1. / fx, ify = 1. / fy;
    for (int base = 0; base < n; base += 1024) {
        int i = base + tid;
        bool keep = i < n && pnp_is_inlier(opts, ipts, i, R, t, fx, fy, cx, cy, thr2);
        int pos = block_compact_pos(keep, wtot, &s_base);
        if (keep) {
            inliers[pos] = i;
            pws[3*pos] = opts[3*i]; pws[3*pos + 1] = opts[3*i + 1]; pws[3*pos + 2] = opts[3*i + 2];
            double x = ((double)ipts[i].x - cx) * ifx, y = ((double)ipts[i].y - cy) * ify;     // undistortPoints, CV_64FC2
            us[2*pos] = x * fx + cx; us[2*pos + 1] = y * fy + cy;
        }
    }
    __syncthreads();
    if (tid == 0) counts[0] = s_base;
}

__device__ long long g_refit_clk[16];
__device__ double g_dbg_small[2][EPNP_SMALL];   // UVO_DBG_REFIT: the fixed-size state of the sequential [0] and the parallel [1] refit of the same inliers       // diagnostic: phase stamps of the last refit (printed when UVO_DBG_PHASE is set)
// inlier refit: one workgroup, block-cooperative EPnP.  ws: pws 3c | us 2c | alphas 4c | pcs 9c | tmp 3c (c = cap)
__global__ __launch_bounds__(256) void k_pnp_refit(PnpBatch b)
{
    const PnpJob& jb = b.job[blockIdx.x];
    double* ws = jb.ws; const int cap = b.cap; double* pose = jb.pose;
    const double fx = b.fx, fy = b.fy, cx = b.cx, cy = b.cy;
    __shared__ double small[EPNP_SMALL];
    extern __shared__ __align__(16) unsigned char refit_smem[];
    double* stage_buf = reinterpret_cast<double*>(refit_smem);       // Epnp<BlockPolicy>::kStageDoubles doubles (dynamic: > 64 KB)
    const int n = jb.countsB[0];
    if (!jb.seq && !b.dbg) return;                                   // k_pnp_refit_fast's job
    using P = BlockPolicy;
    Epnp<P> e;
    e.uc = cx; e.vc = cy; e.fu = fx; e.fv = fy; e.n = n;
    e.s = P::Arr{small}; e.stage = stage_buf; e.clk = blockIdx.x == 0 ? g_refit_clk : nullptr;
    e.pws = P::Arr{ws}; e.us = P::Arr{ws + 3 * (size_t)cap}; e.alphas = P::Arr{ws + 5 * (size_t)cap};
    e.pcs = P::Arr{ws + 9 * (size_t)cap}; e.tmp = P::Arr{ws + 18 * (size_t)cap};
    double rvec[3], tvec[3];
    e.compute_pose(rvec, tvec);
    if (b.dbg) { __syncthreads(); for (int i = threadIdx.x; i < EPNP_SMALL; i += blockDim.x) g_dbg_small[0][i] = small[i]; if (!jb.seq) return; }
    if (threadIdx.x == 0) { pose[0] = rvec[0]; pose[1] = rvec[1]; pose[2] = rvec[2]; pose[3] = tvec[0]; pose[4] = tvec[1]; pose[5] = tvec[2]; jb.ninl_host[0] = n; }
}

// inlier refit, the usual case (>= kFastRefitMin inliers): the workgroup-parallel solver of uvo_epnp_fast.h.  The pose is the
// refit's only output and its contract is 1e-4 relative, so the long sums are tree reductions, M^T M runs on the fp64
// matrix pipe and the small decompositions use fast rotations; with fewer inliers it leaves the job to k_pnp_refit.
__global__ __launch_bounds__(kFastThreads) void k_pnp_refit_fast(PnpBatch b)
{
    const PnpJob& jb = b.job[blockIdx.x];
    __shared__ double lds[kFastLdsDoubles];
    const int n = jb.countsB[0];
    if (jb.seq) return;
    EpnpFast e;
    e.uc = b.cx; e.vc = b.cy; e.fu = b.fx; e.fv = b.fy; e.n = n; e.cap = b.cap; e.ws = jb.ws; e.lds = lds;
    e.clk = blockIdx.x == 0 ? g_refit_clk : nullptr;
    double rvec[3], tvec[3];
    e.compute_pose(rvec, tvec);
    if (b.dbg) { __syncthreads(); for (int i = threadIdx.x; i < EPNP_SMALL; i += blockDim.x) g_dbg_small[1][i] = lds[i]; }
    double* pose = jb.pose;
    if (threadIdx.x == 0) { pose[0] = rvec[0]; pose[1] = rvec[1]; pose[2] = rvec[2]; pose[3] = tvec[0]; pose[4] = tvec[1]; pose[5] = tvec[2]; jb.ninl_host[0] = n; }
}

// ---------------------------------------------------------------- the first RANSAC round without the host (round 3)
// The stereo loop used to hand a pair from its device stage to the lane's worker thread for PnP: wake on an event, draw the
// subsets, launch hypotheses + scoring, wait, replay OpenCV's sequential scan, launch mask + refit, wait -- two host round
// trips and a thread wake-up (~60 us of a 0.8 ms pair).  With a working odometry the scan is over inside the first 64
// hypotheses, so that round now runs SPECULATIVELY on the lane's own stream, queued right behind extract_3Dpoints:
//   k_pnp_hyp_spec    draws its own subsets -- cv::RNG((uint64)-1) is a constant stream, so its raw 32-bit outputs are a table and
//                     only `% G` and the duplicate test depend on the pair -- and solves the 64 EPnP-5 hypotheses
//   k_pnp_score_spec  counts inliers per hypothesis; the last workgroup to finish replays the scan on the device
//   k_pnp_refit_spec  winner's mask, ordered compaction and the workgroup-parallel refit (the bodies of k_pnp_mask and
//                     k_pnp_refit_fast)
// The worker then wakes once, replays the scan over the same counts with the host's libm (log / pow decide the adaptive
// iteration count and must be the reference's) and accepts the device's pose only if the host arrives at the same winner
// with the scan complete; otherwise -- more hypotheses needed, too few inliers for the fast refit, G == 5, a walk that
// ran out of table -- the whole stage runs again the old way.  Results are therefore exactly those of the host-driven path.
static const int kSpecHyp = 64;             // hypotheses of the speculative round (= the host path's first round)
static const int kRngRaw = 1024;            // raw outputs of cv::RNG((uint64)-1) kept on the device (64 subsets need 320 + redraws)
struct PnpSpecState { int ticket, state, best, n_best, niters, nhyp, G, pad; };      // state: 0 not run, 1 ran but the host must redo, 2 pose delivered
struct PnpSpecArgs {
    const int* cn; int min3d, iters; double conf;
    const float* opts; const uvo_point2f* ipts; const unsigned* rng_raw;
    int* subsets; double* models; int* hcount_dev; int* hcount_host;
    int* inliers; double* ws; int* countsB; double* pose; int* ninl_host;
    PnpSpecState* st; PnpSpecState* st_host;
    int cap; double fx, fy, cx, cy; float thr2;
};
// the hypotheses the speculative round evaluates for G points: 0 when the pair does not take the path at all
__device__ __forceinline__ int spec_nhyp(const PnpSpecArgs& a, int G) { return (G > a.min3d && G >= 6) ? (a.iters < kSpecHyp ? a.iters : kSpecHyp) : 0; }

__global__ __launch_bounds__(64) void k_pnp_hyp_spec(PnpSpecArgs a)
{
    const int G = a.cn[CN_G];
    const int nhyp = spec_nhyp(a, G);
    // state 1 = "the models are there", unless some workgroup raises `pad` (its walk ran out of table); both are reset per pair:
    // state here, pad by k_pnp_refit_spec, the ticket by the scan
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.st->G = G; a.st->nhyp = nhyp; a.st->state = nhyp > 0 ? 1 : 0; }
    if ((int)blockIdx.x * kHypGroups >= nhyp) return;
    extern __shared__ __align__(16) unsigned char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    __shared__ int s_sub[kHypGroups][5];
    __shared__ int s_ok;
    const int group = threadIdx.x >> 3, lane = threadIdx.x & 7;
    const int hyp_raw = blockIdx.x * kHypGroups + group;
    const int hyp = hyp_raw < nhyp ? hyp_raw : nhyp - 1;
    if (threadIdx.x == 0) s_ok = 1;
    // getSubset (ptsetreg.cpp): for every iteration five draws uniform(0, G), a draw equal to an earlier one of the same subset
    // is redrawn.  The draws are r_j % G of the constant raw stream; hypothesis h starts at draw 5 h + (redraws before it).  Lane h
    // walks hypothesis h from its assumed start, a wave scan of the redraw counts gives the next assumption, until nothing moves:
    // the starts become final from the front, one pass when no subset of the round has a duplicate (a sequential walk by one lane
    // with a dependent load per draw measured ~100 us on the critical path).
    __shared__ int s_m[kRngRaw];
    for (int j = threadIdx.x; j < kRngRaw; j += 64) s_m[j] = (int)(a.rng_raw[j] % (unsigned)G);
    __syncthreads();
    {
        const int h = threadIdx.x;                        // one lane per hypothesis of the round (64 = the wave)
        int D = 0, cur[5], ok = 1;
        for (int pass = 0; pass < kSpecHyp + 1; pass++) {
            int pos = 5 * h + D, used = 0;
            ok = 1;
            for (int t = 0; t < 5;) {
                if (pos + used >= kRngRaw) { ok = 0; break; }
                const int idx = s_m[pos + used++];
                bool dup = false;
                for (int u = 0; u < t; u++) dup = dup || cur[u] == idx;
                if (!dup) cur[t++] = idx;
            }
            const int extra = ok ? used - 5 : 0;
            int incl = extra;                              // inclusive scan over the wave
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (h >= o) incl += v; }
            const int newD = incl - extra;
            const bool changed = newD != D;
            D = newD;
            if (!__any(changed)) break;
        }
        if (h < nhyp && !ok) s_ok = 0;                    // (s_ok was set below before the walk)
        const int g = h - (int)blockIdx.x * kHypGroups;
        if (g >= 0 && g < kHypGroups) for (int t = 0; t < 5; t++) s_sub[g][t] = cur[t];
    }
    __syncthreads();
    if (!s_ok) { if (threadIdx.x == 0) a.st->pad = 1; return; }       // the walk ran out of table: the host redoes the stage
    using P = GroupPolicy<kHypGroups>;
    using A = P::Arr;
    A base{lds + group};
    Epnp<P> e;
    e.uc = a.cx; e.vc = a.cy; e.fu = a.fx; e.fv = a.fy; e.n = 5;
    e.clk = nullptr;
    e.s = base; e.pws = base + EPNP_SMALL; e.us = e.pws + 15; e.alphas = e.us + 10; e.pcs = e.alphas + 20; e.tmp = e.pcs + 45;
    const double ifx = 1. / a.fx, ify = 1. / a.fy;
    const int gsub = hyp - (int)blockIdx.x * kHypGroups;
    if (lane < 5) {
        const int i = lane, id = s_sub[gsub][i];
        e.pws[3*i] = a.opts[3*id]; e.pws[3*i + 1] = a.opts[3*id + 1]; e.pws[3*i + 2] = a.opts[3*id + 2];
        double x = (double)(float)((a.ipts[id].x - a.cx) * ifx), y = (double)(float)((a.ipts[id].y - a.cy) * ify);
        e.us[2*i] = x * a.fx + a.cx; e.us[2*i + 1] = y * a.fy + a.cy;
        if (hyp_raw < nhyp) a.subsets[hyp * 5 + i] = id;
    }
    __syncthreads();
    double rvec[3], tvec[3];
    e.compute_pose(rvec, tvec);
    if (lane == 0 && hyp_raw < nhyp) {
        double* m = a.models + (size_t)hyp * 6;
        m[0] = rvec[0]; m[1] = rvec[1]; m[2] = rvec[2]; m[3] = tvec[0]; m[4] = tvec[1]; m[5] = tvec[2];
    }
}

// RANSACUpdateNumIters on the device: ocml's log / pow instead of the host's libm -- which is why the host repeats the scan
__device__ __forceinline__ int ransac_update_num_iters_dev(double p, double ep, int modelPoints, int maxIters)
{
    p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, (double)modelPoints);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : cv_round_d(num / denom);
}

__global__ __launch_bounds__(256) void k_pnp_score_spec(PnpSpecArgs a)
{
    const int tid = threadIdx.x, hyp = blockIdx.x;
    const int G = a.st->G, nhyp = a.st->nhyp;
    if (a.st->state != 1 || a.st->pad != 0 || hyp >= nhyp) return;
    __shared__ double sR[9], st[3];
    __shared__ int s_cnt;
    if (tid == 0) {
        const double* m = a.models + (size_t)hyp * 6;
        double R[9]; rodrigues_vec2mat(m, R);
        for (int k = 0; k < 9; k++) sR[k] = R[k];
        st[0] = m[3]; st[1] = m[4]; st[2] = m[5];
        s_cnt = 0;
    }
    __syncthreads();
    double R[9], t[3];
    for (int k = 0; k < 9; k++) R[k] = sR[k];
    t[0] = st[0]; t[1] = st[1]; t[2] = st[2];
    int cnt = 0;
    for (int i = tid; i < G; i += 256) cnt += pnp_is_inlier(a.opts, a.ipts, i, R, t, a.fx, a.fy, a.cx, a.cy, a.thr2) ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if ((tid & 63) == 0) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    if (tid == 0) { a.hcount_host[hyp] = s_cnt; a.hcount_dev[hyp] = s_cnt; }
}

__global__ __launch_bounds__(kFastThreads) void k_pnp_refit_spec(PnpSpecArgs a)
{
    __shared__ double lds[kFastLdsDoubles];
    __shared__ int wtot[16];
    __shared__ int s_base;
    const int tid = threadIdx.x;
    __shared__ int s_state;
    if (tid == 0) {
        // RANSACPointSetRegistrator::run's sequential scan over the round's counts (the scoring kernel has finished)
        int state = a.st->pad != 0 ? 0 : a.st->state;
        if (state == 1) {
            const int modelPoints = 5, G = a.st->G, nhyp = a.st->nhyp;
            int niters = a.iters > 1 ? a.iters : 1, maxGood = 0, best = -1, iter = 0;
            for (; iter < niters && iter < nhyp; iter++) {
                const int good = a.hcount_dev[iter];
                if (good > (maxGood > modelPoints - 1 ? maxGood : modelPoints - 1)) {
                    best = iter; maxGood = good;
                    niters = ransac_update_num_iters_dev(a.conf, (double)(G - good) / G, modelPoints, niters);
                }
            }
            a.st->best = best; a.st->n_best = maxGood; a.st->niters = niters;
            if (iter >= niters && best >= 0 && maxGood >= kFastRefitMin) state = 2;      // the refit goes ahead
        }
        s_state = state;
    }
    __syncthreads();
    const int state = s_state;
    if (state != 2) {                                       // nothing to refit on the device: tell the host (one store over PCIe)
        if (tid == 0) { PnpSpecState h = *a.st; h.state = state == 0 ? 0 : 1; h.pad = 0; *a.st_host = h; a.st->pad = 0; }
        return;
    }
    const int n = a.st->G;
    const double* model = a.models + (size_t)a.st->best * 6;
    double R[9], t[3];
    rodrigues_vec2mat(model, R);
    t[0] = model[3]; t[1] = model[4]; t[2] = model[5];
    if (tid == 0) s_base = 0;
    __syncthreads();
    double* pws = a.ws; double* us = a.ws + 3 * (size_t)a.cap;
    const double ifx = 1. / a.fx, ify = 1. / a.fy;
    for (int base = 0; base < n; base += kFastThreads) {   // k_pnp_mask's body: ascending inlier list + refit inputs
        const int i = base + tid;
        const bool keep = i < n && pnp_is_inlier(a.opts, a.ipts, i, R, t, a.fx, a.fy, a.cx, a.cy, a.thr2);
        const int pos = block_compact_pos(keep, wtot, &s_base);
        if (keep) {
            a.inliers[pos] = i;
            pws[3*pos] = a.opts[3*i]; pws[3*pos + 1] = a.opts[3*i + 1]; pws[3*pos + 2] = a.opts[3*i + 2];
            double x = ((double)a.ipts[i].x - a.cx) * ifx, y = ((double)a.ipts[i].y - a.cy) * ify;     // undistortPoints, CV_64FC2
            us[2*pos] = x * a.fx + a.cx; us[2*pos + 1] = y * a.fy + a.cy;
        }
    }
    __syncthreads();
    const int ninl = s_base;
    if (tid == 0) a.countsB[0] = ninl;
    __syncthreads();
    EpnpFast e;
    e.uc = a.cx; e.vc = a.cy; e.fu = a.fx; e.fv = a.fy; e.n = ninl; e.cap = a.cap; e.ws = a.ws; e.lds = lds;
    e.clk = nullptr;
    double rvec[3], tvec[3];
    e.compute_pose(rvec, tvec);
    if (tid == 0) {
        a.pose[0] = rvec[0]; a.pose[1] = rvec[1]; a.pose[2] = rvec[2]; a.pose[3] = tvec[0]; a.pose[4] = tvec[1]; a.pose[5] = tvec[2];
        a.ninl_host[0] = ninl;
        __threadfence_system();
        PnpSpecState h = *a.st; h.state = 2; *a.st_host = h;
    }
}

// single 5-point solve when npoints == model_points (solvePnPRansac short-cut): reuse k_pnp_hyp with
// the identity subset.

// ---------------------------------------------------------------- host orchestration
static Cam make_cam(const double* R, const double* t, const double* K);
static int extract3d_force_seq() { const char* e = getenv("UVO_EXTRACT3D_SEQ"); return e && atoi(e) != 0; }      // read per call: tests flip it
static TriLane tri_lane(Ctx* c, const int* d_n, bool filter) { return TriLane{ c->d_x1, c->d_x2, d_n, c->d_pts4, filter ? c->d_cam1 : nullptr, filter ? c->d_flag : nullptr }; }
static Ex3Lane ex3_lane(Ctx* c, int slot, const int* d_n, int* counts_host)
{
    return Ex3Lane{ c->d_cam1, c->d_flag, c->d_xc, d_n, c->d_tmp_idx, c->d_good_pts[slot], c->d_good_idx[slot], c->d_opts[slot], c->d_ipts[slot], c->d_counts, counts_host };
}
uvo_status pose_triangulate(Ctx* c, const double* P1, const double* P2, const int* d_n, int n_max)
{
    if (n_max <= 0) return UVO_OK;
    Mat34 a, b; memcpy(a.v, P1, sizeof(a.v)); memcpy(b.v, P2, sizeof(b.v));
    StageTimer t(c, ST_TRIANGULATE);
    TriLanes tl; tl.l[0] = tl.l[1] = tri_lane(c, d_n, false);
    hipLaunchKernelGGL(k_triangulate<false>, dim3((n_max + kTriThreads - 1) / kTriThreads), dim3(kTriThreads), 0, c->stream, a, b, tl, n_max, Cam(), Cam(), 0.0);
    UVO_HIP_TRY(c, hipGetLastError());
    return UVO_OK;
}
// triangulatePoints + extract_3Dpoints on the same point pairs (d_x1, d_x2), as both loops call them: two launches instead of three.
// c2 != nullptr: the same for a second lane's pair in the same two launches (its count is c2's CN_T, its mirror counts_host2); the
// kernels go to c's stream.
uvo_status pose_triangulate_extract3d(Ctx* c, int slot, const double* P1, const double* P2, const double* R1, const double* t1,
                                      const double* R2, const double* t2, const double* K1, const double* K2, const int* d_n, int n_max,
                                      int* counts_host, Ctx* c2, int* counts_host2)
{
    const int nl = c2 ? 2 : 1;
    if (n_max > 0) {
        Mat34 a, b; memcpy(a.v, P1, sizeof(a.v)); memcpy(b.v, P2, sizeof(b.v));
        StageTimer t(c, ST_TRIANGULATE);
        TriLanes tl; tl.l[0] = tri_lane(c, d_n, true); tl.l[1] = c2 ? tri_lane(c2, c2->d_counts + CN_T, true) : tl.l[0];
        hipLaunchKernelGGL(k_triangulate<true>, dim3((n_max + kTriThreads - 1) / kTriThreads, nl), dim3(kTriThreads), 0, c->stream,
                           a, b, tl, n_max, make_cam(R1, t1, K1), make_cam(R2, t2, K2), c->p.REPROJECTION_TOLERANCE);
    }
    StageTimer t(c, ST_EXTRACT3D);
    Ex3Lanes el; el.l[0] = ex3_lane(c, slot, d_n, counts_host); el.l[1] = c2 ? ex3_lane(c2, slot, c2->d_counts + CN_T, counts_host2) : el.l[0];
    hipLaunchKernelGGL(k_extract3d_b, dim3(1, nl), dim3(1024), 0, c->stream, el, n_max, c->p.MIN_NUM_3DPOINTS, extract3d_force_seq());
    UVO_HIP_TRY(c, hipGetLastError());
    return UVO_OK;
}

// triangulatePoints + extract_3Dpoints on the n (device count, at most n_max) point pairs d_in1 / d_in2 with the first camera K [I | 0] and the
// second one candidate *d_best of four (P2x4: 4 x 12, Rx4: 4 x 9, tx4: 4 x 3); outputs as pose_triangulate_extract3d (slot 0)
uvo_status pose_triangulate_extract3d_pick(Ctx* c, const double* P1, const double* P2x4, const double* Rx4, const double* tx4, const double* K,
                                           const int* d_best, const uvo_point2f* d_in1, const uvo_point2f* d_in2, const int* d_n, int n_max)
{
    const double I[9] = {1,0,0,0,1,0,0,0,1}, z[3] = {0,0,0};
    if (n_max > 0) {
        Mat34 a; memcpy(a.v, P1, sizeof(a.v));
        PoseChoice pc;
        for (int k = 0; k < 4; k++) { memcpy(pc.P2[k].v, P2x4 + 12 * k, sizeof(pc.P2[k].v)); pc.c2[k] = make_cam(Rx4 + 9 * k, tx4 + 3 * k, K); }
        StageTimer t(c, ST_TRIANGULATE);
        const TriLane ln{ d_in1, d_in2, d_n, c->d_pts4, c->d_cam1, c->d_flag };
        hipLaunchKernelGGL(k_triangulate_pick, dim3((n_max + kTriThreads - 1) / kTriThreads), dim3(kTriThreads), 0, c->stream, a, pc, d_best, ln, make_cam(I, z, K), c->p.REPROJECTION_TOLERANCE);
    }
    StageTimer t(c, ST_EXTRACT3D);
    Ex3Lane e = ex3_lane(c, 0, d_n, nullptr);
    e.xc = d_in2;
    Ex3Lanes el; el.l[0] = el.l[1] = e;
    hipLaunchKernelGGL(k_extract3d_b, dim3(1, 1), dim3(1024), 0, c->stream, el, n_max, c->p.MIN_NUM_3DPOINTS, extract3d_force_seq());
    UVO_HIP_TRY(c, hipGetLastError());
    return UVO_OK;
}

uvo_status pose_stereo_tail(Ctx* a, Ctx* p, int prev, int curr, int slot, const double* P1, const double* P2, const double* R1, const double* t1,
                            const double* R2, const double* t2, const double* K1, const double* K2, int* counts_host)
{
    const int cap = a->cap;
    Mat34 m1, m2; memcpy(m1.v, P1, sizeof(m1.v)); memcpy(m2.v, P2, sizeof(m2.v));
    int* cn = a->d_counts;
    TailArgs ta;
    ta.m_s = a->d_matches[0]; ta.cn = cn; ta.kL = a->det[0].kps; ta.kR = a->det[1].kps; ta.dL = a->det[0].desc;
    ta.okL = a->d_as_kpsL[curr]; ta.okR = a->d_as_kpsR[curr]; ta.odL = a->d_as_descL[curr]; ta.dim = a->desc_dim();
    ta.as_pts4 = a->d_as_pts4[curr]; ta.as_cam1 = a->d_as_cam1[curr]; ta.as_flag = a->d_as_flag[curr];
    // rows to expect: the keypoints of the last pair whose counts reached the host, plus a quarter (max_kpts before that)
    const Ctx* hm = a->master ? a->master : a;
    int rows = hm->kp_hint > 0 ? hm->kp_hint + hm->kp_hint / 4 : cap;
    rows = rows < 256 ? 256 : (rows > cap ? cap : rows);
    ta.tri_blocks = (rows + kTailTri - 1) / kTailTri;
    ta.gather_blocks = (rows + kTailThreads / 16 - 1) / (kTailThreads / 16);
    ta.ex = Ex3Lane{ p->d_as_cam1[prev], p->d_as_flag[prev], nullptr, cn + CN_T, a->d_tmp_idx, a->d_good_pts[slot], a->d_good_idx[slot], a->d_opts[slot], a->d_ipts[slot],
                     cn, counts_host, a->d_matches[1], a->det[0].kps, p->d_as_pts4[prev], a->d_pts4, a->d_tmp_row };
    // contexts of up to 8192 keypoints: extract_3Dpoints is workgroup 0 of the same launch; larger ones: its any-size kernel first
    static const bool split_env = getenv("UVO_TAIL_SPLIT") != nullptr;      // measurement: extract_3Dpoints in its own launch
    ta.fused = (cap <= kEx3Passes * kTailThreads && !split_env) ? 1 : 0;
    if (ta.fused) { ta.out_pts4 = a->d_pts4; ta.ex.out_pts4 = nullptr; ta.p4_blocks = (rows + kTailThreads - 1) / kTailThreads; }
    else {
        ta.out_pts4 = nullptr; ta.p4_blocks = 0;
        StageTimer t(a, ST_EXTRACT3D);
        Ex3Lanes el; el.l[0] = el.l[1] = ta.ex;
        hipLaunchKernelGGL(k_extract3d_b, dim3(1, 1), dim3(1024), 0, a->stream, el, cap, a->p.MIN_NUM_3DPOINTS, extract3d_force_seq());
    }
    StageTimer t(a, ST_TRIANGULATE);
    hipLaunchKernelGGL(k_stereo_tail, dim3(ta.fused + ta.tri_blocks + ta.p4_blocks + ta.gather_blocks), dim3(kTailThreads), 0, a->stream, m1, m2,
                       make_cam(R1, t1, K1), make_cam(R2, t2, K2), a->p.REPROJECTION_TOLERANCE, ta, a->p.MIN_NUM_3DPOINTS, extract3d_force_seq());
    UVO_HIP_TRY(a, hipGetLastError());
    return UVO_OK;
}
uvo_status pose_as_triangulate(Ctx* c, hipStream_t st, int buf, const int* d_n, int n_max, const double* P1, const double* P2, const double* R1, const double* t1,
                               const double* R2, const double* t2, const double* K1, const double* K2)
{
    if (n_max <= 0) return UVO_OK;
    Mat34 m1, m2; memcpy(m1.v, P1, sizeof(m1.v)); memcpy(m2.v, P2, sizeof(m2.v));
    hipLaunchKernelGGL(k_as_triangulate, dim3((n_max + kTriThreads - 1) / kTriThreads), dim3(kTriThreads), 0, st, m1, m2, make_cam(R1, t1, K1), make_cam(R2, t2, K2),
                       c->p.REPROJECTION_TOLERANCE, c->d_as_kpsL[buf], c->d_as_kpsR[buf], d_n, n_max, c->d_as_pts4[buf], c->d_as_cam1[buf], c->d_as_flag[buf]);
    UVO_HIP_TRY(c, hipGetLastError());
    return UVO_OK;
}

static Cam make_cam(const double* R, const double* t, const double* K)
{
    Cam cm; memcpy(cm.R, R, sizeof(cm.R)); memcpy(cm.t, t, sizeof(cm.t));
    cm.fx = K[0]; cm.fy = K[4]; cm.cx = K[2]; cm.cy = K[5];
    return cm;
}

uvo_status pose_extract3d(Ctx* c, int slot, const double* R1, const double* t1, const double* R2, const double* t2,
                          const double* K1, const double* K2, const int* d_n, int n_max)
{
    int* counts_host = nullptr;
    StageTimer t(c, ST_EXTRACT3D);
    if (n_max > 0) {
        hipLaunchKernelGGL(k_extract3d_a, dim3((n_max + 255) / 256), dim3(256), 0, c->stream, c->d_pts4, c->d_x1, c->d_x2,
                           make_cam(R1, t1, K1), make_cam(R2, t2, K2), c->p.REPROJECTION_TOLERANCE, d_n, n_max, c->d_cam1, c->d_flag);
    }
    Ex3Lanes el; el.l[0] = el.l[1] = ex3_lane(c, slot, d_n, counts_host);
    hipLaunchKernelGGL(k_extract3d_b, dim3(1), dim3(1024), 0, c->stream, el, n_max, c->p.MIN_NUM_3DPOINTS, extract3d_force_seq());
    UVO_HIP_TRY(c, hipGetLastError());
    return UVO_OK;
}

uvo_status pose_reproject_errors(Ctx* c, const double* world, int n, const double* R, const double* t, const double* K,
                                 const uvo_point2f* img, double* err)
{
    // staging: d_cam1 (cap x 3 f64) for the points, d_x1 for the pixels, d_good_pts[0] for the result
    UVO_HIP_TRY(c, hipMemcpyAsync(c->d_cam1, world, sizeof(double) * 3 * n, hipMemcpyHostToDevice, c->stream));
    UVO_HIP_TRY(c, hipMemcpyAsync(c->d_x1, img, sizeof(uvo_point2f) * n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_reproject_errors, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_cam1, c->d_x1, make_cam(R, t, K), n,
                       c->d_good_pts[0]);
    UVO_HIP_TRY(c, hipGetLastError());
    UVO_HIP_TRY(c, hipMemcpyAsync(err, c->d_good_pts[0], sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return UVO_OK;
}

// UVO_DBG_BSTAGE=1: host wall time of the PnP stage's segments, summed over calls (printed by uvo_ctx_destroy)
bool g_bdbg = getenv("UVO_DBG_BSTAGE") != nullptr;
std::atomic<double> g_bstat[16];
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr; int (*pop)() = nullptr;
    Roctx()
    {
        const char* e = getenv("UVO_ROCTX");
        if (!e || !*e || *e == '0') return;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { fprintf(stderr, "uvo: UVO_ROCTX is set but no roctx library could be loaded\n"); return; }
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
const Roctx& roctx() { static Roctx r; return r; }
}
void range_push(const char* name) { const Roctx& r = roctx(); if (r.push) r.push(name); }
void range_pop() { const Roctx& r = roctx(); if (r.pop) r.pop(); }
double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void operator+=(std::atomic<double>& a, double v) { double o = a.load(); while (!a.compare_exchange_weak(o, o + v)) {} }

// the speculative first round of a lane's pair, queued on the lane's own stream behind extract_3Dpoints (see k_pnp_hyp_spec)
static PnpSpecArgs spec_args(Ctx* c, const double* K, int iters, float reprojectionError, double confidence, int min3d)
{
    PnpSpecArgs a;
    memset(&a, 0, sizeof(a));
    a.cn = c->d_counts; a.min3d = min3d; a.iters = iters > 1 ? iters : 1; a.conf = confidence;
    a.opts = c->d_opts[0]; a.ipts = c->d_ipts[0]; a.rng_raw = c->d_rng_raw;
    a.subsets = c->d_subsets; a.models = c->d_models; a.hcount_dev = c->d_hcount; a.hcount_host = c->h_hcount;
    a.inliers = c->d_inliers; a.ws = c->d_refit; a.countsB = c->d_countsB; a.pose = c->h_pose; a.ninl_host = c->h_countsB;
    a.st = static_cast<PnpSpecState*>(c->d_spec); a.st_host = static_cast<PnpSpecState*>(c->h_spec);
    a.cap = c->cap; a.fx = K[0]; a.fy = K[4]; a.cx = K[2]; a.cy = K[5];
    const double threshold = reprojectionError;
    a.thr2 = (float)(threshold * threshold);
    return a;
}
uvo_status pose_pnp_spec_launch(Ctx* c, hipStream_t st, const double* K, int iters, float reprojectionError, double confidence, int min3d)
{
    static_cast<PnpSpecState*>(c->h_spec)->state = -1;     // "not reported yet": also when nothing is queued below, so that the accept
                                                           // step never reads the previous pair's round
    if (iters > kMaxHyp) return UVO_OK;                    // the host path reports it
    const size_t hyp_lds = sizeof(double) * GroupPolicy<kHypGroups>::kStride * kHypPerGroup;
    static std::once_flag attr_once[64];
    std::call_once(attr_once[c->device & 63], [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pnp_hyp_spec), hipFuncAttributeMaxDynamicSharedMemorySize, (int)hyp_lds); });
    const PnpSpecArgs a = spec_args(c, K, iters, reprojectionError, confidence, min3d);
    Ctx::TraceRec* tr = (c->trace_on && c->trace_cur >= 0) ? &c->trace[c->trace_cur] : nullptr;
    if (tr) { tr->b_used = true; (void)hipEventRecord(tr->ev[3], st); }
    hipLaunchKernelGGL(k_pnp_hyp_spec, dim3(kSpecHyp / kHypGroups), dim3(64), hyp_lds, st, a);
    hipLaunchKernelGGL(k_pnp_score_spec, dim3(kSpecHyp), dim3(256), 0, st, a);
    if (tr) (void)hipEventRecord(tr->ev[4], st);
    hipLaunchKernelGGL(k_pnp_refit_spec, dim3(1), dim3(kFastThreads), 0, st, a);
    if (tr) (void)hipEventRecord(tr->ev[5], st);
    UVO_HIP_TRY(c, hipGetLastError());
    return UVO_OK;
}
// After the lane's stream has drained: did the speculative round deliver, and does the host's own replay of the scan (its libm)
// agree?  true: *r holds the stage's result, exactly what pose_pnp_ransac_batch would return.
bool pose_pnp_spec_accept(Ctx* c, int G, int iters, double confidence, PnpResult* r)
{
    const PnpSpecState* h = static_cast<const PnpSpecState*>(c->h_spec);
    if (h->state != 2 || h->G != G) return false;
    const int modelPoints = 5, nhyp = h->nhyp;
    int niters = iters > 1 ? iters : 1, maxGood = 0, best = -1, iter = 0;
    for (; iter < niters && iter < nhyp; iter++) {
        const int good = c->h_hcount[iter];
        if (good > (maxGood > modelPoints - 1 ? maxGood : modelPoints - 1)) {
            best = iter; maxGood = good;
            niters = ransac_update_num_iters(confidence, (double)(G - good) / G, modelPoints, niters);
        }
    }
    if (iter < niters || best < 0 || best != h->best || maxGood != h->n_best || c->h_countsB[0] != maxGood) return false;
    r->st = UVO_OK; r->wrote = 1; r->ok = 1; r->ninl = maxGood;
    memcpy(r->rvec, c->h_pose, sizeof(double) * 3); memcpy(r->tvec, c->h_pose + 3, sizeof(double) * 3);
    return true;
}

// solvePnPRansac for n jobs at once: job i works on the G[i] points already in lanes[i]->d_opts[0] / d_ipts[0] with
// lanes[i]'s PnP buffers; every launch and both host syncs are shared.  Runs on m->pnp_stream.
uvo_status pose_pnp_ransac_batch(Ctx* m, int n, Ctx* const* lanes, const int* G, const double* K, int iterationsCount,
                                 float reprojectionError, double confidence, PnpResult* res)
{
    const int modelPoints = 5;
    hipStream_t st = m->pnp_stream;
    double t_b0 = g_bdbg ? now_us() : 0;
    if (n < 1 || n > kMaxPnpBatch) { m->err = "pnp batch size"; return UVO_INVALID_ARG; }
    const size_t hyp_lds = sizeof(double) * GroupPolicy<kHypGroups>::kStride * kHypPerGroup;
    static std::once_flag attr_once[64];                                               // once per device
    std::call_once(attr_once[m->device & 63], [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pnp_hyp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)hyp_lds); });
    int niters0 = iterationsCount > 1 ? iterationsCount : 1;
    PnpBatch b;
    memset(&b, 0, sizeof(b));
    b.cap = m->cap; b.fx = K[0]; b.fy = K[4]; b.cx = K[2]; b.cy = K[5];
    const double threshold = reprojectionError;
    b.thr2 = (float)(threshold * threshold);
    // Hypotheses are evaluated in rounds: the first kFirstHyp, then -- only if RANSAC's adaptive iteration count still
    // reaches past them after the replayed scan -- all the rest.  The scan visits the same counts in the same order either
    // way, so the chosen model is the one a single launch of all ITERATIONS_COUNT hypotheses would give; with the
    // inlier ratios of a working odometry the count drops to a handful after the first good hypothesis and one round of
    // 8 workgroups replaces 63 (whose 54 KB of LDS each would otherwise sit beside other pairs' stage-A kernels).
    struct Scan { int job, G, niters, maxGood, best, last, iter, computed; uint64_t rng; bool active; };
    Scan sc[kMaxPnpBatch];
    PnpJob base[kMaxPnpBatch];                // full-range pointers of each active job
    int ns = 0;
    for (int i = 0; i < n; i++) {
        PnpResult& r = res[i];
        r.st = UVO_OK; r.wrote = r.ok = r.ninl = 0;
        Ctx* c = lanes[i];
        if (G[i] < 4) { c->err = "solvePnPRansac needs at least 4 points (OpenCV asserts)"; r.st = UVO_TOO_FEW_POINTS; continue; }
        if (G[i] == 4) { c->err = "solvePnPRansac with exactly 4 points takes OpenCV's P3P path, which the reference never reaches; not implemented"; r.st = UVO_TOO_FEW_POINTS; continue; }
        if (niters0 > kMaxHyp) { c->err = "iterations_count exceeds the compiled hypothesis capacity (2048)"; r.st = UVO_CAPACITY; continue; }
        // subsets are read from, and the per-hypothesis counts, the pose and the inlier count written to, pinned host memory
        // by the kernels themselves: the queue holds launches only (each small copy costs ~10 us of stage-B latency)
        PnpJob& j = base[ns];
        j.opts = c->d_opts[0]; j.ipts = c->d_ipts[0]; j.subsets = c->h_subsets; j.models = c->d_models; j.hcount = c->h_hcount;
        j.inliers = c->d_inliers; j.ws = c->d_refit; j.countsB = c->d_countsB; j.pose = c->h_pose; j.ninl_host = c->h_countsB;
        j.G = G[i]; j.nhyp = 0; j.best = 0; j.n_best = 0; j.seq = 0;
        Scan& q = sc[ns++];
        q.job = i; q.G = G[i]; q.niters = G[i] == modelPoints ? 1 : niters0; q.maxGood = 0; q.best = -1; q.last = 0; q.iter = 0; q.computed = 0;
        q.rng = (uint64_t)-1;                 // getSubset (ptsetreg.cpp): cv::RNG((uint64)-1)
        q.active = true;
    }
    if (ns == 0) return UVO_OK;
    Ctx::TraceRec* tr = (n == 1 && lanes[0]->trace_on && lanes[0]->trace_cur >= 0) ? &lanes[0]->trace[lanes[0]->trace_cur] : nullptr;   // UVO_TRACE
    if (tr) { tr->b_used = true; (void)hipEventRecord(tr->ev[3], st); (void)hipEventRecord(tr->ev[4], st); (void)hipEventRecord(tr->ev[5], st); }
    const int kFirstHyp = 64;
    for (int round = 0; ; round++) {
        int slot_of[kMaxPnpBatch], nb = 0, max_hyp = 0;
        for (int k = 0; k < ns; k++) {
            Scan& q = sc[k];
            if (!q.active) continue;
            Ctx* c = lanes[q.job];
            const int count = round == 0 ? (q.niters < kFirstHyp ? q.niters : kFirstHyp) : q.niters - q.computed;
            if (q.G == modelPoints) { for (int t = 0; t < 5; t++) c->h_subsets[t] = t; }
            else {
                // getSubset: uniform(0, count), redraw while duplicate; the generator runs on from round to round
                for (int it = q.computed; it < q.computed + count; it++) {
                    int* sub = c->h_subsets + it * 5;
                    for (int t = 0; t < modelPoints; t++) {
                        int idx_k;
                        for (;;) {
                            idx_k = (int)(rng_next(q.rng) % (uint32_t)q.G);
                            bool dup = false;
                            for (int u = 0; u < t; u++) dup = dup || sub[u] == idx_k;
                            if (!dup) break;
                        }
                        sub[t] = idx_k;
                    }
                }
            }
            PnpJob& j = b.job[nb];
            j = base[k];
            j.subsets = base[k].subsets + (size_t)q.computed * 5; j.models = base[k].models + (size_t)q.computed * 6; j.hcount = base[k].hcount + q.computed;
            j.nhyp = count;
            if (count > max_hyp) max_hyp = count;
            slot_of[nb++] = k;
        }
        if (nb == 0) break;
        b.n = nb;
        {
            StageTimer t(m, ST_PNP_HYP, st);
            hipLaunchKernelGGL(k_pnp_hyp, dim3((max_hyp + kHypGroups - 1) / kHypGroups, nb), dim3(64), hyp_lds, st, b);
            UVO_HIP_TRY(m, hipGetLastError());
        }
        bool any_scored = false;
        for (int s_ = 0; s_ < nb; s_++) any_scored = any_scored || sc[slot_of[s_]].G != modelPoints;
        if (any_scored) {
            StageTimer t(m, ST_PNP_SCORE, st);
            hipLaunchKernelGGL(k_pnp_score, dim3(max_hyp, nb), dim3(256), 0, st, b);
            UVO_HIP_TRY(m, hipGetLastError());
        }
        for (int s_ = 0; s_ < nb; s_++) {
            Ctx* c = lanes[sc[slot_of[s_]].job];
            if (sc[slot_of[s_]].G == modelPoints) UVO_HIP_TRY(m, hipMemcpyAsync(c->h_pose, c->d_models, sizeof(double) * 6, hipMemcpyDeviceToHost, st));
        }
        if (tr) (void)hipEventRecord(tr->ev[4], st);
        UVO_HIP_TRY(m, host_sync(m, st));
        // replay of RANSACPointSetRegistrator::run's sequential scan over the counts that exist so far, per job
        for (int s_ = 0; s_ < nb; s_++) {
            Scan& q = sc[slot_of[s_]];
            Ctx* c = lanes[q.job];
            q.computed += b.job[s_].nhyp;
            if (q.G == modelPoints) { q.active = false; continue; }
            for (; q.iter < q.niters && q.iter < q.computed; q.iter++) {
                q.last = q.iter;
                const int goodCount = c->h_hcount[q.iter];
                if (goodCount > (q.maxGood > modelPoints - 1 ? q.maxGood : modelPoints - 1)) {
                    q.best = q.iter; q.maxGood = goodCount;
                    q.niters = ransac_update_num_iters(confidence, (double)(q.G - goodCount) / q.G, modelPoints, q.niters);
                }
            }
            if (q.iter >= q.niters) q.active = false;       // the scan is over; otherwise it needs hypotheses past `computed`
        }
    }
    if (g_bdbg) { g_bstat[1] += now_us() - t_b0; t_b0 = now_us(); }
    PnpBatch b2 = b;
    b2.dbg = getenv("UVO_DBG_REFIT") != nullptr;
    int idx2[kMaxPnpBatch], nb2 = 0;
    bool need_sync = false;
    for (int k = 0; k < ns; k++) {
        const Scan& q = sc[k];
        Ctx* c = lanes[q.job];
        PnpResult& r = res[q.job];
        if (q.G == modelPoints) {              // npoints == model_points: the single model, all five points inliers
            memcpy(r.rvec, c->h_pose, sizeof(double) * 3); memcpy(r.tvec, c->h_pose + 3, sizeof(double) * 3);
            int ids[5] = {0, 1, 2, 3, 4};
            UVO_HIP_TRY(m, hipMemcpyAsync(c->d_inliers, ids, sizeof(ids), hipMemcpyHostToDevice, st));
            UVO_HIP_TRY(m, hipStreamSynchronize(st));
            r.wrote = 1; r.ok = 1; r.ninl = 5;
            continue;
        }
        if (q.best < 0) {
            // RANSAC failed: OpenCV hands back the last hypothesis' rvec/tvec and no inliers
            UVO_HIP_TRY(m, hipMemcpyAsync(c->h_pose, c->d_models + (size_t)q.last * 6, sizeof(double) * 6, hipMemcpyDeviceToHost, st));
            need_sync = true;
            r.wrote = 2;                      // pose arrives with the final sync
            continue;
        }
        b2.job[nb2] = base[k]; b2.job[nb2].best = q.best; b2.job[nb2].n_best = q.maxGood;
        idx2[nb2++] = q.job;
    }
    if (nb2 > 0) {
        b2.n = nb2;
        StageTimer t(m, ST_PNP_REFIT, st);
        hipLaunchKernelGGL(k_pnp_mask, dim3(nb2), dim3(1024), 0, st, b2);
        const size_t refit_lds = sizeof(double) * Epnp<BlockPolicy>::kStageDoubles;
        static std::once_flag refit_once[64];
        std::call_once(refit_once[m->device & 63], [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pnp_refit), hipFuncAttributeMaxDynamicSharedMemorySize, (int)refit_lds); });
        // the inlier count is known here: it is the winning hypothesis' count (k_pnp_mask repeats the same test)
        static const bool force_seq = getenv("UVO_REFIT_SEQUENTIAL") != nullptr;       // diagnostics: the OpenCV-ordered refit for every job
        bool any_fast = false, any_seq = b2.dbg != 0;
        for (int s = 0; s < nb2; s++) {
            b2.job[s].seq = force_seq || b2.job[s].n_best < kFastRefitMin;
            any_fast = any_fast || !b2.job[s].seq; any_seq = any_seq || b2.job[s].seq;
        }
        if (any_fast) hipLaunchKernelGGL(k_pnp_refit_fast, dim3(nb2), dim3(kFastThreads), 0, st, b2);
        if (any_seq) hipLaunchKernelGGL(k_pnp_refit, dim3(nb2), dim3(256), refit_lds, st, b2);
        UVO_HIP_TRY(m, hipGetLastError());
    }
    if (nb2 > 0) need_sync = true;
    if (tr) (void)hipEventRecord(tr->ev[5], st);
    if (g_bdbg) { g_bstat[2] += now_us() - t_b0; t_b0 = now_us(); }
    if (need_sync) UVO_HIP_TRY(m, host_sync(m, st));
    if (g_bdbg) { g_bstat[3] += now_us() - t_b0; g_bstat[4] += 1; }
    for (int s = 0; s < nb2; s++) {
        Ctx* c = lanes[idx2[s]];
        PnpResult& r = res[idx2[s]];
        memcpy(r.rvec, c->h_pose, sizeof(double) * 3); memcpy(r.tvec, c->h_pose + 3, sizeof(double) * 3);
        r.ninl = c->h_countsB[0]; r.ok = 1; r.wrote = 1;
    }
    for (int i = 0; i < n; i++) if (res[i].wrote == 2) {
        Ctx* c = lanes[i];
        memcpy(res[i].rvec, c->h_pose, sizeof(double) * 3); memcpy(res[i].tvec, c->h_pose + 3, sizeof(double) * 3);
        res[i].wrote = 1;
    }
    if (b2.dbg && nb2 > 0) {
        static double h[2][EPNP_SMALL];
        UVO_HIP_TRY(m, hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dbg_small), sizeof(h)));
        auto show = [&](const char* name, int off, int cnt) {
            fprintf(stderr, "[refit dbg] %-8s", name);
            for (int i = 0; i < cnt; i++) fprintf(stderr, " %.6e|%.6e", h[0][off + i], h[1][off + i]);
            fprintf(stderr, "\n");
        };
        show("cws", EP_CWS, 12); show("D", EP_D, 12); show("ut11", EP_MTM + 132, 12); show("ut10", EP_MTM + 120, 12); show("rho", EP_RHO, 6);
        show("L row0", EP_L, 10);
        for (int br = 0; br < 3; br++) { show("betas", EP_BR + br * EPB_SIZE + EPB_BETAS, 4); show("ccs", EP_BR + br * EPB_SIZE + EPB_CCS, 12); show("R", EP_BR + br * EPB_SIZE + EPB_RS, 9); show("t", EP_BR + br * EPB_SIZE + EPB_TS, 3); show("rep", EP_BR + br * EPB_SIZE + EPB_REP, 1); }
    }
    if (getenv("UVO_DBG_PHASE")) {
        long long k[16];
        UVO_HIP_TRY(m, hipMemcpyFromSymbol(k, HIP_SYMBOL(g_hyp_clk), sizeof(k)));
        fprintf(stderr, "[uvo] pnp batch %d; hyp phases (us): ctrl %.1f bary %.1f mtm %.1f svd12 %.1f betas %.1f pcs %.1f sums %.1f svd3 %.1f reproj %.1f\n", ns,
                (k[1]-k[0])*0.01, (k[2]-k[1])*0.01, (k[3]-k[2])*0.01, (k[4]-k[3])*0.01, (k[5]-k[4])*0.01, (k[6]-k[5])*0.01,
                (k[7]-k[6])*0.01, (k[8]-k[7])*0.01, (k[9]-k[8])*0.01);
        fprintf(stderr, "[uvo]   hyp betas split (us): L %.1f  find_betas %.1f  gauss_newton %.1f  ccs %.1f\n", (k[10]-k[4])*0.01, (k[11]-k[10])*0.01,
                (k[12]-k[11])*0.01, (k[5]-k[12])*0.01);
        UVO_HIP_TRY(m, hipMemcpyFromSymbol(k, HIP_SYMBOL(g_refit_clk), sizeof(k)));
        fprintf(stderr, "[uvo] pnp refit phases (us): ctrl %.1f bary %.1f mtm %.1f svd12 %.1f betas %.1f pcs %.1f sums %.1f svd3 %.1f reproj %.1f\n",
                (k[1]-k[0])*0.01, (k[2]-k[1])*0.01, (k[3]-k[2])*0.01, (k[4]-k[3])*0.01, (k[5]-k[4])*0.01, (k[6]-k[5])*0.01,
                (k[7]-k[6])*0.01, (k[8]-k[7])*0.01, (k[9]-k[8])*0.01);
    }
    return UVO_OK;
}

// single problem (the standalone operator): a batch of one on c's own buffers
uvo_status pose_pnp_ransac(Ctx* c, int slot, int G, const double* K, int iterationsCount, float reprojectionError, double confidence,
                           double* rvec, double* tvec, int* n_inliers, int* ok)
{
    (void)slot;
    PnpResult r;
    Ctx* lanes[1] = { c };
    *n_inliers = 0; *ok = 0;
    UVO_TRY(pose_pnp_ransac_batch(c, 1, lanes, &G, K, iterationsCount, reprojectionError, confidence, &r));
    if (r.st != UVO_OK) return r.st;
    if (r.wrote) { memcpy(rvec, r.rvec, sizeof(r.rvec)); memcpy(tvec, r.tvec, sizeof(r.tvec)); }
    *n_inliers = r.ninl; *ok = r.ok;
    return UVO_OK;
}

}  // namespace uvo
