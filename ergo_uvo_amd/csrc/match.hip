// match.hip -- brute-force L2 2-nearest-neighbour matching + Lowe ratio test on gfx950
// (replaces match_features, VO_utility.cpp:515-573 -> BFMatcher(NORM_L2).knnMatch(k=2)).
//
// Parity needs the exact float value of every distance (nearest/second-nearest order and the
// strict `d0 < ratio*d1` test turn on the last bit), so the contraction is done on the VALU in
// OpenCV's own summation order (normL2Sqr_: 4 lanes x 4 accumulators, mul then add, SSE
// horizontal reduce, sqrt) rather than as a |a|^2+|b|^2-2ab MFMA product, whose rounding differs.
// The work is small next to the detector (3000^2 x 64 x 3 flop ~ 1.7 GFLOP ~ tens of us of VALU
// time); see DESIGN.md for the MFMA-prefilter variant and why it is not the default.
//
// k_match_top2 : grid (query tiles of 256, train chunks of kMatchChunk).  Each thread keeps one
//                query row in registers, the chunk's train rows are staged in LDS and broadcast.
//                Emits the chunk-local top-2 (ties: lower train index first).
// k_match_merge: per query, merges the chunk results in ascending train order with BFMatcher's
//                insertion rule, then applies the ratio test.
// k_match_compact: ordered stream compaction of the surviving matches (single workgroup scan).
#include "uvo_ctx.h"
#include "uvo_math.h"

namespace uvo {

__device__ __forceinline__ float l2_distance64(const float* q /*registers*/, const float* t /*LDS, broadcast*/)
{
    float acc[16];
#pragma unroll
    for (int l = 0; l < 16; l++) acc[l] = 0.f;
#pragma unroll
    for (int j = 0; j < 64; j += 16) {
#pragma unroll
        for (int l = 0; l < 16; l++) { float d = q[j + l] - t[j + l]; acc[l] = d * d + acc[l]; }
    }
    float v0 = ((acc[0] + acc[4]) + acc[8]) + acc[12];
    float v1 = ((acc[1] + acc[5]) + acc[9]) + acc[13];
    float v2 = ((acc[2] + acc[6]) + acc[10]) + acc[14];
    float v3 = ((acc[3] + acc[7]) + acc[11]) + acc[15];
    float d = (v0 + v2) + (v1 + v3);
    return sqrtf(d);
}

// BatchDistInvoker K=2 insertion: enter iff d < worst; shift while prev > d
__device__ __forceinline__ void top2_insert(float d, int j, float& d0, int& i0, float& d1, int& i1)
{
    if (d < d1) {
        if (d0 > d) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
        else { d1 = d; i1 = j; }
    }
}

__global__ __launch_bounds__(256) void k_match_top2(const float* __restrict__ dq, const int* nq_p, int nq_imm,
                                                    const float* __restrict__ dt, const int* nt_p, int nt_imm,
                                                    float4* part, int nq_stride)
{
    const int nq = nq_p ? *nq_p : nq_imm, nt = nt_p ? *nt_p : nt_imm;
    const int q0 = blockIdx.x * 256, t0 = blockIdx.y * kMatchChunk;
    if (q0 >= nq || t0 >= nt) return;
    __shared__ __align__(16) float tile[kMatchChunk * 64];
    const int tid = threadIdx.x;
    const int cnt = min(kMatchChunk, nt - t0);
    {
        const float4* src = reinterpret_cast<const float4*>(dt + (size_t)t0 * 64);
        float4* dst = reinterpret_cast<float4*>(tile);
        for (int i = tid; i < cnt * 16; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int q = q0 + tid;
    if (q >= nq) return;
    float qr[64];
    {
        const float4* src = reinterpret_cast<const float4*>(dq + (size_t)q * 64);
#pragma unroll
        for (int i = 0; i < 16; i++) { float4 v = src[i]; qr[4*i] = v.x; qr[4*i+1] = v.y; qr[4*i+2] = v.z; qr[4*i+3] = v.w; }
    }
    float d0 = FLT_MAX, d1 = FLT_MAX; int i0 = -1, i1 = -1;
    for (int j = 0; j < cnt; j++) {
        float d = l2_distance64(qr, tile + j * 64);
        top2_insert(d, t0 + j, d0, i0, d1, i1);
    }
    part[(size_t)blockIdx.y * nq_stride + q] = make_float4(d0, __int_as_float(i0), d1, __int_as_float(i1));
}

__global__ __launch_bounds__(256) void k_match_merge(const float4* part, const int* nq_p, int nq_imm, const int* nt_p, int nt_imm,
                                                     int nq_stride, int* knn_idx, float* knn_dist)
{
    const int nq = nq_p ? *nq_p : nq_imm, nt = nt_p ? *nt_p : nt_imm;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const int nchunks = (nt + kMatchChunk - 1) / kMatchChunk;
    float d0 = FLT_MAX, d1 = FLT_MAX; int i0 = -1, i1 = -1;
    for (int c = 0; c < nchunks; c++) {
        float4 p = part[(size_t)c * nq_stride + q];
        int a = __float_as_int(p.y), b = __float_as_int(p.w);
        if (a >= 0) top2_insert(p.x, a, d0, i0, d1, i1);
        if (b >= 0) top2_insert(p.z, b, d0, i0, d1, i1);
    }
    knn_idx[2*q] = i0; knn_idx[2*q + 1] = i1;
    knn_dist[2*q] = d0; knn_dist[2*q + 1] = d1;
}

// ratio test + ordered compaction, one workgroup of 1024 threads
__global__ __launch_bounds__(1024) void k_match_compact(const int* knn_idx, const float* knn_dist, const int* nq_p, int nq_imm,
                                                        float ratio, uvo_dmatch* out, int* nout, int out_cap)
{
    const int nq = nq_p ? *nq_p : nq_imm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ int wtot[16];
    __shared__ int s_base;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int base = 0; base < nq; base += 1024) {
        int q = base + tid;
        bool keep = false; int i0 = -1; float d0 = 0.f;
        if (q < nq) {
            i0 = knn_idx[2*q]; int i1 = knn_idx[2*q + 1];
            d0 = knn_dist[2*q]; float d1 = knn_dist[2*q + 1];
            keep = i0 >= 0 && i1 >= 0 && d0 < ratio * d1;
        }
        unsigned long long bal = __ballot(keep);
        int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wtot[wv] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int k = 0; k < wv; k++) off += wtot[k];
        if (keep) {
            int pos = off + before;
            if (pos < out_cap) { uvo_dmatch m; m.queryIdx = q; m.trainIdx = i0; m.imgIdx = 0; m.distance = d0; out[pos] = m; }
        }
        __syncthreads();
        if (tid == 0) { int t = 0; for (int k = 0; k < 16; k++) t += wtot[k]; s_base += t; }
        __syncthreads();
    }
    if (tid == 0) *nout = s_base;      // may exceed out_cap: the host reports UVO_CAPACITY
}

uvo_status match_knn2(Ctx* c, const float* d_q, const int* d_nq, int nq_max, const float* d_t, const int* d_nt, int nt_max)
{
    if (nq_max <= 0 || nt_max <= 0) return UVO_OK;
    if (nq_max > c->cap || nt_max > c->cap) { c->err = "match: descriptor count exceeds the context's max_kpts"; return UVO_CAPACITY; }
    {
        StageTimer t(c, ST_MATCH);
        dim3 grid((nq_max + 255) / 256, (nt_max + kMatchChunk - 1) / kMatchChunk);
        hipLaunchKernelGGL(k_match_top2, grid, dim3(256), 0, c->stream, d_q, d_nq, nq_max, d_t, d_nt, nt_max, c->d_mpart, c->cap);
        UVO_HIP_TRY(c, hipGetLastError());
    }
    {
        StageTimer t(c, ST_MATCH_MERGE);
        hipLaunchKernelGGL(k_match_merge, dim3((nq_max + 255) / 256), dim3(256), 0, c->stream, c->d_mpart, d_nq, nq_max, d_nt, nt_max,
                           c->cap, c->d_knn_idx, c->d_knn_dist);
        UVO_HIP_TRY(c, hipGetLastError());
    }
    return UVO_OK;
}

uvo_status match_ratio_compact(Ctx* c, const int* d_nq, int nq_max, float ratio, uvo_dmatch* d_out, int* d_nout, int out_cap)
{
    StageTimer t(c, ST_MATCH_MERGE);
    hipLaunchKernelGGL(k_match_compact, dim3(1), dim3(1024), 0, c->stream, c->d_knn_idx, c->d_knn_dist, d_nq, nq_max, ratio,
                       d_out, d_nout, out_cap);
    UVO_HIP_TRY(c, hipGetLastError());
    return UVO_OK;
}

}  // namespace uvo
