// match.hip -- brute-force L2 2-nearest-neighbour matching + Lowe ratio test on gfx950
// (replaces match_features, VO_utility.cpp:515-573 -> BFMatcher(NORM_L2).knnMatch(k=2)).
//
// Parity needs the exact float value of the two winning distances (nearest/second-nearest order and the
// strict `d0 < ratio*d1` test turn on the last bit), and OpenCV's value comes from its own summation order
// (normL2Sqr_: 4 lanes x 4 accumulators, mul then add, SSE horizontal reduce, sqrt).  An all-pairs contraction in
// that order is ~190 VALU operations per pair.  So the all-pairs work is done on the matrix cores instead, as a
// SHORTLIST, and the exact order is spent only on the survivors:
//
// k_match_mfma    : S' = |t|^2 - 2 t.q for every (train, query) pair with v_mfma_f32_32x32x16_bf16 on bf16 hi/lo splits of the
//                   floats (three MFMAs per 16 k), tiles of (128 queries, 128 train rows).  Each lane owns one query
//                   column and keeps the four smallest S' of its rows (the row index rides in the low mantissa
//                   bits); per (query, chunk) it emits those four.
// k_match_resolve : 16 lanes per query.  lim = (second-smallest S' over all candidates) + margin.  Every train row that can
//                   be one of the exact two nearest has S' <= lim, so: a chunk whose fourth value is above lim
//                   contributes only its (at most three) candidates below lim; a chunk whose fourth value is not
//                   (near-duplicate descriptors) is scanned row by row.  Those rows are evaluated in OpenCV's
//                   order (the 16 lanes hold the 16 accumulators of normL2Sqr_) and the two smallest by (distance,
//                   train index) are kept -- exactly what BFMatcher's insertion rule leaves after the full scan.
// k_match_compact : ratio test + ordered stream compaction of the surviving matches (single workgroup scan).
//
// Margin: the split-bf16 contraction differs from the real t.q by less than (2^-15 + 2^-16) sum|t_k q_k| <= 2.3e-5 (|q|^2 + |t|^2),
// so S' by twice that; OpenCV's float sum is within ~64 ulp of the real squared distance and the index bits cost 2^-16 of
// |S'|: together below 6.5e-5 (|q|^2 + |t|^2) per value.  lim allows 2 * kMarginRel = 3e-4 of that scale.
#include "uvo_ctx.h"
#include "uvo_math.h"

namespace uvo {

static const int kMfmaChunk = 128;             // train rows per workgroup of k_match_mfma (7 index bits, see shortlist_key)
static const float kMarginRel = 1.5e-4f;
static const float kBig = 3.0e38f;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Shortlist key: S' with the 7 low mantissa bits replaced by the row's index inside its chunk, so that the three
// smallest candidates AND their indices are tracked with min/med3 alone.  The replaced bits change S' by less than
// 2^-16 of its magnitude, which kMarginRel includes.
__device__ __forceinline__ float shortlist_key(float s, int row_in_chunk)
{
    return __uint_as_float((__float_as_uint(s) & ~127u) | (unsigned)row_in_chunk);
}
__device__ __forceinline__ void top4_keys(float v, float& k0, float& k1, float& k2, float& k3)
{
    k3 = __builtin_amdgcn_fmed3f(k2, k3, v);
    k2 = __builtin_amdgcn_fmed3f(k1, k2, v);
    k1 = __builtin_amdgcn_fmed3f(k0, k1, v);
    k0 = fminf(k0, v);
}

// ------------------------------------------------------------------------------------------
// The contraction runs on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16: 16 k per instruction, 32 cycles) with every
// float split in two bf16 terms, x = hi + lo (+ residual below 2^-16 |x|, round-to-nearest both times), and
// t.q ~ hi.hi + hi.lo + lo.hi accumulated in f32: three MFMAs per 16 k instead of eight f32 ones (32x32x2, 64 cycles each),
// 5.3x less matrix time for a product error below 2^-15 + 2^-16 of |t_k q_k| -- inside the margin the exact
// re-evaluation of the shortlist already allows for (kMarginRel).  The chunk's train rows are split once, while they are
// staged in LDS, and |t|^2 is taken from the f32 values there.
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b)       // bits 15:0 = bf16(a), 31:16 = bf16(b), round to nearest even
{
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split_bf16(float x0, float x1, unsigned& hi, unsigned& lo)
{
    hi = cvt_pk_bf16(x0, x1);
    const float h0 = __uint_as_float(hi << 16), h1 = __uint_as_float(hi & 0xFFFF0000u);
    lo = cvt_pk_bf16(x0 - h0, x1 - h1);                                   // the differences are exact in f32
}
__device__ __forceinline__ void split_bf16x8(const float4 v0, const float4 v1, uint4& hi, uint4& lo)
{
    split_bf16(v0.x, v0.y, hi.x, lo.x); split_bf16(v0.z, v0.w, hi.y, lo.y);
    split_bf16(v1.x, v1.y, hi.z, lo.z); split_bf16(v1.z, v1.w, hi.w, lo.w);
}

// D = 64 (SURF) or 128 (SURF_EXTENDED) elements per descriptor.  uint4 per staged row: D/8 of hi (bf16), D/8 of lo, 1 of padding
// (68 or 132 words: the lanes' b128 reads spread over the LDS banks)
template <int D> struct RowQuads { static constexpr int value = D / 4 + 1; };
template <int D> static constexpr size_t match_rows_lds() { return sizeof(uint4) * kMfmaChunk * RowQuads<D>::value; }

// One or two independent matching problems per launch (blockIdx.y): the stereo loop's L -> R and prev -> curr matches are both
// ready to run once the detector has finished, and one launch of ~1060 tiles fills the chip where two of ~530 each did not.
// part: [chunk][cap] (k0, k1, k2, k3): the four smallest shortlist keys; tnmax[chunk]: max |t|^2 of the chunk's rows
struct MatchProb { const float* dq; const int* nq_p; int nq_imm; const float* dt; const int* nt_p; int nt_imm;
                   float4* part; float* tnmax; int* knn_idx; float* knn_dist; };
struct MatchBatch { MatchProb p[4]; int cap; };      // up to four problems: the two matches of two consecutive pairs (two-pair launch)

template <int D>
__global__ __launch_bounds__(256) void k_match_mfma(MatchBatch mb)
{
    const MatchProb& P = mb.p[blockIdx.y];
    const float* __restrict__ dq = P.dq; const float* __restrict__ dt = P.dt;
    float4* part = P.part; float* chunk_tnmax = P.tnmax; const int cap = mb.cap;
    const int nq = P.nq_p ? *P.nq_p : P.nq_imm, nt = P.nt_p ? *P.nt_p : P.nt_imm;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ __align__(16) float s_tn[kMfmaChunk];                     // |t|^2 of the staged rows (kBig past the end: never wins)
    extern __shared__ uint4 s_rows[];                                    // [kMfmaChunk][kRowQuads]: the chunk's train rows as bf16 hi | lo, shared by the four waves
    constexpr int kRowQuads = RowQuads<D>::value, M = D / 16, QW = D / 4;
    __shared__ int s_wmax;
    const int n = lane & 31, h = lane >> 5;
    // the counts live on the device, so the grid is a fixed number of workgroups that walk the (query tile, train chunk)
    // pairs: a cap x cap grid would be 4096 workgroups of which ~530 find work
    const int nqt = (nq + 127) / 128, ntiles = nqt * ((nt + kMfmaChunk - 1) / kMfmaChunk);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tile_q = tile % nqt, tile_c = tile / nqt;
        const int q0 = tile_q * 128 + wave * 32, t0 = tile_c * kMfmaChunk;
        // B operand: this lane's query; MFMA m of a step consumes k = 16m + 8h + j (j = 0..7) from both operands
        const int q = q0 + n;
        uint4 bhi[M], blo[M];
        const float4* src = reinterpret_cast<const float4*>(dt);
        constexpr int kPer = kMfmaChunk * QW / 256, kBatch = 4;
        static_assert(kPer % kBatch == 0, "staging batches");
        float4 vv[kBatch];
        {
            // the query's floats and the first staging batch of the chunk go out together: one memory wait instead of two
            const float4* qsrc = reinterpret_cast<const float4*>(dq + (size_t)min(q, nq - 1) * D);
            float4 qraw[2 * M];
#pragma unroll
            for (int m = 0; m < M; m++) { qraw[2 * m] = qsrc[4 * m + 2 * h]; qraw[2 * m + 1] = qsrc[4 * m + 2 * h + 1]; }
#pragma unroll
            for (int u = 0; u < kBatch; u++) {
                const int e = threadIdx.x + 256 * u, r = e / QW, c4 = e % QW;
                vv[u] = src[(size_t)min(t0 + r, nt - 1) * QW + c4];
            }
#pragma unroll
            for (int m = 0; m < M; m++) split_bf16x8(qraw[2 * m], qraw[2 * m + 1], bhi[m], blo[m]);
        }
        if (threadIdx.x == 0) s_wmax = 0;
        __syncthreads();
        {   // stage the chunk once: 128 rows x D/4 float4, coalesced, split into bf16 hi / lo; rows past the end repeat the last one
            // four loads in flight per thread before the first split (round 3: the loop used to wait for each of its eight -- sixteen
            // for 128-wide rows -- loads in turn, eight memory round trips per tile; eight in flight cost a wave per SIMD)
#pragma unroll 1
            for (int b0 = 0; b0 < kPer; b0 += kBatch) {
                if (b0 > 0) {
#pragma unroll
                    for (int u = 0; u < kBatch; u++) {
                        const int e = threadIdx.x + 256 * (b0 + u), r = e / QW, c4 = e % QW;
                        vv[u] = src[(size_t)min(t0 + r, nt - 1) * QW + c4];
                    }
                }
#pragma unroll
                for (int u = 0; u < kBatch; u++) {
                    const int e = threadIdx.x + 256 * (b0 + u), r = e / QW, c4 = e % QW;
                    const float4 v = vv[u];
                    uint2 hi, lo;
                    split_bf16(v.x, v.y, hi.x, lo.x); split_bf16(v.z, v.w, hi.y, lo.y);
                    uint2* row = reinterpret_cast<uint2*>(s_rows + r * kRowQuads);
                    row[c4] = hi; row[QW + c4] = lo;
                    // |t|^2 of the row (any summation order will do: the margin absorbs it): its D/4 pieces sit on D/4 adjacent lanes
                    float tn = __builtin_fmaf(v.x, v.x, __builtin_fmaf(v.y, v.y, __builtin_fmaf(v.z, v.z, v.w * v.w)));
                    if (QW == 32) tn += __shfl_xor(tn, 16);
                    tn += __shfl_xor(tn, 8); tn += __shfl_xor(tn, 4); tn += __shfl_xor(tn, 2); tn += __shfl_xor(tn, 1);
                    if (c4 == 0) {
                        const bool valid = t0 + r < nt;
                        s_tn[r] = valid ? tn : kBig;
                        if (valid) atomicMax(&s_wmax, __float_as_int(tn));     // non-negative floats order as their bit patterns
                    }
                }
            }
        }
        __syncthreads();
        float k0 = kBig, k1 = kBig, k2 = kBig, k3 = kBig;
        const int tend = min(t0 + kMfmaChunk, nt);
        for (int tb = t0; tb < tend; tb += 32) {
            // A operand: train row tb + n, the same k ranges
            const uint4* rowp = s_rows + (tb - t0 + n) * kRowQuads;
            f32x16 acc;
#pragma unroll
            for (int j = 0; j < 16; j++) acc[j] = 0.f;
#pragma unroll
            for (int m = 0; m < M; m++) {
                const bf16x8 ahi = __builtin_bit_cast(bf16x8, rowp[2 * m + h]), alo = __builtin_bit_cast(bf16x8, rowp[D / 8 + 2 * m + h]);
                const bf16x8 qhi = __builtin_bit_cast(bf16x8, bhi[m]), qlo = __builtin_bit_cast(bf16x8, blo[m]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, qhi, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, qlo, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, qhi, acc, 0, 0, 0);
            }
            // acc[j] = t.q for train row tb + (j&3) + 8(j>>2) + 4h and this lane's query
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int rl = 8*g + 4*h;
                const float4 nn = *reinterpret_cast<const float4*>(s_tn + (tb - t0) + rl);
                const int rc = tb - t0 + rl;
                top4_keys(shortlist_key(__builtin_fmaf(-2.f, acc[4*g + 0], nn.x), rc),     k0, k1, k2, k3);
                top4_keys(shortlist_key(__builtin_fmaf(-2.f, acc[4*g + 1], nn.y), rc + 1), k0, k1, k2, k3);
                top4_keys(shortlist_key(__builtin_fmaf(-2.f, acc[4*g + 2], nn.z), rc + 2), k0, k1, k2, k3);
                top4_keys(shortlist_key(__builtin_fmaf(-2.f, acc[4*g + 3], nn.w), rc + 3), k0, k1, k2, k3);
            }
        }
        // the other half-wave saw the other rows of the same query: merge its four
        {
            float o0 = __shfl_xor(k0, 32), o1 = __shfl_xor(k1, 32), o2 = __shfl_xor(k2, 32), o3 = __shfl_xor(k3, 32);
            top4_keys(o0, k0, k1, k2, k3); top4_keys(o1, k0, k1, k2, k3); top4_keys(o2, k0, k1, k2, k3); top4_keys(o3, k0, k1, k2, k3);
        }
        if (h == 0 && q < nq) part[(size_t)tile_c * cap + q] = make_float4(k0, k1, k2, k3);
        if (tile_q == 0 && threadIdx.x == 0) chunk_tnmax[tile_c] = __int_as_float(s_wmax);      // the largest |t|^2 of the chunk
        __syncthreads();                                        // s_rows is restaged by the next tile
    }
}

// two smallest of the union of two (lo <= hi) pairs
__device__ __forceinline__ void merge_min2(float& b0, float& b1, float o0, float o1)
{
    float n0 = fminf(b0, o0);
    float n1 = fminf(fmaxf(b0, o0), fminf(b1, o1));
    b0 = n0; b1 = n1;
}
// (d, t) lexicographic top-2: what BatchDistInvoker's insertion leaves after visiting every row in index order
__device__ __forceinline__ void top2_lex(float d, int t, float& d0, int& i0, float& d1, int& i1)
{
    if (t < 0) return;
    if (d < d0 || (d == d0 && t < i0)) { d1 = d0; i1 = i0; d0 = d; i0 = t; }
    else if (d < d1 || (d == d1 && t < i1)) { d1 = d; i1 = t; }
}
// normL2Sqr_ + sqrt with the 16 accumulators on the 16 lanes of a group: lane l holds elements 64v + 4l .. 64v + 4l + 3 of the row
// in its v-th float4 (v < D/64); returns on every lane
template <int NV>
__device__ __forceinline__ float group_distance_tv(const float4 (&qv)[NV], const float4 (&tv)[NV], int sub)
{
    // accumulator a (= j mod 16) sums elements a, a+16, a+32, ... in that order; element e lives in float4 e/64 of lane (e%64)/4, slot e%4:
    // lane `sub` computes accumulator a = sub from slot sub%4 of lanes sub/4 + 4k (k = 0..3), first float4 first
    const int base = (threadIdx.x & 63 & ~15);
    const int slot = sub & 3;
    float acc = 0.f;
#pragma unroll
    for (int v = 0; v < NV; v++) {
        const float dx = qv[v].x - tv[v].x, dy = qv[v].y - tv[v].y, dz = qv[v].z - tv[v].z, dw = qv[v].w - tv[v].w;
        const float px = dx * dx, py = dy * dy, pz = dz * dz, pw = dw * dw;       // squared differences of this lane's four elements
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int src = base + (sub >> 2) + 4 * k;           // lane holding element 64v + sub + 16k
            const float vx = __shfl(px, src), vy = __shfl(py, src), vz = __shfl(pz, src), vw = __shfl(pw, src);
            const float term = slot == 0 ? vx : (slot == 1 ? vy : (slot == 2 ? vz : vw));
            acc = term + acc;
        }
    }
    // v[l] = ((acc[l] + acc[4+l]) + acc[8+l]) + acc[12+l], d = (v0 + v2) + (v1 + v3)
    const int l = sub & 3;
    const float a0 = __shfl(acc, base + l), a1 = __shfl(acc, base + 4 + l), a2 = __shfl(acc, base + 8 + l), a3 = __shfl(acc, base + 12 + l);
    const float v = ((a0 + a1) + a2) + a3;                  // lanes with the same l hold the same v
    const float v0 = __shfl(v, base + 0), v1 = __shfl(v, base + 1), v2 = __shfl(v, base + 2), v3 = __shfl(v, base + 3);
    return sqrtf((v0 + v2) + (v1 + v3));
}
template <int NV>
__device__ __forceinline__ void load_row(const float* __restrict__ row, int sub, float4 (&out)[NV])
{
#pragma unroll
    for (int v = 0; v < NV; v++) out[v] = reinterpret_cast<const float4*>(row)[16 * v + sub];
}
template <int NV>
__device__ __forceinline__ float group_distance(const float4 (&qv)[NV], const float* __restrict__ trow, int sub)
{
    float4 tv[NV];
    load_row<NV>(trow, sub, tv);
    return group_distance_tv<NV>(qv, tv, sub);
}

template <int D>
__global__ __launch_bounds__(256) void k_match_resolve(MatchBatch mb)
{
    const MatchProb& P = mb.p[blockIdx.y];
    const float* __restrict__ dq = P.dq; const float* __restrict__ dt = P.dt;
    const float4* part = P.part; const float* chunk_tnmax = P.tnmax; const int cap = mb.cap;
    int* knn_idx = P.knn_idx; float* knn_dist = P.knn_dist;
    const int nq = P.nq_p ? *P.nq_p : P.nq_imm, nt = P.nt_p ? *P.nt_p : P.nt_imm;
    const int sub = threadIdx.x & 15;
    const int q = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (blockIdx.x * 16 >= nq) return;
    const bool live = q < nq;                              // whole groups are live or not
    const int qq = live ? q : nq - 1;
    constexpr int NV = D / 64;
    float4 qv[NV];
    load_row<NV>(dq + (size_t)qq * D, sub, qv);
    float qn = 0.f;
#pragma unroll
    for (int v = 0; v < NV; v++) qn += qv[v].x*qv[v].x + qv[v].y*qv[v].y + qv[v].z*qv[v].z + qv[v].w*qv[v].w;
    qn += __shfl_xor(qn, 8); qn += __shfl_xor(qn, 4); qn += __shfl_xor(qn, 2); qn += __shfl_xor(qn, 1);
    const int nchunks = (nt + kMfmaChunk - 1) / kMfmaChunk;
    // pass 1: two smallest S' over all candidates of this query (lanes stride the chunks)
    float b0 = kBig, b1 = kBig, tnmax = 0.f;
    for (int c = sub; c < nchunks; c += 16) {
        float4 p = part[(size_t)c * cap + qq];
        merge_min2(b0, b1, p.x, p.y);
        tnmax = fmaxf(tnmax, chunk_tnmax[c]);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { merge_min2(b0, b1, __shfl_xor(b0, o), __shfl_xor(b1, o)); tnmax = fmaxf(tnmax, __shfl_xor(tnmax, o)); }
    const float lim = b1 + 2.f * kMarginRel * (qn + tnmax);
    // pass 2: every row that can be among the exact two nearest, evaluated by the whole group
    float d0 = FLT_MAX, d1 = FLT_MAX; int i0 = -1, i1 = -1;
    for (int cb = 0; cb < nchunks; cb += 16) {
        const int c = cb + sub;
        int a = -1, b = -1, e3 = -1; bool scan = false;
        if (c < nchunks) {
            float4 p = part[(size_t)c * cap + qq];
            scan = p.w <= lim;                             // a fourth row is this close: the shortlist may be incomplete
            if (p.x <= lim) a = c * kMfmaChunk + (int)(__float_as_uint(p.x) & 127u);
            if (p.y <= lim) b = c * kMfmaChunk + (int)(__float_as_uint(p.y) & 127u);
            if (p.z <= lim) e3 = c * kMfmaChunk + (int)(__float_as_uint(p.z) & 127u);
            // keys of the padding rows of the last 32-row block (|t|^2 = kBig) only get under lim when the train set has a
            // single row (lim is then a padding key itself): they are not rows
            if (a >= nt) a = -1;
            if (b >= nt) b = -1;
            if (e3 >= nt) e3 = -1;
        }
        const int base = threadIdx.x & 63 & ~15;
        const int gshift = threadIdx.x & 48;                // this group's 16 bits of a wave ballot
        // chunks whose shortlist may be incomplete are scanned row by row (rare: near-duplicate descriptors)
        unsigned gm_scan = (unsigned)(__ballot(scan) >> gshift) & 0xFFFFu;
        while (gm_scan) {
            const int l = __ffs(gm_scan) - 1;
            gm_scan &= gm_scan - 1;
            const int cc = cb + l, te = min((cc + 1) * kMfmaChunk, nt);
            for (int t = cc * kMfmaChunk; t < te; t++) top2_lex(group_distance<NV>(qv, dt + (size_t)t * D, sub), t, d0, i0, d1, i1);
        }
        if (scan) { a = -1; b = -1; e3 = -1; }             // covered by the scan
        // the other chunks' candidates: slot s = 16*which + lane; four train rows are fetched at a time (one memory round trip
        // per four candidates instead of one each) and evaluated by the whole group, in any order (top2_lex is symmetric)
        unsigned long long m = ((unsigned long long)((unsigned)(__ballot(a >= 0) >> gshift) & 0xFFFFu)) |
                               ((unsigned long long)((unsigned)(__ballot(b >= 0) >> gshift) & 0xFFFFu) << 16) |
                               ((unsigned long long)((unsigned)(__ballot(e3 >= 0) >> gshift) & 0xFFFFu) << 32);
        while (m) {
            int cnd[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int sl = m ? __ffsll((long long)m) - 1 : -1;
                if (m) m &= m - 1;
                const int ln = base + (sl & 15);
                const int va = __shfl(a, ln), vb = __shfl(b, ln), vc = __shfl(e3, ln);
                cnd[k] = sl < 0 ? -1 : (sl < 16 ? va : (sl < 32 ? vb : vc));
            }
            float4 tv[4][NV];
#pragma unroll
            for (int k = 0; k < 4; k++) load_row<NV>(dt + (size_t)max(cnd[k], 0) * D, sub, tv[k]);
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (cnd[k] >= 0) top2_lex(group_distance_tv<NV>(qv, tv[k], sub), cnd[k], d0, i0, d1, i1);
        }
    }
    if (live && sub == 0) {
        knn_idx[2*q] = i0; knn_idx[2*q + 1] = i1;
        knn_dist[2*q] = d0; knn_dist[2*q + 1] = d1;
    }
}

// ratio test + ordered compaction, one workgroup of 1024 threads; with two problems they are compacted one after the other (the
// second's query count is the one the first's gate has just written: VO:567 decides whether the triangular matches are used)
// qmap (two-pair launch, the second pair's triangular match): query row q of the compaction is row qmap[q].queryIdx of the matcher's
// query set -- the matcher ran on ALL left descriptors of the pair before (its "after stereo match" set is the subsequence its stereo
// matches select, which did not exist yet when the launch was queued); a row's two nearest neighbours do not depend on the other rows.
struct CompactProb { const int* knn_idx; const float* knn_dist; const int* nq_p; int nq_imm; uvo_dmatch* out; int* nout; int out_cap; GateArgs g; const uvo_dmatch* qmap; };
struct CompactBatch { CompactProb p[4]; int np; float ratio; };
__global__ __launch_bounds__(1024) void k_match_compact(CompactBatch cb)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ int wtot[16];
    __shared__ int s_base;
    for (int pi = 0; pi < cb.np; pi++) {
        const CompactProb& P = cb.p[pi];
        const int* knn_idx = P.knn_idx; const float* knn_dist = P.knn_dist;
        uvo_dmatch* out = P.out; const int out_cap = P.out_cap; const GateArgs g = P.g; const float ratio = cb.ratio;
        if (tid == 0) s_base = 0;
        __syncthreads();
        const int nq = P.nq_p ? *P.nq_p : P.nq_imm;
        for (int base = 0; base < nq; base += 1024) {
            int q = base + tid;
            bool keep = false; int i0 = -1; float d0 = 0.f;
            if (q < nq) {
                const int r = P.qmap ? P.qmap[q].queryIdx : q;
                i0 = knn_idx[2*r]; int i1 = knn_idx[2*r + 1];
                d0 = knn_dist[2*r]; float d1 = knn_dist[2*r + 1];
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
        if (tid == 0) {
            *P.nout = s_base;                // may exceed out_cap: the host reports UVO_CAPACITY
            int* cn = g.cn;
            if (g.mode == 1) {             // VO:567: results_match_curr.size() > MIN_NUM_FEATURES, else the "after stereo match" sets stay empty
                int M = cn[CN_NQA] > 0 ? cn[CN_M] : 0;
                if (cn[CN_NQA] == 0) cn[CN_M] = 0;
                int meff = (M > g.min_features) ? min(M, g.cap) : 0;
                cn[CN_MEFF] = meff;
                *g.as_curr_n = meff;
                cn[CN_NQB] = meff > 0 ? *g.as_prev_n : 0;      // triangular matching only runs inside that branch
            } else if (g.mode == 2) {      // VO:626: results_match_prev_curr.size() > MIN_NUM_FEATURES
                int T = cn[CN_NQB] > 0 ? cn[CN_TRAW] : 0;
                if (cn[CN_NQB] == 0) cn[CN_TRAW] = 0;
                cn[CN_T] = (T > g.min_features) ? min(T, g.cap) : 0;
                cn[CN_G] = 0;
            }
        }
        __threadfence_block();
        __syncthreads();                    // the next problem reads the counters written above
    }
}

static size_t mpart_elems(const Ctx* c) { return (size_t)((c->cap + kMfmaChunk - 1) / kMfmaChunk) * c->cap; }     // float4 per problem
static MatchProb make_prob(Ctx* c, int slot, const float* d_q, const int* d_nq, int nq_max, const float* d_t, const int* d_nt, int nt_max)
{
    MatchProb p;
    p.dq = d_q; p.nq_p = d_nq; p.nq_imm = nq_max; p.dt = d_t; p.nt_p = d_nt; p.nt_imm = nt_max;
    p.part = c->d_mpart + (size_t)slot * mpart_elems(c);
    p.tnmax = c->d_mscratch + (size_t)slot * ((c->cap + kMfmaChunk - 1) / kMfmaChunk + 4);
    p.knn_idx = c->d_knn_idx + (size_t)slot * 2 * c->cap; p.knn_dist = c->d_knn_dist + (size_t)slot * 2 * c->cap;
    return p;
}

template <int D>
static uvo_status match_launch(Ctx* c, const MatchBatch& mb, int np, int nq_max, int nt_max)
{
    {
        StageTimer t(c, ST_MATCH);
        const int tiles_max = ((nq_max + 127) / 128) * ((nt_max + kMfmaChunk - 1) / kMfmaChunk);
        // ~4.5 workgroups per CU over the whole launch, so that the ~1060 tiles of the loop's two 3000 x 3000 problems each get a
        // workgroup of their own (768: 22.8 us -- a quarter of the workgroups walk two tiles --, 1024: 21.2, 1152: 20.3, 1536: 20.6);
        // larger problems loop.  A tile's time is its memory round trips: the staging loop used to wait for each of its eight loads
        // in turn; four in flight, the first batch together with the query's own loads: 20.3 -> 17.8 us (eight in flight: 128
        // VGPRs, three waves per SIMD, fewer resident workgroups than tiles, 20.3).  Measured without effect: the four 32-row
        // blocks of a chunk unrolled two at a time (22.5 us before the grid change; all four: 155 VGPRs, two waves per SIMD, 24.7).
        static const int gtot = getenv("UVO_MATCH_GRID") ? atoi(getenv("UVO_MATCH_GRID")) : 1152;
        const int gmax = (np > 2 ? 2 * gtot : gtot) / np;      // four problems (two-pair launch): twice the tiles
        dim3 grid(tiles_max < gmax ? tiles_max : gmax, np);
        static bool attr_dev[64] = {false};                   // more than 64 KB of LDS (D = 128) has to be asked for, once per device
        bool& attr_set = attr_dev[c->device & 63];
        if (!attr_set && match_rows_lds<D>() > 48 * 1024) {
            UVO_HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_match_mfma<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)match_rows_lds<D>()));
            attr_set = true;
        }
        hipLaunchKernelGGL(k_match_mfma<D>, grid, dim3(256), match_rows_lds<D>(), c->stream, mb);
        UVO_HIP_TRY(c, hipGetLastError());
    }
    {
        StageTimer t(c, ST_MATCH_MERGE);
        hipLaunchKernelGGL(k_match_resolve<D>, dim3((nq_max + 15) / 16, np), dim3(256), 0, c->stream, mb);
        UVO_HIP_TRY(c, hipGetLastError());
    }
    return UVO_OK;
}

// rows of c->desc_dim() floats: SURF::descriptorSize(), 64 or 128 (SURF_EXTENDED)
uvo_status match_knn2(Ctx* c, const float* d_q, const int* d_nq, int nq_max, const float* d_t, const int* d_nt, int nt_max)
{
    if (nq_max <= 0 || nt_max <= 0) return UVO_OK;
    if (nq_max > c->cap || nt_max > c->cap) { c->err = "match: descriptor count exceeds the context's max_kpts"; return UVO_CAPACITY; }
    MatchBatch mb;
    mb.p[0] = mb.p[1] = mb.p[2] = mb.p[3] = make_prob(c, 0, d_q, d_nq, nq_max, d_t, d_nt, nt_max); mb.cap = c->cap;
    return c->desc_dim() == 128 ? match_launch<128>(c, mb, 1, nq_max, nt_max) : match_launch<64>(c, mb, 1, nq_max, nt_max);
}

// Two independent problems in one launch each of the shortlist and the resolve kernels: results of problem s in slot s of the
// shortlist / kNN buffers (match_ratio_compact2 reads them from there).
uvo_status match_knn2_two(Ctx* c, const float* d_q0, const int* d_nq0, const float* d_t0, const int* d_nt0,
                          const float* d_q1, const int* d_nq1, const float* d_t1, const int* d_nt1, int n_max)
{
    if (n_max <= 0) return UVO_OK;
    if (n_max > c->cap) { c->err = "match: descriptor count exceeds the context's max_kpts"; return UVO_CAPACITY; }
    MatchBatch mb;
    mb.p[0] = make_prob(c, 0, d_q0, d_nq0, n_max, d_t0, d_nt0, n_max);
    mb.p[1] = make_prob(c, 1, d_q1, d_nq1, n_max, d_t1, d_nt1, n_max);
    mb.p[2] = mb.p[3] = mb.p[0];
    mb.cap = c->cap;
    return c->desc_dim() == 128 ? match_launch<128>(c, mb, 2, n_max, n_max) : match_launch<64>(c, mb, 2, n_max, n_max);
}

// The four matches of two consecutive pairs (lanes a, b of one stream; pair b follows pair a) in one launch each of the shortlist and
// resolve kernels, then their ratio tests, gates and ordered compactions in one launch (k_match_compact, in dependency order):
//   a: stereo L_a -> R_a, triangular prev-set -> L_a           (as match_knn2_two / match_ratio_compact2 for lane a)
//   b: stereo L_b -> R_b, triangular over ALL rows of L_a -> L_b, compacted through a's stereo matches (CompactProb::qmap): row i
//      of a's "after stereo match" set is row m_a[i].queryIdx of L_a, and that set's length is what a's gate (VO:567) has just written.
// Kernels go to a's stream; each lane's results land in its own shortlist / kNN / match buffers.
uvo_status match_two_pairs(Ctx* a, Ctx* b, const float* d_prev_desc, const int* d_prev_n, const int* d_prev_as_n, int curr_a, int curr_b,
                           float ratio, int min_features)
{
    const int cap = a->cap;
    int* ca = a->d_counts; int* cbn = b->d_counts;
    MatchBatch mb;
    mb.p[0] = make_prob(a, 0, a->det[0].desc, ca + CN_NQA, cap, a->det[1].desc, ca + CN_NR, cap);
    mb.p[1] = make_prob(a, 1, d_prev_desc, d_prev_n, cap, a->det[0].desc, ca + CN_NL, cap);
    mb.p[2] = make_prob(b, 0, b->det[0].desc, cbn + CN_NQA, cap, b->det[1].desc, cbn + CN_NR, cap);
    mb.p[3] = make_prob(b, 1, a->det[0].desc, ca + CN_NL, cap, b->det[0].desc, cbn + CN_NL, cap);
    mb.cap = cap;
    UVO_TRY(a->desc_dim() == 128 ? match_launch<128>(a, mb, 4, cap, cap) : match_launch<64>(a, mb, 4, cap, cap));
    CompactBatch cb;
    const GateArgs ga1 = { 1, ca, min_features, cap, a->d_as_n + curr_a, d_prev_as_n };            // VO:567, pair a
    const GateArgs ga2 = { 2, ca, min_features, cap, nullptr, nullptr };                            // VO:626
    const GateArgs gb1 = { 1, cbn, min_features, cap, b->d_as_n + curr_b, a->d_as_n + curr_a };     // pair b: its "previous" set is a's
    const GateArgs gb2 = { 2, cbn, min_features, cap, nullptr, nullptr };
    cb.p[0] = CompactProb{ mb.p[0].knn_idx, mb.p[0].knn_dist, ca + CN_NQA, cap, a->d_matches[0], ca + CN_M, cap, ga1, nullptr };
    cb.p[1] = CompactProb{ mb.p[1].knn_idx, mb.p[1].knn_dist, ca + CN_NQB, cap, a->d_matches[1], ca + CN_TRAW, cap, ga2, nullptr };
    cb.p[2] = CompactProb{ mb.p[2].knn_idx, mb.p[2].knn_dist, cbn + CN_NQA, cap, b->d_matches[0], cbn + CN_M, cap, gb1, nullptr };
    cb.p[3] = CompactProb{ mb.p[3].knn_idx, mb.p[3].knn_dist, cbn + CN_NQB, cap, b->d_matches[1], cbn + CN_TRAW, cap, gb2, a->d_matches[0] };
    cb.np = 4; cb.ratio = ratio;
    hipLaunchKernelGGL(k_match_compact, dim3(1), dim3(1024), 0, a->stream, cb);
    UVO_HIP_TRY(a, hipGetLastError());
    return UVO_OK;
}

uvo_status match_ratio_compact(Ctx* c, const int* d_nq, int nq_max, float ratio, uvo_dmatch* d_out, int* d_nout, int out_cap,
                               const GateArgs* gate)
{
    StageTimer t(c, ST_MATCH_MERGE);
    CompactBatch cb;
    GateArgs g = { 0, nullptr, 0, 0, nullptr, nullptr };
    if (gate) g = *gate;
    cb.p[0] = CompactProb{ c->d_knn_idx, c->d_knn_dist, d_nq, nq_max, d_out, d_nout, out_cap, g, nullptr };
    cb.p[1] = cb.p[2] = cb.p[3] = cb.p[0]; cb.np = 1; cb.ratio = ratio;
    hipLaunchKernelGGL(k_match_compact, dim3(1), dim3(1024), 0, c->stream, cb);
    UVO_HIP_TRY(c, hipGetLastError());
    return UVO_OK;
}

// the ratio tests + compactions of match_knn2_two's problems, with their gates, in order, in one launch
uvo_status match_ratio_compact2(Ctx* c, float ratio, const int* d_nq0, uvo_dmatch* d_out0, int* d_nout0, const GateArgs& g0,
                                const int* d_nq1, uvo_dmatch* d_out1, int* d_nout1, const GateArgs& g1, int n_max, int out_cap)
{
    StageTimer t(c, ST_MATCH_MERGE);
    CompactBatch cb;
    cb.p[0] = CompactProb{ c->d_knn_idx, c->d_knn_dist, d_nq0, n_max, d_out0, d_nout0, out_cap, g0, nullptr };
    cb.p[1] = CompactProb{ c->d_knn_idx + (size_t)2 * c->cap, c->d_knn_dist + (size_t)2 * c->cap, d_nq1, n_max, d_out1, d_nout1, out_cap, g1, nullptr };
    cb.p[2] = cb.p[3] = cb.p[0];
    cb.np = 2; cb.ratio = ratio;
    hipLaunchKernelGGL(k_match_compact, dim3(1), dim3(1024), 0, c->stream, cb);
    UVO_HIP_TRY(c, hipGetLastError());
    return UVO_OK;
}

// ------------------------------------------------------------------------------------------
// The AKAZE / ORB branch of match_features (VO_utility.cpp:520-524): BFMatcher(NORM_HAMMING).knnMatch(k = 2) on binary
// descriptors of `bytes` bytes (32 for ORB, 61 for AKAZE's MLDB), rows padded with zeros to 64 bytes while they are staged.
// Integer distances, so every order of evaluation gives OpenCV's values; what has to be reproduced is BatchDistInvoker's insertion
// order -- of equal distances the lower train index comes first -- which the (distance, index) lexicographic top-2 does.
//   k_hamming_partial : a thread per query, blockIdx.y = a chunk of 512 train rows staged in LDS: the chunk's best two
//   k_hamming_merge   : a thread per query merges the chunks' pairs in chunk order
// ------------------------------------------------------------------------------------------
static const int kHamChunk = 512;
struct Ham2 { int d0, i0, d1, i1; };
__device__ __forceinline__ void ham_top2(int d, int t, Ham2& r)
{
    if (d < r.d0 || (d == r.d0 && t < r.i0)) { r.d1 = r.d0; r.i1 = r.i0; r.d0 = d; r.i0 = t; }
    else if (d < r.d1 || (d == r.d1 && t < r.i1)) { r.d1 = d; r.i1 = t; }
}
__global__ __launch_bounds__(256) void k_hamming_partial(const uint8_t* __restrict__ dq, int nq, const uint8_t* __restrict__ dt, int nt, int bytes, Ham2* part)
{
    __shared__ uint4 s_rows[kHamChunk * 4];                         // 512 rows x 64 bytes
    const int q = blockIdx.x * 256 + threadIdx.x, t0 = blockIdx.y * kHamChunk, cnt = min(kHamChunk, nt - t0);
    for (int e = threadIdx.x; e < cnt * 16; e += 256) {            // a row as 16 words, bytes past `bytes` are zero
        const int r = e >> 4, wi = e & 15;
        unsigned v = 0;
        for (int b = 0; b < 4; b++) { const int o = wi * 4 + b; if (o < bytes) v |= (unsigned)dt[(size_t)(t0 + r) * bytes + o] << (8 * b); }
        reinterpret_cast<unsigned*>(s_rows)[r * 16 + wi] = v;
    }
    unsigned qw[16];
#pragma unroll
    for (int wi = 0; wi < 16; wi++) {
        unsigned v = 0;
        if (q < nq) for (int b = 0; b < 4; b++) { const int o = wi * 4 + b; if (o < bytes) v |= (unsigned)dq[(size_t)q * bytes + o] << (8 * b); }
        qw[wi] = v;
    }
    __syncthreads();
    Ham2 best = { 0x7FFFFFFF, -1, 0x7FFFFFFF, -1 };
    for (int r = 0; r < cnt; r++) {
        int d = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint4 tv = s_rows[r * 4 + k];                    // the same address for every lane: a broadcast
            d += __popc(qw[4*k] ^ tv.x) + __popc(qw[4*k + 1] ^ tv.y) + __popc(qw[4*k + 2] ^ tv.z) + __popc(qw[4*k + 3] ^ tv.w);
        }
        ham_top2(d, t0 + r, best);
    }
    if (q < nq) part[(size_t)blockIdx.y * nq + q] = best;
}
__global__ __launch_bounds__(256) void k_hamming_merge(const Ham2* part, int nq, int nchunks, int* knn_idx, float* knn_dist)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    Ham2 best = { 0x7FFFFFFF, -1, 0x7FFFFFFF, -1 };
    for (int c = 0; c < nchunks; c++) {
        const Ham2 p = part[(size_t)c * nq + q];
        if (p.i0 >= 0) ham_top2(p.d0, p.i0, best);
        if (p.i1 >= 0) ham_top2(p.d1, p.i1, best);
    }
    knn_idx[2*q] = best.i0; knn_idx[2*q + 1] = best.i1;
    knn_dist[2*q] = best.i0 >= 0 ? (float)best.d0 : FLT_MAX; knn_dist[2*q + 1] = best.i1 >= 0 ? (float)best.d1 : FLT_MAX;
}

// d_q, d_t: device, n x bytes.  Results in slot 0 of the kNN buffers (match_ratio_compact reads them there); the shortlist buffer
// of the L2 matcher is the scratch for the chunks' pairs.
uvo_status match_knn2_hamming(Ctx* c, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int bytes)
{
    if (nq <= 0 || nt <= 0) return UVO_OK;
    if (nq > c->cap || nt > c->cap) { c->err = "match: descriptor count exceeds the context's max_kpts"; return UVO_CAPACITY; }
    if (bytes < 1 || bytes > 64) { c->err = "match (Hamming): descriptors of 1..64 bytes (ORB 32, AKAZE 61)"; return UVO_INVALID_ARG; }
    const int nchunks = (nt + kHamChunk - 1) / kHamChunk;
    Ham2* part = reinterpret_cast<Ham2*>(c->d_mpart);             // (cap / 512) x cap x 16 B fits the (cap / 128) x cap x 16 B shortlist
    StageTimer t(c, ST_MATCH);
    hipLaunchKernelGGL(k_hamming_partial, dim3((nq + 255) / 256, nchunks), dim3(256), 0, c->stream, d_q, nq, d_t, nt, bytes, part);
    hipLaunchKernelGGL(k_hamming_merge, dim3((nq + 255) / 256), dim3(256), 0, c->stream, part, nq, nchunks, c->d_knn_idx, c->d_knn_dist);
    UVO_HIP_TRY(c, hipGetLastError());
    return UVO_OK;
}

}  // namespace uvo
