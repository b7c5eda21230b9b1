/*
 * o_match.c -- CPU ORACLE (test infrastructure): brute-force L2 kNN (k=2) + Lowe ratio.
 * Follows VOU:515-543 (5-arg match_features) / VOU:551-573 (7-arg) and restates
 * [UPSTREAM] features2d BFMatcher::knnMatchImpl, core batchDistance (BatchDistInvoker,
 * batchDistL2_32f) and normL2Sqr_(const float*, const float*, int).  SURVEY.md App. A.2.
 * PARITY UNPINNED vs OpenCV.  The float summation order of normL2Sqr_ depends on the
 * OpenCV build's SIMD width; this oracle fixes the x86-64 baseline (SSE2, 4 lanes x 4
 * accumulators, mul+add not fused) as the canonical order -- only index pairs are a
 * meaningful cross-build parity target (hard part 5).
 */
#include "uvo_oracle.h"
#include <math.h>
#include <float.h>

/* [UPSTREAM] core/src/norm.cpp normL2Sqr_ + batch_distance.cpp batchDistL2_32f (sqrt) */
float orc_l2_distance_f32(const float* a, const float* b, int n)
{
    float acc[16];
    int j = 0, l;
    for (l = 0; l < 16; l++) acc[l] = 0.f;
    for (; j <= n - 16; j += 16)
        for (l = 0; l < 16; l++) { float t = a[j + l] - b[j + l]; acc[l] = t * t + acc[l]; }
    float v[4];
    for (l = 0; l < 4; l++) v[l] = ((acc[l] + acc[4 + l]) + acc[8 + l]) + acc[12 + l];
    float d = (v[0] + v[2]) + (v[1] + v[3]);     /* v_reduce_sum, SSE */
    for (; j < n; j++) { float t = a[j] - b[j]; d += t * t; }
    return sqrtf(d);
}

/* [UPSTREAM] batch_distance.cpp BatchDistInvoker, K = 2: dist init FLT_MAX, idx init -1;
 * a candidate enters iff d < dist[K-1]; shift while dist[k] > d. */
void orc_knn2(const float* d1, int n1, const float* d2, int n2, int dim, int* idx, float* dist)
{
    for (int i = 0; i < n1; i++) {
        float* dp = dist + 2 * i; int* ip = idx + 2 * i;
        dp[0] = dp[1] = FLT_MAX; ip[0] = ip[1] = -1;
        for (int j = 0; j < n2; j++) {
            float d = orc_l2_distance_f32(d1 + (size_t)i * dim, d2 + (size_t)j * dim, dim);
            if (d < dp[1]) {
                int k;
                for (k = 0; k >= 0 && dp[k] > d; k--) { ip[k + 1] = ip[k]; dp[k + 1] = dp[k]; }
                ip[k + 1] = j; dp[k + 1] = d;
            }
        }
    }
}

/* VOU:533-540: keep knn[i][0] iff knn[i][0].distance < ratio_thresh * knn[i][1].distance;
 * matches are APPENDED (no clear).  The reference indexes knn[i][1] unconditionally
 * (UB when the train set has < 2 rows); here such a query yields no match. */
int orc_match_knn2_ratio(const float* d1, int n1, const float* d2, int n2, int dim, float ratio_thresh,
                         orc_dmatch* out, int cap, int* m)
{
    int overflow = 0;
    for (int i = 0; i < n1; i++) {
        int idx[2]; float dist[2];
        orc_knn2(d1 + (size_t)i * dim, 1, d2, n2, dim, idx, dist);
        if (idx[0] < 0 || idx[1] < 0) continue;
        if (dist[0] < ratio_thresh * dist[1]) {
            if (*m < cap) { out[*m].queryIdx = i; out[*m].trainIdx = idx[0]; out[*m].imgIdx = 0; out[*m].distance = dist[0]; (*m)++; }
            else overflow = 1;
        }
    }
    return overflow ? -1 : 0;
}

/* ---- VOU:520-524: the AKAZE / ORB branch, BFMatcher(NORM_HAMMING).knnMatch(k = 2).  [UPSTREAM] batchDistHamming: the number of
 * differing bits over `bytes` bytes, CV_32S, reported in DMatch::distance as float; the same K = 2 insertion as above (strict <, so
 * of equal distances the lower train index stays first). */
static int hamming_bytes(const uint8_t* a, const uint8_t* b, int n)
{
    int d = 0;
    for (int i = 0; i < n; i++) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
    return d;
}
void orc_knn2_hamming(const uint8_t* d1, int n1, const uint8_t* d2, int n2, int bytes, int* idx, float* dist)
{
    for (int i = 0; i < n1; i++) {
        int bd[2] = { 0x7FFFFFFF, 0x7FFFFFFF }; int* ip = idx + 2 * i;
        ip[0] = ip[1] = -1;
        for (int j = 0; j < n2; j++) {
            const int d = hamming_bytes(d1 + (size_t)i * bytes, d2 + (size_t)j * bytes, bytes);
            if (d < bd[1]) {
                int k;
                for (k = 0; k >= 0 && bd[k] > d; k--) { ip[k + 1] = ip[k]; bd[k + 1] = bd[k]; }
                ip[k + 1] = j; bd[k + 1] = d;
            }
        }
        dist[2 * i] = ip[0] >= 0 ? (float)bd[0] : FLT_MAX; dist[2 * i + 1] = ip[1] >= 0 ? (float)bd[1] : FLT_MAX;
    }
}
int orc_match_knn2_ratio_hamming(const uint8_t* d1, int n1, const uint8_t* d2, int n2, int bytes, float ratio_thresh,
                                 orc_dmatch* out, int cap, int* m)
{
    int overflow = 0;
    for (int i = 0; i < n1; i++) {
        int idx[2]; float dist[2];
        orc_knn2_hamming(d1 + (size_t)i * bytes, 1, d2, n2, bytes, idx, dist);
        if (idx[0] < 0 || idx[1] < 0) continue;
        if (dist[0] < ratio_thresh * dist[1]) {
            if (*m < cap) { out[*m].queryIdx = i; out[*m].trainIdx = idx[0]; out[*m].imgIdx = 0; out[*m].distance = dist[0]; (*m)++; }
            else overflow = 1;
        }
    }
    return overflow ? -1 : 0;
}
