// preproc.hip -- get_image on gfx950 (VO_utility.cpp:337-379): the per-frame preprocessing in
// front of the hot path, so that the colour frame is uploaded once and the grey, undistorted, equalised image the
// detector consumes never leaves HBM:
//     cv::resize(INTER_AREA) -> cv::cvtColor(COLOR_RGB2GRAY) -> cv::undistort -> optional cv::CLAHE::apply.
// Integer / fixed-point stages are exact by construction; the float stages keep OpenCV's operation order (area
// resize: taps accumulated in table order per source row, rows accumulated in order; CLAHE blend: the reference
// expression, no FMA contraction).  The undistortion maps depend only on the camera matrices: they are computed once
// (one thread per image row, because the reference advances its running sums by one addition per column) and cached.
//   k_resize_area_c3   : one thread per destination element and channel
//   k_rgb2gray         : (R*9798 + G*19235 + B*3735 + 2^14) >> 15
//   k_undistort_map    : initUndistortRectifyMap per stripe of undistort (CV_16SC2 + 5+5 fraction bits)
//   k_remap_bilinear   : remap INTER_LINEAR, BORDER_CONSTANT 0, 15-bit weights
//   k_clahe_lut        : one workgroup per tile: histogram, clip + redistribute, cumulative LUT
//   k_clahe_apply      : bilinear blend of the four neighbouring tile LUTs
#include "uvo_ctx.h"
#include "uvo_math.h"
#include <string.h>
#include <math.h>

namespace uvo {

struct PreWs {
    uint8_t *rgb = nullptr, *small = nullptr, *gray = nullptr, *und = nullptr, *out = nullptr;
    int16_t* map1 = nullptr; uint16_t* map2 = nullptr; uint8_t* lut = nullptr;
    size_t cap_in = 0, cap_out = 0;
    double mapK[9], mapD[4], mapN[9]; int map_w = 0, map_h = 0; bool map_valid = false;
};

// computeResizeAreaTab for one destination index (same as the descriptor stage's table)
struct PTab { int sx1, sx2; float a_first, a_mid, a_last; int has_first, has_last; };
__device__ __forceinline__ PTab p_area_tab(int dx, int ssize, double scale)
{
    PTab t;
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
__device__ __forceinline__ uint8_t p_sat_u8(float v) { int iv = cv_round_f(v); return (uint8_t)(iv < 0 ? 0 : iv > 255 ? 255 : iv); }

__global__ __launch_bounds__(256) void k_resize_area_c3(const uint8_t* __restrict__ src, int sw, int sh, int stride,
                                                        uint8_t* __restrict__ dst, int dw, int dh, double scale_x, double scale_y,
                                                        int iscale_x, int iscale_y, int fast)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= dw * 3) return;
    const int dx = e / 3, ch = e - dx * 3, dy = blockIdx.y;
    if (fast) {                                          // resizeAreaFast_: integer block sums
        const uint8_t* S = src + (size_t)(dy * iscale_y) * stride + (size_t)(dx * iscale_x) * 3 + ch;
        int sum = 0;
        for (int sy = 0; sy < iscale_y; sy++) for (int sx = 0; sx < iscale_x; sx++) sum += S[(size_t)sy * stride + sx * 3];
        uint8_t r;
        if (iscale_x == 2 && iscale_y == 2) r = (uint8_t)((sum + 2) >> 2);
        else { float sc = 1.f / (iscale_x * iscale_y); r = p_sat_u8(sum * sc); }
        dst[((size_t)dy * dw + dx) * 3 + ch] = r;
        return;
    }
    const PTab tx = p_area_tab(dx, sw, scale_x), ty = p_area_tab(dy, sh, scale_y);
    const int c_begin = tx.has_first ? tx.sx1 - 1 : tx.sx1, c_end = tx.has_last ? tx.sx2 + 1 : tx.sx2;
    const int r_begin = ty.has_first ? ty.sx1 - 1 : ty.sx1, r_end = ty.has_last ? ty.sx2 + 1 : ty.sx2;
    float sum = 0.f;
    for (int r = r_begin; r < r_end; r++) {
        const uint8_t* S = src + (size_t)r * stride + ch;
        float buf = 0.f;
        for (int cc = c_begin; cc < c_end; cc++) {
            float alpha = cc < tx.sx1 ? tx.a_first : (cc < tx.sx2 ? tx.a_mid : tx.a_last);
            buf += S[cc * 3] * alpha;
        }
        float beta = r < ty.sx1 ? ty.a_first : (r < ty.sx2 ? ty.a_mid : ty.a_last);
        sum += beta * buf;
    }
    dst[((size_t)dy * dw + dx) * 3 + ch] = p_sat_u8(sum);
}

__global__ __launch_bounds__(256) void k_rgb2gray(const uint8_t* __restrict__ rgb, int w, int h, int stride, uint8_t* __restrict__ gray)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const uint8_t* s = rgb + (size_t)y * stride + 3 * x;
    gray[(size_t)y * w + x] = (uint8_t)((s[0] * 9798 + s[1] * 19235 + s[2] * 3735 + (1 << 14)) >> 15);
}

struct CamPar { double A[9], D[4], Ar[9]; };

// one thread per image row; stripes of `stripe0` rows re-derive the inverse with Ar(1,2) = v0 - stripe_start
__global__ __launch_bounds__(64) void k_undistort_map(CamPar cp, int cols, int rows, int stripe0, int16_t* __restrict__ map1, uint16_t* __restrict__ map2)
{
    const int y = blockIdx.x * 64 + threadIdx.x;
    if (y >= rows) return;
    const int ys = (y / stripe0) * stripe0, i = y - ys;
    double S[9];
    for (int k = 0; k < 9; k++) S[k] = cp.Ar[k];
    S[5] = cp.Ar[5] - ys;
    double ir[9];
    {   // cv::invert, 3x3 closed form
        double d = S[0]*(S[4]*S[8] - S[5]*S[7]) - S[1]*(S[3]*S[8] - S[5]*S[6]) + S[2]*(S[3]*S[7] - S[4]*S[6]);
        if (d == 0) { for (int k = 0; k < 9; k++) ir[k] = 0; }
        else {
            d = 1./d;
            ir[0] = (S[4]*S[8] - S[5]*S[7]) * d; ir[1] = (S[2]*S[7] - S[1]*S[8]) * d; ir[2] = (S[1]*S[5] - S[2]*S[4]) * d;
            ir[3] = (S[5]*S[6] - S[3]*S[8]) * d; ir[4] = (S[0]*S[8] - S[2]*S[6]) * d; ir[5] = (S[2]*S[3] - S[0]*S[5]) * d;
            ir[6] = (S[3]*S[7] - S[4]*S[6]) * d; ir[7] = (S[1]*S[6] - S[0]*S[7]) * d; ir[8] = (S[0]*S[4] - S[1]*S[3]) * d;
        }
    }
    const double u0 = cp.A[2], v0 = cp.A[5], fx = cp.A[0], fy = cp.A[4];
    const double k1 = cp.D[0], k2 = cp.D[1], p1 = cp.D[2], p2 = cp.D[3];
    const double k3 = 0, k4 = 0, k5 = 0, k6 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    int16_t* m1 = map1 + (size_t)y * cols * 2;
    uint16_t* m2 = map2 + (size_t)y * cols;
    double _x = i*ir[1] + ir[2], _y = i*ir[4] + ir[5], _w = i*ir[7] + ir[8];
    for (int j = 0; j < cols; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
        double w = 1./_w, x = _x*w, yy = _y*w;
        double x2 = x*x, y2 = yy*yy;
        double r2 = x2 + y2, _2xy = 2*x*yy;
        double kr = (1 + ((k3*r2 + k2)*r2 + k1)*r2)/(1 + ((k6*r2 + k5)*r2 + k4)*r2);
        double xd = (x*kr + p1*_2xy + p2*(r2 + 2*x2) + s1*r2 + s2*r2*r2);
        double yd = (yy*kr + p1*(r2 + 2*y2) + p2*_2xy + s3*r2 + s4*r2*r2);
        double invProj = 1.0;
        double u = fx*invProj*xd + u0;
        double v = fy*invProj*yd + v0;
        int iu = cv_round_d(u*32), iv = cv_round_d(v*32);
        m1[j*2] = (int16_t)(iu >> 5); m1[j*2 + 1] = (int16_t)(iv >> 5);
        m2[j] = (uint16_t)((iv & 31)*32 + (iu & 31));
    }
}

__global__ __launch_bounds__(256) void k_remap_bilinear(const uint8_t* __restrict__ src, int sw, int sh, const int16_t* __restrict__ map1,
                                                        const uint16_t* __restrict__ map2, uint8_t* __restrict__ dst, int dw, int dh)
{
    const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
    if (dx >= dw) return;
    const size_t o = (size_t)dy * dw + dx;
    const int sx = map1[o * 2], sy = map1[o * 2 + 1];
    const int f = map2[o], fxi = f & 31, fyi = f >> 5;
    const int w00 = (32 - fxi) * (32 - fyi) * 32, w01 = fxi * (32 - fyi) * 32, w10 = (32 - fxi) * fyi * 32, w11 = fxi * fyi * 32;
    int v00 = 0, v01 = 0, v10 = 0, v11 = 0;
    if (sy >= 0 && sy < sh) { if (sx >= 0 && sx < sw) v00 = src[(size_t)sy * sw + sx]; if (sx + 1 >= 0 && sx + 1 < sw) v01 = src[(size_t)sy * sw + sx + 1]; }
    if (sy + 1 >= 0 && sy + 1 < sh) { if (sx >= 0 && sx < sw) v10 = src[(size_t)(sy + 1) * sw + sx]; if (sx + 1 >= 0 && sx + 1 < sw) v11 = src[(size_t)(sy + 1) * sw + sx + 1]; }
    int val = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
    dst[o] = (uint8_t)(val < 0 ? 0 : val > 255 ? 255 : val);
}

__device__ __forceinline__ int p_reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * (len - 1) - p; }
    return p;
}

// one workgroup per tile (8 x 8 tiles); the image is extended by BORDER_REFLECT_101 when it does not divide
__global__ __launch_bounds__(256) void k_clahe_lut(const uint8_t* __restrict__ src, int w, int h, int tw, int th, int clipLimit, float lutScale,
                                                   uint8_t* __restrict__ lut)
{
    const int k = blockIdx.x, ty = k / 8, tx = k % 8, tid = threadIdx.x;
    __shared__ int hist[256];
    __shared__ int s_red[256];
    hist[tid] = 0;
    __syncthreads();
    for (int e = tid; e < tw * th; e += 256) {
        int y = e / tw, x = e - y * tw;
        int gy = p_reflect101(ty * th + y, h), gx = p_reflect101(tx * tw + x, w);
        atomicAdd(&hist[src[(size_t)gy * w + gx]], 1);
    }
    __syncthreads();
    int hv = hist[tid];
    if (clipLimit > 0) {
        int over = hv > clipLimit ? hv - clipLimit : 0;
        if (hv > clipLimit) hv = clipLimit;
        s_red[tid] = over;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o) s_red[tid] += s_red[tid + o]; __syncthreads(); }
        const int clipped = s_red[0];
        const int redistBatch = clipped / 256;
        int residual = clipped - redistBatch * 256;
        hv += redistBatch;
        if (residual != 0) {
            int step = 256 / residual; if (step < 1) step = 1;
            if (tid % step == 0 && tid / step < residual) hv++;          // for (i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++
        }
        __syncthreads();
    }
    // inclusive prefix sum (integers: any order)
    s_red[tid] = hv;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        int t = tid >= o ? s_red[tid - o] : 0;
        __syncthreads();
        s_red[tid] += t;
        __syncthreads();
    }
    lut[(size_t)k * 256 + tid] = p_sat_u8(s_red[tid] * lutScale);
}

__global__ __launch_bounds__(256) void k_clahe_apply(const uint8_t* __restrict__ src, int w, int h, int tw, int th, const uint8_t* __restrict__ lut,
                                                     uint8_t* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    float tyf = y * inv_th - 0.5f;
    int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
    float ya = tyf - ty1, ya1 = 1.0f - ya;
    ty1 = ty1 > 0 ? ty1 : 0; ty2 = ty2 < 7 ? ty2 : 7;
    float txf = x * inv_tw - 0.5f;
    int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
    float xa = txf - tx1, xa1 = 1.0f - xa;
    tx1 = tx1 > 0 ? tx1 : 0; tx2 = tx2 < 7 ? tx2 : 7;
    const int sv = src[(size_t)y * w + x];
    const uint8_t* p1 = lut + (size_t)ty1 * 8 * 256;
    const uint8_t* p2 = lut + (size_t)ty2 * 8 * 256;
    const int ind1 = tx1 * 256 + sv, ind2 = tx2 * 256 + sv;
    float res = (p1[ind1] * xa1 + p1[ind2] * xa) * ya1 + (p2[ind1] * xa1 + p2[ind2] * xa) * ya;
    dst[(size_t)y * w + x] = p_sat_u8(res);
}

void pre_ws_free(Ctx* c)
{
    PreWs* p = static_cast<PreWs*>(c->pre_ws);
    if (!p) return;
    void* ptrs[] = { p->rgb, p->small, p->gray, p->und, p->out, p->map1, p->map2, p->lut };
    for (void* q : ptrs) (void)hipFree(q);
    delete p;
    c->pre_ws = nullptr;
}

// get_image: result in ws->out (device), dims returned
uvo_status pre_get_image(Ctx* c, const uint8_t* rgb, int w, int h, int stride, int mem, const double* K, const double* dist4, const double* newK,
                         int desired_width, int clahe_on, int clip_limit, const uint8_t** d_out, int* out_w, int* out_h)
{
    if (w <= 0 || h <= 0 || desired_width <= 0 || stride < 3 * w) { c->err = "get_image: bad geometry"; return UVO_INVALID_ARG; }
    const double ratio = (double)w / (double)desired_width;
    const int dw = desired_width, dh = (int)(h / ratio);
    if (dw > w || dh > h || dh <= 0) { c->err = "get_image: enlarging (OpenCV switches INTER_AREA to bilinear there) is not provided"; return UVO_INVALID_ARG; }
    hipStream_t st = c->stream;
    PreWs* p = static_cast<PreWs*>(c->pre_ws);
    if (!p) { p = new PreWs(); c->pre_ws = p; }
    const size_t n_in = (size_t)h * stride, n_out = (size_t)dw * dh;
    if (n_in > p->cap_in) {
        (void)hipFree(p->rgb); p->rgb = nullptr;
        UVO_HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&p->rgb), n_in));
        p->cap_in = n_in;
    }
    if (n_out > p->cap_out) {
        void* old[] = { p->small, p->gray, p->und, p->out, p->map1, p->map2 };
        for (void* q : old) (void)hipFree(q);
        p->small = p->gray = p->und = p->out = nullptr; p->map1 = nullptr; p->map2 = nullptr; p->map_valid = false;
        UVO_HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&p->small), n_out * 3));
        UVO_HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&p->gray), n_out));
        UVO_HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&p->und), n_out));
        UVO_HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&p->out), n_out));
        UVO_HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&p->map1), n_out * 2 * sizeof(int16_t)));
        UVO_HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&p->map2), n_out * sizeof(uint16_t)));
        p->cap_out = n_out;
    }
    if (!p->lut) UVO_HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&p->lut), 64 * 256));
    const uint8_t* d_rgb = rgb;
    int d_stride = stride;
    if (mem != UVO_MEM_DEVICE) {
        UVO_HIP_TRY(c, hipMemcpyAsync(p->rgb, rgb, n_in, hipMemcpyHostToDevice, st));
        d_rgb = p->rgb;
    }
    if (w == dw && h == dh) {
        hipLaunchKernelGGL(k_rgb2gray, dim3((dw + 255) / 256, dh), dim3(256), 0, st, d_rgb, dw, dh, d_stride, p->gray);
    } else {
        const double inv_scale_x = (double)dw / w, inv_scale_y = (double)dh / h;
        const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
        const int iscale_x = cv_round_d(scale_x), iscale_y = cv_round_d(scale_y);
        const int fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
        hipLaunchKernelGGL(k_resize_area_c3, dim3((dw * 3 + 255) / 256, dh), dim3(256), 0, st, d_rgb, w, h, d_stride, p->small, dw, dh,
                           scale_x, scale_y, iscale_x, iscale_y, fast);
        hipLaunchKernelGGL(k_rgb2gray, dim3((dw + 255) / 256, dh), dim3(256), 0, st, p->small, dw, dh, dw * 3, p->gray);
    }
    if (!p->map_valid || p->map_w != dw || p->map_h != dh || memcmp(p->mapK, K, sizeof(p->mapK)) || memcmp(p->mapD, dist4, sizeof(p->mapD)) ||
        memcmp(p->mapN, newK, sizeof(p->mapN))) {
        CamPar cp;
        memcpy(cp.A, K, sizeof(cp.A)); memcpy(cp.D, dist4, sizeof(cp.D)); memcpy(cp.Ar, newK, sizeof(cp.Ar));
        int stripe0 = (1 << 12) / (dw > 1 ? dw : 1);
        if (stripe0 < 1) stripe0 = 1;
        if (stripe0 > dh) stripe0 = dh;
        hipLaunchKernelGGL(k_undistort_map, dim3((dh + 63) / 64), dim3(64), 0, st, cp, dw, dh, stripe0, p->map1, p->map2);
        memcpy(p->mapK, K, sizeof(p->mapK)); memcpy(p->mapD, dist4, sizeof(p->mapD)); memcpy(p->mapN, newK, sizeof(p->mapN));
        p->map_w = dw; p->map_h = dh; p->map_valid = true;
    }
    hipLaunchKernelGGL(k_remap_bilinear, dim3((dw + 255) / 256, dh), dim3(256), 0, st, p->gray, dw, dh, p->map1, p->map2, p->und, dw, dh);
    const uint8_t* result = p->und;
    if (clahe_on) {
        // clahe.cpp: if either side does not divide into 8 tiles BOTH are extended by 8 - (size % 8) (reflect-101) -- that is
        // a full 8 on the side that did divide
        const bool ext = dw % 8 != 0 || dh % 8 != 0;
        const int ew = ext ? dw + (8 - dw % 8) : dw, eh = ext ? dh + (8 - dh % 8) : dh;
        const int tw = ew / 8, th = eh / 8, total = tw * th;
        const float lutScale = (float)255 / total;
        int clipLimit = 0;
        if ((double)clip_limit > 0.0) { clipLimit = (int)((double)clip_limit * total / 256); if (clipLimit < 1) clipLimit = 1; }
        hipLaunchKernelGGL(k_clahe_lut, dim3(64), dim3(256), 0, st, p->und, dw, dh, tw, th, clipLimit, lutScale, p->lut);
        hipLaunchKernelGGL(k_clahe_apply, dim3((dw + 255) / 256, dh), dim3(256), 0, st, p->und, dw, dh, tw, th, p->lut, p->out);
        result = p->out;
    }
    UVO_HIP_TRY(c, hipGetLastError());
    *d_out = result; *out_w = dw; *out_h = dh;
    return UVO_OK;
}

}  // namespace uvo
