// codec.hip -- compressed-image ingest: what from_ros_to_cv_image does before get_image
// (uvo_libraries/src/math_utility.cpp:154-173): cv_bridge::toCvCopy(CompressedImage) = cv::imdecode -> libjpeg, then
// cv::cvtColor(COLOR_BayerBGGR2BGR) when the format string says "bayer".
//
// A JPEG bit stream is sequential by construction (variable-length codes, DC prediction), so the entropy decoding runs on
// the host -- one pass over the compressed bytes into a pinned coefficient buffer -- and everything that is per-block or
// per-pixel runs on the device, HBM-bound byte work with coalesced accesses:
//   k_jpeg_idct    one thread per 8 x 8 block: dequantisation + libjpeg's jpeg_idct_islow (JDCT_ISLOW, libjpeg's and
//                  cv::imdecode's default), the block's 64 samples written to the component plane
//   k_jpeg_colour  one thread per pixel: libjpeg's fancy (triangle) chroma upsampling evaluated at the pixel from the
//                  neighbouring chroma samples + the YCbCr -> RGB fixed-point conversion, B G R interleaved out
//   k_bayer_bggr   COLOR_BayerBGGR2BGR, bilinear, OpenCV's border rule
// Results are byte-identical to libjpeg-turbo's (tests/test_codec.py: fixtures decoded by Pillow's libjpeg-turbo).
// Baseline / extended-sequential Huffman JPEG, 8 bit, 1 or 3 components, sampling factors 1 or 2, restart markers;
// progressive and arithmetic-coded files are refused with UVO_INVALID_ARG (cv::imdecode would decode them on the CPU).
// PNG payloads (round 3; sniffed by signature as cv::imdecode does): zlib inflate and the scanline filters on the host
// (uvo_png.h), k_png_expand on the device: 1 / 2 / 4 / 8-bit grey, palette and RGB(A) samples -> grey / B G R (A) bytes.
#include "uvo_ctx.h"
#include "uvo_png.h"
#include <string.h>
#include <vector>

namespace uvo {

// ------------------------------------------------------------------------------------------ host: headers + entropy decoding
namespace {

const uint8_t kNatural[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                               35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

// canonical Huffman table with a 9-bit first-level lookup (code length + symbol), longer codes by the length walk
struct Huff {
    uint8_t bits[17] = {0}, vals[256] = {0};
    int mincode[18], maxcode[18], valptr[18];
    uint16_t fast[512];                                       // (length << 8) | symbol, 0 = longer than 9 bits
    bool present = false;
    // false: the code lengths do not form a prefix code (more codes of a length than remain): jdhuff.c's JERR_BAD_HUFF_TABLE
    bool build()
    {
        int code = 0, k = 0;
        memset(fast, 0, sizeof(fast));
        present = false;
        for (int l = 1; l <= 16; l++) {
            if (code + bits[l] > (1 << l)) return false;
            valptr[l] = k; mincode[l] = code;
            for (int i = 0; i < bits[l]; i++, k++, code++)
                if (l <= 9) { const int lo = code << (9 - l); for (int f = 0; f < (1 << (9 - l)); f++) fast[lo + f] = (uint16_t)((l << 8) | vals[k]); }
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        present = true;
        return true;
    }
    bool dc_symbols_ok() const { int cnt = 0; for (int l = 1; l <= 16; l++) cnt += bits[l]; for (int i = 0; i < cnt; i++) if (vals[i] > 15) return false; return true; }
};

struct BitReader {
    const uint8_t* p; const uint8_t* end; uint64_t acc = 0; int nbits = 0; int marker = 0;
    void fill()
    {
        while (nbits <= 48) {
            int c = 0;
            if (!marker && p < end) {
                c = *p++;
                if (c == 0xFF) {
                    const int c2 = p < end ? *p : 0xD9;
                    if (c2 == 0) p++;
                    else { marker = c2; p--; c = 0; }        // a marker ends the segment: zeros from here on (as jdhuff.c)
                }
            }
            acc |= (uint64_t)c << (56 - nbits);
            nbits += 8;
        }
    }
    inline int peek(int n) { if (nbits < n) fill(); return (int)(acc >> (64 - n)); }
    inline void skip(int n) { acc <<= n; nbits -= n; }
    inline int get(int n) { if (n == 0) return 0; const int v = peek(n); skip(n); return v; }
    inline int decode(const Huff& h)
    {
        if (nbits < 16) fill();
        const uint16_t f = h.fast[acc >> 55];
        if (f) { skip(f >> 8); return f & 255; }
        int code = (int)(acc >> 54), l = 10;                 // the first ten bits
        for (; l <= 16; l++) {
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) { skip(l); return h.vals[h.valptr[l] + code - h.mincode[l]]; }
            code = (int)(acc >> (63 - l));
        }
        skip(16);
        return 0;                                            // corrupt code: libjpeg warns and goes on with zero
    }
};
inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

struct Comp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, bw = 0, bh = 0; size_t coef_off = 0; int dc_pred = 0; };
struct Jpeg {
    int w = 0, h = 0, ncomp = 0, hmax = 1, vmax = 1, restart = 0, mcux = 0, mcuy = 0;
    uint16_t quant[4][64];
    Huff dc[4], ac[4];
    Comp comp[3];
    size_t total_blocks = 0;
};
inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

}  // namespace

struct CodecWs {
    int16_t* h_coef = nullptr; int16_t* d_coef = nullptr; size_t coef_cap = 0, coef_cap_dev = 0;      // blocks x 64, pinned / device
    uint8_t* d_planes = nullptr; size_t planes_cap = 0;                                // the three component planes (block-padded)
    uint8_t* d_out = nullptr; size_t out_cap = 0;                                      // decoded image (and the Bayer source)
    uint8_t* d_out2 = nullptr; size_t out2_cap = 0;                                    // demosaiced image
    uint16_t* d_quant = nullptr;                                                       // 4 x 64
};

void codec_ws_free(Ctx* c)
{
    CodecWs* w = static_cast<CodecWs*>(c->codec_ws);
    if (!w) return;
    (void)hipHostFree(w->h_coef); (void)hipFree(w->d_coef); (void)hipFree(w->d_planes); (void)hipFree(w->d_out); (void)hipFree(w->d_out2); (void)hipFree(w->d_quant);
    delete w;
    c->codec_ws = nullptr;
}

static uvo_status grow(Ctx* c, void** p, size_t* cap, size_t need, bool pinned)
{
    if (*cap >= need) return UVO_OK;
    if (*p) { if (pinned) (void)hipHostFree(*p); else (void)hipFree(*p); *p = nullptr; *cap = 0; }
    const size_t n = need + need / 4 + 4096;
    UVO_HIP_TRY(c, pinned ? hipHostMalloc(p, n) : hipMalloc(p, n));
    *cap = n;
    return UVO_OK;
}

// Parses the headers and decodes every coefficient into ws->h_coef (natural order, component after component,
// blocks in raster order of the MCU-padded component).  Host only.
// headers_only: stop at the first scan header, after every check of the frame, table and scan headers (the size query of
// uvo_decode_image): nothing is allocated and no coefficient is decoded.
static uvo_status jpeg_entropy_decode(Ctx* c, CodecWs* ws, const uint8_t* data, size_t n, Jpeg* j, bool headers_only = false)
{
    auto bad = [&](const char* m) { c->err = std::string("JPEG: ") + m; return UVO_INVALID_ARG; };
    if (n < 4 || data[0] != 0xFF || data[1] != 0xD8) return bad("not a JPEG stream (no SOI)");
    size_t pos = 2;
    bool have_sof = false;
    while (pos + 4 <= n) {
        if (data[pos] != 0xFF) { pos++; continue; }
        const int m = data[pos + 1];
        if (m == 0xFF) { pos++; continue; }
        pos += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;
        if (pos + 2 > n) return bad("truncated");
        const int len = be16(data + pos);
        if (len < 2 || pos + len > n) return bad("truncated segment");
        const uint8_t* s = data + pos + 2; const int sl = len - 2;
        switch (m) {
        case 0xDB:
            for (int o = 0; o < sl;) {
                const int pq = s[o] >> 4, tq = s[o] & 15; o++;
                if (tq > 3 || o + (pq ? 128 : 64) > sl) return bad("bad quantisation table");
                for (int k = 0; k < 64; k++) j->quant[tq][kNatural[k]] = (uint16_t)(pq ? be16(s + o + 2 * k) : s[o + k]);
                o += pq ? 128 : 64;
            }
            break;
        case 0xC4:
            for (int o = 0; o + 17 <= sl;) {
                const int tc = s[o] >> 4, th = s[o] & 15; o++;
                if (th > 3 || tc > 1) return bad("bad Huffman table id");
                Huff& h = tc ? j->ac[th] : j->dc[th];
                int cnt = 0;
                for (int l = 1; l <= 16; l++) { h.bits[l] = s[o + l - 1]; cnt += h.bits[l]; }
                o += 16;
                if (cnt > 256 || o + cnt > sl) return bad("bad Huffman table");
                memcpy(h.vals, s + o, (size_t)cnt); o += cnt;
                if (!h.build()) return bad("bad Huffman table (not a prefix code)");
            }
            break;
        case 0xC0: case 0xC1: {
            if (have_sof) return bad("second frame header");
            if (sl < 6 || s[0] != 8) return bad("only 8-bit samples are supported");
            j->h = be16(s + 1); j->w = be16(s + 3); j->ncomp = s[5];
            if (j->w <= 0 || j->h <= 0) return bad("empty image");
            if ((long long)j->w * j->h > (1LL << 26)) return bad("images above 64 Mpixel are refused (a damaged header must not ask for gigabytes)");
            if (j->ncomp != 1 && j->ncomp != 3) return bad("only 1 or 3 components are supported");
            if (sl < 6 + 3 * j->ncomp) return bad("truncated frame header");
            j->hmax = j->vmax = 1;
            for (int k = 0; k < j->ncomp; k++) {
                Comp& q = j->comp[k];
                q.id = s[6 + 3 * k]; q.h = s[7 + 3 * k] >> 4; q.v = s[7 + 3 * k] & 15; q.tq = s[8 + 3 * k];
                if (q.h < 1 || q.h > 2 || q.v < 1 || q.v > 2 || q.tq > 3) return bad("sampling factors above 2 are not supported");
                j->hmax = q.h > j->hmax ? q.h : j->hmax; j->vmax = q.v > j->vmax ? q.v : j->vmax;
            }
            if (j->ncomp == 1) { j->comp[0].h = j->comp[0].v = 1; j->hmax = j->vmax = 1; }
            j->mcux = (j->w + 8 * j->hmax - 1) / (8 * j->hmax); j->mcuy = (j->h + 8 * j->vmax - 1) / (8 * j->vmax);
            size_t off = 0;
            for (int k = 0; k < j->ncomp; k++) {
                Comp& q = j->comp[k];
                q.bw = j->mcux * q.h; q.bh = j->mcuy * q.v; q.coef_off = off;
                off += (size_t)q.bw * q.bh;
            }
            j->total_blocks = off;
            if (!headers_only) {
                UVO_TRY(grow(c, reinterpret_cast<void**>(&ws->h_coef), &ws->coef_cap, off * 64 * sizeof(int16_t), true));
                memset(ws->h_coef, 0, off * 64 * sizeof(int16_t));
            }
            have_sof = true;
            break;
        }
        case 0xDD: if (sl < 2) return bad("truncated restart interval"); j->restart = be16(s); break;
        case 0xDA: {
            if (!have_sof) return bad("scan before frame header");
            if (sl < 1 || s[0] != j->ncomp) return bad("non-interleaved scans are not supported");
            if (sl < 1 + 2 * j->ncomp + 3) return bad("truncated scan header");
            for (int i = 0; i < s[0]; i++)
                for (int k = 0; k < j->ncomp; k++)
                    if (j->comp[k].id == s[1 + 2 * i]) { j->comp[k].td = s[2 + 2 * i] >> 4; j->comp[k].ta = s[2 + 2 * i] & 15; }
            for (int k = 0; k < j->ncomp; k++) if (j->comp[k].td > 3 || j->comp[k].ta > 3 || !j->dc[j->comp[k].td].present || !j->ac[j->comp[k].ta].present) return bad("scan refers to a missing Huffman table");
            for (int k = 0; k < j->ncomp; k++) if (!j->dc[j->comp[k].td].dc_symbols_ok()) return bad("bad DC Huffman table (category above 15)");
            if (headers_only) return UVO_OK;
            BitReader b; b.p = data + pos + len; b.end = data + n;
            int left = j->restart;
            for (int my = 0; my < j->mcuy; my++)
                for (int mx = 0; mx < j->mcux; mx++) {
                    if (j->restart && left == 0) {           // RSTn: drop the partial byte, skip the marker, reset the predictors
                        b.acc = 0; b.nbits = 0;
                        if (b.marker >= 0xD0 && b.marker <= 0xD7) { b.p += 2; b.marker = 0; }
                        else { while (b.p + 1 < b.end && !(b.p[0] == 0xFF && b.p[1] >= 0xD0 && b.p[1] <= 0xD7)) b.p++; if (b.p + 1 < b.end) b.p += 2; b.marker = 0; }
                        for (int k = 0; k < j->ncomp; k++) j->comp[k].dc_pred = 0;
                        left = j->restart;
                    }
                    for (int k = 0; k < j->ncomp; k++) {
                        Comp& q = j->comp[k];
                        const Huff& hd = j->dc[q.td]; const Huff& ha = j->ac[q.ta];
                        for (int by = 0; by < q.v; by++)
                            for (int bx = 0; bx < q.h; bx++) {
                                int16_t* blk = ws->h_coef + (q.coef_off + (size_t)(my * q.v + by) * q.bw + (mx * q.h + bx)) * 64;
                                const int t = b.decode(hd);
                                q.dc_pred = (int)((unsigned)q.dc_pred + (unsigned)(t ? extend(b.get(t), t) : 0));   // (wraps on damaged streams)
                                blk[0] = (int16_t)q.dc_pred;
                                for (int kk = 1; kk < 64;) {
                                    const int rs = b.decode(ha), r = rs >> 4, sz = rs & 15;
                                    if (sz == 0) { if (r == 15) { kk += 16; continue; } break; }
                                    kk += r;
                                    if (kk > 63) break;
                                    blk[kNatural[kk]] = (int16_t)extend(b.get(sz), sz);
                                    kk++;
                                }
                            }
                    }
                    if (j->restart) left--;
                }
            return UVO_OK;
        }
        default:
            if (m == 0xC2) return bad("progressive JPEG is not supported (baseline / sequential Huffman only)");
            if (m >= 0xC3 && m <= 0xCF && m != 0xC8) return bad("lossless / arithmetic-coded JPEG is not supported");
            break;
        }
        pos += len;
    }
    return bad("no scan found");
}

// ------------------------------------------------------------------------------------------ device
struct IdctComp { size_t coef_off; int bw, bh, tq; size_t plane_off; };
struct IdctArgs { IdctComp comp[3]; int ncomp; size_t total_blocks; };

#define UVO_DESCALE(x, n) ((int)((unsigned)(x) + (1u << ((n) - 1))) >> (n))
__device__ __forceinline__ uint8_t jpeg_range_limit(int x)
{   // libjpeg's post-IDCT range_limit table read at x & 1023 (jdmaster.c prepare_range_limit_table): [0,128) -> v + 128,
    // [128,512) -> 255, [512,896) -> 0, [896,1024) -> v - 896: clamp(x + 128) for |x| < 512, the table's wrap beyond
    const int v = x & 1023;
    return (uint8_t)(v < 128 ? v + 128 : (v < 512 ? 255 : (v < 896 ? 0 : v - 896)));
}
// jidctint.c's butterfly on eight values (in place); both passes use it with different descaling
// (every add / multiply modulo 2^32: a damaged stream with out-of-range coefficients gives some picture, never undefined behaviour)
__device__ __forceinline__ void islow_1d(const int (&in_s)[8], int (&o)[8])
{
    unsigned in[8];
#pragma unroll
    for (int k = 0; k < 8; k++) in[k] = (unsigned)in_s[k];
    unsigned z2 = in[2], z3 = in[6];
    unsigned z1 = (z2 + z3) * 4433u;
    unsigned tmp2 = z1 + z3 * (unsigned)(-15137), tmp3 = z1 + z2 * 6270u;
    unsigned tmp0 = (in[0] + in[4]) * 8192u, tmp1 = (in[0] - in[4]) * 8192u;
    const unsigned tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; unsigned z4 = tmp1 + tmp3;
    const unsigned z5 = (z3 + z4) * 9633u;
    tmp0 *= 2446u; tmp1 *= 16819u; tmp2 *= 25172u; tmp3 *= 12299u;
    z1 *= (unsigned)(-7373); z2 *= (unsigned)(-20995); z3 *= (unsigned)(-16069); z4 *= (unsigned)(-3196);
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    o[0] = (int)(tmp10 + tmp3); o[7] = (int)(tmp10 - tmp3); o[1] = (int)(tmp11 + tmp2); o[6] = (int)(tmp11 - tmp2);
    o[2] = (int)(tmp12 + tmp1); o[5] = (int)(tmp12 - tmp1); o[3] = (int)(tmp13 + tmp0); o[4] = (int)(tmp13 - tmp0);
}

// Eight threads per block, one column then one row each; the block's workspace lives in LDS ([block in workgroup][8][9] ints)
__global__ __launch_bounds__(256) void k_jpeg_idct(IdctArgs a, const int16_t* __restrict__ coef, const uint16_t* __restrict__ quant, uint8_t* __restrict__ planes)
{
    __shared__ int ws[32][8][9];
    const int lb = threadIdx.x >> 3, t = threadIdx.x & 7;
    const size_t blk = (size_t)blockIdx.x * 32 + lb;
    const bool live = blk < a.total_blocks;
    int ci = 0;
    if (live) { if (a.ncomp == 3 && blk >= a.comp[1].coef_off) ci = blk >= a.comp[2].coef_off ? 2 : 1; }
    const IdctComp cp = a.comp[ci];
    if (live) {
        const int16_t* src = coef + blk * 64;
        const uint16_t* q = quant + cp.tq * 64;
        int in[8], o[8];
#pragma unroll
        for (int r = 0; r < 8; r++) in[r] = (int)src[r * 8 + t] * (int)q[r * 8 + t];           // column t, dequantised
        islow_1d(in, o);
#pragma unroll
        for (int r = 0; r < 8; r++) ws[lb][r][t] = UVO_DESCALE(o[r], 11);                       // CONST_BITS - PASS1_BITS
    }
    __syncthreads();
    if (live) {
        int in[8], o[8];
#pragma unroll
        for (int k = 0; k < 8; k++) in[k] = ws[lb][t][k];                                       // row t
        islow_1d(in, o);
        const size_t local = blk - cp.coef_off;
        const int by = (int)(local / cp.bw), bx = (int)(local - (size_t)by * cp.bw);
        uint8_t* dst = planes + cp.plane_off + ((size_t)by * 8 + t) * ((size_t)cp.bw * 8) + (size_t)bx * 8;
        uint2 pk;
        pk.x = jpeg_range_limit(UVO_DESCALE(o[0], 18)) | (jpeg_range_limit(UVO_DESCALE(o[1], 18)) << 8) | (jpeg_range_limit(UVO_DESCALE(o[2], 18)) << 16) | ((unsigned)jpeg_range_limit(UVO_DESCALE(o[3], 18)) << 24);
        pk.y = jpeg_range_limit(UVO_DESCALE(o[4], 18)) | (jpeg_range_limit(UVO_DESCALE(o[5], 18)) << 8) | (jpeg_range_limit(UVO_DESCALE(o[6], 18)) << 16) | ((unsigned)jpeg_range_limit(UVO_DESCALE(o[7], 18)) << 24);
        *reinterpret_cast<uint2*>(dst) = pk;                                                    // 8 samples of row t (8-byte aligned)
    }
}

struct ColourArgs { size_t plane_off[3]; int pw[3]; int dw[3], dh[3]; int hs[3], vs[3]; int ncomp, w, h; };

// libjpeg's fancy upsampling evaluated at one output position (jdsample.c h2v1 / h2v2 / h1v2), replication for 1 x 1
__device__ __forceinline__ int chroma_at(const uint8_t* __restrict__ p, int pw, int dw, int dh, int hs, int vs, int x, int y)
{
    if (hs == 1 && vs == 1) return p[(size_t)y * pw + x];
    const int sy = y / vs, sx = x / hs;
    const uint8_t* r0 = p + (size_t)sy * pw;
    if (vs == 1) {                                            // h2v1: 3/4 nearer + 1/4 farther, bias 1 (even) / 2 (odd); the ends copy
        if ((x & 1) == 0) return sx == 0 ? r0[0] : (3 * r0[sx] + r0[sx - 1] + 1) >> 2;
        return sx == dw - 1 ? r0[sx] : (3 * r0[sx] + r0[sx + 1] + 2) >> 2;
    }
    int oy = (y & 1) ? sy + 1 : sy - 1;
    oy = oy < 0 ? 0 : (oy > dh - 1 ? dh - 1 : oy);
    const uint8_t* r1 = p + (size_t)oy * pw;
    if (hs == 1) return (3 * r0[x] + r1[x] + ((y & 1) ? 2 : 1)) >> 2;                           // h1v2
    const int thiscol = 3 * r0[sx] + r1[sx];                  // h2v2
    if ((x & 1) == 0) return sx == 0 ? (thiscol * 4 + 8) >> 4 : (thiscol * 3 + 3 * r0[sx - 1] + r1[sx - 1] + 8) >> 4;
    return sx == dw - 1 ? (thiscol * 4 + 7) >> 4 : (thiscol * 3 + 3 * r0[sx + 1] + r1[sx + 1] + 7) >> 4;
}

__global__ __launch_bounds__(256) void k_jpeg_colour(ColourArgs a, const uint8_t* __restrict__ planes, uint8_t* __restrict__ out)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.w || y >= a.h) return;
    const int Y = chroma_at(planes + a.plane_off[0], a.pw[0], a.dw[0], a.dh[0], a.hs[0], a.vs[0], x, y);
    if (a.ncomp == 1) { out[(size_t)y * a.w + x] = (uint8_t)Y; return; }
    const int cb = chroma_at(planes + a.plane_off[1], a.pw[1], a.dw[1], a.dh[1], a.hs[1], a.vs[1], x, y) - 128;
    const int cr = chroma_at(planes + a.plane_off[2], a.pw[2], a.dw[2], a.dh[2], a.hs[2], a.vs[2], x, y) - 128;
    // jdcolor.c: Cr_r_tab, Cb_b_tab, Cb_g_tab + Cr_g_tab (16-bit fixed point, ONE_HALF folded into the Cb table)
    int r = Y + ((91881 * cr + 32768) >> 16);
    int g = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    int b = Y + ((116130 * cb + 32768) >> 16);
    r = r < 0 ? 0 : (r > 255 ? 255 : r); g = g < 0 ? 0 : (g > 255 ? 255 : g); b = b < 0 ? 0 : (b > 255 ? 255 : b);
    uint8_t* o = out + ((size_t)y * a.w + x) * 3;
    o[0] = (uint8_t)b; o[1] = (uint8_t)g; o[2] = (uint8_t)r;
}

// COLOR_BayerBGGR2BGR (= COLOR_BayerRG2BGR), bilinear ([UPSTREAM] imgproc/src/demosaicing.cpp Bayer2RGB_): interior pixels from
// their 3 x 3 neighbourhood; the first / last column copies its neighbour, then the first / last row copies its neighbour.
__global__ __launch_bounds__(256) void k_bayer_bggr(const uint8_t* __restrict__ bayer, int w, int h, int stride, uint8_t* __restrict__ dst)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    uint8_t* o = dst + ((size_t)y * w + x) * 3;
    if (w < 3 || h < 3) { o[0] = o[1] = o[2] = 0; return; }
    const int yy = y < 1 ? 1 : (y > h - 2 ? h - 2 : y), xx = x < 1 ? 1 : (x > w - 2 ? w - 2 : x);      // the border copies the nearest interior pixel
    const uint8_t* r0 = bayer + (size_t)(yy - 1) * stride; const uint8_t* r1 = r0 + stride; const uint8_t* r2 = r1 + stride;
    const bool ey = (yy & 1) == 0, ex = (xx & 1) == 0;
    int B, G, R;
    const int cross = (r0[xx] + r1[xx - 1] + r1[xx + 1] + r2[xx] + 2) >> 2, diag = (r0[xx - 1] + r0[xx + 1] + r2[xx - 1] + r2[xx + 1] + 2) >> 2;
    const int horz = (r1[xx - 1] + r1[xx + 1] + 1) >> 1, vert = (r0[xx] + r2[xx] + 1) >> 1;
    if (ey && ex) { B = r1[xx]; G = cross; R = diag; }
    else if (!ey && !ex) { R = r1[xx]; G = cross; B = diag; }
    else if (ey) { G = r1[xx]; B = horz; R = vert; }
    else { G = r1[xx]; R = horz; B = vert; }
    o[0] = (uint8_t)B; o[1] = (uint8_t)G; o[2] = (uint8_t)R;
}

// ------------------------------------------------------------------------------------------ orchestration
// PNG samples -> what cv::imdecode(IMREAD_UNCHANGED) returns: grey (sub-byte depths scaled as png_set_expand_gray_1_2_4_to_8 does,
// v * 255 / (2^d - 1)), palette -> B G R, RGB -> B G R, RGBA -> B G R A.  One thread per pixel.
struct PngArgs { int w, h, depth, ctype, stride; uint8_t pal[256][4]; };
__global__ __launch_bounds__(256) void k_png_expand(PngArgs a, const uint8_t* __restrict__ rows, uint8_t* __restrict__ out)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.w || y >= a.h) return;
    const uint8_t* src = rows + (size_t)y * a.stride;
    if (a.ctype == 2) {
        uint8_t* d = out + ((size_t)y * a.w + x) * 3;
        d[0] = src[3 * x + 2]; d[1] = src[3 * x + 1]; d[2] = src[3 * x];
    } else if (a.ctype == 6) {
        uint8_t* d = out + ((size_t)y * a.w + x) * 4;
        d[0] = src[4 * x + 2]; d[1] = src[4 * x + 1]; d[2] = src[4 * x]; d[3] = src[4 * x + 3];
    } else {
        int v;                                                // (a packed row is ~w * depth / 8 bytes: src[x] would read past the last one)
        if (a.depth < 8) { const int per = 8 / a.depth, sh = (per - 1 - x % per) * a.depth; v = (src[x / per] >> sh) & ((1 << a.depth) - 1); }
        else v = src[x];
        if (a.ctype == 0) out[(size_t)y * a.w + x] = (uint8_t)(a.depth < 8 ? v * 255 / ((1 << a.depth) - 1) : v);
        else { uint8_t* d = out + ((size_t)y * a.w + x) * 3; d[0] = a.pal[v][0]; d[1] = a.pal[v][1]; d[2] = a.pal[v][2]; }
    }
}

static uvo_status png_decode(Ctx* c, CodecWs* ws, const uint8_t* data, size_t n, const uint8_t** d_out, int* w, int* h, int* channels)
{
    png::Header hd;
    std::vector<uint8_t> idat;
    std::string err;
    if (!png::parse(data, n, &hd, &idat, false, &err)) { c->err = err; return UVO_INVALID_ARG; }
    const size_t stride = png::row_bytes(hd), raw_bytes = stride * (size_t)hd.h;
    UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));                     // the pinned staging buffer of the previous call is free again
    UVO_TRY(grow(c, reinterpret_cast<void**>(&ws->h_coef), &ws->coef_cap, raw_bytes, true));        // (the JPEG path's pinned / device staging buffers, reused)
    if (!png::scanlines(hd, idat, reinterpret_cast<uint8_t*>(ws->h_coef), &err)) { c->err = err; return UVO_INVALID_ARG; }
    { void* p = ws->d_coef; size_t cap = ws->coef_cap_dev; UVO_TRY(grow(c, &p, &cap, raw_bytes, false)); ws->d_coef = static_cast<int16_t*>(p); ws->coef_cap_dev = cap; }
    const int ch = png::channels_out(hd);
    { void* p = ws->d_out; UVO_TRY(grow(c, &p, &ws->out_cap, (size_t)hd.w * hd.h * ch, false)); ws->d_out = static_cast<uint8_t*>(p); }
    UVO_HIP_TRY(c, hipMemcpyAsync(ws->d_coef, ws->h_coef, raw_bytes, hipMemcpyHostToDevice, c->stream));
    PngArgs a;
    a.w = hd.w; a.h = hd.h; a.depth = hd.depth; a.ctype = hd.ctype; a.stride = (int)stride;
    memcpy(a.pal, hd.pal, sizeof(a.pal));
    hipLaunchKernelGGL(k_png_expand, dim3((hd.w + 63) / 64, (hd.h + 3) / 4), dim3(256), 0, c->stream, a, reinterpret_cast<const uint8_t*>(ws->d_coef), ws->d_out);
    UVO_HIP_TRY(c, hipGetLastError());
    *w = hd.w; *h = hd.h; *channels = ch; *d_out = ws->d_out;
    return UVO_OK;
}

uvo_status codec_decode(Ctx* c, const uint8_t* data, size_t n, int bayer, const uint8_t** d_out, int* w, int* h, int* channels)
{
    if (!c->codec_ws) c->codec_ws = new CodecWs();
    CodecWs* ws = static_cast<CodecWs*>(c->codec_ws);
    if (png::is_png(data, n)) {
        UVO_TRY(png_decode(c, ws, data, n, d_out, w, h, channels));
        if (bayer) {
            if (*channels != 1) { c->err = "a bayer-format message must decode to one channel"; return UVO_INVALID_ARG; }
            { void* p = ws->d_out2; UVO_TRY(grow(c, &p, &ws->out2_cap, (size_t)*w * *h * 3, false)); ws->d_out2 = static_cast<uint8_t*>(p); }
            hipLaunchKernelGGL(k_bayer_bggr, dim3((*w + 63) / 64, (*h + 3) / 4), dim3(256), 0, c->stream, ws->d_out, *w, *h, *w, ws->d_out2);
            UVO_HIP_TRY(c, hipGetLastError());
            *channels = 3; *d_out = ws->d_out2;
        }
        UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
        return UVO_OK;
    }
    Jpeg j;
    memset(j.quant, 0, sizeof(j.quant));
    UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));                     // the pinned coefficient buffer of the previous call is free again
    UVO_TRY(jpeg_entropy_decode(c, ws, data, n, &j));
    if (!ws->d_quant) UVO_HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&ws->d_quant), sizeof(j.quant)));
    const size_t coef_bytes = j.total_blocks * 64 * sizeof(int16_t);
    { void* p = ws->d_coef; size_t cap = ws->coef_cap_dev; UVO_TRY(grow(c, &p, &cap, coef_bytes, false)); ws->d_coef = static_cast<int16_t*>(p); ws->coef_cap_dev = cap; }
    IdctArgs ia; ColourArgs ca;
    memset(&ia, 0, sizeof(ia)); memset(&ca, 0, sizeof(ca));
    size_t poff = 0;
    for (int k = 0; k < j.ncomp; k++) {
        const Comp& q = j.comp[k];
        ia.comp[k] = { q.coef_off, q.bw, q.bh, q.tq, poff };
        ca.plane_off[k] = poff; ca.pw[k] = q.bw * 8;
        ca.dw[k] = (j.w * q.h + j.hmax - 1) / j.hmax; ca.dh[k] = (j.h * q.v + j.vmax - 1) / j.vmax;      // jdmaster.c downsampled_width / height
        ca.hs[k] = j.hmax / q.h; ca.vs[k] = j.vmax / q.v;
        poff += (size_t)q.bw * q.bh * 64;
    }
    for (int k = j.ncomp; k < 3; k++) ia.comp[k] = ia.comp[0];
    ia.ncomp = j.ncomp; ia.total_blocks = j.total_blocks;
    ca.ncomp = j.ncomp; ca.w = j.w; ca.h = j.h;
    { void* p = ws->d_planes; UVO_TRY(grow(c, &p, &ws->planes_cap, poff, false)); ws->d_planes = static_cast<uint8_t*>(p); }
    const size_t out_bytes = (size_t)j.w * j.h * j.ncomp;
    { void* p = ws->d_out; UVO_TRY(grow(c, &p, &ws->out_cap, out_bytes, false)); ws->d_out = static_cast<uint8_t*>(p); }
    UVO_HIP_TRY(c, hipMemcpyAsync(ws->d_coef, ws->h_coef, coef_bytes, hipMemcpyHostToDevice, c->stream));
    UVO_HIP_TRY(c, hipMemcpyAsync(ws->d_quant, j.quant, sizeof(j.quant), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_jpeg_idct, dim3((unsigned)((j.total_blocks + 31) / 32)), dim3(256), 0, c->stream, ia, ws->d_coef, ws->d_quant, ws->d_planes);
    hipLaunchKernelGGL(k_jpeg_colour, dim3((j.w + 63) / 64, (j.h + 3) / 4), dim3(256), 0, c->stream, ca, ws->d_planes, ws->d_out);
    UVO_HIP_TRY(c, hipGetLastError());
    *w = j.w; *h = j.h; *channels = j.ncomp; *d_out = ws->d_out;
    if (bayer) {
        if (j.ncomp != 1) { c->err = "a bayer-format message must decode to one channel"; return UVO_INVALID_ARG; }
        { void* p = ws->d_out2; UVO_TRY(grow(c, &p, &ws->out2_cap, (size_t)j.w * j.h * 3, false)); ws->d_out2 = static_cast<uint8_t*>(p); }
        hipLaunchKernelGGL(k_bayer_bggr, dim3((j.w + 63) / 64, (j.h + 3) / 4), dim3(256), 0, c->stream, ws->d_out, j.w, j.h, j.w, ws->d_out2);
        UVO_HIP_TRY(c, hipGetLastError());
        *channels = 3; *d_out = ws->d_out2;
    }
    // j.quant lives on this stack frame and is still being copied: wait for the upload before returning
    UVO_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return UVO_OK;
}

// Size and channel count of what codec_decode would return, from the headers alone (no entropy decoding, no device work)
uvo_status codec_peek(Ctx* c, const uint8_t* data, size_t n, int bayer, int* w, int* h, int* channels)
{
    if (png::is_png(data, n)) {
        png::Header hd; std::string err;
        if (!png::parse(data, n, &hd, nullptr, true, &err)) { c->err = err; return UVO_INVALID_ARG; }
        const int ch = png::channels_out(hd);
        if (bayer && ch != 1) { c->err = "a bayer-format message must decode to one channel"; return UVO_INVALID_ARG; }
        *w = hd.w; *h = hd.h; *channels = bayer ? 3 : ch;
        return UVO_OK;
    }
    Jpeg j;
    memset(j.quant, 0, sizeof(j.quant));
    UVO_TRY(jpeg_entropy_decode(c, nullptr, data, n, &j, true));
    if (bayer && j.ncomp != 1) { c->err = "a bayer-format message must decode to one channel"; return UVO_INVALID_ARG; }
    *w = j.w; *h = j.h; *channels = bayer ? 3 : j.ncomp;
    return UVO_OK;
}

uvo_status codec_bayer(Ctx* c, const uint8_t* bayer, int w, int h, int stride, int mem, const uint8_t** d_out)
{
    if (!c->codec_ws) c->codec_ws = new CodecWs();
    CodecWs* ws = static_cast<CodecWs*>(c->codec_ws);
    { void* p = ws->d_out; UVO_TRY(grow(c, &p, &ws->out_cap, (size_t)w * h, false)); ws->d_out = static_cast<uint8_t*>(p); }
    { void* p = ws->d_out2; UVO_TRY(grow(c, &p, &ws->out2_cap, (size_t)w * h * 3, false)); ws->d_out2 = static_cast<uint8_t*>(p); }
    UVO_HIP_TRY(c, hipMemcpy2DAsync(ws->d_out, w, bayer, stride, w, h, mem == UVO_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_bayer_bggr, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, c->stream, ws->d_out, w, h, w, ws->d_out2);
    UVO_HIP_TRY(c, hipGetLastError());
    *d_out = ws->d_out2;
    return UVO_OK;
}

}  // namespace uvo
