/*
 * o_codec.c -- CPU ORACLE (test infrastructure): the compressed-image ingest in front of get_image
 * (SURVEY.md 8(f) N3): from_ros_to_cv_image, uvo_libraries/src/math_utility.cpp:154-173 =
 * cv_bridge::toCvCopy(sensor_msgs/CompressedImage) -> cv::imdecode -> libjpeg(-turbo), then
 * cv::cvtColor(COLOR_BayerBGGR2BGR) for "bayer" formats.
 *
 * JPEG: baseline / extended sequential Huffman, 8 bit, 1 or 3 components, sampling factors 1 or 2,
 * restart intervals; the decode pipeline of libjpeg with its defaults, which is what cv::imdecode runs:
 *   jdhuff.c   entropy decoding                      -> orc_jpeg_* below
 *   jidctint.c jpeg_idct_islow (JDCT_ISLOW, CONST_BITS 13, PASS1_BITS 2), dequantisation inside
 *   jdsample.c fancy (triangle) upsampling h2v1 / h2v2 / h1v2, replication otherwise
 *   jdcolor.c  YCbCr -> RGB with the 16-bit fixed-point tables
 * Output: BGR interleaved (cv::imdecode's channel order) or one grey channel.
 *
 *   >>> PINNED: unlike the rest of the oracle this file IS checked against a real implementation --
 *   >>> libjpeg-turbo as bundled with Pillow, byte for byte (tests/test_codec.py, fixtures in
 *   >>> tests/golden/jpeg_*.npz made by tests/golden/make_golden_codec.py).
 *
 * Bayer: cv::cvtColor(COLOR_BayerBGGR2BGR) = COLOR_BayerRG2BGR, bilinear ([UPSTREAM] imgproc/src/demosaicing.cpp
 * Bayer2RGB_): interior pixels from the 3 x 3 neighbourhood, first/last column copied from its neighbour, first/last
 * row copied from its neighbour.  Recalled behaviour, medium confidence, NOT pinned (Pillow has no demosaic).
 */
#include "uvo_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ bit reader over entropy-coded data */
typedef struct { const uint8_t* p; const uint8_t* end; uint32_t acc; int nbits; int marker; } bitreader;

static void br_fill(bitreader* b)
{
    while (b->nbits <= 24) {
        int c = 0;
        if (!b->marker && b->p < b->end) {
            c = *b->p++;
            if (c == 0xFF) {
                int c2 = b->p < b->end ? *b->p : 0xD9;
                if (c2 == 0) b->p++;                         /* stuffed zero */
                else { b->marker = c2; b->p--; c = 0; }      /* a marker: feed zeros from here on (jdhuff.c does the same) */
            }
        }
        b->acc |= (uint32_t)c << (24 - b->nbits);
        b->nbits += 8;
    }
}
static int br_peek(bitreader* b, int n) { br_fill(b); return (int)(b->acc >> (32 - n)); }
static void br_skip(bitreader* b, int n) { b->acc <<= n; b->nbits -= n; }
static int br_get(bitreader* b, int n) { if (n == 0) return 0; int v = br_peek(b, n); br_skip(b, n); return v; }

typedef struct { uint8_t bits[17]; uint8_t vals[256]; int mincode[17], maxcode[18], valptr[17]; int present; } hufftab;

static void huff_build(hufftab* h)
{
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        h->valptr[l] = k; h->mincode[l] = code;
        code += h->bits[l]; k += h->bits[l];
        h->maxcode[l] = h->bits[l] ? code - 1 : -1;
        code <<= 1;
    }
    h->maxcode[17] = 0x7FFFFFFF;
}
static int huff_decode(bitreader* b, const hufftab* h)
{
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | br_get(b, 1);
        if (h->maxcode[l] >= 0 && code <= h->maxcode[l] && code >= h->mincode[l]) return h->vals[h->valptr[l] + code - h->mincode[l]];
    }
    return 0;                                                /* corrupt data: libjpeg warns and returns 0 */
}
static int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

static const uint8_t kZigzag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

/* ------------------------------------------------------------------ jidctint.c: jpeg_idct_islow */
#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172
#define CONST_BITS 13
#define PASS1_BITS 2
/* libjpeg computes in JLONG; valid streams stay far inside 32 bits, corrupt ones may not: every add / multiply here is done modulo
 * 2^32 (unsigned), so a damaged stream gives some picture instead of undefined behaviour -- as the SIMD IDCTs of libjpeg-turbo do */
typedef uint32_t wi;
#define DESCALE(x, n) ((wi)((int32_t)((x) + ((wi)1 << ((n) - 1))) >> (n)))

static uint8_t range_limit(int32_t x)
{
    /* IDCT_range_limit[x & RANGE_MASK] as jdmaster.c prepare_range_limit_table lays the post-IDCT table out (RANGE_MASK = 1023,
     * the table is centred on 128): v = x & 1023:  [0, 128) -> v + 128;  [128, 512) -> 255;  [512, 896) -> 0;  [896, 1024) ->
     * v - 896.  Equal to clamp(x + 128, 0, 255) for -512 <= x < 512; beyond that it wraps, which only damaged streams reach. */
    const int32_t v = x & 1023;
    if (v < 128) return (uint8_t)(v + 128);
    if (v < 512) return 255;
    if (v < 896) return 0;
    return (uint8_t)(v - 896);
}

void orc_jpeg_idct_islow(const int16_t* coef /* natural order */, const uint16_t* quant /* natural order */, uint8_t* out, int stride)
{
    wi ws[64];
    for (int c = 0; c < 8; c++) {
        wi in[8];
        for (int r = 0; r < 8; r++) in[r] = (wi)((int32_t)coef[r * 8 + c] * (int32_t)quant[r * 8 + c]);
        wi z2 = in[2], z3 = in[6];
        wi z1 = (z2 + z3) * FIX_0_541196100;
        wi tmp2 = z1 + z3 * (wi)(-FIX_1_847759065);
        wi tmp3 = z1 + z2 * FIX_0_765366865;
        z2 = in[0]; z3 = in[4];
        wi tmp0 = (z2 + z3) * (1 << CONST_BITS), tmp1 = (z2 - z3) * (1 << CONST_BITS);
        wi tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; wi z4 = tmp1 + tmp3;
        wi z5 = (z3 + z4) * FIX_1_175875602;
        tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869; tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
        z1 *= (wi)(-FIX_0_899976223); z2 *= (wi)(-FIX_2_562915447); z3 *= (wi)(-FIX_1_961570560); z4 *= (wi)(-FIX_0_390180644);
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        ws[0 * 8 + c] = DESCALE(tmp10 + tmp3, CONST_BITS - PASS1_BITS); ws[7 * 8 + c] = DESCALE(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
        ws[1 * 8 + c] = DESCALE(tmp11 + tmp2, CONST_BITS - PASS1_BITS); ws[6 * 8 + c] = DESCALE(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
        ws[2 * 8 + c] = DESCALE(tmp12 + tmp1, CONST_BITS - PASS1_BITS); ws[5 * 8 + c] = DESCALE(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
        ws[3 * 8 + c] = DESCALE(tmp13 + tmp0, CONST_BITS - PASS1_BITS); ws[4 * 8 + c] = DESCALE(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
    }
    for (int r = 0; r < 8; r++) {
        const wi* w = ws + r * 8;
        wi z2 = w[2], z3 = w[6];
        wi z1 = (z2 + z3) * FIX_0_541196100;
        wi tmp2 = z1 + z3 * (wi)(-FIX_1_847759065);
        wi tmp3 = z1 + z2 * FIX_0_765366865;
        wi tmp0 = (w[0] + w[4]) * (1 << CONST_BITS), tmp1 = (w[0] - w[4]) * (1 << CONST_BITS);
        wi tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; wi z4 = tmp1 + tmp3;
        wi z5 = (z3 + z4) * FIX_1_175875602;
        tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869; tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
        z1 *= (wi)(-FIX_0_899976223); z2 *= (wi)(-FIX_2_562915447); z3 *= (wi)(-FIX_1_961570560); z4 *= (wi)(-FIX_0_390180644);
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        uint8_t* o = out + (size_t)r * stride;
        const int S = CONST_BITS + PASS1_BITS + 3;
        o[0] = range_limit((int32_t)DESCALE(tmp10 + tmp3, S)); o[7] = range_limit((int32_t)DESCALE(tmp10 - tmp3, S));
        o[1] = range_limit((int32_t)DESCALE(tmp11 + tmp2, S)); o[6] = range_limit((int32_t)DESCALE(tmp11 - tmp2, S));
        o[2] = range_limit((int32_t)DESCALE(tmp12 + tmp1, S)); o[5] = range_limit((int32_t)DESCALE(tmp12 - tmp1, S));
        o[3] = range_limit((int32_t)DESCALE(tmp13 + tmp0, S)); o[4] = range_limit((int32_t)DESCALE(tmp13 - tmp0, S));
    }
}

/* ------------------------------------------------------------------ header parsing + entropy decoding */
typedef struct { int id, h, v, tq, td, ta; int bw, bh; /* blocks per row / column incl. MCU padding */ int16_t* coef; int dc_pred; } jcomp;
typedef struct {
    int w, h, ncomp, hmax, vmax, restart_interval, mcux, mcuy;
    uint16_t quant[4][64];                                   /* natural order */
    hufftab dc[4], ac[4];
    jcomp comp[3];
} jdec;

static int rd16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

/* Decodes the headers and all coefficients.  Returns 0, or a negative error: -1 not a JPEG / truncated, -2 unsupported
 * (progressive, arithmetic, 12 bit, 4 components, sampling factors > 2), -3 out of memory. */
static int jpeg_read(const uint8_t* data, size_t n, jdec* d)
{
    memset(d, 0, sizeof(*d));
    if (n < 4 || data[0] != 0xFF || data[1] != 0xD8) return -1;
    size_t pos = 2;
    int have_sof = 0;
    while (pos + 4 <= n) {
        if (data[pos] != 0xFF) { pos++; continue; }
        const int m = data[pos + 1];
        if (m == 0xFF) { pos++; continue; }
        pos += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;
        if (pos + 2 > n) return -1;
        const int len = rd16(data + pos);
        if (len < 2 || pos + len > n) return -1;
        const uint8_t* s = data + pos + 2; const int sl = len - 2;
        if (m == 0xDB) {                                     /* DQT */
            int o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15; o++;
                if (tq > 3 || pq > 1 || o + (pq ? 128 : 64) > sl) return -1;
                for (int k = 0; k < 64; k++) { d->quant[tq][kZigzag[k]] = (uint16_t)(pq ? rd16(s + o + 2 * k) : s[o + k]); }
                o += pq ? 128 : 64;
            }
        } else if (m == 0xC4) {                              /* DHT */
            int o = 0;
            while (o + 17 <= sl) {
                const int tc = s[o] >> 4, th = s[o] & 15; o++;
                if (th > 3 || tc > 1) return -1;
                hufftab* h = tc ? &d->ac[th] : &d->dc[th];
                int cnt = 0;
                h->bits[0] = 0;
                for (int l = 1; l <= 16; l++) { h->bits[l] = s[o + l - 1]; cnt += h->bits[l]; }
                o += 16;
                if (cnt > 256 || o + cnt > sl) return -1;
                memcpy(h->vals, s + o, (size_t)cnt); o += cnt;
                huff_build(h); h->present = 1;
            }
        } else if (m == 0xC0 || m == 0xC1) {                 /* SOF0 / SOF1 */
            if (have_sof || sl < 6) return -1;               /* (a second frame header would also leak the first one's planes) */
            if (s[0] != 8) return -2;
            d->h = rd16(s + 1); d->w = rd16(s + 3); d->ncomp = s[5];
            if (d->w <= 0 || d->h <= 0) return -1;
            if ((long long)d->w * d->h > (1LL << 26)) return -2;                                  /* 64 Mpixel cap: a damaged header must not ask for gigabytes */
            if (d->ncomp != 1 && d->ncomp != 3) return -2;
            if (sl < 6 + 3 * d->ncomp) return -1;
            for (int c = 0; c < d->ncomp; c++) {
                jcomp* k = &d->comp[c];
                k->id = s[6 + 3 * c]; k->h = s[7 + 3 * c] >> 4; k->v = s[7 + 3 * c] & 15; k->tq = s[8 + 3 * c];
                if (k->h < 1 || k->h > 2 || k->v < 1 || k->v > 2 || k->tq > 3) return -2;
                if (k->h > d->hmax) d->hmax = k->h;
                if (k->v > d->vmax) d->vmax = k->v;
            }
            if (d->ncomp == 1) { d->comp[0].h = d->comp[0].v = 1; d->hmax = d->vmax = 1; }      /* a single component is never interleaved */
            d->mcux = (d->w + 8 * d->hmax - 1) / (8 * d->hmax); d->mcuy = (d->h + 8 * d->vmax - 1) / (8 * d->vmax);
            for (int c = 0; c < d->ncomp; c++) {
                jcomp* k = &d->comp[c];
                k->bw = d->mcux * k->h; k->bh = d->mcuy * k->v;
                k->coef = (int16_t*)calloc((size_t)k->bw * k->bh * 64, sizeof(int16_t));
                if (!k->coef) return -3;
            }
            have_sof = 1;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            return -2;                                       /* progressive, lossless, arithmetic ... */
        } else if (m == 0xDD) {
            if (sl < 2) return -1;
            d->restart_interval = rd16(s);
        } else if (m == 0xDA) {                              /* SOS: one interleaved scan with every component (baseline encoders) */
            if (!have_sof || sl < 1) return -1;
            const int ns = s[0];
            if (ns != d->ncomp) return -2;
            if (sl < 1 + 2 * ns + 3) return -1;
            for (int i = 0; i < ns; i++) {
                const int cid = s[1 + 2 * i];
                for (int c = 0; c < d->ncomp; c++) if (d->comp[c].id == cid) { d->comp[c].td = s[2 + 2 * i] >> 4; d->comp[c].ta = s[2 + 2 * i] & 15; }
            }
            for (int c = 0; c < d->ncomp; c++) {             /* jdhuff.c start_pass: the tables a scan names must exist; DC categories are 0..15 */
                const jcomp* k = &d->comp[c];
                if (k->td > 3 || k->ta > 3 || !d->dc[k->td].present || !d->ac[k->ta].present) return -1;
                const hufftab* h = &d->dc[k->td];
                int cnt = 0;
                for (int l = 1; l <= 16; l++) cnt += h->bits[l];
                for (int i = 0; i < cnt; i++) if (h->vals[i] > 15) return -1;
            }
            bitreader b = { data + pos + len, data + n, 0, 0, 0 };
            int restarts_left = d->restart_interval;
            for (int c = 0; c < d->ncomp; c++) d->comp[c].dc_pred = 0;
            for (int my = 0; my < d->mcuy; my++)
                for (int mx = 0; mx < d->mcux; mx++) {
                    if (d->restart_interval && restarts_left == 0) {
                        /* RSTn: discard the partial byte, skip the marker, reset the predictors */
                        b.acc = 0; b.nbits = 0;
                        if (b.marker >= 0xD0 && b.marker <= 0xD7) { b.p += 2; b.marker = 0; }
                        else { while (b.p + 1 < b.end && !(b.p[0] == 0xFF && b.p[1] >= 0xD0 && b.p[1] <= 0xD7)) b.p++; if (b.p + 1 < b.end) b.p += 2; b.marker = 0; }
                        for (int c = 0; c < d->ncomp; c++) d->comp[c].dc_pred = 0;
                        restarts_left = d->restart_interval;
                    }
                    for (int c = 0; c < d->ncomp; c++) {
                        jcomp* k = &d->comp[c];
                        for (int by = 0; by < k->v; by++)
                            for (int bx = 0; bx < k->h; bx++) {
                                int16_t* blk = k->coef + ((size_t)(my * k->v + by) * k->bw + (mx * k->h + bx)) * 64;
                                int s_ = huff_decode(&b, &d->dc[k->td]);
                                int diff = s_ ? extend(br_get(&b, s_), s_) : 0;
                                k->dc_pred = (int)((unsigned)k->dc_pred + (unsigned)diff);   /* (wraps on damaged streams) */
                                blk[0] = (int16_t)k->dc_pred;
                                for (int kk = 1; kk < 64;) {
                                    const int rs = huff_decode(&b, &d->ac[k->ta]);
                                    const int r = rs >> 4, sz = rs & 15;
                                    if (sz == 0) { if (r == 15) { kk += 16; continue; } break; }
                                    kk += r;
                                    if (kk > 63) break;
                                    blk[kZigzag[kk]] = (int16_t)extend(br_get(&b, sz), sz);
                                    kk++;
                                }
                            }
                    }
                    if (d->restart_interval) restarts_left--;
                }
            return 0;                                        /* everything after the first scan is ignored (baseline has one) */
        }
        pos += len;
    }
    return -1;
}

static void jpeg_free(jdec* d) { for (int c = 0; c < 3; c++) free(d->comp[c].coef); }

/* ------------------------------------------------------------------ jdsample.c */
/* One output row of fancy h2v1 upsampling: in[0..w) -> out[0..2w) */
static void up_h2v1(const uint8_t* in, int w, uint8_t* out)
{
    if (w == 1) { out[0] = out[1] = in[0]; return; }
    out[0] = in[0]; out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
    for (int i = 1; i < w - 1; i++) {
        const int v = in[i] * 3;
        out[2 * i] = (uint8_t)((v + in[i - 1] + 1) >> 2);
        out[2 * i + 1] = (uint8_t)((v + in[i + 1] + 2) >> 2);
    }
    const int v = in[w - 1] * 3;
    out[2 * w - 2] = (uint8_t)((v + in[w - 2] + 1) >> 2); out[2 * w - 1] = in[w - 1];
}
/* One output row of fancy h2v2 upsampling from the nearer (in0) and farther (in1) input rows */
static void up_h2v2(const uint8_t* in0, const uint8_t* in1, int w, uint8_t* out)
{
    int thiscol = in0[0] * 3 + in1[0];
    if (w == 1) { out[0] = (uint8_t)((thiscol * 4 + 8) >> 4); out[1] = (uint8_t)((thiscol * 4 + 7) >> 4); return; }
    int nextcol = in0[1] * 3 + in1[1], lastcol;
    out[0] = (uint8_t)((thiscol * 4 + 8) >> 4); out[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
    lastcol = thiscol; thiscol = nextcol;
    for (int i = 1; i < w - 1; i++) {
        nextcol = in0[i + 1] * 3 + in1[i + 1];
        out[2 * i] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
        out[2 * i + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
        lastcol = thiscol; thiscol = nextcol;
    }
    out[2 * w - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4); out[2 * w - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
}

/* plane: the component's decoded samples (stride pw); dw x dh = its true downsampled size; out: full-resolution w x h */
static void upsample(const uint8_t* plane, int pw, int dw, int dh, int hs, int vs, int w, int h, uint8_t* out)
{
    uint8_t* row = (uint8_t*)malloc((size_t)2 * dw + 2);
    for (int y = 0; y < h; y++) {
        uint8_t* o = out + (size_t)y * w;
        if (hs == 1 && vs == 1) { memcpy(o, plane + (size_t)y * pw, (size_t)w); continue; }
        const int sy = y / vs;
        const uint8_t* in0 = plane + (size_t)sy * pw;
        if (vs == 1) {                                       /* h2v1 */
            up_h2v1(in0, dw, row); memcpy(o, row, (size_t)w);
            continue;
        }
        /* vertical factor 2: output row 2k takes the row above as the farther row, 2k+1 the row below; the image's first and
         * last sample rows stand in for the missing neighbours (jdmainct.c context rows) */
        int other = (y & 1) ? sy + 1 : sy - 1;
        if (other < 0) other = 0;
        if (other > dh - 1) other = dh - 1;
        const uint8_t* in1 = plane + (size_t)other * pw;
        if (hs == 2) { up_h2v2(in0, in1, dw, row); memcpy(o, row, (size_t)w); }
        else {                                               /* h1v2 (libjpeg-turbo h1v2_fancy_upsample) */
            const int bias = (y & 1) ? 2 : 1;
            for (int x = 0; x < w; x++) o[x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
        }
    }
    free(row);
}

/* ------------------------------------------------------------------ public: decode to BGR (3 components) or grey (1) */
int orc_jpeg_decode(const uint8_t* data, size_t n, uint8_t* out, size_t cap, int* w_out, int* h_out, int* channels_out)
{
    jdec d;
    int rc = jpeg_read(data, n, &d);
    if (rc != 0) { jpeg_free(&d); return rc; }
    const int w = d.w, h = d.h;
    if (w_out) *w_out = w;
    if (h_out) *h_out = h;
    if (channels_out) *channels_out = d.ncomp;
    if (!out || cap < (size_t)w * h * d.ncomp) { jpeg_free(&d); return out ? -4 : 0; }
    uint8_t* full[3] = { NULL, NULL, NULL };
    for (int c = 0; c < d.ncomp; c++) {
        jcomp* k = &d.comp[c];
        const int pw = k->bw * 8, ph = k->bh * 8;
        uint8_t* plane = (uint8_t*)malloc((size_t)pw * ph);
        for (int by = 0; by < k->bh; by++)
            for (int bx = 0; bx < k->bw; bx++)
                orc_jpeg_idct_islow(k->coef + ((size_t)by * k->bw + bx) * 64, d.quant[k->tq], plane + (size_t)by * 8 * pw + bx * 8, pw);
        const int dw = (w * k->h + d.hmax - 1) / d.hmax, dh = (h * k->v + d.vmax - 1) / d.vmax;     /* jdmaster.c downsampled_width / height */
        full[c] = (uint8_t*)malloc((size_t)w * h);
        upsample(plane, pw, dw, dh, d.hmax / k->h, d.vmax / k->v, w, h, full[c]);
        free(plane);
    }
    if (d.ncomp == 1) memcpy(out, full[0], (size_t)w * h);
    else {
        /* jdcolor.c build_ycc_rgb_table + ycc_rgb_convert; output order B, G, R */
        for (size_t i = 0; i < (size_t)w * h; i++) {
            const int y = full[0][i], cb = full[1][i] - 128, cr = full[2][i] - 128;
            const int r = y + (int)((91881 * (int32_t)cr + 32768) >> 16);
            const int g = y + (int)(((-22554) * (int32_t)cb + 32768 + (-46802) * (int32_t)cr) >> 16);
            const int b = y + (int)((116130 * (int32_t)cb + 32768) >> 16);
            out[3 * i] = (uint8_t)(b < 0 ? 0 : b > 255 ? 255 : b);
            out[3 * i + 1] = (uint8_t)(g < 0 ? 0 : g > 255 ? 255 : g);
            out[3 * i + 2] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
        }
    }
    for (int c = 0; c < 3; c++) free(full[c]);
    jpeg_free(&d);
    return 0;
}

/* ------------------------------------------------------------------ COLOR_BayerBGGR2BGR ( = COLOR_BayerRG2BGR ), bilinear */
void orc_bayer_bggr2bgr(const uint8_t* bayer, int w, int h, int stride, uint8_t* dst)
{
    const size_t ds = (size_t)w * 3;
    if (w < 3 || h < 3) { memset(dst, 0, ds * h); return; }
    /* BGGR: even rows B G B G .., odd rows G R G R ..  OpenCV walks rows 1 .. h-2 and columns 1 .. w-2 with `blue` and
     * `start_with_green` toggling per row; written out per pixel here */
    for (int y = 1; y < h - 1; y++) {
        const uint8_t* r0 = bayer + (size_t)(y - 1) * stride; const uint8_t* r1 = r0 + stride; const uint8_t* r2 = r1 + stride;
        uint8_t* o = dst + (size_t)y * ds;
        for (int x = 1; x < w - 1; x++) {
            const int ev_y = (y & 1) == 0, ev_x = (x & 1) == 0;
            int B, G, R;
            if (ev_y && ev_x) {                              /* a blue site */
                B = r1[x]; G = (r0[x] + r1[x - 1] + r1[x + 1] + r2[x] + 2) >> 2; R = (r0[x - 1] + r0[x + 1] + r2[x - 1] + r2[x + 1] + 2) >> 2;
            } else if (!ev_y && !ev_x) {                     /* a red site */
                R = r1[x]; G = (r0[x] + r1[x - 1] + r1[x + 1] + r2[x] + 2) >> 2; B = (r0[x - 1] + r0[x + 1] + r2[x - 1] + r2[x + 1] + 2) >> 2;
            } else if (ev_y) {                               /* green on a blue row: blue left/right, red above/below */
                G = r1[x]; B = (r1[x - 1] + r1[x + 1] + 1) >> 1; R = (r0[x] + r2[x] + 1) >> 1;
            } else {                                         /* green on a red row */
                G = r1[x]; R = (r1[x - 1] + r1[x + 1] + 1) >> 1; B = (r0[x] + r2[x] + 1) >> 1;
            }
            o[3 * x] = (uint8_t)B; o[3 * x + 1] = (uint8_t)G; o[3 * x + 2] = (uint8_t)R;
        }
        memcpy(o, o + 3, 3); memcpy(o + 3 * (w - 1), o + 3 * (w - 2), 3);          /* first / last column = its neighbour */
    }
    memcpy(dst, dst + ds, ds); memcpy(dst + (size_t)(h - 1) * ds, dst + (size_t)(h - 2) * ds, ds);     /* first / last row */
}
