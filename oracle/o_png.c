/* o_png.c -- TEST INFRASTRUCTURE (CPU oracle; never linked into the product).
 *
 * PNG payloads of sensor_msgs/CompressedImage as cv_bridge::toCvCopy hands them to the reference
 * (uvo_libraries/src/math_utility.cpp:154-173: toCvCopy -> cv::imdecode(IMREAD_UNCHANGED); OpenCV's grfmt_png.cpp drives
 * libpng with png_set_bgr / png_set_palette_to_rgb): 8-bit grey -> H x W, RGB / palette -> H x W x 3 BGR, RGBA -> H x W x 4 BGRA.
 * PNG is lossless and its decoding is fixed by the specification (RFC 2083 / ISO 15948; DEFLATE: RFC 1951, zlib: RFC 1950), so any
 * conforming decoder gives the same bytes; this one is pinned by Pillow's decoder (zlib), see tests/golden/make_golden_codec.py.
 * A deliberately plain restatement: bit-by-bit canonical Huffman decoding, no tables.
 * Grey and palette images of 1, 2 or 4 bits are expanded as libpng does for OpenCV (png_set_expand_gray_1_2_4_to_8: v * 255 / (2^d - 1)).
 * Refused (error code, never a guess): 16-bit samples, interlacing, grey + alpha, palettes with tRNS. */
#include "uvo_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef struct { const uint8_t* p; size_t n, pos; uint32_t bitbuf; int bitcnt; int err; } bits_t;

static int getbit(bits_t* b)
{
    if (b->bitcnt == 0) {
        if (b->pos >= b->n) { b->err = 1; return 0; }
        b->bitbuf = b->p[b->pos++]; b->bitcnt = 8;
    }
    const int v = b->bitbuf & 1; b->bitbuf >>= 1; b->bitcnt--;
    return v;
}
static uint32_t getbits(bits_t* b, int n) { uint32_t v = 0; for (int i = 0; i < n; i++) v |= (uint32_t)getbit(b) << i; return v; }

typedef struct { uint16_t count[16], symbol[288]; } huff_t;
static int huff_build(huff_t* h, const uint8_t* len, int n)
{
    uint16_t offs[16];
    memset(h->count, 0, sizeof(h->count));
    for (int i = 0; i < n; i++) h->count[len[i]]++;
    if (h->count[0] == n) return 0;                       /* no codes: legal for an unused distance tree */
    int left = 1;
    for (int l = 1; l < 16; l++) { left <<= 1; left -= h->count[l]; if (left < 0) return -1; }
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + h->count[l]);
    for (int i = 0; i < n; i++) if (len[i]) h->symbol[offs[len[i]]++] = (uint16_t)i;
    return left;                                          /* > 0: incomplete code (allowed for single-code distance trees) */
}
static int huff_decode(bits_t* b, const huff_t* h)
{
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; l++) {
        code |= getbit(b);
        const int cnt = h->count[l];
        if (code - cnt < first) return h->symbol[index + (code - first)];
        index += cnt; first += cnt; first <<= 1; code <<= 1;
    }
    b->err = 1;
    return -1;
}

static const uint16_t kLenBase[29] = { 3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258 };
static const uint8_t kLenExtra[29] = { 0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0 };
static const uint16_t kDistBase[30] = { 1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577 };
static const uint8_t kDistExtra[30] = { 0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13 };

/* RFC 1951 inflate of a raw deflate stream into out[0..cap); returns bytes written or -1 */
static long inflate_raw(bits_t* b, uint8_t* out, size_t cap)
{
    size_t o = 0;
    int last;
    do {
        last = getbit(b);
        const int type = (int)getbits(b, 2);
        if (b->err) return -1;
        if (type == 0) {
            b->bitcnt = 0;
            if (b->pos + 4 > b->n) return -1;
            const unsigned len = b->p[b->pos] | (b->p[b->pos + 1] << 8), nlen = b->p[b->pos + 2] | (b->p[b->pos + 3] << 8);
            b->pos += 4;
            if ((len ^ 0xFFFFu) != nlen || b->pos + len > b->n || o + len > cap) return -1;
            memcpy(out + o, b->p + b->pos, len); b->pos += len; o += len;
        } else if (type == 1 || type == 2) {
            huff_t hl, hd;
            uint8_t len[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; i++) len[i] = 8;
                for (; i < 256; i++) len[i] = 9;
                for (; i < 280; i++) len[i] = 7;
                for (; i < 288; i++) len[i] = 8;
                huff_build(&hl, len, 288);
                for (i = 0; i < 30; i++) len[i] = 5;
                huff_build(&hd, len, 30);
            } else {
                static const uint8_t order[19] = { 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15 };
                const int nlen = (int)getbits(b, 5) + 257, ndist = (int)getbits(b, 5) + 1, ncode = (int)getbits(b, 4) + 4;
                if (nlen > 286 || ndist > 30) return -1;
                uint8_t cl[19];
                memset(cl, 0, sizeof(cl));
                for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)getbits(b, 3);
                huff_t hc;
                if (huff_build(&hc, cl, 19) != 0) return -1;
                int idx = 0;
                while (idx < nlen + ndist) {
                    const int sym = huff_decode(b, &hc);
                    if (sym < 0 || b->err) return -1;
                    if (sym < 16) len[idx++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (idx == 0) return -1; val = len[idx - 1]; rep = 3 + (int)getbits(b, 2); }
                        else if (sym == 17) rep = 3 + (int)getbits(b, 3);
                        else rep = 11 + (int)getbits(b, 7);
                        if (idx + rep > nlen + ndist) return -1;
                        while (rep--) len[idx++] = (uint8_t)val;
                    }
                }
                if (len[256] == 0) return -1;
                int e = huff_build(&hl, len, nlen);
                if (e < 0 || (e > 0 && nlen - hl.count[0] != 1)) return -1;
                e = huff_build(&hd, len + nlen, ndist);
                if (e < 0 || (e > 0 && ndist - hd.count[0] != 1)) return -1;
            }
            for (;;) {
                const int sym = huff_decode(b, &hl);
                if (sym < 0 || b->err) return -1;
                if (sym < 256) { if (o >= cap) return -1; out[o++] = (uint8_t)sym; }
                else if (sym == 256) break;
                else {
                    const int s = sym - 257;
                    if (s >= 29) return -1;
                    const int length = kLenBase[s] + (int)getbits(b, kLenExtra[s]);
                    const int ds = huff_decode(b, &hd);
                    if (ds < 0 || ds >= 30 || b->err) return -1;
                    const size_t dist = kDistBase[ds] + getbits(b, kDistExtra[ds]);
                    if (dist > o || o + (size_t)length > cap) return -1;
                    for (int i = 0; i < length; i++, o++) out[o] = out[o - dist];
                }
            }
        } else return -1;
    } while (!last);
    return (long)o;
}

static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static uint32_t crc32_of(const uint8_t* p, size_t n)
{
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) { c ^= p[i]; for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u))); }
    return c ^ 0xFFFFFFFFu;
}
static int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

/* returns 0 and fills w/h/channels; out == NULL: size query.  Error codes: 1 not a PNG / damaged, 2 unsupported kind, 3 capacity */
int orc_png_decode(const uint8_t* data, size_t n, uint8_t* out, size_t cap, int* w_out, int* h_out, int* channels_out)
{
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n' };
    if (n < 8 + 25 || memcmp(data, sig, 8) != 0) return 1;
    size_t pos = 8;
    int w = 0, h = 0, ctype = -1, depth = 8, have_plte = 0, have_trns = 0, seen_ihdr = 0;
    uint8_t pal[256][3];
    uint8_t* z = NULL; size_t zn = 0, zcap = 0;
    int rc = 1;
    for (;;) {
        if (pos + 12 > n) goto done;
        const uint32_t len = be32(data + pos);
        const uint8_t* type = data + pos + 4;
        if (len > n || pos + 12 + len > n) goto done;
        if (crc32_of(type, 4 + (size_t)len) != be32(data + pos + 8 + len)) goto done;
        const uint8_t* d = data + pos + 8;
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13 || seen_ihdr) goto done;
            w = (int)be32(d); h = (int)be32(d + 4);
            depth = d[8]; ctype = d[9];
            if (w <= 0 || h <= 0 || (long long)w * h > (1LL << 26) || d[10] != 0 || d[11] != 0) goto done;
            if (d[12] != 0 || !(ctype == 0 || ctype == 2 || ctype == 3 || ctype == 6)) { rc = 2; goto done; }
            if (!(depth == 8 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) { rc = 2; goto done; }
            seen_ihdr = 1;
        } else if (!seen_ihdr) goto done;
        else if (!memcmp(type, "PLTE", 4)) {
            if (len % 3 || len > 768) goto done;
            memset(pal, 0, sizeof(pal));
            for (uint32_t i = 0; i < len / 3; i++) { pal[i][0] = d[3 * i]; pal[i][1] = d[3 * i + 1]; pal[i][2] = d[3 * i + 2]; }
            have_plte = 1;
        } else if (!memcmp(type, "tRNS", 4)) have_trns = 1;
        else if (!memcmp(type, "IDAT", 4)) {
            if (zn + len > zcap) { zcap = (zn + len) * 2 + 4096; uint8_t* t = (uint8_t*)realloc(z, zcap); if (!t) goto done; z = t; }
            memcpy(z + zn, d, len); zn += len;
        } else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!seen_ihdr || zn < 6) goto done;
    if (ctype == 3 && (!have_plte || have_trns)) { rc = have_plte ? 2 : 1; goto done; }
    {
        const int ch_in = ctype == 0 ? 1 : (ctype == 2 ? 3 : (ctype == 3 ? 1 : 4));
        const int ch_out = ctype == 0 ? 1 : (ctype == 6 ? 4 : 3);
        *w_out = w; *h_out = h; *channels_out = ch_out;
        if (!out) { rc = 0; goto done; }
        if ((size_t)w * h * ch_out > cap) { rc = 3; goto done; }
        /* zlib wrapper (RFC 1950): CMF, FLG, deflate data, Adler-32 */
        if ((z[0] & 0x0F) != 8 || ((z[0] << 8) | z[1]) % 31 != 0 || (z[1] & 0x20)) goto done;
        const size_t stride = ((size_t)w * ch_in * depth + 7) / 8, raw_n = (stride + 1) * (size_t)h;
        uint8_t* raw = (uint8_t*)malloc(raw_n);
        if (!raw) goto done;
        bits_t b = { z + 2, zn - 2, 0, 0, 0, 0 };
        const long got = inflate_raw(&b, raw, raw_n);
        int ok = got == (long)raw_n && !b.err;
        if (ok) {
            uint32_t s1 = 1, s2 = 0;
            for (size_t i = 0; i < raw_n; i++) { s1 = (s1 + raw[i]) % 65521u; s2 = (s2 + s1) % 65521u; }
            b.bitcnt = 0;
            ok = b.pos + 4 <= b.n && be32(b.p + b.pos) == ((s2 << 16) | s1);
        }
        for (int y = 0; ok && y < h; y++) {               /* unfilter in place (RFC 2083 section 6) */
            uint8_t* cur = raw + (size_t)y * (stride + 1) + 1;
            const uint8_t* up = y ? cur - (stride + 1) : NULL;
            const int ft = cur[-1];
            if (ft > 4) { ok = 0; break; }
            const size_t bpp = depth < 8 ? 1 : (size_t)ch_in;     /* filter unit: bytes per complete pixel, at least one */
            for (size_t x = 0; x < stride; x++) {
                const int a = x >= bpp ? cur[x - bpp] : 0, bb = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
                int v = cur[x];
                if (ft == 1) v += a; else if (ft == 2) v += bb; else if (ft == 3) v += (a + bb) >> 1; else if (ft == 4) v += paeth(a, bb, c);
                cur[x] = (uint8_t)v;
            }
        }
        for (int y = 0; ok && y < h; y++) {
            const uint8_t* src = raw + (size_t)y * (stride + 1) + 1;
            uint8_t* dst = out + (size_t)y * w * ch_out;
            for (int x = 0; x < w; x++) {
                int v1 = src[x];                              /* the sample of one-channel images: whole byte, or 1 / 2 / 4 bits, most significant first */
                if (depth < 8) { const int per = 8 / depth, sh = (per - 1 - x % per) * depth; v1 = (src[x / per] >> sh) & ((1 << depth) - 1); }
                if (ctype == 0) dst[x] = (uint8_t)(depth < 8 ? v1 * 255 / ((1 << depth) - 1) : v1);
                else if (ctype == 2) { dst[3 * x] = src[3 * x + 2]; dst[3 * x + 1] = src[3 * x + 1]; dst[3 * x + 2] = src[3 * x]; }
                else if (ctype == 3) { const uint8_t* p = pal[v1]; dst[3 * x] = p[2]; dst[3 * x + 1] = p[1]; dst[3 * x + 2] = p[0]; }
                else { dst[4 * x] = src[4 * x + 2]; dst[4 * x + 1] = src[4 * x + 1]; dst[4 * x + 2] = src[4 * x]; dst[4 * x + 3] = src[4 * x + 3]; }
            }
        }
        free(raw);
        rc = ok ? 0 : 1;
    }
done:
    free(z);
    return rc;
}
