// uvo_png.h -- host half of the PNG path of uvo_decode_image (codec.hip): container, zlib / DEFLATE and scanline filters.
// cv_bridge::toCvCopy(CompressedImage) = cv::imdecode(IMREAD_UNCHANGED) (uvo_libraries/src/math_utility.cpp:154-173) also meets PNG
// payloads ("...; png compressed").  A DEFLATE stream is sequential by construction (variable-length codes, back-references into
// what was just produced), and PNG's Sub / Average / Paeth filters chain every byte to its left and upper neighbours: both run on
// the host, one pass each; what is per-sample -- unpacking 1 / 2 / 4-bit samples, the palette, RGB(A) -> BGR(A) -- runs on the
// device (k_png_expand).  Decoding is fixed by the specification (RFC 2083, RFC 1950, RFC 1951); the result is checked against
// Pillow's decoder byte for byte (tests/test_codec.py).
#pragma once
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>

namespace uvo { namespace png {

struct Header { int w = 0, h = 0, depth = 0, ctype = -1; bool has_plte = false, has_trns = false; uint8_t pal[256][4]; };

// Canonical prefix code, LSB-first as DEFLATE packs it: a 10-bit first-level table (symbol | length << 9), longer codes by a walk
struct Code {
    static const int kFast = 10;
    uint16_t fast[1 << kFast];
    uint16_t count[16], symbol[288];
    int build(const uint8_t* len, int n)              // < 0 over-subscribed, 0 complete (or empty), > 0 incomplete
    {
        memset(count, 0, sizeof(count)); memset(fast, 0, sizeof(fast));
        for (int i = 0; i < n; i++) count[len[i]]++;
        if (count[0] == n) return 0;
        int left = 1;
        for (int l = 1; l < 16; l++) { left = (left << 1) - count[l]; if (left < 0) return -1; }
        uint16_t offs[16]; offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (int i = 0; i < n; i++) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
        // first-level table: the canonical code of every symbol of length <= kFast, bit-reversed, replicated over the unused high bits
        int code = 0, idx = 0;
        for (int l = 1; l <= kFast; l++) {
            for (int k = 0; k < count[l]; k++, code++, idx++) {
                int rev = 0;
                for (int b = 0; b < l; b++) rev |= ((code >> b) & 1) << (l - 1 - b);
                for (int f = rev; f < (1 << kFast); f += 1 << l) fast[f] = (uint16_t)(symbol[idx] | (l << 9));
            }
            code <<= 1;
        }
        return left;
    }
};

struct Bits {
    const uint8_t* p; size_t n, pos = 0; uint64_t acc = 0; int cnt = 0; bool over = false;
    Bits(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
    inline void fill() { while (cnt <= 56) { uint64_t b = 0; if (pos < n) b = p[pos]; else if (pos > n + 8) over = true; pos++; acc |= b << cnt; cnt += 8; } }
    inline uint32_t peek(int k) { if (cnt < k) fill(); return (uint32_t)(acc & ((1ull << k) - 1)); }
    inline void drop(int k) { acc >>= k; cnt -= k; }
    inline uint32_t get(int k) { if (k == 0) return 0; const uint32_t v = peek(k); drop(k); return v; }
    inline int decode(const Code& c)
    {
        if (cnt < 15) fill();
        const uint16_t f = c.fast[acc & ((1u << Code::kFast) - 1)];
        if (f) { drop(f >> 9); return f & 511; }
        int code = 0, first = 0, index = 0;                       // longer than the table: the canonical walk, a bit at a time
        for (int l = 1; l < 16; l++) {
            code |= (int)((acc >> (l - 1)) & 1);
            const int c_l = c.count[l];
            if (code - c_l < first) { drop(l); return c.symbol[index + (code - first)]; }
            index += c_l; first += c_l; first <<= 1; code <<= 1;
        }
        return -1;
    }
    // bytes consumed so far, counting only whole bytes still in the accumulator as unread
    size_t byte_pos() const { return pos - (size_t)(cnt / 8); }
};

inline bool inflate(const uint8_t* z, size_t zn, uint8_t* out, size_t cap, size_t* produced, size_t* consumed)
{
    static const uint16_t len_base[29] = { 3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258 };
    static const uint8_t len_extra[29] = { 0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0 };
    static const uint16_t dist_base[30] = { 1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577 };
    static const uint8_t dist_extra[30] = { 0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13 };
    Bits b(z, zn);
    size_t o = 0;
    Code cl, cd;
    bool last;
    do {
        last = b.get(1) != 0;
        const int type = (int)b.get(2);
        if (type == 0) {
            b.drop(b.cnt & 7);                                        // to the byte boundary
            const uint32_t len = b.get(16), nlen = b.get(16);
            if ((len ^ 0xFFFFu) != nlen || o + len > cap) return false;
            for (uint32_t i = 0; i < len; i++) out[o++] = (uint8_t)b.get(8);
        } else if (type == 1 || type == 2) {
            uint8_t len[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; i++) len[i] = 8;
                for (; i < 256; i++) len[i] = 9;
                for (; i < 280; i++) len[i] = 7;
                for (; i < 288; i++) len[i] = 8;
                cl.build(len, 288);
                for (i = 0; i < 30; i++) len[i] = 5;
                cd.build(len, 30);
            } else {
                static const uint8_t order[19] = { 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15 };
                const int nl = (int)b.get(5) + 257, nd = (int)b.get(5) + 1, nc = (int)b.get(4) + 4;
                if (nl > 286 || nd > 30) return false;
                uint8_t clen[19] = {0};
                for (int i = 0; i < nc; i++) clen[order[i]] = (uint8_t)b.get(3);
                Code cc;
                if (cc.build(clen, 19) != 0) return false;
                int idx = 0;
                while (idx < nl + nd) {
                    const int sym = b.decode(cc);
                    if (sym < 0 || b.over) return false;
                    if (sym < 16) { len[idx++] = (uint8_t)sym; continue; }
                    int rep, val = 0;
                    if (sym == 16) { if (idx == 0) return false; val = len[idx - 1]; rep = 3 + (int)b.get(2); }
                    else if (sym == 17) rep = 3 + (int)b.get(3);
                    else rep = 11 + (int)b.get(7);
                    if (idx + rep > nl + nd) return false;
                    while (rep--) len[idx++] = (uint8_t)val;
                }
                if (len[256] == 0) return false;
                int e = cl.build(len, nl);
                if (e < 0 || (e > 0 && nl - cl.count[0] != 1)) return false;
                e = cd.build(len + nl, nd);
                if (e < 0 || (e > 0 && nd - cd.count[0] != 1)) return false;
            }
            for (;;) {
                const int sym = b.decode(cl);
                if (sym < 0 || b.over) return false;
                if (sym < 256) { if (o >= cap) return false; out[o++] = (uint8_t)sym; continue; }
                if (sym == 256) break;
                const int s = sym - 257;
                if (s >= 29) return false;
                const int length = len_base[s] + (int)b.get(len_extra[s]);
                const int ds = b.decode(cd);
                if (ds < 0 || ds >= 30) return false;
                const size_t dist = dist_base[ds] + b.get(dist_extra[ds]);
                if (dist > o || o + (size_t)length > cap) return false;
                const uint8_t* src = out + o - dist;
                for (int i = 0; i < length; i++) out[o + i] = src[i];       // byte by byte: the ranges may overlap
                o += (size_t)length;
            }
        } else return false;
        if (b.over) return false;
    } while (!last);
    *produced = o; *consumed = b.byte_pos();
    return true;
}

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline uint32_t crc32(const uint8_t* p, size_t n)
{
    struct Table { uint32_t t[256]; Table() { for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u))); t[i] = c; } } };
    static const Table tab;                         // function-local static: initialised once, thread-safe (several contexts decode at once)
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = tab.t[(c ^ p[i]) & 255] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
inline bool is_png(const uint8_t* d, size_t n) { static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n' }; return n >= 8 && memcmp(d, sig, 8) == 0; }

// Walks the chunks: header, palette, and (unless headers_only) the concatenated IDAT payload.  Error text in *err.
inline bool parse(const uint8_t* data, size_t n, Header* hd, std::vector<uint8_t>* idat, bool headers_only, std::string* err)
{
    auto bad = [&](const char* m) { *err = std::string("PNG: ") + m; return false; };
    if (!is_png(data, n)) return bad("no signature");
    size_t pos = 8;
    bool seen = false;
    for (;;) {
        if (pos + 12 > n) return bad("truncated");
        const uint32_t len = be32(data + pos);
        const uint8_t* type = data + pos + 4;
        if (len > n || pos + 12 + (size_t)len > n) return bad("truncated chunk");
        if (crc32(type, 4 + (size_t)len) != be32(data + pos + 8 + len)) return bad("chunk CRC mismatch");
        const uint8_t* d = data + pos + 8;
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13 || seen) return bad("bad IHDR");
            hd->w = (int)be32(d); hd->h = (int)be32(d + 4); hd->depth = d[8]; hd->ctype = d[9];
            if (hd->w <= 0 || hd->h <= 0 || d[10] != 0 || d[11] != 0) return bad("bad IHDR");
            if ((long long)hd->w * hd->h > (1LL << 26)) return bad("images above 64 Mpixel are refused (a damaged header must not ask for gigabytes)");
            if (d[12] != 0) return bad("interlaced (Adam7) files are not supported");
            const int ct = hd->ctype, dp = hd->depth;
            if (!(ct == 0 || ct == 2 || ct == 3 || ct == 6)) return bad("grey + alpha is not supported");
            if (!(dp == 8 || ((ct == 0 || ct == 3) && (dp == 1 || dp == 2 || dp == 4)))) return bad("16-bit samples are not supported");
            seen = true;
        } else if (!seen) return bad("chunk before IHDR");
        else if (!memcmp(type, "PLTE", 4)) {
            if (len % 3 || len > 768) return bad("bad PLTE");
            memset(hd->pal, 0, sizeof(hd->pal));
            for (uint32_t i = 0; i < len / 3; i++) { hd->pal[i][0] = d[3 * i + 2]; hd->pal[i][1] = d[3 * i + 1]; hd->pal[i][2] = d[3 * i]; }      // stored B, G, R
            hd->has_plte = true;
        } else if (!memcmp(type, "tRNS", 4)) hd->has_trns = true;
        else if (!memcmp(type, "IDAT", 4)) { if (headers_only) break; idat->insert(idat->end(), d, d + len); }
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!seen) return bad("no IHDR");
    if (hd->ctype == 3 && !hd->has_plte) return bad("palette image without PLTE");
    if (hd->ctype == 3 && hd->has_trns) return bad("palettes with transparency are not supported");
    return true;
}

inline int channels_in(const Header& h) { return h.ctype == 2 ? 3 : (h.ctype == 6 ? 4 : 1); }
inline int channels_out(const Header& h) { return h.ctype == 0 ? 1 : (h.ctype == 6 ? 4 : 3); }
inline size_t row_bytes(const Header& h) { return ((size_t)h.w * channels_in(h) * h.depth + 7) / 8; }

// zlib stream -> filtered scanlines -> samples: `rows` receives h * row_bytes unfiltered bytes (filter bytes removed)
inline bool scanlines(const Header& h, const std::vector<uint8_t>& z, uint8_t* rows, std::string* err)
{
    auto bad = [&](const char* m) { *err = std::string("PNG: ") + m; return false; };
    if (z.size() < 6 || (z[0] & 0x0F) != 8 || ((z[0] << 8) | z[1]) % 31 != 0 || (z[1] & 0x20)) return bad("bad zlib header");
    const size_t stride = row_bytes(h), raw_n = (stride + 1) * (size_t)h.h;
    std::vector<uint8_t> raw(raw_n);
    size_t produced = 0, consumed = 0;
    if (!inflate(z.data() + 2, z.size() - 2, raw.data(), raw_n, &produced, &consumed) || produced != raw_n) return bad("damaged or short image data");
    uint32_t s1 = 1, s2 = 0;
    for (size_t i = 0; i < raw_n;) {                       // Adler-32, the modulo deferred over blocks of 5552 bytes
        const size_t e = i + 5552 < raw_n ? i + 5552 : raw_n;
        for (; i < e; i++) { s1 += raw[i]; s2 += s1; }
        s1 %= 65521u; s2 %= 65521u;
    }
    if (2 + consumed + 4 > z.size() || be32(z.data() + 2 + consumed) != ((s2 << 16) | s1)) return bad("Adler-32 mismatch");
    const size_t bpp = h.depth < 8 ? 1 : (size_t)channels_in(h);
    for (int y = 0; y < h.h; y++) {
        const uint8_t* src = raw.data() + (size_t)y * (stride + 1);
        uint8_t* cur = rows + (size_t)y * stride;
        const uint8_t* up = y ? cur - stride : nullptr;
        const int ft = src[0];
        src++;
        switch (ft) {
        case 0: memcpy(cur, src, stride); break;
        case 1: for (size_t x = 0; x < stride; x++) cur[x] = (uint8_t)(src[x] + (x >= bpp ? cur[x - bpp] : 0)); break;
        case 2: for (size_t x = 0; x < stride; x++) cur[x] = (uint8_t)(src[x] + (up ? up[x] : 0)); break;
        case 3: for (size_t x = 0; x < stride; x++) cur[x] = (uint8_t)(src[x] + (((x >= bpp ? cur[x - bpp] : 0) + (up ? up[x] : 0)) >> 1)); break;
        case 4:
            for (size_t x = 0; x < stride; x++) {
                const int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
                const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
                cur[x] = (uint8_t)(src[x] + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c)));
            }
            break;
        default: return bad("unknown filter type");
        }
    }
    return true;
}

} }  // namespace uvo::png
