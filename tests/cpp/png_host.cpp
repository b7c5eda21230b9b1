// png_host.cpp -- CPU test driver for the host half of the product's PNG path (ergo_uvo_amd/csrc/uvo_png.h: container, zlib
// inflate, scanline filters).  No GPU involved: the header is plain C++.
//   usage: png_host <in.png> <out.bin>     out: int32 w, h, depth, ctype, row_bytes; then h * row_bytes unfiltered bytes
//   exit 0 ok, 1 refused / damaged (message on stderr), 2 usage
#include <cstdio>
#include <vector>
#include "../../ergo_uvo_amd/csrc/uvo_png.h"

int main(int argc, char** argv)
{
    if (argc != 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint8_t> data;
    uint8_t buf[65536];
    size_t k;
    while ((k = fread(buf, 1, sizeof(buf), f)) > 0) data.insert(data.end(), buf, buf + k);
    fclose(f);
    uvo::png::Header hd;
    std::vector<uint8_t> idat;
    std::string err;
    if (!uvo::png::parse(data.data(), data.size(), &hd, &idat, false, &err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    const size_t stride = uvo::png::row_bytes(hd);
    std::vector<uint8_t> rows(stride * (size_t)hd.h);
    if (!uvo::png::scanlines(hd, idat, rows.data(), &err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    FILE* o = fopen(argv[2], "wb");
    if (!o) return 2;
    const int hdr[5] = { hd.w, hd.h, hd.depth, hd.ctype, (int)stride };
    fwrite(hdr, sizeof(int), 5, o);
    fwrite(rows.data(), 1, rows.size(), o);
    fclose(o);
    return 0;
}
