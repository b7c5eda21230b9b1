// npz_io.h -- minimal .npz reader / writer (stored zip entries of .npy arrays, C order, little endian) for the optional
// OpenCV dumper.  Reads what numpy.savez writes (ZIP_STORED, with or without zip64 extra fields); writes files numpy.load
// reads.  No dependencies beyond the C++17 standard library.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace npz {

struct Array {
    std::string dtype;                 // "<f4", "<f8", "<i4", "|u1", ...
    std::vector<size_t> shape;
    std::vector<unsigned char> bytes;
    size_t count() const { size_t n = 1; for (size_t s : shape) n *= s; return n; }
    size_t itemsize() const { return (size_t)std::stoi(dtype.substr(2)); }
    template <class T> const T* as() const { if (sizeof(T) != itemsize()) throw std::runtime_error("npz: element size mismatch for " + dtype); return reinterpret_cast<const T*>(bytes.data()); }
};
typedef std::map<std::string, Array> File;

template <class T> struct dtype_of;
template <> struct dtype_of<float> { static const char* s() { return "<f4"; } };
template <> struct dtype_of<double> { static const char* s() { return "<f8"; } };
template <> struct dtype_of<int32_t> { static const char* s() { return "<i4"; } };
template <> struct dtype_of<uint8_t> { static const char* s() { return "|u1"; } };
template <> struct dtype_of<uint32_t> { static const char* s() { return "<u4"; } };

template <class T>
inline Array make(const T* data, std::vector<size_t> shape)
{
    Array a; a.dtype = dtype_of<T>::s(); a.shape = shape;
    a.bytes.resize(a.count() * sizeof(T));
    if (a.count()) memcpy(a.bytes.data(), data, a.bytes.size());
    return a;
}

inline uint32_t crc32(const unsigned char* p, size_t n)
{
    static uint32_t table[256]; static bool init = false;
    if (!init) { for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; } init = true; }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

inline std::vector<unsigned char> npy_encode(const Array& a)
{
    std::ostringstream d;
    d << "{'descr': '" << a.dtype << "', 'fortran_order': False, 'shape': (";
    for (size_t i = 0; i < a.shape.size(); i++) d << a.shape[i] << (a.shape.size() == 1 || i + 1 < a.shape.size() ? "," : "") << (i + 1 < a.shape.size() ? " " : "");
    d << "), }";
    std::string dict = d.str();
    size_t total = 10 + dict.size() + 1;
    size_t pad = (64 - total % 64) % 64;
    dict += std::string(pad, ' ') + "\n";
    std::vector<unsigned char> out;
    const unsigned char magic[8] = { 0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0 };
    out.insert(out.end(), magic, magic + 8);
    out.push_back((unsigned char)(dict.size() & 0xFF)); out.push_back((unsigned char)(dict.size() >> 8));
    out.insert(out.end(), dict.begin(), dict.end());
    out.insert(out.end(), a.bytes.begin(), a.bytes.end());
    return out;
}

inline Array npy_decode(const unsigned char* p, size_t n)
{
    if (n < 10 || p[0] != 0x93 || memcmp(p + 1, "NUMPY", 5) != 0) throw std::runtime_error("npz: not an .npy member");
    size_t hlen, off;
    if (p[6] == 1) { hlen = p[8] | (p[9] << 8); off = 10; } else { hlen = p[8] | (p[9] << 8) | (p[10] << 16) | ((size_t)p[11] << 24); off = 12; }
    std::string h(reinterpret_cast<const char*>(p + off), hlen);
    Array a;
    size_t d0 = h.find("'descr':"); d0 = h.find('\'', d0 + 8); size_t d1 = h.find('\'', d0 + 1);
    a.dtype = h.substr(d0 + 1, d1 - d0 - 1);
    if (h.find("'fortran_order': True") != std::string::npos) throw std::runtime_error("npz: Fortran order is not supported");
    size_t s0 = h.find('(', h.find("'shape':")), s1 = h.find(')', s0);
    std::string sh = h.substr(s0 + 1, s1 - s0 - 1);
    std::stringstream ss(sh); std::string tok;
    while (std::getline(ss, tok, ',')) { size_t b = tok.find_first_not_of(' '); if (b != std::string::npos) a.shape.push_back((size_t)std::stoull(tok.substr(b))); }
    a.bytes.assign(p + off + hlen, p + n);
    if (a.dtype == "|b1") a.dtype = "|u1";
    if (a.bytes.size() != a.count() * a.itemsize()) throw std::runtime_error("npz: size mismatch in member");
    return a;
}

inline uint32_t rd32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint64_t rd64(const unsigned char* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

inline File load(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("npz: cannot open " + path);
    std::vector<unsigned char> z((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    File out;
    size_t pos = 0;
    while (pos + 30 <= z.size() && rd32(&z[pos]) == 0x04034b50u) {
        const unsigned char* h = &z[pos];
        const uint16_t flags = rd16(h + 6), method = rd16(h + 8), nlen = rd16(h + 26), xlen = rd16(h + 28);
        uint64_t csize = rd32(h + 18), usize = rd32(h + 22);
        if (method != 0) throw std::runtime_error("npz: compressed members are not supported (use numpy.savez, not savez_compressed)");
        if (flags & 8) throw std::runtime_error("npz: streamed members (data descriptor) are not supported");
        std::string name(reinterpret_cast<const char*>(h + 30), nlen);
        const unsigned char* x = h + 30 + nlen;
        for (size_t o = 0; o + 4 <= xlen;) {                    // zip64 extended information: real sizes
            const uint16_t id = rd16(x + o), sz = rd16(x + o + 2);
            if (id == 1 && sz >= 16) { usize = rd64(x + o + 4); csize = rd64(x + o + 12); }
            o += 4 + sz;
        }
        const size_t data = pos + 30 + nlen + xlen;
        if (data + csize > z.size()) throw std::runtime_error("npz: truncated file");
        if (name.size() > 4 && name.substr(name.size() - 4) == ".npy") name.resize(name.size() - 4);
        out[name] = npy_decode(&z[data], (size_t)usize);
        pos = data + (size_t)csize;
    }
    if (out.empty()) throw std::runtime_error("npz: no members in " + path);
    return out;
}

inline void save(const std::string& path, const File& file)
{
    std::vector<unsigned char> z, cd;
    auto p16 = [](std::vector<unsigned char>& v, uint16_t x) { v.push_back(x & 0xFF); v.push_back(x >> 8); };
    auto p32 = [](std::vector<unsigned char>& v, uint32_t x) { for (int i = 0; i < 4; i++) v.push_back((x >> (8 * i)) & 0xFF); };
    uint16_t count = 0;
    for (const auto& kv : file) {
        const std::string name = kv.first + ".npy";
        const std::vector<unsigned char> body = npy_encode(kv.second);
        if (body.size() > 0xFFFFFFF0ull) throw std::runtime_error("npz: member too large");
        const uint32_t crc = crc32(body.data(), body.size()), off = (uint32_t)z.size(), sz = (uint32_t)body.size();
        p32(z, 0x04034b50u); p16(z, 20); p16(z, 0); p16(z, 0); p16(z, 0); p16(z, 0x21); p32(z, crc); p32(z, sz); p32(z, sz);
        p16(z, (uint16_t)name.size()); p16(z, 0);
        z.insert(z.end(), name.begin(), name.end());
        z.insert(z.end(), body.begin(), body.end());
        p32(cd, 0x02014b50u); p16(cd, 20); p16(cd, 20); p16(cd, 0); p16(cd, 0); p16(cd, 0); p16(cd, 0x21); p32(cd, crc); p32(cd, sz); p32(cd, sz);
        p16(cd, (uint16_t)name.size()); p16(cd, 0); p16(cd, 0); p16(cd, 0); p16(cd, 0); p32(cd, 0); p32(cd, off);
        cd.insert(cd.end(), name.begin(), name.end());
        count++;
    }
    const uint32_t cd_off = (uint32_t)z.size(), cd_size = (uint32_t)cd.size();
    z.insert(z.end(), cd.begin(), cd.end());
    p32(z, 0x06054b50u); p16(z, 0); p16(z, 0); p16(z, count); p16(z, count); p32(z, cd_size); p32(z, cd_off); p16(z, 0);
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("npz: cannot write " + path);
    f.write(reinterpret_cast<const char*>(z.data()), (std::streamsize)z.size());
}

}  // namespace npz
