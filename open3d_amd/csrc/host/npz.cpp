// NPZ codec for the MI355X backend's VoxelBlockGrid::Save / Load: what
// t::io::WriteNpz / ReadNpz (cpp/open3d/t/io/NumpyIO.cpp:157-205 header,
// :360-466 WriteNpzOneTensor, :675-757 ReadNpz) do, host-only C++.
//
// Writing follows the reference byte for byte for arrays below 4 GiB: NPY
// format 1.0, dict `{'descr': '<f4', 'fortran_order': False, 'shape': (n,), }`
// padded with spaces so that 10 + len(dict) is a multiple of 16, entries
// stored (method 0), version-needed 20, zeroed time / date, CRC-32 over
// header + data. Where the reference's 32-bit size / offset fields would wrap
// (one array or the archive beyond 4 GiB -- a 2 M-block grid is ~94 GiB) this
// writer switches that entry / the end record to ZIP64, which numpy reads.
//
// Reading is a superset of the reference's sequential local-header walk: it
// goes through the central directory, accepts ZIP64 extra fields (numpy's
// savez always writes them), data descriptors, stored and deflated entries.
// Little-endian, C-order arrays of the dtypes below only, as the reference.

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>

#include "../common.h"
#include "../npz.h"
#include "o3d_mi355x_host.h"

using namespace o3dmi;

namespace o3dmi {
int NpzDtypeSize(int dt) {
    switch (dt) {
        case O3DMI_F32: case O3DMI_I32: case O3DMI_U32: return 4;
        case O3DMI_F64: case O3DMI_I64: case O3DMI_U64: return 8;
        case O3DMI_U16: case O3DMI_I16: return 2;
        case O3DMI_U8: case O3DMI_I8: case O3DMI_BOOL: return 1;
        default: return 0;
    }
}
}  // namespace o3dmi

namespace {

// NumpyIO.cpp:132-155
char DtypeChar(int dt) {
    switch (dt) {
        case O3DMI_F32: case O3DMI_F64: return 'f';
        case O3DMI_I8: case O3DMI_I16: case O3DMI_I32: case O3DMI_I64: return 'i';
        case O3DMI_U8: case O3DMI_U16: case O3DMI_U32: case O3DMI_U64: return 'u';
        case O3DMI_BOOL: return 'b';
        default: return 0;
    }
}

int DtypeFrom(char type, int64_t word) {
    if (type == 'f') return word == 4 ? O3DMI_F32 : word == 8 ? O3DMI_F64 : -1;
    if (type == 'i')
        return word == 1 ? O3DMI_I8 : word == 2 ? O3DMI_I16
             : word == 4 ? O3DMI_I32 : word == 8 ? O3DMI_I64 : -1;
    if (type == 'u')
        return word == 1 ? O3DMI_U8 : word == 2 ? O3DMI_U16
             : word == 4 ? O3DMI_U32 : word == 8 ? O3DMI_U64 : -1;
    if (type == 'b') return word == 1 ? O3DMI_BOOL : -1;
    return -1;
}

template <typename T>
void Put(std::vector<uint8_t>& v, T x) {
    const uint8_t* p = (const uint8_t*)&x;
    v.insert(v.end(), p, p + sizeof(T));
}
void PutStr(std::vector<uint8_t>& v, const std::string& s) {
    v.insert(v.end(), s.begin(), s.end());
}

// CreateNumpyHeader, NumpyIO.cpp:157-205.
std::vector<uint8_t> NpyHeader(const NpzArray& a) {
    std::string shape;
    if (a.shape.empty()) {
        shape = "()";
    } else if (a.shape.size() == 1) {
        shape = "(" + std::to_string(a.shape[0]) + ",)";
    } else {
        shape = "(" + std::to_string(a.shape[0]);
        for (size_t i = 1; i < a.shape.size(); ++i)
            shape += ", " + std::to_string(a.shape[i]);
        shape += ")";
    }
    std::string dict = std::string("{'descr': '<") + DtypeChar(a.dtype) +
                       std::to_string(NpzDtypeSize(a.dtype)) +
                       "', 'fortran_order': False, 'shape': " + shape + ", }";
    const size_t pad = 16 - (10 + dict.size()) % 16 - 1;
    dict.append(pad, ' ');
    dict.push_back('\n');
    std::vector<uint8_t> h;
    h.push_back(0x93);
    PutStr(h, "NUMPY");
    h.push_back(0x01);
    h.push_back(0x00);
    Put<uint16_t>(h, (uint16_t)dict.size());
    PutStr(h, dict);
    return h;
}

bool ForceZip64() { return std::getenv("O3DMI_NPZ_FORCE_ZIP64") != nullptr; }

struct FileCloser {
    FILE* fp;
    ~FileCloser() {
        if (fp) fclose(fp);
    }
};

int Fail(const std::string& msg) {
    SetLastError(msg);
    return O3DMI_ERR_INVALID_ARG;
}

// ParsePropertyDict, NumpyIO.cpp:207-276.
int ParseNpy(const uint8_t* buf, size_t len, NpzArray* out, size_t* data_off) {
    if (len < 10 || buf[0] != 0x93 || std::memcmp(buf + 1, "NUMPY", 5) != 0)
        return Fail("Invalid Numpy preamble.");
    const int major = buf[6];
    size_t hlen, hoff;
    if (major == 1) {
        hlen = (size_t)buf[8] | ((size_t)buf[9] << 8);
        hoff = 10;
    } else if (major == 2 || major == 3) {
        if (len < 12) return Fail("Truncated .npy header.");
        hlen = (size_t)buf[8] | ((size_t)buf[9] << 8) | ((size_t)buf[10] << 16) |
               ((size_t)buf[11] << 24);
        hoff = 12;
    } else {
        return Fail("Not supported Numpy format version.");
    }
    if (hoff + hlen > len) return Fail("Truncated .npy header.");
    const std::string h((const char*)buf + hoff, hlen);
    size_t loc = h.find("fortran_order");
    if (loc == std::string::npos)
        return Fail("Failed to find header keyword: 'fortran_order'");
    // the value follows "fortran_order': " (16 characters)
    const bool fortran =
            loc + 20 <= h.size() && h.compare(loc + 16, 4, "True") == 0;
    size_t l1 = h.find('('), l2 = h.find(')');
    if (l1 == std::string::npos || l2 == std::string::npos || l2 < l1)
        return Fail("Failed to find header keyword: '(' or ')'");
    out->shape.clear();
    {
        int64_t cur = -1;
        for (size_t i = l1 + 1; i <= l2; ++i) {
            const char c = h[i];
            if (c >= '0' && c <= '9') {
                cur = (cur < 0 ? 0 : cur) * 10 + (c - '0');
            } else if (cur >= 0) {
                out->shape.push_back(cur);
                cur = -1;
            }
        }
    }
    loc = h.find("descr");
    if (loc == std::string::npos)
        return Fail("Failed to find header keyword: 'descr'");
    loc += 9;
    if (loc + 2 >= h.size()) return Fail("Malformed 'descr'.");
    if (!(h[loc] == '<' || h[loc] == '|'))
        return Fail("Only little endian is supported.");
    const char type = h[loc + 1];
    const int64_t word = std::atoi(h.c_str() + loc + 2);
    out->dtype = DtypeFrom(type, word);
    if (out->dtype < 0) return Fail("Unsupported dtype in .npy header.");
    if (fortran && out->shape.size() > 1)
        return Fail("Fortran-order arrays are not supported.");
    *data_off = hoff + hlen;
    return O3DMI_OK;
}

uint16_t R16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t R32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) |
           ((uint32_t)p[3] << 24);
}
uint64_t R64(const uint8_t* p) { return (uint64_t)R32(p) | ((uint64_t)R32(p + 4) << 32); }

bool ReadAt(FILE* fp, uint64_t off, void* dst, size_t n) {
    if (fseeko(fp, (off_t)off, SEEK_SET) != 0) return false;
    return fread(dst, 1, n, fp) == n;
}

}  // namespace

extern "C" {

int o3dmi_npz_create(o3dmi_npz_t** out) {
    O3DMI_REQUIRE(out != nullptr, "out is null");
    *out = new o3dmi_npz();
    return O3DMI_OK;
}

int o3dmi_npz_destroy(o3dmi_npz_t* z) {
    delete z;
    return O3DMI_OK;
}

int o3dmi_npz_add(o3dmi_npz_t* z, const char* name, int dtype, int ndim,
                  const int64_t* shape, const void* data_host) {
    O3DMI_REQUIRE(z && name, "null argument");
    O3DMI_REQUIRE(NpzDtypeSize(dtype) > 0, "Unsupported dtype");
    O3DMI_REQUIRE(ndim >= 0 && ndim <= 8 && (ndim == 0 || shape), "bad shape");
    NpzArray a;
    a.name = name;
    a.dtype = dtype;
    for (int i = 0; i < ndim; ++i) {
        O3DMI_REQUIRE(shape[i] >= 0, "negative dimension");
        a.shape.push_back(shape[i]);
    }
    const size_t bytes = (size_t)a.NumElements() * NpzDtypeSize(dtype);
    O3DMI_REQUIRE(bytes == 0 || data_host != nullptr, "data is null");
    a.data.resize(bytes);
    if (bytes) std::memcpy(a.data.data(), data_host, bytes);
    for (auto& e : z->arrays)
        if (e.name == a.name) {
            e = std::move(a);
            return O3DMI_OK;
        }
    z->arrays.push_back(std::move(a));
    return O3DMI_OK;
}

int o3dmi_npz_count(const o3dmi_npz_t* z) {
    return z ? (int)z->arrays.size() : 0;
}

const char* o3dmi_npz_name(const o3dmi_npz_t* z, int i) {
    if (!z || i < 0 || i >= (int)z->arrays.size()) return nullptr;
    return z->arrays[(size_t)i].name.c_str();
}

int o3dmi_npz_get(const o3dmi_npz_t* z, const char* name, int* dtype, int* ndim,
                  int64_t* shape8, const void** data_host) {
    O3DMI_REQUIRE(z && name, "null argument");
    const NpzArray* a = z->Find(name);
    if (!a) return Fail(std::string("array not found in npz: ") + name);
    if (dtype) *dtype = a->dtype;
    if (ndim) *ndim = (int)a->shape.size();
    if (shape8)
        for (size_t i = 0; i < a->shape.size() && i < 8; ++i) shape8[i] = a->shape[i];
    if (data_host) *data_host = a->data.data();
    return O3DMI_OK;
}

// WriteNpz, NumpyIO.cpp:759-789 (one pass instead of re-opening per tensor).
int o3dmi_npz_write(const o3dmi_npz_t* z, const char* file_name) {
    O3DMI_REQUIRE(z && file_name, "null argument");
    FILE* fp = fopen(file_name, "wb");
    if (!fp) return Fail(std::string("Failed to open file ") + file_name);
    FileCloser closer{fp};
    std::vector<uint8_t> central;
    uint64_t offset = 0;
    const bool force64 = ForceZip64();
    bool any64 = false;
    for (const NpzArray& a : z->arrays) {
        const std::vector<uint8_t> npy = NpyHeader(a);
        const uint64_t nbytes = (uint64_t)a.data.size() + npy.size();
        uint32_t crc = (uint32_t)crc32(0L, npy.data(), (uInt)npy.size());
        {
            // crc32 takes 32-bit lengths: feed large arrays in pieces.
            size_t done = 0;
            while (done < a.data.size()) {
                const size_t n = std::min<size_t>(a.data.size() - done, 1u << 30);
                crc = (uint32_t)crc32(crc, a.data.data() + done, (uInt)n);
                done += n;
            }
        }
        const std::string var_name = a.name + ".npy";
        const bool big = force64 || nbytes >= 0xFFFFFFFFull;
        const bool far = force64 || offset >= 0xFFFFFFFFull;
        any64 = any64 || big || far;
        std::vector<uint8_t> local;
        PutStr(local, "PK");
        Put<uint16_t>(local, 0x0403);
        Put<uint16_t>(local, big ? 45 : 20);  // version needed to extract
        Put<uint16_t>(local, 0);               // general purpose bit flag
        Put<uint16_t>(local, 0);               // compression method: stored
        Put<uint16_t>(local, 0);               // file last mod time
        Put<uint16_t>(local, 0);               // file last mod date
        Put<uint32_t>(local, crc);
        Put<uint32_t>(local, big ? 0xFFFFFFFFu : (uint32_t)nbytes);
        Put<uint32_t>(local, big ? 0xFFFFFFFFu : (uint32_t)nbytes);
        Put<uint16_t>(local, (uint16_t)var_name.size());
        Put<uint16_t>(local, big ? 20 : 0);    // extra field length
        PutStr(local, var_name);
        if (big) {
            Put<uint16_t>(local, 0x0001);
            Put<uint16_t>(local, 16);
            Put<uint64_t>(local, nbytes);  // uncompressed
            Put<uint64_t>(local, nbytes);  // compressed
        }
        // central directory record
        std::vector<uint8_t> extra;
        if (big || far) {
            Put<uint16_t>(extra, 0x0001);
            Put<uint16_t>(extra, (uint16_t)((big ? 16 : 0) + (far ? 8 : 0)));
            if (big) {
                Put<uint64_t>(extra, nbytes);
                Put<uint64_t>(extra, nbytes);
            }
            if (far) Put<uint64_t>(extra, offset);
        }
        PutStr(central, "PK");
        Put<uint16_t>(central, 0x0201);
        Put<uint16_t>(central, (big || far) ? 45 : 20);  // version made by
        Put<uint16_t>(central, (big || far) ? 45 : 20);  // version needed
        Put<uint16_t>(central, 0);
        Put<uint16_t>(central, 0);
        Put<uint16_t>(central, 0);
        Put<uint16_t>(central, 0);
        Put<uint32_t>(central, crc);
        Put<uint32_t>(central, big ? 0xFFFFFFFFu : (uint32_t)nbytes);
        Put<uint32_t>(central, big ? 0xFFFFFFFFu : (uint32_t)nbytes);
        Put<uint16_t>(central, (uint16_t)var_name.size());
        Put<uint16_t>(central, (uint16_t)extra.size());
        Put<uint16_t>(central, 0);  // file comment length
        Put<uint16_t>(central, 0);  // disk number where file starts
        Put<uint16_t>(central, 0);  // internal file attributes
        Put<uint32_t>(central, 0);  // external file attributes
        Put<uint32_t>(central, far ? 0xFFFFFFFFu : (uint32_t)offset);
        PutStr(central, var_name);
        central.insert(central.end(), extra.begin(), extra.end());

        if (fwrite(local.data(), 1, local.size(), fp) != local.size() ||
            fwrite(npy.data(), 1, npy.size(), fp) != npy.size() ||
            (a.data.size() &&
             fwrite(a.data.data(), 1, a.data.size(), fp) != a.data.size()))
            return Fail(std::string("write failed: ") + file_name);
        offset += local.size() + nbytes;
    }
    const uint64_t n = z->arrays.size();
    const uint64_t cd_off = offset, cd_size = central.size();
    std::vector<uint8_t> tail;
    const bool eocd64 = any64 || n >= 0xFFFF || cd_off >= 0xFFFFFFFFull ||
                        cd_size >= 0xFFFFFFFFull;
    if (eocd64) {
        PutStr(tail, "PK");
        Put<uint16_t>(tail, 0x0606);
        Put<uint64_t>(tail, 44);
        Put<uint16_t>(tail, 45);
        Put<uint16_t>(tail, 45);
        Put<uint32_t>(tail, 0);
        Put<uint32_t>(tail, 0);
        Put<uint64_t>(tail, n);
        Put<uint64_t>(tail, n);
        Put<uint64_t>(tail, cd_size);
        Put<uint64_t>(tail, cd_off);
        PutStr(tail, "PK");
        Put<uint16_t>(tail, 0x0706);
        Put<uint32_t>(tail, 0);
        Put<uint64_t>(tail, cd_off + cd_size);
        Put<uint32_t>(tail, 1);
    }
    PutStr(tail, "PK");
    Put<uint16_t>(tail, 0x0605);
    Put<uint16_t>(tail, 0);
    Put<uint16_t>(tail, 0);
    Put<uint16_t>(tail, (uint16_t)(n >= 0xFFFF ? 0xFFFF : n));
    Put<uint16_t>(tail, (uint16_t)(n >= 0xFFFF ? 0xFFFF : n));
    Put<uint32_t>(tail, cd_size >= 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cd_size);
    Put<uint32_t>(tail, (eocd64 && (force64 || cd_off >= 0xFFFFFFFFull))
                                ? 0xFFFFFFFFu
                                : (uint32_t)cd_off);
    Put<uint16_t>(tail, 0);
    if (fwrite(central.data(), 1, central.size(), fp) != central.size() ||
        fwrite(tail.data(), 1, tail.size(), fp) != tail.size())
        return Fail(std::string("write failed: ") + file_name);
    return O3DMI_OK;
}

// ReadNpz, NumpyIO.cpp:675-757. Nothing may throw across the C ABI: a
// malformed file can still make a container allocation fail.
static int NpzReadImpl(const char* file_name, o3dmi_npz_t** out);
int o3dmi_npz_read(const char* file_name, o3dmi_npz_t** out) {
    O3DMI_REQUIRE(file_name && out, "null argument");
    try {
        return NpzReadImpl(file_name, out);
    } catch (const std::exception& e) {
        return Fail(std::string("Malformed npz file: ") + e.what());
    } catch (...) {
        return Fail("Malformed npz file.");
    }
}
}  // extern "C"

static int NpzReadImpl(const char* file_name, o3dmi_npz_t** out) {
    FILE* fp = fopen(file_name, "rb");
    if (!fp) return Fail(std::string("Failed to open file ") + file_name);
    FileCloser closer{fp};
    if (fseeko(fp, 0, SEEK_END) != 0) return Fail("seek failed");
    const uint64_t fsize = (uint64_t)ftello(fp);
    if (fsize < 22) return Fail("Not a zip archive (too short).");
    // End-of-central-directory record: scan the last 64 KiB + 22 bytes.
    const uint64_t scan = std::min<uint64_t>(fsize, 65557);
    std::vector<uint8_t> tail((size_t)scan);
    if (!ReadAt(fp, fsize - scan, tail.data(), (size_t)scan))
        return Fail("read failed");
    int64_t eocd = -1;
    for (int64_t i = (int64_t)scan - 22; i >= 0; --i)
        if (tail[(size_t)i] == 'P' && tail[(size_t)i + 1] == 'K' &&
            tail[(size_t)i + 2] == 5 && tail[(size_t)i + 3] == 6) {
            eocd = i;
            break;
        }
    if (eocd < 0) return Fail("Unsupported zip footer.");
    const uint8_t* e = tail.data() + eocd;
    uint64_t nrecs = R16(e + 10), cd_size = R32(e + 12), cd_off = R32(e + 16);
    if (nrecs == 0xFFFF || cd_size == 0xFFFFFFFFu || cd_off == 0xFFFFFFFFu) {
        // ZIP64 locator sits right before the EOCD.
        if (eocd < 20) return Fail("Unsupported zip footer.");
        const uint8_t* loc = e - 20;
        if (!(loc[0] == 'P' && loc[1] == 'K' && loc[2] == 6 && loc[3] == 7))
            return Fail("Unsupported zip footer.");
        const uint64_t e64_off = R64(loc + 8);
        uint8_t rec[56];
        if (!ReadAt(fp, e64_off, rec, 56) ||
            !(rec[0] == 'P' && rec[1] == 'K' && rec[2] == 6 && rec[3] == 6))
            return Fail("Unsupported zip footer.");
        nrecs = R64(rec + 32);
        cd_size = R64(rec + 40);
        cd_off = R64(rec + 48);
    }
    if (cd_off + cd_size > fsize) return Fail("Corrupt central directory.");
    std::vector<uint8_t> cd((size_t)cd_size);
    if (cd_size && !ReadAt(fp, cd_off, cd.data(), (size_t)cd_size))
        return Fail("read failed");
    auto* z = new o3dmi_npz();
    struct Guard {
        o3dmi_npz* z;
        ~Guard() { delete z; }
    } guard{z};
    size_t p = 0;
    for (uint64_t r = 0; r < nrecs; ++r) {
        if (p + 46 > cd.size() || R32(cd.data() + p) != 0x02014b50u)
            return Fail("Corrupt central directory.");
        const uint8_t* c = cd.data() + p;
        const uint16_t method = R16(c + 10);
        uint64_t csize = R32(c + 20), usize = R32(c + 24);
        const uint16_t nlen = R16(c + 28), xlen = R16(c + 30), clen = R16(c + 32);
        uint64_t lho = R32(c + 42);
        if (p + 46 + nlen + xlen + clen > cd.size())
            return Fail("Corrupt central directory.");
        std::string name((const char*)c + 46, nlen);
        // ZIP64 extended information
        const uint8_t* x = c + 46 + nlen;
        size_t xp = 0;
        while (xp + 4 <= xlen) {
            const uint16_t id = R16(x + xp), sz = R16(x + xp + 2);
            if (id == 0x0001) {
                size_t q = xp + 4;
                if (usize == 0xFFFFFFFFu && q + 8 <= xp + 4 + sz) { usize = R64(x + q); q += 8; }
                if (csize == 0xFFFFFFFFu && q + 8 <= xp + 4 + sz) { csize = R64(x + q); q += 8; }
                if (lho == 0xFFFFFFFFu && q + 8 <= xp + 4 + sz) { lho = R64(x + q); q += 8; }
            }
            xp += 4 + (size_t)sz;
        }
        p += 46 + (size_t)nlen + xlen + clen;
        uint8_t lh[30];
        if (!ReadAt(fp, lho, lh, 30) || R32(lh) != 0x04034b50u)
            return Fail("Failed to read local header in npz.");
        const uint64_t data_off = lho + 30 + R16(lh + 26) + R16(lh + 28);
        if (data_off + csize > fsize) return Fail("Corrupt npz entry.");
        // deflate expands by at most ~1032:1; anything larger is not a size
        // this entry can have (and must not drive the allocation below)
        if (usize > (method == 0 ? csize : csize * 1032 + 65536))
            return Fail("Corrupt npz entry (uncompressed size).");
        std::vector<uint8_t> raw((size_t)usize);
        if (method == 0) {
            if (csize != usize) return Fail("Corrupt stored entry.");
            if (usize && !ReadAt(fp, data_off, raw.data(), (size_t)usize))
                return Fail("Failed to read npy data.");
        } else if (method == 8) {
            std::vector<uint8_t> comp((size_t)csize);
            if (csize && !ReadAt(fp, data_off, comp.data(), (size_t)csize))
                return Fail("Failed to read compressed data.");
            z_stream zs;
            std::memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -MAX_WBITS) != Z_OK)
                return Fail("Failed to decompress data.");
            size_t in_done = 0, out_done = 0;
            int err = Z_OK;
            while (err == Z_OK) {
                if (zs.avail_in == 0 && in_done < comp.size()) {
                    const size_t n = std::min<size_t>(comp.size() - in_done, 1u << 30);
                    zs.next_in = comp.data() + in_done;
                    zs.avail_in = (uInt)n;
                    in_done += n;
                }
                if (zs.avail_out == 0 && out_done < raw.size()) {
                    const size_t n = std::min<size_t>(raw.size() - out_done, 1u << 30);
                    zs.next_out = raw.data() + out_done;
                    zs.avail_out = (uInt)n;
                    out_done += n;
                }
                err = inflate(&zs, Z_NO_FLUSH);
                if (err == Z_BUF_ERROR && zs.avail_out == 0 &&
                    out_done == raw.size())
                    break;
            }
            inflateEnd(&zs);
            if (err != Z_STREAM_END || zs.total_out != raw.size())
                return Fail("Failed to decompress data.");
        } else {
            return Fail("Unsupported zip compression method.");
        }
        NpzArray a;
        // The ".npy" suffix is removed when an npz is read (NumpyIO.cpp:716).
        a.name = name.size() >= 4 && name.compare(name.size() - 4, 4, ".npy") == 0
                         ? name.substr(0, name.size() - 4)
                         : name;
        size_t off = 0;
        int st = ParseNpy(raw.data(), raw.size(), &a, &off);
        if (st) return st;
        const size_t want = (size_t)a.NumElements() * NpzDtypeSize(a.dtype);
        if (off + want > raw.size()) return Fail("Failed to read npy data.");
        a.data.assign(raw.begin() + (ptrdiff_t)off,
                      raw.begin() + (ptrdiff_t)(off + want));
        z->arrays.push_back(std::move(a));
    }
    guard.z = nullptr;
    *out = z;
    return O3DMI_OK;
}
