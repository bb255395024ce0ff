// On-disk output of the hot path's result: VTK ImageData (.vti), the format rho2sdf() hands to ParaView
// (reference: src/DataExport/ExportToVTI.jl:22-67, WriteVTK's vtk_grid(filename, x, y, z) for three ranges).
// Host code only (no device work); part of the library so that non-Julia hosts get the same artefact.
#include <cstdint>
#include <cstdio>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <string>
#include <strings.h>
#include <vector>
#include <zlib.h>

#include "../../include/rho2sdf_hip.h"
#include "r2s_common.hpp"


extern "C" {

// exportSdfToVTI(filename, grid, values, value_label, smooth)
// dimensions = N (* smooth) + 1, origin = AABB_min, spacing = cell_size (/ smooth); one point-data array.
// The file is raw-appended VTK XML (header_type UInt64, little endian), which every VTK reader accepts;
// WriteVTK compresses the same payload with zlib, so the bytes differ but the data set is the same.
int r2s_export_vti(const char* filename, const r2s_grid* grid, const void* values, int32_t is_float32, int64_t n_values,
                   const char* value_label, int32_t smooth)
{
    return r2s_export_vti_z(filename, grid, values, is_float32, n_values, value_label, smooth, 0);
}

// the same file with the payload deflated in 1 MiB blocks (compressor="vtkZLibDataCompressor", what WriteVTK writes by
// default): level 0 = raw appended as above, 1..9 = zlib level
int r2s_export_vti_z(const char* filename, const r2s_grid* grid, const void* values, int32_t is_float32, int64_t n_values,
                     const char* value_label, int32_t smooth, int32_t level)
{
    if (!filename || !grid || !values || !value_label) return fail(R2S_ERR_ARG, "null argument");
    if (level < 0 || level > 9) return fail(R2S_ERR_ARG, "compression level must be 0..9");
    if (smooth < 0) return fail(R2S_ERR_ARG, "smooth must be >= 0 (0 = no refinement)");
    const int64_t s = smooth > 0 ? smooth : 1;
    const int64_t dims[3] = {grid->N[0] * s + 1, grid->N[1] * s + 1, grid->N[2] * s + 1};
    const int64_t n = dims[0] * dims[1] * dims[2];
    if (n_values != n)   // ExportToVTI.jl:47-49
        return fail(R2S_ERR_ARG, "Values vector length (%lld) doesn't match grid dimensions (%lld).", (long long)n_values,
                    (long long)n);
    const double spacing = smooth > 0 ? grid->cell_size / (double)smooth : grid->cell_size;
    std::string path(filename);
    if (path.size() < 4 || path.compare(path.size() - 4, 4, ".vti") != 0) path += ".vti";
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return fail(R2S_ERR_ARG, "cannot open %s for writing", path.c_str());
    const uint64_t nbytes = (uint64_t)n * (is_float32 ? 4u : 8u);
    int ok = fprintf(f,
                     "<?xml version=\"1.0\"?>\n"
                     "<VTKFile type=\"ImageData\" version=\"1.0\" byte_order=\"LittleEndian\" header_type=\"UInt64\"%s>\n"
                     "  <ImageData WholeExtent=\"0 %lld 0 %lld 0 %lld\" Origin=\"%.17g %.17g %.17g\" Spacing=\"%.17g %.17g %.17g\">\n"
                     "    <Piece Extent=\"0 %lld 0 %lld 0 %lld\">\n"
                     "      <PointData>\n"
                     "        <DataArray type=\"%s\" Name=\"%s\" format=\"appended\" offset=\"0\"/>\n"
                     "      </PointData>\n"
                     "    </Piece>\n"
                     "  </ImageData>\n"
                     "  <AppendedData encoding=\"raw\">\n_",
                     level > 0 ? " compressor=\"vtkZLibDataCompressor\"" : "", (long long)dims[0] - 1, (long long)dims[1] - 1, (long long)dims[2] - 1, grid->aabb_min[0],
                     grid->aabb_min[1], grid->aabb_min[2], spacing, spacing, spacing, (long long)dims[0] - 1,
                     (long long)dims[1] - 1, (long long)dims[2] - 1, is_float32 ? "Float32" : "Float64", value_label) > 0;
    if (level == 0) {
        ok = ok && fwrite(&nbytes, sizeof nbytes, 1, f) == 1;
        ok = ok && fwrite(values, 1, (size_t)nbytes, f) == (size_t)nbytes;
    } else {
        // [nblocks, block size, size of the last block, compressed size of every block] then the deflated blocks
        const uint64_t bs = 1u << 20, nb = (nbytes + bs - 1) / bs, last = nbytes - (nb - 1) * bs;
        std::vector<uint64_t> hdr(3 + nb);
        hdr[0] = nb; hdr[1] = bs; hdr[2] = (last == bs) ? 0 : last;
        std::vector<std::vector<unsigned char>> blocks(nb);
        for (uint64_t b = 0; b < nb && ok; ++b) {
            const uint64_t len = (b + 1 < nb) ? bs : last;
            uLongf cap = compressBound((uLong)len);
            blocks[b].resize(cap);
            if (compress2(blocks[b].data(), &cap, (const Bytef*)values + b * bs, (uLong)len, level) != Z_OK) ok = 0;
            blocks[b].resize(cap);
            hdr[3 + b] = cap;
        }
        ok = ok && fwrite(hdr.data(), sizeof(uint64_t), hdr.size(), f) == hdr.size();
        for (uint64_t b = 0; b < nb && ok; ++b) ok = fwrite(blocks[b].data(), 1, blocks[b].size(), f) == blocks[b].size();
    }
    ok = ok && fputs("\n  </AppendedData>\n</VTKFile>\n", f) >= 0;
    if (fclose(f) != 0) ok = 0;
    if (!ok) return fail(R2S_ERR_ARG, "write to %s failed", path.c_str());
    return 0;
}

// exportToVTU(fileName, X, IEN, VTK_CODE, rho)                        src/DataExport/ExportToVTU.jl:2-99
// ASCII UnstructuredGrid with the reference's arrays: Points (Float64; |x| < 1e-20 written as 0), connectivity
// (0-based), offsets, types (VTK_CODE: 12 = hexahedron, 10 = tetra), optional point data "density".
// Numbers are written with 17 significant digits (Julia prints the shortest round-trip form; same values).
int r2s_export_vtu(const char* filename, const double* X, int64_t nnp, const int64_t* IEN, int64_t nel, int32_t nen,
                   int32_t vtk_code, const double* rho_n)
{
    if (!filename || !X || !IEN || nnp <= 0 || nel <= 0 || nen <= 0) return fail(R2S_ERR_ARG, "bad argument");
    FILE* f = fopen(filename, "w");
    if (!f) return fail(R2S_ERR_ARG, "cannot open %s for writing", filename);
    fprintf(f, "<VTKFile type=\"UnstructuredGrid\" version=\"0.1\" byte_order=\"LittleEndian\">\n  <UnstructuredGrid>\n");
    fprintf(f, "    <Piece NumberOfPoints=\"%lld\" NumberOfCells=\"%lld\">\n", (long long)nnp, (long long)nel);
    fprintf(f, "\t  <Points>\n        <DataArray type=\"Float64\" NumberOfComponents=\"3\" format=\"ascii\">\n");
    for (int64_t a = 0; a < nnp; ++a) {
        fprintf(f, "          ");
        for (int i = 0; i < 3; ++i) {
            const double v = X[3 * a + i];
            fprintf(f, " %.17g", (v < 1.0e-20 && v > -1.0e-20) ? 0.0 : v);
        }
        fprintf(f, "\n");
    }
    fprintf(f, "        </DataArray>\n\t  </Points>\n      <Cells>\n");
    fprintf(f, "\t\t  <DataArray type=\"Int32\" Name=\"connectivity\" format=\"ascii\">\n");
    for (int64_t el = 0; el < nel; ++el) {
        fprintf(f, "         ");
        for (int a = 0; a < nen; ++a) fprintf(f, " %lld", (long long)(IEN[el * nen + a] - 1));
        fprintf(f, "\n");
    }
    fprintf(f, "        </DataArray>\n        <DataArray type=\"Int32\" Name=\"offsets\" format=\"ascii\">\n");
    for (int64_t el = 1; el <= nel; ++el) fprintf(f, "          %lld\n", (long long)(el * nen));
    fprintf(f, "        </DataArray>\n        <DataArray type=\"UInt8\" Name=\"types\" format=\"ascii\">\n");
    for (int64_t el = 0; el < nel; ++el) fprintf(f, "          %d\n", (int)vtk_code);
    fprintf(f, "        </DataArray>\n      </Cells>\n");
    if (rho_n) {
        fprintf(f, "      <PointData Scalars=\"scalars\">\n           <DataArray type=\"Float32\" Name=\"density\" Format=\"ascii\">\n");
        for (int64_t a = 0; a < nnp; ++a) fprintf(f, "             %.17g\n", rho_n[a]);
        fprintf(f, "           </DataArray>\n      </PointData>\n");
    }
    fprintf(f, "    </Piece>\n  </UnstructuredGrid>\n</VTKFile>\n");
    const int bad = ferror(f);
    if (fclose(f) != 0 || bad) return fail(R2S_ERR_ARG, "write to %s failed", filename);
    return 0;
}

}  // extern "C"

// ---- import_vtu_mesh (src/DataImport/VTUImport.jl:22-112, density field :117-226) -----------------------
// Every DataArray encoding ReadVTK understands: format="ascii", format="binary" (inline base64) and
// format="appended" (raw or base64 <AppendedData>), each plain or deflated (compressor="vtkZLibDataCompressor"),
// header_type UInt32 / UInt64, little endian.
namespace {

struct VtuArray {
    std::string name, type, format;
    size_t begin = 0, end = 0;   // character range of the values
    long long offset = -1;       // format="appended"
};

struct VtuFile {
    const std::string* s = nullptr;
    bool h64 = false, zlib = false;
    size_t app_begin = std::string::npos;   // first byte after the '_' of <AppendedData>
    bool app_base64 = false;
};

int b64_val(unsigned char c)
{
    if (c >= 'A' && c <= 'Z') return c - 'A';
    if (c >= 'a' && c <= 'z') return c - 'a' + 26;
    if (c >= '0' && c <= '9') return c - '0' + 52;
    if (c == '+') return 62;
    if (c == '/') return 63;
    return -1;
}

// decodes base64 characters from s[pos...) until `want` bytes are out (or the text ends at `end`); whitespace is
// skipped; returns the position after the last consumed 4-character group
size_t b64_decode(const std::string& s, size_t pos, size_t end, size_t want, std::vector<unsigned char>& out)
{
    int q[4], nq = 0;
    while (pos < end && out.size() < want) {
        const unsigned char c = (unsigned char)s[pos++];
        if (isspace(c)) continue;
        if (c == '=') { q[nq++] = 0; }
        else {
            const int v = b64_val(c);
            if (v < 0) break;
            q[nq++] = v;
        }
        if (nq == 4) {
            out.push_back((unsigned char)((q[0] << 2) | (q[1] >> 4)));
            out.push_back((unsigned char)(((q[1] & 15) << 4) | (q[2] >> 2)));
            out.push_back((unsigned char)(((q[2] & 3) << 6) | q[3]));
            nq = 0;
        }
    }
    return pos;
}

// raw bytes of one binary / appended DataArray (header stripped, blocks inflated)
bool vtu_bytes(const VtuFile& F, const VtuArray& A, std::vector<unsigned char>& out)
{
    const std::string& s = *F.s;
    const size_t hs = F.h64 ? 8 : 4;
    const bool appended = strcasecmp(A.format.c_str(), "appended") == 0;
    bool b64 = true;
    size_t pos = A.begin, end = A.end;
    if (appended) {
        if (F.app_begin == std::string::npos || A.offset < 0) return false;
        pos = F.app_begin + (size_t)A.offset;
        end = s.size();
        b64 = F.app_base64;
    }
    auto rd = [&](const unsigned char* p) -> uint64_t {
        uint64_t v = 0;
        memcpy(&v, p, hs);
        return v;
    };
    auto take = [&](size_t nbytes, std::vector<unsigned char>& dst) -> bool {   // next nbytes of the stream
        dst.clear();
        if (b64) {
            // an encoded unit starts on a 4-character boundary: decode whole groups, keep what was asked for
            pos = b64_decode(s, pos, end, (nbytes + 2) / 3 * 3, dst);
            if (dst.size() < nbytes) return false;
            dst.resize(nbytes);
            return true;
        }
        if (pos + nbytes > end) return false;
        dst.assign((const unsigned char*)s.data() + pos, (const unsigned char*)s.data() + pos + nbytes);
        pos += nbytes;
        return true;
    };
    std::vector<unsigned char> h;
    if (!F.zlib) {
        if (!take(hs, h)) return false;
        const uint64_t n = rd(h.data());
        if (b64) {
            // header and data are ONE encoded unit when uncompressed: re-decode from the start
            pos = appended ? F.app_begin + (size_t)A.offset : A.begin;
            std::vector<unsigned char> all;
            pos = b64_decode(s, pos, end, (hs + n + 2) / 3 * 3, all);
            if (all.size() < hs + n) return false;
            out.assign(all.begin() + hs, all.begin() + hs + n);
            return true;
        }
        return take((size_t)n, out);
    }
    if (!take(3 * hs, h)) {
        return false;
    }
    const uint64_t nb = rd(h.data()), bs = rd(h.data() + hs), last = rd(h.data() + 2 * hs);
    if (nb == 0) { out.clear(); return true; }
    if (nb > (1u << 26)) return false;
    // the header (3 + nb words) is one encoded unit
    pos = appended ? F.app_begin + (size_t)A.offset : A.begin;
    if (!take((size_t)(3 + nb) * hs, h)) return false;
    uint64_t csum = 0;
    std::vector<uint64_t> cs(nb);
    for (uint64_t b = 0; b < nb; ++b) { cs[b] = rd(h.data() + (3 + b) * hs); csum += cs[b]; }
    std::vector<unsigned char> comp;
    if (!take((size_t)csum, comp)) return false;
    out.clear();
    size_t cp = 0;
    for (uint64_t b = 0; b < nb; ++b) {
        const uint64_t len = (b + 1 < nb || last == 0) ? bs : last;
        const size_t o = out.size();
        out.resize(o + len);
        uLongf dl = (uLongf)len;
        if (uncompress(out.data() + o, &dl, comp.data() + cp, (uLong)cs[b]) != Z_OK || dl != len) return false;
        cp += cs[b];
    }
    return true;
}

template <class T, class S>
void convert_all(const std::vector<unsigned char>& raw, std::vector<T>& out)
{
    const size_t n = raw.size() / sizeof(S);
    out.resize(n);
    for (size_t i = 0; i < n; ++i) {
        S v;
        memcpy(&v, raw.data() + i * sizeof(S), sizeof(S));
        out[i] = (T)v;
    }
}

template <class T>
bool typed_convert(const std::string& type, const std::vector<unsigned char>& raw, std::vector<T>& out)
{
    if (type == "Float64") convert_all<T, double>(raw, out);
    else if (type == "Float32") convert_all<T, float>(raw, out);
    else if (type == "Int64") convert_all<T, int64_t>(raw, out);
    else if (type == "UInt64") convert_all<T, uint64_t>(raw, out);
    else if (type == "Int32") convert_all<T, int32_t>(raw, out);
    else if (type == "UInt32") convert_all<T, uint32_t>(raw, out);
    else if (type == "Int16") convert_all<T, int16_t>(raw, out);
    else if (type == "UInt16") convert_all<T, uint16_t>(raw, out);
    else if (type == "Int8") convert_all<T, int8_t>(raw, out);
    else if (type == "UInt8") convert_all<T, uint8_t>(raw, out);
    else return false;
    return true;
}

std::string attr_of(const std::string& tag, const char* key)
{
    // case-insensitive attribute name (ExportToVTU.jl writes `Format="ascii"` on its point data)
    const size_t kl = strlen(key);
    for (size_t i = 0; i + kl + 1 < tag.size(); ++i) {
        if ((i == 0 || isspace((unsigned char)tag[i - 1])) && strncasecmp(tag.c_str() + i, key, kl) == 0) {
            size_t j = i + kl;
            while (j < tag.size() && isspace((unsigned char)tag[j])) ++j;
            if (j >= tag.size() || tag[j] != '=') continue;
            ++j;
            while (j < tag.size() && isspace((unsigned char)tag[j])) ++j;
            if (j >= tag.size() || (tag[j] != '"' && tag[j] != '\'')) continue;
            const char q = tag[j];
            const size_t e = tag.find(q, j + 1);
            if (e == std::string::npos) return "";
            return tag.substr(j + 1, e - j - 1);
        }
    }
    return "";
}

// DataArray elements between `from` and `to`
std::vector<VtuArray> data_arrays(const std::string& s, size_t from, size_t to)
{
    std::vector<VtuArray> out;
    size_t p = from;
    while (true) {
        const size_t a = s.find("<DataArray", p);
        if (a == std::string::npos || a >= to) break;
        const size_t b = s.find('>', a);
        if (b == std::string::npos || b >= to) break;
        const std::string tag = s.substr(a, b - a + 1);
        VtuArray A;
        A.name = attr_of(tag, "Name");
        A.type = attr_of(tag, "type");
        A.format = attr_of(tag, "format");
        {
            const std::string off = attr_of(tag, "offset");
            if (!off.empty()) A.offset = atoll(off.c_str());
        }
        if (tag.size() >= 2 && tag[tag.size() - 2] == '/') {   // empty element
            A.begin = A.end = b + 1;
            p = b + 1;
        } else {
            const size_t c = s.find("</DataArray>", b);
            if (c == std::string::npos || c > to) break;
            A.begin = b + 1;
            A.end = c;
            p = c + 12;
        }
        out.push_back(A);
    }
    return out;
}

bool section(const std::string& s, const char* name, size_t& from, size_t& to)
{
    const std::string open = std::string("<") + name, close = std::string("</") + name + ">";
    size_t a = s.find(open);
    while (a != std::string::npos && a + open.size() < s.size() && !(isspace((unsigned char)s[a + open.size()]) || s[a + open.size()] == '>'))
        a = s.find(open, a + 1);   // "<Cells" must not match "<CellData"
    if (a == std::string::npos) return false;
    const size_t b = s.find(close, a);
    if (b == std::string::npos) return false;
    from = a;
    to = b;
    return true;
}

template <class T>
bool parse_numbers(const std::string& s, const VtuArray& A, std::vector<T>& out)
{
    const char* p = s.c_str() + A.begin;
    const char* end = s.c_str() + A.end;
    while (p < end) {
        while (p < end && isspace((unsigned char)*p)) ++p;
        if (p >= end) break;
        char* q = nullptr;
        const double v = strtod(p, &q);
        if (q == p || q > end) return false;
        out.push_back((T)v);
        p = q;
    }
    return true;
}

// values of a DataArray in any of the supported encodings
template <class T>
bool load_array(const VtuFile& F, const VtuArray& A, std::vector<T>& out)
{
    if (A.format.empty() || strcasecmp(A.format.c_str(), "ascii") == 0) return parse_numbers(*F.s, A, out);
    std::vector<unsigned char> raw;
    if (!vtu_bytes(F, A, raw)) return false;
    return typed_convert(A.type, raw, out);
}

}  // namespace

extern "C" {

void r2s_free_vtu_mesh(r2s_vtu_mesh* m)
{
    if (!m) return;
    free(m->X); free(m->IEN); free(m->rho);
    m->X = nullptr; m->IEN = nullptr; m->rho = nullptr;
}

int r2s_import_vtu(const char* filename, r2s_vtu_mesh* out)
{
    if (!filename || !out) return fail(R2S_ERR_ARG, "bad argument");
    memset(out, 0, sizeof *out);
    FILE* f = fopen(filename, "rb");
    if (!f) return fail(R2S_ERR_ARG, "VTU file not found: %s", filename);   // VTUImport.jl:23-25
    std::string s;
    {
        char buf[1 << 16];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    }
    fclose(f);
    if (s.find("UnstructuredGrid") == std::string::npos) return fail(R2S_ERR_ARG, "%s is not a VTK UnstructuredGrid file", filename);
    VtuFile F;
    F.s = &s;
    {
        const size_t v0 = s.find("<VTKFile");
        const size_t v1 = v0 == std::string::npos ? v0 : s.find('>', v0);
        if (v1 != std::string::npos) {
            const std::string tag = s.substr(v0, v1 - v0 + 1);
            F.h64 = attr_of(tag, "header_type") == "UInt64";
            F.zlib = !attr_of(tag, "compressor").empty();
            if (F.zlib && attr_of(tag, "compressor") != "vtkZLibDataCompressor")
                return fail(R2S_ERR_ARG, "%s: compressor %s is not supported (zlib only)", filename, attr_of(tag, "compressor").c_str());
            const std::string bo = attr_of(tag, "byte_order");
            if (!bo.empty() && bo != "LittleEndian") return fail(R2S_ERR_ARG, "%s: byte order %s is not supported", filename, bo.c_str());
        }
        const size_t ap = s.find("<AppendedData");
        if (ap != std::string::npos) {
            const size_t ae = s.find('>', ap);
            if (ae != std::string::npos) {
                F.app_base64 = attr_of(s.substr(ap, ae - ap + 1), "encoding") == "base64";
                const size_t us = s.find('_', ae);
                if (us != std::string::npos) F.app_begin = us + 1;
            }
        }
    }
    size_t a, b;
    if (!section(s, "Points", a, b)) return fail(R2S_ERR_ARG, "%s: no <Points>", filename);
    std::vector<VtuArray> pa = data_arrays(s, a, b);
    if (pa.empty()) return fail(R2S_ERR_ARG, "%s: <Points> holds no DataArray", filename);
    std::vector<double> pts;
    if (!load_array(F, pa[0], pts) || pts.size() % 3 != 0 || pts.empty()) return fail(R2S_ERR_ARG, "%s: malformed point coordinates", filename);
    if (!section(s, "Cells", a, b)) return fail(R2S_ERR_ARG, "%s: no <Cells>", filename);
    std::vector<int64_t> conn, offs, types;
    bool have[3] = {false, false, false};
    for (const VtuArray& A : data_arrays(s, a, b)) {
        std::vector<int64_t>* dst = A.name == "connectivity" ? &conn : (A.name == "offsets" ? &offs : (A.name == "types" ? &types : nullptr));
        if (!dst) continue;
        if (!load_array(F, A, *dst)) return fail(R2S_ERR_ARG, "%s: malformed cell array \"%s\"", filename, A.name.c_str());
        have[A.name == "connectivity" ? 0 : (A.name == "offsets" ? 1 : 2)] = true;
    }
    if (!have[0] || !have[1] || !have[2] || offs.size() != types.size()) return fail(R2S_ERR_ARG, "%s: connectivity / offsets / types missing or inconsistent", filename);
    const int64_t nnp = (int64_t)(pts.size() / 3), ncell = (int64_t)types.size();
    // supported cells: hexahedron (12, 8 nodes) and tetrahedron (10, 4 nodes); the others are skipped (:57-94)
    int64_t nhex = 0, ntet = 0, skipped = 0;
    for (int64_t i = 0; i < ncell; ++i) {
        const int64_t n = offs[i] - (i ? offs[i - 1] : 0);
        if (offs[i] > (int64_t)conn.size() || n < 0) return fail(R2S_ERR_ARG, "%s: offsets run past the connectivity", filename);
        if (types[i] == 12 && n == 8) ++nhex;
        else if (types[i] == 10 && n == 4) ++ntet;
        else ++skipped;
    }
    if (nhex + ntet == 0)
        return fail(R2S_ERR_ARG, "No supported elements found in VTU file. Supported types: Hexahedron (12), Tetrahedron (10)");   // :96-98
    if (nhex && ntet) return fail(R2S_ERR_ARG, "%s mixes hexahedra and tetrahedra: one element type per mesh", filename);
    const int nen = nhex ? 8 : 4;
    const int64_t nel = nhex + ntet;
    out->X = (double*)malloc(sizeof(double) * 3 * (size_t)nnp);
    out->IEN = (int64_t*)malloc(sizeof(int64_t) * (size_t)nen * (size_t)nel);
    out->rho = (double*)malloc(sizeof(double) * (size_t)nel);
    if (!out->X || !out->IEN || !out->rho) { r2s_free_vtu_mesh(out); return fail(R2S_ERR_ARG, "out of memory"); }
    memcpy(out->X, pts.data(), sizeof(double) * pts.size());
    int64_t e = 0;
    for (int64_t i = 0; i < ncell; ++i) {
        const int64_t o0 = i ? offs[i - 1] : 0, n = offs[i] - o0;
        if (!((types[i] == 12 && n == 8) || (types[i] == 10 && n == 4))) continue;
        for (int q = 0; q < nen; ++q) {
            const int64_t node = conn[o0 + q];   // 0-based in the file
            if (node < 0 || node >= nnp) { r2s_free_vtu_mesh(out); return fail(R2S_ERR_ARG, "%s: cell %lld refers to point %lld of %lld", filename, (long long)i, (long long)node, (long long)nnp); }
            out->IEN[e * nen + q] = node + 1;
        }
        ++e;
    }
    // element densities: first cell-data field with one of the usual names, else the first field, else 1.0 (:117-226)
    for (int64_t i = 0; i < nel; ++i) out->rho[i] = 1.0;
    if (section(s, "CellData", a, b)) {
        static const char* names[] = {"density", "Density", "DENSITY", "rho", "Rho", "RHO", "volfrac", "VolFrac", "vol_frac",
                                      "VOLFRAC", "material_density", "element_density", "topology", "design_variable"};
        std::vector<VtuArray> ca = data_arrays(s, a, b);
        const VtuArray* pick = nullptr;
        for (const char* nm : names) {
            for (const VtuArray& A : ca)
                if (A.name == nm) { pick = &A; break; }
            if (pick) break;
        }
        if (!pick && !ca.empty()) pick = &ca[0];
        if (pick) {
            std::vector<double> d;
            if (load_array(F, *pick, d)) {
                // by position over the supported elements; too long: truncated, too short: padded with 1.0 (:183-196)
                for (int64_t i = 0; i < nel && i < (int64_t)d.size(); ++i) out->rho[i] = d[i];
                snprintf(out->density_field, sizeof out->density_field, "%s", pick->name.c_str());
            }
        }
    }
    out->nnp = nnp; out->nel = nel; out->nen = nen;
    out->elem_type = nhex ? R2S_HEX8 : R2S_TET4;
    out->n_skipped = skipped;
    return 0;
}

}  // extern "C"

// ---- MeshInformations(data) for MATLAB level-5 .mat files (src/MeshGrid/MeshInformations.jl:3-12) --------------
// The reference loads the file with MAT.jl and takes `rho`, `msh.X` (3 x nnp) and `msh.IEN` (nen x nel, 0-based in
// the reference's data: IEN .+ 1, :8).  Reader for the level-5 container: 128-byte header, data elements (tag +
// payload, small-element form included), miCOMPRESSED elements (zlib), miMATRIX with numeric and struct classes,
// numeric payloads of any stored integer / floating type.  v7.3 files are HDF5 and are refused with a message.
namespace {

struct MatVar {
    std::string name;
    std::vector<int64_t> dims;
    std::vector<double> val;                // numeric arrays, column-major
    std::vector<std::string> field_names;   // structs (1 x 1)
    std::vector<MatVar> fields;
    bool is_struct = false;
};

struct MatCursor {
    const unsigned char* p;
    size_t n, o = 0;
    bool tag(uint32_t& type, uint32_t& bytes, const unsigned char*& data)
    {
        if (o + 8 > n) return false;
        uint32_t w0, w1;
        memcpy(&w0, p + o, 4);
        memcpy(&w1, p + o + 4, 4);
        if (w0 >> 16) {   // small data element: type in the low half, byte count in the high half, data in the tag
            type = w0 & 0xFFFF;
            bytes = w0 >> 16;
            if (bytes > 4) return false;
            data = p + o + 4;
            o += 8;
            return true;
        }
        type = w0;
        bytes = w1;
        if (o + 8 + (size_t)bytes > n) return false;
        data = p + o + 8;
        o += 8 + (((size_t)bytes + 7) & ~(size_t)7);
        if (type == 15) o = (size_t)(data - p) + bytes;   // compressed elements are not padded
        return true;
    }
};

bool mat_numeric(uint32_t type, const unsigned char* d, uint32_t bytes, std::vector<double>& out)
{
    std::vector<unsigned char> raw(d, d + bytes);
    switch (type) {
    case 1: convert_all<double, int8_t>(raw, out); return true;
    case 2: convert_all<double, uint8_t>(raw, out); return true;
    case 3: convert_all<double, int16_t>(raw, out); return true;
    case 4: convert_all<double, uint16_t>(raw, out); return true;
    case 5: convert_all<double, int32_t>(raw, out); return true;
    case 6: convert_all<double, uint32_t>(raw, out); return true;
    case 7: convert_all<double, float>(raw, out); return true;
    case 9: convert_all<double, double>(raw, out); return true;
    case 12: convert_all<double, int64_t>(raw, out); return true;
    case 13: convert_all<double, uint64_t>(raw, out); return true;
    }
    return false;
}

bool mat_matrix(const unsigned char* d, size_t n, MatVar& v, int depth = 0)
{
    if (depth > 8) return false;
    MatCursor c{d, n};
    uint32_t t, b;
    const unsigned char* q;
    if (!c.tag(t, b, q) || t != 6 || b < 8) return false;   // array flags
    uint32_t flags;
    memcpy(&flags, q, 4);
    const uint32_t cls = flags & 0xFF;
    if (!c.tag(t, b, q) || t != 5) return false;            // dimensions
    v.dims.resize(b / 4);
    for (size_t i = 0; i < v.dims.size(); ++i) { int32_t x; memcpy(&x, q + 4 * i, 4); v.dims[i] = x; }
    if (!c.tag(t, b, q) || t != 1) return false;            // name
    v.name.assign((const char*)q, b);
    if (cls == 2) {                                           // struct
        v.is_struct = true;
        if (!c.tag(t, b, q) || t != 5 || b != 4) return false;
        int32_t flen;
        memcpy(&flen, q, 4);
        if (!c.tag(t, b, q) || t != 1 || flen <= 0) return false;
        const size_t nf = b / (size_t)flen;
        for (size_t i = 0; i < nf; ++i) v.field_names.emplace_back((const char*)q + i * flen, strnlen((const char*)q + i * flen, flen));
        int64_t ne = 1;
        for (int64_t dd : v.dims) ne *= dd;
        if (ne != 1) return false;                            // 1 x 1 structs only
        for (size_t i = 0; i < nf; ++i) {
            if (!c.tag(t, b, q) || t != 14) return false;
            MatVar f;
            if (b && !mat_matrix(q, b, f, depth + 1)) return false;
            f.name = v.field_names[i];
            v.fields.push_back(f);
        }
        return true;
    }
    if (cls >= 6 && cls <= 15) {                              // double, single, (u)int8..64
        if (!c.tag(t, b, q)) return false;
        return mat_numeric(t, q, b, v.val);
    }
    return true;   // other classes (cell, char, sparse ...): kept as an empty variable
}

}  // namespace

extern "C" int r2s_import_mat(const char* filename, r2s_vtu_mesh* out)
{
    if (!filename || !out) return fail(R2S_ERR_ARG, "bad argument");
    memset(out, 0, sizeof *out);
    FILE* f = fopen(filename, "rb");
    if (!f) return fail(R2S_ERR_ARG, "MAT file not found: %s", filename);
    std::vector<unsigned char> s;
    {
        unsigned char buf[1 << 16];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.insert(s.end(), buf, buf + n);
    }
    fclose(f);
    if (s.size() < 128) return fail(R2S_ERR_ARG, "%s is not a MAT file", filename);
    if (memcmp(s.data(), "MATLAB 7.3", 10) == 0) return fail(R2S_ERR_UNSUPPORTED, "%s is a v7.3 (HDF5) MAT file: only level-5 files are read", filename);
    if (memcmp(s.data(), "MATLAB 5.0", 10) != 0) return fail(R2S_ERR_ARG, "%s is not a level-5 MAT file", filename);
    if (!(s[126] == 'I' && s[127] == 'M')) return fail(R2S_ERR_UNSUPPORTED, "%s: big-endian MAT files are not supported", filename);
    std::vector<MatVar> vars;
    MatCursor c{s.data(), s.size()};
    c.o = 128;
    while (c.o + 8 <= s.size()) {
        uint32_t t, b;
        const unsigned char* q;
        if (!c.tag(t, b, q)) return fail(R2S_ERR_ARG, "%s: truncated data element", filename);
        std::vector<unsigned char> inflated;
        if (t == 15) {
            z_stream z;
            memset(&z, 0, sizeof z);
            if (inflateInit(&z) != Z_OK) return fail(R2S_ERR_ARG, "zlib initialisation failed");
            z.next_in = const_cast<Bytef*>(q);
            z.avail_in = b;
            int rc;
            do {
                const size_t o = inflated.size();
                inflated.resize(o + (1 << 20));
                z.next_out = inflated.data() + o;
                z.avail_out = 1 << 20;
                rc = inflate(&z, Z_NO_FLUSH);
                inflated.resize(o + ((1 << 20) - z.avail_out));
            } while (rc == Z_OK);
            inflateEnd(&z);
            if (rc != Z_STREAM_END) return fail(R2S_ERR_ARG, "%s: corrupt compressed element", filename);
            MatCursor ci{inflated.data(), inflated.size()};
            if (!ci.tag(t, b, q)) return fail(R2S_ERR_ARG, "%s: corrupt compressed element", filename);
        }
        if (t != 14) continue;
        MatVar v;
        if (!mat_matrix(q, b, v)) return fail(R2S_ERR_ARG, "%s: unsupported or malformed variable", filename);
        vars.push_back(v);
    }
    const MatVar *rho = nullptr, *msh = nullptr, *X = nullptr, *IEN = nullptr;
    for (const MatVar& v : vars) {
        if (v.name == "rho") rho = &v;
        if (v.name == "msh" && v.is_struct) msh = &v;
    }
    if (!rho) return fail(R2S_ERR_ARG, "%s: variable \"rho\" not found (MeshInformations.jl:4)", filename);
    if (!msh) return fail(R2S_ERR_ARG, "%s: struct \"msh\" not found (MeshInformations.jl:5)", filename);
    for (const MatVar& v : msh->fields) {
        if (v.name == "X") X = &v;
        if (v.name == "IEN") IEN = &v;
    }
    if (!X || !IEN || X->dims.size() != 2 || IEN->dims.size() != 2) return fail(R2S_ERR_ARG, "%s: msh.X / msh.IEN missing", filename);
    if (X->dims[0] != 3) return fail(R2S_ERR_ARG, "%s: msh.X is %lld x %lld, expected 3 x nnp", filename, (long long)X->dims[0], (long long)X->dims[1]);
    const int64_t nnp = X->dims[1], nen = IEN->dims[0], nel = IEN->dims[1];
    if (nen != 8 && nen != 4) return fail(R2S_ERR_ARG, "%s: msh.IEN has %lld rows (8 = HEX8 or 4 = TET4 expected)", filename, (long long)nen);
    if ((int64_t)X->val.size() != 3 * nnp || (int64_t)IEN->val.size() != nen * nel || (int64_t)rho->val.size() != nel)
        return fail(R2S_ERR_ARG, "%s: rho (%zu) / msh.X / msh.IEN (%lld elements) sizes do not fit", filename, rho->val.size(), (long long)nel);
    out->X = (double*)malloc(sizeof(double) * 3 * (size_t)nnp);
    out->IEN = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nen * nel));
    out->rho = (double*)malloc(sizeof(double) * (size_t)nel);
    if (!out->X || !out->IEN || !out->rho) { r2s_free_vtu_mesh(out); return fail(R2S_ERR_ARG, "out of memory"); }
    memcpy(out->X, X->val.data(), sizeof(double) * 3 * (size_t)nnp);            // column-major 3 x nnp = [nnp][3]
    for (int64_t i = 0; i < nen * nel; ++i) out->IEN[i] = (int64_t)IEN->val[i] + 1;   // IEN .+ 1 (MeshInformations.jl:8)
    memcpy(out->rho, rho->val.data(), sizeof(double) * (size_t)nel);
    out->nnp = nnp; out->nel = nel; out->nen = (int32_t)nen;
    out->elem_type = nen == 8 ? R2S_HEX8 : R2S_TET4;
    snprintf(out->density_field, sizeof out->density_field, "rho");
    return 0;
}
