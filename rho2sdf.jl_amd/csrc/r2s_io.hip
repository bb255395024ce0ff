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
    if (!filename || !grid || !values || !value_label) return fail(R2S_ERR_ARG, "null argument");
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
                     "<VTKFile type=\"ImageData\" version=\"1.0\" byte_order=\"LittleEndian\" header_type=\"UInt64\">\n"
                     "  <ImageData WholeExtent=\"0 %lld 0 %lld 0 %lld\" Origin=\"%.17g %.17g %.17g\" Spacing=\"%.17g %.17g %.17g\">\n"
                     "    <Piece Extent=\"0 %lld 0 %lld 0 %lld\">\n"
                     "      <PointData>\n"
                     "        <DataArray type=\"%s\" Name=\"%s\" format=\"appended\" offset=\"0\"/>\n"
                     "      </PointData>\n"
                     "    </Piece>\n"
                     "  </ImageData>\n"
                     "  <AppendedData encoding=\"raw\">\n_",
                     (long long)dims[0] - 1, (long long)dims[1] - 1, (long long)dims[2] - 1, grid->aabb_min[0],
                     grid->aabb_min[1], grid->aabb_min[2], spacing, spacing, spacing, (long long)dims[0] - 1,
                     (long long)dims[1] - 1, (long long)dims[2] - 1, is_float32 ? "Float32" : "Float64", value_label) > 0;
    ok = ok && fwrite(&nbytes, sizeof nbytes, 1, f) == 1;
    ok = ok && fwrite(values, 1, (size_t)nbytes, f) == (size_t)nbytes;
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
// ASCII UnstructuredGrid only (what exportToVTU and most FE exporters write); the reference reads through
// ReadVTK, which also understands binary / appended data - those are refused with a message.
namespace {

struct VtuArray {
    std::string name, type, format;
    size_t begin = 0, end = 0;   // character range of the values
};

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
    if (s.find("<AppendedData") != std::string::npos) return fail(R2S_ERR_ARG, "%s: appended data is not supported (ASCII .vtu only)", filename);
    size_t a, b;
    if (!section(s, "Points", a, b)) return fail(R2S_ERR_ARG, "%s: no <Points>", filename);
    std::vector<VtuArray> pa = data_arrays(s, a, b);
    if (pa.empty()) return fail(R2S_ERR_ARG, "%s: <Points> holds no DataArray", filename);
    if (!pa[0].format.empty() && strcasecmp(pa[0].format.c_str(), "ascii") != 0)
        return fail(R2S_ERR_ARG, "%s: DataArray format \"%s\" is not supported (ASCII .vtu only)", filename, pa[0].format.c_str());
    std::vector<double> pts;
    if (!parse_numbers(s, pa[0], pts) || pts.size() % 3 != 0 || pts.empty()) return fail(R2S_ERR_ARG, "%s: malformed point coordinates", filename);
    if (!section(s, "Cells", a, b)) return fail(R2S_ERR_ARG, "%s: no <Cells>", filename);
    std::vector<int64_t> conn, offs, types;
    bool have[3] = {false, false, false};
    for (const VtuArray& A : data_arrays(s, a, b)) {
        if (!A.format.empty() && strcasecmp(A.format.c_str(), "ascii") != 0)
            return fail(R2S_ERR_ARG, "%s: DataArray format \"%s\" is not supported (ASCII .vtu only)", filename, A.format.c_str());
        std::vector<int64_t>* dst = A.name == "connectivity" ? &conn : (A.name == "offsets" ? &offs : (A.name == "types" ? &types : nullptr));
        if (!dst) continue;
        if (!parse_numbers(s, A, *dst)) return fail(R2S_ERR_ARG, "%s: malformed cell array \"%s\"", filename, A.name.c_str());
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
                if (A.name == nm && (A.format.empty() || strcasecmp(A.format.c_str(), "ascii") == 0)) { pick = &A; break; }
            if (pick) break;
        }
        if (!pick)
            for (const VtuArray& A : ca)
                if (A.format.empty() || strcasecmp(A.format.c_str(), "ascii") == 0) { pick = &A; break; }
        if (pick) {
            std::vector<double> d;
            if (parse_numbers(s, *pick, d)) {
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
