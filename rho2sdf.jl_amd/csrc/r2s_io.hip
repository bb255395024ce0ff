// On-disk output of the hot path's result: VTK ImageData (.vti), the format rho2sdf() hands to ParaView
// (reference: src/DataExport/ExportToVTI.jl:22-67, WriteVTK's vtk_grid(filename, x, y, z) for three ranges).
// Host code only (no device work); part of the library so that non-Julia hosts get the same artefact.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

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
