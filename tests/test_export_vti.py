"""exportSdfToVTI (reference src/DataExport/ExportToVTI.jl:22-67): the file is parsed back here and must
carry the reference's metadata (dimensions N*smooth+1, Origin = AABB_min, Spacing = cell/smooth, point array
name) and the exact values.  Host-only code path: runs without a GPU."""
import re

import numpy as np
import pytest

from conftest import load_fixture


def _read_vti(path):
    raw = open(path, "rb").read()
    head, _, tail = raw.partition(b'<AppendedData encoding="raw">\n_')
    h = head.decode()
    ext = [int(v) for v in re.search(r'WholeExtent="([^"]+)"', h).group(1).split()]
    origin = [float(v) for v in re.search(r'Origin="([^"]+)"', h).group(1).split()]
    spacing = [float(v) for v in re.search(r'Spacing="([^"]+)"', h).group(1).split()]
    arr = re.search(r'<DataArray type="(\w+)" Name="([^"]+)" format="appended" offset="0"/>', h)
    nbytes = int(np.frombuffer(tail[:8], dtype="<u8")[0])
    dt = {"Float32": "<f4", "Float64": "<f8"}[arr.group(1)]
    data = np.frombuffer(tail[8:8 + nbytes], dtype=dt)
    assert tail[8 + nbytes:].strip().endswith(b"</VTKFile>")
    return ext, origin, spacing, arr.group(2), data


@pytest.mark.parametrize("smooth,dtype", [(None, np.float64), (1, np.float32), (2, np.float32)])
def test_vti_round_trip(pkg, tmp_path, smooth, dtype):
    g = pkg.Grid(np.array([-1.0, 0.5, 2.0]), np.array([3.0, 2.5, 2.75]), 9, 2)
    s = smooth or 1
    dims = tuple(int(n) * s + 1 for n in g.c.N)
    rng = np.random.default_rng(5)
    vals = rng.normal(size=dims[2] * dims[1] * dims[0]).astype(dtype)
    path = pkg.exportSdfToVTI(str(tmp_path / "out"), g, vals.reshape(dims[2], dims[1], dims[0]), "distance", smooth)
    assert path.endswith("out.vti")
    ext, origin, spacing, name, data = _read_vti(path)
    assert ext == [0, dims[0] - 1, 0, dims[1] - 1, 0, dims[2] - 1]
    assert origin == [float(v) for v in g.AABB_min]
    assert spacing == [g.cell_size / s] * 3
    assert name == "distance"
    assert data.dtype.itemsize == np.dtype(dtype).itemsize and np.array_equal(data, vals)


def test_vti_length_check(pkg, tmp_path):
    g = pkg.Grid(np.zeros(3), np.ones(3), 4, 1)
    with pytest.raises(pkg._lib.R2SError, match="doesn't match grid dimensions"):    # ExportToVTI.jl:47-49
        pkg.exportSdfToVTI(str(tmp_path / "bad"), g, np.zeros(7), "distance")


def test_result_file_name(pkg, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    g = pkg.Grid(np.zeros(3), np.ones(3), 4, 1)
    dims = tuple(int(n) * 2 + 1 for n in g.c.N)
    p = pkg.export_sdf_results(np.zeros(dims[::-1], dtype=np.float32), g, "beam", 2, True, pkg._lib.HEX8)
    assert p == f"beam_HEX8_B-{round(g.cell_size, 4)}_smooth-2_Interpolation.vti"     # RhoToSDF.jl:268


def test_vtu_round_trip(pkg, tmp_path):
    """exportToVTU (reference src/DataExport/ExportToVTU.jl:2-99): points, 0-based connectivity, offsets, cell types
    and the nodal density array read back from the ASCII file"""
    from conftest import load_fixture
    X, IEN, rho = load_fixture("sphere")
    rn = np.linspace(0.0, 1.0, len(X))
    X = X.copy()
    X[3, 1] = 1e-25                                   # |x| < 1e-20 is written as 0 (ExportToVTU.jl:39-41)
    path = pkg.exportToVTU(str(tmp_path / "mesh.vtu"), X, IEN, 12, rn)
    txt = open(path).read()
    assert f'NumberOfPoints="{len(X)}" NumberOfCells="{len(IEN)}"' in txt
    blocks = re.findall(r"<DataArray[^>]*>\n(.*?)</DataArray>", txt, flags=re.S)
    pts = np.array(blocks[0].split(), dtype=np.float64).reshape(-1, 3)
    conn = np.array(blocks[1].split(), dtype=np.int64).reshape(-1, 8)
    offs = np.array(blocks[2].split(), dtype=np.int64)
    types = np.array(blocks[3].split(), dtype=np.int64)
    dens = np.array(blocks[4].split(), dtype=np.float64)
    Xz = X.copy()
    Xz[np.abs(Xz) < 1e-20] = 0.0
    assert np.array_equal(pts, Xz) and np.array_equal(conn, IEN - 1)
    assert np.array_equal(offs, 8 * np.arange(1, len(IEN) + 1)) and np.all(types == 12)
    assert np.array_equal(dens, rn) and 'Name="density"' in txt


def _vtu(points, conn, offsets, types, cell_data="", fmt="ascii"):
    pts = "\n".join(" ".join(repr(float(v)) for v in p) for p in points)
    return f"""<?xml version="1.0"?>
<VTKFile type="UnstructuredGrid" version="0.1" byte_order="LittleEndian">
  <UnstructuredGrid>
    <Piece NumberOfPoints="{len(points)}" NumberOfCells="{len(types)}">
      <Points>
        <DataArray type="Float64" NumberOfComponents="3" format="{fmt}">
{pts}
        </DataArray>
      </Points>
      <Cells>
        <DataArray type="Int64" Name="connectivity" format="ascii">{" ".join(map(str, conn))}</DataArray>
        <DataArray type="Int64" Name="offsets" format="ascii">{" ".join(map(str, offsets))}</DataArray>
        <DataArray type="UInt8" Name="types" format="ascii">{" ".join(map(str, types))}</DataArray>
      </Cells>
      {cell_data}
    </Piece>
  </UnstructuredGrid>
</VTKFile>
"""


def test_vtu_import_round_trip(pkg, tmp_path):
    """import_vtu_mesh (VTUImport.jl:22-112) reads back what exportToVTU wrote: same X, 1-based IEN; the file
    holds no cell data, so the element densities default to 1.0 (:127-136)"""
    X, IEN, rho = load_fixture("sphere")
    path = pkg.exportToVTU(str(tmp_path / "sphere.vtu"), X, IEN, 12, np.linspace(0, 1, len(X)))
    info = {}
    X2, IEN2, rho2 = pkg.import_vtu_mesh(path, info)
    assert np.array_equal(X2, X) and np.array_equal(IEN2, IEN) and IEN2.dtype == np.int64
    assert np.array_equal(rho2, np.ones(len(IEN))) and info == {"element_type": 0, "n_skipped": 0, "density_field": ""}


def test_vtu_import_cells_and_density_fields(pkg, tmp_path):
    pts = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1)]
    # two tetrahedra, one triangle (type 5: skipped, VTUImport.jl:89-92); densities by position (:183-196)
    conn, offs, types = [0, 1, 2, 3, 1, 2, 3, 4, 0, 1, 2], [4, 8, 11], [10, 10, 5]
    cd = ('<CellData><DataArray type="Float64" Name="stress" format="ascii">9 9 9</DataArray>'
          '<DataArray type="Float64" Name="volfrac" format="ascii">0.25 0.75 0.5</DataArray></CellData>')
    p = tmp_path / "tets.vtu"
    p.write_text(_vtu(pts, conn, offs, types, cd))
    info = {}
    X, IEN, rho = pkg.import_vtu_mesh(str(p), info)
    assert X.shape == (5, 3) and np.array_equal(IEN, [[1, 2, 3, 4], [2, 3, 4, 5]])
    assert np.array_equal(rho, [0.25, 0.75]) and info == {"element_type": 1, "n_skipped": 1, "density_field": "volfrac"}
    # no usual name: the first field is used; too short: padded with 1.0
    p.write_text(_vtu(pts, conn, offs, types, '<CellData><DataArray type="Float32" Name="stress" format="ascii">0.5</DataArray></CellData>'))
    X, IEN, rho = pkg.import_vtu_mesh(str(p), info)
    assert np.array_equal(rho, [0.5, 1.0]) and info["density_field"] == "stress"


def test_vtu_import_errors(pkg, tmp_path):
    with pytest.raises(pkg._lib.R2SError, match="VTU file not found"):             # VTUImport.jl:23-25
        pkg.import_vtu_mesh(str(tmp_path / "missing.vtu"))
    pts = [(0, 0, 0), (1, 0, 0), (0, 1, 0)]
    p = tmp_path / "tri.vtu"
    p.write_text(_vtu(pts, [0, 1, 2], [3], [5]))
    with pytest.raises(pkg._lib.R2SError, match="No supported elements found"):   # :96-98
        pkg.import_vtu_mesh(str(p))
    p.write_text(_vtu(pts + [(0, 0, 1)], [0, 1, 2, 7], [4], [10]))
    with pytest.raises(pkg._lib.R2SError, match="refers to point"):
        pkg.import_vtu_mesh(str(p))
    p.write_text(_vtu(pts + [(0, 0, 1)], [0, 1, 2, 3], [4], [10], fmt="binary"))
    with pytest.raises(pkg._lib.R2SError, match="ASCII .vtu only"):
        pkg.import_vtu_mesh(str(p))
