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
    p.write_text(_vtu(pts + [(0, 0, 1)], [0, 1, 2, 3], [4], [10], fmt="binary"))     # "binary" arrays that hold text
    with pytest.raises(pkg._lib.R2SError, match="malformed"):
        pkg.import_vtu_mesh(str(p))


# ---- binary / appended / compressed .vtu (what ReadVTK reads, VTUImport.jl:33) and zlib .vti, .mat import --------
def _vtk_block(a, h64, compressed, encoding):
    """one DataArray payload as VTK writes it: header (+ deflated blocks), raw bytes or base64 text"""
    import base64
    import zlib
    raw = np.ascontiguousarray(a).tobytes()
    ht = "<u8" if h64 else "<u4"
    if not compressed:
        blob = np.array([len(raw)], dtype=ht).tobytes() + raw
        return blob if encoding == "raw" else base64.b64encode(blob)
    bs = 64                                       # tiny blocks: several of them and a partial last one
    blocks = [raw[i:i + bs] for i in range(0, len(raw), bs)] or [b""]
    comp = [zlib.compress(b) for b in blocks]
    last = len(blocks[-1])
    hdr = np.array([len(blocks), bs, 0 if last == bs else last] + [len(c) for c in comp], dtype=ht).tobytes()
    if encoding == "raw":
        return hdr + b"".join(comp)
    return base64.b64encode(hdr) + base64.b64encode(b"".join(comp))


def _binary_vtu(X, IEN0, rho, mode, h64, compressed):
    nel, nen = IEN0.shape
    arrays = [("Points", "", "Float64", X.astype("<f8"), ' NumberOfComponents="3"'),
              ("Cells", "connectivity", "Int64", IEN0.astype("<i8").ravel(), ""),
              ("Cells", "offsets", "Int32", (nen * np.arange(1, nel + 1)).astype("<i4"), ""),
              ("Cells", "types", "UInt8", np.full(nel, 12 if nen == 8 else 10, dtype="u1"), ""),
              ("CellData", "density", "Float32", rho.astype("<f4"), "")]
    enc = "raw" if mode == "appended-raw" else "base64"
    appended = b""
    tags = {}
    for sec, name, typ, a, extra in arrays:
        blob = _vtk_block(a, h64, compressed, enc)
        if mode == "inline":
            t = f'<DataArray type="{typ}" Name="{name}" format="binary"{extra}>{blob.decode()}</DataArray>'
        else:
            t = f'<DataArray type="{typ}" Name="{name}" format="appended" offset="{len(appended)}"{extra}/>'
            appended += blob
        tags.setdefault(sec, []).append(t)
    attrs = f' header_type="{"UInt64" if h64 else "UInt32"}"' + (' compressor="vtkZLibDataCompressor"' if compressed else "")
    head = (f'<?xml version="1.0"?>\n<VTKFile type="UnstructuredGrid" version="1.0" byte_order="LittleEndian"{attrs}>\n'
            f'<UnstructuredGrid><Piece NumberOfPoints="{len(X)}" NumberOfCells="{nel}">\n'
            f'<Points>{"".join(tags["Points"])}</Points>\n<Cells>{"".join(tags["Cells"])}</Cells>\n'
            f'<CellData>{"".join(tags["CellData"])}</CellData>\n</Piece></UnstructuredGrid>\n').encode()
    if mode != "inline":
        head += f'<AppendedData encoding="{enc}">\n_'.encode() + appended + b"\n</AppendedData>\n"
    return head + b"</VTKFile>\n"


@pytest.mark.parametrize("compressed", [False, True])
@pytest.mark.parametrize("h64", [False, True])
@pytest.mark.parametrize("mode", ["inline", "appended-raw", "appended-base64"])
def test_vtu_import_binary_encodings(pkg, tmp_path, mode, h64, compressed):
    X, IEN, rho = load_fixture("sphere")
    p = tmp_path / "m.vtu"
    p.write_bytes(_binary_vtu(X, IEN - 1, rho, mode, h64, compressed))
    info = {}
    X2, IEN2, rho2 = pkg.import_vtu_mesh(str(p), info)
    assert np.array_equal(X2, X) and np.array_equal(IEN2, IEN)
    assert np.array_equal(rho2, rho.astype(np.float32).astype(np.float64)) and info["density_field"] == "density"


def test_vti_zlib_round_trip(pkg, tmp_path):
    """the compressed appended form WriteVTK writes by default (ExportToVTI.jl:55-64 goes through vtk_grid)"""
    import zlib
    g = pkg.Grid(np.array([-1.0, 0.5, 2.0]), np.array([3.0, 2.5, 2.75]), 130, 2)
    dims = g.dims
    vals = np.random.default_rng(6).normal(size=dims[2] * dims[1] * dims[0])      # > 1 MiB: several blocks
    path = pkg.exportSdfToVTI(str(tmp_path / "z"), g, vals, "distance", None, compress=6)
    raw = open(path, "rb").read()
    head, _, tail = raw.partition(b'<AppendedData encoding="raw">\n_')
    assert b'compressor="vtkZLibDataCompressor"' in head and b'header_type="UInt64"' in head
    nb, bs, last = (int(v) for v in np.frombuffer(tail[:24], dtype="<u8"))
    cs = np.frombuffer(tail[24:24 + 8 * nb], dtype="<u8").astype(int)
    o = 24 + 8 * nb
    out = b""
    for c in cs:
        out += zlib.decompress(tail[o:o + c])
        o += c
    assert nb > 1 and len(out) == (nb - 1) * bs + (last or bs) == vals.nbytes
    assert np.array_equal(np.frombuffer(out, dtype="<f8"), vals)
    assert len(raw) < vals.nbytes and tail[o:].strip().endswith(b"</VTKFile>")


@pytest.mark.parametrize("compress", [True, False])
def test_mat_import(pkg, tmp_path, compress):
    """MeshInformations (MeshInformations.jl:3-12) on MATLAB level-5 files written by scipy.io.savemat in the layout of
    the reference's test data (test/*.mat: `rho` (nel,1) double, struct `msh` with X (3,nnp) double and IEN (8,nel)
    uint16, 0-based) - with and without miCOMPRESSED elements"""
    import scipy.io
    X, IEN, rho = load_fixture("chapadlo")
    p = str(tmp_path / "m.mat")
    scipy.io.savemat(p, {"rho": rho.reshape(-1, 1), "msh": {"X": X.T.copy(), "IEN": (IEN - 1).T.astype(np.uint16)}},
                     do_compression=compress)
    X2, IEN2, rho2 = pkg.MeshInformations(p)
    assert np.array_equal(X2, X) and np.array_equal(IEN2, IEN) and np.array_equal(rho2, rho)
    # TET4 connectivity stored as doubles
    from rho2sdf_jl_amd import synthetic
    Xt, IT, _ = synthetic.tet_mesh(3)
    scipy.io.savemat(p, {"msh": {"IEN": (IT - 1).T.astype(float), "X": Xt.T.copy()}, "rho": np.linspace(0, 1, len(IT))},
                     do_compression=compress)
    X2, IEN2, rho2 = pkg.MeshInformations(p)
    assert np.array_equal(X2, Xt) and np.array_equal(IEN2, IT) and np.array_equal(rho2, np.linspace(0, 1, len(IT)))
    scipy.io.savemat(p, {"rho": rho}, do_compression=compress)
    with pytest.raises(pkg._lib.R2SError, match='struct "msh" not found'):
        pkg.MeshInformations(p)
    with pytest.raises(pkg._lib.R2SError, match="MAT file not found"):
        pkg.MeshInformations(str(tmp_path / "none.mat"))


def _read_vti_stdlib(path):
    """A reader that shares nothing with the library: the XML head through xml.etree, the appended block by the VTK XML
    rules (header_type words; vtkZLibDataCompressor: [blocks, block size, last block size, compressed sizes...] then the
    zlib streams).  What WriteVTK 1.21 writes for `vtk_grid(name, x, y, z)` + `vtk_point_data` (ExportToVTI.jl:55-64):
    ImageData, version 1.0, LittleEndian, UInt64 headers, one appended raw block per array, offset 0 for the first."""
    import xml.etree.ElementTree as ET
    import zlib
    raw = open(path, "rb").read()
    cut = raw.index(b'<AppendedData')
    tag_end = raw.index(b">", cut) + 1
    root = ET.fromstring(raw[:tag_end] + b"</AppendedData></VTKFile>")
    assert raw[tag_end:tag_end + 2] == b"\n_"
    payload = raw[tag_end + 2:]
    assert root.tag == "VTKFile" and root.attrib["type"] == "ImageData" and root.attrib["version"] == "1.0"
    assert root.attrib["byte_order"] == "LittleEndian" and root.attrib["header_type"] == "UInt64"
    img = root.find("ImageData")
    piece = img.find("Piece")
    arrays = piece.find("PointData").findall("DataArray")
    assert len(arrays) == 1 and piece.find("CellData") is None
    da = arrays[0].attrib
    assert da["format"] == "appended" and da["offset"] == "0"
    assert root.find("AppendedData").attrib["encoding"] == "raw"
    dt = {"Float32": "<f4", "Float64": "<f8"}[da["type"]]
    if root.attrib.get("compressor") == "vtkZLibDataCompressor":
        nblocks, bsize, last = (int(v) for v in np.frombuffer(payload[:24], dtype="<u8"))
        csizes = [int(v) for v in np.frombuffer(payload[24:24 + 8 * nblocks], dtype="<u8")]
        pos, out = 24 + 8 * nblocks, b""
        for b, cs in enumerate(csizes):
            blk = zlib.decompress(payload[pos:pos + cs])
            assert len(blk) == (bsize if b < nblocks - 1 or last == 0 else last)
            out += blk
            pos += cs
        data, rest = np.frombuffer(out, dtype=dt), payload[pos:]
    else:
        assert "compressor" not in root.attrib
        nbytes = int(np.frombuffer(payload[:8], dtype="<u8")[0])
        data, rest = np.frombuffer(payload[8:8 + nbytes], dtype=dt), payload[8 + nbytes:]
    assert rest.strip() == b"</AppendedData>\n</VTKFile>".replace(b"\n", b"") or rest.split() == [b"</AppendedData>", b"</VTKFile>"]
    return img.attrib, piece.attrib, da, data


@pytest.mark.parametrize("compress", [0, 1])
@pytest.mark.parametrize("smooth,dtype", [(None, np.float64), (2, np.float32)])
def test_vti_independent_reader(pkg, tmp_path, smooth, dtype, compress):
    """f3: the .vti as a foreign reader sees it - extents, origin, spacing (printed as Float64 whatever the array's type: the
    coordinates are `range(origin, step=spacing)` of Float64, ExportToVTI.jl:50-52), array name / type, block
    sizes and the payload bytes, raw and zlib-compressed"""
    g = pkg.Grid(np.array([-1.0, 0.5, 2.0]), np.array([3.0, 2.5, 2.75]), 9, 2)
    s = smooth or 1
    dims = tuple(int(n) * s + 1 for n in g.c.N)
    rng = np.random.default_rng(11)
    vals = np.round(rng.normal(size=dims[2] * dims[1] * dims[0]), 2).astype(dtype)   # (compressible)
    path = pkg.exportSdfToVTI(str(tmp_path / "f"), g, vals.reshape(dims[2], dims[1], dims[0]), "distance", smooth, compress)
    img, piece, da, data = _read_vti_stdlib(path)
    ext = f"0 {dims[0] - 1} 0 {dims[1] - 1} 0 {dims[2] - 1}"
    assert img["WholeExtent"] == ext and piece["Extent"] == ext
    assert [float(v) for v in img["Origin"].split()] == [float(v) for v in g.AABB_min]
    assert [float(v) for v in img["Spacing"].split()] == [g.cell_size / s] * 3
    # Float64 text: 17 significant digits round-trip, also when the field itself is Float32
    assert all(repr(float(v)) == repr(float(w)) for v, w in zip(img["Origin"].split(), g.AABB_min))
    assert da["Name"] == "distance" and da["type"] == ("Float32" if dtype == np.float32 else "Float64")
    assert data.tobytes() == vals.tobytes()
