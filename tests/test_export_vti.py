"""exportSdfToVTI (reference src/DataExport/ExportToVTI.jl:22-67): the file is parsed back here and must
carry the reference's metadata (dimensions N*smooth+1, Origin = AABB_min, Spacing = cell/smooth, point array
name) and the exact values.  Host-only code path: runs without a GPU."""
import re

import numpy as np
import pytest


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
