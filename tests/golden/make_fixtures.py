#!/usr/bin/env python3
"""Extract the reference's *data* fixtures (test/*.mat) into small .npz files.

Run in the build container only (needs /root/reference); the .npz files are
committed so nothing under /root/reference is read at test/bench time.

Data layout mirrors what `MeshInformations` (reference
src/MeshGrid/MeshInformations.jl:3-12) hands to the pipeline:
  X   (nnp, 3) float64   node coordinates (row i = node i)
  IEN (nel, 8) int64     1-based connectivity, i.e. what Julia sees after the
                         per-file base fix-up documented in SURVEY.md A12
  rho (nel,)   float64   element densities
No reference source code is copied - these are inputs only.
"""
import os, subprocess, tempfile
import numpy as np
import scipy.io

REF = "/root/reference/test"
OUT = os.path.dirname(os.path.abspath(__file__))


def from_v5(name, ien_is_one_based):
    d = scipy.io.loadmat(os.path.join(REF, name + ".mat"))
    rho = np.asarray(d["rho"], dtype=np.float64).reshape(-1)
    X = np.asarray(d["msh"]["X"][0, 0], dtype=np.float64)      # (3, nnp)
    IEN = np.asarray(d["msh"]["IEN"][0, 0], dtype=np.int64)    # (8, nel)
    if not ien_is_one_based:
        IEN = IEN + 1
    return X.T.copy(), IEN.T.copy(), rho


def from_v73(name):
    path = os.path.join(REF, name + ".mat")
    tmp = tempfile.mkdtemp()
    out = {}
    for ds, key in (("/rho", "rho"), ("/msh/X", "X"), ("/msh/IEN", "IEN")):
        f = os.path.join(tmp, key + ".bin")
        subprocess.check_call(["/opt/conda/bin/h5dump", "-d", ds, "-b", "LE", "-o", f, path],
                              stdout=subprocess.DEVNULL)
        out[key] = f
    rho = np.fromfile(out["rho"], dtype="<f8")
    X = np.fromfile(out["X"], dtype="<f8").reshape(-1, 3)
    IEN = np.fromfile(out["IEN"], dtype="<i8").reshape(-1, 8) + 1   # file is 0-based
    return X, IEN, rho


def main():
    sets = {
        "sphere": from_v73("sphere"),
        # beam files are already 1-based (runtests.jl:193 subtracts the +1 again)
        "beam_vfrac_03": from_v5("cantilever_beam_vfrac_03", True),
        "beam_vfrac_04": from_v5("cantilever_beam_vfrac_04", True),
        "chapadlo": from_v5("chapadlo", False),
    }
    for k, (X, IEN, rho) in sets.items():
        assert IEN.min() == 1 and IEN.max() == X.shape[0], (k, IEN.min(), IEN.max(), X.shape)
        assert rho.shape[0] == IEN.shape[0]
        np.savez_compressed(os.path.join(OUT, k + ".npz"), X=X, IEN=IEN.astype(np.int32), rho=rho)
        print(k, X.shape, IEN.shape, rho.shape, float(rho.mean()))


if __name__ == "__main__":
    main()
