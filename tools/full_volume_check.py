#!/usr/bin/env python3
"""Every voxel of the north-star volume (512^3, 97 336 HEX8) against the CPU oracle - not the plane sample of
`bench.py --check`.  The GPU result is written to a scratch file; `--workers` single-thread oracle processes
(test infrastructure, like bench.py's cpu_baseline) each compute the Z planes k % workers == w and compare.
  python tools/full_volume_check.py [--grid 512] [--mesh 46] [--workers 16]"""
import argparse, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as graft

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=512)
ap.add_argument("--mesh", type=int, default=46)
ap.add_argument("--workers", type=int, default=16)
ap.add_argument("--worker", type=int, default=None, help=argparse.SUPPRESS)
ap.add_argument("--volume", default=None, help=argparse.SUPPRESS)
a = ap.parse_args()

graft.load_package()
from rho2sdf_jl_amd import synthetic
X, IEN, rn = synthetic.hex_mesh(a.mesh)
n_max = synthetic.grid_n_max_for_points(a.grid)

if a.worker is not None:      # child: oracle on its planes, compare with the stored GPU volume
    O = graft.load_oracle()
    og = O.grid_make(X.min(0), X.max(0), n_max, 3)
    nx, ny, nz = og.dims
    O.set_k_sampling(a.workers, a.worker)
    dist, _, _ = O.eval_distances(X, IEN, rn, 0.5, og, 1.1, want_xp=False)
    sign = O.sign_detection(X, IEN, rn, 0.5, og)
    want = dist.reshape(nz, ny, nx)[a.worker::a.workers] * sign.reshape(nz, ny, nx)[a.worker::a.workers]
    got = np.load(a.volume, mmap_mode="r")[a.worker::a.workers]
    sent = np.abs(want) > 1e9
    real = ~sent
    rel = np.abs(got[real] - want[real]) / np.maximum(np.abs(want[real]), 1e-300)
    print(json.dumps({"voxels": int(want.size), "real": int(real.sum()),
                      "sentinel_mismatch": int((sent != (np.abs(got) > 1e9)).sum()),
                      "sign_mismatch": int((np.sign(got) != np.sign(want)).sum()),
                      "not_bit_equal": int((got != want).sum()), "max_rel_err": float(rel.max()) if rel.size else 0.0}))
    sys.exit(0)

pkg = graft.load_built()
import torch
g = pkg.Grid(X.min(0), X.max(0), n_max, 3)
nx, ny, nz = g.dims
dev = torch.device("cuda:0")
dX, dI, dR = (torch.from_numpy(t).to(dev) for t in (X, IEN, rn))
out = torch.empty(g.ngp, dtype=torch.float64, device=dev)
pkg.DevicePlan(0).run(dX, dI, dR, 0.5, g, sdf=out)
with tempfile.TemporaryDirectory() as tmp:
    vol = os.path.join(tmp, "gpu_volume.npy")
    np.save(vol, out.cpu().numpy().reshape(nz, ny, nx))
    base = [sys.executable, os.path.abspath(__file__), "--grid", str(a.grid), "--mesh", str(a.mesh),
            "--workers", str(a.workers), "--volume", vol]
    procs = [subprocess.Popen(base + ["--worker", str(w)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for w in range(a.workers)]
    tot = {"voxels": 0, "real": 0, "sentinel_mismatch": 0, "sign_mismatch": 0, "not_bit_equal": 0, "max_rel_err": 0.0}
    for p in procs:
        o, e = p.communicate()
        if p.returncode != 0:
            raise SystemExit("worker failed: " + e[-400:])
        r = json.loads(o.strip().splitlines()[-1])
        for k in tot:
            tot[k] = max(tot[k], r[k]) if k == "max_rel_err" else tot[k] + r[k]
print(json.dumps({"grid": [nx, ny, nz], "elements": int(len(IEN)), **tot}))
