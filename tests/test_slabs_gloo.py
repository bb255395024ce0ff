"""N > 1 path on CPU: world_size 2 and 3 over gloo.  The slab computation is the oracle's full
volume sliced per rank (the kernels need a GPU; the partition / padding / all-gather / trim
logic under test is exactly what bench.py runs over RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_fixture


class NumpyTileOps:
    """CPU stand-ins for r2s_plan_pack_tiles2_dev / r2s_unpack_tiles_dev / r2s_unpack_masks_dev / r2s_fill_dev
    (same tile layout)"""

    def __init__(self, dims, world, rank):
        self.nx, self.ny, self.nz = dims
        self.world, self.rank = world, rank
        self.ntx, self.nty = (self.nx + 3) // 4, (self.ny + 3) // 4

    def _tiles(self, local):
        v = local.view(-1, self.ny, self.nx).numpy()
        out = []
        for tz in range(v.shape[0] // 4):
            for ty in range(self.nty):
                for tx in range(self.ntx):
                    blk = np.full((4, 4, 4), -1.0e10)
                    sub = v[4 * tz:4 * tz + 4, 4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4]
                    kg = 4 * (tz * self.world + self.rank)
                    sub = sub[:max(0, min(4, self.nz - kg))]
                    blk[:sub.shape[0], :sub.shape[1], :sub.shape[2]] = sub
                    if (blk != -1.0e10).any():
                        out.append((((tz * self.world + self.rank) * self.nty + ty) * self.ntx + tx, blk.reshape(-1)))
        return out

    @staticmethod
    def _sign_only(blk):
        return bool((np.abs(blk) == 1.0e10).all())

    def counts(self, local):
        """(tiles that travel as 64 values, tiles that travel as a sign mask)"""
        t = self._tiles(local)
        nm = sum(1 for _, blk in t if self._sign_only(blk))
        return len(t) - nm, nm

    def pack2(self, local, payload, ids, masks, mask_ids):
        nf = nm = 0
        for tid, blk in self._tiles(local):
            if self._sign_only(blk):
                m = 0
                for l in range(64):
                    if blk[l] > 0:
                        m |= 1 << l
                masks[nm] = m - (1 << 64) if m >= (1 << 63) else m     # int64 storage of the 64-bit mask
                mask_ids[nm] = tid
                nm += 1
            else:
                payload[nf * 64:(nf + 1) * 64] = torch.from_numpy(blk)
                ids[nf] = tid
                nf += 1
        return nf, nm

    def unpack(self, payload, ids, n, vol):
        v = vol.view(self.nz, self.ny, self.nx).numpy()
        for w in range(n):
            tid = int(ids[w])
            tx, ty, tz = tid % self.ntx, (tid // self.ntx) % self.nty, tid // (self.ntx * self.nty)
            blk = payload[w * 64:(w + 1) * 64].numpy().reshape(4, 4, 4)
            sub = v[4 * tz:4 * tz + 4, 4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4]
            sub[...] = blk[:sub.shape[0], :sub.shape[1], :sub.shape[2]]

    def unpack_masks(self, masks, mask_ids, n, vol):
        v = vol.view(self.nz, self.ny, self.nx).numpy()
        for w in range(n):
            tid, m = int(mask_ids[w]), int(masks[w]) & ((1 << 64) - 1)
            tx, ty, tz = tid % self.ntx, (tid // self.ntx) % self.nty, tid // (self.ntx * self.nty)
            blk = np.array([1.0e10 if (m >> l) & 1 else -1.0e10 for l in range(64)]).reshape(4, 4, 4)
            sub = v[4 * tz:4 * tz + 4, 4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4]
            sub[...] = blk[:sub.shape[0], :sub.shape[1], :sub.shape[2]]

    @staticmethod
    def fill(t, value):
        t.fill_(value)


class NumpyTileOpsAll(NumpyTileOps):
    """+ the stand-in of r2s_unpack_segments_dev: every rank's segment in one call, counts read from the segment headers
    (a segment whose counts exceed the capacities is skipped, as on the device)"""

    def unpack_all(self, buf, world, seglen, mf, mm, vol):
        from rho2sdf_jl_amd import slabs
        for r in range(world):
            seg = buf[r * seglen:(r + 1) * seglen]
            cf, cm = (int(c) for c in seg[:2].view(torch.int64))
            if cf > mf or cm > mm:
                continue
            payload, ids, masks, mids = slabs.SlabGather._segment_views(seg[2:], mf, mm)
            self.unpack(payload, ids, cf, vol)
            self.unpack_masks(masks, mids, cm, vol)


def _worker(rank, world, port, ref_path, dims, interleaved, sparse=False, device_counts=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    graft.load_package()
    from rho2sdf_jl_amd import slabs
    ref = torch.from_numpy(np.load(ref_path))
    ops = (NumpyTileOpsAll if device_counts else NumpyTileOps)(dims, world, rank)
    sg = slabs.SlabGather(dims, rank, world, torch.device("cpu"), interleaved=interleaved, sparse=sparse, ops=ops)
    plane = dims[0] * dims[1]
    nz = dims[2]

    def compute_slab(k0, k1, out, zstride, zphase):
        if zstride == 1:
            out.copy_(ref[k0 * plane:k1 * plane])
        else:   # what r2s_plan_run_dev(zstride, zphase) returns: the owned 4-plane layers, consecutively
            o = out.view(-1, plane)
            o.fill_(float("nan"))
            for i in range(o.shape[0] // 4):
                for l in range(4):
                    k = 4 * (i * zstride + zphase) + l
                    if k < nz:
                        o[4 * i + l] = ref[k * plane:(k + 1) * plane]
            o[torch.isnan(o)] = -1.0e10      # planes beyond the grid
            nf, nm = ops.counts(out)
            return {"n_active_tiles": nf, "n_sign_only_tiles": nm}

    slabs.run_step(sg, compute_slab)
    ok = torch.equal(sg.volume().reshape(-1), ref)
    if sparse:
        # later steps: no count exchange (capacities agreed on the first step); more tiles than the capacity: one
        # repeated gather, decided identically on every rank from the headers; then steady state again
        full_ref = ref
        assert sg.n_collectives == 2
        ref = full_ref.clone()
        ref.view(nz, -1)[nz // 3:] = -1.0e10                 # far fewer tiles
        slabs.run_step(sg, compute_slab)
        ok = ok and torch.equal(sg.volume().reshape(-1), ref) and sg.n_collectives == 3
        cap = sg._cap
        sg._cap = (1, 1)                                       # pretend the agreed capacity is far too small
        ref = full_ref
        slabs.run_step(sg, compute_slab)
        ok = ok and torch.equal(sg.volume().reshape(-1), ref) and sg.n_collectives == 5 and sg._cap[0] > 1
        slabs.run_step(sg, compute_slab)
        ok = ok and torch.equal(sg.volume().reshape(-1), ref) and sg.n_collectives == 6
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    if flag.item() != 1:
        raise SystemExit(3)


@pytest.mark.parametrize("interleaved", [False, True])
@pytest.mark.parametrize("world", [2, 3])
def test_zslab_allgather(oracle, tmp_path, world, interleaved):
    X, IEN, rho = load_fixture("sphere")
    rn = oracle.dense_in_nodes(X, IEN, rho)
    g = oracle.grid_make(X.min(0), X.max(0), 10)          # 17 planes: not divisible by 2 or 3
    d, _, _ = oracle.eval_distances(X, IEN, rn, 0.5, g, 1.1, want_xp=False)
    sdf = d * oracle.sign_detection(X, IEN, rn, 0.5, g)
    ref_path = str(tmp_path / "ref.npy")
    np.save(ref_path, sdf)
    port = 29500 + (os.getpid() % 500) + world
    mp.spawn(_worker, args=(world, port + (10 if interleaved else 0), ref_path, g.dims, interleaved), nprocs=world, join=True)


def test_slab_bounds():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    graft.load_package()
    from rho2sdf_jl_amd import slabs
    per, b = slabs.slab_bounds(512, 8)
    assert per == 64 and b[0] == (0, 64) and b[7] == (448, 512)
    per, b = slabs.slab_bounds(257, 4)                    # chapadlo config: 257 planes -> 260
    assert per == 65 and b[3] == (195, 257)
    per, b = slabs.slab_bounds(5, 8)
    assert per == 1 and b[4] == (4, 5) and b[5] == (5, 5) and b[7] == (5, 5)
    assert slabs.interleaved_layers(512, 8, 3) == (16, 16)
    assert slabs.interleaved_layers(257, 4, 0) == (17, 17) and slabs.interleaved_layers(257, 4, 1) == (16, 17)
    assert slabs.interleaved_layers(17, 8, 5) == (0, 1)


@pytest.mark.parametrize("device_counts", [False, True])
@pytest.mark.parametrize("world", [2, 3])
def test_sparse_tile_allgather(oracle, tmp_path, world, device_counts):
    """sparse stitching: only non-sentinel 4x4x4 tiles travel, sign-only tiles as 64-bit masks (counts + ONE padded
    all-gather + scatter); device_counts: the scatter reads the counts from the segment headers itself
    (r2s_unpack_segments_dev on the GPU, NumpyTileOpsAll here) instead of waiting for the host to read them"""
    X, IEN, rho = load_fixture("sphere")
    rn = oracle.dense_in_nodes(X, IEN, rho)
    g = oracle.grid_make(X.min(0), X.max(0), 10)
    d, _, _ = oracle.eval_distances(X, IEN, rn, 0.5, g, 1.1, want_xp=False)
    sdf = d * oracle.sign_detection(X, IEN, rn, 0.5, g)
    assert (sdf == -1.0e10).sum() > 1000          # the case has plenty of sentinel tiles to skip
    ref_path = str(tmp_path / "ref.npy")
    np.save(ref_path, sdf)
    port = 29700 + (os.getpid() % 200) + world + (20 if device_counts else 0)
    mp.spawn(_worker, args=(world, port, ref_path, g.dims, True, True, device_counts), nprocs=world, join=True)
