"""Z partition of the regular grid across ranks + the collective that stitches the volume
(SURVEY.md section 8(e)).  One process per GPU; backend "nccl" (= RCCL over xGMI) on MI355X, "gloo"
in the CPU tests.  The slab computation itself is a callable so the same code path is exercised on
CPU (tests) and on GPU (bench.py).

Partitions (equal-sized contributions to ONE all_gather_into_tensor):
  contiguous   rank r computes planes [r*per, (r+1)*per)                 (nz padded to world*per)
  interleaved  rank r computes the 4-plane tile layers t with t % world == r (balanced when the
               material is not uniform in z); after the gather the layers are put back in order

Stitching:
  dense        the whole Float64 volume travels (8 B/voxel)
  sparse       only the 4x4x4 tiles that can differ from the sentinel travel, in ONE all_gather_into_tensor
               (plus a count exchange on the very first step, or when the tile counts outgrow the agreed capacity):
               tiles with band items as 64 values + a tile id, tiles that only carry the sign (every voxel
               +-1e10) as a 64-bit mask + a tile id; every rank pre-fills its volume with the sentinel and
               scatters what it receives.  Interleaved partition only (tile layers are aligned by construction).
"""
TILE = 4  # planes per tile layer (the kernels work on 4x4x4 voxel tiles)
SENTINEL = -1.0e10   # dist = 1e10 (untouched) * sign = -1


def slab_bounds(nz, world):
    """contiguous: equal Z-slabs; nz is padded up to a multiple of `world` so every rank contributes the
    same number of planes.  Returns (planes_per_rank, [(k0, k1)] per rank); trailing ranks may get
    short or empty slabs."""
    per = (nz + world - 1) // world
    return per, [(min(r * per, nz), min((r + 1) * per, nz)) for r in range(world)]


def interleaved_layers(nz, world, rank):
    """interleaved: (layers owned by `rank`, layers per rank after padding)"""
    layers = (nz + TILE - 1) // TILE
    owned = (layers - rank + world - 1) // world if layers > rank else 0
    return owned, (layers + world - 1) // world


class SlabGather:
    """Owns the stitched volume of one rank.

    dense:  rank r writes its part in place at its offset of `gathered`; one all_gather_into_tensor.
    sparse: rank r computes into `local`, packs its non-sentinel tiles, one (padded) all_gather_into_tensor
            of [payloads | ids | sign masks | mask ids], then every rank scatters everything into its
            pre-filled volume.  `ops` supplies pack2(local, payload, ids, masks, mask_ids) -> (n_full, n_mask),
            unpack(payload, ids, n, volume), unpack_masks(masks, mask_ids, n, volume) and fill(volume, value)
            (HIP kernels on the GPU, numpy stand-ins in the CPU tests).
    """

    def __init__(self, dims, rank, world, device, dtype=None, interleaved=False, sparse=False, ops=None):
        import torch
        self.nx, self.ny, self.nz = dims
        self.rank, self.world = rank, world
        self.plane = self.nx * self.ny
        self.device = device
        self.dtype = dtype or torch.float64
        self.interleaved = bool(interleaved) and world > 1
        self.sparse = bool(sparse) and self.interleaved
        self.ops = ops
        if self.sparse and ops is None:
            raise ValueError("sparse stitching needs pack/unpack/fill ops")
        if self.interleaved:
            self.owned_layers, self.layers_per_rank = interleaved_layers(self.nz, world, rank)
            self.per = TILE * self.layers_per_rank
            self.k0, self.k1 = 0, self.nz
            self.my_planes = TILE * self.owned_layers
        else:
            self.per, self.bounds = slab_bounds(self.nz, world)
            self.k0, self.k1 = self.bounds[rank]
            self.my_planes = self.k1 - self.k0
        self._ordered = None
        self._cap = None          # sparse stitching: agreed segment capacities (band tiles, sign-only tiles)
        self._buf = self._hdr = self._mine_hdr = None
        self.n_collectives = 0
        self.last_counts = None
        if self.sparse:
            self.local = torch.empty(max(self.my_planes, 1) * self.plane, dtype=self.dtype, device=device)
            self.full = torch.empty(self.nz * self.plane, dtype=self.dtype, device=device)
            self.gathered = None
            self.mine = self.local
        else:
            self.gathered = torch.empty(world * self.per * self.plane, dtype=self.dtype, device=device)
            self.mine = self.gathered[rank * self.per * self.plane:(rank + 1) * self.per * self.plane]

    @property
    def my_slab(self):
        """view of exactly the voxels this rank computes"""
        return self.mine[:self.my_planes * self.plane]

    @staticmethod
    def _segment_views(seg, mf, mm):
        """views into one rank's segment (a float64 tensor): band-tile payload (mf*64 f64), their ids (mf i32),
        sign-only masks (mm i64), their ids (mm i32); every part starts on an 8-byte boundary"""
        import torch
        hf, hm = (mf + 1) // 2, (mm + 1) // 2
        o = mf * 64
        payload = seg[:o]
        ids = seg[o:o + hf].view(torch.int32)[:mf]
        o += hf
        masks = seg[o:o + mm].view(torch.int64)
        o += mm
        mids = seg[o:o + hm].view(torch.int32)[:mm]
        return payload, ids, masks, mids

    def gather(self, counts=None):
        """counts = (band tiles, sign-only tiles) of this rank's last run (sparse stitching only)"""
        import torch
        import torch.distributed as dist
        if not self.sparse:
            if self.world > 1:
                dist.all_gather_into_tensor(self.gathered, self.mine)
            return
        # ---- sparse: ONE padded all-gather of [counts | payload | ids | masks | mask ids], scatter ----
        # Segment capacities are agreed once (a count exchange on the first step) and kept with 25 % head-room;
        # every segment starts with its rank's two counts, so later steps need no separate exchange.  If some
        # rank's tiles do not fit any more, every rank sees that in the headers it received, all grow the
        # capacities from those (true) counts and the gather is repeated - the same decision everywhere.
        nf_mine, nm_mine = (int(counts[0]), int(counts[1])) if self.my_planes > 0 else (0, 0)
        grow = lambda n: max(1, n + n // 4 + 16)
        if self._cap is None:
            allc = torch.zeros(2 * self.world, dtype=torch.int64, device=self.device)
            mine_cnt = torch.tensor([nf_mine, nm_mine], dtype=torch.int64, device=self.device)
            dist.all_gather_into_tensor(allc, mine_cnt)
            allc = allc.view(self.world, 2).tolist()
            self.n_collectives += 1
            self._cap = (grow(max(int(c[0]) for c in allc)), grow(max(int(c[1]) for c in allc)))
        filled = False
        while True:
            mf, mm = self._cap
            seglen = 2 + mf * 64 + (mf + 1) // 2 + mm + (mm + 1) // 2
            if self._buf is None or self._buf.numel() != self.world * seglen:
                # the exchange buffer lives as long as the capacities: no allocation per step
                self._buf = torch.empty(self.world * seglen, dtype=self.dtype, device=self.device)
                self._hdr = torch.empty((self.world, 2), dtype=torch.int64, pin_memory=(self.device.type == "cuda"))
                self._mine_hdr = torch.empty(2, dtype=torch.int64, pin_memory=(self.device.type == "cuda"))
            buf = self._buf
            seg = buf[self.rank * seglen:(self.rank + 1) * seglen]
            self._mine_hdr[0], self._mine_hdr[1] = nf_mine, nm_mine
            seg[:2].view(torch.int64).copy_(self._mine_hdr, non_blocking=True)
            if nf_mine <= mf and nm_mine <= mm and (nf_mine or nm_mine):
                payload, ids, masks, mids = self._segment_views(seg[2:], mf, mm)
                got = self.ops.pack2(self.my_slab, payload, ids, masks, mids)
                assert tuple(got) == (nf_mine, nm_mine), (got, nf_mine, nm_mine)
            # the collective runs on the communicator's stream: the sentinel pre-fill of the whole volume
            # (1 GB of HBM writes at 512^3) overlaps with it instead of preceding it
            work = dist.all_gather_into_tensor(buf, seg, async_op=True)
            self.n_collectives += 1
            if not filled:
                self.ops.fill(self.full, SENTINEL)
                filled = True
            work.wait()
            hdr_view = buf.view(self.world, seglen)[:, :2]
            if hasattr(self.ops, "unpack_all"):
                # device-side counts: the scatter of every rank's segment is queued (two launches) before the host has
                # seen a single count; the headers come back beside it and are only needed for the overflow decision
                self._hdr.copy_(hdr_view.contiguous().view(torch.int64), non_blocking=True)
                self.ops.unpack_all(buf, self.world, seglen, mf, mm, self.full)
                if self.device.type == "cuda":
                    torch.cuda.current_stream(self.device).synchronize()
                allc = self._hdr.tolist()
                done = all(int(c[0]) <= mf and int(c[1]) <= mm for c in allc)
            else:
                allc = hdr_view.contiguous().view(torch.int64).tolist()
                done = all(int(c[0]) <= mf and int(c[1]) <= mm for c in allc)
                if done:
                    for r, (cf, cm) in enumerate(allc):
                        payload, ids, masks, mids = self._segment_views(buf[r * seglen + 2:(r + 1) * seglen], mf, mm)
                        if cf:
                            self.ops.unpack(payload[:int(cf) * 64], ids[:int(cf)], int(cf), self.full)
                        if cm:
                            self.ops.unpack_masks(masks[:int(cm)], mids[:int(cm)], int(cm), self.full)
            if done:
                break
            self._cap = (grow(max(int(c[0]) for c in allc)), grow(max(int(c[1]) for c in allc)))
            filled = False       # (segments that did fit were scattered already: start the volume again)
        self.last_counts = allc
        self.last_bytes = buf.numel() * buf.element_size()

    def volume(self):
        """the stitched (nz, ny, nx) volume (padding trimmed, tile layers back in lattice order)"""
        if self.sparse:
            return self.full.view(self.nz, self.ny, self.nx)
        if not self.interleaved:
            return self.gathered[:self.nz * self.plane].view(self.nz, self.ny, self.nx)
        import torch
        g = self.gathered.view(self.world, self.layers_per_rank, TILE * self.plane)
        if self._ordered is None:
            self._ordered = torch.empty_like(self.gathered)
        # layer t of rank r is global layer t*world + r
        self._ordered.view(self.layers_per_rank, self.world, TILE * self.plane).copy_(g.transpose(0, 1))
        return self._ordered[:self.nz * self.plane].view(self.nz, self.ny, self.nx)


def run_step(sg, compute_slab):
    """one distributed step: compute my part (if not empty), then the stitching collective.
    compute_slab(k0, k1, out, zstride, zphase) -> stats dict (needs 'n_any_tiles' for sparse stitching)"""
    st = None
    if sg.my_planes > 0:
        if sg.interleaved:
            st = compute_slab(0, sg.nz, sg.my_slab, sg.world, sg.rank)
        else:
            st = compute_slab(sg.k0, sg.k1, sg.my_slab, 1, 0)
    s = st or {}
    sg.gather(counts=(s.get("n_active_tiles", 0), s.get("n_sign_only_tiles", 0)) if sg.sparse else None)
    return st
