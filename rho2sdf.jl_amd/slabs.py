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
  sparse       only the 4x4x4 tiles that can differ from the sentinel travel (64 values + a tile id);
               every rank pre-fills its volume with the sentinel and scatters the tiles it receives.
               Interleaved partition only (tile layers are aligned by construction).
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
            of the tile payloads + ids, then every rank scatters all tiles into its pre-filled volume.
            `ops` supplies pack(local, payload, ids) -> n, unpack(payload, ids, n, volume) and
            fill(volume, value) (HIP kernels on the GPU, numpy stand-ins in the CPU tests).
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

    def gather(self, n_tiles=None):
        import torch
        import torch.distributed as dist
        if not self.sparse:
            if self.world > 1:
                dist.all_gather_into_tensor(self.gathered, self.mine)
            return
        # ---- sparse: counts, padded payload gather, scatter ----
        n_mine = int(n_tiles) if self.my_planes > 0 else 0
        counts = torch.zeros(self.world, dtype=torch.int64, device=self.device)
        mine_cnt = torch.tensor([n_mine], dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(counts, mine_cnt)
        counts = [int(c) for c in counts.tolist()]
        self.last_counts = counts
        m = max(max(counts), 1)
        payload = torch.empty(self.world * m * 64, dtype=self.dtype, device=self.device)
        ids = torch.zeros(self.world * m, dtype=torch.int32, device=self.device)
        my_payload = payload[self.rank * m * 64:(self.rank + 1) * m * 64]
        my_ids = ids[self.rank * m:(self.rank + 1) * m]
        if n_mine:
            got = self.ops.pack(self.my_slab, my_payload, my_ids)
            assert got == n_mine, (got, n_mine)
        self.ops.fill(self.full, SENTINEL)
        dist.all_gather_into_tensor(payload, my_payload)
        dist.all_gather_into_tensor(ids, my_ids)
        for r, c in enumerate(counts):
            if c:
                self.ops.unpack(payload[r * m * 64:(r * m + c) * 64], ids[r * m:r * m + c], c, self.full)

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
    sg.gather(n_tiles=(st or {}).get("n_any_tiles", 0) if sg.sparse else None)
    return st
