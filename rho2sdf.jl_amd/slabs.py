"""Z partition of the regular grid across ranks + the single all-gather that stitches the volume
(SURVEY.md section 8(e)).  One process per GPU; backend "nccl" (= RCCL over xGMI) on MI355X, "gloo"
in the CPU tests.  The slab computation itself is a callable so the same code path is exercised on
CPU (tests) and on GPU (bench.py).

Two partitions, both with equal-sized contributions to ONE all_gather_into_tensor:
  contiguous   rank r computes planes [r*per, (r+1)*per)                 (nz padded to world*per)
  interleaved  rank r computes the 4-plane tile layers t with t % world == r (balanced when the
               material is not uniform in z); after the gather the layers are put back in order
"""
TILE = 4  # planes per tile layer (the kernels work on 4x4x4 voxel tiles)


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
    """Owns the gathered volume; rank r writes its part in place at its offset and one
    all_gather_into_tensor makes every rank hold the whole grid."""

    def __init__(self, dims, rank, world, device, dtype=None, interleaved=False):
        import torch
        self.nx, self.ny, self.nz = dims
        self.rank, self.world = rank, world
        self.plane = self.nx * self.ny
        self.interleaved = bool(interleaved) and world > 1
        if self.interleaved:
            self.owned_layers, self.layers_per_rank = interleaved_layers(self.nz, world, rank)
            self.per = TILE * self.layers_per_rank
            self.k0, self.k1 = 0, self.nz
            self.my_planes = TILE * self.owned_layers
        else:
            self.per, self.bounds = slab_bounds(self.nz, world)
            self.k0, self.k1 = self.bounds[rank]
            self.my_planes = self.k1 - self.k0
        self.gathered = torch.empty(world * self.per * self.plane, dtype=dtype or torch.float64, device=device)
        self.mine = self.gathered[rank * self.per * self.plane:(rank + 1) * self.per * self.plane]
        self._ordered = None

    @property
    def my_slab(self):
        """view of exactly the voxels this rank computes"""
        return self.mine[:self.my_planes * self.plane]

    def gather(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.all_gather_into_tensor(self.gathered, self.mine)

    def volume(self):
        """the stitched (nz, ny, nx) volume (padding trimmed, tile layers back in lattice order)"""
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
    """one distributed step: compute my part (if not empty), then the all-gather.
    compute_slab(k0, k1, out, zstride, zphase)"""
    st = None
    if sg.my_planes > 0:
        if sg.interleaved:
            st = compute_slab(0, sg.nz, sg.my_slab, sg.world, sg.rank)
        else:
            st = compute_slab(sg.k0, sg.k1, sg.my_slab, 1, 0)
    sg.gather()
    return st
