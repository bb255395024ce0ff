"""Z-slab partition of the regular grid across ranks + the single all-gather that stitches
the volume.  One process per GPU; backend "nccl" (= RCCL over xGMI)
on MI355X, "gloo" in the CPU tests.  The slab computation itself is a callable so the same
code path is exercised on CPU (tests) and on GPU (bench.py)."""


def slab_bounds(nz, world):
    """equal Z-slabs; nz is padded up to a multiple of `world` so every rank contributes the
    same number of planes to the all-gather.  Returns (planes_per_rank, [(k0, k1)] per rank);
    trailing ranks may get short or empty slabs."""
    per = (nz + world - 1) // world
    return per, [(min(r * per, nz), min((r + 1) * per, nz)) for r in range(world)]


class SlabGather:
    """Owns the gathered volume; rank r writes its slab in place at its offset and one
    all_gather_into_tensor makes every rank hold the whole grid."""

    def __init__(self, dims, rank, world, device, dtype=None):
        import torch
        self.nx, self.ny, self.nz = dims
        self.rank, self.world = rank, world
        self.plane = self.nx * self.ny
        self.per, self.bounds = slab_bounds(self.nz, world)
        self.k0, self.k1 = self.bounds[rank]
        self.gathered = torch.empty(world * self.per * self.plane, dtype=dtype or torch.float64, device=device)
        self.mine = self.gathered[rank * self.per * self.plane:(rank + 1) * self.per * self.plane]

    @property
    def my_slab(self):
        """view of exactly the voxels this rank computes, planes [k0, k1)"""
        return self.mine[:(self.k1 - self.k0) * self.plane]

    def gather(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.all_gather_into_tensor(self.gathered, self.mine)

    def volume(self):
        """the stitched (nz, ny, nx) volume (padding planes trimmed)"""
        return self.gathered[:self.nz * self.plane].view(self.nz, self.ny, self.nx)


def run_step(sg, compute_slab):
    """one distributed step: compute my slab (if not empty), then the all-gather"""
    st = None
    if sg.k1 > sg.k0:
        st = compute_slab(sg.k0, sg.k1, sg.my_slab)
    sg.gather()
    return st
