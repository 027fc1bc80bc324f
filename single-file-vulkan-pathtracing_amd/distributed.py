"""Multi-GPU sharding of the radiance loop: one process per GPU, pixel tiles interleaved over the
ranks, ONE collective per presented image.

The path shards with no data-path exchange: a pixel's radiance depends only on (pixel, sample
index, scene) (raygen.rgen:47-48, 88-90 touch nothing but the own texel).  Every rank holds the
whole scene and builds the same LBVH; rank r renders the 8x8 pixel tiles with
(tile_x + tile_y) % world == r -- interleaved, because contiguous bands are badly unbalanced
(44 % of the image, the border, terminates after one ray).

Presenting: `pt_film_present` (csrc/present_rccl.hip) packs the tiles a rank owns, gathers them on
the root with ncclSend / ncclRecv over the library's OWN RCCL communicator (W*H*12/N bytes per rank)
and unpacks them into a separate image; a rank's accumulation film is never written by the
collective, so progressive rendering can continue and present again.  `Presenter` below is the glue
under `torch.distributed`: the launcher's process group only carries the 128-byte RCCL unique id
to the ranks (and, in the CPU emulation of the tests, the packed tiles over gloo).
"""
import os

import numpy as np

TILE = 8


def tile_owner(width, height, world):
    """-> int array [H, W]: the rank that renders each pixel (mirror of ensure_work() in
    csrc/film_work.hip ptw_ensure_work)."""
    ty, tx = np.meshgrid(np.arange(height) // TILE, np.arange(width) // TILE, indexing="ij")
    return (tx + ty) % world


def owned_mask(width, height, rank, world):
    return tile_owner(width, height, world) == rank


def tile_list(width, height, rank, world):
    """-> [(tile_x, tile_y)] of a rank in the order the device enumerates them (row major)."""
    tiles_x, tiles_y = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    return [(tx, ty) for ty in range(tiles_y) for tx in range(tiles_x) if (tx + ty) % world == rank]


def pack_tiles_host(film, rank, world):
    """numpy mirror of k_pack_tiles: [n_tiles, 64, 3] float32 (pixels beyond the image edge are 0)."""
    h, w = film.shape[:2]
    tl = tile_list(w, h, rank, world)
    out = np.zeros((len(tl), TILE, TILE, 3), np.float32)
    for k, (tx, ty) in enumerate(tl):
        blk = film[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE]
        out[k, :blk.shape[0], :blk.shape[1]] = blk
    return out.reshape(len(tl), TILE * TILE, 3)


def unpack_tiles_host(packed, image, rank, world):
    """numpy mirror of k_unpack_tiles: scatter a rank's packed tiles into `image` [H, W, 3]."""
    h, w = image.shape[:2]
    for k, (tx, ty) in enumerate(tile_list(w, h, rank, world)):
        blk = packed[k].reshape(TILE, TILE, 3)
        y1, x1 = min((ty + 1) * TILE, h), min((tx + 1) * TILE, w)
        image[ty * TILE:y1, tx * TILE:x1] = blk[:y1 - ty * TILE, :x1 - tx * TILE]
    return image


def gather_present_host(film, rank, world, dst=0, group=None):
    """The presentation collective on host arrays over any torch.distributed backend (the world-2 gloo test): every
    rank packs its tiles, the root gathers them and unpacks.  -> the presented image on `dst`, None elsewhere;
    `film` is not modified."""
    import torch
    import torch.distributed as dist
    mine = torch.from_numpy(pack_tiles_host(film, rank, world))
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return unpack_tiles_host(mine.numpy(), np.zeros_like(film), rank, world)
    h, w = film.shape[:2]
    counts = [len(tile_list(w, h, r, world)) for r in range(world)]
    if rank == dst:
        bufs = [torch.zeros((counts[r], TILE * TILE, 3), dtype=torch.float32) for r in range(world)]
        bufs[dst] = mine
        reqs = [dist.irecv(bufs[r], src=r, group=group) for r in range(world) if r != dst and counts[r]]
        for q in reqs:
            q.wait()
        image = np.zeros_like(film)
        for r in range(world):
            unpack_tiles_host(bufs[r].numpy(), image, r, world)
        return image
    if counts[rank]:
        dist.send(mine, dst=dst, group=group)
    return None


def selftest_film(width, height, rank, world):
    """The film of the presentation self-test: this rank's tiles carry ITS colour -- (rank + 1, 100 + rank, tile x + 1000 * tile y) -- and
    every other pixel is zero, as after a render of its shard."""
    ty, tx = np.meshgrid(np.arange(height) // TILE, np.arange(width) // TILE, indexing="ij")
    film = np.zeros((height, width, 3), np.float32)
    own = (tx + ty) % world == rank
    film[..., 0] = np.where(own, rank + 1, 0)
    film[..., 1] = np.where(own, 100 + rank, 0)
    film[..., 2] = np.where(own, tx + 1000 * ty, 0)
    return film


def selftest_check(image, world):
    """The presented image of the self-test on the root: every pixel must carry the colour of the rank that OWNS its tile.
    -> {"ok", "ranks_seen", "wrong_pixels", "tiles_per_rank"}"""
    h, w = image.shape[:2]
    ty, tx = np.meshgrid(np.arange(h) // TILE, np.arange(w) // TILE, indexing="ij")
    owner = (tx + ty) % world
    good = (image[..., 0] == owner + 1) & (image[..., 1] == 100 + owner) & (image[..., 2] == tx + 1000 * ty)
    tiles = [int(len(tile_list(w, h, r, world))) for r in range(world)]
    seen = sorted(set(int(x) - 1 for x in np.unique(image[..., 0]) if x >= 1))
    return {"ok": bool(good.all()), "ranks_seen": seen, "wrong_pixels": int((~good).sum()), "tiles_per_rank": tiles}


class Presenter:
    """bench.py's glue: this rank's film -> the presented image on rank 0, once per call.

    Real run: the library's own RCCL communicator (`Comm`), created from a unique id that rank 0 makes and
    torch.distributed broadcasts.  PT_BENCH_EMULATE (all ranks on GPU 0, gloo): the same pack / unpack kernels with
    the packed tiles carried by gloo -- RCCL cannot put two ranks on one device."""

    def __init__(self, pt, ctx, film, film_tensor, rank, world, cdev, emulate):
        import torch
        import torch.distributed as dist
        self.pt, self.ctx, self.film, self.rank, self.world, self.emulate = pt, ctx, film, rank, world, emulate
        self.torch, self.dist = torch, dist
        dev = film_tensor.device
        self.image = torch.zeros_like(film_tensor) if rank == 0 else None
        self.comm = None
        self.ranks_seen = None
        self.fallback = None
        if emulate:
            self.counts = [pt.film_tile_count(film, r, world) for r in range(world)]
            self.packed = torch.zeros((max(self.counts[rank], 1), 64, 3), dtype=torch.float32, device=dev)
            self.ranks_seen = dist.get_world_size()
        else:
            # the library's communicator; if ANY rank cannot create it (no librccl to dlopen, ncclCommInitRank refused)
            # every rank falls back to the same gather through torch.distributed's RCCL process group, so that a
            # multi-GPU run still presents its image -- and says so in the bench line (`describe`)
            err = None
            try:
                if os.environ.get("PT_PRESENT_FORCE_TORCH"):   # tests: take the fallback
                    raise RuntimeError("PT_PRESENT_FORCE_TORCH")
                box = [pt.Comm.unique_id() if rank == 0 else None]
            except Exception as e:          # rank 0 could not even make the id: the others must not wait for a peer
                box, err = [None], e
            dist.broadcast_object_list(box, src=0)
            if box[0] is not None:
                try:
                    self.comm = pt.Comm(ctx, box[0], world, rank)
                    self.ranks_seen = self.comm.ranks()
                except Exception as e:
                    err = e
            else:
                err = err or RuntimeError("rank 0 has no RCCL unique id")
            bad = torch.tensor([1 if err is not None else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if int(bad.item()):
                if self.comm:
                    self.comm.close()
                    self.comm = None
                self.fallback = repr(err) if err is not None else "another rank failed to create the communicator"
                self.counts = [pt.film_tile_count(film, r, world) for r in range(world)]
                self.packed = torch.zeros((max(self.counts[rank], 1), 64, 3), dtype=torch.float32, device=dev)
                self.ranks_seen = dist.get_world_size()

    def describe(self):
        if self.emulate:
            return "packed tiles gathered over gloo (emulation)"
        if self.fallback:
            return ("one RCCL gather of the packed tiles to rank 0 through torch.distributed send/recv (the library's own "
                    "communicator could not be created: %s)" % self.fallback)
        return "one RCCL gather of the packed tiles to rank 0 (pt_film_present: ncclSend/ncclRecv, own communicator)"

    def present(self):
        pt, torch, dist = self.pt, self.torch, self.dist
        if self.comm:
            self.comm.present(self.film, self.image.data_ptr() if self.rank == 0 else 0, root=0)
            return self.image
        if not self.emulate:   # fallback: same pack / unpack kernels, device buffers through the process group (RCCL)
            pt.film_pack_tiles(self.film, self.rank, self.world, self.packed.data_ptr())
            torch.cuda.synchronize()
            if self.rank == 0:
                bufs = [self.packed] + [torch.zeros((max(self.counts[r], 1), 64, 3), dtype=torch.float32, device=self.packed.device)
                                        for r in range(1, self.world)]
                ops = [dist.P2POp(dist.irecv, bufs[r], r) for r in range(1, self.world) if self.counts[r]]
                for w in (dist.batch_isend_irecv(ops) if ops else []):
                    w.wait()
                torch.cuda.synchronize()
                for r in range(self.world):
                    pt.film_unpack_tiles(self.film, r, self.world, bufs[r].data_ptr(), self.image.data_ptr())
                return self.image
            if self.counts[self.rank]:
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, self.packed, 0)]):
                    w.wait()
            return None
        pt.film_pack_tiles(self.film, self.rank, self.world, self.packed.data_ptr())
        host = self.packed.cpu()
        if self.rank == 0:
            for r in range(self.world):
                if r == 0:
                    buf = host
                else:
                    buf = torch.zeros((max(self.counts[r], 1), 64, 3), dtype=torch.float32)
                    if self.counts[r]:
                        dist.recv(buf, src=r)
                d = buf.to(self.image.device)
                pt.film_unpack_tiles(self.film, r, self.world, d.data_ptr(), self.image.data_ptr())
            return self.image
        if self.counts[self.rank]:
            dist.send(host, dst=0)
        return None

    def selftest(self, film_tensor):
        """Before any timing: present a rank-coloured film through the very collective the run will use and check on the root that every tile
        arrived from the rank that owns it.  `film_tensor` (this rank's accumulation film, torch-owned) is overwritten and left cleared.
        -> the check's record on rank 0 (with how many ranks the communicator connected), None elsewhere."""
        torch = self.torch
        h, w = film_tensor.shape[:2]
        film_tensor.copy_(torch.from_numpy(selftest_film(w, h, self.rank, self.world)).to(film_tensor.device))
        torch.cuda.synchronize()
        img = self.present()
        torch.cuda.synchronize()
        rec = None
        if self.rank == 0:
            rec = selftest_check(img.cpu().numpy(), self.world)
            rec["rccl_ranks"] = self.ranks_seen
            rec["ok"] = bool(rec["ok"] and self.ranks_seen == self.world and rec["ranks_seen"] == list(range(self.world)))
            rec["path"] = self.describe()
        film_tensor.zero_()
        torch.cuda.synchronize()
        return rec

    def close(self):
        if self.comm:
            self.comm.close()
            self.comm = None


def sum_counters(values, device, group=None):
    """All-reduce a few exact integer counters (ray counts) as int64."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(list(values), dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return [int(x) for x in t.tolist()]
