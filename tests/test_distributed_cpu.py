"""world_size-2 test of the multi-GPU path on CPU (gloo): tile ownership is a partition, and the presentation
collective -- every rank packs the tiles it owns, the root gathers and unpacks them (the host mirror of
pt_film_present's kernels, csrc/present_rccl.hip) -- reproduces the single-device film bit for bit without touching
any rank's accumulation film."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tile_owner_is_a_partition(pt):
    d = importlib.import_module("single-file-vulkan-pathtracing_amd.distributed")
    for (w, h, world) in [(1920, 1080, 8), (256, 256, 2), (250, 131, 3), (7, 5, 4)]:
        own = d.tile_owner(w, h, world)
        assert own.shape == (h, w) and own.min() >= 0 and own.max() < world
        total = sum(d.owned_mask(w, h, r, world).astype(np.int64) for r in range(world))
        assert (total == 1).all()
    # interleaving balances the border (cheap) and the interior (expensive) over the ranks
    own = d.tile_owner(1920, 1080, 8)
    counts = np.bincount(own.ravel(), minlength=8)
    assert counts.max() - counts.min() <= 64 * 135


def _worker(rank, world, port, w, h, out):
    sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = importlib.import_module("single-file-vulkan-pathtracing_amd.distributed")
    full = np.load(out + ".full.npy")
    # what a rank's film holds: its own tiles, exact zeros elsewhere
    mine = np.where(d.owned_mask(w, h, rank, world)[..., None], full, np.float32(0)).astype(np.float32)
    keep = mine.copy()
    first = d.gather_present_host(mine, rank, world, dst=0)
    # the collective leaves the rank's accumulation film alone, so presenting twice gives the same image
    assert mine.tobytes() == keep.tobytes()
    again = d.gather_present_host(mine, rank, world, dst=0)
    rays = d.sum_counters([1000 + rank, 7], "cpu")
    if rank == 0:
        assert first.tobytes() == again.tobytes()
        np.save(out + ".reduced.npy", again)
        np.save(out + ".rays.npy", np.array(rays))
    else:
        assert first is None and again is None
    dist.barrier()
    dist.destroy_process_group()


def test_pack_unpack_round_trip_on_ragged_sizes(pt):
    d = importlib.import_module("single-file-vulkan-pathtracing_amd.distributed")
    rng = np.random.default_rng(3)
    for (w, h, world) in [(64, 40, 2), (61, 37, 3), (7, 5, 4), (200, 9, 8)]:
        full = rng.uniform(0, 4, (h, w, 3)).astype(np.float32)
        image = np.zeros_like(full)
        n = 0
        for r in range(world):
            packed = d.pack_tiles_host(full, r, world)
            n += len(packed)
            d.unpack_tiles_host(packed, image, r, world)
        assert n == ((w + 7) // 8) * ((h + 7) // 8) and image.tobytes() == full.tobytes()


@pytest.mark.parametrize("w,h", [(64, 40), (61, 37)])
def test_gloo_world2_gather_present_is_bit_exact(orc, cornell_oracle, tmp_path, w, h):
    p = orc.default_params(width=w, height=h, spp_per_frame=2, max_depth=4)
    full, _, _, _ = cornell_oracle.render_frame(p)
    out = str(tmp_path / "x")
    np.save(out + ".full.npy", full)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, w, h, out), nprocs=2, join=True)
    red = np.load(out + ".reduced.npy")
    assert red.tobytes() == full.tobytes()
    assert list(np.load(out + ".rays.npy")) == [2001, 14]


def _selftest_worker(rank, world, port, w, h, out):
    sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = importlib.import_module("single-file-vulkan-pathtracing_amd.distributed")
    img = d.gather_present_host(d.selftest_film(w, h, rank, world), rank, world, dst=0)
    if rank == 0:
        rec = d.selftest_check(img, world)
        # ... and the check does find a tile that came from the wrong rank
        bad = img.copy()
        bad[0:8, 8:16] = bad[0:8, 0:8]
        rec["catches_a_swapped_tile"] = not d.selftest_check(bad, world)["ok"]
        import json
        json.dump(rec, open(out, "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_presentation_selftest_over_gloo(tmp_path, world):
    """`bench.py --gpus N --selftest` / `pt_main --ranks N --selftest` present a rank-coloured film before timing and check on the root that every
    tile carries its owner's colour.  The same film, gather and check here over gloo with N = 2, 4, 8 processes (host mirrors of the pack /
    unpack kernels): all ranks seen, no wrong pixel, per-rank tile counts that add up; and a swapped tile IS caught."""
    import json
    w, h = 200, 120
    out = str(tmp_path / "rec.json")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_selftest_worker, args=(world, port, w, h, out), nprocs=world, join=True)
    rec = json.load(open(out))
    assert rec["ok"] and rec["wrong_pixels"] == 0 and rec["ranks_seen"] == list(range(world)), rec
    assert sum(rec["tiles_per_rank"]) == ((w + 7) // 8) * ((h + 7) // 8) and rec["catches_a_swapped_tile"]
