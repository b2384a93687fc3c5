"""N > 1 path on CPU: world_size-2 gloo ranks render their stripe sets, gather to rank 0 and
de-interleave (the same prosper_amd.tiling code bench.py uses on RCCL).  The renderer here is the
oracle because this box has no GPU; the GPU tile render itself is covered by the -m gpu tests."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, world_size, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from oracle import binding as oracle
    from prosper_amd import scenes, structs as S, tiling
    world = scenes.cornell(with_skybox=True)
    w, h = 128, 40
    c = world.camera
    cam, focal = oracle.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], w, h)
    osc = oracle.OracleScene(world, brute_force=True)
    tile = tiling.tile_for_rank(rank, world_size)
    img = None
    for frame in (1, 2):
        flags = S.PC_FLAG_ACCUMULATE | S.PC_FLAG_CLAMP_INDIRECT | S.PC_FLAG_IBL | (S.PC_FLAG_SKIP_HISTORY if frame == 1 else 0)
        pc = S.ReferencePC(0, flags, frame, 1e-5, 1.0, focal, 3, 3)
        img, _ = osc.render(pc, cam, w, h, history=img, tile=tile, threads=2)
    assert img.shape == (h, tiling.local_width(w, rank, world_size), 4)
    mine = torch.from_numpy(img)
    gathered = [torch.empty_like(mine) for _ in range(world_size)] if rank == 0 else None
    dist.gather(mine, gathered, dst=0)
    # max-over-ranks timing pattern of bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world_size
    if rank == 0:
        full = tiling.deinterleave(gathered, w)
        np.save(out_path, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world_size", [2])
def test_stripe_partition_gather_equals_single_rank(tmp_path, oracle, world_size):
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_rank_main, args=(world_size, _free_port(), out), nprocs=world_size, join=True)
    got = np.load(out)
    from prosper_amd import scenes, structs as S
    world = scenes.cornell(with_skybox=True)
    w, h = 128, 40
    c = world.camera
    cam, focal = oracle.camera_uniforms(c["eye"], c["target"], c["up"], c["fov"], c["zN"], c["zF"], w, h)
    osc = oracle.OracleScene(world, brute_force=True)
    img = None
    for frame in (1, 2):
        flags = S.PC_FLAG_ACCUMULATE | S.PC_FLAG_CLAMP_INDIRECT | S.PC_FLAG_IBL | (S.PC_FLAG_SKIP_HISTORY if frame == 1 else 0)
        pc = S.ReferencePC(0, flags, frame, 1e-5, 1.0, focal, 3, 3)
        img, _ = osc.render(pc, cam, w, h, history=img)
    assert (got.view(np.uint32) == img.view(np.uint32)).all()
