"""The N > 1 path on CPU: world_size-2 (and 3) gloo process groups exercise what the multi-GPU
benchmark does around the kernel -- replicate the scene blob with a broadcast, let every rank
produce the rows of its interleaved tiles, gather the compact buffers on rank 0 and restore
frame order.  Without a GPU the rows themselves come from the CPU oracle (test infrastructure),
so the result can be compared with the oracle's render of the whole frame.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, width, height, tile_rows, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import importlib
    import __graft_entry__ as ge
    import oracle as O
    rt = ge.load_package()
    shard = importlib.import_module("raytracer_rust_amd.distributed")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        scene_path = os.path.join(GOLDEN, "test_file", "jalon6", "tp64.test")
        # rank 0 owns the scene: parse, build the blob on the host, broadcast it
        blob = None
        if rank == 0:
            cfg = rt.ParsedConfigState().load_config_file(scene_path)
            host_blob, _ = rt.pack_scene_host(cfg.scene())
            blob = torch.from_numpy(host_blob.copy())
        blob = shard.broadcast_bytes(blob, src=0)
        # every rank checks that it received a well-formed blob of the announced size
        magic, version, total = np.frombuffer(blob.numpy()[:16].tobytes(), dtype=np.uint32)[0], \
            np.frombuffer(blob.numpy()[:16].tobytes(), dtype=np.uint32)[1], np.frombuffer(blob.numpy()[8:16].tobytes(), dtype=np.uint64)[0]
        assert magic == 0x52544231 and version == 2 and int(total) == blob.numel()

        rows = shard.shard_row_indices(height, rank, world, tile_rows)
        assert len(rows) == rt.shard_rows(height, rank, world, tile_rows)
        sc = O.OracleScene.load(scene_path)
        mine, _ = sc.render_tiles(rank, world, tile_rows, width=width, height=height, threads=2)
        stride = shard.max_shard_rows(height, world, tile_rows)
        local = torch.zeros((stride, width), dtype=torch.int32)
        local[:len(rows)] = torch.from_numpy(mine.view(np.int32))
        gathered = [torch.zeros_like(local) for _ in range(world)] if rank == 0 else None
        dist.gather(local, gathered, dst=0)
        if rank == 0:
            frame = shard.assemble_rows_host([g.numpy().view(np.uint32) for g in gathered], height, tile_rows)
            want, _ = sc.render(width=width, height=height, threads=2)
            np.save(out_path, np.array([int(np.array_equal(frame, want))]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,width,height,tile_rows", [(2, 96, 70, 8), (3, 64, 50, 8), (2, 40, 9, 16)])
def test_sharded_frame_over_gloo(tmp_path, world, width, height, tile_rows):
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(world, _free_port(), width, height, tile_rows, out), nprocs=world, join=True)
    assert np.load(out)[0] == 1


def test_assemble_is_the_inverse_of_sharding():
    import importlib
    import __graft_entry__ as ge
    ge.load_package()
    shard = importlib.import_module("raytracer_rust_amd.distributed")
    rng = np.random.default_rng(0)
    for height, world, tile in [(2160, 8, 8), (2161, 8, 8), (5, 8, 8), (100, 3, 16)]:
        frame = rng.integers(0, 2 ** 32, size=(height, 7), dtype=np.uint32)
        stride = shard.max_shard_rows(height, world, tile)
        parts = []
        for s in range(world):
            rows = shard.shard_row_indices(height, s, world, tile)
            p = np.zeros((stride, 7), np.uint32)
            p[:len(rows)] = frame[rows]
            parts.append(p)
        assert np.array_equal(shard.assemble_rows_host(parts, height, tile), frame)
        # tiles of consecutive rows stay together and rotate over the shards
        assert all(((shard.shard_row_indices(height, s, world, tile) // tile) % world == s).all() for s in range(world))
