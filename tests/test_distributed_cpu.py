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


def _worker(rank, world, port, width, height, tile_rows, lanes, out_path):
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
        blob = shard.broadcast_bytes(blob, torch.device("cpu"), src=0)
        # every rank checks that it received a well-formed blob of the announced size
        magic, version, total = np.frombuffer(blob.numpy()[:16].tobytes(), dtype=np.uint32)[0], \
            np.frombuffer(blob.numpy()[:16].tobytes(), dtype=np.uint32)[1], np.frombuffer(blob.numpy()[8:16].tobytes(), dtype=np.uint64)[0]
        assert magic == 0x52544231 and version == 7 and int(total) == blob.numel()

        rows = shard.shard_row_indices(height, rank, world, tile_rows)
        assert len(rows) == rt.shard_rows(height, rank, world, tile_rows)
        sc = O.OracleScene.load(scene_path)

        # The renderer class the GPU benchmark uses, with the two device operations replaced by CPU
        # stand-ins (rows from the oracle, numpy reassembly): pipelining, buffers and collectives are the same code.
        def render_rows(frame_desc, shard_id, nshards, trows, out):
            px, _ = sc.render_tiles(shard_id, nshards, trows, width=frame_desc.width, height=frame_desc.height, threads=2)
            out.zero_()
            out[:px.shape[0]] = torch.from_numpy(px.view(np.int32))

        def assemble(gathered, frame, trows):
            parts = [g.numpy().view(np.uint32) for g in gathered.unbind(0)]
            frame.copy_(torch.from_numpy(shard.assemble_rows_host(parts, frame.shape[0], trows).view(np.int32)))

        class FrameDesc:
            pass

        renderer = shard.ShardedRenderer(torch.device("cpu"), render_rows, assemble, tile_rows=tile_rows, lanes=lanes)
        lanes = renderer.lanes
        outs = []
        sizes = [(width, height)] * (lanes + 1) + [(width // 2, height)]       # more frames than lanes, and another size
        for w, h in sizes:
            fd = FrameDesc()
            fd.width, fd.height = w, h
            outs.append((w, h, renderer.render_frame(fd)))
        renderer.flush()
        if rank == 0:
            ok = True
            for w, h, out in outs[1:]:          # the first frame's tensor was used again when its lane came round
                want, _ = sc.render(width=w, height=h, threads=2)
                ok = ok and np.array_equal(out.numpy().view(np.uint32), want)
            ok = ok and outs[0][2] is outs[lanes][2] and all(outs[0][2] is not outs[k][2] for k in range(1, lanes))
            np.save(out_path, np.array([int(ok)]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,width,height,tile_rows,lanes", [(2, 96, 70, 8, None), (3, 64, 50, 8, None), (2, 40, 9, 16, None), (2, 64, 40, 8, 4), (2, 64, 40, 8, 1)])
def test_sharded_frame_over_gloo(tmp_path, world, width, height, tile_rows, lanes):
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(world, _free_port(), width, height, tile_rows, lanes, out), nprocs=world, join=True)
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


def test_shard_rows_of_the_library_match_the_index_arithmetic():
    import importlib
    import __graft_entry__ as ge
    rt = ge.load_package()
    D = importlib.import_module("raytracer_rust_amd.distributed")
    """rt_shard_rows (pure arithmetic, no GPU) is what the library's own multi-GPU frames and the per-process path size their
    buffers with: every row belongs to exactly one shard, shard 0 never owns fewer rows than another (the stride of the gather
    buffer), and the count agrees with the row lists the reassembly is tested with."""
    L = rt.lib()
    for height in (1, 7, 8, 9, 61, 480, 2160, 4320):
        for n in (1, 2, 3, 4, 8):
            for tile_rows in (1, 7, 8, 16):
                rows = [int(L.rt_shard_rows(height, s, n, tile_rows)) for s in range(n)]
                assert sum(rows) == height
                assert rows[0] == max(rows)
                for s in range(n):
                    assert rows[s] == len(D.shard_row_indices(height, s, n, tile_rows))
    assert L.rt_shard_rows(100, 3, 3, 8) == 0 and L.rt_shard_rows(100, 0, 0, 8) == 0 and L.rt_shard_rows(100, 0, 2, 0) == 0
