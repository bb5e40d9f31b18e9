"""Frame sharding over GPUs, one process per GPU (torch.distributed; "nccl" is RCCL on ROCm).

The reference renders rows as independent tasks (par_chunks_mut, raytracer.rs:57-59).  Here a
frame is cut into tiles of TILE_ROWS rows; tile t belongs to rank t % world (interleaved, so
the expensive middle of the image is spread over all ranks).  The scene blob is built once on
rank 0 and replicated with one broadcast; every frame ends with one gather of the compact
per-rank row buffers to rank 0, which scatters them back into frame order.  No other
collective is involved.

The pure index arithmetic lives in functions that do not need a GPU so that it can be tested
with the gloo backend on CPU.
"""
import numpy as np

from . import TILE_ROWS


def shard_row_indices(height, shard, nshards, tile_rows=TILE_ROWS):
    """Rows of the frame owned by `shard`, in the order they sit in its compact buffer."""
    y = np.arange(height)
    return y[(y // tile_rows) % nshards == shard]


def max_shard_rows(height, nshards, tile_rows=TILE_ROWS):
    return max(len(shard_row_indices(height, s, nshards, tile_rows)) for s in range(nshards))


def assemble_rows_host(parts, height, tile_rows=TILE_ROWS):
    """numpy statement of what rt_assemble_tiles does on the GPU: parts[s] = compact rows of shard s
    (possibly padded at the end) -> frame[height, W]."""
    nshards = len(parts)
    width = parts[0].shape[1]
    frame = np.zeros((height, width), dtype=parts[0].dtype)
    for s, p in enumerate(parts):
        rows = shard_row_indices(height, s, nshards, tile_rows)
        frame[rows] = p[:len(rows)]
    return frame


def broadcast_bytes(buf, src=0, group=None):
    """Broadcast a 1-D uint8 tensor whose length only `src` knows.  Returns the received tensor."""
    import torch
    import torch.distributed as dist
    dev = buf.device if buf is not None else None
    n = torch.tensor([0 if buf is None else buf.numel()], dtype=torch.int64)
    if dist.get_backend(group) == "nccl":
        n = n.cuda()
    dist.broadcast(n, src=src, group=group)
    size = int(n.item())
    if buf is None:
        buf = torch.empty(size, dtype=torch.uint8, device=n.device if dev is None else dev)
    dist.broadcast(buf, src=src, group=group)
    return buf


class ShardedRenderer:
    """One rank of a multi-GPU render.  Rank 0 owns the scene description; every rank renders
    its tiles; rank 0 ends up with the frame."""

    def __init__(self, rt, device, scene=None, tile_rows=TILE_ROWS):
        import torch
        import torch.distributed as dist
        self.rt, self.torch, self.dist = rt, torch, dist
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.tile_rows = tile_rows
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        rt.init(device)
        self.ext_stream = torch.cuda.ExternalStream(rt.stream(), device=self.device)
        self.bvh_info = None
        self._blob = None
        if self.rank == 0:
            assert scene is not None, "rank 0 must be given the scene"
            self.bvh_info = rt.upload_scene(scene)              # RayTracer::new on the owning rank
        if self.world > 1:
            if self.rank == 0:
                _, nbytes = rt.scene_blob()
                buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
                rt.copy_scene_blob(buf.data_ptr(), nbytes)
            else:
                buf = None
            buf = self._bcast(buf)
            if self.rank != 0:
                torch.cuda.synchronize(self.device)
                rt.adopt_scene_blob(buf.data_ptr(), buf.numel())
                self._blob = buf                                 # keeps the adopted memory alive
        self._bufs = {}

    def _bcast(self, buf):
        torch, dist = self.torch, self.dist
        n = torch.tensor([0 if buf is None else buf.numel()], dtype=torch.int64, device=self.device)
        dist.broadcast(n, src=0)
        if buf is None:
            buf = torch.empty(int(n.item()), dtype=torch.uint8, device=self.device)
        dist.broadcast(buf, src=0)
        return buf

    def _buffers(self, width, height):
        key = (width, height)
        if key not in self._bufs:
            torch = self.torch
            stride = max_shard_rows(height, self.world, self.tile_rows)
            local = torch.zeros((stride, width), dtype=torch.int32, device=self.device)
            gathered = frame = None
            if self.rank == 0:
                frame = torch.zeros((height, width), dtype=torch.int32, device=self.device)
                if self.world > 1:
                    gathered = torch.zeros((self.world, stride, width), dtype=torch.int32, device=self.device)
            self._bufs[key] = (stride, local, gathered, frame)
        return self._bufs[key]

    def render_frame(self, frame_desc):
        """Enqueues one frame on the library's stream.  On rank 0 returns the device tensor
        [H, W] (int32 view of 0xFFRRGGBB) that will hold it once the stream has drained."""
        rt, torch, dist = self.rt, self.torch, self.dist
        W, H = frame_desc.width, frame_desc.height
        stride, local, gathered, frame = self._buffers(W, H)
        with torch.cuda.stream(self.ext_stream):
            if self.world == 1:
                rt.render_tiles(frame_desc, 0, 1, frame.data_ptr(), tile_rows=self.tile_rows)
                return frame
            rt.render_tiles(frame_desc, self.rank, self.world, local.data_ptr(), tile_rows=self.tile_rows)
            dist.gather(local, list(gathered.unbind(0)) if self.rank == 0 else None, dst=0)
            if self.rank == 0:
                rt.assemble_tiles(gathered.data_ptr(), stride, self.world, W, H, frame.data_ptr(), tile_rows=self.tile_rows)
        return frame

    def frame_to_host(self, frame):
        self.rt.synchronize()
        self.torch.cuda.synchronize(self.device)
        return frame.cpu().numpy().view(np.uint32)
