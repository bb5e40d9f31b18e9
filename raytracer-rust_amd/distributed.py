"""Frame sharding over GPUs, one process per GPU (torch.distributed; "nccl" is RCCL on ROCm).

The reference renders rows as independent tasks (par_chunks_mut, raytracer.rs:57-59).  Here a
frame is cut into tiles of TILE_ROWS rows; tile t belongs to rank t % world (interleaved, so
the expensive middle of the image is spread over all ranks).  The scene blob is built once on
rank 0 and replicated with one broadcast; every frame ends with one gather of the compact
per-rank row buffers to rank 0, which scatters them back into frame order.  No other
collective is involved.

Frames are pipelined over the library's lanes (a stream with its own work buffers each), two
deep up to four ranks and four deep beyond: frame i+1 is enqueued on the next lane, so its kernel
fills the GPU while the kernel of frame i drains and the gather of frame i is in flight; rank 0
restores frame order of a frame when its lane comes round again (or at flush()).  The smaller a
rank's share of the frame, the more of its launch is the tail of its slowest wavefronts, and the
more frames in flight it takes to fill the GPU.

The index arithmetic and the pipelining logic do not need a GPU: ShardedRenderer takes the two
device operations (render my tiles, restore frame order) as callables, so the same class runs
under the gloo backend on CPU tensors in the tests.
"""
import numpy as np

from . import TILE_ROWS

MAX_LANES = 4                      # lanes of the library (RT_LANES)


def default_lanes(world):
    """Frames in flight per GPU when the frame is cut over `world` ranks (measured on the 4K dragon frame: a
    quarter of the frame fills the GPU with two launches in flight, an eighth needs four)."""
    return 2 if world <= 4 else 4


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


def broadcast_bytes(buf, device, src=0, group=None):
    """Broadcast a 1-D uint8 tensor whose length only `src` knows.  Returns the received tensor."""
    import torch
    import torch.distributed as dist
    n = torch.tensor([0 if buf is None else buf.numel()], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src, group=group)
    if buf is None:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=src, group=group)
    return buf


class ShardedRenderer:
    """One rank of a multi-GPU render.  Every rank renders its tiles; rank 0 ends up with the frame.

    render_rows(frame_desc, rank, world, tile_rows, out[stride, W])   enqueue this rank's tiles
    assemble(gathered[world, stride, W], frame_out[H, W], tile_rows)  restore frame order (rank 0)
    select_lane(lane)                                                 make `lane` the one the two operations above use
    stream_ctx(lane)                                                  context manager entered around everything enqueued for `lane`

    Consecutive frames cycle through `lanes` lanes, each with its own stream and buffers, so the work of
    frame i+1 does not wait for frame i: its kernel fills the GPU while the kernel of frame i drains, and the
    gather of frame i overlaps it.  The tensor render_frame returns holds its frame once flush() has been called
    (or `lanes` more frames have been enqueued) and the device has been synchronised.
    """

    def __init__(self, device, render_rows, assemble, stream_ctx=None, tile_rows=TILE_ROWS, always_gather=False, select_lane=None, lanes=None,
                 group=None):
        import contextlib
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group              # process group of the collectives (None: the default group)
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.device = device
        self.tile_rows = tile_rows
        self._render_rows, self._assemble = render_rows, assemble
        self._stream_ctx = stream_ctx or (lambda lane: contextlib.nullcontext())
        self._select_lane = select_lane or (lambda lane: None)
        self._bufs = {}
        self._gather_lists = {}         # per frame size and lane: the per-rank views of the gather buffer, made once
        self._pending = [None] * MAX_LANES    # per lane: (work, W, H) of the frame whose gather is in flight
        self._slot = 0
        # A single rank normally renders straight into the frame; with always_gather it goes through
        # the same gather and reassembly as many ranks do (to rehearse that path on one GPU).
        self._collectives = self.world > 1 or (always_gather and dist.is_initialized())
        # A single rank without collectives keeps to one lane unless asked otherwise: its launches then run one
        # after the other, and a launch's duration is the time of a frame.
        self.lanes = lanes if lanes is not None else (default_lanes(self.world) if self._collectives else 1)
        assert 1 <= self.lanes <= MAX_LANES

    def buffers(self, width, height):
        """(stride, local[lanes], gathered[lanes], frame[lanes]): one set per lane."""
        key = (width, height)
        if key not in self._bufs:
            torch = self.torch
            stride = max_shard_rows(height, self.world, self.tile_rows)
            L = self.lanes
            local = [torch.zeros((stride, width), dtype=torch.int32, device=self.device) for _ in range(L)]
            gathered, frame = [None] * L, [None] * L
            if self.rank == 0:
                frame = [torch.zeros((height, width), dtype=torch.int32, device=self.device) for _ in range(L)]
                if self._collectives:
                    gathered = [torch.zeros((self.world, stride, width), dtype=torch.int32, device=self.device) for _ in range(L)]
            if torch.device(self.device).type == "cuda":
                torch.cuda.synchronize(self.device)       # the fills run on torch's stream; the library's streams do not wait for it
            self._bufs[key] = (stride, local, gathered, frame)
        return self._bufs[key]

    def _complete(self, lane):
        """Waits (on the lane's stream, not the host) for the gather in flight on `lane` and restores frame order on rank 0.
        Must be called with the lane selected and its stream context entered."""
        if self._pending[lane] is None:
            return
        work, W, H = self._pending[lane]
        self._pending[lane] = None
        stride, local, gathered, frame = self.buffers(W, H)
        work.wait()
        if self.rank == 0:
            self._assemble(gathered[lane], frame[lane], self.tile_rows)

    def next_lane(self):
        return self._slot

    def render_frame(self, frame_desc):
        """Enqueues one frame on the next lane.  On rank 0 returns the tensor [H, W] (int32 view of 0xFFRRGGBB) of that frame."""
        W, H = frame_desc.width, frame_desc.height
        stride, local, gathered, frame = self.buffers(W, H)
        lane = self._slot
        self._slot = (self._slot + 1) % self.lanes
        self._select_lane(lane)
        with self._stream_ctx(lane):
            if not self._collectives:
                self._render_rows(frame_desc, 0, 1, self.tile_rows, frame[lane])
                return frame[lane]
            # the frame rendered on this lane `lanes` frames ago: its gather has had that many kernels' time to finish
            self._complete(lane)
            self._render_rows(frame_desc, self.rank, self.world, self.tile_rows, local[lane])
            views = None
            if self.rank == 0:
                views = self._gather_lists.get((W, H, lane))
                if views is None:
                    views = self._gather_lists[(W, H, lane)] = list(gathered[lane].unbind(0))
            work = self.dist.gather(local[lane], views, dst=0, group=self.group, async_op=True)
            self._pending[lane] = (work, W, H)
        return frame[lane]

    def flush(self):
        for lane in [(self._slot + k) % self.lanes for k in range(self.lanes)]:          # oldest first
            self._select_lane(lane)
            with self._stream_ctx(lane):
                self._complete(lane)
        self._select_lane(self._slot)


def gpu_renderer(rt, device_index, scene=None, tile_rows=TILE_ROWS, always_gather=False, gpu_build=False, lanes=None, group=None):
    """ShardedRenderer on one MI355X: uploads (rank 0) or receives (other ranks) the scene and wires the
    library's tile rendering and frame assembly in.  Returns (renderer, bvh_info or None)."""
    import torch
    import torch.distributed as dist
    device = torch.device("cuda", device_index)
    torch.cuda.set_device(device)
    rt.init(device_index)
    ext = []                                                           # torch work is ordered on the library's streams, one per lane
    for lane in range(MAX_LANES):
        rt.select_lane(lane)
        ext.append(torch.cuda.ExternalStream(rt.stream(), device=device))
    rt.select_lane(0)
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    info, keep = None, None
    if rank == 0:
        assert scene is not None, "rank 0 must be given the scene"
        info = rt.upload_scene(scene, gpu_build=gpu_build)             # RayTracer::new on the owning rank
    if world > 1 or (always_gather and dist.is_initialized()):
        buf = None
        if rank == 0:
            _, nbytes = rt.scene_blob()
            buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            rt.copy_scene_blob(buf.data_ptr(), nbytes)
        buf = broadcast_bytes(buf, device, src=0, group=group)
        if rank != 0:
            torch.cuda.synchronize(device)
            rt.adopt_scene_blob(buf.data_ptr(), buf.numel())
            keep = buf                                                 # the adopted memory must outlive the renderer

    def render_rows(frame_desc, shard, nshards, rows, out):
        rt.render_tiles(frame_desc, shard, nshards, out.data_ptr(), tile_rows=rows)

    def assemble(gathered, frame, rows):
        rt.assemble_tiles(gathered.data_ptr(), gathered.shape[1], gathered.shape[0], frame.shape[1], frame.shape[0], frame.data_ptr(), tile_rows=rows)

    r = ShardedRenderer(device, render_rows, assemble, stream_ctx=lambda lane: torch.cuda.stream(ext[lane]), tile_rows=tile_rows,
                        always_gather=always_gather, select_lane=rt.select_lane, lanes=lanes, group=group)
    r.ext_streams, r._keep_blob = ext, keep
    return r, info
