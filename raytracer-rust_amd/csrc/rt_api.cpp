// rt_api.cpp -- implementation of the C ABI declared in include/rt_amd.h.
//
// One process drives one GPU (rt_init) or several GPUs of one node (rt_init_multi).  The scene lives in one device
// blob (layout.h) per GPU between an upload and rt_shutdown; with several GPUs it is built on the first and replicated
// with one RCCL broadcast.  Work buffers (frame, per-depth fold records, counters) grow on demand and are reused across
// calls, one set per GPU and lane (a lane is a stream with its own buffers: launches on different lanes may overlap on
// the GPU).  A frame on several GPUs is cut into interleaved 8-row tiles, tile t to GPU t % n; every GPU renders its
// tiles into a compact buffer, one grouped RCCL send/receive gathers them on the first GPU, and a kernel there restores
// frame order.  One host thread enqueues everything: nothing between the kernels, the gather and the reassembly of a
// frame waits for the host.  There is no CPU fallback: every entry point that needs the GPU fails with an error when
// HIP does.
#include "../../include/rt_amd.h"

#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bvh_gpu.h"
#include "device_types.h"
#include "errors.h"
#include "host_copy.h"
#include "layout.h"
#include "scene_pack.h"

namespace {

struct RegionKey {                                           // what the regions of a launch depend on
    uint32_t width, height, shard, nshards, tile_rows; uint64_t scene;
    bool operator==(const RegionKey& o) const {
        return width == o.width && height == o.height && shard == o.shard && nshards == o.nshards && tile_rows == o.tile_rows && scene == o.scene;
    }
    bool operator!=(const RegionKey& o) const { return !(*this == o); }
};

// A lane is a stream with the work buffers of the launches made on it.  Launches on one lane run in order;
// launches on different lanes share nothing but the (read-only) scene and may overlap on the GPU, which is how
// a caller that renders frame after frame hides the ramp-up and the drain of every launch (rt_select_lane).
struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float4* d_fold = nullptr; size_t fold_cap = 0;          // records
    unsigned long long* d_counters = nullptr;               // nodes, prims, shadow, secondary, wave-level passes
    // order in which the tile regions of a launch are handed out, and what it was computed for
    // cost: what the next launch measures into, all zero between launches; spare: the other array of that kind (the ordering
    // kernel reads one and clears the other, then they change places); last: the costs of the launch before.  One allocation.
    uint32_t* d_region_order = nullptr; uint32_t* d_region_cost = nullptr; uint32_t* d_region_spare = nullptr; uint32_t* d_region_last = nullptr;
    uint32_t* d_region_block = nullptr; uint32_t region_cap = 0;
    uint32_t region_n = 0;                                  // the first region_n words of cost are known to be zero, nothing else is written (0: cost and spare need clearing)
    RegionKey region_key{}; bool region_key_valid = false;
    // frames over several GPUs: this GPU's compact rows; on the first GPU the gathered rows of all and the frame
    uint32_t* d_rows = nullptr; size_t rows_cap = 0;
    uint32_t* d_gathered = nullptr; size_t gathered_cap = 0;
    uint32_t* d_frame = nullptr; size_t frame_cap = 0;
    hipEvent_t rows_ready = nullptr, gathered = nullptr;    // kernel of this lane done (for the gather) / this GPU's part in the gather done
};
constexpr int N_LANES = RT_LANES;

struct Device {
    int device = -1;
    Lane lanes[N_LANES];
    void* d_blob = nullptr; size_t blob_bytes = 0; bool blob_owned = false;
    DevScene dsc{};
    // Every RCCL operation of this GPU's communicator is enqueued on this one stream, in the order the host issues them,
    // whatever lane the frame belongs to: operations of one communicator then never meet on different streams, and the
    // lanes' kernels and the reassembly are tied to it with events.
    hipStream_t comm_stream = nullptr;
};

struct State {
    bool inited = false;
    int ndev = 0;
    Device devs[RT_MAX_GPUS];
    bool have_comm = false;                                 // rt_init_multi: frames go through the sharded path, scenes are replicated
    ncclComm_t comms[RT_MAX_GPUS] = {};
    int lane = 0;                                           // the lane the next call uses
    bool have_scene = false;
    rtl::BlobHeader hdr{};
    rt_bvh_info info{};
    uint32_t* d_out = nullptr; size_t out_cap = 0;          // pixels (rt_render's frame before it is copied to the host)
    // rt_render's way to the caller's (pageable) memory: pinned staging buffer, filled piece by piece by the copy
    // engine while the pieces that have arrived are copied on by the host's threads
    uint32_t* h_stage = nullptr; size_t stage_cap = 0;
    hipEvent_t piece_done[8] = {};
    bool generic_kernel = false;                            // RT_RENDER_KERNEL=generic: every scene through the kernel that dispatches on the kind of shape (for comparisons)
    bool first_order_raster = false;                        // RT_FIRST_ORDER=raster: the first launch of a frame geometry hands its regions out in index order (for comparisons)
    bool stage_by_copy = false;                             // RT_RENDER_STAGING=copy: rt_render on one GPU copies the finished frame instead of rendering into host memory
    uint64_t scene_generation = 0;                          // bumped whenever another scene becomes resident
};
State g;
Device& first() { return g.devs[0]; }
Lane& cur() { return g.devs[0].lanes[g.lane]; }
Lane& cur(int d) { return g.devs[d].lanes[g.lane]; }

int fail(const std::string& m) { rte::set_error(m); return 1; }
int fail_hip(const char* what, hipError_t e) { rte::set_error(std::string(what) + ": " + hipGetErrorString(e)); return 1; }
int fail_nccl(const char* what, ncclResult_t e) { rte::set_error(std::string(what) + ": " + ncclGetErrorString(e)); return 1; }

#define HIP_TRY(expr)                                                  \
    do {                                                               \
        hipError_t e_ = (expr);                                        \
        if (e_ != hipSuccess) return fail_hip(#expr, e_);              \
    } while (0)
#define NCCL_TRY(expr)                                                 \
    do {                                                               \
        ncclResult_t e_ = (expr);                                      \
        if (e_ != ncclSuccess) return fail_nccl(#expr, e_);            \
    } while (0)

int use(int d) { HIP_TRY(hipSetDevice(g.devs[d].device)); return 0; }

// Scene changes wait for the launches of every lane of every GPU.
int sync_lanes() {
    for (int d = g.ndev - 1; d >= 0; d--) {                  // ends with the first GPU current
        if (use(d)) return 1;
        for (Lane& l : g.devs[d].lanes) if (l.stream) HIP_TRY(hipStreamSynchronize(l.stream));
        if (g.devs[d].comm_stream) HIP_TRY(hipStreamSynchronize(g.devs[d].comm_stream));
    }
    return 0;
}

int grow(uint32_t** p, size_t* cap, size_t pixels) {
    if (pixels <= *cap) return 0;
    if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
    HIP_TRY(hipMalloc((void**)p, pixels * sizeof(uint32_t)));
    *cap = pixels;
    return 0;
}

void release_scene() {
    for (int d = g.ndev - 1; d >= 0; d--) {
        Device& dev = g.devs[d];
        (void)hipSetDevice(dev.device);
        if (dev.d_blob && dev.blob_owned) (void)hipFree(dev.d_blob);
        dev.d_blob = nullptr; dev.blob_bytes = 0; dev.blob_owned = false;
    }
    g.have_scene = false;
}

void point_at_blob(Device& dev, const rtl::BlobHeader& h) {
    const char* base = (const char*)dev.d_blob;
    dev.dsc.nodes = (const uint4*)(base + h.off_nodes);
    dev.dsc.tree = (const float4*)(base + h.off_tree_shapes);
    dev.dsc.list = (const float4*)(base + h.off_list_shapes);
    dev.dsc.mats = (const float4*)(base + h.off_materials);
    dev.dsc.lights = (const float4*)(base + h.off_lights);
    dev.dsc.n_list = h.n_list_shapes; dev.dsc.n_lights = h.n_lights; dev.dsc.root_ref = h.root_ref; dev.dsc.finite_shading = h.finite_shading; dev.dsc.tree_kinds = g.generic_kernel ? 0u : h.tree_kinds;
    for (int k = 0; k < 3; k++) {
        dev.dsc.grid_origin[k] = h.grid_origin[k]; dev.dsc.grid_scale[k] = h.grid_scale[k];
        dev.dsc.inv_limit[k] = std::fmin(1e30f, 1e32f / std::fmax(h.grid_scale[k], 1e-30f));     // 131072 * scale / d stays finite
    }
}

// The blob that has just been put on the first GPU becomes the scene of every GPU the library drives: one broadcast.
int scene_is_resident(const rtl::BlobHeader& h) {
    g.hdr = h;
    point_at_blob(first(), h);
    if (g.have_comm) {
        const size_t bytes = first().blob_bytes;
        for (int d = 1; d < g.ndev; d++) {
            if (use(d)) return 1;
            HIP_TRY(hipMalloc(&g.devs[d].d_blob, bytes));
            g.devs[d].blob_owned = true; g.devs[d].blob_bytes = bytes;
        }
        NCCL_TRY(ncclGroupStart());                                                              // (the upload itself has been waited for)
        for (int d = 0; d < g.ndev; d++)
            NCCL_TRY(ncclBroadcast(g.devs[d].d_blob, g.devs[d].d_blob, bytes, ncclUint8, 0, g.comms[d], g.devs[d].comm_stream));     // in place on every GPU
        NCCL_TRY(ncclGroupEnd());
        for (int d = g.ndev - 1; d >= 0; d--) {
            if (use(d)) return 1;
            HIP_TRY(hipStreamSynchronize(g.devs[d].comm_stream));
            if (d > 0) point_at_blob(g.devs[d], h);
        }
    }
    g.info.node_pairs = h.n_pairs; g.info.leaves = h.n_leaves; g.info.max_depth = h.max_depth; g.info.max_leaf = h.max_leaf;
    g.info.bvh_prims = h.n_tree_shapes; g.info.list_prims = h.n_list_shapes; g.info.sah_cost = h.sah_cost;
    g.have_scene = true;
    g.scene_generation++;
    return 0;
}

int check_frame(const rt_frame* f) {
    if (!f) return fail("rt_frame is NULL");
    if (f->width == 0 || f->height == 0) return fail("Width and height must be greater than zero");
    if ((uint64_t)f->width * f->height > (1ull << 31)) return fail("frame larger than 2^31 pixels");
    return 0;
}

// Global array for the records of the reflection levels that do not fit in LDS.
int ensure_fold(Lane& lane, size_t npix, size_t levels) {
    const size_t limit = (200ull << 30) / sizeof(float4);
    if (levels != 0 && npix > limit / levels) return fail("maxdepth x pixels needs more than 200 GiB of fold records");
    const size_t records = npix * levels;
    if (records <= lane.fold_cap) return 0;
    if (lane.d_fold) { (void)hipFree(lane.d_fold); lane.d_fold = nullptr; lane.fold_cap = 0; }
    HIP_TRY(hipMalloc((void**)&lane.d_fold, records * sizeof(float4)));
    lane.fold_cap = records;
    return 0;
}

// Tile order of a launch.  Wavefront run times are heavy-tailed (on the 4K dragon frame: median 40 us, mean
// 140 us, the slowest 1.4 ms) and the slow ones sit where the tree's shapes are on screen; handed out in plain
// raster order the last of them start late and the launch ends with most of a millisecond of nearly empty GPU.
// So the frame is cut into square regions of tiles (which also keeps the rays in flight on one XCD close
// together in the scene), every launch records the run time of the slowest wavefront of each region, and a
// small ordering kernel behind it turns that into the order in which the next launch of the same frame
// geometry hands its regions out: slowest first, cheap ones last to fill the end.  The first launch of a
// geometry goes by a guess: regions by how many of the tree's shapes, and of their mirror images in the specular
// planes, are seen in and around them (rtk_launch_predict_regions; 4K dragon: a tenth over a frame ordered by measured
// costs, against a quarter to a third in index order).  Only the order of the work changes, never a pixel.
// Regions of a launch of local_rows rows of `width` pixels: side of a region in tiles as a power of two (0: the launch is
// too small to be ordered) and how many regions there are across and down.
struct RegionGrid { uint32_t log2, nx, ny; };
RegionGrid region_grid(uint32_t width, uint32_t local_rows) {
    const uint32_t tiles_x = (width + 7u) >> 3, tiles_y = (local_rows + 7u) >> 3;
    RegionGrid r{region_log2_for((uint64_t)tiles_x * tiles_y), 0u, 0u};
    if (r.log2 == 0) return r;
    const uint32_t side = 1u << r.log2;
    r.nx = (tiles_x + side - 1) / side; r.ny = (tiles_y + side - 1) / side;
    return r;
}

RegionKey region_key(const rt_frame& fr, uint32_t shard, uint32_t nshards, uint32_t tile_rows) {
    RegionKey key{};
    key.width = fr.width; key.height = fr.height;
    key.shard = shard; key.nshards = nshards; key.tile_rows = tile_rows; key.scene = g.scene_generation;
    return key;
}

int reserve_regions(Lane& lane, uint32_t n_regions) {
    if (n_regions <= lane.region_cap) return 0;
    if (lane.d_region_order) { (void)hipFree(lane.d_region_order); lane.d_region_order = nullptr; lane.region_cap = 0; }
    if (lane.d_region_block) { (void)hipFree(lane.d_region_block); lane.d_region_block = lane.d_region_cost = lane.d_region_spare = lane.d_region_last = nullptr; }
    HIP_TRY(hipMalloc((void**)&lane.d_region_order, (size_t)n_regions * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&lane.d_region_block, 3 * (size_t)n_regions * sizeof(uint32_t)));
    lane.d_region_cost = lane.d_region_block; lane.d_region_spare = lane.d_region_block + n_regions; lane.d_region_last = lane.d_region_block + 2 * (size_t)n_regions;
    lane.region_cap = n_regions; lane.region_n = 0;
    return 0;
}

// Orders the regions by the costs in `from`, which is the lane's cost or spare array; afterwards the lane's cost array is
// the one of the two that the kernel has cleared.
int order_regions(Lane& lane, uint32_t* from, uint32_t n) {
    uint32_t* other = from == lane.d_region_cost ? lane.d_region_spare : lane.d_region_cost;
    HIP_TRY(rtk_launch_sort_regions(from, n, lane.d_region_order, lane.d_region_last, other, lane.stream));
    lane.d_region_cost = other; lane.d_region_spare = from;
    return 0;
}

// A launch of another geometry has another number of regions: what lies behind the words that the kernels of the last
// one kept clear is stale.  (Same geometry, new scene: nothing to do.)
int clear_regions_for(Lane& lane, uint32_t n) {
    if (lane.region_n == n) return 0;
    HIP_TRY(hipMemsetAsync(lane.d_region_block, 0, 2 * (size_t)lane.region_cap * sizeof(uint32_t), lane.stream));
    lane.d_region_cost = lane.d_region_block; lane.d_region_spare = lane.d_region_block + lane.region_cap;
    lane.region_n = n;
    return 0;
}

int plan_regions(const Device& dev, Lane& lane, const rt_frame& fr, DevFrame* f) {
    const RegionGrid grid = region_grid(f->width, f->local_rows);
    f->region_log2 = grid.log2;
    f->regions_x = 0; f->n_regions = 0; f->region_order = nullptr; f->region_cost = nullptr;
    if (f->region_log2 == 0) return 0;
    f->regions_x = grid.nx; f->n_regions = grid.nx * grid.ny;
    if (f->n_regions > MAX_SORTED_REGIONS) return 0;          // (frames beyond 200 megapixels: raster order of regions)
    const RegionKey key = region_key(fr, f->shard, f->nshards, f->tile_rows);
    if (!lane.region_key_valid || key != lane.region_key) {
        if (reserve_regions(lane, f->n_regions) || clear_regions_for(lane, f->n_regions)) return 1;
        // No measured costs yet: the order follows a guess from where the tree's shapes are on screen, counted in the (clear)
        // cost array and summed into the spare one; equal costs, i.e. index order, when RT_FIRST_ORDER=raster asks for it.
        if (g.first_order_raster) { if (order_regions(lane, lane.d_region_cost, f->n_regions)) return 1; }
        else {
            HIP_TRY(rtk_launch_predict_regions(&dev.dsc, f, g.hdr.n_tree_shapes, lane.d_region_cost, lane.d_region_spare, lane.stream));
            if (order_regions(lane, lane.d_region_spare, f->n_regions)) return 1;      // (clears the counts)
        }
        lane.region_key = key; lane.region_key_valid = true;
    }
    f->region_order = lane.d_region_order;
    f->region_cost = lane.d_region_cost;
    return 0;
}

// What rt_tile_costs_* address: the launch of `frame` that rt_render / rt_render_frame make on GPU `gpu`.
int tile_cost_target(const rt_frame* frame, int gpu, RegionKey* key, uint32_t* n_regions) {
    if (!g.inited) return fail("rt_init has not been called");
    if (!g.have_scene) return fail("no scene uploaded");
    if (check_frame(frame)) return 1;
    if (gpu < 0 || gpu >= g.ndev) return fail("rt_tile_costs: no such GPU");
    const uint32_t nshards = g.have_comm ? (uint32_t)g.ndev : 1u;
    const uint32_t local_rows = rt_shard_rows(frame->height, (uint32_t)gpu, nshards, RT_TILE_ROWS);
    const RegionGrid grid = region_grid(frame->width, local_rows);
    const uint32_t n = grid.log2 == 0 ? 0u : grid.nx * grid.ny;
    *n_regions = n > MAX_SORTED_REGIONS ? 0u : n;
    *key = region_key(*frame, (uint32_t)gpu, nshards, RT_TILE_ROWS);
    return 0;
}

// The kernels' view of one shard of a frame (the region fields are filled in by plan_regions).
DevFrame dev_frame(const rt_frame& frame, uint32_t shard, uint32_t nshards, uint32_t tile_rows) {
    DevFrame f{};
    std::memcpy(f.pos, frame.pos, sizeof(f.pos)); std::memcpy(f.cv, frame.cv, sizeof(f.cv));
    std::memcpy(f.right, frame.right, sizeof(f.right)); std::memcpy(f.up, frame.up, sizeof(f.up));
    f.pw = frame.pixel_w; f.ph = frame.pixel_h; f.w2 = frame.half_w; f.h2 = frame.half_h;
    f.width = frame.width; f.height = frame.height; f.maxdepth = frame.maxdepth;
    std::memcpy(f.ambient, frame.ambient, sizeof(f.ambient));
    f.shard = shard; f.nshards = nshards; f.tile_rows = tile_rows; f.local_rows = rt_shard_rows(frame.height, shard, nshards, tile_rows);
    return f;
}

// Enqueues the render kernel for one shard on GPU d (current lane) into d_out (memory of that GPU).  `timed`:
// events around the kernel are recorded for collect_stats.
int enqueue_shard(int d, const rt_frame* frame, uint32_t shard, uint32_t nshards, uint32_t tile_rows, uint32_t* d_out, bool timed, int count_work) {
    if (!g.inited) return fail("rt_init has not been called");
    if (!g.have_scene) return fail("no scene uploaded");
    if (check_frame(frame)) return 1;
    if (nshards == 0 || shard >= nshards || tile_rows == 0) return fail("bad shard description");
    DevFrame f = dev_frame(*frame, shard, nshards, tile_rows);
    const uint32_t local_rows = f.local_rows;
    if (use(d)) return 1;
    Lane& lane = cur(d);
    if (local_rows == 0) {
        if (timed) { HIP_TRY(hipEventRecord(lane.ev0, lane.stream)); HIP_TRY(hipEventRecord(lane.ev1, lane.stream)); }
        return 0;
    }
    if (!d_out) return fail("output buffer is NULL");
    size_t npix = (size_t)local_rows * frame->width;
    uint32_t levels = frame->maxdepth > 1 ? frame->maxdepth - 1 : 0;
    const RenderPlan plan = rtk_render_plan((int)g.hdr.max_depth, frame->maxdepth);
    if (plan.lds_levels < levels && ensure_fold(lane, npix, levels - plan.lds_levels)) return 1;
    if (count_work) HIP_TRY(hipMemsetAsync(lane.d_counters, 0, RT_N_COUNTERS * sizeof(unsigned long long), lane.stream));
    if (timed) HIP_TRY(hipEventRecord(lane.ev0, lane.stream));      // (before the planning: the kernels that order a first frame count as part of it)
    if (plan_regions(g.devs[d], lane, *frame, &f)) return 1;
    HIP_TRY(rtk_launch_render(&g.devs[d].dsc, &f, d_out, lane.d_fold, lane.d_counters, (int)g.hdr.max_depth, plan, count_work, lane.stream));
    if (f.region_cost && order_regions(lane, f.region_cost, f.n_regions)) return 1;
    if (timed) HIP_TRY(hipEventRecord(lane.ev1, lane.stream));
    return 0;
}

// Waits for the shard enqueued last on GPU d (timed) and adds its work to *stats; kernel_ms is the longest of the GPUs.
int collect_stats(int d, const rt_frame* frame, uint32_t shard, uint32_t nshards, uint32_t tile_rows, int count_work, rt_stats* stats) {
    if (use(d)) return 1;
    Lane& lane = cur(d);
    HIP_TRY(hipEventSynchronize(lane.ev1));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, lane.ev0, lane.ev1));
    stats->kernel_ms = std::max(stats->kernel_ms, (double)ms);
    const uint32_t local_rows = rt_shard_rows(frame->height, shard, nshards, tile_rows);
    stats->primary += (uint64_t)local_rows * frame->width;
    if (count_work && local_rows) {
        unsigned long long c[RT_N_COUNTERS];
        HIP_TRY(hipMemcpy(c, lane.d_counters, sizeof(c), hipMemcpyDeviceToHost));
        stats->nodes_visited += c[0]; stats->prims_tested += c[1]; stats->shadow += c[2]; stats->secondary += c[3];
        stats->box_passes += c[4]; stats->shape_passes += c[5];
        for (int k = 0; k < 9; k++) stats->phase_passes[k] += c[6 + k];
    }
    return 0;
}

// One shard on the first GPU, as rt_render_tiles and the single-GPU rt_render want it.
int render_shard(const rt_frame* frame, uint32_t shard, uint32_t nshards, uint32_t tile_rows, uint32_t* d_out, rt_stats* stats, int count_work) {
    auto t0 = std::chrono::steady_clock::now();
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (enqueue_shard(0, frame, shard, nshards, tile_rows, d_out, stats != nullptr, count_work)) return 1;
    if (stats) {
        if (collect_stats(0, frame, shard, nshards, tile_rows, count_work, stats)) return 1;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

// A whole frame over every GPU the library drives; the frame ends up in the first GPU's buffer of the current
// lane (*d_frame).  Enqueue only, unless stats are wanted.
int render_frame_all(const rt_frame* frame, const uint32_t** d_frame, rt_stats* stats, int count_work) {
    auto t0 = std::chrono::steady_clock::now();
    if (!g.inited) return fail("rt_init has not been called");
    if (!g.have_scene) return fail("no scene uploaded");
    if (check_frame(frame)) return 1;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    const uint32_t n = (uint32_t)g.ndev, tile_rows = RT_TILE_ROWS, W = frame->width, H = frame->height;
    const size_t npix = (size_t)W * H;
    if (use(0)) return 1;
    Lane& l0 = cur(0);
    if (grow(&l0.d_frame, &l0.frame_cap, npix)) return 1;
    if (!g.have_comm) {                                      // one GPU, no communicator: straight into the frame
        if (enqueue_shard(0, frame, 0, 1, tile_rows, l0.d_frame, stats != nullptr, count_work)) return 1;
    } else {
        const uint32_t stride = rt_shard_rows(H, 0, n, tile_rows);            // shard 0 owns the most rows
        if (grow(&l0.d_gathered, &l0.gathered_cap, (size_t)n * stride * W)) return 1;
        for (uint32_t d = 0; d < n; d++) {
            uint32_t* rows = l0.d_gathered;
            if (d > 0) {
                if (use((int)d)) return 1;
                if (grow(&cur((int)d).d_rows, &cur((int)d).rows_cap, (size_t)stride * W)) return 1;
                rows = cur((int)d).d_rows;
            }
            if (enqueue_shard((int)d, frame, d, n, tile_rows, rows, stats != nullptr, count_work)) return 1;
        }
        if (n > 1) {
            // The gather runs on the communicators' own streams: each waits for its GPU's kernel of this frame, and the
            // reassembly on the first GPU's lane waits for the gather.
            for (uint32_t d = 0; d < n; d++) {
                if (use((int)d)) return 1;
                HIP_TRY(hipEventRecord(cur((int)d).rows_ready, cur((int)d).stream));
                HIP_TRY(hipStreamWaitEvent(g.devs[d].comm_stream, cur((int)d).rows_ready, 0));
            }
            NCCL_TRY(ncclGroupStart());
            for (uint32_t d = 1; d < n; d++) {
                const size_t count = (size_t)rt_shard_rows(H, d, n, tile_rows) * W;
                if (count == 0) continue;
                NCCL_TRY(ncclSend(cur((int)d).d_rows, count, ncclUint32, 0, g.comms[d], g.devs[d].comm_stream));
                NCCL_TRY(ncclRecv(l0.d_gathered + (size_t)d * stride * W, count, ncclUint32, (int)d, g.comms[0], g.devs[0].comm_stream));
            }
            NCCL_TRY(ncclGroupEnd());
            for (uint32_t d = n; d-- > 0;) {                 // later work of the lane (the reassembly; the next frame's kernel, which overwrites the rows) comes after the gather
                if (use((int)d)) return 1;
                HIP_TRY(hipEventRecord(cur((int)d).gathered, g.devs[d].comm_stream));
                HIP_TRY(hipStreamWaitEvent(cur((int)d).stream, cur((int)d).gathered, 0));
            }
        }
        if (use(0)) return 1;
        HIP_TRY(rtk_launch_assemble(l0.d_gathered, stride, n, tile_rows, W, H, l0.d_frame, l0.stream));
    }
    if (stats) {
        const uint32_t shards = g.have_comm ? n : 1u;
        for (uint32_t d = shards; d-- > 0;) if (collect_stats((int)d, frame, d, shards, tile_rows, count_work, stats)) return 1;
        HIP_TRY(hipStreamSynchronize(l0.stream));           // (collect_stats left the first GPU current)
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    if (use(0)) return 1;
    if (d_frame) *d_frame = l0.d_frame;
    return 0;
}

int init_devices(int first_device, int count) {
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (first_device < 0 || count < 1 || first_device + count > n) return fail("rt_init: no such HIP device");
    for (int d = 0; d < count; d++) {
        Device& dev = g.devs[d];
        dev.device = first_device + d;
        g.ndev = d + 1;
        HIP_TRY(hipSetDevice(dev.device));
        for (Lane& l : dev.lanes) {
            HIP_TRY(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreate(&l.ev0));
            HIP_TRY(hipEventCreate(&l.ev1));
            HIP_TRY(hipEventCreateWithFlags(&l.rows_ready, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&l.gathered, hipEventDisableTiming));
            HIP_TRY(hipMalloc((void**)&l.d_counters, 16 * sizeof(unsigned long long)));
        }
        HIP_TRY(hipStreamCreateWithFlags(&dev.comm_stream, hipStreamNonBlocking));
        HIP_TRY(rtk_noalloc_setup(dev.device));
    }
    HIP_TRY(hipSetDevice(first().device));
    g.lane = 0;
    g.inited = true;
    const char* staging = std::getenv("RT_RENDER_STAGING");
    g.stage_by_copy = staging && std::strcmp(staging, "copy") == 0;
    const char* kernel = std::getenv("RT_RENDER_KERNEL");
    g.generic_kernel = kernel && std::strcmp(kernel, "generic") == 0;
    const char* first_order = std::getenv("RT_FIRST_ORDER");
    g.first_order_raster = first_order && std::strcmp(first_order, "raster") == 0;
    return 0;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return rte::last_error(); }

int rti_bound_device(void) { return g.inited ? first().device : -1; }

int rt_init(int device) {
    if (g.inited) {
        if (!g.have_comm && first().device == device) return 0;
        return fail("rt_init: already bound to another device; call rt_shutdown first");
    }
    if (init_devices(device, 1)) { const std::string m = rte::last_error(); rt_shutdown(); return fail(m); }
    return 0;
}

int rt_init_multi(int ngpus) {
    if (g.inited) {
        if (g.have_comm && g.ndev == ngpus) return 0;
        return fail("rt_init_multi: already initialised differently; call rt_shutdown first");
    }
    if (ngpus < 1 || ngpus > RT_MAX_GPUS) return fail("rt_init_multi: between 1 and " + std::to_string(RT_MAX_GPUS) + " GPUs");
    if (init_devices(0, ngpus)) { const std::string m = rte::last_error(); rt_shutdown(); return fail(m); }
    int ids[RT_MAX_GPUS];
    for (int d = 0; d < ngpus; d++) ids[d] = g.devs[d].device;
    ncclResult_t e = ncclCommInitAll(g.comms, ngpus, ids);
    if (e != ncclSuccess) { rt_shutdown(); return fail_nccl("ncclCommInitAll", e); }
    g.have_comm = true;
    HIP_TRY(hipSetDevice(first().device));
    return 0;
}

int rt_device_count(void) { return g.inited ? g.ndev : 0; }

int rt_upload_scene(const rt_scene* scene) {
    if (!g.inited) return fail("rt_init has not been called");
    if (!scene) return fail("rt_scene is NULL");
    rtu::NoInitVector<uint8_t> blob; rt_bvh_info info{}; std::string err;
    if (!rtp::pack_scene(*scene, &blob, &info, &err)) return fail(err);
    rtl::BlobHeader h; std::memcpy(&h, blob.data(), sizeof(h));
    if (!rtp::validate_header(h, blob.size(), &err)) return fail(err);
    if (sync_lanes()) return 1;
    release_scene();
    HIP_TRY(hipMalloc(&first().d_blob, blob.size()));
    first().blob_owned = true; first().blob_bytes = blob.size();
    HIP_TRY(hipMemcpyAsync(first().d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice, cur().stream));
    HIP_TRY(hipStreamSynchronize(cur().stream));
    g.info = info;
    return scene_is_resident(h);
}

int rt_upload_scene_ex(const rt_scene* scene, uint32_t flags) {
    if (!(flags & RT_BUILD_ON_GPU)) return rt_upload_scene(scene);
    auto t0 = std::chrono::steady_clock::now();
    if (!g.inited) return fail("rt_init has not been called");
    if (!scene) return fail("rt_scene is NULL");
    rtp::SceneParts parts; std::string err;
    if (!rtp::split_scene(*scene, &parts, &err)) return fail(err);
    if (!parts.use_tree || parts.tree.size() < 2) return rt_upload_scene(scene);       // nothing to build a tree over
    rtg::SceneParts in{};
    in.tree = parts.tree.data(); in.n_tree = (uint32_t)parts.tree.size();
    in.list = parts.list.data(); in.n_list = (uint32_t)parts.list.size();
    in.mats = parts.mats.data(); in.n_mats = (uint32_t)parts.mats.size();
    in.lights = parts.lights.data(); in.n_lights = (uint32_t)parts.lights.size();
    in.finite_shading = parts.finite_shading; in.tree_kinds = parts.tree_kinds; in.pad_abs = parts.pad_abs; in.depth_limit = rtp::TREE_DEPTH_LIMIT;
    if (sync_lanes()) return 1;
    rtg::BuiltScene built{};
    rtg::BuildStatus st = rtg::build_scene_on_gpu(in, cur().stream, &built, &err);
    if (st == rtg::BUILD_TOO_DEEP) return rt_upload_scene(scene);                       // the host builder bounds the depth
    if (st != rtg::BUILD_OK) return fail(err);
    if (!rtp::validate_header(built.header, built.bytes, &err)) { (void)hipFree(built.d_blob); return fail(err); }
    release_scene();
    first().d_blob = built.d_blob; first().blob_bytes = built.bytes; first().blob_owned = true;
    g.info = rt_bvh_info{};
    if (scene_is_resident(built.header)) return 1;
    g.info.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

int rt_upload_scene_blob(const void* host_blob, size_t bytes) {
    if (!g.inited) return fail("rt_init has not been called");
    std::string err;
    if (!rtp::validate_blob((const uint8_t*)host_blob, bytes, &err)) return fail(err);
    rtl::BlobHeader h; std::memcpy(&h, host_blob, sizeof(h));
    if (sync_lanes()) return 1;
    release_scene();
    HIP_TRY(hipMalloc(&first().d_blob, bytes));
    first().blob_owned = true; first().blob_bytes = bytes;
    HIP_TRY(hipMemcpyAsync(first().d_blob, host_blob, bytes, hipMemcpyHostToDevice, cur().stream));
    HIP_TRY(hipStreamSynchronize(cur().stream));
    g.info = rt_bvh_info{};
    return scene_is_resident(h);
}

int rt_download_scene_blob(void* host_dst, size_t bytes) {
    if (!g.have_scene) return fail("no scene uploaded");
    if (!host_dst || bytes != first().blob_bytes) return fail("rt_download_scene_blob: destination must be exactly the blob size");
    if (use(0)) return 1;
    HIP_TRY(hipMemcpyAsync(host_dst, first().d_blob, bytes, hipMemcpyDeviceToHost, cur().stream));
    HIP_TRY(hipStreamSynchronize(cur().stream));
    return 0;
}

int rt_get_bvh_info(rt_bvh_info* out) {
    if (!g.have_scene) return fail("no scene uploaded");
    if (!out) return fail("rt_bvh_info is NULL");
    *out = g.info;
    return 0;
}

int rt_scene_blob(void** device_ptr, size_t* bytes) {
    if (!g.have_scene) return fail("no scene uploaded");
    if (!device_ptr || !bytes) return fail("rt_scene_blob: NULL argument");
    *device_ptr = first().d_blob; *bytes = first().blob_bytes;
    return 0;
}

int rt_copy_scene_blob(void* device_dst, size_t bytes) {
    if (!g.have_scene) return fail("no scene uploaded");
    if (!device_dst || bytes != first().blob_bytes) return fail("rt_copy_scene_blob: destination must be exactly the blob size");
    if (use(0)) return 1;
    HIP_TRY(hipMemcpyAsync(device_dst, first().d_blob, bytes, hipMemcpyDeviceToDevice, cur().stream));
    HIP_TRY(hipStreamSynchronize(cur().stream));
    return 0;
}

int rt_adopt_scene_blob(void* device_ptr, size_t bytes) {
    if (!g.inited) return fail("rt_init has not been called");
    if (!device_ptr || bytes < sizeof(rtl::BlobHeader)) return fail("rt_adopt_scene_blob: buffer too small");
    if (sync_lanes()) return 1;
    rtl::BlobHeader h;
    HIP_TRY(hipMemcpy(&h, device_ptr, sizeof(h), hipMemcpyDeviceToHost));
    std::string err;
    if (!rtp::validate_header(h, bytes, &err)) return fail(err);
    release_scene();
    first().d_blob = device_ptr; first().blob_bytes = bytes; first().blob_owned = false;
    g.info = rt_bvh_info{};
    return scene_is_resident(h);
}

uint32_t rt_shard_rows(uint32_t height, uint32_t shard, uint32_t nshards, uint32_t tile_rows) {
    if (nshards == 0 || tile_rows == 0 || shard >= nshards) return 0;
    uint64_t rows = 0;
    uint32_t ntiles = (height + tile_rows - 1) / tile_rows;
    for (uint32_t t = shard; t < ntiles; t += nshards) {
        uint32_t r0 = t * tile_rows, r1 = r0 + tile_rows;
        if (r1 > height) r1 = height;
        rows += r1 - r0;
    }
    return (uint32_t)rows;
}

int rt_render_tiles(const rt_frame* frame, uint32_t shard, uint32_t nshards, uint32_t tile_rows, uint32_t* d_out, rt_stats* stats,
                    int count_work) {
    return render_shard(frame, shard, nshards, tile_rows, d_out, stats, count_work);
}

int rt_render_frame(const rt_frame* frame, const uint32_t** d_frame, rt_stats* stats, int count_work) {
    return render_frame_all(frame, d_frame, stats, count_work);
}

int rt_render(const rt_frame* frame, uint32_t* out_pixels, rt_stats* stats, int count_work) {
    auto t0 = std::chrono::steady_clock::now();
    if (!g.inited) return fail("rt_init has not been called");
    if (check_frame(frame)) return 1;
    if (!out_pixels) return fail("output buffer is NULL");
    if (use(0)) return 1;
    size_t npix = (size_t)frame->width * frame->height;
    if (npix > g.stage_cap) {
        if (g.h_stage) { (void)hipHostFree(g.h_stage); g.h_stage = nullptr; g.stage_cap = 0; }
        HIP_TRY(hipHostMalloc((void**)&g.h_stage, npix * sizeof(uint32_t), hipHostMallocMapped));
        g.stage_cap = npix;
    }
    rt_stats local{};
    const uint32_t* d_src = nullptr;
    if (g.have_comm) {
        if (render_frame_all(frame, &d_src, &local, count_work)) return 1;
    } else if (!g.stage_by_copy) {
        // One GPU: the kernel stores its pixels straight into the pinned staging buffer (host memory mapped into the GPU's
        // address space), so the frame crosses the bus while it is being rendered, 4 bytes per pixel spread over the whole
        // launch, instead of in one copy behind it.  What is left after the kernel is the host's copy into the caller's memory.
        void* d_stage = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&d_stage, g.h_stage, 0));
        if (render_shard(frame, 0, 1, RT_TILE_ROWS, (uint32_t*)d_stage, &local, count_work)) return 1;      // waits for the kernel (stats)
        rtu::parallel_copy(out_pixels, g.h_stage, npix * sizeof(uint32_t));
        if (stats) {
            *stats = local;
            stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        return 0;
    } else {
        if (grow(&g.d_out, &g.out_cap, npix)) return 1;
        if (render_shard(frame, 0, 1, RT_TILE_ROWS, g.d_out, &local, count_work)) return 1;
        d_src = g.d_out;
    }
    const size_t n_pieces = npix >= (1u << 20) ? 8 : 1, per_piece = (npix + n_pieces - 1) / n_pieces;
    for (size_t k = 0; k < n_pieces; k++) {
        const size_t a = std::min(npix, k * per_piece), b = std::min(npix, a + per_piece);
        if (!g.piece_done[k]) HIP_TRY(hipEventCreateWithFlags(&g.piece_done[k], hipEventDisableTiming));
        if (b > a) HIP_TRY(hipMemcpyAsync(g.h_stage + a, d_src + a, (b - a) * sizeof(uint32_t), hipMemcpyDeviceToHost, cur().stream));
        HIP_TRY(hipEventRecord(g.piece_done[k], cur().stream));
    }
    for (size_t k = 0; k < n_pieces; k++) {
        const size_t a = std::min(npix, k * per_piece), b = std::min(npix, a + per_piece);
        HIP_TRY(hipEventSynchronize(g.piece_done[k]));
        rtu::parallel_copy(out_pixels + a, g.h_stage + a, (b - a) * sizeof(uint32_t));
    }
    if (stats) {
        *stats = local;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

int rt_assemble_tiles(const uint32_t* d_gathered, uint32_t rows_per_shard_stride, uint32_t nshards, uint32_t tile_rows, uint32_t width,
                      uint32_t height, uint32_t* d_frame) {
    if (!g.inited) return fail("rt_init has not been called");
    if (!d_gathered || !d_frame) return fail("rt_assemble_tiles: NULL buffer");
    if (nshards == 0 || tile_rows == 0 || width == 0 || height == 0) return fail("rt_assemble_tiles: bad geometry");
    for (uint32_t s = 0; s < nshards; s++)
        if (rt_shard_rows(height, s, nshards, tile_rows) > rows_per_shard_stride) return fail("rt_assemble_tiles: stride smaller than a shard");
    if (use(0)) return 1;
    HIP_TRY(rtk_launch_assemble(d_gathered, rows_per_shard_stride, nshards, tile_rows, width, height, d_frame, cur().stream));
    return 0;
}

int rt_compare_frames(const uint32_t* d_a, uint32_t width_a, uint32_t height_a, const uint32_t* d_b, uint32_t width_b, uint32_t height_b,
                      uint32_t* d_diff, uint64_t* count) {
    if (!g.inited) return fail("rt_init has not been called");
    if (height_a != height_b || width_a != width_b) return fail("Images have different dimensions");            // mod.rs:55-57
    if (!d_a || !d_b) return fail("rt_compare_frames: NULL image");
    if (((uintptr_t)d_a | (uintptr_t)d_b | (uintptr_t)d_diff) & 15u) return fail("rt_compare_frames: buffers must be 16-byte aligned");
    if (use(0)) return 1;
    HIP_TRY(hipMemsetAsync(cur().d_counters, 0, sizeof(unsigned long long), cur().stream));
    HIP_TRY(rtk_launch_compare(d_a, d_b, (size_t)width_a * height_a, d_diff, cur().d_counters, cur().stream));
    unsigned long long c = 0;
    HIP_TRY(hipMemcpyAsync(&c, cur().d_counters, sizeof(c), hipMemcpyDeviceToHost, cur().stream));
    HIP_TRY(hipStreamSynchronize(cur().stream));
    if (count) *count = c;
    return 0;
}

int rt_pack_rgb(const uint32_t* d_pixels, uint32_t width, uint32_t height, uint8_t* d_rgb) {
    if (!g.inited) return fail("rt_init has not been called");
    if (!d_pixels || !d_rgb) return fail("rt_pack_rgb: NULL buffer");
    if (((uintptr_t)d_pixels & 15u) || ((uintptr_t)d_rgb & 3u)) return fail("rt_pack_rgb: pixels must be 16-byte and rgb 4-byte aligned");
    if (use(0)) return 1;
    HIP_TRY(rtk_launch_pack_rgb(d_pixels, (size_t)width * height, d_rgb, cur().stream));
    return 0;
}

int rt_tile_costs_count(const rt_frame* frame, int gpu, uint32_t* count) {
    RegionKey key; uint32_t n = 0;
    if (!count) return fail("rt_tile_costs_count: count is NULL");
    if (tile_cost_target(frame, gpu, &key, &n)) return 1;
    *count = n;
    return 0;
}

int rt_tile_costs_export(const rt_frame* frame, int gpu, uint32_t* costs, uint32_t count) {
    RegionKey key; uint32_t n = 0;
    if (tile_cost_target(frame, gpu, &key, &n)) return 1;
    if (count != n) return fail("rt_tile_costs_export: count differs from rt_tile_costs_count");
    if (n == 0) return 0;
    if (!costs) return fail("rt_tile_costs_export: costs is NULL");
    if (use(gpu)) return 1;
    Lane& lane = cur(gpu);
    if (!lane.region_key_valid || key != lane.region_key)
        return fail("rt_tile_costs_export: this frame has not been rendered on the current lane since the scene was uploaded");
    HIP_TRY(hipStreamSynchronize(lane.stream));
    HIP_TRY(hipMemcpy(costs, lane.d_region_last, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return use(0);
}

int rt_tile_costs_guess(const rt_frame* frame, int gpu, uint32_t* costs, uint32_t count) {
    RegionKey key; uint32_t n = 0;
    if (tile_cost_target(frame, gpu, &key, &n)) return 1;
    if (count != n) return fail("rt_tile_costs_guess: count differs from rt_tile_costs_count");
    if (n == 0) return 0;
    if (!costs) return fail("rt_tile_costs_guess: costs is NULL");
    if (use(gpu)) return 1;
    Lane& lane = cur(gpu);
    const uint32_t nshards = g.have_comm ? (uint32_t)g.ndev : 1u;
    DevFrame f = dev_frame(*frame, (uint32_t)gpu, nshards, RT_TILE_ROWS);
    const RegionGrid grid = region_grid(f.width, f.local_rows);
    f.region_log2 = grid.log2; f.regions_x = grid.nx; f.n_regions = n;
    uint32_t* d_guess = nullptr;                              // (the lane's own arrays belong to its launches)
    HIP_TRY(hipMalloc((void**)&d_guess, 2 * (size_t)n * sizeof(uint32_t)));
    hipError_t e = hipMemsetAsync(d_guess + n, 0, (size_t)n * sizeof(uint32_t), lane.stream);
    if (e == hipSuccess) e = rtk_launch_predict_regions(&g.devs[gpu].dsc, &f, g.hdr.n_tree_shapes, d_guess + n, d_guess, lane.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(costs, d_guess, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, lane.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(lane.stream);
    (void)hipFree(d_guess);
    if (e != hipSuccess) return fail_hip("rt_tile_costs_guess", e);
    return use(0);
}

int rt_tile_costs_import(const rt_frame* frame, int gpu, const uint32_t* costs, uint32_t count) {
    RegionKey key; uint32_t n = 0;
    if (tile_cost_target(frame, gpu, &key, &n)) return 1;
    if (count != n) return fail("rt_tile_costs_import: count differs from rt_tile_costs_count");
    if (n == 0) return 0;
    if (!costs) return fail("rt_tile_costs_import: costs is NULL");
    if (use(gpu)) return 1;
    Lane& lane = cur(gpu);
    HIP_TRY(hipStreamSynchronize(lane.stream));
    if (reserve_regions(lane, n) || clear_regions_for(lane, n)) return 1;
    HIP_TRY(hipStreamSynchronize(lane.stream));
    HIP_TRY(hipMemcpy(lane.d_region_spare, costs, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (order_regions(lane, lane.d_region_spare, n)) return 1;
    lane.region_key = key; lane.region_key_valid = true;
    return use(0);
}

int rt_select_lane(int lane) {
    if (!g.inited) return fail("rt_init has not been called");
    if (lane < 0 || lane >= N_LANES) return fail("rt_select_lane: the lanes are 0 .. " + std::to_string(N_LANES - 1));
    g.lane = lane;
    return 0;
}

void* rt_stream(void) { return g.inited ? (void*)cur().stream : nullptr; }

int rt_synchronize(void) {
    if (!g.inited) return fail("rt_init has not been called");
    for (int d = g.ndev - 1; d >= 0; d--) {                 // the first GPU last: its stream ends a frame, and it stays current
        if (use(d)) return 1;
        HIP_TRY(hipStreamSynchronize(cur(d).stream));
    }
    return 0;
}

int rt_wait_event(void* hip_event) {
    if (!g.inited) return fail("rt_init has not been called");
    if (!hip_event) return fail("rt_wait_event: event is NULL");
    if (use(0)) return 1;
    HIP_TRY(hipStreamWaitEvent(cur().stream, (hipEvent_t)hip_event, 0));
    return 0;
}

int rt_record_event(void* hip_event) {
    if (!g.inited) return fail("rt_init has not been called");
    if (!hip_event) return fail("rt_record_event: event is NULL");
    if (use(0)) return 1;
    HIP_TRY(hipEventRecord((hipEvent_t)hip_event, cur().stream));
    return 0;
}

void rt_shutdown(void) {
    if (g.ndev == 0) { g = State(); return; }
    (void)sync_lanes();
    release_scene();
    if (g.have_comm) for (int d = 0; d < g.ndev; d++) if (g.comms[d]) (void)ncclCommDestroy(g.comms[d]);
    (void)hipSetDevice(first().device);
    if (g.d_out) (void)hipFree(g.d_out);
    if (g.h_stage) (void)hipHostFree(g.h_stage);
    for (hipEvent_t e : g.piece_done) if (e) (void)hipEventDestroy(e);
    for (int d = 0; d < g.ndev; d++) {
        (void)hipSetDevice(g.devs[d].device);
        for (Lane& l : g.devs[d].lanes) {
            if (l.d_fold) (void)hipFree(l.d_fold);
            if (l.d_counters) (void)hipFree(l.d_counters);
            if (l.d_region_order) (void)hipFree(l.d_region_order);
            if (l.d_region_block) (void)hipFree(l.d_region_block);
            if (l.d_rows) (void)hipFree(l.d_rows);
            if (l.d_gathered) (void)hipFree(l.d_gathered);
            if (l.d_frame) (void)hipFree(l.d_frame);
            if (l.ev0) (void)hipEventDestroy(l.ev0);
            if (l.ev1) (void)hipEventDestroy(l.ev1);
            if (l.rows_ready) (void)hipEventDestroy(l.rows_ready);
            if (l.gathered) (void)hipEventDestroy(l.gathered);
            if (l.stream) (void)hipStreamDestroy(l.stream);
        }
        if (g.devs[d].comm_stream) (void)hipStreamDestroy(g.devs[d].comm_stream);
    }
    g = State();
}

}  // extern "C"
