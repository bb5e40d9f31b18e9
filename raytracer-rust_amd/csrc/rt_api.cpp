// rt_api.cpp -- implementation of the C ABI declared in include/rt_amd.h.
//
// One process drives one GPU.  The scene lives in one device blob (layout.h) between
// rt_upload_scene / rt_adopt_scene_blob and rt_shutdown; work buffers (frame, per-depth fold
// records, counters) grow on demand and are reused across calls, one set per lane (a stream
// with its own buffers: launches on different lanes may overlap on the GPU).  There is no CPU fallback:
// every entry point that needs the GPU fails with an error when HIP does.
#include "../../include/rt_amd.h"

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

struct RegionKey { uint32_t width, height, shard, nshards, tile_rows; uint64_t scene; };   // what the regions of a launch depend on

// A lane is a stream with the work buffers of the launches made on it.  Launches on one lane run in order;
// launches on different lanes share nothing but the (read-only) scene and may overlap on the GPU, which is how
// a caller that renders frame after frame hides the ramp-up and the drain of every launch (rt_select_lane).
struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float4* d_fold = nullptr; size_t fold_cap = 0;          // records
    unsigned long long* d_counters = nullptr;               // nodes, prims, shadow, secondary, wave-level passes
    // order in which the tile regions of a launch are handed out, and what it was computed for
    uint32_t* d_region_order = nullptr; uint32_t* d_region_cost = nullptr; uint32_t region_cap = 0;
    RegionKey region_key{}; bool region_key_valid = false;
};
constexpr int N_LANES = RT_LANES;

struct State {
    bool inited = false;
    int device = -1;
    Lane lanes[N_LANES];
    int lane = 0;                                           // the lane the next call uses
    void* d_blob = nullptr; size_t blob_bytes = 0; bool blob_owned = false; bool have_scene = false;
    rtl::BlobHeader hdr{};
    DevScene dsc{};
    rt_bvh_info info{};
    uint32_t* d_out = nullptr; size_t out_cap = 0;          // pixels (rt_render's frame before it is copied to the host)
    // rt_render's way to the caller's (pageable) memory: pinned staging buffer, filled piece by piece by the copy
    // engine while the pieces that have arrived are copied on by the host's threads
    uint32_t* h_stage = nullptr; size_t stage_cap = 0;
    hipEvent_t piece_done[8] = {};
    uint64_t scene_generation = 0;                          // bumped whenever another scene becomes resident
};
State g;
Lane& cur() { return g.lanes[g.lane]; }

int fail(const std::string& m) { rte::set_error(m); return 1; }
int fail_hip(const char* what, hipError_t e) { rte::set_error(std::string(what) + ": " + hipGetErrorString(e)); return 1; }

#define HIP_TRY(expr)                                                  \
    do {                                                               \
        hipError_t e_ = (expr);                                        \
        if (e_ != hipSuccess) return fail_hip(#expr, e_);              \
    } while (0)

// Scene changes wait for the launches of every lane.
int sync_lanes() {
    for (Lane& l : g.lanes) if (l.stream) HIP_TRY(hipStreamSynchronize(l.stream));
    return 0;
}

void release_scene() {
    if (g.d_blob && g.blob_owned) (void)hipFree(g.d_blob);
    g.d_blob = nullptr; g.blob_bytes = 0; g.blob_owned = false; g.have_scene = false;
}

void point_at_blob(void* d_blob, const rtl::BlobHeader& h) {
    const char* base = (const char*)d_blob;
    g.hdr = h;
    g.dsc.nodes = (const uint4*)(base + h.off_nodes);
    g.dsc.tree = (const float4*)(base + h.off_tree_shapes);
    g.dsc.list = (const float4*)(base + h.off_list_shapes);
    g.dsc.mats = (const float4*)(base + h.off_materials);
    g.dsc.lights = (const float4*)(base + h.off_lights);
    g.dsc.n_list = h.n_list_shapes; g.dsc.n_lights = h.n_lights; g.dsc.root_ref = h.root_ref; g.dsc.finite_shading = h.finite_shading;
    for (int k = 0; k < 3; k++) {
        g.dsc.grid_origin[k] = h.grid_origin[k]; g.dsc.grid_scale[k] = h.grid_scale[k];
        g.dsc.inv_limit[k] = std::fmin(1e30f, 1e33f / std::fmax(h.grid_scale[k], 1e-30f));
    }
    g.info.node_pairs = h.n_pairs; g.info.leaves = h.n_leaves; g.info.max_depth = h.max_depth; g.info.max_leaf = h.max_leaf;
    g.info.bvh_prims = h.n_tree_shapes; g.info.list_prims = h.n_list_shapes; g.info.sah_cost = h.sah_cost;
    g.have_scene = true;
    g.scene_generation++;
}

int check_frame(const rt_frame* f) {
    if (!f) return fail("rt_frame is NULL");
    if (f->width == 0 || f->height == 0) return fail("Width and height must be greater than zero");
    if ((uint64_t)f->width * f->height > (1ull << 31)) return fail("frame larger than 2^31 pixels");
    return 0;
}

// Global array for the records of the reflection levels, for frames whose records do not fit in LDS.
int ensure_fold(size_t npix, size_t levels) {
    const size_t limit = (200ull << 30) / sizeof(float4);
    if (levels != 0 && npix > limit / levels) return fail("maxdepth x pixels needs more than 200 GiB of fold records");
    const size_t records = npix * levels;
    if (records <= cur().fold_cap) return 0;
    if (cur().d_fold) { (void)hipFree(cur().d_fold); cur().d_fold = nullptr; cur().fold_cap = 0; }
    HIP_TRY(hipMalloc((void**)&cur().d_fold, records * sizeof(float4)));
    cur().fold_cap = records;
    return 0;
}

// Tile order of a launch.  Wavefront run times are heavy-tailed (on the 4K dragon frame: median 40 us, mean
// 140 us, the slowest 1.4 ms) and the slow ones sit where the tree's shapes are on screen; handed out in plain
// raster order the last of them start late and the launch ends with most of a millisecond of nearly empty GPU.
// So the frame is cut into square regions of tiles (which also keeps the rays in flight on one XCD close
// together in the scene), every launch records the run time of the slowest wavefront of each region, and a
// one-workgroup sort behind the kernel turns that into the order in which the next launch of the same frame
// geometry hands its regions out: slowest first, cheap ones last to fill the end.  The first launch of a
// geometry uses raster order.  Only the order of the work changes, never a pixel.
int plan_regions(const rt_frame& fr, DevFrame* f) {
    const uint32_t tiles_x = (f->width + 7u) >> 3, tiles_y = (f->local_rows + 7u) >> 3;
    f->region_log2 = region_log2_for((uint64_t)tiles_x * tiles_y);
    f->regions_x = 0; f->n_regions = 0; f->region_order = nullptr; f->region_cost = nullptr;
    if (f->region_log2 == 0) return 0;
    const uint32_t side = 1u << f->region_log2;
    const uint32_t rx = (tiles_x + side - 1) / side, ry = (tiles_y + side - 1) / side;
    f->regions_x = rx; f->n_regions = rx * ry;
    if (f->n_regions > MAX_SORTED_REGIONS) return 0;          // (frames beyond 200 megapixels: raster order of regions)
    RegionKey key{};
    key.width = fr.width; key.height = fr.height;
    key.shard = f->shard; key.nshards = f->nshards; key.tile_rows = f->tile_rows; key.scene = g.scene_generation;
    if (!cur().region_key_valid || std::memcmp(&key, &cur().region_key, sizeof(key)) != 0) {
        if (f->n_regions > cur().region_cap) {
            if (cur().d_region_order) { (void)hipFree(cur().d_region_order); cur().d_region_order = nullptr; cur().region_cap = 0; }
            if (cur().d_region_cost) { (void)hipFree(cur().d_region_cost); cur().d_region_cost = nullptr; }
            HIP_TRY(hipMalloc((void**)&cur().d_region_order, (size_t)f->n_regions * sizeof(uint32_t)));
            HIP_TRY(hipMalloc((void**)&cur().d_region_cost, (size_t)f->n_regions * sizeof(uint32_t)));
            cur().region_cap = f->n_regions;
        }
        // All costs equal: the sort leaves the regions in index order.
        HIP_TRY(hipMemsetAsync(cur().d_region_cost, 0, (size_t)f->n_regions * sizeof(uint32_t), cur().stream));
        HIP_TRY(rtk_launch_sort_regions(cur().d_region_cost, f->n_regions, cur().d_region_order, cur().stream));
        cur().region_key = key; cur().region_key_valid = true;
    }
    HIP_TRY(hipMemsetAsync(cur().d_region_cost, 0, (size_t)f->n_regions * sizeof(uint32_t), cur().stream));
    f->region_order = cur().d_region_order;
    f->region_cost = cur().d_region_cost;
    return 0;
}

// Launches the render kernel for one shard into d_out (device).  With stats != NULL the call
// is timed with HIP events on the library's stream and waits for completion.
int render_shard(const rt_frame* frame, uint32_t shard, uint32_t nshards, uint32_t tile_rows, uint32_t* d_out, rt_stats* stats, int count_work) {
    auto t0 = std::chrono::steady_clock::now();
    if (!g.inited) return fail("rt_init has not been called");
    if (!g.have_scene) return fail("no scene uploaded");
    if (check_frame(frame)) return 1;
    if (nshards == 0 || shard >= nshards || tile_rows == 0) return fail("bad shard description");
    uint32_t local_rows = rt_shard_rows(frame->height, shard, nshards, tile_rows);
    DevFrame f{};
    std::memcpy(f.pos, frame->pos, sizeof(f.pos)); std::memcpy(f.cv, frame->cv, sizeof(f.cv));
    std::memcpy(f.right, frame->right, sizeof(f.right)); std::memcpy(f.up, frame->up, sizeof(f.up));
    f.pw = frame->pixel_w; f.ph = frame->pixel_h; f.w2 = frame->half_w; f.h2 = frame->half_h;
    f.width = frame->width; f.height = frame->height; f.maxdepth = frame->maxdepth;
    std::memcpy(f.ambient, frame->ambient, sizeof(f.ambient));
    f.shard = shard; f.nshards = nshards; f.tile_rows = tile_rows; f.local_rows = local_rows;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (local_rows == 0) return 0;
    if (!d_out) return fail("output buffer is NULL");
    size_t npix = (size_t)local_rows * frame->width;
    uint32_t levels = frame->maxdepth > 1 ? frame->maxdepth - 1 : 0;
    if (!rtk_fold_in_lds((int)g.hdr.max_depth, frame->maxdepth) && ensure_fold(npix, levels)) return 1;
    if (count_work) HIP_TRY(hipMemsetAsync(cur().d_counters, 0, 8 * sizeof(unsigned long long), cur().stream));
    if (plan_regions(*frame, &f)) return 1;
    if (stats) HIP_TRY(hipEventRecord(cur().ev0, cur().stream));
    HIP_TRY(rtk_launch_render(&g.dsc, &f, d_out, cur().d_fold, cur().d_counters, (int)g.hdr.max_depth, count_work, cur().stream));
    if (f.region_cost) HIP_TRY(rtk_launch_sort_regions(f.region_cost, f.n_regions, cur().d_region_order, cur().stream));
    if (stats) {
        HIP_TRY(hipEventRecord(cur().ev1, cur().stream));
        HIP_TRY(hipEventSynchronize(cur().ev1));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, cur().ev0, cur().ev1));
        stats->kernel_ms = ms;
        stats->primary = npix;
        if (count_work) {
            unsigned long long c[8];
            HIP_TRY(hipMemcpy(c, cur().d_counters, sizeof(c), hipMemcpyDeviceToHost));
            stats->nodes_visited = c[0]; stats->prims_tested = c[1]; stats->shadow = c[2]; stats->secondary = c[3];
            stats->box_passes = c[4]; stats->shape_passes = c[5];
        }
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return rte::last_error(); }

int rti_bound_device(void) { return g.inited ? g.device : -1; }

int rt_init(int device) {
    if (g.inited) {
        if (g.device == device) return 0;
        return fail("rt_init: already bound to another device; call rt_shutdown first");
    }
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail("rt_init: no such HIP device");
    HIP_TRY(hipSetDevice(device));
    for (Lane& l : g.lanes) {
        HIP_TRY(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreate(&l.ev0));
        HIP_TRY(hipEventCreate(&l.ev1));
        HIP_TRY(hipMalloc((void**)&l.d_counters, 8 * sizeof(unsigned long long)));
    }
    HIP_TRY(rtk_noalloc_setup(device));
    g.lane = 0;
    g.device = device; g.inited = true;
    return 0;
}

int rt_upload_scene(const rt_scene* scene) {
    if (!g.inited) return fail("rt_init has not been called");
    if (!scene) return fail("rt_scene is NULL");
    rtu::NoInitVector<uint8_t> blob; rt_bvh_info info{}; std::string err;
    if (!rtp::pack_scene(*scene, &blob, &info, &err)) return fail(err);
    rtl::BlobHeader h; std::memcpy(&h, blob.data(), sizeof(h));
    if (!rtp::validate_header(h, blob.size(), &err)) return fail(err);
    HIP_TRY(hipSetDevice(g.device));
    if (sync_lanes()) return 1;
    release_scene();
    HIP_TRY(hipMalloc(&g.d_blob, blob.size()));
    g.blob_owned = true; g.blob_bytes = blob.size();
    HIP_TRY(hipMemcpyAsync(g.d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice, cur().stream));
    HIP_TRY(hipStreamSynchronize(cur().stream));
    point_at_blob(g.d_blob, h);
    g.info = info;
    return 0;
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
    in.finite_shading = parts.finite_shading; in.pad_abs = parts.pad_abs; in.depth_limit = rtp::TREE_DEPTH_LIMIT;
    HIP_TRY(hipSetDevice(g.device));
    if (sync_lanes()) return 1;
    rtg::BuiltScene built{};
    rtg::BuildStatus st = rtg::build_scene_on_gpu(in, cur().stream, &built, &err);
    if (st == rtg::BUILD_TOO_DEEP) return rt_upload_scene(scene);                       // the host builder bounds the depth
    if (st != rtg::BUILD_OK) return fail(err);
    if (!rtp::validate_header(built.header, built.bytes, &err)) { (void)hipFree(built.d_blob); return fail(err); }
    release_scene();
    g.d_blob = built.d_blob; g.blob_bytes = built.bytes; g.blob_owned = true;
    g.info = rt_bvh_info{};
    point_at_blob(g.d_blob, built.header);
    g.info.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

int rt_upload_scene_blob(const void* host_blob, size_t bytes) {
    if (!g.inited) return fail("rt_init has not been called");
    std::string err;
    if (!rtp::validate_blob((const uint8_t*)host_blob, bytes, &err)) return fail(err);
    rtl::BlobHeader h; std::memcpy(&h, host_blob, sizeof(h));
    HIP_TRY(hipSetDevice(g.device));
    if (sync_lanes()) return 1;
    release_scene();
    HIP_TRY(hipMalloc(&g.d_blob, bytes));
    g.blob_owned = true; g.blob_bytes = bytes;
    HIP_TRY(hipMemcpyAsync(g.d_blob, host_blob, bytes, hipMemcpyHostToDevice, cur().stream));
    HIP_TRY(hipStreamSynchronize(cur().stream));
    g.info = rt_bvh_info{};
    point_at_blob(g.d_blob, h);
    return 0;
}

int rt_download_scene_blob(void* host_dst, size_t bytes) {
    if (!g.have_scene) return fail("no scene uploaded");
    if (!host_dst || bytes != g.blob_bytes) return fail("rt_download_scene_blob: destination must be exactly the blob size");
    HIP_TRY(hipSetDevice(g.device));
    HIP_TRY(hipMemcpyAsync(host_dst, g.d_blob, bytes, hipMemcpyDeviceToHost, cur().stream));
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
    *device_ptr = g.d_blob; *bytes = g.blob_bytes;
    return 0;
}

int rt_copy_scene_blob(void* device_dst, size_t bytes) {
    if (!g.have_scene) return fail("no scene uploaded");
    if (!device_dst || bytes != g.blob_bytes) return fail("rt_copy_scene_blob: destination must be exactly the blob size");
    HIP_TRY(hipSetDevice(g.device));
    HIP_TRY(hipMemcpyAsync(device_dst, g.d_blob, bytes, hipMemcpyDeviceToDevice, cur().stream));
    HIP_TRY(hipStreamSynchronize(cur().stream));
    return 0;
}

int rt_adopt_scene_blob(void* device_ptr, size_t bytes) {
    if (!g.inited) return fail("rt_init has not been called");
    if (!device_ptr || bytes < sizeof(rtl::BlobHeader)) return fail("rt_adopt_scene_blob: buffer too small");
    HIP_TRY(hipSetDevice(g.device));
    if (sync_lanes()) return 1;
    rtl::BlobHeader h;
    HIP_TRY(hipMemcpy(&h, device_ptr, sizeof(h), hipMemcpyDeviceToHost));
    std::string err;
    if (!rtp::validate_header(h, bytes, &err)) return fail(err);
    release_scene();
    g.d_blob = device_ptr; g.blob_bytes = bytes; g.blob_owned = false;
    point_at_blob(device_ptr, h);
    g.info.build_ms = 0.0;
    return 0;
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
    if (g.inited) HIP_TRY(hipSetDevice(g.device));
    return render_shard(frame, shard, nshards, tile_rows, d_out, stats, count_work);
}

int rt_render(const rt_frame* frame, uint32_t* out_pixels, rt_stats* stats, int count_work) {
    auto t0 = std::chrono::steady_clock::now();
    if (!g.inited) return fail("rt_init has not been called");
    if (check_frame(frame)) return 1;
    if (!out_pixels) return fail("output buffer is NULL");
    HIP_TRY(hipSetDevice(g.device));
    size_t npix = (size_t)frame->width * frame->height;
    if (npix > g.out_cap) {
        if (g.d_out) { (void)hipFree(g.d_out); g.d_out = nullptr; g.out_cap = 0; }
        HIP_TRY(hipMalloc((void**)&g.d_out, npix * sizeof(uint32_t)));
        g.out_cap = npix;
    }
    if (npix > g.stage_cap) {
        if (g.h_stage) { (void)hipHostFree(g.h_stage); g.h_stage = nullptr; g.stage_cap = 0; }
        HIP_TRY(hipHostMalloc((void**)&g.h_stage, npix * sizeof(uint32_t), hipHostMallocDefault));
        g.stage_cap = npix;
    }
    rt_stats local{};
    if (render_shard(frame, 0, 1, 8, g.d_out, &local, count_work)) return 1;
    const size_t n_pieces = npix >= (1u << 20) ? 8 : 1, per_piece = (npix + n_pieces - 1) / n_pieces;
    for (size_t k = 0; k < n_pieces; k++) {
        const size_t a = std::min(npix, k * per_piece), b = std::min(npix, a + per_piece);
        if (!g.piece_done[k]) HIP_TRY(hipEventCreateWithFlags(&g.piece_done[k], hipEventDisableTiming));
        if (b > a) HIP_TRY(hipMemcpyAsync(g.h_stage + a, g.d_out + a, (b - a) * sizeof(uint32_t), hipMemcpyDeviceToHost, cur().stream));
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
    HIP_TRY(hipSetDevice(g.device));
    HIP_TRY(rtk_launch_assemble(d_gathered, rows_per_shard_stride, nshards, tile_rows, width, height, d_frame, cur().stream));
    return 0;
}

int rt_compare_frames(const uint32_t* d_a, uint32_t width_a, uint32_t height_a, const uint32_t* d_b, uint32_t width_b, uint32_t height_b,
                      uint32_t* d_diff, uint64_t* count) {
    if (!g.inited) return fail("rt_init has not been called");
    if (height_a != height_b || width_a != width_b) return fail("Images have different dimensions");            // mod.rs:55-57
    if (!d_a || !d_b) return fail("rt_compare_frames: NULL image");
    if (((uintptr_t)d_a | (uintptr_t)d_b | (uintptr_t)d_diff) & 15u) return fail("rt_compare_frames: buffers must be 16-byte aligned");
    HIP_TRY(hipSetDevice(g.device));
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
    HIP_TRY(hipSetDevice(g.device));
    HIP_TRY(rtk_launch_pack_rgb(d_pixels, (size_t)width * height, d_rgb, cur().stream));
    return 0;
}

int rt_select_lane(int lane) {
    if (!g.inited) return fail("rt_init has not been called");
    if (lane < 0 || lane >= N_LANES) return fail("rt_select_lane: the lanes are 0 .. " + std::to_string(N_LANES - 1));
    g.lane = lane;
    return 0;
}

void* rt_stream(void) { return (void*)cur().stream; }

int rt_synchronize(void) {
    if (!g.inited) return fail("rt_init has not been called");
    HIP_TRY(hipStreamSynchronize(cur().stream));
    return 0;
}

void rt_shutdown(void) {
    if (!g.inited) return;
    (void)hipSetDevice(g.device);
    (void)sync_lanes();
    release_scene();
    if (g.d_out) (void)hipFree(g.d_out);
    if (g.h_stage) (void)hipHostFree(g.h_stage);
    for (hipEvent_t e : g.piece_done) if (e) (void)hipEventDestroy(e);
    for (Lane& l : g.lanes) {
        if (l.d_fold) (void)hipFree(l.d_fold);
        if (l.d_counters) (void)hipFree(l.d_counters);
        if (l.d_region_order) (void)hipFree(l.d_region_order);
        if (l.d_region_cost) (void)hipFree(l.d_region_cost);
        if (l.ev0) (void)hipEventDestroy(l.ev0);
        if (l.ev1) (void)hipEventDestroy(l.ev1);
        if (l.stream) (void)hipStreamDestroy(l.stream);
    }
    g = State();
}

}  // extern "C"
