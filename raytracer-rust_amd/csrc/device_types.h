// device_types.h -- kernel argument blocks, shared by render_kernels.hip and rt_api.cpp.
#ifndef RT_DEVICE_TYPES_H
#define RT_DEVICE_TYPES_H

#include <hip/hip_runtime.h>
#include <stdint.h>

// Pointers into the resident scene blob (layout.h).
struct DevScene {
    const uint4* nodes;     // 2 uint4 per node pair: {lo.x|hi.x<<16, lo.y|hi.y<<16, lo.z|hi.z<<16, ref} per child
    const float4* tree;     // 3 float4 per tree shape, leaves contiguous
    const float4* list;     // 3 float4 per always-tested shape
    const float4* mats;     // 2 float4 per material
    const float4* lights;   // 2 float4 per light
    uint32_t n_list, n_lights, root_ref;
    uint32_t finite_shading;   // 1 when every material and light value is finite (lets zero contributions be skipped)
    uint32_t tree_kinds;       // bit k set: the tree holds shapes of kind k (layout.h); only triangles selects the kernel built for that
    float grid_origin[3], grid_scale[3];   // node box coordinate q on axis k = grid_origin[k] + q * grid_scale[k]
    float inv_limit[3];                    // bound on |1 / direction| per axis that keeps grid_scale / direction and its multiples finite
};

// rt_frame plus the shard this launch renders.
struct DevFrame {
    float pos[3], cv[3], right[3], up[3];
    float pw, ph, w2, h2;
    uint32_t width, height, maxdepth;
    float ambient[3];
    uint32_t shard, nshards, tile_rows, local_rows;
    // Pixel-tile order of the launch.  region_log2 == 0: block b renders tile b in row-major order.
    // Otherwise the frame is cut into square regions of (1 << region_log2)^2 tiles and the blocks
    // that share an XCD (b, b + 8, b + 16, ...) walk through one region after another.
    uint32_t region_log2, regions_x, n_regions;
    // Order in which the regions are handed out (n_regions entries, device memory; nullptr = region i at position i)
    // and, per region, the run time of its slowest wavefront in this launch (10 ns units; nullptr = not recorded),
    // from which the order of the next launch of the same frame geometry is made.
    const uint32_t* region_order;
    uint32_t* region_cost;
};


constexpr uint32_t MAX_SORTED_REGIONS = 4096;
#define RT_N_COUNTERS 15                         // 64-bit work counters of an instrumented launch (render_kernels.hip, rt_api.cpp)

// Region size for a launch of tiles_x x tiles_y pixel tiles: 0 = row-major tile order.
// Frames of fewer than 2048 tiles keep row-major order; otherwise the smallest regions (from 8x8 tiles up) that
// number at most MAX_SORTED_REGIONS, so that the order can follow measured costs as finely as possible.
inline uint32_t region_log2_for(uint64_t tiles) {
    if (tiles < (1u << 11)) return 0u;
    uint32_t l = 3;
    while ((tiles >> (2 * l)) > 3072u && l < 12u) l++;     // leaves room for the partial regions at the frame's edges
    return l;
}

// The render kernel runs at RT_WAVES wavefronts per SIMD: 5, i.e. 96 vector registers (a few bytes per lane spilled
// outside the inner loops) and 7.5 KiB of LDS per wavefront.  The kernel is bound by vector ALU issue and by the chain of
// dependent traversal steps of each wavefront; a fifth wavefront hides more of the latter than its spills cost (4K dragon
// 3.24 against 3.40 ms at four, 8K mesh 37 against 41 ms), a sixth (80 registers) spills inside the loops and loses.
// LDS of one wavefront: a traversal stack of one entry per tree level plus two sentinels (256 bytes per entry), and behind
// it the records of the first reflection levels (1 KiB per level), as many as fit; the records of deeper levels, which
// fewer pixels reach, go to a global array.  Last come the 512 bytes of tables of the specular power (specular_pow.h).
#define RT_POW_LDS_BYTES 512u
#ifndef RT_WAVES
#define RT_WAVES 5
#endif
// An even share of the 160 KiB of a compute unit less half a KiB: with 8 192 bytes per wavefront only four fit a SIMD
// (measured: 7 680 bytes run at five, 8 192 at four).
constexpr uint32_t RT_LDS_PER_WAVE = ((160u * 1024u / (4u * RT_WAVES)) & ~255u) - 512u;
struct RenderPlan { uint32_t lds_levels; };          // reflection levels 0 .. lds_levels - 1 keep their records in LDS
inline uint32_t rtk_stack_entries(int stack_depth) { return (uint32_t)(stack_depth < 6 ? 8 : stack_depth + 2); }     // two sentinels
inline RenderPlan rtk_render_plan(int stack_depth, uint32_t maxdepth) {
    RenderPlan p;
    const uint32_t levels = maxdepth > 1 ? maxdepth - 1u : 0u, stack_bytes = rtk_stack_entries(stack_depth) * 256u + RT_POW_LDS_BYTES;
    const uint32_t room = stack_bytes < RT_LDS_PER_WAVE ? (RT_LDS_PER_WAVE - stack_bytes) / 1024u : 0u;
    p.lds_levels = levels < room ? levels : room;
#ifdef RT_FOLD_GLOBAL
    p.lds_levels = 0;
#endif
    return p;
}

// regions[order[i]] by decreasing cost into `order` (n <= MAX_SORTED_REGIONS), ties by index; the costs are copied to d_last,
// d_cost is left as it is and d_zeroed (n words, another array than d_cost) is cleared for the next launch to measure into.
extern "C" hipError_t rtk_launch_sort_regions(const uint32_t* d_cost, uint32_t n, uint32_t* d_order, uint32_t* d_last, uint32_t* d_zeroed, hipStream_t stream);

// Writes a guess at every region's cost to d_cost (n_regions words) from where the tree's shapes are on screen; for the
// first launch of a frame geometry.  f: as for the launch, region fields filled in.  d_seen: n_regions words, all zero
// (they come back holding counts).
extern "C" hipError_t rtk_launch_predict_regions(const DevScene* sc, const DevFrame* f, uint32_t n_tree_shapes, uint32_t* d_seen, uint32_t* d_cost, hipStream_t stream);

extern "C" hipError_t rtk_launch_render(const DevScene* sc, const DevFrame* f, uint32_t* d_out, float4* d_fold,
                                        unsigned long long* d_counters, int stack_depth, RenderPlan plan, int count, hipStream_t stream);
extern "C" hipError_t rtk_launch_assemble(const uint32_t* d_gathered, uint32_t stride_rows, uint32_t nshards, uint32_t tile_rows,
                                          uint32_t width, uint32_t height, uint32_t* d_frame, hipStream_t stream);
// image_kernels.hip; pointers must be 16-byte aligned (d_rgb: 4-byte).
extern "C" hipError_t rtk_launch_compare(const uint32_t* d_a, const uint32_t* d_b, size_t n, uint32_t* d_diff, unsigned long long* d_count,
                                         hipStream_t stream);
// The scratch array of the entry point that allocates nothing (image_kernels.hip): looked up once per device, at rt_init.
extern "C" hipError_t rtk_noalloc_setup(int device);
// rt_api.cpp: the device the library is bound to, -1 before rt_init.
extern "C" int rti_bound_device(void);
extern "C" hipError_t rtk_launch_pack_rgb(const uint32_t* d_pixels, size_t n, uint8_t* d_rgb, hipStream_t stream);

#endif
