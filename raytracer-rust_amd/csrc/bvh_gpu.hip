// bvh_gpu.hip -- the tree of RayTracer::new built on the GPU.
//
// The reference builds its bounding volume hierarchy on the CPU before it starts the clock
// (Bvh::build_par, src/raytracer/raytracer.rs:32; main.rs:11-15).  The host builder of this
// library (bvh_build.cpp, binned SAH) does the same in about a second per five million
// triangles; this one takes milliseconds, at the price of a tree that is somewhat slower to
// traverse (Morton-order radix tree instead of a surface-area split search).  Pixels do not
// depend on the tree: it only culls (layout.h, render_kernels.hip).
//
// Steps, all on one stream, every kernel a plain data-parallel pass that reads only what earlier
// launches wrote (no inter-block hand-offs inside a launch):
//   1. boxes of the shapes (shape.rs:205-234), bounds of the boxes and of their centres
//   2. 63-bit Morton code of every centre, radix sort (hipCUB)
//   3. binary radix tree over the sorted codes (Karras 2012): node i covers a contiguous range
//   4. bottom-up in rounds: a node whose children are finished takes the union of their boxes, its
//      count, its SAH cost, whether it is better kept as one leaf (at most MAX_LEAF shapes), and
//      the height of the tree below it.  One launch per round; a tree of height h needs h rounds.
//   5. every node learns whether it survives (no ancestor became a leaf); survivors are numbered by
//      a prefix sum
//   6. survivors become 32-byte node pairs on the 16-bit scene grid, shapes are gathered into
//      leaf order, both written straight into the final scene blob
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "bvh_gpu.h"
#include "layout.h"

namespace rtg {

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint32_t LEAF_BIT = 0x80000000u;     // child reference: bit 31 set = shape at that sorted position
constexpr uint32_t DEAD = 0, ALIVE = 1, LEAF_ROOT = 2;
constexpr float COST_BOX = 1.0f, COST_SHAPE = 1.0f;   // relative cost of one box-pair step and one shape test in the render kernel

// Monotone map float -> uint32 so that unsigned min/max atomics order floats.
__device__ __forceinline__ uint32_t ordered(float f) { uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__host__ __device__ inline float unordered(uint32_t u) {
    uint32_t b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float f;
#ifdef __HIP_DEVICE_COMPILE__
    f = __uint_as_float(b);
#else
    memcpy(&f, &b, 4);
#endif
    return f;
}

struct Bounds {             // [0..2] min, [3..5] max of the shape boxes; [6..8] min, [9..11] max of their centres; [12] non-finite flag
    uint32_t v[16];
};

__global__ __launch_bounds__(256) void shape_boxes(const rtl::ShapeRec* __restrict__ rec, uint32_t n, float4* __restrict__ lo, float4* __restrict__ hi,
                                                   Bounds* __restrict__ bounds) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY}, c[3] = {0, 0, 0};
    bool bad = false;
    if (i < n) {
        const rtl::ShapeRec r = rec[i];
        const uint32_t kind = r.kind_mat & 3u;
        for (int k = 0; k < 3; k++) {
            if (kind == rtl::KIND_SPHERE) { l[k] = r.q0[k] - r.q1[0]; h[k] = r.q0[k] + r.q1[0]; }        // shape.rs:208-216
            else { l[k] = fminf(fminf(r.q0[k], r.q1[k]), r.q2[k]); h[k] = fmaxf(fmaxf(r.q0[k], r.q1[k]), r.q2[k]); }   // shape.rs:217-234
            c[k] = 0.5f * l[k] + 0.5f * h[k];
            bad = bad || !isfinite(l[k]) || !isfinite(h[k]);
        }
        lo[i] = make_float4(l[0], l[1], l[2], 0.0f);
        hi[i] = make_float4(h[0], h[1], h[2], 0.0f);
    }
    // Wave reduction, then one atomic per wave and value.
    float cl[3], ch[3];
    for (int k = 0; k < 3; k++) { cl[k] = i < n ? c[k] : INFINITY; ch[k] = i < n ? c[k] : -INFINITY; }
    for (int off = 32; off > 0; off >>= 1)
        for (int k = 0; k < 3; k++) {
            l[k] = fminf(l[k], __shfl_down(l[k], off, 64)); h[k] = fmaxf(h[k], __shfl_down(h[k], off, 64));
            cl[k] = fminf(cl[k], __shfl_down(cl[k], off, 64)); ch[k] = fmaxf(ch[k], __shfl_down(ch[k], off, 64));
        }
    const bool any_bad = __any(bad);
    if (any_bad && (threadIdx.x & 63u) == 0) atomicOr(&bounds->v[12], 1u);
    if ((threadIdx.x & 63u) == 0 && !any_bad)
        for (int k = 0; k < 3; k++) {
            atomicMin(&bounds->v[k], ordered(l[k])); atomicMax(&bounds->v[3 + k], ordered(h[k]));
            atomicMin(&bounds->v[6 + k], ordered(cl[k])); atomicMax(&bounds->v[9 + k], ordered(ch[k]));
        }
}

__device__ __forceinline__ uint64_t spread21(uint64_t v) {   // 21 bits -> every third bit
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x001F00000000FFFFull;
    v = (v | (v << 16)) & 0x001F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__global__ __launch_bounds__(256) void morton_codes(const float4* __restrict__ lo, const float4* __restrict__ hi, uint32_t n,
                                                    const Bounds* __restrict__ bounds, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 l = lo[i], h = hi[i];
    const float c[3] = {0.5f * l.x + 0.5f * h.x, 0.5f * l.y + 0.5f * h.y, 0.5f * l.z + 0.5f * h.z};
    // Cubic cells: one scale for the three axes, so that a flat scene is not split across its thin side
    // as often as along the others.
    float ext = 0.0f;
    for (int k = 0; k < 3; k++) ext = fmaxf(ext, unordered(bounds->v[9 + k]) - unordered(bounds->v[6 + k]));
    uint64_t q[3];
    for (int k = 0; k < 3; k++) {
        const float cmin = unordered(bounds->v[6 + k]);
        float u = ext > 0.0f ? (c[k] - cmin) / ext : 0.0f;
        u = fminf(fmaxf(u, 0.0f), 1.0f);
        q[k] = (uint64_t)fminf(u * 2097152.0f, 2097151.0f);
    }
    keys[i] = spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2);
    vals[i] = i;
}

__global__ __launch_bounds__(256) void gather_boxes(const float4* __restrict__ lo, const float4* __restrict__ hi, const uint32_t* __restrict__ order, uint32_t n,
                                                    float4* __restrict__ slo, float4* __restrict__ shi) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t i = order[p];
    slo[p] = lo[i]; shi[p] = hi[i];
}

// Length of the common prefix of the keys at sorted positions i and j (equal keys: continue with the positions themselves).
__device__ __forceinline__ int common_prefix(const uint64_t* __restrict__ keys, uint32_t n, int i, int j) {
    if (j < 0 || j >= (int)n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

struct Topology {
    uint32_t* left; uint32_t* right;       // children of internal node i (LEAF_BIT | sorted position, or internal index)
    uint32_t* parent;                      // parent of internal node i (NONE for the root, node 0)
    uint32_t* first; uint32_t* last;       // sorted positions covered by internal node i
};

__global__ __launch_bounds__(256) void radix_tree(const uint64_t* __restrict__ keys, uint32_t n, Topology t) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= (int)n - 1) return;
    // Direction of the range and its other end (Karras 2012, section 3).
    const int d = common_prefix(keys, n, i, i + 1) - common_prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = common_prefix(keys, n, i, i - d);
    int lmax = 2;
    while (common_prefix(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int s = lmax >> 1; s >= 1; s >>= 1)
        if (common_prefix(keys, n, i, i + (l + s) * d) > dmin) l += s;
    const int j = i + l * d;
    // Split position: the highest differing bit inside the range.
    const int dnode = common_prefix(keys, n, i, j);
    int sp = 0;
    for (int s = (l + 1) >> 1;; s = (s + 1) >> 1) {
        if (common_prefix(keys, n, i, i + (sp + s) * d) > dnode) sp += s;
        if (s <= 1) break;
    }
    const int gamma = i + sp * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const uint32_t lc = lo == gamma ? (LEAF_BIT | (uint32_t)gamma) : (uint32_t)gamma;
    const uint32_t rc = hi == gamma + 1 ? (LEAF_BIT | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    t.left[i] = lc; t.right[i] = rc; t.first[i] = (uint32_t)lo; t.last[i] = (uint32_t)hi;
    if (!(lc & LEAF_BIT)) t.parent[lc] = (uint32_t)i;
    if (!(rc & LEAF_BIT)) t.parent[rc] = (uint32_t)i;
    if (i == 0) t.parent[0] = NONE;
}

struct NodeData {
    float4* lo; float4* hi;                // box of internal node i (union of the shape boxes below it, not padded)
    float* cost;                           // SAH cost of the best subtree below i, times its area
    uint32_t* height;                      // node pairs on the longest path below and including i (0 when i becomes a leaf)
    uint32_t* leafable;                    // 1 when i is better kept as one leaf
    uint32_t* done;                        // round in which i was finished (0 = not yet)
};

__device__ __forceinline__ float half_area(float4 l, float4 h) {
    const float dx = h.x - l.x, dy = h.y - l.y, dz = h.z - l.z;
    return dx * dy + dy * dz + dz * dx;
}

// One round of the bottom-up pass: finishes every node whose children were finished in earlier rounds.
__global__ __launch_bounds__(256) void fit_round(uint32_t n, uint32_t round, Topology t, NodeData d, const float4* __restrict__ slo, const float4* __restrict__ shi,
                                                 uint32_t* __restrict__ remaining) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool finished = false;
    if (i < n - 1 && d.done[i] == 0) {
        const uint32_t c[2] = {t.left[i], t.right[i]};
        float4 l[2], h[2]; float cost[2]; uint32_t height[2];
        bool ready = true;
        for (int k = 0; k < 2; k++) {
            if (c[k] & LEAF_BIT) {
                const uint32_t p = c[k] & ~LEAF_BIT;
                l[k] = slo[p]; h[k] = shi[p];
                cost[k] = COST_SHAPE * half_area(l[k], h[k]); height[k] = 0;
            } else {
                const uint32_t r = d.done[c[k]];
                if (r == 0 || r >= round) { ready = false; l[k] = h[k] = make_float4(0, 0, 0, 0); cost[k] = 0.0f; height[k] = 0; }   // not finished before this launch
                else { l[k] = d.lo[c[k]]; h[k] = d.hi[c[k]]; cost[k] = d.cost[c[k]]; height[k] = d.height[c[k]]; }
            }
        }
        if (ready) {
            const float4 lo = make_float4(fminf(l[0].x, l[1].x), fminf(l[0].y, l[1].y), fminf(l[0].z, l[1].z), 0.0f);
            const float4 hi = make_float4(fmaxf(h[0].x, h[1].x), fmaxf(h[0].y, h[1].y), fmaxf(h[0].z, h[1].z), 0.0f);
            const uint32_t count = t.last[i] - t.first[i] + 1u;
            const float area = half_area(lo, hi);
            const float split_cost = COST_BOX * area + cost[0] + cost[1];
            const float leaf_cost = COST_SHAPE * (float)count * area;
            const bool leafable = count <= rtl::MAX_LEAF && leaf_cost <= split_cost;
            d.lo[i] = lo; d.hi[i] = hi;
            d.cost[i] = leafable ? leaf_cost : split_cost;
            d.height[i] = leafable ? 0u : 1u + max(height[0], height[1]);
            d.leafable[i] = leafable ? 1u : 0u;
            d.done[i] = round;
            finished = true;
        }
    }
    // One update of the counter per wavefront.
    const unsigned long long m = __ballot(finished);
    if (m && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m)) atomicSub(remaining, (uint32_t)__popcll(m));
}

// A node survives as a node pair unless it or an ancestor became a leaf.  Only nodes with at most
// MAX_LEAF shapes can become leaves, so the walk up stops as soon as an ancestor is larger.
__global__ __launch_bounds__(256) void classify(uint32_t n, Topology t, const uint32_t* __restrict__ leafable, uint32_t* __restrict__ state,
                                                uint32_t* __restrict__ alive_flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    uint32_t s = leafable[i] ? LEAF_ROOT : ALIVE;
    if (t.last[i] - t.first[i] + 1u <= rtl::MAX_LEAF) {
        for (uint32_t a = t.parent[i]; a != NONE && t.last[a] - t.first[a] + 1u <= rtl::MAX_LEAF; a = t.parent[a])
            if (leafable[a]) { s = DEAD; break; }
    }
    state[i] = s;
    alive_flag[i] = s == ALIVE ? 1u : 0u;
}

struct Grid { float origin[3], scale[3], pad_abs; };

// Outward padding as the host builder applies it, then outward rounding onto the 16-bit grid with one
// extra step of margin for the rounding of this very computation.
__device__ __forceinline__ void quantise_box(const Grid& g, float4 lo, float4 hi, rtl::QuantChild* out) {
    const float l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
    for (int k = 0; k < 3; k++) {
        const float pad = fmaxf(fabsf(l[k]), fabsf(h[k])) * 4e-7f + g.pad_abs;
        const float ql = floorf((l[k] - pad - g.origin[k]) / g.scale[k]) - 1.0f;
        const float qh = ceilf((h[k] + pad - g.origin[k]) / g.scale[k]) + 1.0f;
        out->lo[k] = (uint16_t)fminf(fmaxf(ql, 0.0f), rtl::GRID_STEPS);
        out->hi[k] = (uint16_t)fminf(fmaxf(qh, 0.0f), rtl::GRID_STEPS);
    }
}

struct EmitStats { uint32_t n_leaves, max_leaf; };

__global__ __launch_bounds__(256) void emit_pairs(uint32_t n, Topology t, NodeData d, const uint32_t* __restrict__ state, const uint32_t* __restrict__ pair_index,
                                                  const float4* __restrict__ slo, const float4* __restrict__ shi, Grid g, rtl::QuantPair* __restrict__ pairs,
                                                  EmitStats* __restrict__ stats) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1 || state[i] != ALIVE) return;
    rtl::QuantPair out;
    const uint32_t c[2] = {t.left[i], t.right[i]};
    for (int k = 0; k < 2; k++) {
        float4 lo, hi; uint32_t ref, leaf_count = 0;
        if (c[k] & LEAF_BIT) {
            const uint32_t p = c[k] & ~LEAF_BIT;
            lo = slo[p]; hi = shi[p]; ref = rtl::REF_LEAF | (p << 3); leaf_count = 1;
        } else {
            lo = d.lo[c[k]]; hi = d.hi[c[k]];
            if (state[c[k]] == ALIVE) ref = pair_index[c[k]];
            else { leaf_count = t.last[c[k]] - t.first[c[k]] + 1u; ref = rtl::REF_LEAF | (t.first[c[k]] << 3) | (leaf_count - 1u); }
        }
        quantise_box(g, lo, hi, &out.c[k]);
        out.c[k].ref = ref;
        if (leaf_count) { atomicAdd(&stats->n_leaves, 1u); atomicMax(&stats->max_leaf, leaf_count); }
    }
    pairs[pair_index[i]] = out;
}

__global__ __launch_bounds__(256) void gather_records(const rtl::ShapeRec* __restrict__ rec, const uint32_t* __restrict__ order, uint32_t n,
                                                      rtl::ShapeRec* __restrict__ out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    // 48-byte records as three 16-byte words
    const float4* src = (const float4*)(rec + order[p]);
    float4* dst = (float4*)(out + p);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct DeviceBuffers {               // everything the build needs besides the blob, freed when the build ends
    std::vector<void*> ptrs;
    ~DeviceBuffers() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T> hipError_t alloc(T** p, size_t count) {
        hipError_t e = hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
};

#define GPU_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) { *err = std::string(#expr) + ": " + hipGetErrorString(e_); return BUILD_FAILED; } \
    } while (0)

}  // namespace

BuildStatus build_scene_on_gpu(const SceneParts& in, hipStream_t stream, BuiltScene* out, std::string* err) {
    const uint32_t n = in.n_tree;
    if (n < 2 || n > rtl::MAX_TREE_SHAPES) { *err = "GPU tree build needs between 2 and 2^27 shapes"; return BUILD_FAILED; }
    const uint32_t blocks_n = (n + 255) / 256, blocks_i = (n - 1 + 255) / 256;
    DeviceBuffers mem;
    rtl::ShapeRec* d_rec; float4 *d_lo, *d_hi, *d_slo, *d_shi; Bounds* d_bounds;
    uint64_t *d_keys, *d_keys_sorted; uint32_t *d_vals, *d_order;
    GPU_TRY(mem.alloc(&d_rec, n)); GPU_TRY(mem.alloc(&d_lo, n)); GPU_TRY(mem.alloc(&d_hi, n)); GPU_TRY(mem.alloc(&d_slo, n)); GPU_TRY(mem.alloc(&d_shi, n));
    GPU_TRY(mem.alloc(&d_bounds, 1)); GPU_TRY(mem.alloc(&d_keys, n)); GPU_TRY(mem.alloc(&d_keys_sorted, n)); GPU_TRY(mem.alloc(&d_vals, n)); GPU_TRY(mem.alloc(&d_order, n));
    GPU_TRY(hipMemcpyAsync(d_rec, in.tree, (size_t)n * sizeof(rtl::ShapeRec), hipMemcpyHostToDevice, stream));

    // 1. boxes and bounds
    Bounds init;
    for (int k = 0; k < 16; k++) init.v[k] = 0;
    for (int k = 0; k < 3; k++) { init.v[k] = init.v[6 + k] = 0xFFFFFFFFu; init.v[3 + k] = init.v[9 + k] = 0u; }
    GPU_TRY(hipMemcpyAsync(d_bounds, &init, sizeof(init), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(shape_boxes, dim3(blocks_n), dim3(256), 0, stream, d_rec, n, d_lo, d_hi, d_bounds);
    GPU_TRY(hipGetLastError());
    // 2. Morton codes, sort
    hipLaunchKernelGGL(morton_codes, dim3(blocks_n), dim3(256), 0, stream, d_lo, d_hi, n, d_bounds, d_keys, d_vals);
    GPU_TRY(hipGetLastError());
    size_t sort_bytes = 0;
    GPU_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, d_keys, d_keys_sorted, d_vals, d_order, (int)n, 0, 63, stream));
    uint8_t* d_sort_temp; GPU_TRY(mem.alloc(&d_sort_temp, sort_bytes));
    GPU_TRY(hipcub::DeviceRadixSort::SortPairs(d_sort_temp, sort_bytes, d_keys, d_keys_sorted, d_vals, d_order, (int)n, 0, 63, stream));
    hipLaunchKernelGGL(gather_boxes, dim3(blocks_n), dim3(256), 0, stream, d_lo, d_hi, d_order, n, d_slo, d_shi);
    GPU_TRY(hipGetLastError());
    Bounds bounds;
    GPU_TRY(hipMemcpyAsync(&bounds, d_bounds, sizeof(bounds), hipMemcpyDeviceToHost, stream));
    GPU_TRY(hipStreamSynchronize(stream));
    if (bounds.v[12]) { *err = "rt_scene: non-finite shape coordinates"; return BUILD_FAILED; }

    // 3. radix tree
    Topology t; NodeData d;
    GPU_TRY(mem.alloc(&t.left, n)); GPU_TRY(mem.alloc(&t.right, n)); GPU_TRY(mem.alloc(&t.parent, n)); GPU_TRY(mem.alloc(&t.first, n)); GPU_TRY(mem.alloc(&t.last, n));
    GPU_TRY(mem.alloc(&d.lo, n)); GPU_TRY(mem.alloc(&d.hi, n)); GPU_TRY(mem.alloc(&d.cost, n)); GPU_TRY(mem.alloc(&d.height, n));
    GPU_TRY(mem.alloc(&d.leafable, n)); GPU_TRY(mem.alloc(&d.done, n));
    hipLaunchKernelGGL(radix_tree, dim3(blocks_i), dim3(256), 0, stream, d_keys_sorted, n, t);
    GPU_TRY(hipGetLastError());

    // 4. bottom-up rounds; the count of unfinished nodes comes back every few rounds
    uint32_t* d_remaining; GPU_TRY(mem.alloc(&d_remaining, 1));
    uint32_t remaining = n - 1;
    GPU_TRY(hipMemsetAsync(d.done, 0, (size_t)n * sizeof(uint32_t), stream));
    GPU_TRY(hipMemcpyAsync(d_remaining, &remaining, sizeof(remaining), hipMemcpyHostToDevice, stream));
    const uint32_t MAX_ROUNDS = 160;      // height of the radix tree is at most 63 + 28 key and position bits
    uint32_t round = 1;
    while (remaining != 0 && round <= MAX_ROUNDS) {
        for (int k = 0; k < 8; k++, round++) {
            hipLaunchKernelGGL(fit_round, dim3(blocks_i), dim3(256), 0, stream, n, round, t, d, d_slo, d_shi, d_remaining);
            GPU_TRY(hipGetLastError());
        }
        GPU_TRY(hipMemcpyAsync(&remaining, d_remaining, sizeof(remaining), hipMemcpyDeviceToHost, stream));
        GPU_TRY(hipStreamSynchronize(stream));
    }
    if (remaining != 0) { *err = "internal: the bottom-up pass of the GPU tree build did not finish"; return BUILD_FAILED; }

    // 5. survivors and their numbering
    uint32_t *d_state, *d_alive, *d_pair_index;
    GPU_TRY(mem.alloc(&d_state, n)); GPU_TRY(mem.alloc(&d_alive, n)); GPU_TRY(mem.alloc(&d_pair_index, n));
    hipLaunchKernelGGL(classify, dim3(blocks_i), dim3(256), 0, stream, n, t, d.leafable, d_state, d_alive);
    GPU_TRY(hipGetLastError());
    size_t scan_bytes = 0;
    GPU_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_alive, d_pair_index, (int)(n - 1), stream));
    uint8_t* d_scan_temp; GPU_TRY(mem.alloc(&d_scan_temp, scan_bytes));
    GPU_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan_temp, scan_bytes, d_alive, d_pair_index, (int)(n - 1), stream));
    uint32_t last_index = 0, last_alive = 0, root_height = 0, root_state = 0; float root_cost = 0.0f;
    GPU_TRY(hipMemcpyAsync(&last_index, d_pair_index + (n - 2), 4, hipMemcpyDeviceToHost, stream));
    GPU_TRY(hipMemcpyAsync(&last_alive, d_alive + (n - 2), 4, hipMemcpyDeviceToHost, stream));
    GPU_TRY(hipMemcpyAsync(&root_height, d.height, 4, hipMemcpyDeviceToHost, stream));
    GPU_TRY(hipMemcpyAsync(&root_state, d_state, 4, hipMemcpyDeviceToHost, stream));
    GPU_TRY(hipMemcpyAsync(&root_cost, d.cost, 4, hipMemcpyDeviceToHost, stream));
    GPU_TRY(hipStreamSynchronize(stream));
    const uint32_t n_pairs = last_index + last_alive;
    if (root_state != ALIVE || n_pairs == 0) { *err = "internal: the root of the GPU tree is not a node pair"; return BUILD_FAILED; }
    if (root_height > in.depth_limit) return BUILD_TOO_DEEP;

    // 6. the blob
    rtl::BlobHeader h{};
    h.magic = rtl::BLOB_MAGIC; h.version = rtl::BLOB_VERSION;
    h.n_pairs = n_pairs; h.n_tree_shapes = n; h.n_list_shapes = in.n_list; h.n_materials = in.n_mats; h.n_lights = in.n_lights;
    h.root_ref = 0;                       // node 0 is the root and the first survivor
    h.max_depth = root_height; h.finite_shading = in.finite_shading;
    size_t off = sizeof(rtl::BlobHeader);
    h.off_nodes = off; off = align256(off + (size_t)n_pairs * sizeof(rtl::QuantPair));
    h.off_tree_shapes = off; off = align256(off + (size_t)n * sizeof(rtl::ShapeRec));
    h.off_list_shapes = off; off = align256(off + (size_t)in.n_list * sizeof(rtl::ShapeRec));
    h.off_materials = off; off = align256(off + (size_t)in.n_mats * sizeof(rtl::MaterialRec));
    h.off_lights = off; off = align256(off + (size_t)in.n_lights * sizeof(rtl::LightRec));
    h.total_bytes = off;
    // Grid over the padded box of all tree shapes, as the host builder lays it out.
    Grid g; g.pad_abs = in.pad_abs;
    float area = 0.0f, ext[3];
    for (int k = 0; k < 3; k++) {
        const float lo = unordered(bounds.v[k]), hi = unordered(bounds.v[3 + k]);
        ext[k] = hi - lo;
        const float pad = fmaxf(fabsf(lo), fabsf(hi)) * 4e-7f + in.pad_abs;
        g.origin[k] = nextafterf(lo - pad, -INFINITY);
        float sc = (float)(((double)(hi + pad) - (double)g.origin[k]) / (double)rtl::GRID_STEPS);
        sc = nextafterf(nextafterf(sc, INFINITY), INFINITY);
        if (!(sc > 1e-30f)) sc = 1e-30f;
        g.scale[k] = sc; h.grid_origin[k] = g.origin[k]; h.grid_scale[k] = sc;
    }
    area = ext[0] * ext[1] + ext[1] * ext[2] + ext[2] * ext[0];
    h.sah_cost = area > 0.0f ? root_cost / area : 0.0f;

    uint8_t* d_blob = nullptr;
    GPU_TRY(hipMalloc((void**)&d_blob, h.total_bytes));
    struct BlobGuard { uint8_t* p; ~BlobGuard() { if (p) (void)hipFree(p); } } guard{d_blob};
    GPU_TRY(hipMemsetAsync(d_blob, 0, h.total_bytes, stream));
    EmitStats* d_stats; GPU_TRY(mem.alloc(&d_stats, 1));
    GPU_TRY(hipMemsetAsync(d_stats, 0, sizeof(EmitStats), stream));
    hipLaunchKernelGGL(emit_pairs, dim3(blocks_i), dim3(256), 0, stream, n, t, d, d_state, d_pair_index, d_slo, d_shi, g,
                       (rtl::QuantPair*)(d_blob + h.off_nodes), d_stats);
    GPU_TRY(hipGetLastError());
    hipLaunchKernelGGL(gather_records, dim3(blocks_n), dim3(256), 0, stream, d_rec, d_order, n, (rtl::ShapeRec*)(d_blob + h.off_tree_shapes));
    GPU_TRY(hipGetLastError());
    if (in.n_list) GPU_TRY(hipMemcpyAsync(d_blob + h.off_list_shapes, in.list, (size_t)in.n_list * sizeof(rtl::ShapeRec), hipMemcpyHostToDevice, stream));
    if (in.n_mats) GPU_TRY(hipMemcpyAsync(d_blob + h.off_materials, in.mats, (size_t)in.n_mats * sizeof(rtl::MaterialRec), hipMemcpyHostToDevice, stream));
    if (in.n_lights) GPU_TRY(hipMemcpyAsync(d_blob + h.off_lights, in.lights, (size_t)in.n_lights * sizeof(rtl::LightRec), hipMemcpyHostToDevice, stream));
    EmitStats stats;
    GPU_TRY(hipMemcpyAsync(&stats, d_stats, sizeof(stats), hipMemcpyDeviceToHost, stream));
    GPU_TRY(hipStreamSynchronize(stream));
    h.n_leaves = stats.n_leaves; h.max_leaf = stats.max_leaf;
    GPU_TRY(hipMemcpyAsync(d_blob, &h, sizeof(h), hipMemcpyHostToDevice, stream));
    GPU_TRY(hipStreamSynchronize(stream));
    out->d_blob = d_blob; out->bytes = h.total_bytes; out->header = h;
    guard.p = nullptr;
    return BUILD_OK;
}

}  // namespace rtg
