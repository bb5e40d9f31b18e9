// bvh_gpu.hip -- the tree of RayTracer::new built on the GPU.
//
// The reference builds its bounding volume hierarchy on the CPU before it starts the clock
// (Bvh::build_par, src/raytracer/raytracer.rs:32; main.rs:11-15).  The host builder of this
// library (bvh_build.cpp, binned SAH) does the same in about a second per five million
// triangles; this one takes tens of milliseconds, for a tree that is a few per cent slower to
// traverse.  Pixels do not depend on the tree: it only culls (layout.h, render_kernels.hip).
//
// Steps, all on one stream, every kernel a plain data-parallel pass that reads only what earlier
// launches wrote (no inter-block hand-offs inside a launch):
//   1. boxes of the shapes (shape.rs:205-234), bounds of the boxes and of their centres
//   2. 63-bit Morton code of every centre, radix sort (hipCUB)
//   3. binary radix tree over the sorted codes (Karras 2012): node i covers a contiguous range.
//      It is used for grouping only.
//   4. groups: the largest radix nodes with at most GROUP shapes (level 1), then the largest with
//      at most GROUP groups of the level before (levels 2..), until the root is a group; each
//      level is one marking pass, one prefix sum and one listing pass over the positions, and one
//      pass for the boxes of the groups
//   5. from the top level down, one wavefront per group: binned surface-area split search over
//      the group's items in LDS, node pairs written to temporary slots on the 16-bit scene grid,
//      every item told where its own subtree has to be hooked in and how deep it sits; at level 1
//      the items are shapes, leaves hold up to MAX_LEAF of them and the shapes of a group are
//      put into leaf order
//   6. used slots numbered by a prefix sum, node pairs and shape records written in final order
//      straight into the scene blob
// Radix nodes survive as tree nodes only above the last level of groups (never, unless the
// scene has more than GROUP^LEVELS shapes, its shapes pile up on few Morton codes, or the
// caller asked for fewer levels); their boxes come from bottom-up rounds, one launch per round.
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

// A fixed number of workgroups strides over the shapes, so that the bounds take a few thousand atomics, not one set per wavefront of shapes.
constexpr uint32_t BOXES_BLOCKS = 1024;

__global__ __launch_bounds__(256) void shape_boxes(const rtl::ShapeRec* __restrict__ rec, uint32_t n, float4* __restrict__ lo, float4* __restrict__ hi,
                                                   Bounds* __restrict__ bounds) {
    float bl[3] = {INFINITY, INFINITY, INFINITY}, bh[3] = {-INFINITY, -INFINITY, -INFINITY};
    float cl[3] = {INFINITY, INFINITY, INFINITY}, ch[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool bad = false;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const rtl::ShapeRec r = rec[i];
        const uint32_t kind = r.kind_mat & 3u;
        float l[3], h[3];
        for (int k = 0; k < 3; k++) {
            if (kind == rtl::KIND_SPHERE) { l[k] = r.q0[k] - r.q1[0]; h[k] = r.q0[k] + r.q1[0]; }        // shape.rs:208-216
            else { l[k] = fminf(fminf(r.q0[k], r.q1[k]), r.q2[k]); h[k] = fmaxf(fmaxf(r.q0[k], r.q1[k]), r.q2[k]); }   // shape.rs:217-234
            const float c = 0.5f * l[k] + 0.5f * h[k];
            bad = bad || !isfinite(l[k]) || !isfinite(h[k]);
            bl[k] = fminf(bl[k], l[k]); bh[k] = fmaxf(bh[k], h[k]); cl[k] = fminf(cl[k], c); ch[k] = fmaxf(ch[k], c);
        }
        lo[i] = make_float4(l[0], l[1], l[2], 0.0f);
        hi[i] = make_float4(h[0], h[1], h[2], 0.0f);
    }
    // Wave reduction, then one atomic per wave and value.
    for (int off = 32; off > 0; off >>= 1)
        for (int k = 0; k < 3; k++) {
            bl[k] = fminf(bl[k], __shfl_down(bl[k], off, 64)); bh[k] = fmaxf(bh[k], __shfl_down(bh[k], off, 64));
            cl[k] = fminf(cl[k], __shfl_down(cl[k], off, 64)); ch[k] = fmaxf(ch[k], __shfl_down(ch[k], off, 64));
        }
    const bool any_bad = __any(bad);
    if (any_bad && (threadIdx.x & 63u) == 0) atomicOr(&bounds->v[12], 1u);
    if ((threadIdx.x & 63u) == 0 && !any_bad && bl[0] <= bh[0])
        for (int k = 0; k < 3; k++) {
            atomicMin(&bounds->v[k], ordered(bl[k])); atomicMax(&bounds->v[3 + k], ordered(bh[k]));
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
    uint32_t* leaf_parent;                 // internal node above the shape at a sorted position
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
    if (!(lc & LEAF_BIT)) t.parent[lc] = (uint32_t)i; else t.leaf_parent[lc & ~LEAF_BIT] = (uint32_t)i;
    if (!(rc & LEAF_BIT)) t.parent[rc] = (uint32_t)i; else t.leaf_parent[rc & ~LEAF_BIT] = (uint32_t)i;
    if (i == 0) t.parent[0] = NONE;
}

struct NodeData {
    float4* lo; float4* hi;                // box of internal node i (union of the shape boxes below it, not padded); only for nodes above the groups
    uint32_t* done;                        // round in which i was finished (0 = not yet)
};

__device__ __forceinline__ float half_area(float4 l, float4 h) {
    const float dx = h.x - l.x, dy = h.y - l.y, dz = h.z - l.z;
    return dx * dy + dy * dz + dz * dx;
}

struct Grid { float origin[3], scale[3], pad_abs; };

// Outward padding as the host builder applies it, then outward rounding onto the 16-bit grid with one
// extra step of margin for the rounding of this very computation.
__device__ __forceinline__ void quantise_box(const Grid& g, const float* l, const float* h, rtl::QuantChild* out) {
    for (int k = 0; k < 3; k++) {
        const float pad = fmaxf(fabsf(l[k]), fabsf(h[k])) * 4e-7f + g.pad_abs;
        const float ql = floorf((l[k] - pad - g.origin[k]) / g.scale[k]) - 1.0f;
        const float qh = ceilf((h[k] + pad - g.origin[k]) / g.scale[k]) + 1.0f;
        out->lo[k] = (uint16_t)fminf(fmaxf(ql, 0.0f), rtl::GRID_STEPS);
        out->hi[k] = (uint16_t)fminf(fmaxf(qh, 0.0f), rtl::GRID_STEPS);
    }
}

// The radix tree only groups; the tree that is rendered comes from a surface-area split search inside the groups.
// Level 1: the items are the shapes, a group is a largest radix node with at most GROUP shapes (or a single shape
// hanging directly under a larger node).  Level k + 1: the items are the groups of level k, a group is a largest
// radix node with at most GROUP of them.  The levels are built from the top down, each group handing to each of its
// items the place its subtree has to be hooked into (`link`) and the number of node pairs above it (`depth`).  What is
// left above the last level (radix nodes with more than GROUP items of it; none for scenes of up to GROUP^LEVELS
// shapes spread evenly) stays as the radix tree made it.
#ifndef RT_GROUP
#define RT_GROUP 256
#endif
constexpr uint32_t GROUP = RT_GROUP;
constexpr int LEVELS = 4;               // most levels of groups
#ifndef RT_BINS
#define RT_BINS 16
#endif
constexpr int BINS = RT_BINS;

struct ItemIndex {                 // items of one level: index of the item that holds a sorted position
    const uint32_t* incl;          // inclusive prefix sum of the item starts (item = incl[p] - 1); nullptr: every position is an item
    __device__ __forceinline__ uint32_t of(uint32_t p) const { return incl ? incl[p] - 1u : p; }
    __device__ __forceinline__ uint32_t under(const Topology& t, uint32_t i) const { return of(t.last[i]) - of(t.first[i]) + 1u; }
};

// Starts of the groups of this level = starts of the items of the next.
__global__ __launch_bounds__(256) void mark_groups(uint32_t n, Topology t, ItemIndex items, uint32_t* __restrict__ start) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n - 1) {
        const uint32_t up = t.parent[i];
        if (items.under(t, i) <= GROUP && (up == NONE || items.under(t, up) > GROUP)) start[t.first[i]] = 1u;
    }
    if (i < n && items.under(t, t.leaf_parent[i]) > GROUP) start[i] = 1u;      // a shape (then always an item by itself) directly under a larger node
}

struct ItemList {                  // items of one level above the shapes, in the order of their sorted positions
    uint32_t* first; uint32_t* last;        // sorted positions covered
    float4* lo; float4* hi;                 // box, not padded
    uint32_t* link;                         // temporary slot * 2 + side of the child reference that waits for this item's subtree (NONE: the root)
    uint32_t* depth;                        // node pairs above it
    uint32_t count;
};

__global__ __launch_bounds__(256) void list_groups(uint32_t n, Topology t, ItemIndex items, ItemIndex groups, ItemList out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n - 1) {
        const uint32_t up = t.parent[i];
        if (items.under(t, i) <= GROUP && (up == NONE || items.under(t, up) > GROUP)) {
            const uint32_t j = groups.of(t.first[i]);
            out.first[j] = t.first[i]; out.last[j] = t.last[i]; out.link[j] = NONE; out.depth[j] = 0u;
        }
    }
    if (i < n && items.under(t, t.leaf_parent[i]) > GROUP) {
        const uint32_t j = groups.of(i);
        out.first[j] = i; out.last[j] = i; out.link[j] = NONE; out.depth[j] = 0u;
    }
}

// Box of every group from the boxes of its items, one wavefront per group.
__global__ __launch_bounds__(256) void group_boxes(ItemList groups, ItemIndex index, const float4* __restrict__ ilo, const float4* __restrict__ ihi) {
    const uint32_t c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (c >= groups.count) return;
    const uint32_t a = index.of(groups.first[c]), b = index.of(groups.last[c]);
    float l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t j = a + lane; j <= b && j - a < GROUP; j += 64) {
        const float4 x = ilo[j], y = ihi[j];
        l[0] = fminf(l[0], x.x); l[1] = fminf(l[1], x.y); l[2] = fminf(l[2], x.z); h[0] = fmaxf(h[0], y.x); h[1] = fmaxf(h[1], y.y); h[2] = fmaxf(h[2], y.z);
    }
    for (int off = 32; off > 0; off >>= 1)
        for (int k = 0; k < 3; k++) { l[k] = fminf(l[k], __shfl_xor(l[k], off, 64)); h[k] = fmaxf(h[k], __shfl_xor(h[k], off, 64)); }
    if (lane == 0) { groups.lo[c] = make_float4(l[0], l[1], l[2], 0.0f); groups.hi[c] = make_float4(h[0], h[1], h[2], 0.0f); }
}

// Radix nodes above the last level of groups: how many there are, and their boxes bottom-up in rounds (one launch
// per round; a node is finished when its children were finished in earlier rounds; groups and shapes are finished
// from the start).
__global__ __launch_bounds__(256) void count_upper(uint32_t n, Topology t, ItemIndex items, uint32_t* __restrict__ remaining) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long m = __ballot(i < n - 1 && items.under(t, i) > GROUP);
    if (m && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m)) atomicAdd(remaining, (uint32_t)__popcll(m));
}

__global__ __launch_bounds__(256) void fit_upper_round(uint32_t n, uint32_t round, Topology t, NodeData d, const float4* __restrict__ slo, const float4* __restrict__ shi,
                                                       ItemIndex items, ItemIndex groups, ItemList top, uint32_t* __restrict__ remaining) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool finished = false;
    if (i < n - 1 && items.under(t, i) > GROUP && d.done[i] == 0) {
        const uint32_t c[2] = {t.left[i], t.right[i]};
        float4 l[2], h[2];
        bool ready = true;
        for (int k = 0; k < 2; k++) {
            if (c[k] & LEAF_BIT) { const uint32_t p = c[k] & ~LEAF_BIT; l[k] = slo[p]; h[k] = shi[p]; }
            else if (items.under(t, c[k]) <= GROUP) { const uint32_t j = groups.of(t.first[c[k]]); l[k] = top.lo[j]; h[k] = top.hi[j]; }
            else {
                const uint32_t r = d.done[c[k]];
                if (r == 0 || r >= round) { ready = false; l[k] = h[k] = make_float4(0, 0, 0, 0); }   // not finished before this launch
                else { l[k] = d.lo[c[k]]; h[k] = d.hi[c[k]]; }
            }
        }
        if (ready) {
            d.lo[i] = make_float4(fminf(l[0].x, l[1].x), fminf(l[0].y, l[1].y), fminf(l[0].z, l[1].z), 0.0f);
            d.hi[i] = make_float4(fmaxf(h[0].x, h[1].x), fmaxf(h[0].y, h[1].y), fmaxf(h[0].z, h[1].z), 0.0f);
            d.done[i] = round;
            finished = true;
        }
    }
    // One update of the counter per wavefront.
    const unsigned long long m = __ballot(finished);
    if (m && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m)) atomicSub(remaining, (uint32_t)__popcll(m));
}

struct BuildStats { uint32_t n_leaves, max_leaf, max_depth, root_ref; unsigned long long cost_micro; };

// Temporary node pairs.  A group whose first item has index a puts its k-th pair into slot base(level) + a + k
// (a group of m items makes at most m - 1 pairs); `used` marks the slots that hold a pair.
struct TempTree {
    rtl::QuantPair* pairs; uint32_t* used;
    uint32_t* order_out;          // sorted position -> shape index, after the level-1 groups reordered their ranges
};

__device__ __forceinline__ void hook(const TempTree& tmp, BuildStats* stats, uint32_t link, uint32_t ref) {
    if (link == NONE) stats->root_ref = ref; else tmp.pairs[link >> 1].c[link & 1u].ref = ref;
}

// Binned surface-area split search over the items of one group, one wavefront per group, everything in LDS.
// The rules of the host builder (bvh_build.cpp): cost of a split = area-weighted weights of the two sides; with shapes
// as items a node of at most MAX_LEAF of them stays a leaf when that is cheaper; halving by position (Morton order)
// when the depth budget runs out or no split separates anything.  SHAPES: level 1, leaves hold shapes and the shapes of
// the group are put into leaf order.  Otherwise a leaf is one item and receives link and depth for the level below.
template <bool SHAPES>
__global__ __launch_bounds__(64) void build_groups(ItemList groups, ItemIndex index, const float4* __restrict__ ilo, const float4* __restrict__ ihi,
                                                   const uint32_t* __restrict__ item_first, const uint32_t* __restrict__ item_last,
                                                   uint32_t* __restrict__ item_link, uint32_t* __restrict__ item_depth, uint32_t slot_base,
                                                   const uint32_t* __restrict__ order_in, Grid g, TempTree tmp, uint32_t depth_limit, uint32_t reserve,
                                                   float root_area, BuildStats* __restrict__ stats) {
    __shared__ float blo[3][GROUP], bhi[3][GROUP];
    __shared__ uint16_t perm[GROUP], scratch[GROUP];
    __shared__ uint8_t weight[SHAPES ? 1 : GROUP];
    __shared__ uint32_t bin_lo[3][BINS][3], bin_hi[3][BINS][3], bin_cnt[3][BINS];
    // Pending nodes.  The smaller child is always taken next, so every node that waits is at least as large as all
    // that is above it on the stack: at most log2(GROUP) + 1 wait at any time.
    constexpr int STACK = 40;
    __shared__ uint16_t task_start[STACK], task_count[STACK], task_depth[STACK];
    __shared__ uint32_t task_link[STACK];
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    if (c >= groups.count) return;
    const uint32_t a = index.of(groups.first[c]), m_all = index.of(groups.last[c]) - a + 1u;
    if (m_all > GROUP) return;                                        // cannot happen: a group is defined by this bound
    for (uint32_t j = lane; j < m_all; j += 64) {
        const float4 l = ilo[a + j], h = ihi[a + j];
        blo[0][j] = l.x; blo[1][j] = l.y; blo[2][j] = l.z; bhi[0][j] = h.x; bhi[1][j] = h.y; bhi[2][j] = h.z;
        perm[j] = (uint16_t)j;
        // a subtree over s shapes costs about its depth to pass through
        if (!SHAPES) weight[j] = (uint8_t)(32 - __clz((int)(item_last[a + j] - item_first[a + j] + 1u)));
    }
    if (lane == 0) { task_start[0] = 0; task_count[0] = (uint16_t)m_all; task_depth[0] = (uint16_t)min(groups.depth[c], 60000u); task_link[0] = groups.link[c]; }
    __syncthreads();
    uint32_t n_tasks = 1, next_slot = slot_base + a, leaves = 0, max_leaf = 0, max_depth = 0;
    double cost = 0.0;
    while (n_tasks > 0 && n_tasks + 2 <= (uint32_t)STACK) {
        n_tasks--;
        const uint32_t start = task_start[n_tasks], m = task_count[n_tasks], depth = task_depth[n_tasks], link = task_link[n_tasks];
        __syncthreads();
        // Box of the node and of its centres.
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t j = lane; j < m; j += 64) {
            const uint32_t s = perm[start + j];
            for (int k = 0; k < 3; k++) {
                const float l = blo[k][s], h = bhi[k][s], ctr = 0.5f * l + 0.5f * h;
                lo[k] = fminf(lo[k], l); hi[k] = fmaxf(hi[k], h); clo[k] = fminf(clo[k], ctr); chi[k] = fmaxf(chi[k], ctr);
            }
        }
        for (int off = 32; off > 0; off >>= 1)
            for (int k = 0; k < 3; k++) {
                lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
                clo[k] = fminf(clo[k], __shfl_xor(clo[k], off, 64)); chi[k] = fmaxf(chi[k], __shfl_xor(chi[k], off, 64));
            }
        const float area = (hi[0] - lo[0]) * (hi[1] - lo[1]) + (hi[1] - lo[1]) * (hi[2] - lo[2]) + (hi[2] - lo[2]) * (hi[0] - lo[0]);
        // Depth budget: halving down to leaves takes `need` more levels here and `reserve` in the levels below.
        const uint32_t leaf_items = SHAPES ? rtl::MAX_LEAF : 1u;
        uint32_t need = 0;
        while ((leaf_items << need) < m) need++;
        const bool force_median = depth + need + 1u + reserve >= depth_limit;
        bool make_leaf = m == 1u || (SHAPES && force_median && m <= rtl::MAX_LEAF);
        int axis = -1; float threshold = 0.0f, scale = 0.0f, origin = 0.0f;     // items with bin(axis) <= threshold go left
        if (!make_leaf && !force_median) {
            for (uint32_t j = lane; j < 3u * BINS; j += 64) {
                const uint32_t ax = j / BINS, b = j % BINS;
                for (int k = 0; k < 3; k++) { bin_lo[ax][b][k] = 0xFFFFFFFFu; bin_hi[ax][b][k] = 0u; }
                bin_cnt[ax][b] = 0u;
            }
            __syncthreads();
            float inv[3];
            for (int k = 0; k < 3; k++) {
                inv[k] = chi[k] > clo[k] ? (float)BINS * 0.999999f / (chi[k] - clo[k]) : 0.0f;
                if (!(inv[k] < INFINITY)) inv[k] = 0.0f;
            }
            for (uint32_t j = lane; j < m; j += 64) {
                const uint32_t s = perm[start + j];
                for (int ax = 0; ax < 3; ax++) {
                    if (inv[ax] == 0.0f) continue;
                    int b = (int)((0.5f * blo[ax][s] + 0.5f * bhi[ax][s] - clo[ax]) * inv[ax]);
                    b = min(max(b, 0), BINS - 1);
                    for (int k = 0; k < 3; k++) { atomicMin(&bin_lo[ax][b][k], ordered(blo[k][s])); atomicMax(&bin_hi[ax][b][k], ordered(bhi[k][s])); }
                    atomicAdd(&bin_cnt[ax][b], SHAPES ? 1u : (uint32_t)weight[s]);
                }
            }
            __syncthreads();
            // Candidate split after bin b of axis ax, spread over the lanes.
            float best = INFINITY; int best_id = -1;
            for (uint32_t cand = lane; cand < 3u * BINS; cand += 64) {
                const uint32_t ax = cand / BINS, b = cand % BINS;
                if (b + 1 < BINS && inv[ax] != 0.0f) {
                    float l0[3] = {INFINITY, INFINITY, INFINITY}, h0[3] = {-INFINITY, -INFINITY, -INFINITY}, l1[3] = {INFINITY, INFINITY, INFINITY}, h1[3] = {-INFINITY, -INFINITY, -INFINITY};
                    uint32_t n0 = 0, n1 = 0;
                    for (uint32_t q = 0; q < (uint32_t)BINS; q++) {
                        if (bin_cnt[ax][q] == 0u) continue;
                        for (int k = 0; k < 3; k++) {
                            const float l = unordered(bin_lo[ax][q][k]), h = unordered(bin_hi[ax][q][k]);
                            if (q <= b) { l0[k] = fminf(l0[k], l); h0[k] = fmaxf(h0[k], h); } else { l1[k] = fminf(l1[k], l); h1[k] = fmaxf(h1[k], h); }
                        }
                        if (q <= b) n0 += bin_cnt[ax][q]; else n1 += bin_cnt[ax][q];
                    }
                    if (n0 && n1) {
                        const float a0 = (h0[0] - l0[0]) * (h0[1] - l0[1]) + (h0[1] - l0[1]) * (h0[2] - l0[2]) + (h0[2] - l0[2]) * (h0[0] - l0[0]);
                        const float a1 = (h1[0] - l1[0]) * (h1[1] - l1[1]) + (h1[1] - l1[1]) * (h1[2] - l1[2]) + (h1[2] - l1[2]) * (h1[0] - l1[0]);
                        const float v = a0 * (float)n0 + a1 * (float)n1;
                        if (v < best) { best = v; best_id = (int)cand; }       // overflow or NaN: not a candidate
                    }
                }
            }
            for (int off = 32; off > 0; off >>= 1) {
                const float ob = __shfl_xor(best, off, 64); const int oi = __shfl_xor(best_id, off, 64);
                if (oi >= 0 && (best_id < 0 || ob < best || (ob == best && oi < best_id))) { best = ob; best_id = oi; }
            }
            best = __shfl(best, 0, 64); best_id = __shfl(best_id, 0, 64);  // one decision for the whole wavefront
            if (best_id >= 0) {
                const float split_cost = COST_BOX + COST_SHAPE * (area > 0.0f ? best / area : (float)m);
                if (SHAPES && m <= rtl::MAX_LEAF && COST_SHAPE * (float)m <= split_cost) make_leaf = true;
                else { axis = best_id / BINS; threshold = (float)(best_id % BINS); scale = inv[axis]; origin = clo[axis]; }
            } else if (SHAPES && m <= rtl::MAX_LEAF) {
                make_leaf = true;                                     // all centres coincide and the shapes fit one leaf
            }
        }
        if (make_leaf) {
            if (lane == 0) {
                if (SHAPES) hook(tmp, stats, link, rtl::REF_LEAF | ((a + start) << 3) | (m - 1u));
                else { item_link[a + perm[start]] = link; item_depth[a + perm[start]] = depth; }
            }
            if (SHAPES) {
                leaves++; max_leaf = max(max_leaf, m); max_depth = max(max_depth, depth);
                cost += (double)COST_SHAPE * (double)m * (double)area;
            }
            continue;
        }
        // Partition perm[start, start + m): by bin on the chosen axis, or in halves by position.
        uint32_t nl = 0;
        if (axis >= 0) {
            for (uint32_t base = 0; base < m; base += 64) {
                const uint32_t j = base + lane;
                bool left = false;
                if (j < m) {
                    const uint32_t s = perm[start + j];
                    int b = (int)((0.5f * blo[axis][s] + 0.5f * bhi[axis][s] - origin) * scale);
                    b = min(max(b, 0), BINS - 1);
                    left = (float)b <= threshold;
                }
                nl += (uint32_t)__popcll(__ballot(left));
            }
            uint32_t at_left = 0, at_right = nl;
            for (uint32_t base = 0; base < m; base += 64) {
                const uint32_t j = base + lane;
                bool left = false, right = false; uint32_t s = 0;
                if (j < m) {
                    s = perm[start + j];
                    int b = (int)((0.5f * blo[axis][s] + 0.5f * bhi[axis][s] - origin) * scale);
                    b = min(max(b, 0), BINS - 1);
                    left = (float)b <= threshold; right = !left;
                }
                const unsigned long long ml = __ballot(left), mr = __ballot(right), below = (1ull << lane) - 1ull;
                if (left) scratch[at_left + (uint32_t)__popcll(ml & below)] = (uint16_t)s;
                if (right) scratch[at_right + (uint32_t)__popcll(mr & below)] = (uint16_t)s;
                at_left += (uint32_t)__popcll(ml); at_right += (uint32_t)__popcll(mr);
            }
            __syncthreads();
            for (uint32_t j = lane; j < m; j += 64) perm[start + j] = scratch[j];
            __syncthreads();
        }
        if (nl == 0u || nl >= m) nl = m / 2u;
        // Boxes of the two sides.
        float l0[3] = {INFINITY, INFINITY, INFINITY}, h0[3] = {-INFINITY, -INFINITY, -INFINITY}, l1[3] = {INFINITY, INFINITY, INFINITY}, h1[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t j = lane; j < m; j += 64) {
            const uint32_t s = perm[start + j];
            for (int k = 0; k < 3; k++) {
                if (j < nl) { l0[k] = fminf(l0[k], blo[k][s]); h0[k] = fmaxf(h0[k], bhi[k][s]); }
                else { l1[k] = fminf(l1[k], blo[k][s]); h1[k] = fmaxf(h1[k], bhi[k][s]); }
            }
        }
        for (int off = 32; off > 0; off >>= 1)
            for (int k = 0; k < 3; k++) {
                l0[k] = fminf(l0[k], __shfl_xor(l0[k], off, 64)); h0[k] = fmaxf(h0[k], __shfl_xor(h0[k], off, 64));
                l1[k] = fminf(l1[k], __shfl_xor(l1[k], off, 64)); h1[k] = fmaxf(h1[k], __shfl_xor(h1[k], off, 64));
            }
        const uint32_t slot = next_slot++;
        if (lane == 0) {
            rtl::QuantPair p;
            quantise_box(g, l0, h0, &p.c[0]); quantise_box(g, l1, h1, &p.c[1]);
            p.c[0].ref = rtl::REF_EMPTY; p.c[1].ref = rtl::REF_EMPTY;     // hooked in when the children are made
            tmp.pairs[slot] = p;
            tmp.used[slot] = 1u;
            hook(tmp, stats, link, slot);
            const uint32_t big = nl >= m - nl ? 0u : 1u;              // side that waits
            for (uint32_t e = 0; e < 2u; e++) {
                const uint32_t side = e == 0 ? big : 1u - big;
                task_start[n_tasks + e] = (uint16_t)(side == 0 ? start : start + nl); task_count[n_tasks + e] = (uint16_t)(side == 0 ? nl : m - nl);
                task_depth[n_tasks + e] = (uint16_t)(depth + 1u); task_link[n_tasks + e] = slot * 2u + side;
            }
        }
        n_tasks += 2;
        cost += (double)COST_BOX * (double)area;
        __syncthreads();
    }
    if (n_tasks > 0 && lane == 0) atomicMax(&stats->max_depth, 0x7FFFFFFFu);   // stack bound broken (cannot happen): reported as a tree that is too deep
    if (SHAPES)                                                       // the order of the shapes inside the group's range follows the leaves
        for (uint32_t j = lane; j < m_all; j += 64) tmp.order_out[a + j] = order_in[a + perm[j]];
    if (lane == 0) {
        if (SHAPES) { atomicAdd(&stats->n_leaves, leaves); atomicMax(&stats->max_leaf, max_leaf); atomicMax(&stats->max_depth, max_depth); }
        if (root_area > 0.0f) atomicAdd(&stats->cost_micro, (unsigned long long)(cost / (double)root_area * 1e6));
    }
}

// Radix nodes above the last level (more than GROUP of its items) become node pairs as they are, in the temporary
// slots slot_base + i, and hand link and depth to the groups below them.
__global__ __launch_bounds__(256) void emit_upper(uint32_t n, Topology t, NodeData d, const float4* __restrict__ slo, const float4* __restrict__ shi,
                                                  ItemIndex items, ItemIndex groups, ItemList top, uint32_t slot_base, Grid g, TempTree tmp,
                                                  float root_area, BuildStats* __restrict__ stats) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1 || items.under(t, i) <= GROUP) return;
    rtl::QuantPair out;
    const uint32_t c[2] = {t.left[i], t.right[i]};
    uint32_t depth = 1;                                            // pairs from the root down to and including this one
    for (uint32_t up = t.parent[i]; up != NONE && depth < 1024u; up = t.parent[up]) depth++;
    for (int k = 0; k < 2; k++) {
        float4 lo, hi; uint32_t ref = rtl::REF_EMPTY, below;       // a group's reference is hooked in when the group is built
        if (c[k] & LEAF_BIT) { const uint32_t p = c[k] & ~LEAF_BIT; lo = slo[p]; hi = shi[p]; below = p; }
        else {
            below = t.first[c[k]];
            if (items.under(t, c[k]) > GROUP) { ref = slot_base + c[k]; lo = d.lo[c[k]]; hi = d.hi[c[k]]; }
            else { const uint32_t j = groups.of(below); lo = top.lo[j]; hi = top.hi[j]; }
        }
        if (ref == rtl::REF_EMPTY) { const uint32_t j = groups.of(below); top.link[j] = (slot_base + i) * 2u + (uint32_t)k; top.depth[j] = depth; }
        const float l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
        quantise_box(g, l, h, &out.c[k]);
        out.c[k].ref = ref;
    }
    tmp.pairs[slot_base + i] = out;
    tmp.used[slot_base + i] = 1u;
    if (i == 0) stats->root_ref = slot_base;
    if (root_area > 0.0f) atomicAdd(&stats->cost_micro, (unsigned long long)((double)COST_BOX * half_area(d.lo[i], d.hi[i]) / (double)root_area * 1e6));
}

// Temporary slots -> consecutive node pairs of the blob, references renumbered.
__global__ __launch_bounds__(256) void compact_pairs(uint32_t slots, const rtl::QuantPair* __restrict__ in, const uint32_t* __restrict__ used,
                                                     const uint32_t* __restrict__ new_index, rtl::QuantPair* __restrict__ out) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= slots || !used[s]) return;
    rtl::QuantPair p = in[s];
    for (int k = 0; k < 2; k++)
        if (!(p.c[k].ref & rtl::REF_LEAF)) p.c[k].ref = p.c[k].ref < slots ? new_index[p.c[k].ref] : rtl::REF_EMPTY;
    out[new_index[s]] = p;
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
    hipLaunchKernelGGL(shape_boxes, dim3(std::min(blocks_n, BOXES_BLOCKS)), dim3(256), 0, stream, d_rec, n, d_lo, d_hi, d_bounds);
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
    Topology t;
    GPU_TRY(mem.alloc(&t.left, n)); GPU_TRY(mem.alloc(&t.right, n)); GPU_TRY(mem.alloc(&t.parent, n)); GPU_TRY(mem.alloc(&t.leaf_parent, n)); GPU_TRY(mem.alloc(&t.first, n));
    GPU_TRY(mem.alloc(&t.last, n));
    hipLaunchKernelGGL(radix_tree, dim3(blocks_i), dim3(256), 0, stream, d_keys_sorted, n, t);
    GPU_TRY(hipGetLastError());

    // Grid over the padded box of all tree shapes, as the host builder lays it out.
    rtl::BlobHeader h{};
    Grid g; g.pad_abs = in.pad_abs;
    float ext[3];
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
    float root_area = ext[0] * ext[1] + ext[1] * ext[2] + ext[2] * ext[0];
    if (!std::isfinite(root_area)) root_area = 0.0f;

    // 4. groups, level by level from the shapes up: lists[k] holds the groups of level k + 1 = the items of level k + 2
    uint32_t* d_start; GPU_TRY(mem.alloc(&d_start, n));
    ItemIndex index[LEVELS + 1]; index[0].incl = nullptr;
    ItemList lists[LEVELS];
    int levels = 0;
    uint32_t items_in = n;                  // items of the level being grouped
    const int max_levels = in.group_levels >= 1 && in.group_levels <= (uint32_t)LEVELS ? (int)in.group_levels : LEVELS;
    while (levels < max_levels) {
        uint32_t* d_incl; GPU_TRY(mem.alloc(&d_incl, n));
        GPU_TRY(hipMemsetAsync(d_start, 0, (size_t)n * sizeof(uint32_t), stream));
        hipLaunchKernelGGL(mark_groups, dim3(blocks_n), dim3(256), 0, stream, n, t, index[levels], d_start);
        GPU_TRY(hipGetLastError());
        size_t need_bytes = 0;
        GPU_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, need_bytes, d_start, d_incl, (int)n, stream));
        uint8_t* d_scan_temp; GPU_TRY(mem.alloc(&d_scan_temp, need_bytes));
        GPU_TRY(hipcub::DeviceScan::InclusiveSum(d_scan_temp, need_bytes, d_start, d_incl, (int)n, stream));
        uint32_t count = 0;
        GPU_TRY(hipMemcpyAsync(&count, d_incl + (n - 1), 4, hipMemcpyDeviceToHost, stream));
        GPU_TRY(hipStreamSynchronize(stream));
        if (count == 0 || count > items_in) { *err = "internal: the GPU tree build lost its groups"; return BUILD_FAILED; }
        index[levels + 1].incl = d_incl;
        ItemList& l = lists[levels];
        l.count = count;
        GPU_TRY(mem.alloc(&l.first, count)); GPU_TRY(mem.alloc(&l.last, count)); GPU_TRY(mem.alloc(&l.lo, count)); GPU_TRY(mem.alloc(&l.hi, count));
        GPU_TRY(mem.alloc(&l.link, count)); GPU_TRY(mem.alloc(&l.depth, count));
        hipLaunchKernelGGL(list_groups, dim3(blocks_n), dim3(256), 0, stream, n, t, index[levels], index[levels + 1], l);
        GPU_TRY(hipGetLastError());
        hipLaunchKernelGGL(group_boxes, dim3((count + 3) / 4), dim3(256), 0, stream, l, index[levels], levels == 0 ? (const float4*)d_slo : (const float4*)lists[levels - 1].lo,
                           levels == 0 ? (const float4*)d_shi : (const float4*)lists[levels - 1].hi);
        GPU_TRY(hipGetLastError());
        levels++;
        items_in = count;
        if (count == 1) break;              // the root of the radix tree is a group: nothing is left above
    }

    // 5. temporary slots: one range per level (as many slots as the level has items), one for the radix nodes left above
    uint32_t slot_base[LEVELS + 1];
    uint64_t slots64 = 0;
    for (int k = 0; k < levels; k++) { slot_base[k] = (uint32_t)slots64; slots64 += k == 0 ? n : lists[k - 1].count; }
    const bool upper = lists[levels - 1].count > 1;
    slot_base[levels] = (uint32_t)slots64;
    if (upper) slots64 += n - 1;
    if (slots64 >= (1ull << 30)) { *err = "GPU tree build: too many temporary node pairs"; return BUILD_FAILED; }
    const uint32_t slots = (uint32_t)slots64;
    TempTree tmp;
    uint32_t* d_new_index;
    GPU_TRY(mem.alloc(&tmp.pairs, slots)); GPU_TRY(mem.alloc(&tmp.used, slots)); GPU_TRY(mem.alloc(&d_new_index, slots)); GPU_TRY(mem.alloc(&tmp.order_out, n));
    GPU_TRY(hipMemsetAsync(tmp.used, 0, (size_t)slots * sizeof(uint32_t), stream));
    BuildStats* d_stats; GPU_TRY(mem.alloc(&d_stats, 1));
    GPU_TRY(hipMemsetAsync(d_stats, 0, sizeof(BuildStats), stream));
    NodeData d{};
    if (upper) {
        // boxes of the radix nodes that stay, bottom-up; the count of unfinished nodes comes back every few rounds
        GPU_TRY(mem.alloc(&d.lo, n)); GPU_TRY(mem.alloc(&d.hi, n)); GPU_TRY(mem.alloc(&d.done, n));
        uint32_t* d_remaining; GPU_TRY(mem.alloc(&d_remaining, 1));
        uint32_t remaining = 0;
        GPU_TRY(hipMemsetAsync(d.done, 0, (size_t)n * sizeof(uint32_t), stream));
        GPU_TRY(hipMemsetAsync(d_remaining, 0, sizeof(uint32_t), stream));
        hipLaunchKernelGGL(count_upper, dim3(blocks_i), dim3(256), 0, stream, n, t, index[levels - 1], d_remaining);
        GPU_TRY(hipGetLastError());
        GPU_TRY(hipMemcpyAsync(&remaining, d_remaining, sizeof(remaining), hipMemcpyDeviceToHost, stream));
        GPU_TRY(hipStreamSynchronize(stream));
        const uint32_t MAX_ROUNDS = 160;      // height of the radix tree is at most 63 + 28 key and position bits
        uint32_t round = 1;
        while (remaining != 0 && round <= MAX_ROUNDS) {
            for (int k = 0; k < 8; k++, round++) {
                hipLaunchKernelGGL(fit_upper_round, dim3(blocks_i), dim3(256), 0, stream, n, round, t, d, d_slo, d_shi, index[levels - 1], index[levels], lists[levels - 1], d_remaining);
                GPU_TRY(hipGetLastError());
            }
            GPU_TRY(hipMemcpyAsync(&remaining, d_remaining, sizeof(remaining), hipMemcpyDeviceToHost, stream));
            GPU_TRY(hipStreamSynchronize(stream));
        }
        if (remaining != 0) { *err = "internal: the bottom-up pass of the GPU tree build did not finish"; return BUILD_FAILED; }
        hipLaunchKernelGGL(emit_upper, dim3(blocks_i), dim3(256), 0, stream, n, t, d, d_slo, d_shi, index[levels - 1], index[levels], lists[levels - 1],
                           slot_base[levels], g, tmp, root_area, d_stats);
        GPU_TRY(hipGetLastError());
    }
    // from the top level down; each level needs at most 8 node pairs of depth below it (5 for the shapes: leaves of 8)
    for (int k = levels - 1; k >= 1; k--) {
        hipLaunchKernelGGL(build_groups<false>, dim3(lists[k].count), dim3(64), 0, stream, lists[k], index[k], (const float4*)lists[k - 1].lo, (const float4*)lists[k - 1].hi,
                           (const uint32_t*)lists[k - 1].first, (const uint32_t*)lists[k - 1].last, lists[k - 1].link, lists[k - 1].depth, slot_base[k],
                           (const uint32_t*)nullptr, g, tmp, in.depth_limit, (uint32_t)(5 + 8 * (k - 1)), root_area, d_stats);
        GPU_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(build_groups<true>, dim3(lists[0].count), dim3(64), 0, stream, lists[0], index[0], (const float4*)d_slo, (const float4*)d_shi,
                       (const uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, slot_base[0],
                       (const uint32_t*)d_order, g, tmp, in.depth_limit, 0u, root_area, d_stats);
    GPU_TRY(hipGetLastError());
    size_t need_bytes = 0;
    GPU_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, need_bytes, tmp.used, d_new_index, (int)slots, stream));
    uint8_t* d_scan_temp; GPU_TRY(mem.alloc(&d_scan_temp, need_bytes));
    GPU_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan_temp, need_bytes, tmp.used, d_new_index, (int)slots, stream));
    uint32_t last_index = 0, last_used = 0;
    BuildStats stats;
    GPU_TRY(hipMemcpyAsync(&last_index, d_new_index + (slots - 1), 4, hipMemcpyDeviceToHost, stream));
    GPU_TRY(hipMemcpyAsync(&last_used, tmp.used + (slots - 1), 4, hipMemcpyDeviceToHost, stream));
    GPU_TRY(hipMemcpyAsync(&stats, d_stats, sizeof(stats), hipMemcpyDeviceToHost, stream));
    GPU_TRY(hipStreamSynchronize(stream));
    const uint32_t n_pairs = last_index + last_used;
    if (stats.max_depth > in.depth_limit) return BUILD_TOO_DEEP;
    uint32_t root_ref = stats.root_ref;
    if (!(root_ref & rtl::REF_LEAF)) {
        if (root_ref >= slots || n_pairs == 0) { *err = "internal: the root of the GPU tree is not among its node pairs"; return BUILD_FAILED; }
        GPU_TRY(hipMemcpyAsync(&root_ref, d_new_index + root_ref, 4, hipMemcpyDeviceToHost, stream));
        GPU_TRY(hipStreamSynchronize(stream));
    } else if (n_pairs != 0) { *err = "internal: a single-leaf GPU tree with node pairs"; return BUILD_FAILED; }

    // 7. the blob
    h.magic = rtl::BLOB_MAGIC; h.version = rtl::BLOB_VERSION;
    h.n_pairs = n_pairs; h.n_tree_shapes = n; h.n_list_shapes = in.n_list; h.n_materials = in.n_mats; h.n_lights = in.n_lights;
    h.root_ref = root_ref;
    h.max_depth = stats.max_depth; h.finite_shading = in.finite_shading;
    h.n_leaves = stats.n_leaves; h.max_leaf = stats.max_leaf;
    h.sah_cost = (float)((double)stats.cost_micro * 1e-6);
    size_t off = sizeof(rtl::BlobHeader);
    h.off_nodes = off; off = align256(off + (size_t)n_pairs * sizeof(rtl::QuantPair));
    h.off_tree_shapes = off; off = align256(off + (size_t)n * sizeof(rtl::ShapeRec));
    h.off_list_shapes = off; off = align256(off + (size_t)in.n_list * sizeof(rtl::ShapeRec));
    h.off_materials = off; off = align256(off + (size_t)in.n_mats * sizeof(rtl::MaterialRec));
    h.off_lights = off; off = align256(off + (size_t)in.n_lights * sizeof(rtl::LightRec));
    h.total_bytes = off;

    uint8_t* d_blob = nullptr;
    GPU_TRY(hipMalloc((void**)&d_blob, h.total_bytes));
    struct BlobGuard { uint8_t* p; ~BlobGuard() { if (p) (void)hipFree(p); } } guard{d_blob};
    GPU_TRY(hipMemsetAsync(d_blob, 0, h.total_bytes, stream));
    hipLaunchKernelGGL(compact_pairs, dim3((slots + 255) / 256), dim3(256), 0, stream, slots, tmp.pairs, tmp.used, d_new_index, (rtl::QuantPair*)(d_blob + h.off_nodes));
    GPU_TRY(hipGetLastError());
    hipLaunchKernelGGL(gather_records, dim3(blocks_n), dim3(256), 0, stream, d_rec, tmp.order_out, n, (rtl::ShapeRec*)(d_blob + h.off_tree_shapes));
    GPU_TRY(hipGetLastError());
    if (in.n_list) GPU_TRY(hipMemcpyAsync(d_blob + h.off_list_shapes, in.list, (size_t)in.n_list * sizeof(rtl::ShapeRec), hipMemcpyHostToDevice, stream));
    if (in.n_mats) GPU_TRY(hipMemcpyAsync(d_blob + h.off_materials, in.mats, (size_t)in.n_mats * sizeof(rtl::MaterialRec), hipMemcpyHostToDevice, stream));
    if (in.n_lights) GPU_TRY(hipMemcpyAsync(d_blob + h.off_lights, in.lights, (size_t)in.n_lights * sizeof(rtl::LightRec), hipMemcpyHostToDevice, stream));
    GPU_TRY(hipMemcpyAsync(d_blob, &h, sizeof(h), hipMemcpyHostToDevice, stream));
    GPU_TRY(hipStreamSynchronize(stream));
    out->d_blob = d_blob; out->bytes = h.total_bytes; out->header = h;
    guard.p = nullptr;
    return BUILD_OK;
}

}  // namespace rtg
