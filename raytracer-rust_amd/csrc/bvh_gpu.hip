// bvh_gpu.hip -- the tree of RayTracer::new built on the GPU.
//
// The reference builds its bounding volume hierarchy on the CPU before it starts the clock
// (Bvh::build_par, src/raytracer/raytracer.rs:32; main.rs:11-15).  The host builder of this
// library (bvh_build.cpp) runs a binned surface-area split search from the root down, about a
// second per eight million triangles; this file runs the same search on the GPU in a tenth of
// that.  Pixels do not depend on the tree: it only culls (layout.h, render_kernels.hip).
//
// Steps, all on one stream, every kernel a plain data-parallel pass that reads only what earlier
// launches wrote (no inter-block hand-offs inside a launch):
//   1. boxes of the shapes (shape.rs:205-234), bounds of the boxes and of their centres
//   2. 63-bit Morton code of every centre, radix sort (hipCUB): the order the shapes start in, so
//      that what lies together in space lies together in memory
//   3. large nodes (more than GROUP shapes), one tree level per round: every node is cut into
//      chunks of CHUNK shapes, a workgroup per chunk bins its shapes (3 axes x BINS bins in LDS,
//      merged into the node's bins with atomics), a wavefront per node picks the cheapest split
//      and turns the chunks' bin counts into the places their shapes go to, a workgroup per chunk
//      moves them (stable) and collects the boxes of the two sides, a thread per node writes the
//      node pair and hands its children to the next round or, once they have at most GROUP
//      shapes, to step 4
//   4. small nodes: a wavefront per group of at most GROUP shapes runs the same search down to
//      the leaves with everything in LDS, and puts the shapes of its group into leaf order
//   5. used slots numbered by a prefix sum; node pairs and shape records written in final order
//      straight into the scene blob
// Node pairs are written on the 16-bit scene grid from the start, into temporary slots; whoever
// makes a node is told which child reference of which pair waits for it (`link`) and how many
// pairs lie above it (`depth`, which is what bounds the tree: halving by position takes over
// where the split search would break the traversal stack's depth).
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
constexpr float COST_BOX = 1.0f, COST_SHAPE = 1.0f;   // relative cost of one box-pair step and one shape test in the render kernel (10 M-triangle mesh at 8K with the final kernel: 31.9 ms; 0.6: 34.5, 0.8: 32.2, 1.3: 32.1)

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

__device__ __forceinline__ float half_area3(const float* l, const float* h) {
    const float dx = h[0] - l[0], dy = h[1] - l[1], dz = h[2] - l[2];
    return dx * dy + dy * dz + dz * dx;
}

struct Grid { float origin[3], scale[3], inv_scale[3], pad_abs; };

// Outward padding as the host builder applies it, then outward rounding onto the scene grid (layout.h) with one
// extra step of margin for the rounding of this very computation.
__device__ __forceinline__ void quantise_box(const Grid& g, const float* l, const float* h, rtl::QuantChild* out) {
    for (int k = 0; k < 3; k++) {
        const float pad = fmaxf(fabsf(l[k]), fabsf(h[k])) * 4e-7f + g.pad_abs;
        // (times the rounded reciprocal: off by less than a hundredth of a step at the far end of the grid, the extra step covers it)
        const float ul = floorf((l[k] - pad - g.origin[k]) * g.inv_scale[k]) - 1.0f;
        const float uh = ceilf((h[k] + pad - g.origin[k]) * g.inv_scale[k]) + 1.0f;
        out->q[k][0] = (uint16_t)rtl::grid_code_below(fminf(fmaxf(ul, 0.0f), rtl::GRID_UNITS));
        out->q[k][1] = (uint16_t)rtl::grid_code_above(fminf(fmaxf(uh, 0.0f), rtl::GRID_UNITS));
    }
}

constexpr uint32_t GROUP = 256;          // most shapes of a node that one wavefront finishes in LDS
constexpr uint32_t CHUNK = 2048;         // shapes of a large node that one workgroup bins and moves
constexpr int BINS = 16;                 // 8, 16 and 32 bins per axis give the same frame times within 1 %

struct Bins {                            // per axis and bin: box of the shapes whose centre falls into it (ordered uints), their number
    uint32_t lo[3][BINS][3], hi[3][BINS][3], cnt[3][BINS];
};

// nb: bins in use per axis (BINS for large nodes; small nodes take fewer, as the host builder does: most nodes of a
// tree hold a handful of shapes, and clearing and sweeping all bins for each of them would be most of the work).
__device__ __forceinline__ int bin_of(float lo, float hi, float origin, float scale, int nb = BINS) {
    const int b = (int)((0.5f * lo + 0.5f * hi - origin) * scale);
    return min(max(b, 0), nb - 1);
}

__device__ __forceinline__ void bin_scales(const float* clo, const float* chi, float* inv, int nb = BINS) {
    for (int k = 0; k < 3; k++) {
        inv[k] = chi[k] > clo[k] ? (float)nb * 0.999999f / (chi[k] - clo[k]) : 0.0f;
        if (!(inv[k] < INFINITY)) inv[k] = 0.0f;
    }
}

// Cheapest split "bins 0..b of axis ax to the left" over all axes with a usable scale, candidates spread over the
// lanes of the calling wavefront.  Every lane returns the same: *best = sum over the sides of area x count (INFINITY
// when nothing separates), result = ax * BINS + b or -1.
__device__ int best_split(const Bins& bins, const float* inv, uint32_t lane, float* best_out, int nb = BINS) {
    float best = INFINITY; int best_id = -1;
    for (uint32_t cand = lane; cand < 3u * BINS; cand += 64) {
        const uint32_t ax = cand / BINS, b = cand % BINS;
        if (b + 1 < (uint32_t)nb && inv[ax] != 0.0f) {
            float l0[3] = {INFINITY, INFINITY, INFINITY}, h0[3] = {-INFINITY, -INFINITY, -INFINITY}, l1[3] = {INFINITY, INFINITY, INFINITY}, h1[3] = {-INFINITY, -INFINITY, -INFINITY};
            uint32_t n0 = 0, n1 = 0;
            for (uint32_t q = 0; q < (uint32_t)nb; q++) {
                const uint32_t c = bins.cnt[ax][q];
                if (c == 0u) continue;
                for (int k = 0; k < 3; k++) {
                    const float l = unordered(bins.lo[ax][q][k]), h = unordered(bins.hi[ax][q][k]);
                    if (q <= b) { l0[k] = fminf(l0[k], l); h0[k] = fmaxf(h0[k], h); } else { l1[k] = fminf(l1[k], l); h1[k] = fmaxf(h1[k], h); }
                }
                if (q <= b) n0 += c; else n1 += c;
            }
            if (n0 && n1) {
                const float v = half_area3(l0, h0) * (float)n0 + half_area3(l1, h1) * (float)n1;
                if (v < best) { best = v; best_id = (int)cand; }       // overflow or NaN: not a candidate
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64); const int oi = __shfl_xor(best_id, off, 64);
        if (oi >= 0 && (best_id < 0 || ob < best || (ob == best && oi < best_id))) { best = ob; best_id = oi; }
    }
    *best_out = __shfl(best, 0, 64);
    return __shfl(best_id, 0, 64);                                     // one decision for the whole wavefront
}

__device__ __forceinline__ float first_lane(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// Halving by position instead of a split search: when the levels a balanced subtree still needs (leaves of
// MAX_LEAF shapes) would otherwise not fit under the depth limit.
__device__ __forceinline__ bool must_halve(uint32_t depth, uint32_t count, uint32_t depth_limit) {
    uint32_t need = 0;
    while ((rtl::MAX_LEAF << need) < count) need++;
    return depth + need + 1u >= depth_limit;
}

struct BuildStats { uint32_t n_leaves, max_leaf, max_depth, root_ref; unsigned long long cost_micro; uint32_t n_groups, pad; };

// Temporary node pairs.  A group whose first shape sits at position a puts its k-th pair into slot a + k (a group of
// m shapes makes at most m - 1 pairs); the large nodes of a round use one slot each behind those.  `used` marks the
// slots that hold a pair.
struct TempTree {
    rtl::QuantPair* pairs; uint32_t* used;
    uint32_t* order_out;          // final position -> shape index
};

__device__ __forceinline__ void hook(const TempTree& tmp, BuildStats* stats, uint32_t link, uint32_t ref) {
    if (link == NONE) stats->root_ref = ref; else tmp.pairs[link >> 1].c[link & 1u].ref = ref;
}

// ------------------------------------------------------------------------------------------------ large nodes
struct BigNode {
    uint32_t first, count;             // its shapes: positions [first, first + count) of the permutation
    uint32_t link, depth;
    uint32_t clo[3], chi[3];           // bounds of the centres of its shapes (ordered uints)
};

struct SideBounds { uint32_t lo[3], hi[3], clo[3], chi[3]; };    // box and centre bounds of one side of a split (ordered uints)

struct Split {
    int axis;                          // -1: halves by position
    float origin, scale; uint32_t bin; // shapes with bin_of(axis) <= bin go left
    uint32_t n_left;
};

struct Level {                         // work arrays of one round; n_nodes nodes, their chunks numbered node after node
    const BigNode* nodes; uint32_t n_nodes;
    const uint32_t* chunk_off;         // first chunk of node i; [n_nodes] = number of chunks
    Bins* bins;                        // per node
    uint32_t* chunk_cnt;               // per chunk: 3 x BINS counts
    uint32_t* chunk_left;              // per chunk: shapes that go left in the chunks of the same node before it
    Split* split;                      // per node
    SideBounds* sides;                 // per node x 2
    uint32_t* n_big;                   // per node: children that are large nodes again
};

__global__ __launch_bounds__(256) void level_begin(Level lv, uint32_t* __restrict__ n_chunks_of) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < lv.n_nodes) n_chunks_of[t] = (lv.nodes[t].count + CHUNK - 1) / CHUNK;
    if (t == lv.n_nodes) n_chunks_of[t] = 0u;
    // bins and side bounds start empty
    const uint32_t per_node = 3u * BINS;
    if (t < lv.n_nodes * per_node) {
        const uint32_t i = t / per_node, ax = (t % per_node) / BINS, b = t % BINS;
        for (int k = 0; k < 3; k++) { lv.bins[i].lo[ax][b][k] = 0xFFFFFFFFu; lv.bins[i].hi[ax][b][k] = 0u; }
        lv.bins[i].cnt[ax][b] = 0u;
    }
    if (t < lv.n_nodes * 2u)
        for (int k = 0; k < 3; k++) { lv.sides[t].lo[k] = lv.sides[t].clo[k] = 0xFFFFFFFFu; lv.sides[t].hi[k] = lv.sides[t].chi[k] = 0u; }
}

// Node that chunk c belongs to: the last i with chunk_off[i] <= c.
__device__ __forceinline__ uint32_t node_of_chunk(const Level& lv, uint32_t c) {
    uint32_t a = 0, b = lv.n_nodes;          // chunk_off[a] <= c < chunk_off[b]
    while (b - a > 1u) { const uint32_t m = (a + b) >> 1; if (lv.chunk_off[m] <= c) a = m; else b = m; }
    return a;
}

__global__ __launch_bounds__(256) void bin_chunks(Level lv, const uint32_t* __restrict__ idx, const float4* __restrict__ slo, const float4* __restrict__ shi,
                                                  uint32_t depth_limit) {
    __shared__ Bins bins;
    __shared__ uint32_t s_node;
    const uint32_t c = blockIdx.x;
    if (c >= lv.chunk_off[lv.n_nodes]) return;
    if (threadIdx.x == 0) s_node = node_of_chunk(lv, c);
    for (uint32_t j = threadIdx.x; j < 3u * BINS; j += blockDim.x) {
        const uint32_t ax = j / BINS, b = j % BINS;
        for (int k = 0; k < 3; k++) { bins.lo[ax][b][k] = 0xFFFFFFFFu; bins.hi[ax][b][k] = 0u; }
        bins.cnt[ax][b] = 0u;
    }
    __syncthreads();
    const uint32_t i = s_node;
    const BigNode nd = lv.nodes[i];
    if (must_halve(nd.depth, nd.count, depth_limit)) return;          // no split search for this node
    const uint32_t s = nd.first + (c - lv.chunk_off[i]) * CHUNK, e = min(nd.first + nd.count, s + CHUNK);
    float clo[3], chi[3], inv[3];
    for (int k = 0; k < 3; k++) { clo[k] = unordered(nd.clo[k]); chi[k] = unordered(nd.chi[k]); }
    bin_scales(clo, chi, inv);
    for (uint32_t p = s + threadIdx.x; p < e; p += blockDim.x) {
        const uint32_t sh = idx[p];
        const float4 l = slo[sh], h = shi[sh];
        const float bl[3] = {l.x, l.y, l.z}, bh[3] = {h.x, h.y, h.z};
        for (int ax = 0; ax < 3; ax++) {
            if (inv[ax] == 0.0f) continue;
            const int b = bin_of(bl[ax], bh[ax], clo[ax], inv[ax]);
            for (int k = 0; k < 3; k++) { atomicMin(&bins.lo[ax][b][k], ordered(bl[k])); atomicMax(&bins.hi[ax][b][k], ordered(bh[k])); }
            atomicAdd(&bins.cnt[ax][b], 1u);
        }
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < 3u * BINS; j += blockDim.x) {
        const uint32_t ax = j / BINS, b = j % BINS, n = bins.cnt[ax][b];
        lv.chunk_cnt[(size_t)c * (3u * BINS) + j] = n;
        if (n) {
            for (int k = 0; k < 3; k++) { atomicMin(&lv.bins[i].lo[ax][b][k], bins.lo[ax][b][k]); atomicMax(&lv.bins[i].hi[ax][b][k], bins.hi[ax][b][k]); }
            atomicAdd(&lv.bins[i].cnt[ax][b], n);
        }
    }
}

// One wavefront per node: the split, and for every chunk of the node how many shapes of the chunks before it go left.
__global__ __launch_bounds__(256) void split_nodes(Level lv, uint32_t depth_limit) {
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (i >= lv.n_nodes) return;
    const BigNode nd = lv.nodes[i];
    Split sp; sp.axis = -1; sp.origin = 0.0f; sp.scale = 0.0f; sp.bin = 0u; sp.n_left = nd.count / 2u;
    if (!must_halve(nd.depth, nd.count, depth_limit)) {
        float clo[3], chi[3], inv[3], best;
        for (int k = 0; k < 3; k++) { clo[k] = unordered(nd.clo[k]); chi[k] = unordered(nd.chi[k]); }
        bin_scales(clo, chi, inv);
        const int id = best_split(lv.bins[i], inv, lane, &best);
        if (id >= 0) { sp.axis = id / BINS; sp.bin = (uint32_t)(id % BINS); sp.origin = clo[sp.axis]; sp.scale = inv[sp.axis]; }
    }
    const uint32_t c0 = lv.chunk_off[i], c1 = lv.chunk_off[i + 1];
    uint32_t running = 0;
    for (uint32_t base = c0; base < c1; base += 64) {
        const uint32_t c = base + lane;
        uint32_t mine = 0;
        if (c < c1) {
            if (sp.axis >= 0) {
                const uint32_t* cnt = lv.chunk_cnt + (size_t)c * (3u * BINS) + (uint32_t)sp.axis * BINS;
                for (uint32_t b = 0; b <= sp.bin; b++) mine += cnt[b];
            } else {                                                    // halves by position
                const uint32_t s = (c - c0) * CHUNK, e = min(nd.count, s + CHUNK), half = nd.count / 2u;
                mine = min(e, half) - min(s, half);
            }
        }
        uint32_t incl = mine;
        for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(incl, off, 64); if (lane >= (uint32_t)off) incl += v; }
        if (c < c1) lv.chunk_left[c] = running + incl - mine;
        running += __shfl(incl, 63, 64);
    }
    if (sp.axis >= 0) sp.n_left = running;
    if (lane == 0) {
        lv.split[i] = sp;
        lv.n_big[i] = (sp.n_left > GROUP ? 1u : 0u) + (nd.count - sp.n_left > GROUP ? 1u : 0u);
        if (i == 0) lv.n_big[lv.n_nodes] = 0u;
    }
}

// A workgroup per chunk moves the chunk's shapes to their side of the split, keeping their order, and collects the
// boxes and centre bounds of the two sides.
__global__ __launch_bounds__(256) void move_chunks(Level lv, const uint32_t* __restrict__ idx, uint32_t* __restrict__ idx_out, const float4* __restrict__ slo,
                                                   const float4* __restrict__ shi) {
    __shared__ uint32_t s_node, s_wave[4][2];
    __shared__ SideBounds s_side[2];
    const uint32_t c = blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (c >= lv.chunk_off[lv.n_nodes]) return;
    if (threadIdx.x == 0) s_node = node_of_chunk(lv, c);
    if (threadIdx.x < 2)
        for (int k = 0; k < 3; k++) { s_side[threadIdx.x].lo[k] = s_side[threadIdx.x].clo[k] = 0xFFFFFFFFu; s_side[threadIdx.x].hi[k] = s_side[threadIdx.x].chi[k] = 0u; }
    __syncthreads();
    const uint32_t i = s_node;
    const BigNode nd = lv.nodes[i];
    const Split sp = lv.split[i];
    const uint32_t s = nd.first + (c - lv.chunk_off[i]) * CHUNK, e = min(nd.first + nd.count, s + CHUNK);
    uint32_t at_left = nd.first + lv.chunk_left[c];                                  // next free place on either side
    uint32_t at_right = nd.first + sp.n_left + ((s - nd.first) - lv.chunk_left[c]);
    for (uint32_t base = s; base < e; base += blockDim.x) {
        const uint32_t p = base + threadIdx.x;
        bool left = false, right = false; uint32_t sh = 0;
        float bl[3] = {0, 0, 0}, bh[3] = {0, 0, 0};
        if (p < e) {
            sh = idx[p];
            const float4 l = slo[sh], h = shi[sh];
            bl[0] = l.x; bl[1] = l.y; bl[2] = l.z; bh[0] = h.x; bh[1] = h.y; bh[2] = h.z;
            left = sp.axis >= 0 ? (uint32_t)bin_of(bl[sp.axis], bh[sp.axis], sp.origin, sp.scale) <= sp.bin : p - nd.first < sp.n_left;
            right = !left;
        }
        const unsigned long long ml = __ballot(left), mr = __ballot(right), below = (1ull << lane) - 1ull;
        if (lane == 0) { s_wave[wave][0] = (uint32_t)__popcll(ml); s_wave[wave][1] = (uint32_t)__popcll(mr); }
        __syncthreads();
        uint32_t before_l = 0, before_r = 0, all_l = 0, all_r = 0;
        for (uint32_t w = 0; w < 4; w++) {
            if (w < wave) { before_l += s_wave[w][0]; before_r += s_wave[w][1]; }
            all_l += s_wave[w][0]; all_r += s_wave[w][1];
        }
        if (left) idx_out[at_left + before_l + (uint32_t)__popcll(ml & below)] = sh;
        if (right) idx_out[at_right + before_r + (uint32_t)__popcll(mr & below)] = sh;
        at_left += all_l; at_right += all_r;
        if (p < e) {
            SideBounds& sb = s_side[left ? 0 : 1];
            for (int k = 0; k < 3; k++) {
                const uint32_t ctr = ordered(0.5f * bl[k] + 0.5f * bh[k]);
                atomicMin(&sb.lo[k], ordered(bl[k])); atomicMax(&sb.hi[k], ordered(bh[k])); atomicMin(&sb.clo[k], ctr); atomicMax(&sb.chi[k], ctr);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 6) {
        const uint32_t side = threadIdx.x / 3, k = threadIdx.x % 3;
        SideBounds& out = lv.sides[2u * i + side];
        if (s_side[side].lo[k] <= s_side[side].hi[k]) {                // the chunk had shapes on this side
            atomicMin(&out.lo[k], s_side[side].lo[k]); atomicMax(&out.hi[k], s_side[side].hi[k]);
            atomicMin(&out.clo[k], s_side[side].clo[k]); atomicMax(&out.chi[k], s_side[side].chi[k]);
        }
    }
}

__global__ __launch_bounds__(256) void copy_chunks(Level lv, const uint32_t* __restrict__ from, uint32_t* __restrict__ to) {
    __shared__ uint32_t s_node;
    const uint32_t c = blockIdx.x;
    if (c >= lv.chunk_off[lv.n_nodes]) return;
    if (threadIdx.x == 0) s_node = node_of_chunk(lv, c);
    __syncthreads();
    const BigNode nd = lv.nodes[s_node];
    const uint32_t s = nd.first + (c - lv.chunk_off[s_node]) * CHUNK, e = min(nd.first + nd.count, s + CHUNK);
    for (uint32_t p = s + threadIdx.x; p < e; p += blockDim.x) to[p] = from[p];
}

struct GroupList {                     // nodes of at most GROUP shapes, in the order they came up
    uint32_t* first; uint32_t* count; uint32_t* link; uint32_t* depth;
};

// A thread per node: its node pair, and its children handed on.
__global__ __launch_bounds__(256) void emit_nodes(Level lv, const uint32_t* __restrict__ next_index, BigNode* __restrict__ next, GroupList groups, uint32_t slot_base,
                                                  Grid g, TempTree tmp, float root_area, BuildStats* __restrict__ stats) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lv.n_nodes) return;
    const BigNode nd = lv.nodes[i];
    const Split sp = lv.split[i];
    const uint32_t slot = slot_base + i;
    rtl::QuantPair pair;
    float ul[3] = {INFINITY, INFINITY, INFINITY}, uh[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t big_at = next_index[i];
    for (uint32_t side = 0; side < 2u; side++) {
        const SideBounds& sb = lv.sides[2u * i + side];
        float l[3], h[3];
        for (int k = 0; k < 3; k++) { l[k] = unordered(sb.lo[k]); h[k] = unordered(sb.hi[k]); ul[k] = fminf(ul[k], l[k]); uh[k] = fmaxf(uh[k], h[k]); }
        quantise_box(g, l, h, &pair.c[side]);
        pair.c[side].ref = rtl::REF_EMPTY;                             // hooked in when the child is made
        const uint32_t first = side == 0 ? nd.first : nd.first + sp.n_left, count = side == 0 ? sp.n_left : nd.count - sp.n_left;
        if (count > GROUP) {
            BigNode ch; ch.first = first; ch.count = count; ch.link = slot * 2u + side; ch.depth = nd.depth + 1u;
            for (int k = 0; k < 3; k++) { ch.clo[k] = sb.clo[k]; ch.chi[k] = sb.chi[k]; }
            next[big_at++] = ch;
        } else {
            const uint32_t j = atomicAdd(&stats->n_groups, 1u);
            groups.first[j] = first; groups.count[j] = count; groups.link[j] = slot * 2u + side; groups.depth[j] = nd.depth + 1u;
        }
    }
    tmp.pairs[slot] = pair;
    tmp.used[slot] = 1u;
    hook(tmp, stats, nd.link, slot);
    if (root_area > 0.0f) atomicAdd(&stats->cost_micro, (unsigned long long)((double)COST_BOX * (double)half_area3(ul, uh) / (double)root_area * 1e6));
}

__global__ __launch_bounds__(256) void iota(uint32_t* __restrict__ a, uint32_t* __restrict__ b, uint32_t n) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) { a[p] = p; b[p] = p; }
}

// Shape boxes and shape indices in the order of the permutation the large nodes left behind.
__global__ __launch_bounds__(256) void gather_by_index(const uint32_t* __restrict__ idx, uint32_t n, const float4* __restrict__ slo, const float4* __restrict__ shi,
                                                       const uint32_t* __restrict__ order, float4* __restrict__ plo, float4* __restrict__ phi, uint32_t* __restrict__ order_p) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t sh = idx[p];
    plo[p] = slo[sh]; phi[p] = shi[sh]; order_p[p] = order[sh];
}

// ------------------------------------------------------------------------------------------------ small nodes
// Binned surface-area split search over the shapes of one group, one wavefront per group, everything in LDS.
// The rules of the host builder (bvh_build.cpp): cost of a split = area-weighted counts of the two sides; a node of
// at most MAX_LEAF shapes stays a leaf when that is cheaper; halving by position when the depth budget runs out or no
// split separates anything.  The shapes of the group are put into leaf order.
__global__ __launch_bounds__(64) void build_groups(GroupList groups, uint32_t n_groups, const float4* __restrict__ plo, const float4* __restrict__ phi,
                                                   const uint32_t* __restrict__ order_in, Grid g, TempTree tmp, uint32_t depth_limit,
                                                   float root_area, BuildStats* __restrict__ stats) {
    __shared__ float blo[3][GROUP], bhi[3][GROUP];
    __shared__ uint16_t perm[GROUP], scratch[GROUP];
    __shared__ Bins bins;
    // Pending nodes.  The smaller child is always taken next, so every node that waits is at least as large as all
    // that is above it on the stack: at most log2(GROUP) + 1 wait at any time.
    constexpr int STACK = 40;
    __shared__ uint16_t task_start[STACK], task_count[STACK], task_depth[STACK];
    __shared__ uint32_t task_link[STACK];
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    if (c >= n_groups) return;
    const uint32_t a = groups.first[c], m_all = groups.count[c];
    if (m_all == 0u || m_all > GROUP) return;                         // cannot happen: a group is defined by this bound
    for (uint32_t j = lane; j < m_all; j += 64) {
        const float4 l = plo[a + j], h = phi[a + j];
        blo[0][j] = l.x; blo[1][j] = l.y; blo[2][j] = l.z; bhi[0][j] = h.x; bhi[1][j] = h.y; bhi[2][j] = h.z;
        perm[j] = (uint16_t)j;
    }
    if (lane == 0) { task_start[0] = 0; task_count[0] = (uint16_t)m_all; task_depth[0] = (uint16_t)min(groups.depth[c], 60000u); task_link[0] = groups.link[c]; }
    __syncthreads();
    uint32_t n_tasks = 1, next_slot = a, leaves = 0, max_leaf = 0, max_depth = 0;
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
        // The node's shapes sit in lanes 0 .. min(m, 64) - 1: the butterfly only needs the levels that span them (most
        // nodes hold a handful of shapes), and lane 0 then holds the result for everyone.
        const int top = m > 32u ? 32 : m > 1u ? 1 << (31 - __clz((int)m - 1)) : 0;
        for (int off = top; off > 0; off >>= 1)
            for (int k = 0; k < 3; k++) {
                lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
                clo[k] = fminf(clo[k], __shfl_xor(clo[k], off, 64)); chi[k] = fmaxf(chi[k], __shfl_xor(chi[k], off, 64));
            }
        for (int k = 0; k < 3; k++) { lo[k] = first_lane(lo[k]); hi[k] = first_lane(hi[k]); clo[k] = first_lane(clo[k]); chi[k] = first_lane(chi[k]); }
        const float area = half_area3(lo, hi);
        const bool halve = must_halve(depth, m, depth_limit);
        bool make_leaf = m == 1u || (halve && m <= rtl::MAX_LEAF);
        int axis = -1, nb = BINS; uint32_t split_bin = 0; float scale = 0.0f, origin = 0.0f;
        if (!make_leaf && !halve) {
            nb = m <= 8u ? 4 : m <= 32u ? 8 : BINS;
            for (uint32_t j = lane; j < 3u * (uint32_t)nb; j += 64) {
                const uint32_t ax = j / (uint32_t)nb, b = j % (uint32_t)nb;
                for (int k = 0; k < 3; k++) { bins.lo[ax][b][k] = 0xFFFFFFFFu; bins.hi[ax][b][k] = 0u; }
                bins.cnt[ax][b] = 0u;
            }
            __syncthreads();
            float inv[3];
            bin_scales(clo, chi, inv, nb);
            for (uint32_t j = lane; j < m; j += 64) {
                const uint32_t s = perm[start + j];
                for (int ax = 0; ax < 3; ax++) {
                    if (inv[ax] == 0.0f) continue;
                    const int b = bin_of(blo[ax][s], bhi[ax][s], clo[ax], inv[ax], nb);
                    for (int k = 0; k < 3; k++) { atomicMin(&bins.lo[ax][b][k], ordered(blo[k][s])); atomicMax(&bins.hi[ax][b][k], ordered(bhi[k][s])); }
                    atomicAdd(&bins.cnt[ax][b], 1u);
                }
            }
            __syncthreads();
            float best;
            const int best_id = best_split(bins, inv, lane, &best, nb);
            if (best_id >= 0) {
                const float split_cost = COST_BOX + COST_SHAPE * (area > 0.0f ? best / area : (float)m);
                if (m <= rtl::MAX_LEAF && COST_SHAPE * (float)m <= split_cost) make_leaf = true;
                else { axis = best_id / BINS; split_bin = (uint32_t)(best_id % BINS); scale = inv[axis]; origin = clo[axis]; }
            } else if (m <= rtl::MAX_LEAF) {
                make_leaf = true;                                     // all centres coincide and the shapes fit one leaf
            }
        }
        if (make_leaf) {
            if (lane == 0) hook(tmp, stats, link, rtl::REF_LEAF | ((a + start) << 3) | (m - 1u));
            leaves++; max_leaf = max(max_leaf, m); max_depth = max(max_depth, depth);
            cost += (double)COST_SHAPE * (double)m * (double)area;
            continue;
        }
        // Partition perm[start, start + m): by bin on the chosen axis, or in halves by position.
        uint32_t nl = 0;
        if (axis >= 0) {
            for (uint32_t base = 0; base < m; base += 64) {
                const uint32_t j = base + lane;
                bool left = false;
                if (j < m) { const uint32_t s = perm[start + j]; left = (uint32_t)bin_of(blo[axis][s], bhi[axis][s], origin, scale, nb) <= split_bin; }
                nl += (uint32_t)__popcll(__ballot(left));
            }
            uint32_t at_left = 0, at_right = nl;
            for (uint32_t base = 0; base < m; base += 64) {
                const uint32_t j = base + lane;
                bool left = false, right = false; uint32_t s = 0;
                if (j < m) { s = perm[start + j]; left = (uint32_t)bin_of(blo[axis][s], bhi[axis][s], origin, scale, nb) <= split_bin; right = !left; }
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
        for (int off = top; off > 0; off >>= 1)
            for (int k = 0; k < 3; k++) {
                l0[k] = fminf(l0[k], __shfl_xor(l0[k], off, 64)); h0[k] = fmaxf(h0[k], __shfl_xor(h0[k], off, 64));
                l1[k] = fminf(l1[k], __shfl_xor(l1[k], off, 64)); h1[k] = fmaxf(h1[k], __shfl_xor(h1[k], off, 64));
            }
        for (int k = 0; k < 3; k++) { l0[k] = first_lane(l0[k]); h0[k] = first_lane(h0[k]); l1[k] = first_lane(l1[k]); h1[k] = first_lane(h1[k]); }
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
    for (uint32_t j = lane; j < m_all; j += 64) tmp.order_out[a + j] = order_in[a + perm[j]];
    if (lane == 0) {
        atomicAdd(&stats->n_leaves, leaves); atomicMax(&stats->max_leaf, max_leaf); atomicMax(&stats->max_depth, max_depth);
        if (root_area > 0.0f) atomicAdd(&stats->cost_micro, (unsigned long long)(cost / (double)root_area * 1e6));
    }
}

// Temporary slots -> consecutive node pairs of the blob, references renumbered.
__global__ __launch_bounds__(256) void compact_pairs(uint32_t slots, const rtl::QuantPair* __restrict__ in, const uint32_t* __restrict__ used,
                                                     const uint32_t* __restrict__ new_index, rtl::QuantPair* __restrict__ out) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= slots || !used[s]) return;
    rtl::QuantPair p = in[s];
    for (int k = 0; k < 2; k++)
        if (!(p.c[k].ref & rtl::REF_LEAF)) p.c[k].ref = p.c[k].ref < slots ? rtl::inner_ref(new_index[p.c[k].ref]) : rtl::REF_EMPTY;
    out[new_index[s]] = p;
}

__global__ __launch_bounds__(256) void gather_records(const rtl::ShapeRec* __restrict__ rec, const uint32_t* __restrict__ order, uint32_t n,
                                                      rtl::ShapeRec* __restrict__ out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    // 48-byte records as three 16-byte words; triangles leave as {v0, v1 - v0, v2 - v0} (rtl::to_blob_form)
    const float4* src = (const float4*)(rec + order[p]);
    float4* dst = (float4*)(out + p);
    const float4 q0 = src[0];
    float4 q1 = src[1], q2 = src[2];
    if ((__float_as_uint(q0.w) & 3u) == rtl::KIND_TRIANGLE) {
        q1.x = q1.x - q0.x; q1.y = q1.y - q0.y; q1.z = q1.z - q0.z;
        q2.x = q2.x - q0.x; q2.y = q2.y - q0.y; q2.z = q2.z - q0.z;
    }
    dst[0] = q0; dst[1] = q1; dst[2] = q2;
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
    if (n < 2 || n > rtl::MAX_TREE_SHAPES) { *err = "GPU tree build needs between 2 and 2^26 shapes"; return BUILD_FAILED; }
    const uint32_t blocks_n = (n + 255) / 256;
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
    // 2. Morton codes, sort; from here on a shape is known by its place in the sorted order
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

    // Grid over the padded box of all tree shapes, as the host builder lays it out.
    rtl::BlobHeader h{};
    Grid g; g.pad_abs = in.pad_abs;
    float ext[3];
    for (int k = 0; k < 3; k++) {
        const float lo = unordered(bounds.v[k]), hi = unordered(bounds.v[3 + k]);
        ext[k] = hi - lo;
        const float pad = fmaxf(fabsf(lo), fabsf(hi)) * 4e-7f + in.pad_abs;
        g.origin[k] = nextafterf(lo - pad, -INFINITY);
        float sc = (float)(((double)(hi + pad) - (double)g.origin[k]) / (double)rtl::GRID_UNITS);
        sc = nextafterf(nextafterf(sc, INFINITY), INFINITY);
        if (!(sc > 1e-30f)) sc = 1e-30f;
        g.scale[k] = sc; g.inv_scale[k] = 1.0f / sc; h.grid_origin[k] = g.origin[k]; h.grid_scale[k] = sc;
    }
    float root_area = ext[0] * ext[1] + ext[1] * ext[2] + ext[2] * ext[0];
    if (!std::isfinite(root_area)) root_area = 0.0f;

    // Temporary slots: [0, n) for the groups, [n, 2n) for the large nodes (there are far fewer, but how many is not known beforehand).
    const uint32_t slots = 2u * n;
    TempTree tmp;
    uint32_t* d_new_index;
    GPU_TRY(mem.alloc(&tmp.pairs, slots)); GPU_TRY(mem.alloc(&tmp.used, slots)); GPU_TRY(mem.alloc(&d_new_index, slots)); GPU_TRY(mem.alloc(&tmp.order_out, n));
    GPU_TRY(hipMemsetAsync(tmp.used, 0, (size_t)slots * sizeof(uint32_t), stream));
    BuildStats* d_stats; GPU_TRY(mem.alloc(&d_stats, 1));
    GPU_TRY(hipMemsetAsync(d_stats, 0, sizeof(BuildStats), stream));

    // 3. large nodes, a level per round
    const uint32_t max_nodes = n / (GROUP + 1u) + 1u;                  // nodes of a level are disjoint and hold more than GROUP shapes each
    const uint32_t max_chunks = n / CHUNK + max_nodes + 1u;
    BigNode *d_nodes_a, *d_nodes_b; uint32_t *d_idx, *d_idx_out, *d_n_chunks_of, *d_chunk_off, *d_next_index;
    Level lv{};
    GroupList groups;
    GPU_TRY(mem.alloc(&d_nodes_a, max_nodes + 2)); GPU_TRY(mem.alloc(&d_nodes_b, max_nodes + 2)); GPU_TRY(mem.alloc(&d_idx, n)); GPU_TRY(mem.alloc(&d_idx_out, n));
    GPU_TRY(mem.alloc(&d_n_chunks_of, max_nodes + 2)); GPU_TRY(mem.alloc(&d_chunk_off, max_nodes + 2)); GPU_TRY(mem.alloc(&d_next_index, max_nodes + 2));
    GPU_TRY(mem.alloc(&lv.bins, max_nodes)); GPU_TRY(mem.alloc(&lv.chunk_cnt, (size_t)max_chunks * 3u * BINS)); GPU_TRY(mem.alloc(&lv.chunk_left, max_chunks));
    GPU_TRY(mem.alloc(&lv.split, max_nodes)); GPU_TRY(mem.alloc(&lv.sides, 2 * (size_t)max_nodes)); GPU_TRY(mem.alloc(&lv.n_big, max_nodes + 2));
    GPU_TRY(mem.alloc(&groups.first, n)); GPU_TRY(mem.alloc(&groups.count, n)); GPU_TRY(mem.alloc(&groups.link, n)); GPU_TRY(mem.alloc(&groups.depth, n));
    size_t scan_bytes = 0;
    GPU_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_n_chunks_of, d_chunk_off, (int)(max_nodes + 1), stream));
    uint8_t* d_scan_temp; GPU_TRY(mem.alloc(&d_scan_temp, scan_bytes));
    // the permutation starts as the sorted order
    hipLaunchKernelGGL(iota, dim3(blocks_n), dim3(256), 0, stream, d_idx, d_idx_out, n);
    GPU_TRY(hipGetLastError());
    uint32_t n_nodes = 0, nodes_before = 0;
    if (n > GROUP) {
        BigNode root{};
        root.first = 0; root.count = n; root.link = NONE; root.depth = 0;
        for (int k = 0; k < 3; k++) { root.clo[k] = bounds.v[6 + k]; root.chi[k] = bounds.v[9 + k]; }
        GPU_TRY(hipMemcpyAsync(d_nodes_a, &root, sizeof(root), hipMemcpyHostToDevice, stream));
        GPU_TRY(hipStreamSynchronize(stream));                         // the source is on this function's stack
        n_nodes = 1;
    } else {
        const uint32_t one[4] = {0u, n, NONE, 0u};                      // the whole scene is one group
        GPU_TRY(hipMemcpyAsync(groups.first, &one[0], 4, hipMemcpyHostToDevice, stream)); GPU_TRY(hipMemcpyAsync(groups.count, &one[1], 4, hipMemcpyHostToDevice, stream));
        GPU_TRY(hipMemcpyAsync(groups.link, &one[2], 4, hipMemcpyHostToDevice, stream)); GPU_TRY(hipMemcpyAsync(groups.depth, &one[3], 4, hipMemcpyHostToDevice, stream));
        const uint32_t one_group = 1u;
        GPU_TRY(hipMemcpyAsync(&d_stats->n_groups, &one_group, 4, hipMemcpyHostToDevice, stream));
        GPU_TRY(hipStreamSynchronize(stream));                         // the sources are on this function's stack
    }
    BigNode *cur = d_nodes_a, *next = d_nodes_b;
    for (uint32_t round = 0; n_nodes > 0; round++) {
        if (round > in.depth_limit || nodes_before + n_nodes > n) return BUILD_TOO_DEEP;
        lv.nodes = cur; lv.n_nodes = n_nodes; lv.chunk_off = d_chunk_off;
        const uint32_t chunk_blocks = std::min<uint64_t>(max_chunks, (uint64_t)n / CHUNK + n_nodes);   // at most one partly filled chunk per node
        const uint32_t init_threads = std::max(n_nodes * 3u * BINS, n_nodes + 1u);
        hipLaunchKernelGGL(level_begin, dim3((init_threads + 255) / 256), dim3(256), 0, stream, lv, d_n_chunks_of);
        GPU_TRY(hipGetLastError());
        GPU_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan_temp, scan_bytes, d_n_chunks_of, d_chunk_off, (int)(n_nodes + 1), stream));
        hipLaunchKernelGGL(bin_chunks, dim3(chunk_blocks), dim3(256), 0, stream, lv, d_idx, d_slo, d_shi, in.depth_limit);
        GPU_TRY(hipGetLastError());
        hipLaunchKernelGGL(split_nodes, dim3((n_nodes + 3) / 4), dim3(256), 0, stream, lv, in.depth_limit);
        GPU_TRY(hipGetLastError());
        GPU_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan_temp, scan_bytes, lv.n_big, d_next_index, (int)(n_nodes + 1), stream));
        hipLaunchKernelGGL(move_chunks, dim3(chunk_blocks), dim3(256), 0, stream, lv, d_idx, d_idx_out, d_slo, d_shi);
        GPU_TRY(hipGetLastError());
        hipLaunchKernelGGL(copy_chunks, dim3(chunk_blocks), dim3(256), 0, stream, lv, d_idx_out, d_idx);
        GPU_TRY(hipGetLastError());
        hipLaunchKernelGGL(emit_nodes, dim3((n_nodes + 255) / 256), dim3(256), 0, stream, lv, d_next_index, next, groups, n + nodes_before, g, tmp, root_area, d_stats);
        GPU_TRY(hipGetLastError());
        uint32_t n_next = 0;
        GPU_TRY(hipMemcpyAsync(&n_next, d_next_index + n_nodes, 4, hipMemcpyDeviceToHost, stream));
        GPU_TRY(hipStreamSynchronize(stream));
        if (n_next > max_nodes) { *err = "internal: the GPU tree build made more large nodes than fit a level"; return BUILD_FAILED; }
        nodes_before += n_nodes;
        n_nodes = n_next;
        std::swap(cur, next);
    }
    uint32_t n_groups = 0;
    GPU_TRY(hipMemcpyAsync(&n_groups, &d_stats->n_groups, 4, hipMemcpyDeviceToHost, stream));
    GPU_TRY(hipStreamSynchronize(stream));
    if (n_groups == 0 || n_groups > n) { *err = "internal: the GPU tree build lost its groups"; return BUILD_FAILED; }

    // 4. small nodes
    uint32_t* d_order_p;
    GPU_TRY(mem.alloc(&d_order_p, n));
    hipLaunchKernelGGL(gather_by_index, dim3(blocks_n), dim3(256), 0, stream, d_idx, n, d_slo, d_shi, d_order, d_lo, d_hi, d_order_p);     // d_lo / d_hi are free again
    GPU_TRY(hipGetLastError());
    hipLaunchKernelGGL(build_groups, dim3(n_groups), dim3(64), 0, stream, groups, n_groups, (const float4*)d_lo, (const float4*)d_hi, (const uint32_t*)d_order_p, g, tmp,
                       in.depth_limit, root_area, d_stats);
    GPU_TRY(hipGetLastError());

    // 5. numbering of the slots in use
    size_t need_bytes = 0;
    GPU_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, need_bytes, tmp.used, d_new_index, (int)slots, stream));
    uint8_t* d_scan_temp2; GPU_TRY(mem.alloc(&d_scan_temp2, need_bytes));
    GPU_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan_temp2, need_bytes, tmp.used, d_new_index, (int)slots, stream));
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

    // 6. the blob
    h.magic = rtl::BLOB_MAGIC; h.version = rtl::BLOB_VERSION;
    h.n_pairs = n_pairs; h.n_tree_shapes = n; h.n_list_shapes = in.n_list; h.n_materials = in.n_mats; h.n_lights = in.n_lights;
    h.root_ref = rtl::inner_ref(root_ref);
    h.max_depth = stats.max_depth; h.finite_shading = in.finite_shading; h.tree_kinds = in.tree_kinds;
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
