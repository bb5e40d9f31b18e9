// bvh_build.cpp -- host builder of the GPU bounding volume hierarchy.
//
// Takes the place of Bvh::build_par (reference src/raytracer/raytracer.rs:32; the bvh crate
// builds a SAH binary tree with one shape per leaf).  In the reference the tree only selects
// candidates that are then tested exactly (raytracer.rs:110-120), so any tree whose culling
// is conservative gives the same image; this one is laid out for the GPU instead:
// binned-SAH binary tree, up to 8 shapes per leaf, stored as 64-byte child-pair records
// (layout.h), leaves contiguous in a reordered shape array.
#include "bvh_build.h"

#include <queue>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <omp.h>
#include <parallel/algorithm>

namespace rtb {

namespace {

constexpr int NBINS = 32;
// Cost model of the surface area heuristic.  Measured on the 4K dragon frame, ratios from 1:3 to
// 2:1 and leaf caps of 2, 4 and 8 all land within 12 % of each other; 1:1 with a cap of 8 is best (2.47 ms; a shape at
// 0.5 or 1.5 node pairs: 2.52 and 2.66; a cap of 2: 2.62).  RT_BVH_COST_SHAPE and RT_BVH_LEAF_CAP in the environment
// override the two for such measurements (scripts/tree_sweep.sh).
constexpr float COST_TRAVERSE = 1.0f;   // one node pair fetch + two slab tests
static const float COST_SHAPE = getenv("RT_BVH_COST_SHAPE") ? (float)atof(getenv("RT_BVH_COST_SHAPE")) : 1.0f;      // one exact shape test
static const uint32_t LEAF_CAP = getenv("RT_BVH_LEAF_CAP") ? std::min(std::max((uint32_t)atoi(getenv("RT_BVH_LEAF_CAP")), 1u), rtl::MAX_LEAF) : rtl::MAX_LEAF;
constexpr uint32_t PARALLEL_MIN = 8192; // subtrees above this size become OpenMP tasks
constexpr uint32_t BIG_NODE = 1u << 18; // nodes above this size are themselves binned and partitioned in parallel
constexpr uint32_t BIG_CHUNK = 1u << 15;
constexpr uint32_t REINSERT_MAX_SHAPES = 1u << 18;   // topology optimisation (Reinserter) up to this many shapes

struct Box {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<float>::infinity(); hi[k] = -lo[k]; } }
    void grow(const float* l, const float* h) {
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], l[k]); hi[k] = std::max(hi[k], h[k]); }
    }
    void grow(const Box& b) { grow(b.lo, b.hi); }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.0f) || !(dy >= 0.0f) || !(dz >= 0.0f)) return 0.0f;
        // Huge extents (a sphere of radius 1e4 next to unit spheres) overflow nothing in float,
        // but keep the arithmetic in double so that ratios stay meaningful.
        return (float)((double)dx * dy + (double)dy * dz + (double)dz * dx);
    }
};

struct Builder {
    const float* lo; const float* hi;      // n x 3 shape boxes
    uint32_t n;
    uint32_t depth_limit;
    float pad_abs;
    rtu::NoInitVector<uint32_t> idx;       // permutation being partitioned
    rtu::NoInitVector<float> ctr;          // n x 3 centroids
    rtu::NoInitVector<rtl::NodePair> pairs;
    std::atomic<uint32_t> next_pair{0};
    std::atomic<uint32_t> n_leaves{0}, max_depth{0}, max_leaf{0};

    rtu::NoInitVector<uint32_t> tmp;       // scratch of the parallel partition (same positions as idx)
    const uint32_t* perm = nullptr;        // builder position -> index of the caller's box (boxes are pre-sorted along a Morton curve)

    // Runs body(chunk, begin, end) over [first, first + count) in BIG_CHUNK pieces as OpenMP tasks and waits.
    template <class F>
    static void for_chunks(uint32_t first, uint32_t count, F body) {
        const uint32_t nchunks = (count + BIG_CHUNK - 1) / BIG_CHUNK;
#pragma omp taskloop grainsize(1) default(shared)
        for (uint32_t c = 0; c < nchunks; c++) {
            const uint32_t b = first + c * BIG_CHUNK, e = std::min(first + count, b + BIG_CHUNK);
            body(c, b, e);
        }
    }

    Box bounds(uint32_t first, uint32_t count) const {
        Box b; b.reset();
        if (count < BIG_NODE) {
            for (uint32_t i = first; i < first + count; i++) b.grow(lo + 3 * (size_t)idx[i], hi + 3 * (size_t)idx[i]);
            return b;
        }
        std::vector<Box> part((count + BIG_CHUNK - 1) / BIG_CHUNK);
        for_chunks(first, count, [&](uint32_t c, uint32_t s, uint32_t e) {
            Box l; l.reset();
            for (uint32_t i = s; i < e; i++) l.grow(lo + 3 * (size_t)idx[i], hi + 3 * (size_t)idx[i]);
            part[c] = l;
        });
        for (const Box& l : part) b.grow(l);
        return b;
    }

    void write_child(rtl::ChildBox& c, const Box& b, uint32_t ref, uint32_t count) {
        for (int k = 0; k < 3; k++) {
            // Outward padding keeps culling conservative: against rounding in the slab test, and
            // against exact shape tests that accept rays passing a rounding error outside the shape.
            float m = std::max(std::fabs(b.lo[k]), std::fabs(b.hi[k]));
            float pad = m * 4e-7f + pad_abs;
            c.lo[k] = b.lo[k] - pad; c.hi[k] = b.hi[k] + pad;
        }
        c.ref = ref; c.count = count;
    }

    static void atomic_max(std::atomic<uint32_t>& a, uint32_t v) {
        uint32_t cur = a.load(std::memory_order_relaxed);
        while (cur < v && !a.compare_exchange_weak(cur, v, std::memory_order_relaxed)) {}
    }

    uint32_t make_leaf(uint32_t first, uint32_t count, uint32_t depth) {
        n_leaves.fetch_add(1, std::memory_order_relaxed);
        atomic_max(max_depth, depth);
        atomic_max(max_leaf, count);
        // Shapes inside a leaf in input order, so that the result does not depend on how the threads interleaved.
        std::sort(idx.begin() + first, idx.begin() + first + count, [&](uint32_t a, uint32_t b) { return perm[a] < perm[b]; });
        return rtl::make_leaf_ref(first, count);
    }

    // Returns the split position (first < mid < first + count) or 0 when a leaf is cheaper.
    uint32_t choose_split(uint32_t first, uint32_t count, const Box& node_box, bool force_median) {
        const bool big = count >= BIG_NODE;
        const uint32_t nchunks = (count + BIG_CHUNK - 1) / BIG_CHUNK;
        Box cb; cb.reset();
        if (!big) {
            for (uint32_t i = first; i < first + count; i++) { const float* c = &ctr[3 * (size_t)idx[i]]; cb.grow(c, c); }
        } else {
            std::vector<Box> part(nchunks);
            for_chunks(first, count, [&](uint32_t c, uint32_t s, uint32_t e) {
                Box l; l.reset();
                for (uint32_t i = s; i < e; i++) { const float* p = &ctr[3 * (size_t)idx[i]]; l.grow(p, p); }
                part[c] = l;
            });
            for (const Box& l : part) cb.grow(l);
        }
        int axis_by_extent = 0;
        for (int k = 1; k < 3; k++) if (cb.hi[k] - cb.lo[k] > cb.hi[axis_by_extent] - cb.lo[axis_by_extent]) axis_by_extent = k;

        auto median_split = [&](int axis) -> uint32_t {
            uint32_t mid = first + count / 2;
            std::nth_element(idx.begin() + first, idx.begin() + mid, idx.begin() + first + count,
                             [&](uint32_t p, uint32_t q) {
                                 float a = ctr[3 * (size_t)p + axis], b = ctr[3 * (size_t)q + axis];
                                 return a != b ? a < b : perm[p] < perm[q];
                             });
            return mid;
        };
        if (force_median) return median_split(axis_by_extent);

        float parent_area = node_box.half_area();
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1, best_bin = -1;
        // One pass bins every shape on all three axes (the shape's box is fetched once).
        struct AxisBins { Box bb[3][NBINS]; uint32_t bc[3][NBINS]; };
        // Small nodes use fewer bins: most nodes of a tree hold a handful of shapes, and clearing and
        // sweeping 3 x 32 bins for each of them would cost more than everything else in the build.
        const int nbin = count <= 8 ? 4 : count <= 32 ? 8 : count <= 128 ? 16 : NBINS;
        float c0v[3], scalev[3]; bool usable[3];
        for (int axis = 0; axis < 3; axis++) {
            c0v[axis] = cb.lo[axis];
            usable[axis] = cb.hi[axis] > cb.lo[axis];
            scalev[axis] = usable[axis] ? (float)nbin / (cb.hi[axis] - cb.lo[axis]) : 0.0f;
        }
        auto clear_bins = [nbin](AxisBins& a) { for (int k = 0; k < 3; k++) for (int b = 0; b < nbin; b++) { a.bb[k][b].reset(); a.bc[k][b] = 0; } };
        auto bin_range = [&](uint32_t s0, uint32_t e0, AxisBins& a) {
            for (uint32_t i = s0; i < e0; i++) {
                const uint32_t s = idx[i];
                const float* pl = lo + 3 * (size_t)s; const float* ph = hi + 3 * (size_t)s; const float* pc = &ctr[3 * (size_t)s];
                for (int axis = 0; axis < 3; axis++) {
                    if (!usable[axis]) continue;
                    int b = (int)((pc[axis] - c0v[axis]) * scalev[axis]);
                    b = std::min(std::max(b, 0), nbin - 1);
                    a.bb[axis][b].grow(pl, ph); a.bc[axis][b]++;
                }
            }
        };
        AxisBins bins; clear_bins(bins);
        if (!big) {
            bin_range(first, first + count, bins);
        } else {
            std::vector<AxisBins> part(nchunks);
            for_chunks(first, count, [&](uint32_t c, uint32_t s0, uint32_t e0) { clear_bins(part[c]); bin_range(s0, e0, part[c]); });
            for (const AxisBins& a : part)
                for (int k = 0; k < 3; k++)
                    for (int b = 0; b < nbin; b++) if (a.bc[k][b]) { bins.bb[k][b].grow(a.bb[k][b]); bins.bc[k][b] += a.bc[k][b]; }
        }
        for (int axis = 0; axis < 3; axis++) {
            if (!usable[axis]) continue;
            const Box* bb = bins.bb[axis]; const uint32_t* bc = bins.bc[axis];
            float right_area[NBINS]; uint32_t right_cnt[NBINS];
            Box acc; acc.reset(); uint32_t cnt = 0;
            for (int b = nbin - 1; b >= 1; b--) { acc.grow(bb[b]); cnt += bc[b]; right_area[b] = acc.half_area(); right_cnt[b] = cnt; }
            acc.reset(); cnt = 0;
            for (int b = 0; b < nbin - 1; b++) {
                acc.grow(bb[b]); cnt += bc[b];
                if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                float cost = acc.half_area() * (float)cnt + right_area[b + 1] * (float)right_cnt[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
            }
        }
        if (best_axis < 0) {
            // All centroids coincide: nothing to bin.  Keep a leaf if it fits, else halve by index.
            if (count <= LEAF_CAP) return 0;
            return first + count / 2;
        }
        float split_cost = COST_TRAVERSE + COST_SHAPE * (parent_area > 0.0f ? best_cost / parent_area : (float)count);
        float leaf_cost = COST_SHAPE * (float)count;
        if (count <= LEAF_CAP && leaf_cost <= split_cost) return 0;
        const float c0 = c0v[best_axis], scale = scalev[best_axis];
        auto goes_left = [&](uint32_t s) {
            int b = (int)((ctr[3 * (size_t)s + best_axis] - c0) * scale);
            b = std::min(std::max(b, 0), nbin - 1);
            return b <= best_bin;
        };
        uint32_t mid;
        if (!big) {
            auto it = std::partition(idx.begin() + first, idx.begin() + first + count, goes_left);
            mid = (uint32_t)(it - idx.begin());
        } else {
            // Count per chunk, prefix, scatter into the scratch array, copy back: all in chunk-parallel passes.
            std::vector<uint32_t> nleft(nchunks), lpos(nchunks), rpos(nchunks);
            for_chunks(first, count, [&](uint32_t c, uint32_t s0, uint32_t e0) {
                uint32_t k = 0;
                for (uint32_t i = s0; i < e0; i++) k += goes_left(idx[i]) ? 1u : 0u;
                nleft[c] = k;
            });
            uint32_t total_left = 0;
            for (uint32_t c = 0; c < nchunks; c++) { lpos[c] = total_left; total_left += nleft[c]; }
            uint32_t racc = 0;
            for (uint32_t c = 0; c < nchunks; c++) {
                const uint32_t len = std::min(first + count, first + (c + 1) * BIG_CHUNK) - (first + c * BIG_CHUNK);
                rpos[c] = total_left + racc; racc += len - nleft[c];
            }
            for_chunks(first, count, [&](uint32_t c, uint32_t s0, uint32_t e0) {
                uint32_t l = first + lpos[c], r = first + rpos[c];
                for (uint32_t i = s0; i < e0; i++) { const uint32_t s = idx[i]; if (goes_left(s)) tmp[l++] = s; else tmp[r++] = s; }
            });
            for_chunks(first, count, [&](uint32_t, uint32_t s0, uint32_t e0) {
                std::memcpy(&idx[s0], &tmp[s0], (size_t)(e0 - s0) * sizeof(uint32_t));
            });
            mid = first + total_left;
        }
        if (mid == first || mid == first + count) return median_split(axis_by_extent);
        return mid;
    }

    static double area_of(const rtl::ChildBox& c) {
        const double dx = (double)c.hi[0] - c.lo[0], dy = (double)c.hi[1] - c.lo[1], dz = (double)c.hi[2] - c.lo[2];
        return dx * dy + dy * dz + dz * dx;
    }

    // Builds the subtree over idx[first, first+count) and returns its reference; *cost receives the surface-area cost
    // of the subtree times `area`, the surface measure of the box it is stored with (summed along the tree, so the
    // value does not depend on how the threads interleaved).
    uint32_t build(uint32_t first, uint32_t count, uint32_t depth, const Box& node_box, double area, double* cost) {
        *cost = (double)COST_SHAPE * (double)count * area;          // as a leaf
        if (count == 1) return make_leaf(first, 1, depth);
        // Depth budget: a balanced subtree over `count` shapes needs ceil(log2(count / MAX_LEAF)) more levels.
        uint32_t need = 0; while (((uint64_t)rtl::MAX_LEAF << need) < count) need++;
        bool force_median = depth + need + 1 >= depth_limit;
        if (force_median && count <= rtl::MAX_LEAF) return make_leaf(first, count, depth);
        uint32_t mid = choose_split(first, count, node_box, force_median);
        if (mid == 0) return make_leaf(first, count, depth);
        uint32_t me = next_pair.fetch_add(1, std::memory_order_relaxed);
        uint32_t nl = mid - first, nr = first + count - mid;
        Box bl = bounds(first, nl), br = bounds(mid, nr);
        rtl::NodePair& p = pairs[me];
        write_child(p.c[0], bl, 0, 0);
        write_child(p.c[1], br, 0, 0);
        const double al = area_of(p.c[0]), ar = area_of(p.c[1]);
        uint32_t rl = 0, rr = 0;
        double cl = 0.0, cr = 0.0;
        if (count >= PARALLEL_MIN) {
#pragma omp task shared(rl, cl) firstprivate(first, nl, depth, bl, al)
            rl = build(first, nl, depth + 1, bl, al, &cl);
#pragma omp task shared(rr, cr) firstprivate(mid, nr, depth, br, ar)
            rr = build(mid, nr, depth + 1, br, ar, &cr);
#pragma omp taskwait
        } else {
            rl = build(first, nl, depth + 1, bl, al, &cl);
            rr = build(mid, nr, depth + 1, br, ar, &cr);
        }
        p.c[0].ref = rl; p.c[0].count = (rl & rtl::REF_LEAF) ? (rl & 7u) + 1u : 0u;
        p.c[1].ref = rr; p.c[1].count = (rr & rtl::REF_LEAF) ? (rr & 7u) + 1u : 0u;
        *cost = (double)COST_TRAVERSE * area + cl + cr;
        return me;
    }
};


// ------------------------------------------------------------------ topology optimisation by reinsertion
// The top-down build decides every split with what it knows at that node.  Afterwards the tree can still be improved
// by taking subtrees out and putting them back where they enlarge the boxes above them least (Bittner, Hapala,
// Havran: "Fast insertion-based optimization of bounding volume hierarchies", 2013): pick the inner nodes that look
// worst (large, with children much smaller than themselves), remove such a node, which frees it and its parent and
// leaves its two children as loose subtrees, and reinsert each subtree at the position found by a branch-and-bound
// search over the tree for the smallest increase of the summed surface area.  Leaves are never opened, so the
// order of the shapes stays as the build left it.  Depth is kept within the limit of the traversal stacks.
struct Reinserter {
    static constexpr uint32_t NONE = 0xFFFFFFFFu;
    uint32_t n_inner = 0, root = NONE, depth_limit = 0;
    uint32_t undone = 0;                  // reinsertions taken back for want of a place within the depth limit
    uint32_t refuse_every = 0, searches = 0;   // tests: every refuse_every-th search for a place is made to fail
    std::vector<Box> box;                 // inner nodes 0 .. n_inner-1, leaves from n_inner on
    std::vector<uint32_t> parent, left, right, leaf_ref, height;     // height: node pairs on the longest way down (0 for a leaf)
    std::vector<double> area;

    bool inner(uint32_t v) const { return v < n_inner; }
    static double area_of_box(const Box& b) { return (double)b.half_area(); }

    void load(const rtu::NoInitVector<rtl::NodePair>& pairs, uint32_t root_ref) {
        n_inner = (uint32_t)pairs.size();
        const uint32_t total = 2 * n_inner + 1;
        box.resize(total); parent.assign(total, NONE); left.assign(n_inner, NONE); right.assign(n_inner, NONE);
        leaf_ref.assign(total, 0u); height.assign(total, 0u); area.assign(total, 0.0);
        uint32_t next_leaf = n_inner;
        for (uint32_t i = 0; i < n_inner; i++)
            for (int c = 0; c < 2; c++) {
                const rtl::ChildBox& cb = pairs[i].c[c];
                uint32_t v;
                if (cb.ref & rtl::REF_LEAF) { v = next_leaf++; leaf_ref[v] = cb.ref; } else v = cb.ref;
                for (int k = 0; k < 3; k++) { box[v].lo[k] = cb.lo[k]; box[v].hi[k] = cb.hi[k]; }
                parent[v] = i;
                (c == 0 ? left : right)[i] = v;
            }
        root = root_ref;
        box[root].reset(); box[root].grow(box[left[root]]); box[root].grow(box[right[root]]);
        for (uint32_t v = 0; v < total; v++) area[v] = area_of_box(box[v]);
        // heights bottom-up: children of pair i may have any index, so iterate over a depth-first order
        std::vector<uint32_t> order; order.reserve(n_inner);
        std::vector<uint32_t> stack{root};
        while (!stack.empty()) { uint32_t v = stack.back(); stack.pop_back(); order.push_back(v); if (inner(left[v])) stack.push_back(left[v]); if (inner(right[v])) stack.push_back(right[v]); }
        for (size_t i = order.size(); i-- > 0;) { uint32_t v = order[i]; height[v] = 1u + std::max(height[left[v]], height[right[v]]); }
    }

    // Boxes, areas and heights from v upwards.
    void refit(uint32_t v) {
        for (; v != NONE; v = parent[v]) {
            Box b; b.reset(); b.grow(box[left[v]]); b.grow(box[right[v]]);
            box[v] = b; area[v] = area_of_box(b);
            height[v] = 1u + std::max(height[left[v]], height[right[v]]);
        }
    }

    // Where the loose subtree t costs least: returns the node it should become the sibling of.
    uint32_t best_position(uint32_t t) const {
        struct Item { double induced; uint32_t node, depth; bool operator<(const Item& o) const { return induced > o.induced; } };
        std::priority_queue<Item> queue;
        queue.push({0.0, root, 0u});
        double best = std::numeric_limits<double>::infinity();
        uint32_t best_node = NONE;
        const double at = area[t];
        while (!queue.empty()) {
            const Item it = queue.top(); queue.pop();
            if (it.induced + at >= best) break;
            Box u = box[it.node]; u.grow(box[t]);
            const double direct = area_of_box(u), total = it.induced + direct;
            // the new pair sits where it.node sat: everything below moves one level down
            const bool fits = it.depth + 1u + std::max(height[it.node], height[t]) <= depth_limit;
            if (fits && total < best) { best = total; best_node = it.node; }
            if (inner(it.node)) {
                const double below = total - area[it.node];
                if (below + at < best) { queue.push({below, left[it.node], it.depth + 1u}); queue.push({below, right[it.node], it.depth + 1u}); }
            }
        }
        return best_node;
    }

    void replace_child(uint32_t p, uint32_t old_child, uint32_t new_child) {
        if (left[p] == old_child) left[p] = new_child; else right[p] = new_child;
    }

    // The free node f becomes the parent of x and t, in x's place.
    void attach(uint32_t f, uint32_t x, uint32_t t) {
        const uint32_t xp = parent[x];
        left[f] = x; right[f] = t; parent[f] = xp; parent[x] = f; parent[t] = f;
        if (xp == NONE) root = f; else replace_child(xp, x, f);
        refit(f);
    }

    // Undoes attach(f, x, t): x takes f's place again, f and t are loose.
    void detach(uint32_t f) {
        const uint32_t x = left[f], fp = parent[f];
        parent[x] = fp;
        if (fp == NONE) root = x; else { replace_child(fp, f, x); refit(fp); }
    }

    // Takes inner node v out, reinserts its two children.  Returns false, with the tree as it was, when v sits too close to
    // the root or when one of the children finds no place that keeps the tree within the depth limit (the first always
    // does, its old place; the second need not once the first sits somewhere else).
    bool reinsert_children_of(uint32_t v) {
        const uint32_t p = parent[v];
        if (v == root || p == NONE || p == root) return false;
        const uint32_t g = parent[p], s = left[p] == v ? right[p] : left[p];
        const uint32_t child[2] = {left[v], right[v]};
        uint32_t loose[2] = {child[0], child[1]}, spare[2] = {v, p};
        if (area[loose[0]] < area[loose[1]]) std::swap(loose[0], loose[1]);          // the larger one first
        replace_child(g, p, s); parent[s] = g;
        refit(g);
        for (int k = 0; k < 2; k++) {
            uint32_t x = best_position(loose[k]);
            if (refuse_every != 0 && ++searches % refuse_every == 0) x = NONE;       // (tests: the rare case on demand)
            if (x == NONE) {                                                         // put everything back
                undone++;
                if (k == 1) detach(spare[0]);
                left[v] = child[0]; right[v] = child[1]; parent[child[0]] = v; parent[child[1]] = v; parent[v] = NONE;
                refit(v);
                attach(p, s, v);
                return false;
            }
            attach(spare[k], x, loose[k]);
        }
        return true;
    }

    // One sweep: every inner node once, in the order of decreasing badness as it was when the sweep began.
    void run(uint32_t sweeps) {
        std::vector<std::pair<double, uint32_t>> scored;
        for (uint32_t sweep = 0; sweep < sweeps; sweep++) {
            scored.clear();
            for (uint32_t v = 0; v < n_inner; v++) {
                if (v == root || parent[v] == root) continue;
                const double a = area[v], al = area[left[v]], ar = area[right[v]];
                const double smallest = std::max(std::min(al, ar), 1e-300), mean = std::max(0.5 * (al + ar), 1e-300);
                scored.push_back({a * (a / smallest) * (a / mean), v});
            }
            std::sort(scored.begin(), scored.end(),
                      [](const std::pair<double, uint32_t>& x, const std::pair<double, uint32_t>& y) { return x.first != y.first ? x.first > y.first : x.second < y.second; });
            // every node in the first sweep, the worse half in the later ones (where most of what is left to gain sits)
            const size_t take = sweep == 0 ? scored.size() : scored.size() / 2;
            for (size_t i = 0; i < take; i++) reinsert_children_of(scored[i].second);
        }
    }

    // Back into node pairs, numbered in depth-first order from the root (pair 0).
    void store(BvhResult* out) const {
        rtu::NoInitVector<rtl::NodePair> pairs(n_inner);
        std::vector<uint32_t> number(n_inner, NONE);
        uint32_t next = 0;
        std::vector<uint32_t> stack{root};
        while (!stack.empty()) {
            const uint32_t v = stack.back(); stack.pop_back();
            number[v] = next++;
            if (inner(right[v])) stack.push_back(right[v]);
            if (inner(left[v])) stack.push_back(left[v]);
        }
        double cost = 0.0;
        for (uint32_t v = 0; v < n_inner; v++) {
            if (number[v] == NONE) continue;
            rtl::NodePair& p = pairs[number[v]];
            // The child with the smaller box first: shadow rays search the children in stored order and stop at the first
            // occluder, which the denser child yields sooner (1 % fewer box tests on the dragon frame than the other way round).
            uint32_t kids[2] = {left[v], right[v]};
            if (area[kids[0]] > area[kids[1]]) std::swap(kids[0], kids[1]);
            for (int c = 0; c < 2; c++) {
                const uint32_t k = kids[c];
                for (int a = 0; a < 3; a++) { p.c[c].lo[a] = box[k].lo[a]; p.c[c].hi[a] = box[k].hi[a]; }
                if (inner(k)) { p.c[c].ref = number[k]; p.c[c].count = 0; }
                else { p.c[c].ref = leaf_ref[k]; p.c[c].count = (leaf_ref[k] & 7u) + 1u; cost += (double)COST_SHAPE * p.c[c].count * area[k]; }
            }
            cost += (double)COST_TRAVERSE * area[v];
        }
        out->pairs.swap(pairs);
        out->root_ref = 0;
        out->max_depth = height[root];
        out->sah_cost = area[root] > 0.0 ? cost / area[root] : 0.0;
    }
};

}  // namespace

bool build_bvh(const float* lo, const float* hi, uint32_t n, uint32_t depth_limit, float pad_abs, BvhResult* out, std::string* err) {
    out->pairs.clear(); out->order.clear();
    out->root_ref = rtl::REF_EMPTY; out->max_depth = 0; out->max_leaf = 0; out->n_leaves = 0; out->sah_cost = 0.0;
    if (n == 0) return true;
    if (n > rtl::MAX_TREE_SHAPES) { *err = "too many shapes for the tree's leaf references"; return false; }
    if (depth_limit < 8) { *err = "depth limit too small"; return false; }
    {
        uint32_t need = 0; while (((uint64_t)rtl::MAX_LEAF << need) < n) need++;
        if (need + 2 > depth_limit) { *err = "scene too large for the traversal stack depth"; return false; }
    }
    // Pre-sort the boxes along a Morton curve and build over the sorted copies: every subtree then works
    // on a compact stretch of memory instead of gathering its boxes from all over the input.
    rtu::NoInitVector<uint32_t> perm(n);
    rtu::NoInitVector<float> slo(3 * (size_t)n), shi(3 * (size_t)n);
    {
        const double inf = std::numeric_limits<double>::infinity();
        double clo0 = inf, clo1 = inf, clo2 = inf, chi0 = -inf, chi1 = -inf, chi2 = -inf;
#pragma omp parallel for schedule(static) reduction(min : clo0, clo1, clo2) reduction(max : chi0, chi1, chi2) if (n >= BIG_NODE)
        for (int64_t i = 0; i < (int64_t)n; i++) {
            const double c0 = 0.5 * ((double)lo[3 * (size_t)i] + (double)hi[3 * (size_t)i]), c1 = 0.5 * ((double)lo[3 * (size_t)i + 1] + (double)hi[3 * (size_t)i + 1]),
                         c2 = 0.5 * ((double)lo[3 * (size_t)i + 2] + (double)hi[3 * (size_t)i + 2]);
            clo0 = std::min(clo0, c0); chi0 = std::max(chi0, c0); clo1 = std::min(clo1, c1); chi1 = std::max(chi1, c1); clo2 = std::min(clo2, c2); chi2 = std::max(chi2, c2);
        }
        const double clo[3] = {clo0, clo1, clo2}, chi[3] = {chi0, chi1, chi2};
        rtu::NoInitVector<uint64_t> keyed(n);
#pragma omp parallel for schedule(static) if (n >= BIG_NODE)
        for (int64_t i = 0; i < (int64_t)n; i++) {
            uint64_t code = 0;
            uint32_t q[3];
            for (int k = 0; k < 3; k++) {
                const double c = 0.5 * ((double)lo[3 * (size_t)i + k] + (double)hi[3 * (size_t)i + k]);
                const double ext = chi[k] - clo[k];
                double u = ext > 0 ? (c - clo[k]) / ext : 0.0;
                u = std::min(std::max(u, 0.0), 1.0);
                q[k] = (uint32_t)(u * 1023.0);
            }
            for (int bit = 9; bit >= 0; bit--)
                for (int k = 0; k < 3; k++) code = (code << 1) | ((q[k] >> bit) & 1u);
            keyed[(size_t)i] = (code << 32) | (uint64_t)i;           // ties keep input order
        }
        if (n >= BIG_NODE) __gnu_parallel::sort(keyed.begin(), keyed.end()); else std::sort(keyed.begin(), keyed.end());
#pragma omp parallel for schedule(static) if (n >= BIG_NODE)
        for (int64_t i = 0; i < (int64_t)n; i++) {
            const uint32_t src = (uint32_t)(keyed[(size_t)i] & 0xFFFFFFFFull);
            perm[(size_t)i] = src;
            for (int k = 0; k < 3; k++) { slo[3 * (size_t)i + k] = lo[3 * (size_t)src + k]; shi[3 * (size_t)i + k] = hi[3 * (size_t)src + k]; }
        }
    }
    Builder b;
    b.lo = slo.data(); b.hi = shi.data(); b.n = n; b.depth_limit = depth_limit; b.pad_abs = pad_abs; b.perm = perm.data();
    b.idx.resize(n); b.ctr.resize(3 * (size_t)n);
    if (n >= BIG_NODE) b.tmp.resize(n);
#pragma omp parallel for schedule(static) if (n >= BIG_NODE)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        b.idx[(size_t)i] = (uint32_t)i;
        for (int k = 0; k < 3; k++) b.ctr[3 * (size_t)i + k] = 0.5f * slo[3 * (size_t)i + k] + 0.5f * shi[3 * (size_t)i + k];
    }
    b.pairs.resize(n > 1 ? n - 1 : 1);
    Box root; root.reset();
    uint32_t root_ref = 0;
    double root_cost = 0.0, root_area = 0.0;
#pragma omp parallel
#pragma omp single
    {
        root = b.bounds(0, n);
        root_area = root.half_area();
        root_ref = b.build(0, n, 0, root, root_area, &root_cost);
    }
    b.pairs.resize(b.next_pair.load());
    out->pairs.swap(b.pairs);
    out->order.resize(n);
#pragma omp parallel for schedule(static) if (n >= BIG_NODE)
    for (int64_t i = 0; i < (int64_t)n; i++) out->order[(size_t)i] = perm[b.idx[(size_t)i]];
    out->root_ref = root_ref;
    out->max_depth = b.max_depth.load(); out->max_leaf = b.max_leaf.load(); out->n_leaves = b.n_leaves.load();
    out->sah_cost = root_area > 0.0 ? root_cost / root_area : 0.0;
    // Topology optimisation for trees of a size where it takes about a second (bigger scenes are built on the GPU): three
    // sweeps.  On the 100 000-triangle dragon: 0.8 s, surface-area cost 43.5 -> 42.2, 4.5 % fewer boxes tested per frame
    // and 10 % fewer wavefront passes (the longest rays gain most).  RT_BVH_REINSERT=<sweeps> in the environment
    // overrides the number (0: none).
    static const int sweeps = getenv("RT_BVH_REINSERT") ? atoi(getenv("RT_BVH_REINSERT")) : 3;
    if (sweeps > 0 && !(root_ref & rtl::REF_LEAF) && out->pairs.size() >= 8 && n <= REINSERT_MAX_SHAPES) {
        Reinserter r;
        r.depth_limit = out->max_depth;               // no deeper than the build made it: the traversal stacks are sized by it
        if (const char* refuse = getenv("RT_BVH_TEST_REFUSE_EVERY")) r.refuse_every = (uint32_t)atoi(refuse);
        r.load(out->pairs, root_ref);
        r.store(out);                                 // numbered from the root down, whatever order the build's threads took their pairs in
        r.load(out->pairs, out->root_ref);
        r.run((uint32_t)sweeps);
        r.store(out);
        if (getenv("RT_BVH_TRACE")) fprintf(stderr, "reinsertion: %u shapes, depth limit %u, %u reinsertions taken back\n", n, r.depth_limit, r.undone);
    }
    return true;
}

}  // namespace rtb
