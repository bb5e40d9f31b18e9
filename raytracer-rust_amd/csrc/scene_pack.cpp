// scene_pack.cpp -- flattened scene (rt_scene) -> device blob (layout.h).
//
// Host half of RayTracer::new (reference src/raytracer/raytracer.rs:29-38): computes the
// shape boxes the way shape.rs:205-234 does, builds the tree (bvh_build.cpp) and writes
// node pairs, reordered shape records, the material table and the lights into one buffer.
#include "scene_pack.h"

#include <omp.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <unordered_map>

#include "bvh_build.h"

namespace rtp {

namespace {

struct MatKey {
    uint32_t w[7];
    bool operator==(const MatKey& o) const { return std::memcmp(w, o.w, sizeof(w)) == 0; }
};
struct MatHash {
    size_t operator()(const MatKey& k) const {
        uint64_t h = 1469598103934665603ull;
        for (int i = 0; i < 7; i++) { h ^= k.w[i]; h *= 1099511628211ull; }
        return (size_t)h;
    }
};

struct Flat {              // one shape in file order
    uint32_t kind, idx_in_kind, order, mat;
};

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

void fill_record(const rt_scene& s, const Flat& f, rtl::ShapeRec* r) {
    std::memset(r, 0, sizeof(*r));
    r->kind_mat = f.kind | (f.mat << 2);
    r->order = f.order;
    size_t i = f.idx_in_kind;
    if (f.kind == rtl::KIND_TRIANGLE) {
        for (int k = 0; k < 3; k++) { r->q0[k] = s.tri_v0[3 * i + k]; r->q1[k] = s.tri_v1[3 * i + k]; r->q2[k] = s.tri_v2[3 * i + k]; }
    } else if (f.kind == rtl::KIND_SPHERE) {
        for (int k = 0; k < 3; k++) r->q0[k] = s.sph_center[3 * i + k];
        r->q1[0] = s.sph_radius[i];
    } else {
        for (int k = 0; k < 3; k++) { r->q0[k] = s.pl_point[3 * i + k]; r->q2[k] = s.pl_normal[3 * i + k]; }
    }
}

}  // namespace

uint32_t kinds_of(const rtl::ShapeRec* records, size_t n) {
    uint32_t kinds = 0;
#pragma omp parallel for schedule(static) reduction(| : kinds) if (n > 100000)
    for (int64_t i = 0; i < (int64_t)n; i++) kinds |= 1u << (records[i].kind_mat & 3u);
    return kinds;
}

bool validate_header(const rtl::BlobHeader& h, size_t bytes, std::string* err) {
    if (h.magic != rtl::BLOB_MAGIC || h.version != rtl::BLOB_VERSION) { *err = "scene blob: bad magic or version"; return false; }
    if (h.total_bytes != bytes) { *err = "scene blob: size does not match its header"; return false; }
    // (no sum or product that could wrap: offsets and counts come from the file)
    auto fits = [&](uint64_t off, uint64_t n, uint64_t each) { return off % 16 == 0 && off >= sizeof(rtl::BlobHeader) && off <= bytes && n <= (bytes - off) / each; };
    if (!fits(h.off_nodes, h.n_pairs, sizeof(rtl::QuantPair)) || !fits(h.off_tree_shapes, h.n_tree_shapes, sizeof(rtl::ShapeRec)) ||
        !fits(h.off_list_shapes, h.n_list_shapes, sizeof(rtl::ShapeRec)) || !fits(h.off_materials, h.n_materials, sizeof(rtl::MaterialRec)) ||
        !fits(h.off_lights, h.n_lights, sizeof(rtl::LightRec))) {
        *err = "scene blob: section outside the buffer";
        return false;
    }
    if (h.n_pairs >= (1u << 26)) { *err = "scene blob: too many node pairs for 31-bit offsets"; return false; }
    if (h.n_tree_shapes > rtl::MAX_TREE_SHAPES) { *err = "scene blob: too many tree shapes for 32-bit offsets"; return false; }
    if (h.max_depth > TREE_DEPTH_LIMIT) { *err = "scene blob: tree deeper than the traversal stack"; return false; }
    if (h.root_ref != rtl::REF_EMPTY) {
        if (h.root_ref & rtl::REF_LEAF) {
            uint32_t first = (h.root_ref & ~rtl::REF_LEAF) >> 3, cnt = (h.root_ref & 7u) + 1u;
            if ((uint64_t)first + cnt > h.n_tree_shapes) { *err = "scene blob: root leaf outside the shape array"; return false; }
        } else if ((h.root_ref & 31u) || (h.root_ref >> 5) >= h.n_pairs) { *err = "scene blob: root outside the node array"; return false; }
    }
    return true;
}

// Full check of a blob that comes from outside the builder (a cache file): beyond the header,
// every reference the kernels follow must stay inside its array and the tree must be a tree
// (every pair reached exactly once, no deeper than the traversal stack), so that a damaged file
// yields an error instead of a GPU fault or an endless traversal.
bool validate_blob(const uint8_t* blob, size_t bytes, std::string* err) {
    if (!blob || bytes < sizeof(rtl::BlobHeader)) { *err = "scene blob: smaller than its header"; return false; }
    rtl::BlobHeader h; std::memcpy(&h, blob, sizeof(h));
    if (!validate_header(h, bytes, err)) return false;
    for (int k = 0; k < 3; k++)
        if (!std::isfinite(h.grid_origin[k]) || !std::isfinite(h.grid_scale[k])) { *err = "scene blob: node grid is not finite"; return false; }
    auto check_shapes = [&](uint64_t off, uint32_t n) {
        const rtl::ShapeRec* r = (const rtl::ShapeRec*)(blob + off);
        bool ok = true;
#pragma omp parallel for schedule(static) reduction(&& : ok) if (n > 100000)
        for (int64_t i = 0; i < (int64_t)n; i++) ok = ok && (r[i].kind_mat & 3u) <= rtl::KIND_PLANE && (r[i].kind_mat >> 2) < h.n_materials;
        return ok;
    };
    if (!check_shapes(h.off_tree_shapes, h.n_tree_shapes) || !check_shapes(h.off_list_shapes, h.n_list_shapes)) {
        *err = "scene blob: shape with an unknown kind or material";
        return false;
    }
    // (the header's word chooses the kernel: a tree said to hold only triangles is tested as triangles)
    if (h.tree_kinds != kinds_of((const rtl::ShapeRec*)(blob + h.off_tree_shapes), h.n_tree_shapes)) { *err = "scene blob: kinds of the tree's shapes differ from the header"; return false; }
    if (h.root_ref == rtl::REF_EMPTY || (h.root_ref & rtl::REF_LEAF)) {
        if (h.n_pairs != 0) { *err = "scene blob: node pairs without a root"; return false; }
        return true;
    }
    const rtl::QuantPair* pairs = (const rtl::QuantPair*)(blob + h.off_nodes);
    std::vector<uint8_t> seen(h.n_pairs, 0);
    std::vector<std::pair<uint32_t, uint32_t>> stack;   // pair index, depth
    stack.push_back({h.root_ref >> 5, 1});
    uint64_t reached = 0;
    while (!stack.empty()) {
        auto [idx, depth] = stack.back(); stack.pop_back();
        if (seen[idx]) { *err = "scene blob: node pair referenced twice"; return false; }
        seen[idx] = 1; reached++;
        if (depth > h.max_depth || depth > TREE_DEPTH_LIMIT) { *err = "scene blob: tree deeper than its header says"; return false; }
        for (int c = 0; c < 2; c++) {
            uint32_t ref = pairs[idx].c[c].ref;
            if (ref == rtl::REF_EMPTY) continue;
            if (ref & rtl::REF_LEAF) {
                uint32_t first = (ref & ~rtl::REF_LEAF) >> 3, cnt = (ref & 7u) + 1u;
                if ((uint64_t)first + cnt > h.n_tree_shapes) { *err = "scene blob: leaf outside the shape array"; return false; }
            } else {
                if ((ref & 31u) || (ref >> 5) >= h.n_pairs) { *err = "scene blob: child outside the node array"; return false; }
                stack.push_back({ref >> 5, depth + 1});
            }
        }
    }
    if (reached != h.n_pairs) { *err = "scene blob: unreachable node pairs"; return false; }
    return true;
}

bool split_scene(const rt_scene& s, SceneParts* out, std::string* err) {
    if ((s.ntri && (!s.tri_v0 || !s.tri_v1 || !s.tri_v2 || !s.tri_mat)) || (s.nsph && (!s.sph_center || !s.sph_radius || !s.sph_mat)) ||
        (s.npl && (!s.pl_point || !s.pl_normal || !s.pl_mat)) || (s.nlights && !s.lights)) {
        *err = "rt_scene: a non-empty array is NULL";
        return false;
    }
    const uint64_t total = (uint64_t)s.ntri + s.nsph + s.npl;
    if (total > rtl::MAX_TREE_SHAPES) { *err = "rt_scene: too many shapes"; return false; }
    const uint32_t bounded = s.ntri + s.nsph;
    out->use_tree = bounded > SMALL_SCENE;

    // Material ids in file-independent but deterministic order (triangles, spheres, planes).  Shapes copy
    // the parser's current material (config_builder.rs:401-403): runs of equal materials are the rule,
    // so the table is only consulted where a run ends.
    std::unordered_map<MatKey, uint32_t, MatHash> mat_ids;
    rtu::NoInitVector<uint32_t> tri_ids(s.ntri), sph_ids(s.nsph), pl_ids(s.npl);
    auto assign_ids = [&](const float* mats, uint32_t n, uint32_t* ids) {
        if (n == 0) return;
        // where a run of equal materials starts: all threads, each over one stretch, stretches in order
        const int nthreads = n > 100000 ? omp_get_max_threads() : 1;
        std::vector<std::vector<uint32_t>> found((size_t)nthreads);
#pragma omp parallel num_threads(nthreads)
        {
            const int t = omp_get_thread_num(), nt = omp_get_num_threads();
            const uint32_t a = (uint32_t)((uint64_t)n * (uint64_t)t / (uint64_t)nt), b = (uint32_t)((uint64_t)n * (uint64_t)(t + 1) / (uint64_t)nt);
            if (t < nthreads)
                for (uint32_t i = a; i < b; i++)
                    if (i == 0 || std::memcmp(mats + 7 * (size_t)i, mats + 7 * (size_t)(i - 1), 28) != 0) found[(size_t)t].push_back(i);
        }
        std::vector<uint32_t> starts, start_id;
        for (const auto& f : found) starts.insert(starts.end(), f.begin(), f.end());
        start_id.reserve(starts.size());
        for (uint32_t i : starts) {                          // in input order, so that the table does not depend on the threads
            const float* m = mats + 7 * (size_t)i;
            MatKey k; std::memcpy(k.w, m, sizeof(k.w));
            auto it = mat_ids.find(k);
            if (it == mat_ids.end()) {
                rtl::MaterialRec r{};
                r.diffuse[0] = m[0]; r.diffuse[1] = m[1]; r.diffuse[2] = m[2]; r.specular[0] = m[3]; r.specular[1] = m[4]; r.specular[2] = m[5];
                r.shininess = m[6];
                it = mat_ids.emplace(k, (uint32_t)out->mats.size()).first;
                out->mats.push_back(r);
            }
            start_id.push_back(it->second);
        }
        // every shape takes the id of the run it lies in
        const int64_t CHUNK = 1 << 16;
#pragma omp parallel for schedule(static) if (n > 100000)
        for (int64_t c = 0; c < ((int64_t)n + CHUNK - 1) / CHUNK; c++) {
            const uint32_t a = (uint32_t)(c * CHUNK), b = (uint32_t)std::min<int64_t>((int64_t)n, (c + 1) * CHUNK);
            size_t run = (size_t)(std::upper_bound(starts.begin(), starts.end(), a) - starts.begin()) - 1;     // starts[0] == 0
            for (uint32_t i = a; i < b; i++) {
                if (run + 1 < starts.size() && starts[run + 1] == i) run++;
                ids[i] = start_id[run];
            }
        }
    };
    assign_ids(s.tri_mat, s.ntri, tri_ids.data());
    assign_ids(s.sph_mat, s.nsph, sph_ids.data());
    assign_ids(s.pl_mat, s.npl, pl_ids.data());
    if (out->mats.size() >= (1u << 30)) { *err = "rt_scene: too many distinct materials"; return false; }

    auto flat_of = [&](uint32_t j) {       // bounded shape j: triangles first, then spheres
        if (j < s.ntri) return Flat{rtl::KIND_TRIANGLE, j, s.tri_order ? s.tri_order[j] : j, tri_ids[j]};
        const uint32_t i = j - s.ntri;
        return Flat{rtl::KIND_SPHERE, i, s.sph_order ? s.sph_order[i] : s.ntri + i, sph_ids[i]};
    };
    auto box_of = [](const rtl::ShapeRec& r, int k, float* lo, float* hi) {
        if ((r.kind_mat & 3u) == rtl::KIND_SPHERE) { *lo = r.q0[k] - r.q1[0]; *hi = r.q0[k] + r.q1[0]; }
        else { *lo = std::min(std::min(r.q0[k], r.q1[k]), r.q2[k]); *hi = std::max(std::max(r.q0[k], r.q1[k]), r.q2[k]); }
    };
    // Extent of the bulk of the scene from a sample of the box centres (5th..95th percentile), for the
    // absolute padding and for spotting the few shapes far larger than everything else.
    const size_t ntree = out->use_tree ? bounded : 0;
    const size_t stride = std::max<size_t>(1, ntree / 65536);
    float extent = 0.0f;
    for (int k = 0; k < 3 && ntree > 0; k++) {
        std::vector<float> c;
        for (size_t j = 0; j < ntree; j += stride) {
            rtl::ShapeRec r; fill_record(s, flat_of((uint32_t)j), &r);
            float lo, hi; box_of(r, k, &lo, &hi); c.push_back(0.5f * lo + 0.5f * hi);
        }
        const size_t a = c.size() / 20, b = c.size() - 1 - c.size() / 20;
        std::nth_element(c.begin(), c.begin() + (ptrdiff_t)a, c.end()); const float p5 = c[a];
        std::nth_element(c.begin(), c.begin() + (ptrdiff_t)b, c.end()); const float p95 = c[b];
        if (std::isfinite(p95 - p5)) extent = std::max(extent, p95 - p5);
    }
    out->pad_abs = extent * 4e-6f;

    // The records, and on the way the shapes far larger than the bulk.
    out->tree.resize(bounded);
    std::vector<uint32_t> big;
#pragma omp parallel for schedule(static) if (bounded > 100000)
    for (int64_t j = 0; j < (int64_t)bounded; j++) {
        rtl::ShapeRec& r = out->tree[(size_t)j];
        fill_record(s, flat_of((uint32_t)j), &r);
        if (extent > 0.0f) {
            bool oversize = false;
            for (int k = 0; k < 3; k++) { float lo, hi; box_of(r, k, &lo, &hi); oversize = oversize || hi - lo > 32.0f * extent; }
            if (oversize) {
#pragma omp critical
                big.push_back((uint32_t)j);
            }
        }
    }
    if (!out->use_tree) { out->list.assign(out->tree.begin(), out->tree.end()); out->tree.clear(); }      // a handful of shapes: all of them are tested for every ray
    for (uint32_t i = 0; i < s.npl; i++) {
        rtl::ShapeRec r; fill_record(s, Flat{rtl::KIND_PLANE, i, s.pl_order ? s.pl_order[i] : s.ntri + s.nsph + i, pl_ids[i]}, &r);
        out->list.push_back(r);
    }
    if (!big.empty() && big.size() <= SMALL_SCENE && ntree - big.size() > SMALL_SCENE) {
        std::sort(big.begin(), big.end());
        for (size_t b = big.size(); b-- > 0;) {          // the order of the tree records is free: fill the hole with the last one
            out->list.push_back(out->tree[big[b]]);
            out->tree[big[b]] = out->tree.back();
            out->tree.pop_back();
        }
    }
    std::sort(out->list.begin(), out->list.end(), [](const rtl::ShapeRec& a, const rtl::ShapeRec& b) { return a.order < b.order; });
    for (rtl::ShapeRec& r : out->list) rtl::to_blob_form(&r);      // nothing needs their boxes: straight into the form the blob keeps

    out->lights.resize(s.nlights);
    bool finite = true;
    for (const rtl::MaterialRec& m : out->mats)
        for (int k = 0; k < 3; k++) finite = finite && std::isfinite(m.diffuse[k]) && std::isfinite(m.specular[k]);
    for (uint32_t i = 0; i < s.nlights; i++) {
        const float* l = s.lights + 7 * (size_t)i;
        if (!(l[0] == 0.0f || l[0] == 1.0f)) { *err = "rt_scene: light kind must be 0 (point) or 1 (directional)"; return false; }
        rtl::LightRec& r = out->lights[i];
        r.kind = l[0]; r.v[0] = l[1]; r.v[1] = l[2]; r.v[2] = l[3]; r.color[0] = l[4]; r.color[1] = l[5]; r.color[2] = l[6]; r.pad = 0.0f;
        for (int k = 1; k < 7; k++) finite = finite && std::isfinite(l[k]);
    }
    out->finite_shading = finite ? 1u : 0u;
    out->tree_kinds = kinds_of(out->tree.data(), out->tree.size());
    return true;
}

bool pack_scene(const rt_scene& s, rtu::NoInitVector<uint8_t>* blob, rt_bvh_info* info, std::string* err) {
    auto t0 = std::chrono::steady_clock::now();
    SceneParts parts;
    if (!split_scene(s, &parts, err)) return false;
    const std::vector<rtl::ShapeRec>& list_shapes = parts.list;
    const std::vector<rtl::MaterialRec>& mats = parts.mats;

    // Boxes of the tree shapes (shape.rs:208-234).
    const size_t nt = parts.tree.size();
    rtu::NoInitVector<float> lo(3 * nt), hi(3 * nt);
    bool finite_boxes = true;
#pragma omp parallel for schedule(static) reduction(&& : finite_boxes) if (nt > 100000)
    for (int64_t jj = 0; jj < (int64_t)nt; jj++) {
        const size_t j = (size_t)jj;
        const rtl::ShapeRec& r = parts.tree[j];
        for (int k = 0; k < 3; k++) {
            if ((r.kind_mat & 3u) == rtl::KIND_SPHERE) { lo[3 * j + k] = r.q0[k] - r.q1[0]; hi[3 * j + k] = r.q0[k] + r.q1[0]; }
            else {
                lo[3 * j + k] = std::min(std::min(r.q0[k], r.q1[k]), r.q2[k]);
                hi[3 * j + k] = std::max(std::max(r.q0[k], r.q1[k]), r.q2[k]);
            }
            if (!std::isfinite(lo[3 * j + k]) || !std::isfinite(hi[3 * j + k])) finite_boxes = false;
        }
    }
    if (!finite_boxes) { *err = "rt_scene: non-finite shape coordinates"; return false; }

    rtb::BvhResult bvh;
    if (!rtb::build_bvh(lo.data(), hi.data(), (uint32_t)nt, TREE_DEPTH_LIMIT, parts.pad_abs, &bvh, err)) return false;

    // Assemble the blob.
    rtl::BlobHeader h{};
    h.magic = rtl::BLOB_MAGIC; h.version = rtl::BLOB_VERSION;
    h.n_pairs = (uint32_t)bvh.pairs.size(); h.n_tree_shapes = (uint32_t)nt; h.n_list_shapes = (uint32_t)list_shapes.size();
    h.n_materials = (uint32_t)mats.size(); h.n_lights = s.nlights;
    h.root_ref = rtl::inner_ref(bvh.root_ref); h.max_depth = bvh.max_depth; h.max_leaf = bvh.max_leaf; h.n_leaves = bvh.n_leaves; h.sah_cost = (float)bvh.sah_cost;
    size_t off = sizeof(rtl::BlobHeader);
    h.off_nodes = off; off = align256(off + (size_t)h.n_pairs * sizeof(rtl::QuantPair));
    h.off_tree_shapes = off; off = align256(off + nt * sizeof(rtl::ShapeRec));
    h.off_list_shapes = off; off = align256(off + list_shapes.size() * sizeof(rtl::ShapeRec));
    h.off_materials = off; off = align256(off + mats.size() * sizeof(rtl::MaterialRec));
    h.off_lights = off; off = align256(off + (size_t)s.nlights * sizeof(rtl::LightRec));
    h.total_bytes = off;
    h.finite_shading = parts.finite_shading; h.tree_kinds = parts.tree_kinds;
    blob->resize(off);
    {   // every section is written in full below; only the gaps up to the next 256-byte boundary need zeros
        const size_t ends[5] = {h.off_nodes + (size_t)h.n_pairs * sizeof(rtl::QuantPair), h.off_tree_shapes + nt * sizeof(rtl::ShapeRec),
                                h.off_list_shapes + list_shapes.size() * sizeof(rtl::ShapeRec), h.off_materials + mats.size() * sizeof(rtl::MaterialRec),
                                h.off_lights + (size_t)s.nlights * sizeof(rtl::LightRec)};
        for (size_t e : ends) std::memset(blob->data() + e, 0, align256(e) - e);
    }
    // Quantise the (already padded) builder boxes outwards onto the scene grid (layout.h).
    {
        // The two children of the root bound everything below them.
        double glo[3] = {0, 0, 0}, ghi[3] = {0, 0, 0};
        if (!bvh.pairs.empty() && !(bvh.root_ref & rtl::REF_LEAF)) {
            const rtl::NodePair& p = bvh.pairs[bvh.root_ref];
            for (int k = 0; k < 3; k++) { glo[k] = std::min(p.c[0].lo[k], p.c[1].lo[k]); ghi[k] = std::max(p.c[0].hi[k], p.c[1].hi[k]); }
        }
        for (int k = 0; k < 3; k++) {
            h.grid_origin[k] = std::nextafter((float)glo[k], -INFINITY);
            float sc = (float)((ghi[k] - (double)h.grid_origin[k]) / (double)rtl::GRID_UNITS);
            sc = std::nextafter(std::nextafter(sc, INFINITY), INFINITY);
            if (!(sc > 1e-30f)) sc = 1e-30f;
            h.grid_scale[k] = sc;
        }
        rtl::QuantPair* qp = (rtl::QuantPair*)(blob->data() + h.off_nodes);
        bool ok = true;
#pragma omp parallel for schedule(static) reduction(&& : ok) if (h.n_pairs > 100000)
        for (int64_t i = 0; i < (int64_t)h.n_pairs; i++) {
            for (int c = 0; c < 2; c++) {
                const rtl::ChildBox& b = bvh.pairs[(size_t)i].c[c];
                for (int k = 0; k < 3; k++) {
                    const double o = h.grid_origin[k], sc = h.grid_scale[k];
                    const double ul = std::min(std::max(((double)b.lo[k] - o) / sc, 0.0), (double)rtl::GRID_UNITS);
                    const double uh = std::min(std::max(((double)b.hi[k] - o) / sc, 0.0), (double)rtl::GRID_UNITS);
                    uint32_t ql = rtl::grid_code_below((float)std::floor(ul)), qh = rtl::grid_code_above((float)std::ceil(uh));
                    while (ql > 0 && o + (double)rtl::grid_units(ql) * sc > (double)b.lo[k]) ql -= 1;         // the grid plane must not lie inside the box
                    while (qh < 65535u && o + (double)rtl::grid_units(qh) * sc < (double)b.hi[k]) qh += 1;
                    if (o + (double)rtl::grid_units(ql) * sc > (double)b.lo[k] || o + (double)rtl::grid_units(qh) * sc < (double)b.hi[k]) ok = false;
                    qp[i].c[c].q[k][0] = (uint16_t)ql; qp[i].c[c].q[k][1] = (uint16_t)qh;
                }
                qp[i].c[c].ref = rtl::inner_ref(b.ref);
            }
        }
        if (!ok) { *err = "internal: a node box does not fit the quantisation grid"; return false; }
    }
    std::memcpy(blob->data(), &h, sizeof(h));        // after the grid has been filled in
    rtl::ShapeRec* ts = (rtl::ShapeRec*)(blob->data() + h.off_tree_shapes);
#pragma omp parallel for schedule(static) if (nt > 100000)
    for (int64_t j = 0; j < (int64_t)nt; j++) { ts[j] = parts.tree[bvh.order[(size_t)j]]; rtl::to_blob_form(&ts[j]); }
    if (!list_shapes.empty()) std::memcpy(blob->data() + h.off_list_shapes, list_shapes.data(), list_shapes.size() * sizeof(rtl::ShapeRec));
    if (!mats.empty()) std::memcpy(blob->data() + h.off_materials, mats.data(), mats.size() * sizeof(rtl::MaterialRec));
    if (!parts.lights.empty()) std::memcpy(blob->data() + h.off_lights, parts.lights.data(), parts.lights.size() * sizeof(rtl::LightRec));
    if (info) {
        info->node_pairs = h.n_pairs; info->leaves = h.n_leaves; info->max_depth = h.max_depth; info->max_leaf = h.max_leaf;
        info->bvh_prims = h.n_tree_shapes; info->list_prims = h.n_list_shapes; info->sah_cost = bvh.sah_cost;
        info->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return true;
}

}  // namespace rtp
