// rt_oracle.cpp -- CPU restatement of the reference ray tracer's render path.
//
// TEST INFRASTRUCTURE ONLY.  This file is the checker that the HIP path is compared
// against; nothing in the product library links, loads or calls it.  Only tests/,
// __graft_entry__.smoke() and the cpu_baseline leg of bench.py may use it.
//
// Parity status: PINNED.  The reference (Rust) cannot be compiled in this environment,
// so this restatement is pinned by the reference's own committed renders: the 20
// test_file/jalon{3,5,6}/tp*.png goldens (reference comparator reports 0 differing
// pixels on all of them) and dragon3.png, the reference's 640x480 render of
// final_avec_bonus.scene (307200/307200 pixels bit-identical).  tests/test_oracle_golden.py
// re-checks those pins.
//
// All arithmetic is IEEE binary32, evaluated strictly in the written order.  Build with
// -ffp-contract=off and without -ffast-math (see oracle/Makefile).
//
// What each part follows (paths relative to the reference tree):
//   vector primitives            nalgebra 0.34.1 Vector3<f32> as used by the files below
//   scene parser                 src/raytracer/config/config_builder.rs:127-460
//   camera direction             src/raytracer/config/camera.rs:11-13
//   frame setup + pixel loop     src/raytracer/raytracer.rs:40-79
//   quantise                     src/raytracer/raytracer.rs:85-91
//   closest hit / shadows /
//   shading / reflection         src/raytracer/raytracer.rs:99-202
//   sphere / plane / triangle    src/raytracer/config/shape.rs:59-97, 99-133, 135-192
//   shape bounding boxes         src/raytracer/config/shape.rs:205-245
//   image comparator             src/imgcomparator/mod.rs:54-119
//
// The bvh crate (0.12.0, Cargo.lock:158-167) is not part of the reference tree.  In the
// reference it only selects *candidates* (raytracer.rs:110,141): every candidate is then
// intersected exactly and the minimum distance wins, so the image does not depend on the
// tree as long as culling is conservative.  Mode 0 below (no tree, every shape is a
// candidate) is therefore the ground truth; modes 1 and 2 use this file's own tree and
// exist for the CPU timing baseline and to cross-check that pruning does not change
// pixels.  Closest-hit ties on exactly equal distance depend on the crate's candidate
// order and are unpinned; here the lowest file-order index wins.

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cerrno>
#include <string>
#include <vector>
#include <algorithm>
#include <fstream>
#include <limits>
#include <omp.h>

namespace {

// ------------------------------------------------------------------ vector primitives
struct V3 { float x, y, z; };

inline V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
inline V3 vadd(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 vsub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 vscale(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 vcmul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 vneg(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline float vdot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 vcross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float vnorm(V3 a) { return sqrtf(vdot(a, a)); }
// Reciprocal of the norm, then three multiplies: the only form that reproduces
// dragon3.png bit for bit (dividing each component matches 82 % of its pixels).
inline V3 vnormalize(V3 a) { float inv = 1.0f / vnorm(a); return vscale(a, inv); }

// ------------------------------------------------------------------ scene model
enum Kind : int { SPHERE = 0, TRIANGLE = 1, PLANE = 2 };

struct Shape {            // shape.rs:5-32 (node_index dropped: the tree is local to this file)
    int kind;
    V3 a, b, c;           // sphere: a=center, b.x=radius | triangle: v0,v1,v2 | plane: a=point, b=normal
    V3 diffuse, specular;
    float shininess;
};

struct Light {            // light.rs:3-6
    int directional;      // 0 = point (v = position), 1 = directional (v = unit direction towards the light)
    V3 v, color;
};

struct Scene {            // config_builder.rs:14-24
    uint32_t width = 800, height = 600;
    std::string output = "output.png";
    V3 cam_pos = v3(0, 0, 0), cam_look = v3(0, 0, 1), cam_up = v3(0, 1, 0);
    float fov = 60.0f;
    V3 ambient = v3(0, 0, 0);
    uint32_t maxdepth = 1, maxverts = 0;
    std::vector<Shape> shapes;
    std::vector<Light> lights;
    // tree (this file's own; built lazily)
    struct Node { float lo[3], hi[3]; int left, right; int first, count; };
    std::vector<Node> nodes;
    std::vector<int> order;      // tree leaf order -> shape index (bounded shapes only)
    std::vector<int> unbounded;  // planes: always candidates
    int tree_leaf = 0;           // leaf size the current tree was built with
};

// ------------------------------------------------------------------ parser
// Rust's str::parse::<f32> grammar: [+-]? (digits [. digits?]? | . digits) ([eE] [+-]? digits)?
// or inf / infinity / nan, case-insensitive.  No surrounding blanks, no hex.
bool parse_f32(const std::string& s, float* out, std::string* err) {
    size_t i = 0, n = s.size();
    if (n == 0) { *err = "cannot parse float from empty string"; return false; }
    if (s[i] == '+' || s[i] == '-') i++;
    std::string rest = s.substr(i);
    std::string low = rest;
    for (auto& ch : low) ch = (char)tolower((unsigned char)ch);
    bool ok = false;
    if (low == "inf" || low == "infinity" || low == "nan") ok = true;
    else {
        size_t j = 0, m = rest.size();
        size_t d0 = 0, d1 = 0;
        while (j < m && isdigit((unsigned char)rest[j])) { j++; d0++; }
        if (j < m && rest[j] == '.') { j++; while (j < m && isdigit((unsigned char)rest[j])) { j++; d1++; } }
        if (d0 + d1 > 0) {
            if (j < m && (rest[j] == 'e' || rest[j] == 'E')) {
                j++;
                if (j < m && (rest[j] == '+' || rest[j] == '-')) j++;
                size_t de = 0;
                while (j < m && isdigit((unsigned char)rest[j])) { j++; de++; }
                ok = (de > 0 && j == m);
            } else ok = (j == m);
        }
    }
    if (!ok) { *err = "invalid float literal"; return false; }
    *out = strtof(s.c_str(), nullptr);   // glibc strtof is correctly rounded, like Rust's parser
    return true;
}

bool parse_uint(const std::string& s, uint64_t maxv, uint64_t* out, std::string* err) {
    size_t i = 0, n = s.size();
    if (n == 0) { *err = "cannot parse integer from empty string"; return false; }
    if (s[0] == '+') i = 1;
    if (i == n) { *err = "invalid digit found in string"; return false; }
    uint64_t v = 0;
    for (; i < n; i++) {
        if (!isdigit((unsigned char)s[i])) { *err = "invalid digit found in string"; return false; }
        unsigned d = (unsigned)(s[i] - '0');
        if (v > (maxv - d) / 10) { *err = "number too large to fit in target type"; return false; }
        v = v * 10 + d;
    }
    *out = v;
    return true;
}

std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) a++;
    while (b > a && isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

std::vector<std::string> split_sp(const std::string& s) {
    std::vector<std::string> out;
    size_t p = 0;
    for (;;) {
        size_t q = s.find(' ', p);
        if (q == std::string::npos) { out.push_back(s.substr(p)); break; }
        out.push_back(s.substr(p, q - p));
        p = q + 1;
    }
    return out;
}

bool parse_floats(const std::string& param, size_t want, const char* fmt_err, float* out, std::string* err) {
    std::vector<std::string> t = split_sp(param);
    if (t.size() != want) { *err = fmt_err; return false; }
    for (size_t i = 0; i < want; i++) if (!parse_f32(t[i], &out[i], err)) return false;
    return true;
}

bool rgb_ok(float r, float g, float b) {   // config_builder.rs:376-381
    return r >= 0.0f && r <= 1.0f && g >= 0.0f && g <= 1.0f && b >= 0.0f && b <= 1.0f;
}

struct ParserState {      // config_builder.rs:111-116
    V3 diffuse = v3(0, 0, 0), specular = v3(0, 0, 0);
    float shininess = 0.0f;
    std::vector<V3> vertices;
};

bool parse_line(ParserState& st, const std::string& line, Scene& sc, std::string* err) {
    std::string t = trim(line);
    if (t.empty() || t[0] == '#') return true;                       // :153
    std::vector<std::string> parts = split_sp(line);                 // :156
    if (parts.size() < 2) return true;                               // :157
    std::string key = trim(parts[0]);
    std::string param = trim(line.substr(key.size()));               // :158
    float f[10];
    const char* rgb_err = "RGB values must be between 0.0 and 1.0";
    if (key == "size") {                                             // :249-262
        std::vector<std::string> d = split_sp(param);
        if (d.size() != 2) { *err = "Invalid size format"; return false; }
        uint64_t w, h;
        if (!parse_uint(d[0], 0xFFFFFFFFull, &w, err) || !parse_uint(d[1], 0xFFFFFFFFull, &h, err)) return false;
        if (w == 0 || h == 0) { *err = "Width and height must be greater than zero"; return false; }
        sc.width = (uint32_t)w; sc.height = (uint32_t)h;
    } else if (key == "output") {                                    // :368-374
        if (param.empty()) { *err = "Output file name cannot be empty"; return false; }
        sc.output = param;
    } else if (key == "camera") {                                    // :309-341
        if (!parse_floats(param, 10, "Invalid camera format", f, err)) return false;
        if (f[9] < 1.0f || f[9] > 179.0f) { *err = "Field of view (fov) must be between 1 and 179 degrees"; return false; }
        sc.cam_pos = v3(f[0], f[1], f[2]); sc.cam_look = v3(f[3], f[4], f[5]); sc.cam_up = v3(f[6], f[7], f[8]);
        sc.fov = f[9];
    } else if (key == "ambient") {                                   // :343-355
        if (!parse_floats(param, 3, "Invalid ambient light format", f, err)) return false;
        if (!rgb_ok(f[0], f[1], f[2])) { *err = rgb_err; return false; }
        sc.ambient = v3(f[0], f[1], f[2]);
    } else if (key == "sphere") {                                    // :383-406
        if (!parse_floats(param, 4, "Invalid sphere format", f, err)) return false;
        if (f[3] <= 0.0f) { *err = "Sphere radius must be greater than zero"; return false; }
        Shape s{}; s.kind = SPHERE; s.a = v3(f[0], f[1], f[2]); s.b = v3(f[3], 0, 0); s.c = v3(0, 0, 0);
        s.diffuse = st.diffuse; s.specular = st.specular; s.shininess = st.shininess;
        sc.shapes.push_back(s);
    } else if (key == "tri") {                                       // :408-433
        std::vector<std::string> d = split_sp(param);
        if (d.size() != 3) { *err = "Invalid triangle format"; return false; }
        uint64_t idx[3];
        for (int k = 0; k < 3; k++) if (!parse_uint(d[k], 0xFFFFFFFFFFFFFFFFull, &idx[k], err)) return false;
        for (int k = 0; k < 3; k++) if (idx[k] >= st.vertices.size()) { *err = "Triangle vertex index out of bounds"; return false; }
        Shape s{}; s.kind = TRIANGLE; s.a = st.vertices[idx[0]]; s.b = st.vertices[idx[1]]; s.c = st.vertices[idx[2]];
        s.diffuse = st.diffuse; s.specular = st.specular; s.shininess = st.shininess;
        sc.shapes.push_back(s);
    } else if (key == "plane") {                                     // :435-460
        if (!parse_floats(param, 6, "Invalid plane format", f, err)) return false;
        Shape s{}; s.kind = PLANE; s.a = v3(f[0], f[1], f[2]); s.b = vnormalize(v3(f[3], f[4], f[5])); s.c = v3(0, 0, 0);
        s.diffuse = st.diffuse; s.specular = st.specular; s.shininess = st.shininess;
        sc.shapes.push_back(s);
    } else if (key == "point") {                                     // :264-283
        if (!parse_floats(param, 6, "Invalid point light format", f, err)) return false;
        if (!rgb_ok(f[3], f[4], f[5])) { *err = rgb_err; return false; }
        Light l; l.directional = 0; l.v = v3(f[0], f[1], f[2]); l.color = v3(f[3], f[4], f[5]);
        sc.lights.push_back(l);
    } else if (key == "directional") {                               // :285-307
        if (!parse_floats(param, 6, "Invalid directional light format", f, err)) return false;
        if (!rgb_ok(f[3], f[4], f[5])) { *err = rgb_err; return false; }
        Light l; l.directional = 1; l.v = vnormalize(v3(f[0], f[1], f[2])); l.color = v3(f[3], f[4], f[5]);
        sc.lights.push_back(l);
    } else if (key == "diffuse") {                                   // :196-212
        if (!parse_floats(param, 3, "Invalid Vector3 format", f, err)) return false;
        st.diffuse = v3(f[0], f[1], f[2]);     // the state is updated before validation, as in the reference
        if (!rgb_ok(f[0], f[1], f[2])) { *err = rgb_err; return false; }
        if ((f[0] + sc.ambient.x) > 1.0f || (f[1] + sc.ambient.y) > 1.0f || (f[2] + sc.ambient.z) > 1.0f) {
            *err = "Sum of diffuse color and ambient light components must not exceed 1.0"; return false;
        }
    } else if (key == "specular") {                                  // :213-221
        if (!parse_floats(param, 3, "Invalid Vector3 format", f, err)) return false;
        st.specular = v3(f[0], f[1], f[2]);
        if (f[0] < 0.0f || f[1] < 0.0f || f[2] < 0.0f) { *err = "Specular color components must be non-negative"; return false; }
    } else if (key == "shininess") {                                 // :222-227
        if (!parse_f32(param, &f[0], err)) return false;
        st.shininess = f[0];
        if (f[0] < 0.0f) { *err = "Shininess must be non-negative"; return false; }
    } else if (key == "maxdepth") {                                  // :228-230
        uint64_t v; if (!parse_uint(param, 0xFFFFFFFFull, &v, err)) return false;
        sc.maxdepth = (uint32_t)v;
    } else if (key == "maxverts") {                                  // :231-234
        uint64_t v; if (!parse_uint(param, 0xFFFFFFFFull, &v, err)) return false;
        sc.maxverts = (uint32_t)v;
    } else if (key == "vertex") {                                    // :235-241
        if (!parse_floats(param, 3, "Invalid Vector3 format", f, err)) return false;
        if (st.vertices.size() >= (size_t)sc.maxverts) { *err = "Exceeded maximum number of vertices (maxverts)"; return false; }
        st.vertices.push_back(v3(f[0], f[1], f[2]));
    } else {
        *err = "Unknown configuration key: " + key;                  // :242-244
        return false;
    }
    return true;
}

// ------------------------------------------------------------------ intersections
struct Hit {              // shape.rs:39-47 (material referenced by shape index instead of copied)
    float distance;
    V3 normal, point;
    int shape;
    bool back;
};

inline bool hit_sphere(const Shape& s, V3 o, V3 d, Hit* h) {         // shape.rs:72-96
    V3 oc = vsub(o, s.a);
    float half_b = vdot(oc, d);
    float c = vdot(oc, oc) - s.b.x * s.b.x;
    float disc = half_b * half_b - c;
    if (disc < 0.0f) return false;
    float t = -half_b - sqrtf(disc);
    if (t < 0.0f) return false;
    V3 p = vadd(o, vscale(d, t));
    h->distance = t; h->point = p; h->normal = vnormalize(vsub(p, s.a)); h->back = false;
    return true;
}

inline bool hit_plane(const Shape& s, V3 o, V3 d, Hit* h) {          // shape.rs:112-132
    float denom = vdot(s.b, d);
    if (fabsf(denom) < 1e-6f) return false;
    float t = vdot(vsub(s.a, o), s.b) / denom;
    if (t < 0.0f) return false;
    h->distance = t; h->point = vadd(o, vscale(d, t)); h->normal = s.b; h->back = false;
    return true;
}

inline bool hit_triangle(const Shape& s, V3 o, V3 d, Hit* h) {       // shape.rs:149-191
    V3 e1 = vsub(s.b, s.a);
    V3 e2 = vsub(s.c, s.a);
    V3 hh = vcross(d, e2);
    float a = vdot(e1, hh);
    if (fabsf(a) < 1e-6f) return false;
    float f = 1.0f / a;
    V3 sv = vsub(o, s.a);
    float u = f * vdot(sv, hh);
    if (u < 0.0f || u > 1.0f) return false;
    V3 q = vcross(sv, e1);
    float v = f * vdot(d, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = f * vdot(e2, q);
    if (t < 0.0f) return false;
    V3 n = vnormalize(vcross(e1, e2));
    h->distance = t; h->point = vadd(o, vscale(d, t)); h->normal = n; h->back = vdot(n, d) > 0.0f;
    return true;
}

inline bool hit_shape(const Shape& s, V3 o, V3 d, Hit* h) {          // shape.rs:50-56
    switch (s.kind) {
        case SPHERE: return hit_sphere(s, o, d, h);
        case PLANE: return hit_plane(s, o, d, h);
        default: return hit_triangle(s, o, d, h);
    }
}

// ------------------------------------------------------------------ this file's own tree
void shape_box(const Shape& s, float lo[3], float hi[3]) {           // shape.rs:205-245
    if (s.kind == SPHERE) {
        lo[0] = s.a.x - s.b.x; lo[1] = s.a.y - s.b.x; lo[2] = s.a.z - s.b.x;
        hi[0] = s.a.x + s.b.x; hi[1] = s.a.y + s.b.x; hi[2] = s.a.z + s.b.x;
    } else if (s.kind == TRIANGLE) {
        lo[0] = fminf(fminf(s.a.x, s.b.x), s.c.x); hi[0] = fmaxf(fmaxf(s.a.x, s.b.x), s.c.x);
        lo[1] = fminf(fminf(s.a.y, s.b.y), s.c.y); hi[1] = fmaxf(fmaxf(s.a.y, s.b.y), s.c.y);
        lo[2] = fminf(fminf(s.a.z, s.b.z), s.c.z); hi[2] = fmaxf(fmaxf(s.a.z, s.b.z), s.c.z);
    } else {
        for (int k = 0; k < 3; k++) { lo[k] = -1e10f; hi[k] = 1e10f; }
    }
}

struct BuildItem { float lo[3], hi[3], ctr[3]; int shape; };

int build_rec(Scene& sc, std::vector<BuildItem>& items, int first, int count, int leaf) {
    Scene::Node nd;
    for (int k = 0; k < 3; k++) { nd.lo[k] = std::numeric_limits<float>::infinity(); nd.hi[k] = -nd.lo[k]; }
    for (int i = first; i < first + count; i++)
        for (int k = 0; k < 3; k++) { nd.lo[k] = fminf(nd.lo[k], items[i].lo[k]); nd.hi[k] = fmaxf(nd.hi[k], items[i].hi[k]); }
    // Pad outwards so that culling stays conservative against rounding in the slab test.
    for (int k = 0; k < 3; k++) {
        float m = fmaxf(fabsf(nd.lo[k]), fabsf(nd.hi[k]));
        float pad = m * 4e-7f + 1e-30f;
        nd.lo[k] -= pad; nd.hi[k] += pad;
    }
    nd.left = nd.right = -1; nd.first = first; nd.count = count;
    int me = (int)sc.nodes.size();
    sc.nodes.push_back(nd);
    if (count <= leaf) return me;
    // Split at the median of the centroids along the axis of largest centroid extent.
    float clo[3], chi[3];
    for (int k = 0; k < 3; k++) { clo[k] = std::numeric_limits<float>::infinity(); chi[k] = -clo[k]; }
    for (int i = first; i < first + count; i++)
        for (int k = 0; k < 3; k++) { clo[k] = fminf(clo[k], items[i].ctr[k]); chi[k] = fmaxf(chi[k], items[i].ctr[k]); }
    int axis = 0;
    if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
    if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
    int mid = first + count / 2;
    std::nth_element(items.begin() + first, items.begin() + mid, items.begin() + first + count,
                     [axis](const BuildItem& p, const BuildItem& q) {
                         if (p.ctr[axis] != q.ctr[axis]) return p.ctr[axis] < q.ctr[axis];
                         return p.shape < q.shape;
                     });
    int l = build_rec(sc, items, first, mid - first, leaf);
    int r = build_rec(sc, items, mid, first + count - mid, leaf);
    sc.nodes[me].left = l; sc.nodes[me].right = r; sc.nodes[me].count = 0;
    return me;
}

void build_tree(Scene& sc, int leaf) {
    if (sc.tree_leaf == leaf) return;
    sc.nodes.clear(); sc.order.clear(); sc.unbounded.clear();
    std::vector<BuildItem> items;
    for (size_t i = 0; i < sc.shapes.size(); i++) {
        if (sc.shapes[i].kind == PLANE) { sc.unbounded.push_back((int)i); continue; }
        BuildItem it; shape_box(sc.shapes[i], it.lo, it.hi);
        for (int k = 0; k < 3; k++) it.ctr[k] = 0.5f * it.lo[k] + 0.5f * it.hi[k];
        it.shape = (int)i; items.push_back(it);
    }
    if (!items.empty()) build_rec(sc, items, 0, (int)items.size(), leaf);
    sc.order.resize(items.size());
    for (size_t i = 0; i < items.size(); i++) sc.order[i] = items[i].shape;
    sc.tree_leaf = leaf;
}

// Conservative slab test: is the padded box possibly entered on [0, tlimit]?
inline bool box_hit(const Scene::Node& n, V3 o, V3 inv, float tlimit) {
    float t0 = 0.0f, t1 = tlimit;
    const float oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
    for (int k = 0; k < 3; k++) {
        float a = (n.lo[k] - oo[k]) * ii[k], b = (n.hi[k] - oo[k]) * ii[k];
        float tn = fminf(a, b), tf = fmaxf(a, b);     // fminf/fmaxf drop a NaN operand (0 * inf)
        t0 = fmaxf(t0, tn * 0.9999995f); t1 = fminf(t1, tf * 1.0000005f);
    }
    return t0 <= t1;
}

struct Counters { uint64_t primary = 0, shadow = 0, secondary = 0, nodes = 0, prims = 0; };

enum Mode : int { MODE_ALL = 0, MODE_TREE_COLLECT = 1, MODE_TREE_PRUNED = 2 };

struct Tracer {
    const Scene& sc;
    int mode;
    Counters cnt;
    std::vector<int> cand;     // candidate list reused across rays (mode 1)
    std::vector<int> stack;
    Tracer(const Scene& s, int m) : sc(s), mode(m) { stack.reserve(128); }

    void collect(V3 o, V3 d) {                       // what bvh.traverse does at raytracer.rs:110,141
        cand.clear();
        for (int i : sc.unbounded) cand.push_back(i);
        if (sc.nodes.empty()) return;
        V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        stack.clear(); stack.push_back(0);
        const float inf = std::numeric_limits<float>::infinity();
        while (!stack.empty()) {
            int ni = stack.back(); stack.pop_back();
            const Scene::Node& n = sc.nodes[ni];
            cnt.nodes++;
            if (!box_hit(n, o, inv, inf)) continue;
            if (n.left < 0) { for (int i = n.first; i < n.first + n.count; i++) cand.push_back(sc.order[i]); }
            else { stack.push_back(n.right); stack.push_back(n.left); }
        }
    }

    // raytracer.rs:113-120.  min_by keeps the earlier element unless it compares Greater.
    bool closest(V3 o, V3 d, Hit* best) {
        bool found = false; int best_idx = -1;
        auto consider = [&](int i) {
            Hit h; cnt.prims++;
            if (!hit_shape(sc.shapes[i], o, d, &h)) return;
            h.shape = i;
            if (!found) { *best = h; best_idx = i; found = true; return; }
            if (best->distance > h.distance || (best->distance == h.distance && i < best_idx)) { *best = h; best_idx = i; }
        };
        if (mode == MODE_ALL) {
            for (size_t i = 0; i < sc.shapes.size(); i++) consider((int)i);
        } else if (mode == MODE_TREE_COLLECT) {
            collect(o, d);
            for (int i : cand) consider(i);
        } else {
            for (int i : sc.unbounded) consider(i);
            if (!sc.nodes.empty()) {
                V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                const float inf = std::numeric_limits<float>::infinity();
                stack.clear(); stack.push_back(0);
                while (!stack.empty()) {
                    int ni = stack.back(); stack.pop_back();
                    const Scene::Node& n = sc.nodes[ni];
                    cnt.nodes++;
                    // Slack on the pruning bound: a later candidate whose computed distance is
                    // smaller must never be culled because of rounding in an earlier one.
                    float lim = found ? best->distance * 1.0005f + 1e-4f : inf;
                    if (!box_hit(n, o, inv, lim)) continue;
                    if (n.left < 0) { for (int i = n.first; i < n.first + n.count; i++) consider(sc.order[i]); }
                    else { stack.push_back(n.right); stack.push_back(n.left); }
                }
            }
        }
        return found;
    }

    // raytracer.rs:143-159
    inline bool occludes(int i, V3 o, V3 d, bool hit_back, bool point_light, float light_dist) {
        Hit s; cnt.prims++;
        if (!hit_shape(sc.shapes[i], o, d, &s)) return false;
        if (s.distance < 1e-6f) return false;
        if (hit_back && s.back) return false;
        if (point_light) return s.distance < light_dist;
        return true;
    }

    bool in_shadow(V3 o, V3 d, bool hit_back, bool point_light, float light_dist) {
        if (mode == MODE_ALL) {
            bool any = false;
            for (size_t i = 0; i < sc.shapes.size(); i++) any = occludes((int)i, o, d, hit_back, point_light, light_dist) || any;
            return any;
        } else if (mode == MODE_TREE_COLLECT) {
            collect(o, d);
            bool any = false;
            for (int i : cand) any = occludes(i, o, d, hit_back, point_light, light_dist) || any;
            return any;
        }
        for (int i : sc.unbounded) if (occludes(i, o, d, hit_back, point_light, light_dist)) return true;
        if (sc.nodes.empty()) return false;
        V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        float lim = point_light ? light_dist * 1.0005f + 1e-4f : std::numeric_limits<float>::infinity();
        stack.clear(); stack.push_back(0);
        while (!stack.empty()) {
            int ni = stack.back(); stack.pop_back();
            const Scene::Node& n = sc.nodes[ni];
            cnt.nodes++;
            if (!box_hit(n, o, inv, lim)) continue;
            if (n.left < 0) {
                for (int i = n.first; i < n.first + n.count; i++)
                    if (occludes(sc.order[i], o, d, hit_back, point_light, light_dist)) return true;
            } else { stack.push_back(n.right); stack.push_back(n.left); }
        }
        return false;
    }

    // raytracer.rs:99-202
    V3 find_color_recursive(V3 origin, V3 direction, uint32_t depth) {
        if (depth > sc.maxdepth) return v3(0, 0, 0);
        Hit hit;
        if (!closest(origin, direction, &hit)) return v3(0, 0, 0);
        const Shape& sh = sc.shapes[hit.shape];
        V3 acc = v3(0, 0, 0);
        for (const Light& light : sc.lights) {
            V3 light_dir = light.directional ? light.v : vnormalize(vsub(light.v, hit.point));
            V3 so = vadd(hit.point, vscale(hit.normal, 1e-6f));
            float light_dist = light.directional ? 0.0f : vnorm(vsub(light.v, hit.point));
            cnt.shadow++;
            if (in_shadow(so, light_dir, hit.back, !light.directional, light_dist)) continue;
            float n_dot_l = fmaxf(vdot(hit.normal, light_dir), 0.0f);
            V3 diffuse = vscale(sh.diffuse, n_dot_l);
            V3 view_dir = vneg(direction);
            V3 half_vector = vnormalize(vadd(light_dir, view_dir));
            float n_dot_h = fmaxf(vdot(hit.normal, half_vector), 0.0f);
            float factor;
            if (sh.shininess == 1.0f) factor = n_dot_h;
            else if (sh.shininess == 0.0f) factor = n_dot_l > 0.0f ? n_dot_h : 0.0f;
            else factor = n_dot_l > 0.0f ? powf(n_dot_h, sh.shininess) : 0.0f;
            V3 specular = vscale(sh.specular, factor);
            acc = vadd(acc, vcmul(vadd(diffuse, specular), light.color));
        }
        V3 final_color = vadd(acc, sc.ambient);
        bool reflective = sh.specular.x > 0.0f || sh.specular.y > 0.0f || sh.specular.z > 0.0f;
        if (reflective && depth + 1 < sc.maxdepth) {
            float k = 2.0f * vdot(direction, hit.normal);
            V3 reflect_dir = vsub(direction, vscale(hit.normal, k));
            V3 reflect_origin = vadd(hit.point, vscale(hit.normal, 1e-6f));
            cnt.secondary++;
            V3 reflected = find_color_recursive(reflect_origin, reflect_dir, depth + 1);
            final_color = vadd(final_color, vcmul(sh.specular, reflected));
        }
        return final_color;
    }

    uint32_t find_color(V3 origin, V3 direction) {                   // raytracer.rs:85-91
        cnt.primary++;
        V3 c = find_color_recursive(origin, direction, 0);
        uint32_t r = (uint32_t)roundf(fminf(fmaxf(c.x, 0.0f), 1.0f) * 255.0f);
        uint32_t g = (uint32_t)roundf(fminf(fmaxf(c.y, 0.0f), 1.0f) * 255.0f);
        uint32_t b = (uint32_t)roundf(fminf(fmaxf(c.z, 0.0f), 1.0f) * 255.0f);
        return (255u << 24) | (r << 16) | (g << 8) | b;
    }
};

struct FrameSetup { V3 pos, cv, right, up; float pw, ph, w2, h2; };

FrameSetup frame_setup(const Scene& sc, uint32_t W, uint32_t H) {     // raytracer.rs:46-55, camera.rs:11-13
    FrameSetup f;
    f.pos = sc.cam_pos;
    f.cv = vnormalize(vnormalize(vsub(sc.cam_look, sc.cam_pos)));
    f.right = vnormalize(vcross(f.cv, sc.cam_up));
    f.up = vnormalize(vcross(f.right, f.cv));
    const float PI_F = 3.14159265358979323846f;
    float fovrad = sc.fov * PI_F / 180.0f;
    f.ph = tanf(fovrad / 2.0f);
    f.pw = f.ph * ((float)W / (float)H);
    f.w2 = (float)W / 2.0f;
    f.h2 = (float)H / 2.0f;
    return f;
}

thread_local std::string g_err;

}  // namespace

// ====================================================================== C interface (ctypes)
extern "C" {

struct oracle_stats {
    uint64_t primary, shadow, secondary, nodes_visited, prims_tested;
    double render_ms;
    int threads;
};

const char* oracle_last_error() { return g_err.c_str(); }

void* oracle_scene_new() { return new Scene(); }
void oracle_scene_free(void* h) { delete (Scene*)h; }

// Parse a .scene/.test file.  Returns NULL and sets oracle_last_error() on failure.
void* oracle_load_scene(const char* path) {
    std::ifstream in(path);
    if (!in) { g_err = std::string("No such file or directory (os error 2): ") + path; return nullptr; }
    Scene* sc = new Scene();
    ParserState st;
    std::string line, err;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!parse_line(st, line, *sc, &err)) { g_err = err; delete sc; return nullptr; }
    }
    return sc;
}

// Parse scene text held in memory (same grammar).
void* oracle_parse_scene(const char* text) {
    Scene* sc = new Scene();
    ParserState st;
    std::string err, all(text);
    size_t p = 0;
    while (p <= all.size()) {
        size_t q = all.find('\n', p);
        if (q == std::string::npos) q = all.size();
        if (q == all.size() && p == q) break;
        std::string line = all.substr(p, q - p);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!parse_line(st, line, *sc, &err)) { g_err = err; delete sc; return nullptr; }
        p = q + 1;
    }
    return sc;
}

void oracle_scene_info(void* h, uint32_t* out /* width,height,maxdepth,nshapes,nlights,ntri,nsph,npl */) {
    Scene* sc = (Scene*)h;
    out[0] = sc->width; out[1] = sc->height; out[2] = sc->maxdepth;
    out[3] = (uint32_t)sc->shapes.size(); out[4] = (uint32_t)sc->lights.size();
    uint32_t nt = 0, ns = 0, np = 0;
    for (auto& s : sc->shapes) { if (s.kind == TRIANGLE) nt++; else if (s.kind == SPHERE) ns++; else np++; }
    out[5] = nt; out[6] = ns; out[7] = np;
}

void oracle_scene_output(void* h, char* buf, size_t cap) {
    Scene* sc = (Scene*)h;
    snprintf(buf, cap, "%s", sc->output.c_str());
}

// Programmatic construction (synthetic scenes too large for a text file).
void oracle_scene_set_frame(void* h, uint32_t w, uint32_t hgt, uint32_t maxdepth, const float* cam10, const float* ambient3) {
    Scene* sc = (Scene*)h;
    sc->width = w; sc->height = hgt; sc->maxdepth = maxdepth;
    sc->cam_pos = v3(cam10[0], cam10[1], cam10[2]); sc->cam_look = v3(cam10[3], cam10[4], cam10[5]);
    sc->cam_up = v3(cam10[6], cam10[7], cam10[8]); sc->fov = cam10[9];
    sc->ambient = v3(ambient3[0], ambient3[1], ambient3[2]);
}
// kind: 0 sphere (g = cx cy cz r), 1 triangle (g = 9 floats), 2 plane (g = px py pz nx ny nz, normal normalised here)
void oracle_scene_add_shapes(void* h, int kind, uint64_t n, const float* g, const float* mat7, int mat_per_shape) {
    Scene* sc = (Scene*)h;
    size_t stride = kind == SPHERE ? 4 : (kind == TRIANGLE ? 9 : 6);
    sc->shapes.reserve(sc->shapes.size() + n);
    for (uint64_t i = 0; i < n; i++) {
        const float* p = g + i * stride; const float* m = mat7 + (mat_per_shape ? i * 7 : 0);
        Shape s{}; s.kind = kind;
        if (kind == SPHERE) { s.a = v3(p[0], p[1], p[2]); s.b = v3(p[3], 0, 0); s.c = v3(0, 0, 0); }
        else if (kind == TRIANGLE) { s.a = v3(p[0], p[1], p[2]); s.b = v3(p[3], p[4], p[5]); s.c = v3(p[6], p[7], p[8]); }
        else { s.a = v3(p[0], p[1], p[2]); s.b = vnormalize(v3(p[3], p[4], p[5])); s.c = v3(0, 0, 0); }
        s.diffuse = v3(m[0], m[1], m[2]); s.specular = v3(m[3], m[4], m[5]); s.shininess = m[6];
        sc->shapes.push_back(s);
    }
    sc->tree_leaf = 0;
}
void oracle_scene_add_light(void* h, int directional, const float* v3in, const float* color3) {
    Scene* sc = (Scene*)h;
    Light l; l.directional = directional;
    l.v = directional ? vnormalize(v3(v3in[0], v3in[1], v3in[2])) : v3(v3in[0], v3in[1], v3in[2]);
    l.color = v3(color3[0], color3[1], color3[2]);
    sc->lights.push_back(l);
}

// Flattened view of the parsed scene, so that a test can hand exactly the same numbers to
// the code under test without going through a second parser.
// shapes: 17 floats each = kind, a(3), b(3), c(3), diffuse(3), specular(3), shininess
void oracle_scene_export_shapes(void* h, float* out) {
    Scene* sc = (Scene*)h;
    for (size_t i = 0; i < sc->shapes.size(); i++) {
        const Shape& s = sc->shapes[i]; float* o = out + i * 17;
        o[0] = (float)s.kind;
        o[1] = s.a.x; o[2] = s.a.y; o[3] = s.a.z; o[4] = s.b.x; o[5] = s.b.y; o[6] = s.b.z;
        o[7] = s.c.x; o[8] = s.c.y; o[9] = s.c.z;
        o[10] = s.diffuse.x; o[11] = s.diffuse.y; o[12] = s.diffuse.z;
        o[13] = s.specular.x; o[14] = s.specular.y; o[15] = s.specular.z; o[16] = s.shininess;
    }
}
// lights: 7 floats each = directional flag, v(3), color(3); camera: pos, look_at, up, fov, ambient(3)
void oracle_scene_export_lights(void* h, float* out) {
    Scene* sc = (Scene*)h;
    for (size_t i = 0; i < sc->lights.size(); i++) {
        const Light& l = sc->lights[i]; float* o = out + i * 7;
        o[0] = (float)l.directional; o[1] = l.v.x; o[2] = l.v.y; o[3] = l.v.z;
        o[4] = l.color.x; o[5] = l.color.y; o[6] = l.color.z;
    }
}
void oracle_scene_export_camera(void* h, float* out13) {
    Scene* sc = (Scene*)h;
    out13[0] = sc->cam_pos.x; out13[1] = sc->cam_pos.y; out13[2] = sc->cam_pos.z;
    out13[3] = sc->cam_look.x; out13[4] = sc->cam_look.y; out13[5] = sc->cam_look.z;
    out13[6] = sc->cam_up.x; out13[7] = sc->cam_up.y; out13[8] = sc->cam_up.z;
    out13[9] = sc->fov; out13[10] = sc->ambient.x; out13[11] = sc->ambient.y; out13[12] = sc->ambient.z;
}

// Frame constants of raytracer.rs:46-55 for a W x H frame: pos, cv, right, up (3 each), pw, ph, w2, h2.
void oracle_frame_setup(void* h, uint32_t W, uint32_t H, float* out16) {
    FrameSetup f = frame_setup(*(Scene*)h, W, H);
    const V3* v[4] = {&f.pos, &f.cv, &f.right, &f.up};
    for (int i = 0; i < 4; i++) { out16[3 * i] = v[i]->x; out16[3 * i + 1] = v[i]->y; out16[3 * i + 2] = v[i]->z; }
    out16[12] = f.pw; out16[13] = f.ph; out16[14] = f.w2; out16[15] = f.h2;
}

namespace {
int render_rows(Scene* sc, uint32_t W, uint32_t H, const std::vector<uint32_t>& rows, int mode, int threads,
                uint32_t* out, oracle_stats* stats) {
    if (mode < 0 || mode > 2) { g_err = "unknown mode"; return 1; }
    if (mode == MODE_TREE_COLLECT) build_tree(*sc, 1);
    if (mode == MODE_TREE_PRUNED) build_tree(*sc, 4);
    if (threads <= 0) threads = omp_get_max_threads();
    FrameSetup f = frame_setup(*sc, W, H);
    Counters total;
    double t0 = omp_get_wtime();
#pragma omp parallel num_threads(threads)
    {
        Tracer tr(*sc, mode);
#pragma omp for schedule(dynamic, 1)
        for (int64_t yy = 0; yy < (int64_t)rows.size(); yy++) {
            uint32_t y = rows[(size_t)yy];
            float b = (f.ph * (f.h2 - ((float)y + 0.5f))) / f.h2;                      // raytracer.rs:61
            for (uint32_t x = 0; x < W; x++) {
                float a = (f.pw * (((float)x + 0.5f) - f.w2)) / f.w2;                  // raytracer.rs:64
                V3 d = vnormalize(vadd(vadd(vscale(f.right, a), vscale(f.up, b)), f.cv));  // raytracer.rs:66
                out[(size_t)yy * W + x] = tr.find_color(f.pos, d);
            }
        }
#pragma omp critical
        {
            total.primary += tr.cnt.primary; total.shadow += tr.cnt.shadow; total.secondary += tr.cnt.secondary;
            total.nodes += tr.cnt.nodes; total.prims += tr.cnt.prims;
        }
    }
    double t1 = omp_get_wtime();
    if (stats) {
        stats->primary = total.primary; stats->shadow = total.shadow; stats->secondary = total.secondary;
        stats->nodes_visited = total.nodes; stats->prims_tested = total.prims;
        stats->render_ms = (t1 - t0) * 1e3; stats->threads = threads;
    }
    return 0;
}
}  // namespace

// Render rows [row0, row0+nrows) of a W x H frame (W = H = 0: the scene's own size) into
// out[nrows * W] (0xFFRRGGBB, row-major, first row on top).
// mode 0: every shape is a candidate (ground truth)
// mode 1: tree, collect every candidate, then test them all (what raytracer.rs:110-159 does)
// mode 2: tree with distance pruning and shadow early exit
int oracle_render(void* h, uint32_t W, uint32_t H, uint32_t row0, uint32_t nrows, int mode, int threads,
                  uint32_t* out, oracle_stats* stats) {
    Scene* sc = (Scene*)h;
    if (W == 0 || H == 0) { W = sc->width; H = sc->height; }
    if (row0 + nrows > H) { g_err = "row range outside the frame"; return 1; }
    std::vector<uint32_t> rows(nrows);
    for (uint32_t i = 0; i < nrows; i++) rows[i] = row0 + i;
    return render_rows(sc, W, H, rows, mode, threads, out, stats);
}

// Render the interleaved row tiles of one shard: tile t (tile_rows rows) belongs to shard
// t % nshards; the shard's rows are written compactly in increasing order.  Returns the
// number of rows through *nrows_out (out may be NULL to only ask for it).
int oracle_render_tiles(void* h, uint32_t W, uint32_t H, uint32_t shard, uint32_t nshards, uint32_t tile_rows, int mode,
                        int threads, uint32_t* out, uint32_t* nrows_out, oracle_stats* stats) {
    Scene* sc = (Scene*)h;
    if (W == 0 || H == 0) { W = sc->width; H = sc->height; }
    if (nshards == 0 || shard >= nshards || tile_rows == 0) { g_err = "bad shard description"; return 1; }
    std::vector<uint32_t> rows;
    for (uint32_t y = 0; y < H; y++) if ((y / tile_rows) % nshards == shard) rows.push_back(y);
    if (nrows_out) *nrows_out = (uint32_t)rows.size();
    if (!out) return 0;
    return render_rows(sc, W, H, rows, mode, threads, out, stats);
}

// Image::compare (imgcomparator/mod.rs:54-89) plus the two statistics the parity target is
// stated in.  a, b: n pixels, top byte ignored.  diff_out (optional): per-pixel packed
// channel differences, 0 where every channel differs by at most 1.
// out4: [0] pixels the reference comparator counts (any channel differing by more than 1),
//       [1] pixels whose 24-bit RGB is identical, [2] largest channel difference,
//       [3] pixels with some channel differing by more than 2.
void oracle_compare(const uint32_t* a, const uint32_t* b, uint64_t n, uint32_t* diff_out, uint64_t* out4) {
    uint64_t refdiff = 0, same = 0, maxd = 0, over2 = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint32_t p = a[i], q = b[i];
        int dr = abs((int)((p >> 16) & 0xFF) - (int)((q >> 16) & 0xFF));
        int dg = abs((int)((p >> 8) & 0xFF) - (int)((q >> 8) & 0xFF));
        int db = abs((int)(p & 0xFF) - (int)(q & 0xFF));
        uint32_t d = 0;
        if (!(dr <= 1 && dg <= 1 && db <= 1)) d = ((uint32_t)dr << 16) | ((uint32_t)dg << 8) | (uint32_t)db;
        if (d != 0) refdiff++;
        if (dr == 0 && dg == 0 && db == 0) same++;
        int m = std::max(dr, std::max(dg, db));
        if ((uint64_t)m > maxd) maxd = (uint64_t)m;
        if (m > 2) over2++;
        if (diff_out) diff_out[i] = d;
    }
    out4[0] = refdiff; out4[1] = same; out4[2] = maxd; out4[3] = over2;
}

float oracle_powf(float x, float y) { return powf(x, y); }

}  // extern "C"
