// render_kernels.hip -- the per-pixel render loop as gfx950 kernels.
//
// One thread owns one pixel start to finish: camera ray generation, closest hit over the
// always-tested list and the tree, one shadow ray per light, Lambert + Blinn-Phong shading and
// the bounded mirror recursion of the reference, unrolled into a forward pass that records one
// {local colour, material} per depth and a backward fold (so that the float operations happen
// in the reference's order).  A wavefront covers an 8x8 pixel tile and walks the path levels
// together; once few of its paths are left, the rays that depend on one hit only (its shadow
// rays and the closest-hit ray of its next level) are traced side by side, one per lane, by
// whichever lanes are free, and the pixel's own lane shades with the answers.
//
// Arithmetic contract.  Everything that decides a pixel is IEEE binary32 evaluated in exactly
// the order the reference evaluates it, with no multiply-add fusion (this file is compiled
// with -ffp-contract=off and says so again below), correctly rounded division and square
// root, and denormals preserved:
//   dot(a,b)     = (a.x*b.x + a.y*b.y) + a.z*b.z
//   cross(a,b)   = (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x)
//   normalize(a) = a * (1.0f / sqrtf(dot(a,a)))
// Box culling is the only place with freedom: it is conservative (padded boxes, slack on the
// pruning distance), so it never removes a shape the exact test would accept.
//
// The kernel is built twice: a tree that holds nothing but triangles (the blob header says what it holds) is rendered
// by a build without the dispatch on the kind of shape and with the triangle test as straight-line code; its frames
// are those of the generic build (RT_RENDER_KERNEL=generic in the environment selects that one for every scene).
//
// Reference code each part follows (paths relative to the reference tree):
//   pixel loop / ray generation   src/raytracer/raytracer.rs:57-72
//   quantise                      src/raytracer/raytracer.rs:85-91
//   closest hit                   src/raytracer/raytracer.rs:109-120
//   shadow test and guards        src/raytracer/raytracer.rs:126-159
//   shading                       src/raytracer/raytracer.rs:160-181
//   reflection                    src/raytracer/raytracer.rs:183-196
//   sphere / plane / triangle     src/raytracer/config/shape.rs:72-96, 112-132, 149-191
// The one library call of the path, powf (raytracer.rs:173), is the C library's algorithm in binary64 (specular_pow.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#ifdef RT_DIAG_LDS_PAD
#include <cstdlib>
#endif

#include "device_types.h"
#include "layout.h"
#include "specular_pow.h"

#pragma clang fp contract(off)


namespace {

using rtl::KIND_PLANE;
using rtl::KIND_SPHERE;
using rtl::KIND_TRIANGLE;
using rtl::REF_EMPTY;
using rtl::REF_LEAF;

constexpr uint32_t NO_PRIM = 0xFFFFFFFFu;
constexpr uint32_t LIST_FLAG = 0x80000000u;
constexpr float INF = __builtin_huge_valf();

// ------------------------------------------------------------------ vector primitives
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 v3(const float4 q) { return v3(q.x, q.y, q.z); }
__device__ __forceinline__ V3 v3(const float* p) { return v3(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 vadd(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 vscale(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 vcmul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 vneg(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float vdot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 vcross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float vnorm(V3 a) { return sqrtf(vdot(a, a)); }
__device__ __forceinline__ V3 vnormalize(V3 a) { float inv = 1.0f / vnorm(a); return vscale(a, inv); }

// Loads whose address is the same in every lane (the always-tested shapes, the lights) go through the scalar
// data cache: read through the constant address space the compiler issues s_load for them, which keeps them
// out of the vector memory path that the per-lane node and shape loads need.  The scene is never written by a kernel.
typedef float RawF4 __attribute__((ext_vector_type(4)));
typedef const RawF4 __attribute__((address_space(4)))* UniformF4;
__device__ __forceinline__ float4 uniform_load(const float4* base, uint32_t index) {
    const RawF4 v = ((UniformF4)(uintptr_t)base)[index];
    return make_float4(v.x, v.y, v.z, v.w);
}

struct Counters {
    uint32_t nodes, prims, shadow, secondary, wave_node_iters, wave_prim_iters;
    uint32_t wave_closest, wave_hit, wave_shadow, wave_shade;      // wavefront-level passes through the phases of a path level
    uint32_t wave_shadow_node_iters, wave_shadow_prim_iters;       // the share of shadow rays (traced light by light) in wave_node_iters / wave_prim_iters
    uint32_t wave_rounds, wave_round_node_iters, wave_round_prim_iters;   // side-by-side rounds, and their share in wave_node_iters / wave_prim_iters
};

// Diagnostic builds (-DRT_MARKS, compiled to assembly only) leave comments in the instruction stream, from which
// scripts/phase_table.py counts the instructions of each phase.
#ifdef RT_MARKS
#define MARK(name) asm volatile("; MARK " name)
#define MARK_AT(name, site) asm volatile("; MARK " name "_%c0" : : "i"(site))      // a mark inside a function that is inlined at several sites
#else
#define MARK(name)
#define MARK_AT(name, site)
#endif
enum Site { SITE_CLOSEST_LIST, SITE_CLOSEST_TREE, SITE_SHADOW_LIST, SITE_SHADOW_TREE, SITE_ROUND_LIST, SITE_ROUND_TREE };

// True on exactly one of the currently active lanes.
__device__ __forceinline__ bool first_active_lane() { return (threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)__ballot(1)) - 1); }

// ------------------------------------------------------------------ exact shape tests
// Each returns whether the reference's test yields Some(..) and the distance it would carry.

// e1 = v1 - v0 and e2 = v2 - v0 (shape.rs:149-150) come ready-made from the shape record.
__device__ __forceinline__ bool test_triangle(V3 v0, V3 e1, V3 e2, V3 o, V3 d, float* t_out) {   // shape.rs:151-176
    V3 h = vcross(d, e2);
    float a = vdot(e1, h);
    if (fabsf(a) < 1e-6f) return false;
    float f = 1.0f / a;
    V3 s = vsub(o, v0);
    float u = f * vdot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    V3 q = vcross(s, e1);
    float v = f * vdot(d, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = f * vdot(e2, q);
    if (t < 0.0f) return false;
    *t_out = t;
    return true;
}

// The same test as straight-line code: every quantity is computed, the four rejections are combined at the end.  Each condition
// sees the value the reference would have computed had it got that far, and the reference rejects as soon as one of them
// holds, so the outcome (and t for an accepted hit) is the same; what a rejected candidate leaves in *t_out is not used.
// In a wavefront that tests a leaf per lane some lane nearly always reaches the end, so leaving early saves little, while
// the exec-mask bookkeeping of four nested exits costs more scalar instructions than the test has vector ones.
__device__ __forceinline__ bool test_triangle_straight(V3 v0, V3 e1, V3 e2, V3 o, V3 d, float* t_out) {   // shape.rs:151-176
    const V3 h = vcross(d, e2);
    const float a = vdot(e1, h);
    const float f = 1.0f / a;
    const V3 s = vsub(o, v0);
    const float u = f * vdot(s, h);
    const V3 q = vcross(s, e1);
    const float v = f * vdot(d, q);
    const float t = f * vdot(e2, q);
    *t_out = t;
    return !(fabsf(a) < 1e-6f) & !(u < 0.0f || u > 1.0f) & !(v < 0.0f || u + v > 1.0f) & !(t < 0.0f);
}

__device__ __forceinline__ bool test_sphere(V3 c, float radius, V3 o, V3 d, float* t_out) {      // shape.rs:72-83
    V3 oc = vsub(o, c);
    float half_b = vdot(oc, d);
    float cc = vdot(oc, oc) - radius * radius;
    float disc = half_b * half_b - cc;
    if (disc < 0.0f) return false;
    float t = -half_b - sqrtf(disc);
    if (t < 0.0f) return false;
    *t_out = t;
    return true;
}

__device__ __forceinline__ bool test_plane(V3 p, V3 n, V3 o, V3 d, float* t_out) {               // shape.rs:112-120
    float denom = vdot(n, d);
    if (fabsf(denom) < 1e-6f) return false;
    float t = vdot(vsub(p, o), n) / denom;
    if (t < 0.0f) return false;
    *t_out = t;
    return true;
}

// TRI: the caller knows the shape to be a triangle (a tree that holds nothing else): no dispatch, straight-line test.
template <int SITE, bool TRI = false>
__device__ __forceinline__ bool test_shape(const float4 q0, const float4 q1, const float4 q2, V3 o, V3 d, float* t_out) {
    if (TRI) { MARK_AT("triangle", SITE); return test_triangle_straight(v3(q0), v3(q1), v3(q2), o, d, t_out); }
    uint32_t kind = __float_as_uint(q0.w) & 3u;
    if (kind == KIND_TRIANGLE) { MARK_AT("triangle", SITE); return test_triangle(v3(q0), v3(q1), v3(q2), o, d, t_out); }
    if (kind == KIND_SPHERE) { MARK_AT("sphere", SITE); return test_sphere(v3(q0), q1.x, o, d, t_out); }
    MARK_AT("plane", SITE);
    return test_plane(v3(q0), v3(q2), o, d, t_out);
}

// is_back_face of the reference's Intersection: only triangles can be back faces (shape.rs:94,131,181).
__device__ __forceinline__ bool shape_back_face(const float4 q0, const float4 q1, const float4 q2, V3 d) {
    uint32_t kind = __float_as_uint(q0.w) & 3u;
    if (kind != KIND_TRIANGLE) return false;
    V3 n = vnormalize(vcross(v3(q1), v3(q2)));       // the record holds the edges
    return vdot(n, d) > 0.0f;
}

// ------------------------------------------------------------------ box culling (conservative)
// Node boxes are 16-bit coordinates on the scene grid: plane q on an axis is origin + units(q) * scale (layout.h).
// Along a ray o + t d its parameter is (origin + units*scale - o) / d = units * A + B with, per ray and axis,
// A = scale / d and B = (origin - o) / d.  A child record keeps the low and the high plane of an axis in one word;
// one byte permute per plane builds the float x = 32768 + units(q) from it (bytes {0, q.lo, q.hi, 0x47}: no
// integer-to-float conversion), and the ray's selector for that permute already picks the plane the ray enters
// through (the low one when d >= 0) or the one it leaves through, so no minimum / maximum per axis is needed
// afterwards.  The parameter is then x * A + (B - 32768 A): one packed multiply-add gives entry and exit of an
// axis together.  With e = 2^-24: 1/d comes from the hardware reciprocal (off by at most 2e), so A is within 3e and
// B within 4e of their exact values, and B - 32768 A, the shifted constant and the multiply-add round once each,
// every time to a value below 131072 |A| + |B|; with units up to 98302 that is an error below
// e (688122 |A| + 7 |B|) in the parameter.  delta = 2^-21 * (131072 |A| + |B|) = e (1048576 |A| + 8 |B|) covers it:
// the entry constant is lowered and the exit constant raised by delta, so the computed interval always contains
// the exact one.  Nothing here decides a pixel: it only
// has to keep every shape reachable that the exact tests would accept.
typedef float f2 __attribute__((ext_vector_type(2)));
struct RayBox {
    f2 Ax, Ay, Az;            // {A, A} per axis
    f2 Bx, By, Bz;            // {entry constant, exit constant} per axis
    uint32_t sel_in[3], sel_out[3];   // permute selectors: which half of the axis word is entered / left through
};

__device__ __forceinline__ float safe_inverse(float d, float limit) {
    // A zero (or denormal) direction component would make A and B infinite and their sum NaN;
    // a huge finite slope keeps the slab semantics (inside: unbounded interval, outside: empty).
    // The hardware reciprocal (within one unit in the last place) instead of a division: culling only has to be
    // conservative, and the slack on the slab constants covers it (make_axis).
    float inv = __builtin_amdgcn_rcpf(d);
    return fminf(fmaxf(inv, -limit), limit);
}

__device__ __forceinline__ void make_axis(float scale, float origin, float limit, float o, float d, f2* A, f2* B, uint32_t* sel_in, uint32_t* sel_out) {
    const float inv = safe_inverse(d, limit);
    const float a = scale * inv;
    const float b = fminf(fmaxf((origin - o) * inv, -1e37f), 1e37f);
    const float delta = 4.76837158203125e-07f * (131072.0f * fabsf(a) + fabsf(b));      // 2^-21 (...)
    const float shifted = b - 32768.0f * a;
    *A = f2{a, a};
    *B = f2{shifted - delta, shifted + delta};
    // selector bytes, lowest first: 0x0C -> 0x00; 0/1 or 2/3 -> low or high half of the axis word; 7 -> 0x47 of the constant
    const bool forward = inv >= 0.0f;
    *sel_in = forward ? 0x0701000Cu : 0x0703020Cu;
    *sel_out = forward ? 0x0703020Cu : 0x0701000Cu;
}

__device__ __forceinline__ RayBox make_raybox(const DevScene& sc, V3 o, V3 d) {
    RayBox r;
    make_axis(sc.grid_scale[0], sc.grid_origin[0], sc.inv_limit[0], o.x, d.x, &r.Ax, &r.Bx, &r.sel_in[0], &r.sel_out[0]);
    make_axis(sc.grid_scale[1], sc.grid_origin[1], sc.inv_limit[1], o.y, d.y, &r.Ay, &r.By, &r.sel_in[1], &r.sel_out[1]);
    make_axis(sc.grid_scale[2], sc.grid_origin[2], sc.inv_limit[2], o.z, d.z, &r.Az, &r.Bz, &r.sel_in[2], &r.sel_out[2]);
    return r;
}

__device__ __forceinline__ f2 planes_of(uint32_t axis_word, uint32_t sel_in, uint32_t sel_out) {
    const uint32_t K = 0x47000000u;       // exponent byte of 32768 + units(q)
    return f2{__uint_as_float(__builtin_amdgcn_perm(K, axis_word, sel_in)), __uint_as_float(__builtin_amdgcn_perm(K, axis_word, sel_out))};
}

__device__ __forceinline__ bool slab(const uint4 c, const RayBox& rb, float tlim, float* tnear) {
    const f2 tx = __builtin_elementwise_fma(planes_of(c.x, rb.sel_in[0], rb.sel_out[0]), rb.Ax, rb.Bx);
    const f2 ty = __builtin_elementwise_fma(planes_of(c.y, rb.sel_in[1], rb.sel_out[1]), rb.Ay, rb.By);
    const f2 tz = __builtin_elementwise_fma(planes_of(c.z, rb.sel_in[2], rb.sel_out[2]), rb.Az, rb.Bz);
    const float tn = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, 0.0f));
    // The limit goes in through a bare v_min_f32: the compiler would otherwise re-canonicalise it in every pass
    // (it cannot see through the loop that the value is never a signalling NaN).
    float zlim;
    asm("v_min_f32 %0, %1, %2" : "=v"(zlim) : "v"(tz.y), "v"(tlim));
    const float tf = fminf(fminf(tx.y, ty.y), zlim);
    *tnear = tn;
    return tn <= tf;
}

// Pruning bound for a closest hit at computed distance t: shapes whose computed distance could
// still come out smaller (their exact test rounds differently) must stay reachable.
__device__ __forceinline__ float prune_limit(float t) { return t * 1.0005f + 1e-4f; }

struct Best { float t; uint32_t prim; uint32_t order; };

// raytracer.rs:113-120: smallest distance wins; on exactly equal distances the shape that
// comes first in the scene file wins.
__device__ __forceinline__ void consider(Best* b, float t, uint32_t prim, uint32_t order) {
    bool take = (b->prim == NO_PRIM) || (b->t > t) || (b->t == t && order < b->order);
    if (take) { b->t = t; b->prim = prim; b->order = order; }
}

template <bool COUNT, bool TRI>
__device__ __forceinline__ void closest_hit(const DevScene& sc, V3 o, V3 d, uint32_t* stack, Best* best, Counters* cnt) {
    best->t = INF; best->prim = NO_PRIM; best->order = 0xFFFFFFFFu;
    MARK("closest_list");
    for (uint32_t i = 0; i < sc.n_list; i++) {
        const float4 q0 = uniform_load(sc.list, 3 * i), q1 = uniform_load(sc.list, 3 * i + 1), q2 = uniform_load(sc.list, 3 * i + 2);
        float t;
        if (COUNT) cnt->prims++;
        const bool hit = test_shape<SITE_CLOSEST_LIST>(q0, q1, q2, o, d, &t);
        MARK_AT("tested", SITE_CLOSEST_LIST);
        if (hit) consider(best, t, LIST_FLAG | i, __float_as_uint(q1.w));
    }
    uint32_t cur = sc.root_ref;
    if (cur == REF_EMPTY) return;
    MARK("closest_raybox");
    const RayBox rb = make_raybox(sc, o, d);
    float tlim = best->prim == NO_PRIM ? INF : prune_limit(best->t);
    // Entries 0 and 1 of every lane's stack hold REF_EMPTY: popping needs no emptiness test, and the two entries below
    // the top can always be read.  `low` points two entries below the next free one (low[128]), so that all three
    // accesses of a step take non-negative offsets, which fit the immediate field of the LDS instructions.
    uint32_t* low = stack;
    // All lanes descend through inner nodes until each has reached a leaf (or run out of nodes),
    // then all test their leaves together.  A lane that reaches its leaf early sets it aside and
    // keeps descending towards its next one instead of idling (it then tests both).
    // A step is straight-line code: the two entries below the top are read while the node is on its way (a pop, and a
    // pop right after a pop, then cost no LDS round trip of their own), the rest are selects.
    // The descent is a loop the whole wavefront goes round together, with the step under a condition of the lane's own
    // (a lane whose `cur` is a leaf sits the rounds out): a loop that lanes leave one by one costs a dozen scalar
    // instructions of mask bookkeeping per round more (2 % of the frame).
    uint32_t held = 0;                                  // leaf set aside (leaf references are never 0)
    for (;;) {
        MARK("closest_box");
        do {
            if (!(cur & REF_LEAF)) {
                const uint4* n = (const uint4*)((const char*)sc.nodes + cur);   // an inner reference is the pair's byte offset; 32 bits keep the address in one register
                const uint4 c0 = n[0], c1 = n[1];
                uint32_t below = low[64], below2 = low[0];
                asm volatile("" : "+v"(below), "+v"(below2));       // keeps the reads here, under the node loads, instead of behind the box tests
                float t0, t1;
                bool h0 = slab(c0, rb, tlim, &t0);
                bool h1 = slab(c1, rb, tlim, &t1);
                if (COUNT) { cnt->nodes += 2; if (first_active_lane()) cnt->wave_node_iters++; }
                const uint32_t r0 = c0.w, r1 = c1.w;
                const bool both = h0 && h1, any = h0 || h1, swap = t1 < t0;
                const uint32_t far_child = swap ? r0 : r1;
                low[128] = far_child;                               // a push when both are hit; otherwise a store to the free entry above the top, cheaper than a branch around it
                // where the step leads, and what a pop would return from there
                const uint32_t next = both ? (swap ? r1 : r0) : (any ? (h0 ? r0 : r1) : below);
                const uint32_t under = both ? far_child : (any ? below : below2);
                // a lane without a leaf in hand sets the one it has reached aside and goes on with what lies under it
                const bool take = held == 0 && (next & REF_LEAF) && next != REF_EMPTY;
                held = take ? next : held;
                cur = take ? under : next;
                low = (uint32_t*)((char*)low + ((both ? 256 : (any ? 0 : -256)) + (take ? -256 : 0)));   // 256 bytes: one entry of every lane
            }
            // until nobody is still looking for a first leaf: no lane with empty hands and an inner node before it (one compare per lane)
        } while (__builtin_amdgcn_ballot_w64((held | (cur & REF_LEAF)) == 0u) != 0);
        MARK("closest_shape");
        if (held == 0 && (cur & REF_LEAF) && cur != REF_EMPTY) {  // the root or a popped entry can be a leaf
            held = cur;
            low -= 64; cur = low[128];
        }
        while (held != 0) {
            uint32_t first = (held & ~REF_LEAF) >> 3, count = (held & 7u) + 1u;
            uint32_t below = low[64];
            asm volatile("" : "+v"(below));                     // read under the shape loads
            for (uint32_t k = 0; k < count; k++) {
                const float4* p = (const float4*)((const char*)sc.tree + (first + k) * 48u);      // fewer than 2^26 shapes: the byte offset fits 32 bits and the address one register
                const float4 q0 = p[0], q1 = p[1], q2 = p[2];
                float t;
                if (COUNT) { cnt->prims++; if (first_active_lane()) cnt->wave_prim_iters++; }
                const bool hit = test_shape<SITE_CLOSEST_TREE, TRI>(q0, q1, q2, o, d, &t);
                MARK_AT("tested", SITE_CLOSEST_TREE);
                if (hit) consider(best, t, first + k, __float_as_uint(q1.w));
            }
            tlim = prune_limit(best->t);                        // (infinite while nothing has been hit)
            held = 0;
            if ((cur & REF_LEAF) && cur != REF_EMPTY) {
                held = cur;
                low -= 64; cur = below;
            }
        }
        if (cur == REF_EMPTY) break;
    }
}

// The guards of raytracer.rs:143-159 for a candidate whose exact test gave `hit` and distance t.  Most candidates of a shadow
// ray are missed by every lane, so the guards sit behind a branch (measured against selects: 1 % in their favour); the
// other one is around the occluder's own facing, which is needed for back-face hits only.
template <int SITE>
__device__ __forceinline__ bool occludes_at(bool hit, const float4 q0, const float4 q1, const float4 q2, V3 d, float t, bool hit_back, bool point_light, float light_dist) {
    if (!hit) return false;
    bool blocks = !(t < 1e-6f) & (!point_light | (t < light_dist));
    if (blocks && hit_back) {
        MARK_AT("occluder_back_face", SITE);
        blocks = !shape_back_face(q0, q1, q2, d);
    }
    MARK_AT("guards", SITE);
    return blocks;
}

// raytracer.rs:143-159 for one candidate.
template <int SITE>
__device__ __forceinline__ bool occludes(const float4 q0, const float4 q1, const float4 q2, V3 o, V3 d, bool hit_back, bool point_light,
                                         float light_dist) {
    float t;
    const bool hit = test_shape<SITE>(q0, q1, q2, o, d, &t);
    MARK_AT("tested", SITE);
    if (!hit) return false;
    if (t < 1e-6f) return false;
    if (hit_back) {
        MARK_AT("occluder_back_face", SITE);
        if (shape_back_face(q0, q1, q2, d)) return false;
    }
    MARK_AT("guards", SITE);
    if (point_light) return t < light_dist;
    return true;
}

template <bool COUNT, bool TRI>
__device__ __forceinline__ bool in_shadow(const DevScene& sc, V3 o, V3 d, bool hit_back, bool point_light, float light_dist, uint32_t* stack,
                                          Counters* cnt) {
    MARK("shadow_list");
    for (uint32_t i = 0; i < sc.n_list; i++) {
        if (COUNT) cnt->prims++;
        if (occludes<SITE_SHADOW_LIST>(uniform_load(sc.list, 3 * i), uniform_load(sc.list, 3 * i + 1), uniform_load(sc.list, 3 * i + 2), o, d, hit_back, point_light, light_dist)) return true;
    }
    uint32_t cur = sc.root_ref;
    if (cur == REF_EMPTY) return false;
    MARK("shadow_raybox");
    const RayBox rb = make_raybox(sc, o, d);
    float tlim = point_light ? prune_limit(light_dist) : INF;
    // The loop of closest_hit without the ordering: all lanes descend until each has a leaf in hand, and a lane that reaches its leaf
    // early sets it aside and goes on towards its next one instead of idling, then tests both.  (What it descends on behalf of a leaf
    // that turns out to hold an occluder is wasted; the lanes it no longer keeps waiting are worth more.)
    uint32_t* low = stack;                              // as in closest_hit: the next free entry is low[128]
    uint32_t held = 0;
    for (;;) {
        MARK("shadow_box");
        do {
            if (!(cur & REF_LEAF)) {
                const uint4* n = (const uint4*)((const char*)sc.nodes + cur);   // an inner reference is the pair's byte offset; 32 bits keep the address in one register
                const uint4 c0 = n[0], c1 = n[1];
                uint32_t below = low[64], below2 = low[0];
                asm volatile("" : "+v"(below), "+v"(below2));       // keeps the reads here, under the node loads, instead of behind the box tests
                float t0, t1;
                bool h0 = slab(c0, rb, tlim, &t0);
                bool h1 = slab(c1, rb, tlim, &t1);
                if (COUNT) { cnt->nodes += 2; if (first_active_lane()) { cnt->wave_node_iters++; cnt->wave_shadow_node_iters++; } }
                const uint32_t r0 = c0.w, r1 = c1.w;
                const bool both = h0 && h1, any = h0 || h1;
                low[128] = r1;                                      // as in closest_hit: a push, or a store to the free entry above the top
                const uint32_t next = h0 ? r0 : (h1 ? r1 : below);
                const uint32_t under = both ? r1 : (any ? below : below2);
                const bool take = held == 0 && (next & REF_LEAF) && next != REF_EMPTY;
                held = take ? next : held;
                cur = take ? under : next;
                low = (uint32_t*)((char*)low + ((both ? 256 : (any ? 0 : -256)) + (take ? -256 : 0)));   // 256 bytes: one entry of every lane
            }
        } while (__builtin_amdgcn_ballot_w64((held | (cur & REF_LEAF)) == 0u) != 0);
        MARK("shadow_shape");
        if (held == 0 && (cur & REF_LEAF) && cur != REF_EMPTY) {  // the root or a popped entry can be a leaf
            held = cur;
            low -= 64; cur = low[128];
        }
        while (held != 0) {
            uint32_t first = (held & ~REF_LEAF) >> 3, count = (held & 7u) + 1u;
            uint32_t below = low[64];
            asm volatile("" : "+v"(below));                     // read under the shape loads
            for (uint32_t k = 0; k < count; k++) {
                const float4* p = (const float4*)((const char*)sc.tree + (first + k) * 48u);      // fewer than 2^26 shapes: the byte offset fits 32 bits and the address one register
                const float4 q0 = p[0], q1 = p[1], q2 = p[2];
                float t;
                if (COUNT) { cnt->prims++; if (first_active_lane()) { cnt->wave_prim_iters++; cnt->wave_shadow_prim_iters++; } }
                const bool hit = test_shape<SITE_SHADOW_TREE, TRI>(q0, q1, q2, o, d, &t);
                MARK_AT("tested", SITE_SHADOW_TREE);
                if (occludes_at<SITE_SHADOW_TREE>(hit, q0, q1, q2, d, t, hit_back, point_light, light_dist)) return true;
            }
            held = 0;
            if ((cur & REF_LEAF) && cur != REF_EMPTY) {
                held = cur;
                low -= 64; cur = below;
            }
        }
        if (cur == REF_EMPTY) break;
    }
    return false;
}


// Most rounds of 64 jobs a side-by-side round may take (0: never, every ray is traced by its own lane, kind after kind).
// One: the jobs of a level have to fit the wavefront (4K dragon: 2.51 ms; never 2.66; two rounds in place of three kinds,
// which paid while the loops of the kinds were slower, 2.55).
#ifndef RT_SIDE_ROUNDS_MAX
#define RT_SIDE_ROUNDS_MAX 1u
#endif
// One ray of either kind through the loop of closest_hit: a closest-hit ray as there, a shadow ray (any_hit) with the guards of
// raytracer.rs:143-159 and an end at the first occluder.  For rounds in which the lanes of a wavefront carry rays of both kinds.
// Result: *best for a closest-hit ray; the return value says whether a shadow ray found an occluder.
template <bool COUNT, bool TRI>
__device__ __forceinline__ bool trace_job(const DevScene& sc, V3 o, V3 d, bool any_hit, bool hit_back, bool point_light, float light_dist,
                                          uint32_t* stack, Best* best, Counters* cnt) {
    best->t = INF; best->prim = NO_PRIM; best->order = 0xFFFFFFFFu;
    bool occluded = false;
    MARK("round_list");
    for (uint32_t i = 0; i < sc.n_list && !occluded; i++) {
        const float4 q0 = uniform_load(sc.list, 3 * i), q1 = uniform_load(sc.list, 3 * i + 1), q2 = uniform_load(sc.list, 3 * i + 2);
        float t;
        if (COUNT) cnt->prims++;
        if (!test_shape<SITE_ROUND_LIST>(q0, q1, q2, o, d, &t)) continue;
        MARK_AT("tested", SITE_ROUND_LIST);
        if (!any_hit) consider(best, t, LIST_FLAG | i, __float_as_uint(q1.w));
        else occluded = occludes_at<SITE_ROUND_LIST>(true, q0, q1, q2, d, t, hit_back, point_light, light_dist);
    }
    uint32_t cur = occluded ? REF_EMPTY : sc.root_ref;
    if (cur == REF_EMPTY) return occluded;
    MARK("round_raybox");
    const RayBox rb = make_raybox(sc, o, d);
    float tlim = any_hit ? (point_light ? prune_limit(light_dist) : INF) : (best->prim == NO_PRIM ? INF : prune_limit(best->t));
    uint32_t* low = stack;
    uint32_t held = 0;
    for (;;) {
        MARK("round_box");
        do {
            if (!(cur & REF_LEAF)) {
                const uint4* n = (const uint4*)((const char*)sc.nodes + cur);
                const uint4 c0 = n[0], c1 = n[1];
                uint32_t below = low[64], below2 = low[0];
                asm volatile("" : "+v"(below), "+v"(below2));
                float t0, t1;
                bool h0 = slab(c0, rb, tlim, &t0);
                bool h1 = slab(c1, rb, tlim, &t1);
                if (COUNT) { cnt->nodes += 2; if (first_active_lane()) { cnt->wave_node_iters++; cnt->wave_round_node_iters++; } }
                const uint32_t r0 = c0.w, r1 = c1.w;
                const bool both = h0 && h1, any = h0 || h1, swap = t1 < t0;
                const uint32_t far_child = swap ? r0 : r1;
                low[128] = far_child;
                const uint32_t next = both ? (swap ? r1 : r0) : (any ? (h0 ? r0 : r1) : below);
                const uint32_t under = both ? far_child : (any ? below : below2);
                const bool take = held == 0 && (next & REF_LEAF) && next != REF_EMPTY;
                held = take ? next : held;
                cur = take ? under : next;
                low = (uint32_t*)((char*)low + ((both ? 256 : (any ? 0 : -256)) + (take ? -256 : 0)));
            }
        } while (__builtin_amdgcn_ballot_w64((held | (cur & REF_LEAF)) == 0u) != 0);
        MARK("round_shape");
        if (held == 0 && (cur & REF_LEAF) && cur != REF_EMPTY) {  // the root or a popped entry can be a leaf
            held = cur;
            low -= 64; cur = low[128];
        }
        while (held != 0) {
            uint32_t first = (held & ~REF_LEAF) >> 3, count = (held & 7u) + 1u;
            uint32_t below = low[64];
            asm volatile("" : "+v"(below));
            for (uint32_t k = 0; k < count && !occluded; k++) {
                const float4* p = (const float4*)((const char*)sc.tree + (first + k) * 48u);      // fewer than 2^26 shapes: the byte offset fits 32 bits and the address one register
                const float4 q0 = p[0], q1 = p[1], q2 = p[2];
                float t;
                if (COUNT) { cnt->prims++; if (first_active_lane()) { cnt->wave_prim_iters++; cnt->wave_round_prim_iters++; } }
                const bool hit = test_shape<SITE_ROUND_TREE, TRI>(q0, q1, q2, o, d, &t);
                MARK_AT("tested", SITE_ROUND_TREE);
                if (!hit) continue;
                if (!any_hit) consider(best, t, first + k, __float_as_uint(q1.w));
                else occluded = occludes_at<SITE_ROUND_TREE>(true, q0, q1, q2, d, t, hit_back, point_light, light_dist);
            }
            tlim = any_hit ? tlim : prune_limit(best->t);
            held = 0;
            if (occluded) { cur = REF_EMPTY; break; }
            if ((cur & REF_LEAF) && cur != REF_EMPTY) {
                held = cur;
                low -= 64; cur = below;
            }
        }
        if (cur == REF_EMPTY) break;
    }
    return occluded;
}

// The value `v` has in lane `from` (every lane names its own source).
__device__ __forceinline__ float pull(float v, uint32_t from) { return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(from << 2), __float_as_int(v))); }
__device__ __forceinline__ uint32_t pull(uint32_t v, uint32_t from) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from << 2), (int)v); }

__device__ __forceinline__ uint32_t quantise(V3 c) {                                             // raytracer.rs:85-91
    uint32_t r = (uint32_t)roundf(fminf(fmaxf(c.x, 0.0f), 1.0f) * 255.0f);
    uint32_t g = (uint32_t)roundf(fminf(fmaxf(c.y, 0.0f), 1.0f) * 255.0f);
    uint32_t b = (uint32_t)roundf(fminf(fmaxf(c.z, 0.0f), 1.0f) * 255.0f);
    return (255u << 24) | (r << 16) | (g << 8) | b;
}

}  // namespace

// Local row r of a shard -> row of the frame.  Tile t (tile_rows rows) belongs to shard t % nshards.
__device__ __forceinline__ uint32_t frame_row(const DevFrame& f, uint32_t r) {
    uint32_t lt = r / f.tile_rows, within = r - lt * f.tile_rows;
    return (lt * f.nshards + f.shard) * f.tile_rows + within;
}

// Traversal stacks, [entry][lane]: sized at launch for the tree at hand (its depth plus the sentinel entry).
extern __shared__ uint32_t s_stack[];

// raytracer.rs:173.  On the C library's main path (a positive normal base: every n.h that is not exactly zero) its own
// algorithm in binary64 with the tables in LDS; +0 to a positive power is +0; whatever else comes in goes to the device library.
__device__ __forceinline__ float specular_power(float base, float exponent, const uint64_t* tables) {
#ifndef RT_LIBRARY_POWF      // (for comparisons: everything through the device library)
    if (rtpow::pow_specular_domain(base, exponent)) { MARK("shade_pow_main_path"); return rtpow::pow_specular(base, exponent, tables); }
#endif
    MARK("shade_pow_off_the_main_path");
    if (__float_as_uint(base) == 0u && exponent > 0.0f && exponent < INF) return 0.0f;
    return powf(base, exponent);
}

// FOLD_LDS: the {local colour, material} records of the reflection levels live in LDS behind the stacks
// ([level][lane], 16 bytes each), those of the deeper levels in the global `fold` array ([level - lds_levels][pixel]).
// TRI: the tree holds nothing but triangles (the builders say so in the blob header).
template <bool COUNT, bool TRI>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RT_WAVES, RT_WAVES))) void render_kernel(DevScene sc, DevFrame f, uint32_t* __restrict__ out,
                                                                                                        float4* __restrict__ fold, unsigned long long* __restrict__ counters,
                                                                                                        uint32_t stack_entries, uint32_t lds_levels) {
    const uint32_t lane = threadIdx.x;
    float4* const fold_lds = (float4*)(s_stack + stack_entries * 64u) + lane;
    uint64_t* const pow_tables = (uint64_t*)(s_stack + stack_entries * 64u + lds_levels * 256u);     // behind the records: RT_POW_LDS_BYTES
    static_assert(rtpow::TABLE_WORDS == 64 && RT_POW_LDS_BYTES == 64 * sizeof(uint64_t), "one table word per lane");
    pow_tables[lane] = rtpow::TABLES[lane];
    // Which 8x8 pixel tile this wavefront renders.  Workgroups are dealt round-robin over the 8
    // XCDs, each with an L2 of its own: in row-major order every XCD would touch every part of
    // the scene.  For big frames the blocks of one XCD instead sweep a square region of tiles at a
    // time, so that the rays in flight on an XCD share the nodes and shapes in its L2.
    uint32_t tile_x, tile_y, region = 0;
    const unsigned long long started = f.region_cost ? __builtin_amdgcn_s_memrealtime() : 0ull;
    if (f.region_log2 == 0) {
        const uint32_t tiles_x = (f.width + 7u) >> 3;
        tile_x = blockIdx.x % tiles_x; tile_y = blockIdx.x / tiles_x;
    } else {
        const uint32_t per_log2 = 2u * f.region_log2, side_mask = (1u << f.region_log2) - 1u;
        const uint32_t xcd_slot = blockIdx.x & 7u, j = blockIdx.x >> 3;
        const uint32_t turn = ((j >> per_log2) << 3) + xcd_slot, t = j & ((1u << per_log2) - 1u);
        if (turn >= f.n_regions) return;
        region = f.region_order ? f.region_order[turn] : turn;
        const uint32_t rx = region % f.regions_x, ry = region / f.regions_x;
        tile_x = (rx << f.region_log2) + (t & side_mask); tile_y = (ry << f.region_log2) + (t >> f.region_log2);
    }
    const uint32_t x = tile_x * 8u + (lane & 7u), r = tile_y * 8u + (lane >> 3);
    uint32_t* stack = s_stack + lane;
    stack[0] = REF_EMPTY; stack[64] = REF_EMPTY;
    Counters cnt = {};
    const bool live = x < f.width && r < f.local_rows;
    // The wavefront walks the path levels together (the level is the same number in every lane); `alive` says whether this lane's
    // path still goes on.  Once so few paths are left that the shadow rays of every live hit AND the closest-hit ray of its next
    // level fit the 64 lanes, they are traced side by side, one ray per lane, by whoever has a lane to spare: those rays
    // depend on nothing but the hit.  The round lasts as long as its slowest ray instead of the slowest closest-hit ray plus the
    // slowest ray to each light.  Each job rebuilds its ray from the owner's hit with the owner's arithmetic; the owner shades
    // with the answers, light after light in the reference's order.
    {
        const uint32_t y = frame_row(f, r);
        const size_t pix = (size_t)r * f.width + x;
        const size_t npix = (size_t)f.local_rows * f.width;
        MARK("raygen");
        float b = (f.ph * (f.h2 - ((float)y + 0.5f))) / f.h2;
        float a = (f.pw * (((float)x + 0.5f) - f.w2)) / f.w2;
        V3 right = v3(f.right), up = v3(f.up), cv = v3(f.cv);
        V3 d = vnormalize(vadd(vadd(vscale(right, a), vscale(up, b)), cv));
        V3 o = v3(f.pos);
        const V3 ambient = v3(f.ambient);

        V3 cur = v3(0.0f, 0.0f, 0.0f);   // colour returned by the deepest level
        bool alive = live;
        uint32_t levels = 0;             // reflection records this lane has left behind
        Best next = {0.0f, NO_PRIM, 0u};
        bool have_next = false;          // (the same in every lane) the closest hit of this level was found in the round before
        for (uint32_t depth = 0; __builtin_amdgcn_ballot_w64(alive) != 0; depth++) {
            Best best = next;
            if (alive && !have_next) {
                if (COUNT && first_active_lane()) cnt.wave_closest++;
                closest_hit<COUNT, TRI>(sc, o, d, stack, &best, &cnt);
            }
            MARK("hit");
            if (alive && best.prim == NO_PRIM) { cur = v3(0.0f, 0.0f, 0.0f); alive = false; }   // raytracer.rs:199-201
            V3 point = v3(0.0f, 0.0f, 0.0f), normal = point, shadow_origin = point, diffuse_color = point, specular_color = point;
            float shininess = 0.0f;
            uint32_t mat = 0;
            bool back = false, goes_on = false;
            if (alive) {
                // Rebuild the reference's Intersection for the winner (shape.rs:84-95, 122-132, 178-191).
                if (COUNT && first_active_lane()) cnt.wave_hit++;
                const float4* rec = (best.prim & LIST_FLAG) ? sc.list + 3 * (size_t)(best.prim & ~LIST_FLAG) : sc.tree + 3 * (size_t)best.prim;
                const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
                const uint32_t km = __float_as_uint(q0.w), kind = km & 3u;
                mat = km >> 2;
                point = vadd(o, vscale(d, best.t));
                if (kind == KIND_TRIANGLE) {
                    normal = vnormalize(vcross(v3(q1), v3(q2)));      // the record holds the edges
                    back = vdot(normal, d) > 0.0f;
                } else if (kind == KIND_SPHERE) {
                    normal = vnormalize(vsub(point, v3(q0)));
                } else {
                    normal = v3(q2);
                }
                const float4 m0 = sc.mats[2 * (size_t)mat], m1 = sc.mats[2 * (size_t)mat + 1];
                diffuse_color = v3(m0); specular_color = v3(m1);
                shininess = m0.w;
                shadow_origin = vadd(point, vscale(normal, 1e-6f));                             // raytracer.rs:133
                const bool reflective = specular_color.x > 0.0f || specular_color.y > 0.0f || specular_color.z > 0.0f;
                goes_on = reflective && depth + 1u < f.maxdepth;                                // raytracer.rs:187
            }

            // ---- the rays that depend on this hit only, side by side when they fit
            const unsigned long long alive_mask = __builtin_amdgcn_ballot_w64(alive);
            const uint32_t n_alive = (uint32_t)__popcll(alive_mask);
            // Worth it when the jobs take fewer rounds of 64 than the one-after-the-other way takes passes.  The jobs of a hit: a shadow
            // ray per light, and the closest-hit ray of the next level unless no path of the wavefront goes on (the last level).
            const uint32_t jobs_per_hit = sc.n_lights + (__builtin_amdgcn_ballot_w64(goes_on) != 0 ? 1u : 0u);
            const uint32_t n_jobs = n_alive * jobs_per_hit, n_rounds = (n_jobs + 63u) >> 6;
            const bool side_by_side = n_alive != 0u && n_rounds <= RT_SIDE_ROUNDS_MAX && n_rounds < jobs_per_hit && sc.n_lights <= 31u;      // the same in every lane
            uint32_t lit = 0;            // side by side: bit li set = light li is not occluded
            have_next = false;
            if (side_by_side) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(alive_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)alive_mask, 0u));
                // lane numbers of the live lanes, packed from lane 0 upwards (the others from lane 63 downwards)
                const uint32_t table = (uint32_t)__builtin_amdgcn_ds_permute((int)((alive ? rank : 63u - (lane - rank)) << 2), (int)lane);
                const uint32_t first_job = rank * jobs_per_hit;                                 // (meaningful in live lanes)
                for (uint32_t round = 0; round < n_rounds; round++) {
                    MARK("round_setup");
                    if (COUNT && first_active_lane()) cnt.wave_rounds++;
                    const uint32_t job = round * 64u + lane, hit_of_job = job / jobs_per_hit, kind = job - hit_of_job * jobs_per_hit;
                    const bool mine = job < n_jobs;
                    const uint32_t owner = pull(table, mine ? hit_of_job : 0u);
                    const V3 at = v3(pull(point.x, owner), pull(point.y, owner), pull(point.z, owner));
                    const V3 nrm = v3(pull(normal.x, owner), pull(normal.y, owner), pull(normal.z, owner));
                    const V3 dir_in = v3(pull(d.x, owner), pull(d.y, owner), pull(d.z, owner));
                    const uint32_t flags = pull((back ? 1u : 0u) | (goes_on ? 2u : 0u), owner);
                    const float shin = pull(shininess, owner);
                    const bool shadow_job = kind < sc.n_lights;
                    bool need = mine, point_light = false;
                    float light_dist = 0.0f;
                    V3 ray_dir;
                    if (shadow_job) {
                        const float4 l0 = sc.lights[2 * (size_t)(mine ? kind : 0u)];
                        point_light = l0.w == 0.0f;
                        if (point_light) {
                            V3 to_light = vsub(v3(l0), at);
                            ray_dir = vnormalize(to_light);
                            light_dist = vnorm(to_light);
                        } else {
                            ray_dir = v3(l0);
                        }
                        const float n_dot_l = fmaxf(vdot(nrm, ray_dir), 0.0f);
                        if (sc.finite_shading && n_dot_l == 0.0f && shin != 1.0f) need = false; // contributes nothing either way (see the light loop below)
                    } else {
                        need = need && (flags & 2u) != 0u;
                        const float k2 = 2.0f * vdot(dir_in, nrm);                              // raytracer.rs:188-190
                        ray_dir = vsub(dir_in, vscale(nrm, k2));
                    }
                    Best found = {0.0f, NO_PRIM, 0u};
                    bool occluded = false;
                    if (need) {
                        occluded = trace_job<COUNT, TRI>(sc, vadd(at, vscale(nrm, 1e-6f)), ray_dir, shadow_job, (flags & 1u) != 0u, point_light, light_dist, stack, &found, &cnt);
                    }
                    MARK("round_collect");
                    const uint32_t word = shadow_job ? ((need && !occluded) ? 1u : 0u) : found.prim;
                    // every live lane collects the answers of its jobs that were in this round
                    for (uint32_t li = 0; li < sc.n_lights; li++) {
                        const uint32_t answer = pull(word, (first_job + li) & 63u);
                        if (((first_job + li) >> 6) == round) lit |= answer << li;
                    }
                    if (jobs_per_hit > sc.n_lights) {                                           // (the same in every lane)
                        const uint32_t next_prim = pull(word, (first_job + sc.n_lights) & 63u);
                        const float next_t = pull(found.t, (first_job + sc.n_lights) & 63u);
                        if (((first_job + sc.n_lights) >> 6) == round) { next.prim = next_prim; next.t = next_t; }
                    }
                }
                have_next = true;
            }

            if (alive) {
                V3 acc = v3(0.0f, 0.0f, 0.0f);
                for (uint32_t li = 0; li < sc.n_lights; li++) {                                  // raytracer.rs:126-179
                    MARK("light");
                    const float4 l0 = uniform_load(sc.lights, 2 * li), l1 = uniform_load(sc.lights, 2 * li + 1);
                    const bool point_light = l0.w == 0.0f;
                    V3 light_dir; float light_dist = 0.0f;
                    if (point_light) {
                        V3 to_light = vsub(v3(l0), point);
                        light_dir = vnormalize(to_light);
                        light_dist = vnorm(to_light);
                    } else {
                        light_dir = v3(l0);
                    }
                    if (COUNT) cnt.shadow++;
                    float n_dot_l = fmaxf(vdot(normal, light_dir), 0.0f);
                    // A light at or below the horizon of the surface adds (diffuse*0 + specular*0) (*) colour = +0
                    // whether or not it is occluded (unless shininess == 1, where the factor ignores n.l), so
                    // its shadow ray need not be traced.  Only valid while every factor is finite.
                    if (sc.finite_shading && n_dot_l == 0.0f && shininess != 1.0f) continue;
                    if (side_by_side) {
                        if (!((lit >> li) & 1u)) continue;
                    } else {
                        if (COUNT && first_active_lane()) cnt.wave_shadow++;
                        if (in_shadow<COUNT, TRI>(sc, shadow_origin, light_dir, back, point_light, light_dist, stack, &cnt)) continue;
                    }
                    MARK("shade");
                    if (COUNT && first_active_lane()) cnt.wave_shade++;
                    V3 diffuse = vscale(diffuse_color, n_dot_l);
                    V3 view_dir = vneg(d);
                    V3 half_vector = vnormalize(vadd(light_dir, view_dir));
                    float n_dot_h = fmaxf(vdot(normal, half_vector), 0.0f);
                    float factor;
                    if (shininess == 1.0f) factor = n_dot_h;
                    else if (shininess == 0.0f) factor = n_dot_l > 0.0f ? n_dot_h : 0.0f;
                    else { MARK("shade_pow"); factor = n_dot_l > 0.0f ? specular_power(n_dot_h, shininess, pow_tables) : 0.0f; }
                    MARK("shade_sum");
                    V3 specular = vscale(specular_color, factor);
                    acc = vadd(acc, vcmul(vadd(diffuse, specular), v3(l1)));
                }
                MARK("level_end");
                V3 local = vadd(acc, ambient);                                                   // raytracer.rs:181
                if (!goes_on) { cur = local; alive = false; }
                else {
                    const float4 record = make_float4(local.x, local.y, local.z, __uint_as_float(mat));
                    if (depth < lds_levels) fold_lds[depth * 64u] = record;
                    else fold[(size_t)(depth - lds_levels) * npix + pix] = record;
                    levels = depth + 1u;
                    float k2 = 2.0f * vdot(d, normal);                                           // raytracer.rs:188-190
                    d = vsub(d, vscale(normal, k2));
                    o = shadow_origin;
                    if (COUNT) cnt.secondary++;
                }
            }
        }
        if (live) {
            // Backward fold: final_k = local_k + specular_k (*) final_{k+1}                       raytracer.rs:194-195
            MARK("fold");
            while (levels > 0) {
                levels--;
                const float4 rec = levels < lds_levels ? fold_lds[levels * 64u] : fold[(size_t)(levels - lds_levels) * npix + pix];
                const float4 m1 = sc.mats[2 * (size_t)__float_as_uint(rec.w) + 1];
                cur = vadd(v3(rec), vcmul(v3(m1), cur));
            }
            MARK("quantise");
            out[pix] = quantise(cur);
        }
    }
    MARK("epilogue");
    if (f.region_cost && lane == 0) atomicMax(&f.region_cost[region], (uint32_t)(__builtin_amdgcn_s_memrealtime() - started));
    if (COUNT) {
        unsigned long long v[RT_N_COUNTERS] = {cnt.nodes, cnt.prims, cnt.shadow, cnt.secondary, cnt.wave_node_iters, cnt.wave_prim_iters,
                                               cnt.wave_closest, cnt.wave_hit, cnt.wave_shadow, cnt.wave_shade, cnt.wave_shadow_node_iters,
                                               cnt.wave_shadow_prim_iters, cnt.wave_rounds, cnt.wave_round_node_iters, cnt.wave_round_prim_iters};
        for (int k = 0; k < RT_N_COUNTERS; k++) {
            unsigned long long s = v[k];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0 && s) atomicAdd(&counters[k], s);
        }
    }
}

// Regions by decreasing measured cost, ties by index: wavefront i counts the regions that come before region i (higher
// cost, or the same cost and a lower index) and files i at that place.  n wavefronts with n / 64 loads each from an array
// the L2 holds: a few microseconds, where one workgroup sorting in LDS took 22 (which every frame waits for).
// The costs move on to `last` (what rt_tile_costs_export hands out).  Nothing may write `cost` while the wavefronts
// read it, so the array that the next launch measures into is another one, `zeroed`, which this kernel clears:
// no separate clearing launch per frame.
__global__ __launch_bounds__(64) void rank_regions_kernel(const uint32_t* __restrict__ cost, uint32_t n, uint32_t* __restrict__ order,
                                                          uint32_t* __restrict__ last, uint32_t* __restrict__ zeroed) {
    const uint32_t i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const uint32_t mine = cost[i];
    uint32_t before = 0;
    for (uint32_t j = lane; j < n; j += 64u) {
        const uint32_t c = cost[j];
        before += (c > mine) | ((c == mine) & (j < i));
    }
    for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off, 64);
    if (lane == 0) { order[before] = i; last[i] = mine; zeroed[i] = 0u; }
}

// A guess at the costs of the tile regions for the first launch of a frame geometry, which has no measured ones yet:
// where the tree's shapes are on screen, seen directly and in the specular planes of the always-tested list (the slow
// wavefronts of a frame are those whose paths keep bouncing between shapes and their mirror images; raster order starts
// the last of them late and ends the launch with a nearly empty GPU).  Every `step`-th tree shape is projected through
// the frame's camera, as it is (counting twice: its paths start on it) and mirrored in each such plane (once), into the
// region it lands in; counts are collected per workgroup in LDS and added to seen[] (all zero before).  A region's cost is then the sum
// over the five by five regions around it: what the slowest wavefront of a region costs varies a lot between
// neighbours, and a smooth guess orders them better than a sharp one (measured on the 4K dragon frame from two
// cameras).  Only the order of the work depends on this; the launch then measures the real costs for the ones after it.
__device__ __forceinline__ void count_on_screen(const DevFrame& f, uint32_t regions_y, V3 p, uint32_t weight, uint32_t* hist) {
    const V3 q = vsub(p, v3(f.pos));
    const float depth = vdot(q, v3(f.cv));
    if (!(depth > 0.0f)) return;
    const float px = (vdot(q, v3(f.right)) / depth) / f.pw * f.w2 + f.w2, py = f.h2 - (vdot(q, v3(f.up)) / depth) / f.ph * f.h2;
    if (!(px >= 0.0f && px < (float)f.width && py >= 0.0f && py < (float)f.height)) return;
    // row of the frame -> row of this shard's compact buffer (the shards interleave tile by tile, so a row that belongs
    // to another shard stands for its neighbours that belong to this one)
    const uint32_t y = (uint32_t)py, tile = y / f.tile_rows, local = (tile / f.nshards) * f.tile_rows + (y - tile * f.tile_rows);
    const uint32_t rx = (uint32_t)px >> (3u + f.region_log2), ry = local >> (3u + f.region_log2);
    if (rx < f.regions_x && ry < regions_y) atomicAdd(&hist[ry * f.regions_x + rx], weight);
}

__global__ __launch_bounds__(256) void count_shapes_on_screen_kernel(DevScene sc, DevFrame f, uint32_t n_shapes, uint32_t step, uint32_t* __restrict__ seen) {
    __shared__ uint32_t hist[MAX_SORTED_REGIONS];
    for (uint32_t i = threadIdx.x; i < f.n_regions; i += blockDim.x) hist[i] = 0u;
    __syncthreads();
    const uint32_t regions_y = f.n_regions / f.regions_x;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s * step < n_shapes; s += (uint64_t)gridDim.x * blockDim.x) {
        const V3 p = v3(sc.tree[3 * (size_t)(s * step)]);       // first vertex of a triangle, centre of a sphere
        count_on_screen(f, regions_y, p, 2u, hist);
        if (f.maxdepth > 1u)
            for (uint32_t i = 0; i < sc.n_list; i++) {
                const float4 l0 = uniform_load(sc.list, 3 * i), l2 = uniform_load(sc.list, 3 * i + 2);
                const uint32_t km = __float_as_uint(l0.w);
                if ((km & 3u) != KIND_PLANE) continue;
                const float4 m1 = sc.mats[2 * (size_t)(km >> 2) + 1];
                if (!(m1.x > 0.0f || m1.y > 0.0f || m1.z > 0.0f)) continue;
                const V3 n = v3(l2);
                count_on_screen(f, regions_y, vsub(p, vscale(n, 2.0f * vdot(vsub(p, v3(l0)), n))), 1u, hist);
            }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < f.n_regions; i += blockDim.x)
        if (hist[i]) atomicAdd(&seen[i], hist[i]);
}

__global__ __launch_bounds__(256) void smooth_region_costs_kernel(const uint32_t* __restrict__ seen, uint32_t regions_x, uint32_t n_regions, uint32_t* __restrict__ cost) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_regions) return;
    const int regions_y = (int)(n_regions / regions_x), x = (int)(i % regions_x), y = (int)(i / regions_x);
    unsigned long long sum = 0;
    for (int dy = -2; dy <= 2; dy++)
        for (int dx = -2; dx <= 2; dx++)
            if (x + dx >= 0 && x + dx < (int)regions_x && y + dy >= 0 && y + dy < regions_y) sum += seen[(uint32_t)(y + dy) * regions_x + (uint32_t)(x + dx)];
    cost[i] = (uint32_t)(sum < 0xFFFFFFFFull ? sum : 0xFFFFFFFFull);
}

// Scatters the compact per-shard row buffers back into frame order on the gathering GPU.
__global__ void assemble_kernel(const uint32_t* __restrict__ gathered, uint32_t stride_rows, uint32_t nshards, uint32_t tile_rows,
                                uint32_t width, uint32_t height, uint32_t* __restrict__ frame) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)width * height;
    if (i >= n) return;
    uint32_t y = (uint32_t)(i / width), x = (uint32_t)(i - (size_t)y * width);
    uint32_t tile = y / tile_rows, within = y - tile * tile_rows;
    uint32_t shard = tile % nshards, local_tile = tile / nshards;
    size_t src = ((size_t)shard * stride_rows + (size_t)local_tile * tile_rows + within) * width + x;
    frame[i] = gathered[src];
}

// ------------------------------------------------------------------ launchers (called from rt_api.cpp)
extern "C" hipError_t rtk_launch_render(const DevScene* sc, const DevFrame* f, uint32_t* d_out, float4* d_fold,
                                        unsigned long long* d_counters, int stack_depth, RenderPlan plan, int count, hipStream_t stream) {
    uint32_t tiles_x = (f->width + 7u) >> 3, tiles_y = (f->local_rows + 7u) >> 3;
    uint64_t blocks = (uint64_t)tiles_x * tiles_y;
    if (blocks == 0) return hipSuccess;
    if (f->region_log2 > 0) {             // the caller planned regions (see region_log2_for): whole regions for all 8 XCD slots
        const uint64_t side = 1ull << f->region_log2;
        blocks = (((uint64_t)f->n_regions + 7u) / 8u) * 8u * side * side;
    }
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    dim3 grid((uint32_t)blocks), block(64);
    // One stack entry per tree level plus the sentinel.
    const uint32_t entries = rtk_stack_entries(stack_depth);
    if (entries > 65) return hipErrorInvalidValue;
    size_t lds_bytes = (size_t)entries * 64 * sizeof(uint32_t);
    const uint32_t levels = f->maxdepth > 1 ? f->maxdepth - 1u : 0u;
    if (plan.lds_levels > levels) return hipErrorInvalidValue;
    lds_bytes += (size_t)plan.lds_levels * 64 * sizeof(float4) + RT_POW_LDS_BYTES;
    if (lds_bytes > 64u * 1024u || (plan.lds_levels < levels && !d_fold)) return hipErrorInvalidValue;
#ifdef RT_DIAG_LDS_PAD
    { static const char* pad = getenv("RT_DIAG_LDS_PAD"); if (pad) lds_bytes = std::max(lds_bytes, (size_t)atoi(pad)); }   // occupancy experiments (scripts/ab_occ.sh)
#endif
    const bool tri = sc->tree_kinds == (1u << KIND_TRIANGLE);
    auto kernel = count ? (tri ? render_kernel<true, true> : render_kernel<true, false>) : (tri ? render_kernel<false, true> : render_kernel<false, false>);
    hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, *sc, *f, d_out, d_fold, d_counters, entries, plan.lds_levels);
    return hipGetLastError();
}

extern "C" hipError_t rtk_launch_sort_regions(const uint32_t* d_cost, uint32_t n, uint32_t* d_order, uint32_t* d_last, uint32_t* d_zeroed, hipStream_t stream) {
    if (n == 0 || n > MAX_SORTED_REGIONS || d_cost == d_zeroed) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rank_regions_kernel, dim3(n), dim3(64), 0, stream, d_cost, n, d_order, d_last, d_zeroed);
    return hipGetLastError();
}

extern "C" hipError_t rtk_launch_predict_regions(const DevScene* sc, const DevFrame* f, uint32_t n_tree_shapes, uint32_t* d_seen, uint32_t* d_cost, hipStream_t stream) {
    if (f->n_regions == 0) return hipSuccess;
    if (f->n_regions > MAX_SORTED_REGIONS || f->regions_x == 0 || f->n_regions % f->regions_x != 0 || f->nshards == 0 || f->tile_rows == 0) return hipErrorInvalidValue;
    if (n_tree_shapes != 0) {                                 // (no tree: nothing to go by, the costs come out equal)
        const uint32_t max_samples = 1u << 18, step = (n_tree_shapes + max_samples - 1u) / max_samples, samples = (n_tree_shapes + step - 1u) / step;
        const uint32_t blocks = std::min((samples + 255u) / 256u, 256u);
        hipLaunchKernelGGL(count_shapes_on_screen_kernel, dim3(blocks), dim3(256), 0, stream, *sc, *f, n_tree_shapes, step, d_seen);
    }
    hipLaunchKernelGGL(smooth_region_costs_kernel, dim3((f->n_regions + 255u) / 256u), dim3(256), 0, stream, d_seen, f->regions_x, f->n_regions, d_cost);
    return hipGetLastError();
}

extern "C" hipError_t rtk_launch_assemble(const uint32_t* d_gathered, uint32_t stride_rows, uint32_t nshards, uint32_t tile_rows,
                                          uint32_t width, uint32_t height, uint32_t* d_frame, hipStream_t stream) {
    size_t n = (size_t)width * height;
    if (n == 0) return hipSuccess;
    uint32_t blocks = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(assemble_kernel, dim3(blocks), dim3(256), 0, stream, d_gathered, stride_rows, nshards, tile_rows, width, height, d_frame);
    return hipGetLastError();
}
