// layout.h -- device-side scene layout shared by the host builder and the HIP kernels.
//
// Everything the kernels read lives in ONE contiguous device blob (so that a scene can be
// replicated to another GPU with a single broadcast):
//
//   [BlobHeader 256 B][node pairs][tree shapes][list shapes][materials][lights]
//
// every section 256-byte aligned.
//
//   node pair  32 B  two 16-byte child records side by side: {lo.x | hi.x << 16, lo.y | hi.y << 16,
//                    lo.z | hi.z << 16, ref}, 16-bit coordinates on a grid that spans the box of all tree
//                    shapes (header: grid_origin, grid_scale).  A coordinate q stands for the plane
//                    origin + units(q) * scale with units(q) = q below 32768 and 2 q - 32768 from there
//                    on: 98302 units per axis, planes one unit apart in the first third of the range and
//                    two units apart beyond.  That is what the bit pattern gives for free: one byte permute
//                    turns a coordinate into the float with the bytes {0, q.lo, q.hi, 0x47}, which is
//                    32768 + units(q) (the top bit of q is the lowest exponent bit), with no integer-to-float
//                    conversion; and because low and high plane of an axis share a word, the same permute,
//                    with a per-ray selector, also sorts them into the plane the ray enters through and the
//                    one it leaves through.  Boxes are rounded outwards onto the grid, so they only grow (by
//                    less than two units: 0.002 % of the scene extent).  One 16-byte load per child.
//   shape      48 B  three float4.  q0.w carries kind | material << 2, q1.w the shape's
//                    position in the scene file (tie break), q2.w is spare.
//                      triangle: q0.xyz = v0, q1.xyz = v1 - v0, q2.xyz = v2 - v0   (the two edges every
//                                test starts from, shape.rs:149-150, subtracted once in binary32 when the
//                                blob is written: the same bits the per-test subtraction would give)
//                      sphere:   q0.xyz = center, q1.x = radius
//                      plane:    q0.xyz = point,  q2.xyz = unit normal
//   material   32 B  {diffuse.rgb, shininess} {specular.rgb, 0}
//   light      32 B  {vec.xyz, kind} {colour.rgb, 0}     kind 0 = point, 1 = directional
#ifndef RT_LAYOUT_H
#define RT_LAYOUT_H

#include <math.h>
#include <stdint.h>

namespace rtl {

constexpr uint32_t KIND_SPHERE = 0, KIND_TRIANGLE = 1, KIND_PLANE = 2;

// Child reference.  Inner (bit 31 clear): the node pair's byte offset in the node array, 32 x its index, so that a
// traversal step adds it to the base address as it is (the builders work with plain indices; inner_ref converts).
// Leaf: bit 31 set, bits 30..3 = first shape of the leaf in the tree-shape array, bits 2..0 = count - 1.
constexpr uint32_t REF_LEAF = 0x80000000u;
constexpr uint32_t REF_EMPTY = 0xFFFFFFFFu;   // never produced for a real leaf (would need first = 2^28 - 1)
constexpr uint32_t MAX_LEAF = 8;
constexpr uint32_t MAX_TREE_SHAPES = 1u << 26;   // fewer node pairs than 2^26: their byte offsets leave bit 31 of a reference free; and the byte offset of a shape (48 x its index) fits 32 bits

inline uint32_t make_leaf_ref(uint32_t first, uint32_t count) { return REF_LEAF | (first << 3) | (count - 1); }
// A builder's reference (pair index or leaf) -> the blob's.
constexpr uint32_t inner_ref(uint32_t builder_ref) { return (builder_ref & REF_LEAF) ? builder_ref : builder_ref << 5; }

// Builder-side record (full precision), quantised by scene_pack.cpp into QuantPair.
struct ChildBox { float lo[3]; uint32_t ref; float hi[3]; uint32_t count; };
struct NodePair { ChildBox c[2]; };
static_assert(sizeof(NodePair) == 64, "builder node pair is 64 bytes");

struct QuantChild { uint16_t q[3][2]; uint32_t ref; };   // q[axis][0] = low plane, q[axis][1] = high plane
struct QuantPair { QuantChild c[2]; };
static_assert(sizeof(QuantChild) == 16 && sizeof(QuantPair) == 32, "device node pair must be 32 bytes");
constexpr float GRID_UNITS = 98302.0f;        // units(65535)
#if defined(__HIPCC__)
#define RT_LAYOUT_FN __host__ __device__ inline
#else
#define RT_LAYOUT_FN inline
#endif
RT_LAYOUT_FN float grid_units(uint32_t q) { return q < 32768u ? (float)q : (float)(2u * q - 32768u); }
// Coordinate of the last plane at or below u units / the first plane at or above it (u already clamped to [0, GRID_UNITS]).
RT_LAYOUT_FN uint32_t grid_code_below(float u) { return u < 32768.0f ? (uint32_t)floorf(u) : 32768u + (uint32_t)floorf((u - 32768.0f) * 0.5f); }
RT_LAYOUT_FN uint32_t grid_code_above(float u) { return u <= 32768.0f ? (uint32_t)ceilf(u) : 32768u + (uint32_t)ceilf((u - 32768.0f) * 0.5f); }

struct ShapeRec { float q0[3]; uint32_t kind_mat; float q1[3]; uint32_t order; float q2[3]; uint32_t spare; };
static_assert(sizeof(ShapeRec) == 48, "shape record must be 48 bytes");

// A record as the builders handle it (triangles with their three vertices) -> as the blob stores it.
RT_LAYOUT_FN void to_blob_form(ShapeRec* r) {
    if ((r->kind_mat & 3u) != KIND_TRIANGLE) return;
    for (int k = 0; k < 3; k++) { r->q1[k] = r->q1[k] - r->q0[k]; r->q2[k] = r->q2[k] - r->q0[k]; }
}

struct MaterialRec { float diffuse[3]; float shininess; float specular[3]; float pad; };
struct LightRec { float v[3]; float kind; float color[3]; float pad; };

constexpr uint32_t BLOB_MAGIC = 0x52544231u;   // "RTB1"
constexpr uint32_t BLOB_VERSION = 7;

struct BlobHeader {
    uint32_t magic, version;
    uint64_t total_bytes;
    uint64_t off_nodes, off_tree_shapes, off_list_shapes, off_materials, off_lights;
    uint32_t n_pairs, n_tree_shapes, n_list_shapes, n_materials, n_lights;
    uint32_t root_ref;
    uint32_t max_depth, max_leaf, n_leaves;
    float sah_cost;
    uint32_t finite_shading;   // 1 when every material and light value is finite
    float grid_origin[3], grid_scale[3];   // node box coordinate q on axis k = grid_origin[k] + q * grid_scale[k]
    uint32_t tree_kinds;       // bit k set: the tree holds shapes of kind k (a tree of triangles only gets the kernel built for that)
    uint32_t pad[64 - 32];
};
static_assert(sizeof(BlobHeader) == 256, "blob header must be 256 bytes");

}  // namespace rtl
#endif
