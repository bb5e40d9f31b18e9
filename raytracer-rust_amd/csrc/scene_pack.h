// scene_pack.h -- turns the flattened scene of the C ABI into the device blob of layout.h.
#ifndef RT_SCENE_PACK_H
#define RT_SCENE_PACK_H

#include <stdint.h>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rt_amd.h"
#include "layout.h"
#include "no_init_vector.h"

namespace rtp {

// Scenes with at most this many bounded shapes skip the tree: every shape is tested for
// every ray, straight from the list (the jalon3/5/6 scenes have at most 6).
constexpr uint32_t SMALL_SCENE = 8;

// Longest root-to-leaf path the builder may produce: the deepest traversal stack a kernel offers
// (64 entries) less the sentinel entry at its bottom.
constexpr uint32_t TREE_DEPTH_LIMIT = 63;



// The same scene split into the parts of a blob without a tree, for the GPU tree builder: shape
// records of the tree shapes in input order (triangles, then spheres; triangles still with their three
// vertices, which the builders take the boxes from: whoever writes them into a blob applies
// rtl::to_blob_form), the always-tested list in file order and already in blob form, the material
// table, the lights, and the padding rule of the node boxes.
// use_tree is false for scenes so small that every shape goes to the list (then pack_scene is the way).
// The whole job of the host path: split, tree build (bvh_build.cpp), blob.  Padding between the sections of the blob is zero.
bool pack_scene(const rt_scene& s, rtu::NoInitVector<uint8_t>* blob, rt_bvh_info* info, std::string* err);

struct SceneParts {
    rtu::NoInitVector<rtl::ShapeRec> tree;
    std::vector<rtl::ShapeRec> list;
    std::vector<rtl::MaterialRec> mats;
    std::vector<rtl::LightRec> lights;
    uint32_t finite_shading = 1;
    uint32_t tree_kinds = 0;      // bit k set: `tree` holds shapes of kind k
    float pad_abs = 0.0f;
    bool use_tree = false;
};
bool split_scene(const rt_scene& s, SceneParts* out, std::string* err);
// Which kinds of shape the n records hold, one bit per kind (layout.h).
uint32_t kinds_of(const rtl::ShapeRec* records, size_t n);

// Checks a blob header copied back from the device before the kernels are pointed at it.
bool validate_header(const rtl::BlobHeader& h, size_t bytes, std::string* err);

// Checks a whole blob in host memory that did not come from pack_scene in this process (a cache
// file): header, every reference, tree shape.
bool validate_blob(const uint8_t* blob, size_t bytes, std::string* err);

}  // namespace rtp
#endif
