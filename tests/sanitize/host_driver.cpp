// host_driver.cpp -- drives the host-side C ABI (include/rt_host.h) through its good and its bad paths.
// Built by tests/test_host_sanitizers.py with AddressSanitizer and UndefinedBehaviorSanitizer together with
// the host sources of the library; any finding aborts the program, which fails the test.
//   host_driver <directory with the golden scenes> <scratch directory>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../include/rt_host.h"
#include "../../raytracer-rust_amd/csrc/bvh_build.h"
#include "../../raytracer-rust_amd/csrc/errors.h"

static int checks = 0;
#define REQUIRE(cond)                                                                            \
    do {                                                                                         \
        checks++;                                                                                \
        if (!(cond)) { std::fprintf(stderr, "%s:%d: %s failed (%s)\n", __FILE__, __LINE__, #cond, rte::last_error()); std::exit(2); } \
    } while (0)

static std::string slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    return std::string(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
}
static void spill(const std::string& path, const std::string& bytes) {
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    f.write(bytes.data(), (std::streamsize)bytes.size());
}

static uint32_t lcg(uint32_t* s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

// Parses, flattens, packs (tree build included), validates, then damages the blob in many places: validation may
// say yes or no, it must not read outside the buffer.
static void scene_round(const std::string& path, const std::string& scratch) {
    rth_config* c = rth_load_config_file(path.c_str());
    REQUIRE(c != nullptr);
    rth_config_info info;
    REQUIRE(rth_config_get_info(c, &info) == 0);
    REQUIRE(rth_config_output_file(c) != nullptr);
    rt_scene sc; rt_frame fr;
    REQUIRE(rth_config_scene(c, &sc) == 0);
    REQUIRE(rth_config_frame(c, &fr) == 0);
    REQUIRE(sc.ntri == info.ntri && sc.nsph == info.nsph && sc.npl == info.npl && fr.width == info.width);
    void* blob = nullptr; size_t bytes = 0; rt_bvh_info bi;
    REQUIRE(rth_pack_scene(&sc, &blob, &bytes, &bi) == 0);
    REQUIRE(rth_validate_blob(blob, bytes) == 0);
    std::vector<uint8_t> copy((const uint8_t*)blob, (const uint8_t*)blob + bytes);
    uint32_t seed = 12345u + (uint32_t)bytes;
    for (int round = 0; round < 200; round++) {
        std::vector<uint8_t> bad = copy;
        const size_t at = round < 60 ? (size_t)round * 4 : lcg(&seed) % bytes;            // the header first, then anywhere
        bad[at] ^= (uint8_t)(1u << (lcg(&seed) % 8));
        if (round % 3 == 0) bad[at] = 0xFF;
        (void)rth_validate_blob(bad.data(), bad.size());
        checks++;
    }
    for (size_t cut : {(size_t)0, (size_t)17, (size_t)255, (size_t)256, bytes / 2, bytes - 1}) { (void)rth_validate_blob(copy.data(), cut); checks++; }
    // section offsets close to 2^64: offset + count * size wraps round to a small number, the check must not
    for (size_t field = 16; field <= 48; field += 8)
        for (uint64_t off : std::initializer_list<uint64_t>{0xFFFFFFFFFFFFFF00ull, 0xFFFFFFFFFFFFFFF0ull, 0x8000000000000000ull, (uint64_t)bytes, (uint64_t)bytes + 16}) {
            std::vector<uint8_t> bad = copy;
            std::memcpy(bad.data() + field, &off, 8);
            const bool in_place = off == (uint64_t)bytes;              // an empty section may end exactly at the end
            const int rc = rth_validate_blob(bad.data(), bad.size());
            REQUIRE(rc != 0 || in_place);
            checks++;
        }
    // scene cache: write, map, read back, and a damaged / truncated file
    const std::string cache = scratch + "/scene.rtcache";
    REQUIRE(rth_scene_cache_write(cache.c_str(), c, blob, bytes, path.c_str()) == 0);
    rth_scene_cache* h = rth_scene_cache_open(cache.c_str());
    REQUIRE(h != nullptr);
    size_t got = 0; const void* p = rth_scene_cache_blob(h, &got);
    REQUIRE(p != nullptr && got == bytes && std::memcmp(p, blob, bytes) == 0);
    REQUIRE(rth_scene_cache_is_fresh(h, path.c_str()) == 1);
    rth_config* c2 = rth_scene_cache_config(h);
    REQUIRE(c2 != nullptr);
    rt_frame fr2; REQUIRE(rth_config_frame(c2, &fr2) == 0 && std::memcmp(&fr, &fr2, sizeof(fr)) == 0);
    rth_config_free(c2);
    rth_scene_cache_close(h);
    std::string file = slurp(cache);
    for (size_t cut : {(size_t)0, (size_t)100, (size_t)4096, file.size() - 7}) {
        spill(cache, file.substr(0, cut));
        REQUIRE(rth_scene_cache_open(cache.c_str()) == nullptr);
    }
    for (size_t at : {(size_t)3, (size_t)20, (size_t)200, (size_t)5000, file.size() - 1}) {
        std::string bad = file; bad[at % bad.size()] ^= 0x40;
        spill(cache, bad);
        rth_scene_cache* hb = rth_scene_cache_open(cache.c_str());
        if (hb) rth_scene_cache_close(hb);                                               // a flipped bit in unused header padding is accepted
        checks++;
    }
    rth_free(blob);
    rth_config_free(c);
}

// The grammar's error paths: every prefix of a good file cut inside a line, and lines with garbage spliced in.
static void parser_round(const std::string& path) {
    const std::string text = slurp(path);
    REQUIRE(!text.empty());
    uint32_t seed = 99u;
    const size_t limit = text.size() < 6000 ? text.size() : 6000;
    for (size_t cut = 0; cut < limit; cut += 1 + lcg(&seed) % 7) {
        const std::string part = text.substr(0, cut);
        rth_config* c = rth_parse_config_text(part.c_str());
        if (c) rth_config_free(c);
        checks++;
    }
    const char* junk[] = {"sphere 1 2", "tri 0 1 999999", "vertex a b c", "size -3 5", "maxverts 1e99", "camera 1 2 3", "tri 0 0", "point 1 2 3 4 5",
                          "bogus 1", "shininess", "size 4294967297 3", "\xff\xfe\x00garbage", "vertex 1 2 3 4", "maxdepth -1", "tri -1 0 1", "plane 0 0 0 0 0"};
    for (const char* j : junk) {
        for (size_t pos : {(size_t)0, text.size() / 3, text.size()}) {
            size_t at = text.find('\n', pos); at = at == std::string::npos ? text.size() : at + 1;
            const std::string mixed = text.substr(0, at) + j + "\n" + text.substr(at);
            rth_config* c = rth_parse_config_text(mixed.c_str());
            if (c) rth_config_free(c);
            checks++;
        }
    }
    REQUIRE(rth_load_config_file((path + ".does-not-exist").c_str()) == nullptr);
}

static void png_round(const std::string& scratch) {
    const uint32_t w = 67, h = 41;
    std::vector<uint32_t> px((size_t)w * h);
    uint32_t seed = 7u;
    for (uint32_t& p : px) p = lcg(&seed) | 0xFF000000u;
    const std::string path = scratch + "/a.png";
    REQUIRE(rth_save_png(path.c_str(), px.data(), w, h) == 0);
    uint32_t* back = nullptr; uint32_t bw = 0, bh = 0;
    REQUIRE(rth_load_png(path.c_str(), &back, &bw, &bh) == 0 && bw == w && bh == h);
    for (size_t i = 0; i < px.size(); i++) REQUIRE((back[i] & 0xFFFFFFu) == (px[i] & 0xFFFFFFu));
    uint64_t count = 1; std::vector<uint32_t> diff(px.size());
    REQUIRE(rth_image_compare(px.data(), w, h, back, bw, bh, diff.data(), &count) == 0 && count == 0);
    REQUIRE(rth_image_compare(px.data(), w, h, back, bw, bh - 1, nullptr, &count) != 0);
    rth_free(back);
    std::vector<uint8_t> rgb((size_t)w * h * 3, 0x5A);
    REQUIRE(rth_save_png_rgb((scratch + "/b.png").c_str(), rgb.data(), w, h) == 0);
    const std::string file = slurp(path);
    for (size_t cut = 0; cut < file.size(); cut += 1 + file.size() / 97) {                 // truncated files
        spill(path, file.substr(0, cut));
        uint32_t* p = nullptr; uint32_t a = 0, b = 0;
        if (rth_load_png(path.c_str(), &p, &a, &b) == 0) rth_free(p);
        checks++;
    }
    for (int round = 0; round < 300; round++) {                                             // damaged bytes (CRCs catch most; the parser sees the rest)
        std::string bad = file;
        bad[lcg(&seed) % bad.size()] ^= (char)(1u << (lcg(&seed) % 8));
        spill(path, bad);
        uint32_t* p = nullptr; uint32_t a = 0, b = 0;
        if (rth_load_png(path.c_str(), &p, &a, &b) == 0) rth_free(p);
        checks++;
    }
}

// The tree builder under depth limits that bind: the top-down build has to switch to median splits, and the reinsertion
// sweeps behind it find subtrees with no place left within the limit and have to put them back.  The tree must cover every
// shape exactly once, hold every shape's box inside the boxes above it, and stay within the limit.
static uint32_t walk_tree(const rtb::BvhResult& t, uint32_t ref, const float* lo, const float* hi, const float* blo, const float* bhi,
                          std::vector<uint32_t>& seen, uint32_t depth) {
    if (ref & rtl::REF_LEAF) {
        const uint32_t first = (ref & ~rtl::REF_LEAF) >> 3, count = (ref & 7u) + 1u;
        for (uint32_t k = first; k < first + count; k++) {
            REQUIRE(k < t.order.size());
            const uint32_t shape = t.order[k];
            seen[shape]++;
            if (blo) for (int a = 0; a < 3; a++) REQUIRE(blo[a] <= lo[3 * shape + a] && bhi[a] >= hi[3 * shape + a]);
        }
        return depth;
    }
    REQUIRE(ref < t.pairs.size());
    uint32_t deepest = 0;
    for (int c = 0; c < 2; c++) {
        const rtl::ChildBox& cb = t.pairs[ref].c[c];
        if (blo) for (int a = 0; a < 3; a++) REQUIRE(blo[a] <= cb.lo[a] && bhi[a] >= cb.hi[a]);
        deepest = std::max(deepest, walk_tree(t, cb.ref, lo, hi, cb.lo, cb.hi, seen, depth + 1));
    }
    return deepest;
}

static void builder_round() {
    uint32_t seed = 99;
    for (uint32_t n : {40u, 700u, 5000u}) {
        std::vector<float> lo(3 * n), hi(3 * n);
        for (uint32_t i = 0; i < n; i++) {                                    // clusters of very different size along a curve
            const float scale = std::pow(1.003f, (float)(lcg(&seed) % 2000)), cx = scale * 3.0f, cy = (float)(lcg(&seed) % 100) * 0.01f * scale;
            for (int a = 0; a < 3; a++) {
                const float c = a == 0 ? cx : (a == 1 ? cy : 0.0f), r = scale * (0.01f + (float)(lcg(&seed) % 100) * 0.002f);
                lo[3 * i + a] = c - r; hi[3 * i + a] = c + r;
            }
        }
        uint32_t need = 1;
        while ((8u << need) < n) need++;                                      // a balanced tree of 8-shape leaves is this deep
        for (uint32_t limit : {need + 2u, need + 3u, need + 5u, 63u, 62u}) {
            if (limit < 8) continue;
            // last round: every third search of the reinsertion sweeps for a new place is refused, so subtrees that were
            // taken out are put back where they were, first and second child alike
            if (limit == 62u) setenv("RT_BVH_TEST_REFUSE_EVERY", "3", 1); else unsetenv("RT_BVH_TEST_REFUSE_EVERY");
            rtb::BvhResult tree; std::string err;
            REQUIRE(rtb::build_bvh(lo.data(), hi.data(), n, limit, 0.0f, &tree, &err));
            std::vector<uint32_t> seen(n, 0u);
            const uint32_t depth = walk_tree(tree, tree.root_ref, lo.data(), hi.data(), nullptr, nullptr, seen, 0);
            for (uint32_t i = 0; i < n; i++) REQUIRE(seen[i] == 1);
            REQUIRE(depth <= limit && tree.max_depth <= limit && depth == tree.max_depth);
            checks++;
        }
    }
    unsetenv("RT_BVH_TEST_REFUSE_EVERY");
}

int main(int argc, char** argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: host_driver <golden dir> <scratch dir>\n"); return 64; }
    const std::string golden = argv[1], scratch = argv[2];
    const char* scenes[] = {"test_file/jalon3/tp31.test", "test_file/jalon5/tp53.test", "test_file/jalon5/tp55.test", "test_file/jalon6/tp63.test",
                            "test_file/jalon6/tp64.test", "test.scene", "final_avec_bonus.scene"};
    for (const char* s : scenes) scene_round(golden + "/" + s, scratch);
    for (const char* s : {"test_file/jalon6/tp64.test", "test_file/jalon5/tp53.test", "test.scene"}) parser_round(golden + "/" + s);
    png_round(scratch);
    builder_round();
    std::printf("host_driver: %d checks, no findings\n", checks);
    return 0;
}
