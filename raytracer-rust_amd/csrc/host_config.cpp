// host_config.cpp -- C++ twin of the reference's scene parser, scene model and comparator,
// behind the C ABI of include/rt_host.h.
//
//   ParsedConfigState / Config      src/raytracer/config/config_builder.rs:14-24, 111-461
//   Shape / Light / Camera          src/raytracer/config/shape.rs:5-32, light.rs:3-6, camera.rs:3-8
//   Image::compare                  src/imgcomparator/mod.rs:54-89
//
// The parser keeps the reference's behaviour line for line in *effect*: stateful current
// material and vertex pool, split on single spaces, silent skip of one-token lines, the same
// validation rules and the same error strings.  It stores shapes directly in the
// structure-of-arrays form rt_scene wants instead of the reference's enum-of-structs.
#include "../../include/rt_host.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "errors.h"
#include "layout.h"
#include "scene_pack.h"

struct rth_config {
    uint32_t width = 800, height = 600;                    // config_builder.rs:131-132
    std::string output_file = "output.png";                // :133
    float cam_pos[3] = {0, 0, 0}, cam_look[3] = {0, 0, 1}, cam_up[3] = {0, 1, 0}, cam_fov = 60.0f;   // :134-139
    float ambient[3] = {0, 0, 0};                          // :140
    uint32_t maxdepth = 1, maxverts = 0;                   // :141-142
    // scene_objects, one array set per kind, plus each shape's position in the file
    std::vector<float> tri_v0, tri_v1, tri_v2, tri_mat, sph_center, sph_radius, sph_mat, pl_point, pl_normal, pl_mat;
    std::vector<uint32_t> tri_order, sph_order, pl_order;
    uint32_t nshapes = 0;
    std::vector<float> lights;                             // 7 floats each
    // parser state (config_builder.rs:111-116)
    float cur_diffuse[3] = {0, 0, 0}, cur_specular[3] = {0, 0, 0}, cur_shininess = 0.0f;
    std::vector<float> vertices;
};

namespace {

// ---- number grammar of Rust's str::parse ------------------------------------------------
bool rust_f32(const char* b, const char* e, float* out, const char** err) {
    if (b == e) { *err = "cannot parse float from empty string"; return false; }
    const char* p = b;
    const bool negative = *p == '-';
    if (*p == '+' || *p == '-') p++;
    auto ieq = [&](const char* lit) {
        size_t n = std::strlen(lit);
        if ((size_t)(e - p) != n) return false;
        for (size_t i = 0; i < n; i++) if ((p[i] | 0x20) != lit[i]) return false;
        return true;
    };
    if (ieq("inf") || ieq("infinity")) { *out = negative ? -HUGE_VALF : HUGE_VALF; return true; }
    if (ieq("nan")) { *out = std::nanf(""); return true; }
    bool ok = false;
    const char* q = p; int mant = 0;
    while (q < e && (unsigned)(*q - '0') < 10u) { q++; mant++; }
    if (q < e && *q == '.') { q++; while (q < e && (unsigned)(*q - '0') < 10u) { q++; mant++; } }
    if (mant > 0) {
        if (q < e && (*q == 'e' || *q == 'E')) {
            q++;
            if (q < e && (*q == '+' || *q == '-')) q++;
            int ex = 0;
            while (q < e && (unsigned)(*q - '0') < 10u) { q++; ex++; }
            ok = ex > 0 && q == e;
        } else ok = q == e;
    }
    if (!ok) { *err = "invalid float literal"; return false; }
    // Correctly rounded, like Rust's parser.  from_chars reports overflow/underflow without a
    // value; strtof then gives the infinity / subnormal / zero that Rust returns.
    float v;
    auto r = std::from_chars(p, e, v);
    if (r.ec != std::errc() || r.ptr != e) { std::string tmp(p, e); v = std::strtof(tmp.c_str(), nullptr); }
    *out = negative ? -v : v;
    return true;
}

bool rust_uint(const char* b, const char* e, uint64_t limit, uint64_t* out, const char** err) {
    if (b == e) { *err = "cannot parse integer from empty string"; return false; }
    if (*b == '+') b++;
    if (b == e) { *err = "invalid digit found in string"; return false; }
    uint64_t v = 0;
    for (const char* p = b; p < e; p++) {
        if ((unsigned)(*p - '0') >= 10u) { *err = "invalid digit found in string"; return false; }
        uint64_t d = (uint64_t)(*p - '0');
        if (v > (limit - d) / 10) { *err = "number too large to fit in target type"; return false; }
        v = v * 10 + d;
    }
    *out = v;
    return true;
}

struct Span { const char* b; const char* e; };

inline bool is_space(char ch) { return ch == ' ' || (unsigned)(ch - 9) < 5u; }

Span trimmed(const char* b, const char* e) {
    while (b < e && is_space(*b)) b++;
    while (e > b && is_space(e[-1])) e--;
    return {b, e};
}

// str::split(' '): n separators give n + 1 pieces, empty pieces included.  Stores at most `cap`
// pieces and returns how many there are.
size_t split_spaces(Span s, Span* out, size_t cap) {
    size_t n = 0;
    const char* p = s.b;
    for (const char* q = s.b;; q++) {
        if (q == s.e || *q == ' ') { if (n < cap) out[n] = {p, q}; n++; p = q + 1; if (q == s.e) break; }
    }
    return n;
}

bool floats(Span param, size_t want, const char* shape_err, float* out, const char** err) {
    Span t[12];
    if (split_spaces(param, t, 12) != want) { *err = shape_err; return false; }
    for (size_t i = 0; i < want; i++) if (!rust_f32(t[i].b, t[i].e, &out[i], err)) return false;
    return true;
}

bool rgb_in_range(const float* c) {                                   // config_builder.rs:376-381
    for (int k = 0; k < 3; k++) if (!(c[k] >= 0.0f && c[k] <= 1.0f)) return false;
    return true;
}
const char* const RGB_ERR = "RGB values must be between 0.0 and 1.0";

void normalize3(float* v) {    // nalgebra normalize as the reference's arithmetic behaves: reciprocal norm, three multiplies
    float n = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    float inv = 1.0f / n;
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
}

// ---- the parser -------------------------------------------------------------------------
// The reference reads line by line through one mutable state (config_builder.rs:127-248).  Here
// the text (a memory-mapped file) is cut into chunks at line ends and read in three steps, the
// first and last in parallel over the chunks:
//   1. scan: every line becomes a record {key, numbers}; whatever can be decided from the line
//      alone (formats, number grammar, fixed ranges) is decided here;
//   2. one pass over the chunk summaries gives every chunk the state it starts from (vertex and
//      shape counts so far, current material, ambient, maxverts);
//   3. apply: vertices first, then shapes, written straight into their final arrays; the checks
//      that need the state (maxverts, vertex index bounds, diffuse + ambient) happen here.
// The reported error is the one of the earliest failing line, as in a sequential read.
enum Key : uint8_t {
    K_SIZE, K_OUTPUT, K_CAMERA, K_AMBIENT, K_SPHERE, K_TRI, K_PLANE, K_POINT, K_DIRECTIONAL, K_DIFFUSE, K_SPECULAR, K_SHININESS,
    K_MAXDEPTH, K_MAXVERTS, K_VERTEX, K_ERROR
};
constexpr int N_FLOATS[] = {0, 0, 10, 3, 4, 0, 6, 6, 6, 3, 3, 1, 0, 0, 3, 0};
constexpr int N_UINTS[] = {2, 0, 0, 0, 0, 3, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0};

struct Material { float diffuse[3] = {0, 0, 0}, specular[3] = {0, 0, 0}, shininess = 0.0f; };

struct Chunk {
    Span text{nullptr, nullptr};
    std::vector<uint8_t> keys;
    std::vector<float> fl;
    std::vector<uint32_t> ui;
    std::vector<std::string> outputs;                 // parameters of `output` lines, in order
    size_t n_vertex = 0, n_tri = 0, n_sph = 0, n_pl = 0, n_light = 0;
    std::string error; size_t error_rec = SIZE_MAX;   // earliest failing record of this chunk
    // state the chunk starts from (step 2)
    size_t v0 = 0, tri0 = 0, sph0 = 0, pl0 = 0, light0 = 0;
    Material mat0; float ambient0[3] = {0, 0, 0}; uint32_t maxverts0 = 0;

    void fail_at(size_t rec, const std::string& m) { if (rec < error_rec) { error_rec = rec; error = m; } }
};

Key key_of(Span k, bool* known) {
    static const struct { const char* name; Key key; } table[] = {
        {"tri", K_TRI}, {"vertex", K_VERTEX}, {"sphere", K_SPHERE}, {"plane", K_PLANE}, {"diffuse", K_DIFFUSE}, {"specular", K_SPECULAR},
        {"shininess", K_SHININESS}, {"point", K_POINT}, {"directional", K_DIRECTIONAL}, {"size", K_SIZE}, {"output", K_OUTPUT},
        {"camera", K_CAMERA}, {"ambient", K_AMBIENT}, {"maxdepth", K_MAXDEPTH}, {"maxverts", K_MAXVERTS}};
    size_t n = (size_t)(k.e - k.b);
    for (const auto& t : table)
        if (std::strlen(t.name) == n && std::memcmp(t.name, k.b, n) == 0) { *known = true; return t.key; }
    *known = false;
    return K_ERROR;
}

// One line -> at most one record.  False when the line is in error (recorded in the chunk).
bool scan_line(Chunk& c, const char* lb, const char* le) {                      // config_builder.rs:152-248
    Span t = trimmed(lb, le);
    if (t.b == t.e || *t.b == '#') return true;
    Span parts[2];
    if (split_spaces({lb, le}, parts, 2) < 2) return true;
    Span key_sp = trimmed(parts[0].b, parts[0].e);
    Span param = trimmed(lb + (key_sp.e - key_sp.b), le);
    const size_t rec = c.keys.size();
    const char* err = nullptr;
    auto bad = [&](const std::string& m) { c.keys.push_back(K_ERROR); c.fail_at(rec, m); return false; };
    bool known;
    Key key = key_of(key_sp, &known);
    if (!known) return bad("Unknown configuration key: " + std::string(key_sp.b, key_sp.e));
    float f[10]; uint64_t u[3]; Span d[4];
    switch (key) {
    case K_SIZE:
        if (split_spaces(param, d, 4) != 2) return bad("Invalid size format");
        if (!rust_uint(d[0].b, d[0].e, 0xFFFFFFFFull, &u[0], &err) || !rust_uint(d[1].b, d[1].e, 0xFFFFFFFFull, &u[1], &err)) return bad(err);
        if (u[0] == 0 || u[1] == 0) return bad("Width and height must be greater than zero");
        break;
    case K_OUTPUT:
        if (param.b == param.e) return bad("Output file name cannot be empty");
        c.outputs.emplace_back(param.b, param.e);
        break;
    case K_CAMERA:
        if (!floats(param, 10, "Invalid camera format", f, &err)) return bad(err);
        if (f[9] < 1.0f || f[9] > 179.0f) return bad("Field of view (fov) must be between 1 and 179 degrees");
        break;
    case K_AMBIENT:
        if (!floats(param, 3, "Invalid ambient light format", f, &err)) return bad(err);
        if (!rgb_in_range(f)) return bad(RGB_ERR);
        break;
    case K_SPHERE:
        if (!floats(param, 4, "Invalid sphere format", f, &err)) return bad(err);
        if (f[3] <= 0.0f) return bad("Sphere radius must be greater than zero");
        c.n_sph++;
        break;
    case K_TRI:
        if (split_spaces(param, d, 4) != 3) return bad("Invalid triangle format");
        for (int k = 0; k < 3; k++) {
            if (!rust_uint(d[k].b, d[k].e, ~0ull, &u[k], &err)) return bad(err);
            if (u[k] > 0xFFFFFFFFull) u[k] = 0xFFFFFFFFull;   // the pool holds at most maxverts < 2^32 vertices: out of bounds either way
        }
        c.n_tri++;
        break;
    case K_PLANE:
        if (!floats(param, 6, "Invalid plane format", f, &err)) return bad(err);
        normalize3(f + 3);                                            // config_builder.rs:445-450
        c.n_pl++;
        break;
    case K_POINT: case K_DIRECTIONAL:
        if (!floats(param, 6, key == K_DIRECTIONAL ? "Invalid directional light format" : "Invalid point light format", f, &err)) return bad(err);
        if (!rgb_in_range(f + 3)) return bad(RGB_ERR);
        if (key == K_DIRECTIONAL) normalize3(f);                      // config_builder.rs:303-306
        c.n_light++;
        break;
    case K_DIFFUSE:
        if (!floats(param, 3, "Invalid Vector3 format", f, &err)) return bad(err);
        if (!rgb_in_range(f)) return bad(RGB_ERR);
        break;
    case K_SPECULAR:
        if (!floats(param, 3, "Invalid Vector3 format", f, &err)) return bad(err);
        if (f[0] < 0.0f || f[1] < 0.0f || f[2] < 0.0f) return bad("Specular color components must be non-negative");
        break;
    case K_SHININESS:
        if (!rust_f32(param.b, param.e, f, &err)) return bad(err);
        if (f[0] < 0.0f) return bad("Shininess must be non-negative");
        break;
    case K_MAXDEPTH: case K_MAXVERTS:
        if (!rust_uint(param.b, param.e, 0xFFFFFFFFull, &u[0], &err)) return bad(err);
        break;
    case K_VERTEX:
        if (!floats(param, 3, "Invalid Vector3 format", f, &err)) return bad(err);
        c.n_vertex++;
        break;
    case K_ERROR: break;
    }
    c.keys.push_back(key);
    c.fl.insert(c.fl.end(), f, f + N_FLOATS[key]);
    for (int k = 0; k < N_UINTS[key]; k++) c.ui.push_back((uint32_t)u[k]);
    return true;
}

void scan_chunk(Chunk& c) {
    const char* p = c.text.b;
    while (p < c.text.e) {
        const char* nl = (const char*)std::memchr(p, '\n', (size_t)(c.text.e - p));
        const char* le = nl ? nl : c.text.e;
        const char* next = nl ? nl + 1 : c.text.e;
        if (le > p && le[-1] == '\r') le--;                                     // BufRead::lines strips "\r\n" too
        if (!scan_line(c, p, le)) return;                                       // nothing after the first error matters
        p = next;
    }
}

// Walks the records of a chunk in order.  `on(key, f, u, rec)` returns false to stop.
template <class F>
void walk(const Chunk& c, F on) {
    const float* f = c.fl.data(); const uint32_t* u = c.ui.data();
    for (size_t r = 0; r < c.keys.size(); r++) {
        Key k = (Key)c.keys[r];
        if (k == K_ERROR || !on(k, f, u, r)) return;
        f += N_FLOATS[k]; u += N_UINTS[k];
    }
}

rth_config* parse_text(const char* text, size_t bytes) {
    // Chunks of at least 1 MiB, cut after a line end.
    const size_t target = std::max<size_t>(1u << 20, bytes / 1024 + 1);
    std::vector<Chunk> chunks;
    for (size_t pos = 0; pos < bytes;) {
        size_t end = std::min(bytes, pos + target);
        if (end < bytes) {
            const char* nl = (const char*)std::memchr(text + end, '\n', bytes - end);
            end = nl ? (size_t)(nl - text) + 1 : bytes;
        }
        chunks.emplace_back();
        chunks.back().text = {text + pos, text + end};
        pos = end;
    }
    const long nchunks = (long)chunks.size();
#pragma omp parallel for schedule(dynamic, 1) if (nchunks > 1)
    for (long i = 0; i < nchunks; i++) scan_chunk(chunks[i]);

    // Step 2: the state every chunk starts from, and the scalar settings (last one wins).
    rth_config* c = new rth_config();
    {
        size_t v = 0, tri = 0, sph = 0, pl = 0, light = 0;
        Material mat; float ambient[3] = {0, 0, 0}; uint32_t maxverts = 0;
        for (Chunk& ch : chunks) {
            ch.v0 = v; ch.tri0 = tri; ch.sph0 = sph; ch.pl0 = pl; ch.light0 = light;
            ch.mat0 = mat; std::memcpy(ch.ambient0, ambient, 12); ch.maxverts0 = maxverts;
            v += ch.n_vertex; tri += ch.n_tri; sph += ch.n_sph; pl += ch.n_pl; light += ch.n_light;
            size_t out_i = 0;
            walk(ch, [&](Key k, const float* f, const uint32_t* u, size_t) {
                switch (k) {   // only the rare keys do anything here
                case K_SIZE: c->width = u[0]; c->height = u[1]; break;
                case K_OUTPUT: c->output_file = ch.outputs[out_i++]; break;
                case K_CAMERA: std::memcpy(c->cam_pos, f, 12); std::memcpy(c->cam_look, f + 3, 12); std::memcpy(c->cam_up, f + 6, 12); c->cam_fov = f[9]; break;
                case K_AMBIENT: std::memcpy(ambient, f, 12); break;
                case K_DIFFUSE: std::memcpy(mat.diffuse, f, 12); break;
                case K_SPECULAR: std::memcpy(mat.specular, f, 12); break;
                case K_SHININESS: mat.shininess = f[0]; break;
                case K_MAXDEPTH: c->maxdepth = u[0]; break;
                case K_MAXVERTS: maxverts = u[0]; break;
                default: break;
                }
                return true;
            });
        }
        std::memcpy(c->ambient, ambient, 12); c->maxverts = maxverts;
        std::memcpy(c->cur_diffuse, mat.diffuse, 12); std::memcpy(c->cur_specular, mat.specular, 12); c->cur_shininess = mat.shininess;
        if ((uint64_t)tri + sph + pl > 0xFFFFFFFFull) { rte::set_error("more than 2^32 - 1 shapes"); delete c; return nullptr; }
        c->vertices.resize(3 * v);
        c->tri_v0.resize(3 * tri); c->tri_v1.resize(3 * tri); c->tri_v2.resize(3 * tri); c->tri_mat.resize(7 * tri); c->tri_order.resize(tri);
        c->sph_center.resize(3 * sph); c->sph_radius.resize(sph); c->sph_mat.resize(7 * sph); c->sph_order.resize(sph);
        c->pl_point.resize(3 * pl); c->pl_normal.resize(3 * pl); c->pl_mat.resize(7 * pl); c->pl_order.resize(pl);
        c->lights.resize(7 * light);
        c->nshapes = (uint32_t)(tri + sph + pl);
    }

    // Step 3a: the vertex pool (config_builder.rs:236-243) and the diffuse + ambient rule (:386-392).
#pragma omp parallel for schedule(dynamic, 1) if (nchunks > 1)
    for (long i = 0; i < nchunks; i++) {
        Chunk& ch = chunks[i];
        size_t v = ch.v0; uint32_t maxverts = ch.maxverts0; float ambient[3];
        std::memcpy(ambient, ch.ambient0, 12);
        walk(ch, [&](Key k, const float* f, const uint32_t* u, size_t rec) {
            if (k == K_VERTEX) {
                if (v >= (size_t)maxverts) { ch.fail_at(rec, "Exceeded maximum number of vertices (maxverts)"); return false; }
                std::memcpy(&c->vertices[3 * v], f, 12); v++;
            } else if (k == K_MAXVERTS) maxverts = u[0];
            else if (k == K_AMBIENT) std::memcpy(ambient, f, 12);
            else if (k == K_DIFFUSE) {
                if ((f[0] + ambient[0]) > 1.0f || (f[1] + ambient[1]) > 1.0f || (f[2] + ambient[2]) > 1.0f) {
                    ch.fail_at(rec, "Sum of diffuse color and ambient light components must not exceed 1.0"); return false;
                }
            }
            return true;
        });
    }
    // Step 3b: shapes and lights into their final arrays.
#pragma omp parallel for schedule(dynamic, 1) if (nchunks > 1)
    for (long i = 0; i < nchunks; i++) {
        Chunk& ch = chunks[i];
        size_t v = ch.v0, tri = ch.tri0, sph = ch.sph0, pl = ch.pl0, light = ch.light0;
        Material mat = ch.mat0;
        auto put_material = [&](float* dst) { std::memcpy(dst, mat.diffuse, 12); std::memcpy(dst + 3, mat.specular, 12); dst[6] = mat.shininess; };
        walk(ch, [&](Key k, const float* f, const uint32_t* u, size_t rec) {
            if (rec >= ch.error_rec) return false;
            const uint32_t order = (uint32_t)(tri + sph + pl);
            switch (k) {
            case K_VERTEX: v++; break;
            case K_TRI:
                for (int j = 0; j < 3; j++) if ((size_t)u[j] >= v) { ch.fail_at(rec, "Triangle vertex index out of bounds"); return false; }
                std::memcpy(&c->tri_v0[3 * tri], &c->vertices[3 * (size_t)u[0]], 12);
                std::memcpy(&c->tri_v1[3 * tri], &c->vertices[3 * (size_t)u[1]], 12);
                std::memcpy(&c->tri_v2[3 * tri], &c->vertices[3 * (size_t)u[2]], 12);
                put_material(&c->tri_mat[7 * tri]); c->tri_order[tri] = order; tri++;
                break;
            case K_SPHERE:
                std::memcpy(&c->sph_center[3 * sph], f, 12); c->sph_radius[sph] = f[3];
                put_material(&c->sph_mat[7 * sph]); c->sph_order[sph] = order; sph++;
                break;
            case K_PLANE:
                std::memcpy(&c->pl_point[3 * pl], f, 12); std::memcpy(&c->pl_normal[3 * pl], f + 3, 12);
                put_material(&c->pl_mat[7 * pl]); c->pl_order[pl] = order; pl++;
                break;
            case K_POINT: case K_DIRECTIONAL:
                c->lights[7 * light] = k == K_DIRECTIONAL ? 1.0f : 0.0f;
                std::memcpy(&c->lights[7 * light + 1], f, 24); light++;
                break;
            case K_DIFFUSE: std::memcpy(mat.diffuse, f, 12); break;
            case K_SPECULAR: std::memcpy(mat.specular, f, 12); break;
            case K_SHININESS: mat.shininess = f[0]; break;
            default: break;
            }
            return true;
        });
    }
    for (const Chunk& ch : chunks)
        if (ch.error_rec != SIZE_MAX) { rte::set_error(ch.error); delete c; return nullptr; }
    return c;
}

int fail(const char* m) { rte::set_error(m); return 1; }

}  // namespace

extern "C" {

rth_config* rth_load_config_file(const char* path) {
    if (!path) { rte::set_error("path is NULL"); return nullptr; }
    int fd = open(path, O_RDONLY);
    if (fd < 0) { rte::set_error(std::string(std::strerror(errno)) + " (os error " + std::to_string(errno) + "): " + path); return nullptr; }
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); rte::set_error(std::string("not a regular file: ") + path); return nullptr; }
    if (st.st_size == 0) { close(fd); return parse_text("", 0); }
    void* map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) { rte::set_error(std::string("mmap failed: ") + path); return nullptr; }
    (void)madvise(map, (size_t)st.st_size, MADV_SEQUENTIAL);
    rth_config* c = parse_text((const char*)map, (size_t)st.st_size);
    munmap(map, (size_t)st.st_size);
    return c;
}

rth_config* rth_parse_config_text(const char* text) {
    if (!text) { rte::set_error("text is NULL"); return nullptr; }
    return parse_text(text, std::strlen(text));
}

void rth_config_free(rth_config* c) { delete c; }

int rth_config_get_info(const rth_config* c, rth_config_info* out) {
    if (!c || !out) return fail("rth_config_get_info: NULL argument");
    out->width = c->width; out->height = c->height; out->maxdepth = c->maxdepth; out->maxverts = c->maxverts;
    out->ntri = (uint32_t)c->tri_order.size(); out->nsph = (uint32_t)c->sph_order.size(); out->npl = (uint32_t)c->pl_order.size();
    out->nlights = (uint32_t)(c->lights.size() / 7);
    std::memcpy(out->camera_position, c->cam_pos, 12); std::memcpy(out->camera_look_at, c->cam_look, 12);
    std::memcpy(out->camera_up, c->cam_up, 12); out->camera_fov = c->cam_fov;
    std::memcpy(out->ambient, c->ambient, 12);
    return 0;
}

const char* rth_config_output_file(const rth_config* c) { return c ? c->output_file.c_str() : ""; }

int rth_config_set_size(rth_config* c, uint32_t width, uint32_t height) {
    if (!c) return fail("rth_config_set_size: NULL config");
    if (width == 0 || height == 0) return fail("Width and height must be greater than zero");
    c->width = width; c->height = height;
    return 0;
}

int rth_config_scene(const rth_config* c, rt_scene* out) {
    if (!c || !out) return fail("rth_config_scene: NULL argument");
    std::memset(out, 0, sizeof(*out));
    out->ntri = (uint32_t)c->tri_order.size(); out->nsph = (uint32_t)c->sph_order.size(); out->npl = (uint32_t)c->pl_order.size();
    out->tri_v0 = c->tri_v0.data(); out->tri_v1 = c->tri_v1.data(); out->tri_v2 = c->tri_v2.data(); out->tri_mat = c->tri_mat.data();
    out->sph_center = c->sph_center.data(); out->sph_radius = c->sph_radius.data(); out->sph_mat = c->sph_mat.data();
    out->pl_point = c->pl_point.data(); out->pl_normal = c->pl_normal.data(); out->pl_mat = c->pl_mat.data();
    out->tri_order = c->tri_order.data(); out->sph_order = c->sph_order.data(); out->pl_order = c->pl_order.data();
    out->lights = c->lights.data(); out->nlights = (uint32_t)(c->lights.size() / 7);
    return 0;
}

int rth_config_frame(const rth_config* c, rt_frame* out) {
    if (!c || !out) return fail("rth_config_frame: NULL argument");
    return rt_frame_from_camera(c->cam_pos, c->cam_look, c->cam_up, c->cam_fov, c->width, c->height, c->maxdepth, c->ambient, out);
}

int rth_image_compare(const uint32_t* a, uint32_t wa, uint32_t ha, const uint32_t* b, uint32_t wb, uint32_t hb, uint32_t* diff,
                      uint64_t* count) {
    if (ha != hb || wa != wb) return fail("Images have different dimensions");            // mod.rs:55-57
    if (!a || !b) return fail("rth_image_compare: NULL image");
    uint64_t total = 0; size_t n = (size_t)wa * ha;
    for (size_t i = 0; i < n; i++) {
        uint32_t d = 0;
        if (a[i] != b[i]) {                                                               // mod.rs:63-78
            int dr = std::abs((int)((a[i] >> 16) & 0xFF) - (int)((b[i] >> 16) & 0xFF));
            int dg = std::abs((int)((a[i] >> 8) & 0xFF) - (int)((b[i] >> 8) & 0xFF));
            int db = std::abs((int)(a[i] & 0xFF) - (int)(b[i] & 0xFF));
            if (!(dr <= 1 && dg <= 1 && db <= 1)) d = ((uint32_t)dr << 16) | ((uint32_t)dg << 8) | (uint32_t)db;
        }
        if (d != 0) total++;
        if (diff) diff[i] = d;
    }
    if (count) *count = total;
    return 0;
}

int rth_pack_scene(const rt_scene* scene, void** blob, size_t* bytes, rt_bvh_info* info) {
    if (!scene || !blob || !bytes) return fail("rth_pack_scene: NULL argument");
    rtu::NoInitVector<uint8_t> buf; std::string err; rt_bvh_info local{};
    if (!rtp::pack_scene(*scene, &buf, &local, &err)) { rte::set_error(err); return 1; }
    void* p = std::malloc(buf.size());
    if (!p) return fail("out of memory");
    std::memcpy(p, buf.data(), buf.size());
    *blob = p; *bytes = buf.size();
    if (info) *info = local;
    return 0;
}

void rth_free(void* p) { std::free(p); }

int rth_validate_blob(const void* blob, size_t bytes) {
    std::string err;
    if (!rtp::validate_blob((const uint8_t*)blob, bytes, &err)) { rte::set_error(err); return 1; }
    return 0;
}

// ---- binary scene cache ---------------------------------------------------------------------
// One file holds what RayTracer::new (raytracer.rs:29-38) derives from a scene file, so that a
// second start neither parses the text nor builds the tree: a 4096-byte header with the Config
// fields the render loop reads (config_builder.rs:14-24 minus the shapes) and the identity of
// the source file, then the device blob (layout.h) as it is uploaded.

}  // extern "C"

struct rth_scene_cache { void* map = nullptr; size_t bytes = 0; };

namespace {

struct CacheHeader {
    char magic[8];
    uint32_t version, header_bytes;
    uint64_t blob_bytes, blob_hash;
    uint64_t source_bytes; int64_t source_mtime_ns;
    uint32_t width, height, maxdepth, maxverts;
    float cam_pos[3], cam_look[3], cam_up[3], cam_fov, ambient[3];
    uint32_t ntri, nsph, npl, nlights;
    char output_file[1024];
    uint32_t reserved;             // zero (spelt out so that the struct has no padding: every byte of it is written and hashed)
    uint64_t header_hash;          // of the whole header block with this field zero: damage to the Config fields is noticed too
};
static_assert(offsetof(CacheHeader, header_hash) == 1160 && sizeof(CacheHeader) == 1168, "cache header layout (no padding)");
constexpr size_t CACHE_HEADER_BYTES = 4096;
// The layout of this header (high half) and of the blob behind it (low half): a cache written before either changed is
// "written by another version", which callers treat like a stale one (from_scene_file rebuilds it from the scene file).
constexpr uint32_t CACHE_VERSION = (3u << 16) | rtl::BLOB_VERSION;
const char CACHE_MAGIC[8] = {'R', 'T', 'S', 'C', 'A', 'C', 'H', 'E'};
static_assert(sizeof(CacheHeader) <= CACHE_HEADER_BYTES, "cache header must fit its block");

// Four interleaved multiply-rotate lanes over 8-byte words (memory-speed; guards against damaged files, not against intent).
uint64_t hash64(const void* data, size_t bytes) {
    const uint8_t* p = (const uint8_t*)data;
    uint64_t h[4] = {0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull};
    size_t i = 0;
    for (; i + 32 <= bytes; i += 32)
        for (int k = 0; k < 4; k++) {
            uint64_t w; std::memcpy(&w, p + i + 8 * k, 8);
            h[k] = (h[k] ^ w) * 0x100000001B3ull; h[k] = (h[k] << 29) | (h[k] >> 35);
        }
    uint64_t tail = bytes;
    for (; i < bytes; i++) tail = (tail ^ p[i]) * 0x100000001B3ull;
    return ((h[0] ^ (h[1] << 1)) * 0x9E3779B97F4A7C15ull) ^ ((h[2] ^ (h[3] << 1)) * 0xC2B2AE3D27D4EB4Full) ^ tail;
}

bool source_identity(const char* path, uint64_t* bytes, int64_t* mtime_ns) {
    struct stat st;
    if (!path || stat(path, &st) != 0) return false;
    *bytes = (uint64_t)st.st_size;
    *mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
    return true;
}

bool write_all(int fd, const void* data, size_t bytes) {
    const char* p = (const char*)data;
    while (bytes) {
        ssize_t n = write(fd, p, std::min<size_t>(bytes, 1u << 30));
        if (n < 0) { if (errno == EINTR) continue; return false; }
        p += n; bytes -= (size_t)n;
    }
    return true;
}

}  // namespace

extern "C" {

int rth_scene_cache_write(const char* path, const rth_config* c, const void* blob, size_t bytes, const char* source_path) {
    if (!path || !c || !blob) return fail("rth_scene_cache_write: NULL argument");
    if (bytes < sizeof(rtl::BlobHeader)) return fail("rth_scene_cache_write: not a scene blob");
    if (c->output_file.size() >= sizeof(CacheHeader::output_file)) return fail("rth_scene_cache_write: output file name too long");
    std::vector<char> block(CACHE_HEADER_BYTES, 0);
    CacheHeader h;
    std::memset(&h, 0, sizeof(h));
    std::memcpy(h.magic, CACHE_MAGIC, 8);
    h.version = CACHE_VERSION; h.header_bytes = (uint32_t)CACHE_HEADER_BYTES;
    h.blob_bytes = bytes; h.blob_hash = hash64(blob, bytes);
    if (!source_identity(source_path, &h.source_bytes, &h.source_mtime_ns)) { h.source_bytes = 0; h.source_mtime_ns = 0; }
    h.width = c->width; h.height = c->height; h.maxdepth = c->maxdepth; h.maxverts = c->maxverts;
    std::memcpy(h.cam_pos, c->cam_pos, 12); std::memcpy(h.cam_look, c->cam_look, 12); std::memcpy(h.cam_up, c->cam_up, 12);
    h.cam_fov = c->cam_fov; std::memcpy(h.ambient, c->ambient, 12);
    h.ntri = (uint32_t)c->tri_order.size(); h.nsph = (uint32_t)c->sph_order.size(); h.npl = (uint32_t)c->pl_order.size();
    h.nlights = (uint32_t)(c->lights.size() / 7);
    std::memcpy(h.output_file, c->output_file.data(), c->output_file.size());
    std::memcpy(block.data(), &h, sizeof(h));
    h.header_hash = hash64(block.data(), block.size());
    std::memcpy(block.data(), &h, sizeof(h));
    // Written under a temporary name and renamed, so that a reader never sees half a file.
    std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid());
    int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { rte::set_error(std::string(std::strerror(errno)) + ": " + tmp); return 1; }
    bool ok = write_all(fd, block.data(), block.size()) && write_all(fd, blob, bytes);
    ok = (close(fd) == 0) && ok;
    if (ok && rename(tmp.c_str(), path) != 0) ok = false;
    if (!ok) { rte::set_error(std::string("cannot write scene cache: ") + std::strerror(errno)); unlink(tmp.c_str()); return 1; }
    return 0;
}

rth_scene_cache* rth_scene_cache_open(const char* path) {
    if (!path) { rte::set_error("path is NULL"); return nullptr; }
    int fd = open(path, O_RDONLY);
    if (fd < 0) { rte::set_error(std::string(std::strerror(errno)) + " (os error " + std::to_string(errno) + "): " + path); return nullptr; }
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || (size_t)st.st_size < CACHE_HEADER_BYTES) {
        close(fd); rte::set_error("scene cache: not a cache file"); return nullptr;
    }
    void* map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) { rte::set_error("scene cache: mmap failed"); return nullptr; }
    CacheHeader h; std::memcpy(&h, map, sizeof(h));
    const char* why = nullptr;
    if (std::memcmp(h.magic, CACHE_MAGIC, 8) != 0) why = "scene cache: not a cache file";
    else if (h.version != CACHE_VERSION || h.header_bytes != CACHE_HEADER_BYTES) why = "scene cache: written by another version";
    else if (h.blob_bytes != (uint64_t)st.st_size - CACHE_HEADER_BYTES) why = "scene cache: truncated";
    else if (std::memchr(h.output_file, 0, sizeof(h.output_file)) == nullptr) why = "scene cache: damaged header";
    else if (hash64((const char*)map + CACHE_HEADER_BYTES, (size_t)h.blob_bytes) != h.blob_hash) why = "scene cache: checksum mismatch";
    else {
        std::vector<char> block((const char*)map, (const char*)map + CACHE_HEADER_BYTES);
        std::memset(block.data() + offsetof(CacheHeader, header_hash), 0, sizeof(h.header_hash));      // hashed with this field zero
        if (hash64(block.data(), block.size()) != h.header_hash) why = "scene cache: damaged header";
    }
    if (why) { munmap(map, (size_t)st.st_size); rte::set_error(why); return nullptr; }
    rth_scene_cache* out = new rth_scene_cache();
    out->map = map; out->bytes = (size_t)st.st_size;
    return out;
}

const void* rth_scene_cache_blob(const rth_scene_cache* h, size_t* bytes) {
    if (!h || !bytes) { rte::set_error("rth_scene_cache_blob: NULL argument"); return nullptr; }
    *bytes = h->bytes - CACHE_HEADER_BYTES;
    return (const char*)h->map + CACHE_HEADER_BYTES;
}

rth_config* rth_scene_cache_config(const rth_scene_cache* h) {
    if (!h) { rte::set_error("rth_scene_cache_config: NULL cache"); return nullptr; }
    CacheHeader ch; std::memcpy(&ch, h->map, sizeof(ch));
    rth_config* c = new rth_config();
    c->width = ch.width; c->height = ch.height; c->maxdepth = ch.maxdepth; c->maxverts = ch.maxverts;
    std::memcpy(c->cam_pos, ch.cam_pos, 12); std::memcpy(c->cam_look, ch.cam_look, 12); std::memcpy(c->cam_up, ch.cam_up, 12);
    c->cam_fov = ch.cam_fov; std::memcpy(c->ambient, ch.ambient, 12);
    c->output_file = ch.output_file;
    return c;
}

int rth_scene_cache_is_fresh(const rth_scene_cache* h, const char* source_path) {
    if (!h) return 0;
    CacheHeader ch; std::memcpy(&ch, h->map, sizeof(ch));
    uint64_t bytes; int64_t mtime;
    if (!source_identity(source_path, &bytes, &mtime)) return 0;
    return (ch.source_bytes == bytes && ch.source_mtime_ns == mtime && (bytes != 0 || mtime != 0)) ? 1 : 0;
}

void rth_scene_cache_close(rth_scene_cache* h) {
    if (!h) return;
    if (h->map) munmap(h->map, h->bytes);
    delete h;
}

}  // extern "C"
