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

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "errors.h"
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
bool rust_f32(const char* b, const char* e, float* out, std::string* err) {
    if (b == e) { *err = "cannot parse float from empty string"; return false; }
    const char* p = b;
    if (*p == '+' || *p == '-') p++;
    auto ieq = [&](const char* lit) {
        size_t n = std::strlen(lit);
        if ((size_t)(e - p) != n) return false;
        for (size_t i = 0; i < n; i++) if (std::tolower((unsigned char)p[i]) != lit[i]) return false;
        return true;
    };
    bool ok = ieq("inf") || ieq("infinity") || ieq("nan");
    if (!ok) {
        const char* q = p; int mant = 0;
        while (q < e && std::isdigit((unsigned char)*q)) { q++; mant++; }
        if (q < e && *q == '.') { q++; while (q < e && std::isdigit((unsigned char)*q)) { q++; mant++; } }
        if (mant > 0) {
            if (q < e && (*q == 'e' || *q == 'E')) {
                q++;
                if (q < e && (*q == '+' || *q == '-')) q++;
                int ex = 0;
                while (q < e && std::isdigit((unsigned char)*q)) { q++; ex++; }
                ok = ex > 0 && q == e;
            } else ok = q == e;
        }
    }
    if (!ok) { *err = "invalid float literal"; return false; }
    std::string tmp(b, e);
    *out = std::strtof(tmp.c_str(), nullptr);
    return true;
}

bool rust_uint(const char* b, const char* e, uint64_t limit, uint64_t* out, std::string* err) {
    if (b == e) { *err = "cannot parse integer from empty string"; return false; }
    if (*b == '+') b++;
    if (b == e) { *err = "invalid digit found in string"; return false; }
    uint64_t v = 0;
    for (const char* p = b; p < e; p++) {
        if (!std::isdigit((unsigned char)*p)) { *err = "invalid digit found in string"; return false; }
        uint64_t d = (uint64_t)(*p - '0');
        if (v > (limit - d) / 10) { *err = "number too large to fit in target type"; return false; }
        v = v * 10 + d;
    }
    *out = v;
    return true;
}

struct Span { const char* b; const char* e; };

Span trimmed(const char* b, const char* e) {
    while (b < e && std::isspace((unsigned char)*b)) b++;
    while (e > b && std::isspace((unsigned char)e[-1])) e--;
    return {b, e};
}

// str::split(' '): n separators give n + 1 pieces, empty pieces included.
std::vector<Span> split_spaces(Span s) {
    std::vector<Span> out;
    const char* p = s.b;
    for (const char* q = s.b;; q++) {
        if (q == s.e || *q == ' ') { out.push_back({p, q}); p = q + 1; if (q == s.e) break; }
    }
    return out;
}

bool floats(Span param, size_t want, const char* shape_err, float* out, std::string* err) {
    std::vector<Span> t = split_spaces(param);
    if (t.size() != want) { *err = shape_err; return false; }
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

void push_material(const rth_config& c, std::vector<float>* mat) {
    mat->insert(mat->end(), c.cur_diffuse, c.cur_diffuse + 3);
    mat->insert(mat->end(), c.cur_specular, c.cur_specular + 3);
    mat->push_back(c.cur_shininess);
}

bool parse_line(rth_config& c, const std::string& line, std::string* err) {     // config_builder.rs:152-248
    const char* lb = line.data(); const char* le = lb + line.size();
    Span t = trimmed(lb, le);
    if (t.b == t.e || *t.b == '#') return true;
    std::vector<Span> parts = split_spaces({lb, le});
    if (parts.size() < 2) return true;
    Span key_sp = trimmed(parts[0].b, parts[0].e);
    std::string key(key_sp.b, key_sp.e);
    Span param = trimmed(lb + key.size(), le);
    float f[10];
    if (key == "size") {
        std::vector<Span> d = split_spaces(param);
        if (d.size() != 2) { *err = "Invalid size format"; return false; }
        uint64_t w, h;
        if (!rust_uint(d[0].b, d[0].e, 0xFFFFFFFFull, &w, err) || !rust_uint(d[1].b, d[1].e, 0xFFFFFFFFull, &h, err)) return false;
        if (w == 0 || h == 0) { *err = "Width and height must be greater than zero"; return false; }
        c.width = (uint32_t)w; c.height = (uint32_t)h;
    } else if (key == "output") {
        if (param.b == param.e) { *err = "Output file name cannot be empty"; return false; }
        c.output_file.assign(param.b, param.e);
    } else if (key == "camera") {
        if (!floats(param, 10, "Invalid camera format", f, err)) return false;
        if (f[9] < 1.0f || f[9] > 179.0f) { *err = "Field of view (fov) must be between 1 and 179 degrees"; return false; }
        std::memcpy(c.cam_pos, f, 12); std::memcpy(c.cam_look, f + 3, 12); std::memcpy(c.cam_up, f + 6, 12); c.cam_fov = f[9];
    } else if (key == "ambient") {
        if (!floats(param, 3, "Invalid ambient light format", f, err)) return false;
        if (!rgb_in_range(f)) { *err = RGB_ERR; return false; }
        std::memcpy(c.ambient, f, 12);
    } else if (key == "sphere") {
        if (!floats(param, 4, "Invalid sphere format", f, err)) return false;
        if (f[3] <= 0.0f) { *err = "Sphere radius must be greater than zero"; return false; }
        c.sph_center.insert(c.sph_center.end(), f, f + 3); c.sph_radius.push_back(f[3]);
        push_material(c, &c.sph_mat); c.sph_order.push_back(c.nshapes++);
    } else if (key == "tri") {
        std::vector<Span> d = split_spaces(param);
        if (d.size() != 3) { *err = "Invalid triangle format"; return false; }
        uint64_t idx[3];
        for (int k = 0; k < 3; k++) if (!rust_uint(d[k].b, d[k].e, ~0ull, &idx[k], err)) return false;
        size_t nv = c.vertices.size() / 3;
        for (int k = 0; k < 3; k++) if (idx[k] >= nv) { *err = "Triangle vertex index out of bounds"; return false; }
        c.tri_v0.insert(c.tri_v0.end(), &c.vertices[3 * idx[0]], &c.vertices[3 * idx[0]] + 3);
        c.tri_v1.insert(c.tri_v1.end(), &c.vertices[3 * idx[1]], &c.vertices[3 * idx[1]] + 3);
        c.tri_v2.insert(c.tri_v2.end(), &c.vertices[3 * idx[2]], &c.vertices[3 * idx[2]] + 3);
        push_material(c, &c.tri_mat); c.tri_order.push_back(c.nshapes++);
    } else if (key == "plane") {
        if (!floats(param, 6, "Invalid plane format", f, err)) return false;
        normalize3(f + 3);                                            // config_builder.rs:445-450
        c.pl_point.insert(c.pl_point.end(), f, f + 3); c.pl_normal.insert(c.pl_normal.end(), f + 3, f + 6);
        push_material(c, &c.pl_mat); c.pl_order.push_back(c.nshapes++);
    } else if (key == "point" || key == "directional") {
        bool dir = key == "directional";
        if (!floats(param, 6, dir ? "Invalid directional light format" : "Invalid point light format", f, err)) return false;
        if (!rgb_in_range(f + 3)) { *err = RGB_ERR; return false; }
        if (dir) normalize3(f);                                       // config_builder.rs:303-306
        c.lights.push_back(dir ? 1.0f : 0.0f);
        c.lights.insert(c.lights.end(), f, f + 6);
    } else if (key == "diffuse") {
        if (!floats(param, 3, "Invalid Vector3 format", f, err)) return false;
        std::memcpy(c.cur_diffuse, f, 12);
        if (!rgb_in_range(f)) { *err = RGB_ERR; return false; }
        if ((f[0] + c.ambient[0]) > 1.0f || (f[1] + c.ambient[1]) > 1.0f || (f[2] + c.ambient[2]) > 1.0f) {
            *err = "Sum of diffuse color and ambient light components must not exceed 1.0"; return false;
        }
    } else if (key == "specular") {
        if (!floats(param, 3, "Invalid Vector3 format", f, err)) return false;
        std::memcpy(c.cur_specular, f, 12);
        if (f[0] < 0.0f || f[1] < 0.0f || f[2] < 0.0f) { *err = "Specular color components must be non-negative"; return false; }
    } else if (key == "shininess") {
        if (!rust_f32(param.b, param.e, f, err)) return false;
        c.cur_shininess = f[0];
        if (f[0] < 0.0f) { *err = "Shininess must be non-negative"; return false; }
    } else if (key == "maxdepth" || key == "maxverts") {
        uint64_t v;
        if (!rust_uint(param.b, param.e, 0xFFFFFFFFull, &v, err)) return false;
        if (key == "maxdepth") c.maxdepth = (uint32_t)v; else { c.maxverts = (uint32_t)v; c.vertices.reserve(3 * (size_t)v); }
    } else if (key == "vertex") {
        if (!floats(param, 3, "Invalid Vector3 format", f, err)) return false;
        if (c.vertices.size() / 3 >= (size_t)c.maxverts) { *err = "Exceeded maximum number of vertices (maxverts)"; return false; }
        c.vertices.insert(c.vertices.end(), f, f + 3);
    } else {
        *err = "Unknown configuration key: " + key;
        return false;
    }
    return true;
}

rth_config* parse_stream(std::istream& in) {
    rth_config* c = new rth_config();
    std::string line, err;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();    // BufRead::lines strips "\r\n" too
        if (!parse_line(*c, line, &err)) { rte::set_error(err); delete c; return nullptr; }
    }
    return c;
}

int fail(const char* m) { rte::set_error(m); return 1; }

}  // namespace

extern "C" {

rth_config* rth_load_config_file(const char* path) {
    if (!path) { rte::set_error("path is NULL"); return nullptr; }
    std::ifstream in(path, std::ios::binary);
    if (!in) { rte::set_error(std::string("No such file or directory (os error 2): ") + path); return nullptr; }
    return parse_stream(in);
}

rth_config* rth_parse_config_text(const char* text) {
    if (!text) { rte::set_error("text is NULL"); return nullptr; }
    std::istringstream in{std::string(text)};
    return parse_stream(in);
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
    std::vector<uint8_t> buf; std::string err; rt_bvh_info local{};
    if (!rtp::pack_scene(*scene, &buf, &local, &err)) { rte::set_error(err); return 1; }
    void* p = std::malloc(buf.size());
    if (!p) return fail("out of memory");
    std::memcpy(p, buf.data(), buf.size());
    *blob = p; *bytes = buf.size();
    if (info) *info = local;
    return 0;
}

void rth_free(void* p) { std::free(p); }

}  // extern "C"
