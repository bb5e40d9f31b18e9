// frame.cpp -- host-side helpers of the render calls that need no GPU: the per-frame constants of RayTracer::render
// and the copy of a finished frame out of the pinned staging buffer.
#include "../../include/rt_amd.h"

#include <cmath>
#include <cstring>
#include <string>

#include "errors.h"
#include "host_copy.h"

namespace {
int fail_msg(const std::string& m) { rte::set_error(m); return 1; }
}  // namespace

namespace rtu {

void parallel_copy(void* dst, const void* src, size_t bytes) {
    const size_t PIECE = (size_t)1 << 20;
    const long long pieces = (long long)((bytes + PIECE - 1) / PIECE);
#pragma omp parallel for schedule(static) if (pieces >= 4) num_threads(8)
    for (long long p = 0; p < pieces; p++) {
        const size_t a = (size_t)p * PIECE, n = bytes - a < PIECE ? bytes - a : PIECE;
        std::memcpy((char*)dst + a, (const char*)src + a, n);
    }
}

}  // namespace rtu

extern "C" {

int rt_frame_from_camera(const float position[3], const float look_at[3], const float up[3], float fov_degrees, uint32_t width,
                         uint32_t height, uint32_t maxdepth, const float ambient[3], rt_frame* out) {
    if (!position || !look_at || !up || !ambient || !out) return fail_msg("rt_frame_from_camera: NULL argument");
    if (width == 0 || height == 0) return fail_msg("Width and height must be greater than zero");
    // Strict binary32 in the reference's order (raytracer.rs:46-55, camera.rs:11-13).
    struct V { float x, y, z; };
    auto sub = [](V a, V b) { return V{a.x - b.x, a.y - b.y, a.z - b.z}; };
    auto dot = [](V a, V b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; };
    auto cross = [](V a, V b) { return V{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; };
    auto normalize = [&](V a) { float inv = 1.0f / sqrtf(dot(a, a)); return V{a.x * inv, a.y * inv, a.z * inv}; };
    V pos{position[0], position[1], position[2]}, look{look_at[0], look_at[1], look_at[2]}, upv{up[0], up[1], up[2]};
    V cv = normalize(normalize(sub(look, pos)));
    V right = normalize(cross(cv, upv));
    V v = normalize(cross(right, cv));
    const float PI_F = 3.14159265358979323846f;
    float fovrad = fov_degrees * PI_F / 180.0f;
    float ph = tanf(fovrad / 2.0f);
    float pw = ph * ((float)width / (float)height);
    out->pos[0] = pos.x; out->pos[1] = pos.y; out->pos[2] = pos.z;
    out->cv[0] = cv.x; out->cv[1] = cv.y; out->cv[2] = cv.z;
    out->right[0] = right.x; out->right[1] = right.y; out->right[2] = right.z;
    out->up[0] = v.x; out->up[1] = v.y; out->up[2] = v.z;
    out->pixel_w = pw; out->pixel_h = ph;
    out->half_w = (float)width / 2.0f; out->half_h = (float)height / 2.0f;
    out->width = width; out->height = height; out->maxdepth = maxdepth;
    out->ambient[0] = ambient[0]; out->ambient[1] = ambient[1]; out->ambient[2] = ambient[2];
    return 0;
}

}  // extern "C"
