/* rt_host.h -- host-side companions of the render path, C ABI.
 *
 * The reference keeps scene parsing, the scene model and image comparison in Rust
 * (src/raytracer/config/config_builder.rs, src/imgcomparator/mod.rs).  A Rust host would keep
 * using its own; these twins exist so that hosts without the Rust crate (the Python package, its
 * tests and the benchmark) load the same .scene files, flatten them into
 * rt_scene / rt_frame exactly as the Rust side would, and compare images the way the
 * reference's tests do.  Nothing here touches the GPU.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Config (config_builder.rs:14-24) as parsed by ParsedConfigState::load_config_file
 * (config_builder.rs:127-150).  Opaque; owns every array rth_config_scene hands out. */
typedef struct rth_config rth_config;

/* Returns NULL on error; rt_last_error() then carries the reference's Err(String) text
 * (for example "Unknown configuration key: foo", "Invalid sphere format"). */
rth_config* rth_load_config_file(const char* path);
/* Same grammar, text already in memory. */
rth_config* rth_parse_config_text(const char* text);
void rth_config_free(rth_config* c);

typedef struct rth_config_info {
    uint32_t width, height, maxdepth, maxverts;
    uint32_t ntri, nsph, npl, nlights;
    float camera_position[3], camera_look_at[3], camera_up[3], camera_fov;
    float ambient[3];
} rth_config_info;
int rth_config_get_info(const rth_config* c, rth_config_info* out);
/* Config.output_file; the pointer stays valid until the config is freed. */
const char* rth_config_output_file(const rth_config* c);
/* Overrides Config.width / Config.height (what editing the `size` line of the file does). */
int rth_config_set_size(rth_config* c, uint32_t width, uint32_t height);

/* The scene flattened for rt_upload_scene; arrays stay valid until the config is freed. */
int rth_config_scene(const rth_config* c, rt_scene* out);
/* The frame constants of RayTracer::render for this config (raytracer.rs:46-55). */
int rth_config_frame(const rth_config* c, rt_frame* out);

/* Image::compare (imgcomparator/mod.rs:54-89).  Pixels are packed 0x??RRGGBB, the top byte is
 * ignored.  Fails with "Images have different dimensions" when the sizes differ.  `diff`
 * (optional, w*h) receives the difference image, `count` the number of pixels with a
 * channel differing by more than 1. */
int rth_image_compare(const uint32_t* a, uint32_t wa, uint32_t ha, const uint32_t* b, uint32_t wb, uint32_t hb,
                      uint32_t* diff, uint64_t* count);

/* Host-only build of the device scene blob (tree + shape records), for inspection and
 * tests without a GPU.  Free with rth_free. */
int rth_pack_scene(const rt_scene* scene, void** blob, size_t* bytes, rt_bvh_info* info);
void rth_free(void* p);
/* The check rt_upload_scene_blob applies before a blob reaches the GPU: 0 when every section,
 * node and shape reference is in range and the tree is a tree no deeper than the traversal stack. */
int rth_validate_blob(const void* blob, size_t bytes);

/* PNG files.  save_image (src/imgcomparator/mod.rs:150-166): 0x??RRGGBB pixels (or R,G,B bytes as
 * rt_pack_rgb produces them) to an 8-bit RGB PNG, rows compressed in parallel.  file_to_image
 * (mod.rs:128-148): any non-interlaced 8/16-bit PNG to 0x00RRGGBB pixels (alpha dropped);
 * *pixels is released with rth_free. */
int rth_save_png(const char* path, const uint32_t* pixels, uint32_t width, uint32_t height);
int rth_save_png_rgb(const char* path, const uint8_t* rgb, uint32_t width, uint32_t height);
int rth_load_png(const char* path, uint32_t** pixels, uint32_t* width, uint32_t* height);

/* Binary scene cache.  The reference parses the text and rebuilds its tree on every start
 * (main.rs:9-11, raytracer.rs:29-38); for a 10-million-triangle scene that is seconds of work
 * against milliseconds of rendering.  One cache file holds the Config fields the render loop
 * reads (config_builder.rs:14-24 without the shapes), the identity (size, modification time) of
 * the scene file it was made from, and the device blob ready for rt_upload_scene_blob.
 *   write:  parse, rth_pack_scene (or rt_upload_scene + rt_download_scene_blob), rth_scene_cache_write
 *   read:   rth_scene_cache_open (maps the file, checks header and checksum), rth_scene_cache_is_fresh,
 *           rt_upload_scene_blob(rth_scene_cache_blob(..)), rth_scene_cache_config for rth_config_frame */
typedef struct rth_scene_cache rth_scene_cache;
int rth_scene_cache_write(const char* path, const rth_config* c, const void* blob, size_t bytes, const char* source_path);
rth_scene_cache* rth_scene_cache_open(const char* path);                       /* NULL on failure */
const void* rth_scene_cache_blob(const rth_scene_cache* h, size_t* bytes);     /* valid until close */
rth_config* rth_scene_cache_config(const rth_scene_cache* h);                  /* settings only, no shapes; rth_config_free */
int rth_scene_cache_is_fresh(const rth_scene_cache* h, const char* source_path);   /* 1 when source_path is the file it was made from, unchanged */
void rth_scene_cache_close(rth_scene_cache* h);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
