/* rt_amd.h -- C ABI of the MI355X render path.
 *
 * The reference (a Rust CPU ray tracer) has no FFI of its own.  Its render seam is the
 * method pair
 *     RayTracer::new(config: Config) -> RayTracer          src/raytracer/raytracer.rs:29
 *     RayTracer::render(&self) -> Result<Image, String>    src/raytracer/raytracer.rs:40
 * and these entry points are what a patched `new`/`render` would bind through
 * `extern "C"` (one `rt_upload_scene` in `new`, one `rt_render` in `render`).
 *
 * Conventions
 *   - every struct is plain data, little endian, float = IEEE binary32;
 *   - arrays passed in are owned by the caller and only borrowed for the duration of the call;
 *   - the library owns device memory between rt_upload_scene and rt_shutdown;
 *   - every function returns 0 on success; on failure it returns non-zero and
 *     rt_last_error() describes it (the Err(String) of the reference); nothing aborts;
 *   - calls come from one thread at a time and are blocking, like the reference's render(), except where a
 *     function says it only enqueues work on rt_stream() (rt_render_tiles without stats, rt_assemble_tiles, rt_pack_rgb);
 *   - the library works on streams of its own (rt_stream() is the current one), which do not wait for the default stream
 *     or for anybody else's: device buffers handed to a call must be complete (whatever filled them has finished, or is
 *     ordered before rt_stream() by an event: rt_wait_event), and results of calls that only enqueue are there after
 *     rt_synchronize() or behind an event recorded with rt_record_event;
 *   - a process drives one GPU (rt_init(device)) or the first n GPUs of its node (rt_init_multi(n)): the scene is then
 *     replicated with one RCCL broadcast and every frame is cut into interleaved row tiles over the GPUs and gathered on
 *     the first, all inside rt_upload_scene / rt_render, so the caller's code is the same.  Jobs of one process per GPU
 *     shard the frame themselves (rt_render_tiles, rt_assemble_tiles).
 *   - pixels are 0xFFRRGGBB, row-major, first row on top (src/raytracer/raytracer.rs:85-91,
 *     src/imgcomparator/mod.rs:14-25).
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Per-frame constants: the values RayTracer::render computes once before its pixel loop
 * (src/raytracer/raytracer.rs:46-55) plus the Config fields the loop reads. */
typedef struct rt_frame {
    float pos[3];        /* camera.position                                   raytracer.rs:68  */
    float cv[3];         /* camera_vector = normalize(camera.direction())     raytracer.rs:46  */
    float right[3];      /* normal_to_plane = normalize(cv x up)              raytracer.rs:47  */
    float up[3];         /* v = normalize(right x cv)                         raytracer.rs:48  */
    float pixel_w;       /* pixel_width                                       raytracer.rs:52  */
    float pixel_h;       /* pixel_height = tan(fovrad / 2)                    raytracer.rs:51  */
    float half_w;        /* img_width_by_2                                    raytracer.rs:54  */
    float half_h;        /* img_height_by_2                                   raytracer.rs:55  */
    uint32_t width;      /* config.width                                                       */
    uint32_t height;     /* config.height                                                      */
    uint32_t maxdepth;   /* config.maxdepth                                   raytracer.rs:187 */
    float ambient[3];    /* config.ambient                                    raytracer.rs:181 */
} rt_frame;

/* Fills an rt_frame from the camera of the scene file exactly as raytracer.rs:46-55 and
 * camera.rs:11-13 do (strict binary32, tanf on the host). */
int rt_frame_from_camera(const float position[3], const float look_at[3], const float up[3], float fov_degrees,
                         uint32_t width, uint32_t height, uint32_t maxdepth, const float ambient[3],
                         rt_frame* out);

/* Flattened scene: Config.scene_objects and Config.lights (config_builder.rs:22-23,
 * shape.rs:5-32, light.rs:3-6) as structure-of-arrays.  Shapes keep file order inside each
 * kind; `*_order` gives each shape's position in the file (used only to break exact
 * distance ties the way a file-order scan would; may be NULL = kinds concatenated
 * triangles, spheres, planes). */
typedef struct rt_scene {
    const float* tri_v0;        /* 3 * ntri : Triangle.v0                                  */
    const float* tri_v1;        /* 3 * ntri : Triangle.v1                                  */
    const float* tri_v2;        /* 3 * ntri : Triangle.v2                                  */
    const float* sph_center;    /* 3 * nsph : Sphere.center                                */
    const float* sph_radius;    /*     nsph : Sphere.radius                                */
    const float* pl_point;      /* 3 * npl  : Plane.point                                  */
    const float* pl_normal;     /* 3 * npl  : Plane.normal, already normalised by the parser (config_builder.rs:445-450) */
    const float* tri_mat;       /* 7 * ntri : diffuse rgb, specular rgb, shininess         */
    const float* sph_mat;       /* 7 * nsph                                                */
    const float* pl_mat;        /* 7 * npl                                                 */
    const uint32_t* tri_order;  /* ntri or NULL */
    const uint32_t* sph_order;  /* nsph or NULL */
    const uint32_t* pl_order;   /* npl  or NULL */
    uint32_t ntri, nsph, npl;
    const float* lights;        /* 7 * nlights : kind (0 = point, 1 = directional), vec3 (position, or unit
                                   direction towards the light, normalised by the parser), colour rgb   */
    uint32_t nlights;
} rt_scene;

typedef struct rt_stats {
    uint64_t primary;        /* primary rays   = pixels rendered by this call               */
    uint64_t shadow;         /* shadow rays    = one per (hit, light)     raytracer.rs:126-141 */
    uint64_t secondary;      /* reflection rays actually spawned          raytracer.rs:192  */
    uint64_t nodes_visited;  /* bounding boxes tested (two per node pair visited)           */
    uint64_t prims_tested;   /* exact shape tests (48 bytes each)                           */
    uint64_t box_passes;     /* instrumented runs: wavefront-level passes of the box-test loop; lane use =
                                nodes_visited / 2 / (64 * box_passes)                                   */
    uint64_t shape_passes;   /* the same for the shape-test loop: prims_tested / (64 * shape_passes)  */
    uint64_t phase_passes[9];/* instrumented runs: wavefront-level passes through the phases of a path level: closest-hit
                                searches started, hits shaded, shadow rays traced light by light, lights shaded (not in
                                shadow), the share of those shadow rays in box_passes and in shape_passes, side-by-side
                                rounds (shadow rays and next closest-hit rays of few live paths, one per lane), and their
                                share in box_passes and in shape_passes                                            */
    double kernel_ms;        /* render kernel(s), HIP events on the library's stream        */
    double total_ms;         /* whole call, host clock                                      */
} rt_stats;

/* rt_build information about the tree the library built for the resident scene. */
typedef struct rt_bvh_info {
    uint32_t node_pairs;     /* 32-byte records: two 16-byte child boxes (16-bit grid coordinates) each */
    uint32_t leaves;
    uint32_t max_depth;      /* node pairs on the longest root-to-leaf path (at most 63)    */
    uint32_t max_leaf;       /* largest number of shapes in one leaf                        */
    uint32_t bvh_prims;      /* shapes inside the tree                                      */
    uint32_t list_prims;     /* shapes always tested (planes; everything in tiny scenes)    */
    double build_ms;         /* host time of the upload call that made the scene resident; 0 for adopted or cached blobs */
    double sah_cost;
} rt_bvh_info;

/* Binds the library to HIP device `device` and creates its streams.  Replaces nothing in the
 * reference (it has no device); called once before RayTracer::new. */
int rt_init(int device);

/* The same for the first `ngpus` GPUs of the node (1 .. RT_MAX_GPUS), driven by this one process and thread (the caller of
 * the reference's render() is a single-threaded main, src/main.rs:11-14): creates streams on every GPU and one RCCL
 * communicator per GPU (ncclCommInitAll).  From then on every upload replicates the scene to all GPUs with one
 * ncclBroadcast, and rt_render / rt_render_frame cut each frame into tiles of RT_TILE_ROWS rows, tile t to GPU t % ngpus,
 * gather the rows on the first GPU with one grouped ncclSend / ncclRecv and restore frame order there.  Device pointers
 * the other entry points take or return (rt_render_tiles, rt_scene_blob, rt_compare_frames, ...) refer to the first GPU. */
#define RT_MAX_GPUS 8
#define RT_TILE_ROWS 8
int rt_init_multi(int ngpus);
int rt_device_count(void);   /* GPUs the library drives; 0 before rt_init */

/* RayTracer::new (raytracer.rs:29-38): builds the bounding volume hierarchy on the host
 * (the reference calls Bvh::build_par at raytracer.rs:32), lays the scene out for the GPU
 * and uploads it.  The scene stays resident until the next upload or rt_shutdown. */
int rt_upload_scene(const rt_scene* scene);

/* rt_upload_scene with options.  RT_BUILD_ON_GPU builds the tree on the GPU instead of on the host: the
 * same binned surface-area split search from the root down, a tree level per round of kernels while the
 * nodes are large and a wavefront per node in LDS once they hold at most 256 shapes.  A tenth of a second
 * instead of a second and a half for ten million triangles, for a tree of the same quality.  The frame
 * does not depend on the choice.  Falls back to the host builder by itself for tiny scenes and for a tree
 * deeper than the traversal stack. */
#define RT_BUILD_ON_GPU 1u
int rt_upload_scene_ex(const rt_scene* scene, uint32_t flags);

int rt_get_bvh_info(rt_bvh_info* out);

/* The resident scene as one contiguous device blob (for replicating it to other GPUs with
 * a broadcast) and the inverse: adopt a blob that was received into device memory owned by
 * the caller, which must stay valid until the next upload/adopt/shutdown. */
int rt_scene_blob(void** device_ptr, size_t* bytes);
int rt_copy_scene_blob(void* device_dst, size_t bytes);   /* device-to-device copy of the blob, waits for completion */
int rt_adopt_scene_blob(void* device_ptr, size_t bytes);

/* RayTracer::new from a scene that was laid out earlier (rth_pack_scene, or a cache file read
 * with rth_scene_cache_open): skips the tree build the reference repeats on every start
 * (Bvh::build_par, raytracer.rs:32).  The blob is checked in full before it reaches the GPU
 * (sections, every node and shape reference, tree shape), so a damaged file gives an error. */
int rt_upload_scene_blob(const void* host_blob, size_t bytes);
/* The resident blob copied to host memory (bytes as reported by rt_scene_blob), e.g. to write a cache. */
int rt_download_scene_blob(void* host_dst, size_t bytes);

/* RayTracer::render (raytracer.rs:40-79): renders the whole frame into host memory,
 * out_pixels[width * height].  `stats` may be NULL; when `count_work` is non-zero the
 * instrumented kernel also fills nodes_visited / prims_tested / shadow / secondary
 * (slower; never used for timing). */
int rt_render(const rt_frame* frame, uint32_t* out_pixels, rt_stats* stats, int count_work);

/* RayTracer::render with the frame left in device memory (of the first GPU): *d_frame points at width * height pixels in
 * a buffer of the library that stays valid until the next frame on the same lane.  With stats == NULL the call only
 * enqueues (on the current lane of every GPU; rt_synchronize waits for them); otherwise it waits and fills *stats
 * (kernel_ms: the longest of the GPUs' kernels; counts: all GPUs together). */
int rt_render_frame(const rt_frame* frame, const uint32_t** d_frame, rt_stats* stats, int count_work);

/* The same loop restricted to the row tiles owned by one shard: tile t (rows
 * [t*tile_rows, (t+1)*tile_rows)) belongs to shard t % nshards.  Rows of the owned tiles are
 * written compactly, in increasing order, to the DEVICE buffer d_out (rt_shard_rows() rows of
 * `width` pixels).  Rows of the frame are independent in the reference too
 * (par_chunks_mut, raytracer.rs:57-59).  Asynchronous on rt_stream() unless stats != NULL. */
int rt_render_tiles(const rt_frame* frame, uint32_t shard, uint32_t nshards, uint32_t tile_rows,
                    uint32_t* d_out, rt_stats* stats, int count_work);
uint32_t rt_shard_rows(uint32_t height, uint32_t shard, uint32_t nshards, uint32_t tile_rows);

/* Inverse of the sharding on the gathering GPU: d_gathered holds nshards compact buffers of
 * `rows_per_shard_stride` rows each (shard s at offset s * stride * width pixels); writes the
 * assembled frame to d_frame[width * height]. */
int rt_assemble_tiles(const uint32_t* d_gathered, uint32_t rows_per_shard_stride, uint32_t nshards,
                      uint32_t tile_rows, uint32_t width, uint32_t height, uint32_t* d_frame);

/* Image::compare (src/imgcomparator/mod.rs:54-89) on two frames in device memory: *count = number of
 * pixels with a colour channel differing by more than 1; d_diff (may be NULL) receives the difference
 * image (dr << 16 | dg << 8 | db for those pixels, 0 elsewhere).  Different dimensions are the
 * reference's error "Images have different dimensions".  Buffers 16-byte aligned.  Waits for the result. */
int rt_compare_frames(const uint32_t* d_a, uint32_t width_a, uint32_t height_a, const uint32_t* d_b, uint32_t width_b, uint32_t height_b,
                      uint32_t* d_diff, uint64_t* count);

/* rt_compare_frames for callers that cannot allocate: code under HIP graph capture, serving loops and harnesses with a
 * fixed memory budget.  No counterpart in the reference.  Same kernel, but its pixel counter lives in a static device
 * array of the library (256 bytes per GPU, held until the process ends) instead of memory from hipMalloc, and the count
 * is delivered to DEVICE memory: *d_count (8-byte aligned) receives it once the work enqueued on rt_stream() has run; the
 * call itself only enqueues (a memset of the counter, the kernel, an 8-byte device-to-device copy) and calls no
 * hipMalloc, hipFree or other allocating routine.  At most RT_NOALLOC_MAX_PIXELS pixels.  Every call on a GPU uses
 * that GPU's one array: calls that overlap on one GPU (different lanes, other threads) race; calls on one lane do not. */
#define RT_NOALLOC_MAX_PIXELS 2147483648ull
int rt_compare_frames_noalloc(const uint32_t* d_a, uint32_t width_a, uint32_t height_a, const uint32_t* d_b, uint32_t width_b,
                              uint32_t height_b, uint32_t* d_diff, uint64_t* d_count);

/* The per-pixel loop of save_image (src/imgcomparator/mod.rs:156-162) on the device: 0x??RRGGBB words to
 * R,G,B byte triples (3 * width * height bytes) for a PNG encoder.  Asynchronous on rt_stream(). */
int rt_pack_rgb(const uint32_t* d_pixels, uint32_t width, uint32_t height, uint8_t* d_rgb);

/* Lanes.  The library has RT_LANES lanes, each a HIP stream with its own work buffers; every render, assemble,
 * compare and pack call uses the lane selected last (lane 0 after rt_init), and rt_stream / rt_synchronize
 * refer to it.  Calls on one lane run in order.  Calls on different lanes share only the read-only scene and
 * may overlap on the GPU: a caller that renders frame after frame cycles through lanes (and output buffers), so
 * that the ramp-up and the long tail of one launch are filled by the next (a launch lasts at least as long as
 * its slowest wavefront, which matters most for the small launches of a frame cut over many GPUs: an eighth of
 * the 4K dragon frame takes 1.01 ms per frame on one lane, 0.56 on two, 0.50 on four).  Scene changes wait
 * for all lanes.  No counterpart in the reference (its frames are rendered one call at a time). */
#define RT_LANES 4
int rt_select_lane(int lane);

/* Tile order across runs.  Every launch measures what its regions of tiles cost and the next launch of the same frame
 * size hands the dearest out first, which shortens its tail; the first frame after an upload has nothing to go by (on the
 * 4K dragon frame, handed out in plain raster order: 0.5 ms more than the 2.5 of a later frame) and is ordered by a guess
 * instead: regions by how many of the tree's shapes, and of their mirror images in the specular planes of the scene, are seen
 * in and around them (rt_tile_costs_guess returns that guess, for inspection; it changes nothing).  A host that renders the same scene file again in a later run (the reference's
 * main renders one frame per run, main.rs:11-15) can keep the costs of the last frame beside its scene cache and hand them
 * back before the first frame.  They address the launch rt_render / rt_render_frame make for `frame` on GPU `gpu`
 * (0 <= gpu < rt_device_count()) on the current lane: rt_tile_costs_count gives the number of regions (0: the frame is too
 * small to be ordered, nothing to save), rt_tile_costs_export copies the costs of the last launch (it waits for the lane;
 * fails if that frame size has not been rendered on the lane since the upload), rt_tile_costs_import sets the order of the
 * next launch from costs saved earlier.  Costs are relative weights: any values are safe, they change the order of the
 * work and never a pixel.  No counterpart in the reference. */
int rt_tile_costs_count(const rt_frame* frame, int gpu, uint32_t* count);
int rt_tile_costs_export(const rt_frame* frame, int gpu, uint32_t* costs, uint32_t count);
int rt_tile_costs_import(const rt_frame* frame, int gpu, const uint32_t* costs, uint32_t count);
int rt_tile_costs_guess(const rt_frame* frame, int gpu, uint32_t* costs, uint32_t count);

/* The HIP stream (hipStream_t) of the current lane on the first GPU, for callers that order their own
 * device work or events against it (a frame over several GPUs ends on it), and a wait for the current lane of every GPU. */
void* rt_stream(void);
int rt_synchronize(void);

/* Ordering against work on other streams without waiting on the host.  rt_wait_event: everything enqueued on the current
 * lane after this call runs after `hip_event` (a hipEvent_t the caller has recorded on the stream that fills the buffers it
 * is about to hand in: a memset or a kernel of its own on the default stream, say).  rt_record_event: records the caller's
 * event behind everything enqueued on the current lane so far (a frame over several GPUs ends there), for the caller's
 * stream to wait on before it reads a result.  Both refer to the first GPU's stream of the current lane and return at
 * once.  No counterpart in the reference (it has one address space and no streams). */
int rt_wait_event(void* hip_event);
int rt_record_event(void* hip_event);

/* Err(String) of the last failing call on this thread. */
const char* rt_last_error(void);

/* Frees the resident scene, work buffers and the stream. */
void rt_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
