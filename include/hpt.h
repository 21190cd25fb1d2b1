/* hpt.h -- C ABI of the MI355X-native path-tracing hot path (libhpt.so).
 *
 * Drop-in boundary for the reference renderer's unidirectional path-tracing launch
 * API.  Every entry point takes plain pointers and sizes; the scene records are the
 * reference's own host PODs, byte for byte:
 *
 *   light    144 B  reference include/geometric.cuh:73-78   (CudaLight)
 *   sphere   100 B  reference include/geometric.cuh:29-35   (CudaSphere)
 *   triangle 120 B  reference include/geometric.cuh:37-42   (CudaTriangle)
 *   camera    84 B  reference include/geometric.cuh:67-69   (CudaCamera)
 *   image    W*H*3 float32, row-major, row 0 = top, linear RGB mean radiance
 *            (reference src/pt_cu.cu:30,248)
 *
 * What each entry point replaces in the reference:
 *
 *   hpt_pt_render_wrapper     pt_render_wrapper, include/pt_cu.cuh:6-13 (defined
 *                             src/pt_cu.cu:255-297): alloc + upload + render + download
 *                             in one blocking call.  include/hpt_reference_api.hpp
 *                             declares the C++-linkage adapter of the same name.
 *   hpt_scene_create/destroy  the per-call cudaMalloc/cudaMemcpy/cudaFree of the scene,
 *                             src/pt_cu.cu:270-278,292-296, hoisted so a caller that
 *                             renders repeatedly (reference src/main.cpp:416) uploads and
 *                             builds the BVH once.
 *   hpt_render_pt             cuda_path_trace_kernel launch + D2H, src/pt_cu.cu:282-290.
 *   hpt_render_pt_device      same, leaving the result in device memory on a caller
 *                             stream (no reference equivalent; used for multi-GPU tiling
 *                             and for timing with inputs resident in HBM).
 *
 * All functions return 0 on success, non-zero on error; hpt_last_error() describes the
 * last error of the calling thread.  Calls on one scene handle are not re-entrant.
 */
#ifndef HPT_H
#define HPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HPT_OK 0
#define HPT_ERR_INVALID 1
#define HPT_ERR_DEVICE 2
#define HPT_ERR_NOMEM 3

#define HPT_LIGHT_BYTES 144
#define HPT_SPHERE_BYTES 100
#define HPT_TRIANGLE_BYTES 120
#define HPT_CAMERA_BYTES 84

typedef struct hpt_scene hpt_scene;

/* Render parameters that the reference fixes at compile time or leaves to the clock. */
typedef struct hpt_params {
    uint64_t seed;            /* RNG stream key (the reference seeds cuRAND from time(NULL), pt_cu.cu:282) */
    int32_t sample_offset;    /* global index of this call's first sample (progressive rendering) */
    int32_t max_delta;        /* cap on free delta bounces per sample (reference: uncapped, pt_cu.cu:228); 0 -> 64 */
    int32_t rank;             /* image-tile partition: this device renders tiles t with t % world == rank */
    int32_t world;            /* number of devices sharing the image; 0 or 1 -> whole image */
    int32_t tile;             /* tile edge in pixels, multiple of 8; 0 -> 32 */
    int32_t samples_per_pass; /* samples of every local pixel in flight at once; 0 -> auto */
    int32_t flags;            /* HPT_FLAG_* */
    int32_t reserved;
} hpt_params;

#define HPT_FLAG_BRUTE_FORCE 1   /* scan every primitive instead of the BVH (tests) */
#define HPT_FLAG_COUNT_WORK 2    /* count BVH boxes/triangles tested (slower; fills hpt_stats) */
#define HPT_FLAG_OUTPUT_SUM 4    /* leave the per-pixel sum over this call's samples, not the mean */
#define HPT_FLAG_TIME_KERNELS 8  /* bracket every kernel launch with HIP events (fills hpt_stats.ms_*) */
#define HPT_FLAG_RUSSIAN_ROULETTE 16 /* PT: unbiased roulette after every non-delta bounce, survival
                                      * q = clamp(max throughput channel, 0.05, 1).  The reference has no
                                      * roulette: off by default; it costs one extra uniform per
                                      * bounce, so images differ from the roulette-free ones sample by sample */
#define HPT_FLAG_SINGLE_PIPELINE 32 /* PT: one pass in flight at a time (default: two passes of a render run
                                      * concurrently on two streams with a workspace each; same image) */
#define HPT_FLAG_NO_HOST_WAIT 64    /* hpt_render_pt_device / hpt_render_bdpt_device never wait for the device.  Every
                                      * iteration up to eye_depth + max_delta is enqueued whether or not a path is still
                                      * alive; the iterations past eye_depth get small fixed grids (8 workgroups per CU;
                                      * the trace, resume and shade kernels of the PT path and the extend, connect and
                                      * reduce kernels of the BDPT path walk their queues with a stride; k_bdpt_vertex keeps
                                      * its one-chunk-per-workgroup grid, whose workgroups return at once on an empty
                                      * queue), so an iteration that finds its queue empty costs a few microseconds per
                                      * launch: config 3 with the default max_delta of 64, of which it needs 4, renders
                                      * in the same 131 ms either way.  What it costs is a fixed ~200 (PT) / ~260 (BDPT)
                                      * launches per pass, i.e. about a millisecond on a render of a few milliseconds --
                                      * hence opt-in; the default looks at a 4-byte counter every other tail iteration.
                                      * Same image */

typedef struct hpt_stats {
    uint64_t samples;         /* camera samples traced by the last render */
    uint64_t closest_rays;    /* closest-hit rays */
    uint64_t shadow_rays;     /* any-hit rays */
    uint64_t boxes_closest;   /* child boxes slab-tested by closest-hit rays (2 per inner node); COUNT_WORK only */
    uint64_t tris_closest;    /* triangle tests by closest-hit rays; COUNT_WORK only */
    uint64_t boxes_shadow;    /* same, any-hit rays */
    uint64_t tris_shadow;
    uint64_t path_iters;      /* (path, bounce) shading steps */
    double ms_total;          /* device time first-to-last kernel of the last render (HIP events) */
    double ms_extend, ms_shade, ms_connect, ms_other;   /* per-kernel-class sums; TIME_KERNELS only */
    uint32_t n_extend, n_shade, n_connect, n_other;     /* launches per class */
    uint32_t bvh_nodes, bvh_depth, n_tris, n_materials;
    double ms_bvh_build, ms_upload;
    /* SIMD efficiency of the traversal loops (COUNT_WORK only): lane_steps = loop trips summed over
     * lanes, wave_steps = 64 x the longest lane's trips summed over the wave's rays */
    uint64_t lane_steps_closest, wave_steps_closest, lane_steps_shadow, wave_steps_shadow;   /* inner-node trips */
    uint64_t leaf_lane_closest, leaf_wave_closest, leaf_lane_shadow, leaf_wave_shadow;       /* leaf trips */
    /* split trace step: the first launch gives every ray split_budget node steps, the second (resume)
     * launch finishes the rays that needed more */
    double ms_resume;                 /* resume launches; TIME_KERNELS only (ms_extend / ms_connect then hold the first launches) */
    uint32_t n_resume, split_budget;  /* split_budget 0: single-launch trace steps */
    uint64_t traced_rays_last_pass, long_rays_last_pass;   /* rays entering the trace steps of the last pass / set aside for resume */
    /* connection stage of a bidirectional render (COUNT_WORK only; reference loop src/cpu_bdpt.cpp:387-440): candidate (eye vertex,
     * light vertex) pairs, pairs that pass the culls (zero throughput, distance, cosines, emission cone), shadow rays traced (both
     * BSDF values non-zero), unoccluded ones; and the work of those shadow rays: BVH nodes visited (64 B each), triangle, sphere
     * and group-box tests */
    uint64_t bd_pairs, bd_survivors, bd_shadow_rays, bd_unoccluded, bd_nodes, bd_tris, bd_spheres, bd_group_boxes;
} hpt_stats;

const char *hpt_last_error(void);

/* Number of visible HIP devices (<0 on error). */
int hpt_device_count(void);

/* Flattens the reference records into the device layout, builds the BVH on the host and
 * uploads everything to the current HIP device.  Records are copied; the caller keeps
 * ownership of its arrays. */
int hpt_scene_create(const void *lights, int num_lights,
                     const void *spheres, int num_spheres,
                     const void *triangles, int num_triangles,
                     hpt_scene **out_scene);
void hpt_scene_destroy(hpt_scene *scene);

/* Number of float3 slots of the packed local framebuffer for this (W, H, params) -- the
 * size hpt_render_pt_device writes and hpt_untile reads per rank. */
int64_t hpt_local_pixels(int W, int H, const hpt_params *params);

/* Blocking render into a caller-owned host image of W*H*3 floats (whole image:
 * params->world must be 0 or 1). */
int hpt_render_pt(hpt_scene *scene, const void *camera, int W, int H,
                  int eye_depth, int spp, const hpt_params *params, float *host_image);

/* Render of this rank's tiles into device memory: d_local holds hpt_local_pixels() float3 records in local
 * tile order.  Everything is enqueued on hip_stream (plus one stream of the scene, joined back before the
 * call returns); the result is complete when the stream reaches the end of what the call enqueued.  The call
 * returns without waiting for the device as long as no path outlives eye_depth iterations; paths kept alive by
 * free delta bounces (mirror, glass: reference src/pt_cu.cu:228) need further iterations whose number only the
 * device knows, and for those the calling thread waits on a 4-byte read-back every other iteration -- i.e. on
 * scenes with delta materials the call MAY BLOCK THE HOST for most of the render's duration.  It never blocks
 * the device: both pipelines of a render keep running while the host waits.  With HPT_FLAG_NO_HOST_WAIT the call
 * only enqueues (blind launches instead of read-backs). */
int hpt_render_pt_device(hpt_scene *scene, const void *camera, int W, int H,
                         int eye_depth, int spp, const hpt_params *params,
                         void *d_local, void *hip_stream);

/* Scatters `world` packed local framebuffers, laid out [rank][local pixel] in d_gathered,
 * into the row-major W*H image d_image (both device pointers). */
int hpt_untile(const void *d_gathered, void *d_image, int W, int H,
               const hpt_params *params, void *hip_stream);

/* One-shot equivalent of the reference's pt_render_wrapper (include/pt_cu.cuh:6-13):
 * scene_min/scene_max/light_depth/light_sample are accepted and ignored there too
 * (src/pt_cu.cu:259-262).  seed < 0 -> seed from the clock like the reference.
 *
 * The reference uploads the scene and allocates its buffers on every call (src/pt_cu.cu:270-296), and
 * its interactive front-end calls the wrapper once per frame (src/main.cpp:416).  Here the two one-shot
 * wrappers keep the last scene (device records, BVH, workspace) and reuse it when the next call on the
 * same device passes byte-identical light/sphere/triangle arrays; anything else rebuilds.  The images
 * are the same either way.  hpt_wrapper_cache_clear() releases the kept scene; HPT_WRAPPER_CACHE=0 in
 * the environment switches the reuse off. */
void hpt_wrapper_cache_clear(void);
int hpt_pt_render_wrapper(const void *lights, int num_lights,
                          const void *spheres, int num_spheres,
                          const void *triangles, int num_triangles,
                          const float scene_min[3], const float scene_max[3],
                          const void *camera, float *host_image, int W, int H,
                          int light_depth, int light_sample, int eye_depth, int spp,
                          int64_t seed);

/* ---- bidirectional estimator of the reference's CPU renderer ------------------------------------
 * Replaces bdpt_render_wrapper (reference include/bdpt_cu.cuh:30-37, src/bdpt_cu.cu:538-674) and
 * run_cuda_bdpt (include/bdpt_cu_helper.h:6).  What is computed follows run_cpu_bdpt (reference
 * src/cpu_bdpt.cpp:173-488) wherever it and the CUDA BDPT kernel disagree: the CPU scene
 * model with its one-level group boxes, nl*spl light subpaths carrying illum/spl, every eye vertex
 * connected to every light vertex with the ratio-sum MIS weight.
 *
 * hpt_scene_set_groups hands over the scene file's grouping (kind 0 sphere / 1 triangle, index into
 * the arrays given to hpt_scene_create, group id; insertion order), which decides tie-breaks and the
 * per-group box culls of the CPU model.  Without it: one group, spheres then triangles. */
int hpt_scene_set_groups(hpt_scene *scene, const int32_t *obj_kind, const int32_t *obj_index,
                         const int32_t *obj_group, int num_objects);
int hpt_render_bdpt(hpt_scene *scene, const void *camera, int W, int H, int eye_depth, int light_depth,
                    int spp, int spl, const hpt_params *params, float *host_image);
int hpt_render_bdpt_device(hpt_scene *scene, const void *camera, int W, int H, int eye_depth, int light_depth,
                           int spp, int spl, const hpt_params *params, void *d_local, void *hip_stream);
/* One-shot, argument list of the reference's bdpt_render_wrapper; the lights arrive with illum already
 * divided by light_sample (reference src/bdpt_cu_helper.cpp:60-62), which is undone here. */
int hpt_bdpt_render_wrapper(const void *lights, int num_lights, const void *spheres, int num_spheres,
                            const void *triangles, int num_triangles,
                            const float scene_min[3], const float scene_max[3],
                            const void *camera, float *host_image, int W, int H,
                            int light_depth, int light_sample, int eye_depth, int spp, int spl, int64_t seed);

int hpt_get_stats(const hpt_scene *scene, hpt_stats *out);

/* ---- multi-device fan-out inside the blocking call -----------------------------------------------
 * The reference's launch API is one blocking call per frame (run_cuda_pt -> pt_render_wrapper, reference
 * src/pt_cu_helper.cpp:66-77, src/pt_cu.cu:255-297, one device).  hpt_multi_* is the same call over the
 * devices of one node, in ONE process: the scene is flattened and its BVH built once and uploaded to every
 * device; every device renders its image tiles (hpt_params.rank/world are set internally) on its own
 * stream, driven by its own host thread; the packed local framebuffers are gathered on device_ids[0] with
 * one ncclGather per device inside one RCCL group (librccl.so is dlopen'ed on first use; each sender uses
 * its own xGMI link to the root), un-tiled there and copied to host_image.  The image is bit-identical to
 * hpt_render_pt's for any number of devices.
 *   device_ids   num_devices HIP device ordinals, NULL = 0 .. num_devices-1; num_devices <= 0 = all visible
 *   exchange     0 = RCCL (distinct devices; fails if RCCL cannot be loaded or initialised)
 *                1 = hipMemcpyPeerAsync into the root's buffer (boxes without RCCL; tests that place
 *                    several ranks on one device, which an RCCL communicator refuses)
 * hpt_wrapper_set_devices(n) (or HPT_DEVICES=n in the environment, read once) makes the two one-shot
 * wrappers -- hence the reference's unmodified run_cuda_pt / run_cuda_bdpt -- render on n devices this way;
 * 0 returns to the environment's value, 1 to a single device. */
typedef struct hpt_multi hpt_multi;
int hpt_multi_create(const void *lights, int num_lights, const void *spheres, int num_spheres,
                     const void *triangles, int num_triangles,
                     const int *device_ids, int num_devices, int exchange, hpt_multi **out);
void hpt_multi_destroy(hpt_multi *multi);
int hpt_multi_num_devices(const hpt_multi *multi);
int hpt_multi_set_groups(hpt_multi *multi, const int32_t *obj_kind, const int32_t *obj_index,
                         const int32_t *obj_group, int num_objects);
int hpt_multi_render_pt(hpt_multi *multi, const void *camera, int W, int H, int eye_depth, int spp,
                        const hpt_params *params, float *host_image);
int hpt_multi_render_bdpt(hpt_multi *multi, const void *camera, int W, int H, int eye_depth, int light_depth,
                          int spp, int spl, const hpt_params *params, float *host_image);
/* device time of every rank's render (HIP events), of the exchange step, and host wall time of the last call */
int hpt_multi_get_timing(const hpt_multi *multi, double *render_ms_per_device, double *gather_ms, double *total_ms);
int hpt_wrapper_set_devices(int num_devices);

/* Function-level probe of the device BSDF code (tests): evaluates, for n caller-given records, the
 * device versions of bsdf_evaluate / bsdf_pdf / bsdf_sample, FrDielectric, FrSchlick, the GGX D / Lambda / G terms,
 * the visible-normal sample, the local frame, sin/cos(2 pi u), is_valid_color and clamp_radiance (reference
 * include/geometric.cuh:119-235, 419-562) exactly as the shading kernel calls them.  24 floats in and 40 floats out per
 * record; the layout is documented at k_probe_functions (path_tracing_amd/csrc/pt_kernels.hip) and mirrored by
 * oracle_function_kats (oracle/pt_oracle.cpp). */
int hpt_probe_functions(const float *records_in, int n, float *results_out);

/* ---- 8-bit output stage on the device -------------------------------------------------------------
 * Replaces the per-pixel loop of the reference CLI, src/main_cli.cpp:225-242: per channel clamp to [0, 1],
 * pow(x, 1/2.2), x 255, truncate.  The device looks the byte up in a table of 255 thresholds that the host
 * computes with its own powf, so the bytes are exactly those of the host loop (hpt_tonemap_reference runs
 * that loop; hpt_tonemap_table returns thresholds[k] = the smallest float whose byte is >= k, [0] = -inf).
 *   d_linear_rgb  3 floats per pixel (the image hpt_untile / hpt_render_* produce), device memory
 *   d_rgb8        3 bytes per pixel, device memory, 4-byte aligned
 *   bgr           non-zero: the reference's cv::Vec3b order (B, G, R); zero: R, G, B
 * hpt_tonemap_host takes and returns host buffers (upload, kernel, download). */
int hpt_tonemap(const void *d_linear_rgb, void *d_rgb8, int64_t num_pixels, int bgr, void *hip_stream);
int hpt_tonemap_host(const float *linear_rgb, unsigned char *rgb8, int64_t num_pixels, int bgr);
void hpt_tonemap_table(float thresholds_out[256]);
void hpt_tonemap_reference(const float *linear_rgb, unsigned char *rgb8, int64_t num_pixels, int bgr);

/* ---- the acceleration structure, exported (tests) -------------------------------------
 * The reference has no acceleration structure (include/geometric.cuh:293-388 scan every primitive); the tree the
 * kernels walk is this library's own, and the work counts the bench's roofline is built from (hpt_stats.boxes_*,
 * tris_*) are counts of walking it.  These two calls hand the tree out so that an independent walker -- the oracle's
 * host traversal, oracle/pt_oracle.cpp -- can reproduce those counts ray by ray.
 *   qnodes_out  num_nodes records of 32 B: six words of uint16 grid coordinates (coordinate = qorigin + q * qscale) --
 *               per axis x, y, z one word with the lower plane of the left child's box in its low half and of the right
 *               child's in its high half, then one word with the two upper planes the same way -- then the two uint32
 *               child codes: bit 31 set = leaf,
 *               (code & 0x7FFFFFFF) >> 3 = first triangle slot, (code & 7) + 1 = triangle count; 0xFFFFFFFF = no
 *               child; otherwise the index of an inner node.  Node 0 is the root.
 *   tris_out    num_tris records of 48 B in leaf order: v0 | scan ordinal, v1 - v0 | material, v2 - v0 | flags
 * Either output may be NULL (sizes only); the caps are in bytes and must cover what is written.
 * hpt_bvh_export_host builds on the host only (no device needed) exactly what hpt_scene_create would upload;
 * hpt_scene_export_bvh copies back what the scene's device actually holds. */
typedef struct hpt_bvh_info {
    int32_t num_nodes, num_tris, bvh_depth, num_rounds;   /* num_rounds = spheres + light balls: the first triangle ordinal */
    float qorigin[3], qscale[3];
} hpt_bvh_info;
int hpt_bvh_export_host(const void *lights, int num_lights, const void *spheres, int num_spheres,
                        const void *triangles, int num_triangles, hpt_bvh_info *info,
                        void *qnodes_out, size_t qnodes_cap, void *tris_out, size_t tris_cap);
int hpt_scene_export_bvh(const hpt_scene *scene, hpt_bvh_info *info,
                         void *qnodes_out, size_t qnodes_cap, void *tris_out, size_t tris_cap);

/* Ray-level probes of the intersection kernels (tests): n rays, origins/directions as
 * packed float3.  prim is the reference scan ordinal (spheres, then light balls, then
 * triangles in input order), -1 on a miss; t is 1e20f on a miss. */
int hpt_trace_closest(hpt_scene *scene, const float *origins, const float *dirs, int n,
                      int flags, float *t_out, int32_t *prim_out);
/* Shadow segments p1 -> p2 with the reference's (1e-3, dist-1e-3) range; visible_out[i] = 1
 * when no opaque primitive blocks the segment. */
int hpt_trace_visibility(hpt_scene *scene, const float *p1, const float *p2, int n,
                         int flags, int32_t *visible_out);

#ifdef __cplusplus
}
#endif
#endif /* HPT_H */
