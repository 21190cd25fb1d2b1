// Launch interface between the C ABI (hpt_api.cpp) and the HIP kernels (pt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "hpt_scene.h"

namespace hpt {

struct SceneDev {
    const float4 *nodes;        // BvhNode as 4 x float4
    const uint4 *qnodes;        // QBvhNode as 2 x uint4 (k_trace)
    const uint4 *wnodes;        // WideNode as 4 x uint4 (development: four-wide resume launch)
    int wide_depth;
    float qorigin[3], qscale[3];
    const float4 *tris;         // DevTriangle as 3 x float4, leaf order
    const float4 *tri_frames;   // per triangle 4 x float4: shading normal and both local frames (launch_tri_frames)
    const DevRound *rounds;     // spheres then light balls
    const DevMaterial *mats;
    const DevLight *lights;
    int num_rounds, num_spheres, num_lights, num_tris, num_mats, num_nodes;
};

// Path state, structure of arrays indexed by path slot (one slot per (pixel, sample) in flight).
struct PathBuf {
    float4 *org_eta;     // ray origin xyz | current medium eta
    float4 *dir_flags;   // ray direction xyz | bit 0 last_is_delta, bits 8-15 depth, bits 16-23 delta count
    float4 *thr;         // throughput xyz | unused
    float4 *col;         // radiance accumulated by this sample xyz | unused
    uint2 *rng;          // PCG32 state
    uint2 *hit;          // extend result: as_uint(t) | primitive code
};

// Pending shadow ray of a path, structure of arrays indexed by path slot (the shadow queue
// itself is a compacted list of path slots).
struct ShadowBuf {
    float4 *org_max;     // p1 xyz | max_d
    float4 *dir;         // unit direction xyz | unused
    float4 *contrib;     // clamped contribution if unoccluded xyz | unused
};

// Image tiling shared by ray generation, resolve and untile (see DESIGN.md "Tiling").
struct Tiling {
    int W, H, tile, tiles_x, tiles_y, ntiles, rank, world;
    int n_local;         // packed local framebuffer slots = ceil(ntiles / world) * tile * tile
};

struct CameraDev { float eye[3], UL[3], dx[3], dy[3]; };

// What it takes to recompute the primary ray of a path slot (reference src/pt_cu.cu:36-46): handed to the PRIMARY
// variants of k_trace / k_shade, which run iteration 0 of a pass without a generate launch.
struct PrimaryGen { Tiling tl; CameraDev cam; uint32_t first_sample; uint32_t pad; uint64_t seed; };

struct WorkCounters {    // device counters, COUNT_WORK only
    unsigned long long boxes_closest, tris_closest, boxes_shadow, tris_shadow, closest_rays, shadow_rays, path_iters, samples,
                       lane_steps_closest, wave_steps_closest, lane_steps_shadow, wave_steps_shadow,
                       leaf_lane_closest, leaf_wave_closest, leaf_lane_shadow, leaf_wave_shadow,
                       // connection stage of the bidirectional path (k_bdpt_connect): candidate (eye vertex, light vertex) pairs, pairs that
                       // pass the culls, shadow rays traced, unoccluded ones; node visits, triangle, sphere and group-box tests of those rays
                       bd_pairs, bd_survivors, bd_shadow_rays, bd_unoccluded, bd_nodes, bd_tris, bd_spheres, bd_group_boxes;
};

// primitive code in PathBuf::hit.y
constexpr uint32_t kHitMiss = 0xFFFFFFFFu;
constexpr uint32_t kHitRoundFlag = 0x80000000u;   // | index into rounds; else triangle slot

constexpr int kBlock = 256;
// k_trace tuning for launches whose rays are all long (the resume launch of a split step; every launch of a
// scene whose split is held off): rays per workgroup, idle lanes that trigger a refill, early-leaf-break threshold
constexpr uint32_t kLongChunk = 4096;       // with two passes in flight: 2048 -> 4096 = -3.0 % / -2.1 % on 100 k / 20 k random triangles, -0.5 % on the sphere scenes
                                            // (one pass in flight: +2 % on random triangles: a launch's tail is then nobody's to fill)
constexpr int kLongRefillMin = 16;
constexpr int kLongNodeMin = 8;
constexpr int kResumeLdsLevels = 12; // stack levels of the resume launch kept in LDS when a deep-stack buffer is given (deeper: global memory)
constexpr uint32_t kResumeMaxGroups = 8192;   // grid cap of the resume launch
constexpr int kDeepLevels = 48;      // global-memory stack levels per lane of the resume launch (the four-wide walk stacks up to three children per step)
constexpr int kWideRefillMin = 24;   // the same two thresholds for the four-wide resume launch (A/B grid 16/24/32 x 12/16/24 on configs 3 and 5)
constexpr int kWideNodeMin = 16;
#ifndef HPT_TRACE_BUDGET             // development A/B: `make variant EXTRA="-DHPT_TRACE_BUDGET=7 -DHPT_TOP_LEVELS=7"`
#define HPT_TRACE_BUDGET 6
#endif
#ifndef HPT_TOP_LEVELS
#define HPT_TOP_LEVELS 6
#endif
constexpr int kTraceBudget = HPT_TRACE_BUDGET;   // node steps a ray gets in the first trace launch before it is set aside
constexpr int kTopLevels = HPT_TOP_LEVELS;       // a budget of at most this many steps keeps a ray among the first kTopNodes nodes
constexpr int kTopNodes = 1 << kTopLevels;       // (breadth-first order, scene_build.cpp): 2^kTopLevels - 1 = 63 nodes of 32 B, staged in LDS

void launch_generate(hipStream_t s, const Tiling &tl, const CameraDev &cam, PathBuf pb,
                     uint32_t *qcount, int samples_this_pass, uint32_t first_sample, uint64_t seed,
                     WorkCounters *wc);
void launch_extend(hipStream_t s, const SceneDev &sc, PathBuf pb, const uint32_t *queue, const uint32_t *qcount,
                   uint32_t max_items, int flags, WorkCounters *wc);
void launch_shade(hipStream_t s, const SceneDev &sc, PathBuf pb, const uint32_t *queue, const uint32_t *qcount,
                  uint32_t max_items, uint32_t *next_queue, uint32_t *next_count, ShadowBuf sb, uint32_t *squeue,
                  uint32_t *scount, int max_depth, int max_delta, int roulette, WorkCounters *wc, const PrimaryGen *primary = nullptr,
                  uint32_t max_groups = 0);        // max_groups != 0 caps the grid (the kernels walk their chunks with a stride)
void launch_connect(hipStream_t s, const SceneDev &sc, PathBuf pb, ShadowBuf sb, const uint32_t *squeue,
                    const uint32_t *scount, uint32_t max_items, int flags, WorkCounters *wc);
// merged closest-hit (equeue) + any-hit (squeue) launch; either queue may be absent (null count)
// split != null && split->budget > 0: rays needing more than `budget` node steps are set aside into the
// long queues (their counters must be zero on entry) and finished by a second launch
struct TraceSplit { uint32_t *equeue, *ecount, *squeue, *scount; int budget; };
void launch_trace(hipStream_t s, const SceneDev &sc, PathBuf pb, ShadowBuf sb, const uint32_t *equeue,
                  const uint32_t *ecount, uint32_t max_extend, const uint32_t *squeue, const uint32_t *scount,
                  uint32_t max_shadow, int stack_levels, int flags, int tuning, WorkCounters *wc,
                  const TraceSplit *split = nullptr, const PrimaryGen *primary = nullptr, uint32_t max_groups = 0);
void launch_trace_resume(hipStream_t s, const SceneDev &sc, PathBuf pb, ShadowBuf sb, bool extend, bool shadow,
                         uint32_t max_items, int stack_levels, WorkCounters *wc, const TraceSplit &split,
                         const PrimaryGen *primary = nullptr, uint32_t max_groups = 0, uint32_t *deep_stack = nullptr,
                         bool tiny_lds_share = false, bool wide = false, int dev_tuning = 0);
// words of the buffer launch_trace_resume's deep_stack needs (one column per lane of its largest grid)
size_t resume_deep_stack_words();
// fills frames[4 * num_tris] from tris (once per scene)
void launch_tri_frames(hipStream_t s, const float4 *tris, int num_tris, float4 *frames);
void launch_resolve(hipStream_t s, const Tiling &tl, PathBuf pb, float4 *accum, int samples_this_pass);
void launch_finalize(hipStream_t s, const Tiling &tl, const float4 *accum, float *d_local, float scale);
void launch_untile(hipStream_t s, const Tiling &tl, const float *d_gathered, float *d_image);

// 8-bit output stage: num_values = 3 * pixels floats in, as many bytes out (d_bytes 4-byte aligned); d_thresholds: 256 floats
void launch_tonemap(hipStream_t s, const float *d_linear, void *d_bytes, unsigned long long num_values, int bgr, const float *d_thresholds);

// function-level probe (tests): 24 floats in, 40 floats out per record (layout: k_probe_functions)
void launch_probe_functions(hipStream_t s, const float *d_in, int n, float *d_out);
// ray-level probes (tests)
void launch_probe_closest(hipStream_t s, const SceneDev &sc, const float *org, const float *dir, int n, int flags,
                          float *t_out, int32_t *prim_out);
void launch_probe_visibility(hipStream_t s, const SceneDev &sc, const float *p1, const float *p2, int n, int flags,
                             int32_t *vis_out);

} // namespace hpt
