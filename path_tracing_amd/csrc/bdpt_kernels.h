// Launch interface of the bidirectional (cpu_bdpt-estimator) kernels (bdpt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "hpt_scene.h"
#include "pt_kernels.h"

namespace hpt {

struct BdptSceneDev {
    const float4 *nodes;            // BvhNode as 4 x float4 (all groups)
    const float4 *tris;             // DevTriangle as 3 x float4; ordinal = CPU iteration order
    const DevRound *spheres;        // pad[0] = CPU iteration order
    const DevGroup *groups;
    const DevMaterial *mats;
    const DevLight *lights;
    int num_groups, num_lights, num_mats, stack_levels;   // stack_levels: deepest group tree (traversal stack entries per lane)
    float scene_min[3], scene_max[3];
};

// Light-subpath vertex (reference include/bdpt_cu.cuh:6-18, fields the CPU estimator uses): 80 B
struct LightVertexDev {
    float pos[3]; float pdf_fwd;
    float normal[3]; float pdf_rev;
    float thr[3]; float source_cutoff;
    float base[3]; float roughness;          // material at the vertex
    float metallic; float eta; uint32_t flags; uint32_t pad;    // flags: 1 is_light_source, 2 is_parallel
};
static_assert(sizeof(LightVertexDev) == 80, "layout");

// What a connection needs of a light vertex beyond the vertex itself, computed once per vertex instead of once per
// (eye vertex, light vertex) pair (k_bdpt_light_ctx): the shading frame of its normal with the direction towards the
// previous vertex in that frame (the BSDF value fL, src/cpu_bdpt.cpp:405-410) and the same for the re-normalised normal
// the MIS weight uses (cpu_calculate_mis_weight, src/cpu_bdpt.cpp:119-135), Lambda(wo) of both and the diffuse lobe.
// Same expressions as the per-pair code they replace, so the bits do not change.  112 B.
struct LightVertexCtx {
    float T[3], B[3], wo_l[3];          // frame of `normal`, direction to the previous vertex in it
    float nt[3], Tn[3], Bn[3], wo_t[3]; // normalize(normal), its frame, the MIS direction (the normal itself at the source) in it
    float lam_l, lam_t;                 // ggx_lambda of wo_l / wo_t
    float diffuse[3];                   // base / pi * (1 - metallic)
    float pad[2];
};
static_assert(sizeof(LightVertexCtx) == 112, "layout");

// Eye-path state beyond PathBuf, structure of arrays by path slot.
struct BdptPathBuf {
    float4 *last_pos_pdf;     // last vertex position xyz | last_pdf_omega
    float4 *last_normal;      // last vertex normal xyz | unused
    float4 *vtx_pos;          // current vertex: position xyz | material roughness
    float4 *vtx_nrm;          // normal xyz | material metallic
    float4 *vtx_thr;          // throughput at the vertex xyz | material eta
    float4 *vtx_wo;           // direction back along the eye ray xyz | as_float(depth)
    float4 *vtx_base;         // material base colour xyz | unused
    float4 *hist_pos_eta;     // [depth][slot]: vertex position xyz | material eta
    float2 *hist_pdf;         // [depth][slot]: pdf_fwd, pdf_rev (final values)
    float4 *contrib;          // [slot][light vertex]: clamped contribution xyz | unused; written for the pairs that pass the culls only
    unsigned long long *valid; // [slot][64-vertex chunk]: bit j = pair (slot, chunk * 64 + j) passed the culls and has a table entry
    float4 *ectx;             // [7][slot]: the eye vertex's shading contexts (k_bdpt_vertex): frame of the normal, wo in it,
                              // Lambda(wo), diffuse lobe | normalize(normal), its frame, the MIS direction in it, its Lambda
};

void launch_bdpt_light_trace(hipStream_t s, const BdptSceneDev &sc, LightVertexDev *lv, int total_paths, int light_depth,
                             int spl, uint64_t seed, int max_delta);
void launch_bdpt_generate(hipStream_t s, const Tiling &tl, const CameraDev &cam, PathBuf pb, BdptPathBuf bp, uint32_t *qcount,
                          int samples_this_pass, uint32_t first_sample, uint64_t seed);
// max_groups != 0 caps the grid of extend / connect / reduce (they walk their queues with a stride)
void launch_bdpt_extend(hipStream_t s, const BdptSceneDev &sc, PathBuf pb, const uint32_t *queue, const uint32_t *qcount,
                        uint32_t max_items, uint32_t max_groups = 0);
void launch_bdpt_light_ctx(hipStream_t s, const LightVertexDev *lv, LightVertexCtx *ctx, int n_lv, int light_depth);
void launch_bdpt_vertex(hipStream_t s, const BdptSceneDev &sc, PathBuf pb, BdptPathBuf bp, const uint32_t *queue,
                        const uint32_t *qcount, uint32_t max_items, uint32_t *next_queue, uint32_t *next_count,
                        uint32_t *cqueue, uint32_t *ccount, int eye_depth, int max_delta, uint32_t slots, const float eye[3]);
void launch_bdpt_connect(hipStream_t s, const BdptSceneDev &sc, PathBuf pb, BdptPathBuf bp, const LightVertexDev *lv,
                         const LightVertexCtx *lctx, int n_lv, int light_depth, const uint32_t *cqueue, const uint32_t *ccount,
                         uint32_t max_items, uint32_t slots, uint32_t max_groups = 0, WorkCounters *wc = nullptr);
void launch_bdpt_reduce(hipStream_t s, PathBuf pb, BdptPathBuf bp, int n_lv, const uint32_t *cqueue, const uint32_t *ccount,
                        uint32_t max_items, uint32_t max_groups = 0);

} // namespace hpt
