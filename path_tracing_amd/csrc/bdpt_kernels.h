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
    int num_groups, num_lights, num_mats, pad;
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
    float4 *contrib;          // [slot][light vertex]: clamped contribution xyz | unused
};

void launch_bdpt_light_trace(hipStream_t s, const BdptSceneDev &sc, LightVertexDev *lv, int total_paths, int light_depth,
                             int spl, uint64_t seed, int max_delta);
void launch_bdpt_generate(hipStream_t s, const Tiling &tl, const CameraDev &cam, PathBuf pb, BdptPathBuf bp, uint32_t *qcount,
                          int samples_this_pass, uint32_t first_sample, uint64_t seed);
void launch_bdpt_extend(hipStream_t s, const BdptSceneDev &sc, PathBuf pb, const uint32_t *queue, const uint32_t *qcount,
                        uint32_t max_items);
void launch_bdpt_vertex(hipStream_t s, const BdptSceneDev &sc, PathBuf pb, BdptPathBuf bp, const uint32_t *queue,
                        const uint32_t *qcount, uint32_t max_items, uint32_t *next_queue, uint32_t *next_count,
                        uint32_t *cqueue, uint32_t *ccount, int eye_depth, int max_delta, uint32_t slots);
void launch_bdpt_connect(hipStream_t s, const BdptSceneDev &sc, PathBuf pb, BdptPathBuf bp, const LightVertexDev *lv,
                         int n_lv, int light_depth, const uint32_t *cqueue, const uint32_t *ccount, uint32_t max_items,
                         const float eye[3], uint32_t slots);
void launch_bdpt_reduce(hipStream_t s, PathBuf pb, BdptPathBuf bp, int n_lv, const uint32_t *cqueue, const uint32_t *ccount,
                        uint32_t max_items);

} // namespace hpt
