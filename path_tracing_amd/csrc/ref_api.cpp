// libhpt_ref.so: C++-linkage adapters with the reference's own symbol names, forwarding to the
// C ABI (include/hpt.h).  Host-only translation unit (g++), no HIP headers: the by-value float3
// here must be CUDA's plain struct, not HIP's vector type, for the mangled name to match.
#include "../../include/hpt_reference_api.hpp"
#include "../../include/hpt.h"

#include <cstdio>
#include <cstdlib>

void pt_render_wrapper(
    const CudaLight *cuda_lights, int num_lights,
    const CudaSphere *cuda_spheres, int num_spheres,
    const CudaTriangle *cuda_triangles, int num_triangles,
    float3 scene_min, float3 scene_max,
    const CudaCamera cuda_camera, float3 *cuda_image, int W, int H,
    int light_depth, int light_sample, int eye_depth, int spp){
    long long seed = -1;                                   // clock, like the reference
    if(const char *e = getenv("HPT_SEED")){ char *endp = nullptr; long long v = strtoll(e, &endp, 10); if(endp != e && v >= 0) seed = v; }
    const float mn[3] = { scene_min.x, scene_min.y, scene_min.z }, mx[3] = { scene_max.x, scene_max.y, scene_max.z };
    int rc = hpt_pt_render_wrapper(cuda_lights, num_lights, cuda_spheres, num_spheres, cuda_triangles, num_triangles,
                                   mn, mx, &cuda_camera, &cuda_image->x, W, H, light_depth, light_sample, eye_depth, spp, seed);
    if(rc != HPT_OK) fprintf(stderr, "pt_render_wrapper: %s\n", hpt_last_error());
}

void bdpt_render_wrapper(
    const CudaLight *cuda_lights, int num_lights,
    const CudaSphere *cuda_spheres, int num_spheres,
    const CudaTriangle *cuda_triangles, int num_triangles,
    float3 scene_min, float3 scene_max,
    const CudaCamera cuda_camera, float3 *cuda_image, int W, int H,
    int light_depth, int light_sample, int eye_depth, int spp, int spl){
    long long seed = -1;
    if(const char *e = getenv("HPT_SEED")){ char *endp = nullptr; long long v = strtoll(e, &endp, 10); if(endp != e && v >= 0) seed = v; }
    const float mn[3] = { scene_min.x, scene_min.y, scene_min.z }, mx[3] = { scene_max.x, scene_max.y, scene_max.z };
    int rc = hpt_bdpt_render_wrapper(cuda_lights, num_lights, cuda_spheres, num_spheres, cuda_triangles, num_triangles,
                                     mn, mx, &cuda_camera, &cuda_image->x, W, H, light_depth, light_sample, eye_depth, spp, spl, seed);
    if(rc != HPT_OK) fprintf(stderr, "bdpt_render_wrapper: %s\n", hpt_last_error());
}
