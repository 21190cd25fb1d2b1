// hpt_reference_api.hpp -- the reference renderer's own entry points, re-exported with C++
// linkage by libhpt_ref.so so that the reference's unmodified helper
// (reference src/pt_cu_helper.cpp:66-77, which calls pt_render_wrapper) links against this
// library instead of the reference's CUDA translation unit src/pt_cu.cu.
//
// Replaces:  void pt_render_wrapper(...)   reference include/pt_cu.cuh:6-13, defined src/pt_cu.cu:255-297
//
// The mangled name depends on the type NAMES (CudaLight, CudaSphere, CudaTriangle, CudaCamera,
// float3), and the by-value arguments on their field types, so the records are declared here
// with the reference's names and layouts (reference include/geometric.cuh:15-51,67-78; CUDA's
// float3 is a plain struct of three floats).  Do not include this header together with the
// reference's geometric.cuh or with HIP/CUDA vector-type headers: it is for translation units
// that need the declarations only (libhpt_ref.so itself, tests, new callers).
#ifndef HPT_REFERENCE_API_HPP
#define HPT_REFERENCE_API_HPP

struct float3 { float x, y, z; };

struct CudaMaterial_Old { float3 Kd, Kg, Ks; float glossy, exp, refract, reflect; };      // 52 B
struct CudaMaterial { float3 base_color; float roughness, metallic, eta; int type; };      // 28 B
struct CudaSphere { float3 center; float r; CudaMaterial_Old mtl_old; CudaMaterial mtl; int id; };        // 100 B
struct CudaTriangle { float3 v0, v1, v2; CudaMaterial_Old mtl_old; CudaMaterial mtl; int id; };           // 120 B
struct CudaCamera { float3 eye, U, V, W, UL, dx, dy; };                                                     // 84 B
struct CudaLight { float3 pos, dir, illum; CudaSphere light_ball; float cutoff; int is_parallel; };       // 144 B

static_assert(sizeof(float3) == 12 && sizeof(CudaSphere) == 100 && sizeof(CudaTriangle) == 120, "layout");
static_assert(sizeof(CudaLight) == 144 && sizeof(CudaCamera) == 84, "layout");

// Same contract as the reference: blocking; host arrays in, host image out (W*H float3, row-major,
// row 0 = top, mean linear radiance); scene_min/scene_max/light_depth/light_sample are ignored
// (src/pt_cu.cu:259-262); the random streams are seeded from the clock (src/pt_cu.cu:282) unless
// the environment variable HPT_SEED holds a non-negative integer.  Errors are printed to stderr
// and the image is left untouched (the reference checks nothing, src/pt_cu.cu:270-296).
void pt_render_wrapper(
    const CudaLight *cuda_lights, int num_lights,
    const CudaSphere *cuda_spheres, int num_spheres,
    const CudaTriangle *cuda_triangles, int num_triangles,
    float3 scene_min, float3 scene_max,
    const CudaCamera cuda_camera, float3 *cuda_image, int W, int H,
    int light_depth, int light_sample, int eye_depth, int spp);


// Replaces:  void bdpt_render_wrapper(...)   reference include/bdpt_cu.cuh:30-37, defined src/bdpt_cu.cu:538-674
// The lights arrive with illum / light_sample (reference src/bdpt_cu_helper.cpp:60-62).  The image is the
// estimator of the reference's CPU renderer run_cpu_bdpt (src/cpu_bdpt.cpp), which the CUDA kernel differs
// from in the ways listed in SURVEY.md Q19; one implicit group (no grouping crosses this signature).
void bdpt_render_wrapper(
    const CudaLight *cuda_lights, int num_lights,
    const CudaSphere *cuda_spheres, int num_spheres,
    const CudaTriangle *cuda_triangles, int num_triangles,
    float3 scene_min, float3 scene_max,
    const CudaCamera cuda_camera, float3 *cuda_image, int W, int H,
    int light_depth, int light_sample, int eye_depth, int spp, int spl /*sample per light*/);

#endif
