// Host-side mirror of the reference's front-end for the PT path, free of glm / OpenCV / CUDA
// headers so it builds on the MI355X box: scene model, scene-file parser, camera set-up,
// flattening helper and launch helper with the reference's own names.
//
//   Material / Object / Sphere / Triangle / AABB / Camera   reference include/object.h:28-33,40-80,94-110
//   parse_scene                                             reference src/main_cli.cpp:99-141 (grammar: SURVEY Appendix A)
//   init_camera                                             reference src/main_cli.cpp:25-40
//   move_data_to_cuda_pt / run_cuda_pt                      reference include/pt_cu_helper.h:5-6, src/pt_cu_helper.cpp:12-77
//
// The device records are the ones of include/hpt_reference_api.hpp (CudaLight, CudaSphere, ...).
#pragma once
#include "../../../include/hpt_reference_api.hpp"

#include <iosfwd>
#include <map>
#include <string>
#include <vector>

namespace hpt_host {

struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float &operator[](int i){ return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};

struct Material { vec3 base_color; float roughness = 0, metallic = 0, eta = 0; };    // include/object.h:28-33

class Object {                                                                        // include/object.h:40-52
public:
    Material mtl;
    int obj_id = 0;
    virtual ~Object() = default;
};
class Sphere : public Object { public: float r = 1.0f; vec3 center; vec3 scale{1, 1, 1}; };
class Triangle : public Object { public: vec3 vert[3]; };

class AABB {                                                                          // include/object.h:94-102
public:
    vec3 min{99999.f, 99999.f, 99999.f}, max{-99999.f, -99999.f, -99999.f};
    std::vector<Object *> objs;
    void add_obj(Object *obj);                                                        // src/object.cpp:123-146
};

struct Camera { vec3 eye, look_at, view_up; float fov = 50.0f; };                     // include/object.h:104-110

struct SceneFile {
    Camera camera;
    std::pair<int, int> resolution{200, 200};
    std::map<int, AABB> groups;
    std::vector<CudaLight> lights;
    int tri_cnt = 0, ball_cnt = 0;
    std::vector<Object *> owned;          // freed by the destructor (the reference leaks them)
    SceneFile() = default;
    SceneFile(const SceneFile &) = delete;
    SceneFile &operator=(const SceneFile &) = delete;
    ~SceneFile();
};

// Token-by-token grammar of the reference CLI; returns false if the file cannot be opened.
bool parse_scene(std::istream &input, SceneFile &out);
bool parse_scene_file(const std::string &path, SceneFile &out);
// Minimal Wavefront OBJ reader ('v' and 'f' with fan triangulation): appends the faces as
// triangles of `mtl` to group `group_id` (the role the reference gave tiny_obj_loader, which it
// compiles but never calls: src/tiny_obj_loader.cpp:1-2).  Returns the number of triangles.
int append_obj(const std::string &path, const Material &mtl, int group_id, SceneFile &scene, std::string *err);

void init_camera(const Camera &camera, float F, int W, int H, vec3 &UL, vec3 &dx, vec3 &dy);
CudaCamera make_cuda_camera(const Camera &camera, float F, int W, int H);            // src/main_cli.cpp:155-166

// 8-bit output stage of the reference CLI (src/main_cli.cpp:223-254): clamp [0,1], pow 1/2.2,
// x255 truncated; rows top to bottom.  PNG is written with zlib only (no OpenCV); ".pfm" writes
// the linear float image instead.
bool write_image(const std::string &path, const float3 *linear_rgb, int W, int H, std::string *err);

} // namespace hpt_host

// The reference's helper API, same names and argument meaning (include/pt_cu_helper.h:5-6).
// Differences kept deliberately: move_data_to_cuda_pt REPLACES the previously moved scene
// instead of appending to it (the reference's pt_ns vectors are never cleared).
void move_data_to_cuda_pt(std::map<int, hpt_host::AABB> groups, std::vector<CudaLight> &cuda_lights, int light_sample);
void run_cuda_pt(CudaCamera cam, float3 *image_buffer, int light_depth, int eye_depth, int W, int H, int spp);
// BDPT twins (include/bdpt_cu_helper.h:5-6, src/bdpt_cu_helper.cpp:13-81); the scene file's grouping is kept
// and handed to the device scene, so the result is run_cpu_bdpt's estimator on the same groups.
void move_data_to_cuda_bdpt(std::map<int, hpt_host::AABB> groups, std::vector<CudaLight> &cuda_lights, int light_sample);
void run_cuda_bdpt(CudaCamera cam, float3 *image_buffer, int light_depth, int eye_depth, int W, int H, int spp, int spl = 1);
