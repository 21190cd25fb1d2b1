// Host-side front-end of the PT / BDPT hot path, free of glm / OpenCV / CUDA headers so it builds on the
// MI355X box.  What it stands in for in the reference:
//
//   scene text grammar      reference src/main_cli.cpp:99-141 -- here a buffered tokenizer over
//                           the mapped file (text_cursor.hpp), 10^6 'T' lines in a fraction of a second
//   OBJ ingestion           the role the reference gave include/tiny_obj_loader.h:607-628 (compiled, never called)
//   object model            reference include/object.h:28-33,40-80,94-110 -- kept as flat arrays; the reference-shaped
//                           std::map<int, AABB> view is built on demand for the helper API
//   camera                  reference src/main_cli.cpp:25-40,155-166
//   helper API              reference include/pt_cu_helper.h:5-6, include/bdpt_cu_helper.h:5-6
//   output stage            reference src/main_cli.cpp:223-254: tone-map on the device (hpt_tonemap), PNG through
//                           zlib, PFM for the linear image
//
// The device records are the ones of include/hpt_reference_api.hpp (CudaLight, CudaSphere, ...).
#pragma once
#include "../../../include/hpt_reference_api.hpp"

#include <cstdint>
#include <map>
#include <memory>
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

// ---- the reference-shaped object model (what move_data_to_cuda_* take) ------------------------------
class Object {                                                                        // include/object.h:40-52
public:
    enum Kind { kSphere = 0, kTriangle = 1 };
    Material mtl;
    int obj_id = 0;
    virtual ~Object() = default;
    virtual Kind kind() const = 0;
};
class Sphere : public Object { public: float r = 1.0f; vec3 center; Kind kind() const override { return kSphere; } };
class Triangle : public Object { public: vec3 vert[3]; Kind kind() const override { return kTriangle; } };

class AABB {                                                                          // include/object.h:94-102
public:
    vec3 min{99999.f, 99999.f, 99999.f}, max{-99999.f, -99999.f, -99999.f};
    std::vector<Object *> objs;
    void add_obj(Object *obj);      // grows the box (sphere: centre +- r, triangle: its vertices) and appends
};

struct Camera { vec3 eye, look_at, view_up; float fov = 50.0f; };                     // include/object.h:104-110

// ---- a parsed scene: flat storage in file order --------------------------------------------------------
struct SceneItem {            // one 'S' or 'T' line (or OBJ face), in the order read
    uint8_t kind;             // Object::Kind
    int32_t group;            // 'G' in force
    uint32_t material;        // index into SceneFile::materials ('M' in force)
    uint32_t geom;            // index into sphere_geom (4 floats each) or tri_geom (9 floats each)
    int32_t obj_id;           // running id, as the reference numbers its objects
};

struct SceneFile {
    Camera camera;
    std::pair<int, int> resolution{200, 200};
    std::vector<Material> materials;        // [0] = the all-zero material in force before the first 'M'
    std::vector<SceneItem> items;
    std::vector<float> sphere_geom;         // cx cy cz r
    std::vector<float> tri_geom;            // x0 y0 z0 x1 y1 z1 x2 y2 z2
    std::vector<CudaLight> lights;          // as parsed: dir not normalised, cutoff in radians
    int tri_cnt = 0, ball_cnt = 0;
    double parse_ms = 0.0;                  // wall time of the last parse_scene_file / append_obj

    SceneFile();
    // the reference-shaped view: groups in id order, objects in file order inside a group.  Built on first use and
    // owned by this SceneFile (the reference leaks its objects).
    std::map<int, AABB> &groups();
private:
    std::map<int, AABB> groups_;
    std::vector<std::unique_ptr<Object>> pool_;
    size_t grouped_items_ = 0;
};

// Scene text (grammar: SURVEY Appendix A).  parse_scene_file returns false only if the file cannot be opened.
void parse_scene_text(const char *begin, const char *end, SceneFile &out);
bool parse_scene_file(const std::string &path, SceneFile &out);
// Wavefront OBJ: 'v x y z' and 'f a b c ...' (a, a/t, a//n, a/t/n; negative = relative; polygons fanned from their
// first vertex); everything else is skipped.  Faces become triangles of `mtl` in group `group_id`.  Returns the number
// of triangles added, -1 with *err set on an unreadable file or a bad index.
int append_obj(const std::string &path, const Material &mtl, int group_id, SceneFile &scene, std::string *err);

// boundary records of a scene: groups in id order, file order inside a group, spheres and triangles split into their
// arrays (what move_data_to_cuda_pt does through the object model, reference src/pt_cu_helper.cpp:12-64), light
// directions normalised (normalize_cuda, src/geometric.cu:54-57); `order` receives kind / index / group per object
struct FlatScene {
    std::vector<CudaSphere> spheres;
    std::vector<CudaTriangle> triangles;
    std::vector<CudaLight> lights;
    std::vector<int32_t> kind, index, group;
};
void flatten_scene(const SceneFile &scene, FlatScene &out);

CudaCamera make_cuda_camera(const Camera &camera, float fov_deg, int W, int H);     // src/main_cli.cpp:25-40,155-166

// 8-bit output stage (src/main_cli.cpp:223-254): tone-map on the device, rows top to bottom; ".png" through zlib,
// ".pfm" writes the linear float image instead (bottom-up rows, as the format wants)
bool write_image(const std::string &path, const float3 *linear_rgb, int W, int H, std::string *err);
bool write_png_rgb8(const std::string &path, const unsigned char *rgb, int W, int H, std::string *err);

} // namespace hpt_host

// The reference's helper API, same names and argument meaning (include/pt_cu_helper.h:5-6).  Kept different on
// purpose: move_data_to_cuda_pt REPLACES the previously moved scene (the reference's vectors are never cleared, SURVEY
// Q17); the seed comes from hpt_host::g_run_params unless g_seed_from_clock.
void move_data_to_cuda_pt(std::map<int, hpt_host::AABB> groups, std::vector<CudaLight> &cuda_lights, int light_sample);
void run_cuda_pt(CudaCamera cam, float3 *image_buffer, int light_depth, int eye_depth, int W, int H, int spp);
// BDPT twins (include/bdpt_cu_helper.h:5-6); the grouping is handed to the device scene, so the result is
// run_cpu_bdpt's estimator on the same groups.
void move_data_to_cuda_bdpt(std::map<int, hpt_host::AABB> groups, std::vector<CudaLight> &cuda_lights, int light_sample);
void run_cuda_bdpt(CudaCamera cam, float3 *image_buffer, int light_depth, int eye_depth, int W, int H, int spp, int spl = 1);
