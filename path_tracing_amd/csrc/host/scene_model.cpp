#include "scene_model.hpp"
#include "../../../include/hpt.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <zlib.h>

namespace hpt_host {

namespace {
constexpr float kPi = 3.14159265358979323846f;
inline vec3 operator+(vec3 a, vec3 b){ return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b){ return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(float s, vec3 a){ return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator/(vec3 a, float s){ return {a.x / s, a.y / s, a.z / s}; }
inline float dot(vec3 a, vec3 b){ return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b){ return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline vec3 normalize(vec3 a){ float inv = 1.0f / std::sqrt(dot(a, a)); return inv * a; }   // glm: v * inversesqrt(dot)
inline float3 to_f3(vec3 v){ return float3{v.x, v.y, v.z}; }
std::istream &operator>>(std::istream &is, vec3 &v){ return is >> v.x >> v.y >> v.z; }
std::istream &operator>>(std::istream &is, float3 &v){ return is >> v.x >> v.y >> v.z; }
}

void AABB::add_obj(Object *obj){
    if(const Sphere *s = dynamic_cast<const Sphere *>(obj)){
        for(int a = 0; a < 3; ++a){
            min[a] = std::min({min[a], s->center[a] + s->r, s->center[a] - s->r});
            max[a] = std::max({max[a], s->center[a] + s->r, s->center[a] - s->r});
        }
    } else if(const Triangle *t = dynamic_cast<const Triangle *>(obj)){
        for(int i = 0; i < 3; ++i) for(int a = 0; a < 3; ++a){
            min[a] = std::min(min[a], t->vert[i][a]);
            max[a] = std::max(max[a], t->vert[i][a]);
        }
    }
    objs.push_back(obj);
}

SceneFile::~SceneFile(){ for(Object *o : owned) delete o; }

bool parse_scene(std::istream &input, SceneFile &sc){
    char t;
    Material mtl;
    int group_id = 0, obj_id = 0;
    while(input >> t){
        if(t == 'E'){ input >> sc.camera.eye; }
        else if(t == 'V'){ input >> sc.camera.look_at >> sc.camera.view_up; }
        else if(t == 'F'){ input >> sc.camera.fov; }
        else if(t == 'R'){ input >> sc.resolution.first >> sc.resolution.second; }
        else if(t == 'S'){
            Sphere *s = new Sphere;
            input >> s->center >> s->r;
            s->mtl = mtl; s->obj_id = obj_id++;
            sc.owned.push_back(s);
            sc.groups[group_id].add_obj(s);
            sc.ball_cnt++;
        }
        else if(t == 'T'){
            Triangle *tri = new Triangle;
            for(int i = 0; i < 3; i++) input >> tri->vert[i];
            tri->mtl = mtl; tri->obj_id = obj_id++;
            sc.owned.push_back(tri);
            sc.groups[group_id].add_obj(tri);
            sc.tri_cnt++;
        }
        else if(t == 'M'){ input >> mtl.base_color >> mtl.roughness >> mtl.metallic >> mtl.eta; }
        else if(t == 'G'){ input >> group_id; }
        else if(t == '/'){
            input >> t;
            if(t == '/'){ std::string trash; std::getline(input, trash); continue; }
        }
        else if(t == 'L'){
            CudaLight light; memset(&light, 0, sizeof light);
            float cutoff_deg = 0;
            input >> light.pos >> light.dir >> light.illum >> cutoff_deg;
            light.cutoff = cutoff_deg * 0.01745329251994329576923690768489f;          // glm::radians
            input >> light.is_parallel >> light.light_ball.r;
            light.light_ball.center = light.pos;
            light.light_ball.mtl_old.Kd = light.illum;
            sc.lights.push_back(light);
        }
    }
    return true;
}

bool parse_scene_file(const std::string &path, SceneFile &out){
    std::ifstream f(path);
    if(!f.is_open()) return false;
    return parse_scene(f, out);
}

int append_obj(const std::string &path, const Material &mtl, int group_id, SceneFile &scene, std::string *err){
    std::ifstream f(path);
    if(!f.is_open()){ if(err) *err = "cannot open " + path; return -1; }
    std::vector<vec3> verts;
    int added = 0, next_id = 0;
    for(auto &g : scene.groups) for(Object *o : g.second.objs) next_id = std::max(next_id, o->obj_id + 1);
    std::string line;
    while(std::getline(f, line)){
        std::istringstream ls(line);
        std::string tag;
        if(!(ls >> tag)) continue;
        if(tag == "v"){ vec3 v; ls >> v.x >> v.y >> v.z; verts.push_back(v); }
        else if(tag == "f"){
            std::vector<int> idx; std::string tok;
            while(ls >> tok){
                int i = atoi(tok.c_str());                       // "i", "i/j", "i//k", "i/j/k": the vertex index leads
                if(i < 0) i = (int) verts.size() + i + 1;
                if(i < 1 || i > (int) verts.size()){ if(err) *err = "face index out of range in " + path; return -1; }
                idx.push_back(i - 1);
            }
            for(size_t k = 1; k + 1 < idx.size(); ++k){
                Triangle *tri = new Triangle;
                tri->vert[0] = verts[idx[0]]; tri->vert[1] = verts[idx[k]]; tri->vert[2] = verts[idx[k + 1]];
                tri->mtl = mtl; tri->obj_id = next_id++;
                scene.owned.push_back(tri);
                scene.groups[group_id].add_obj(tri);
                scene.tri_cnt++; added++;
            }
        }
    }
    return added;
}

void init_camera(const Camera &camera, float F, int W, int H, vec3 &UL, vec3 &dx, vec3 &dy){
    float aspect = float(W) / float(H);
    float theta = F * kPi / 180.0f;
    float half_height = (float) std::tan((double) (theta / 2));      // main_cli.cpp:30 calls ::tan on a float
    float half_width = aspect * half_height;
    vec3 w = normalize(camera.eye - camera.look_at);
    vec3 u = normalize(cross(camera.view_up, w));
    vec3 v = cross(w, u);
    UL = camera.eye - half_width * u + half_height * v - w;
    dx = ((2 * half_width) * u) / float(W);
    dy = ((-2 * half_height) * v) / float(H);
}

CudaCamera make_cuda_camera(const Camera &camera, float F, int W, int H){
    vec3 UL, dx, dy;
    init_camera(camera, F, W, H, UL, dx, dy);
    CudaCamera cam; memset(&cam, 0, sizeof cam);
    cam.eye = to_f3(camera.eye); cam.UL = to_f3(UL); cam.dx = to_f3(dx); cam.dy = to_f3(dy);
    return cam;
}

namespace {
uint32_t crc_of(const unsigned char *p, size_t n, uint32_t c){ return (uint32_t) crc32(c, p, (uInt) n); }
void put32(std::vector<unsigned char> &v, uint32_t x){ v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void chunk(std::vector<unsigned char> &out, const char tag[4], const std::vector<unsigned char> &data){
    put32(out, (uint32_t) data.size());
    size_t start = out.size();
    out.insert(out.end(), tag, tag + 4);
    out.insert(out.end(), data.begin(), data.end());
    put32(out, crc_of(out.data() + start, out.size() - start, 0));
}
}

bool write_image(const std::string &path, const float3 *img, int W, int H, std::string *err){
    bool pfm = path.size() > 4 && path.substr(path.size() - 4) == ".pfm";
    FILE *f = fopen(path.c_str(), "wb");
    if(!f){ if(err) *err = "cannot open " + path; return false; }
    if(pfm){
        fprintf(f, "PF\n%d %d\n-1.0\n", W, H);
        for(int j = H - 1; j >= 0; --j) fwrite(&img[(size_t) j * W], sizeof(float3), W, f);      // PFM rows go bottom-up
        fclose(f);
        return true;
    }
    std::vector<unsigned char> raw((size_t) H * (1 + 3 * (size_t) W));
    for(int j = 0; j < H; ++j){
        unsigned char *row = &raw[(size_t) j * (1 + 3 * (size_t) W)];
        row[0] = 0;
        for(int i = 0; i < W; ++i){
            const float3 &p = img[(size_t) j * W + i];
            float r = std::pow(std::max(0.0f, std::min(p.x, 1.0f)), 1.0f / 2.2f);
            float g = std::pow(std::max(0.0f, std::min(p.y, 1.0f)), 1.0f / 2.2f);
            float b = std::pow(std::max(0.0f, std::min(p.z, 1.0f)), 1.0f / 2.2f);
            row[1 + 3 * i + 0] = (unsigned char) (r * 255.0f);
            row[1 + 3 * i + 1] = (unsigned char) (g * 255.0f);
            row[1 + 3 * i + 2] = (unsigned char) (b * 255.0f);
        }
    }
    uLongf clen = compressBound((uLong) raw.size());
    std::vector<unsigned char> comp(clen);
    if(compress2(comp.data(), &clen, raw.data(), (uLong) raw.size(), 6) != Z_OK){ fclose(f); if(err) *err = "zlib failure"; return false; }
    comp.resize(clen);
    std::vector<unsigned char> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<unsigned char> ihdr;
    put32(ihdr, (uint32_t) W); put32(ihdr, (uint32_t) H);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", comp);
    chunk(out, "IEND", {});
    fwrite(out.data(), 1, out.size(), f);
    fclose(f);
    return true;
}

} // namespace hpt_host

// ---- the reference's helper API --------------------------------------------------------------
namespace pt_ns {
std::vector<CudaSphere> cuda_spheres;
std::vector<CudaTriangle> cuda_triangles;
std::vector<CudaLight> cuda_lights;
float3 scene_max_bound = {-1e9f, -1e9f, -1e9f};
float3 scene_min_bound = {1e9f, 1e9f, 1e9f};
int light_sample = 0;
hpt_scene *scene = nullptr;            // device-resident copy, rebuilt by move_data_to_cuda_pt
}

static CudaMaterial to_cmtl(const hpt_host::Material &m){                              // src/geometric.cu:31-52
    CudaMaterial cm;
    cm.base_color = float3{m.base_color.x, m.base_color.y, m.base_color.z};
    cm.roughness = m.roughness; cm.metallic = m.metallic; cm.eta = m.eta;
    cm.type = cm.eta > 0.0f ? 1 : (cm.metallic > 0.0f ? 2 : 3);
    return cm;
}

void move_data_to_cuda_pt(std::map<int, hpt_host::AABB> groups, std::vector<CudaLight> &lights, int light_sample){
    using namespace hpt_host;
    pt_ns::cuda_spheres.clear(); pt_ns::cuda_triangles.clear(); pt_ns::cuda_lights.clear();
    pt_ns::scene_max_bound = {-1e9f, -1e9f, -1e9f}; pt_ns::scene_min_bound = {1e9f, 1e9f, 1e9f};
    auto grow = [](float x, float y, float z){
        pt_ns::scene_max_bound.x = std::max(pt_ns::scene_max_bound.x, x); pt_ns::scene_min_bound.x = std::min(pt_ns::scene_min_bound.x, x);
        pt_ns::scene_max_bound.y = std::max(pt_ns::scene_max_bound.y, y); pt_ns::scene_min_bound.y = std::min(pt_ns::scene_min_bound.y, y);
        pt_ns::scene_max_bound.z = std::max(pt_ns::scene_max_bound.z, z); pt_ns::scene_min_bound.z = std::min(pt_ns::scene_min_bound.z, z);
    };
    for(auto &g : groups){
        for(Object *obj : g.second.objs){
            if(const Sphere *sph = dynamic_cast<const Sphere *>(obj)){
                CudaSphere c; memset(&c, 0, sizeof c);
                c.center = float3{sph->center.x, sph->center.y, sph->center.z}; c.r = sph->r;
                c.mtl = to_cmtl(sph->mtl); c.id = sph->obj_id;
                pt_ns::cuda_spheres.push_back(c);
                grow(sph->center.x + sph->r, sph->center.y + sph->r, sph->center.z + sph->r);
                grow(sph->center.x - sph->r, sph->center.y - sph->r, sph->center.z - sph->r);
            } else if(const Triangle *tri = dynamic_cast<const Triangle *>(obj)){
                CudaTriangle c; memset(&c, 0, sizeof c);
                c.v0 = float3{tri->vert[0].x, tri->vert[0].y, tri->vert[0].z};
                c.v1 = float3{tri->vert[1].x, tri->vert[1].y, tri->vert[1].z};
                c.v2 = float3{tri->vert[2].x, tri->vert[2].y, tri->vert[2].z};
                c.mtl = to_cmtl(tri->mtl); c.id = tri->obj_id;
                pt_ns::cuda_triangles.push_back(c);
                for(int i = 0; i < 3; ++i) grow(tri->vert[i].x, tri->vert[i].y, tri->vert[i].z);
            }
        }
    }
    for(auto l : lights){
        float len = std::sqrt(l.dir.x * l.dir.x + l.dir.y * l.dir.y + l.dir.z * l.dir.z);     // normalize_cuda, src/geometric.cu:54-57
        l.dir = float3{l.dir.x / len, l.dir.y / len, l.dir.z / len};
        pt_ns::cuda_lights.push_back(l);
    }
    pt_ns::light_sample = light_sample;
    if(pt_ns::scene){ hpt_scene_destroy(pt_ns::scene); pt_ns::scene = nullptr; }
    int rc = hpt_scene_create(pt_ns::cuda_lights.data(), (int) pt_ns::cuda_lights.size(),
                              pt_ns::cuda_spheres.data(), (int) pt_ns::cuda_spheres.size(),
                              pt_ns::cuda_triangles.data(), (int) pt_ns::cuda_triangles.size(), &pt_ns::scene);
    if(rc != HPT_OK){ std::cerr << "move_data_to_cuda_pt: " << hpt_last_error() << std::endl; pt_ns::scene = nullptr; return; }
    std::cout << "moved" << std::endl;
}

namespace hpt_host { hpt_params g_run_params = {1, 0, 0, 0, 0, 0, 0, 0, 0}; bool g_seed_from_clock = true; }

void run_cuda_pt(CudaCamera cam, float3 *image_buffer, int light_depth, int eye_depth, int W, int H, int spp){
    (void) light_depth;
    if(!pt_ns::scene){ std::cerr << "run_cuda_pt: no scene moved to the device" << std::endl; return; }
    hpt_params p = hpt_host::g_run_params;
    if(hpt_host::g_seed_from_clock) p.seed = (uint64_t) time(nullptr);                // reference: time(NULL), src/pt_cu.cu:282
    int rc = hpt_render_pt(pt_ns::scene, &cam, W, H, eye_depth, spp, &p, &image_buffer->x);
    if(rc != HPT_OK) std::cerr << "run_cuda_pt: " << hpt_last_error() << std::endl;
}

namespace bdpt_ns {
std::vector<CudaSphere> cuda_spheres;
std::vector<CudaTriangle> cuda_triangles;
std::vector<CudaLight> cuda_lights;
int light_sample = 0;
hpt_scene *scene = nullptr;
}

void move_data_to_cuda_bdpt(std::map<int, hpt_host::AABB> groups, std::vector<CudaLight> &lights, int light_sample){
    using namespace hpt_host;
    bdpt_ns::cuda_spheres.clear(); bdpt_ns::cuda_triangles.clear(); bdpt_ns::cuda_lights.clear();
    std::vector<int32_t> kind, index, group;
    for(auto &g : groups){
        for(Object *obj : g.second.objs){
            if(const Sphere *sph = dynamic_cast<const Sphere *>(obj)){
                CudaSphere c; memset(&c, 0, sizeof c);
                c.center = float3{sph->center.x, sph->center.y, sph->center.z}; c.r = sph->r; c.mtl = to_cmtl(sph->mtl); c.id = sph->obj_id;
                kind.push_back(0); index.push_back((int32_t) bdpt_ns::cuda_spheres.size()); group.push_back(g.first);
                bdpt_ns::cuda_spheres.push_back(c);
            } else if(const Triangle *tri = dynamic_cast<const Triangle *>(obj)){
                CudaTriangle c; memset(&c, 0, sizeof c);
                c.v0 = float3{tri->vert[0].x, tri->vert[0].y, tri->vert[0].z}; c.v1 = float3{tri->vert[1].x, tri->vert[1].y, tri->vert[1].z};
                c.v2 = float3{tri->vert[2].x, tri->vert[2].y, tri->vert[2].z}; c.mtl = to_cmtl(tri->mtl); c.id = tri->obj_id;
                kind.push_back(1); index.push_back((int32_t) bdpt_ns::cuda_triangles.size()); group.push_back(g.first);
                bdpt_ns::cuda_triangles.push_back(c);
            }
        }
    }
    for(auto l : lights){
        float len = std::sqrt(l.dir.x * l.dir.x + l.dir.y * l.dir.y + l.dir.z * l.dir.z);
        l.dir = float3{l.dir.x / len, l.dir.y / len, l.dir.z / len};
        // the reference divides illum by light_sample here (src/bdpt_cu_helper.cpp:60-62) for its CUDA kernel's
        // light_sample-times larger vertex pool; the estimator behind run_cuda_bdpt is run_cpu_bdpt's, which takes
        // the undivided flux (src/cpu_bdpt.cpp:255), so the lights are handed over as they are
        bdpt_ns::cuda_lights.push_back(l);
    }
    bdpt_ns::light_sample = light_sample;
    if(bdpt_ns::scene){ hpt_scene_destroy(bdpt_ns::scene); bdpt_ns::scene = nullptr; }
    int rc = hpt_scene_create(bdpt_ns::cuda_lights.data(), (int) bdpt_ns::cuda_lights.size(),
                              bdpt_ns::cuda_spheres.data(), (int) bdpt_ns::cuda_spheres.size(),
                              bdpt_ns::cuda_triangles.data(), (int) bdpt_ns::cuda_triangles.size(), &bdpt_ns::scene);
    if(rc == HPT_OK) rc = hpt_scene_set_groups(bdpt_ns::scene, kind.data(), index.data(), group.data(), (int) kind.size());
    if(rc != HPT_OK){ std::cerr << "move_data_to_cuda_bdpt: " << hpt_last_error() << std::endl; if(bdpt_ns::scene){ hpt_scene_destroy(bdpt_ns::scene); bdpt_ns::scene = nullptr; } }
}

void run_cuda_bdpt(CudaCamera cam, float3 *image_buffer, int light_depth, int eye_depth, int W, int H, int spp, int spl){
    if(!bdpt_ns::scene){ std::cerr << "run_cuda_bdpt: no scene moved to the device" << std::endl; return; }
    hpt_params p = hpt_host::g_run_params;
    if(hpt_host::g_seed_from_clock) p.seed = (uint64_t) time(nullptr);
    int rc = hpt_render_bdpt(bdpt_ns::scene, &cam, W, H, eye_depth, light_depth, spp, spl, &p, &image_buffer->x);
    if(rc != HPT_OK) std::cerr << "run_cuda_bdpt: " << hpt_last_error() << std::endl;
}

// ---- C entry points for tests (flattening + camera through the C++ mirror) -------------------
extern "C" {
int hpt_host_flatten_scene_file(const char *path, int *nl, int *ns, int *nt, const void **lights, const void **spheres,
                                const void **tris, float *camera_rgb /* CudaCamera, 21 floats */, int W, int H, int *res_wh){
    using namespace hpt_host;
    static std::vector<CudaLight> L; static std::vector<CudaSphere> S; static std::vector<CudaTriangle> T;
    SceneFile sc;
    if(!parse_scene_file(path, sc)) return 1;
    // same flattening as move_data_to_cuda_pt, without touching the device
    L.clear(); S.clear(); T.clear();
    for(auto &g : sc.groups) for(Object *obj : g.second.objs){
        if(const Sphere *sph = dynamic_cast<const Sphere *>(obj)){
            CudaSphere c; memset(&c, 0, sizeof c);
            c.center = float3{sph->center.x, sph->center.y, sph->center.z}; c.r = sph->r; c.mtl = to_cmtl(sph->mtl); c.id = sph->obj_id; S.push_back(c);
        } else if(const Triangle *tri = dynamic_cast<const Triangle *>(obj)){
            CudaTriangle c; memset(&c, 0, sizeof c);
            c.v0 = float3{tri->vert[0].x, tri->vert[0].y, tri->vert[0].z}; c.v1 = float3{tri->vert[1].x, tri->vert[1].y, tri->vert[1].z};
            c.v2 = float3{tri->vert[2].x, tri->vert[2].y, tri->vert[2].z}; c.mtl = to_cmtl(tri->mtl); c.id = tri->obj_id; T.push_back(c);
        }
    }
    for(auto l : sc.lights){
        float len = std::sqrt(l.dir.x * l.dir.x + l.dir.y * l.dir.y + l.dir.z * l.dir.z);
        l.dir = float3{l.dir.x / len, l.dir.y / len, l.dir.z / len};
        L.push_back(l);
    }
    *nl = (int) L.size(); *ns = (int) S.size(); *nt = (int) T.size();
    *lights = L.data(); *spheres = S.data(); *tris = T.data();
    CudaCamera cam = make_cuda_camera(sc.camera, 50.0f, W, H);
    memcpy(camera_rgb, &cam, sizeof cam);
    res_wh[0] = sc.resolution.first; res_wh[1] = sc.resolution.second;
    return 0;
}
int hpt_host_write_image(const char *path, const float *rgb, int W, int H){
    std::string err;
    return hpt_host::write_image(path, (const float3 *) rgb, W, H, &err) ? 0 : 1;
}
}
