// Host front-end of the PT / BDPT hot path: scene text + OBJ ingestion, flattening into the boundary records,
// camera, the reference-named helper API and the output stage.  See scene_model.hpp for what each piece replaces.
#include "scene_model.hpp"
#include "text_cursor.hpp"
#include "../../../include/hpt.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <iostream>
#include <type_traits>
#include <zlib.h>

namespace hpt_host {

namespace {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kRadiansPerDegree = 0.01745329251994329576923690768489f;      // glm::radians<float>

double now_ms(){
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace

// ---- object model ------------------------------------------------------------------------------------

void AABB::add_obj(Object *obj){
    auto cover = [this](float x, float y, float z){
        const float p[3] = { x, y, z };
        for(int a = 0; a < 3; ++a){ min[a] = std::min(min[a], p[a]); max[a] = std::max(max[a], p[a]); }
    };
    if(obj->kind() == Object::kSphere){
        const Sphere *s = static_cast<const Sphere *>(obj);
        cover(s->center.x + s->r, s->center.y + s->r, s->center.z + s->r);
        cover(s->center.x - s->r, s->center.y - s->r, s->center.z - s->r);
    } else {
        const Triangle *t = static_cast<const Triangle *>(obj);
        for(const vec3 &v : t->vert) cover(v.x, v.y, v.z);
    }
    objs.push_back(obj);
}

SceneFile::SceneFile(){ materials.emplace_back(); }

std::map<int, AABB> &SceneFile::groups(){
    for(; grouped_items_ < items.size(); ++grouped_items_){
        const SceneItem &it = items[grouped_items_];
        std::unique_ptr<Object> obj;
        if(it.kind == Object::kSphere){
            const float *g = &sphere_geom[(size_t) it.geom * 4];
            auto s = std::make_unique<Sphere>();
            s->center = vec3(g[0], g[1], g[2]); s->r = g[3];
            obj = std::move(s);
        } else {
            const float *g = &tri_geom[(size_t) it.geom * 9];
            auto t = std::make_unique<Triangle>();
            for(int k = 0; k < 3; ++k) t->vert[k] = vec3(g[3 * k], g[3 * k + 1], g[3 * k + 2]);
            obj = std::move(t);
        }
        obj->mtl = materials[it.material];
        obj->obj_id = it.obj_id;
        groups_[it.group].add_obj(obj.get());
        pool_.push_back(std::move(obj));
    }
    return groups_;
}

// ---- scene text ----------------------------------------------------------------------------------------
// One tag character, then that tag's numbers.  A character that is no tag is dropped and the next one is tried,
// which is how the extra numbers behind some 'M' lines of mis_test.txt disappear (digit by digit) in the reference.

void parse_scene_text(const char *begin, const char *end, SceneFile &sc){
    const double t0 = now_ms();
    TextCursor in(begin, end);
    uint32_t material = (uint32_t) sc.materials.size() - 1;       // material in force
    int group = 0;
    int next_id = 0;
    for(const SceneItem &it : sc.items) next_id = std::max(next_id, it.obj_id + 1);
    // a rough reservation from the input size: a 'T' line is ~100 bytes
    sc.tri_geom.reserve(sc.tri_geom.size() + (size_t) (end - begin) / 96 * 9);
    char tag;
    float v[12];
    while(in.tag(tag)){
        switch(tag){
        case 'T':
            if(in.f32s(v, 9)){
                sc.items.push_back(SceneItem{ (uint8_t) Object::kTriangle, group, material, (uint32_t) (sc.tri_geom.size() / 9), next_id++ });
                sc.tri_geom.insert(sc.tri_geom.end(), v, v + 9);
                sc.tri_cnt++;
            }
            break;
        case 'S':
            if(in.f32s(v, 4)){
                sc.items.push_back(SceneItem{ (uint8_t) Object::kSphere, group, material, (uint32_t) (sc.sphere_geom.size() / 4), next_id++ });
                sc.sphere_geom.insert(sc.sphere_geom.end(), v, v + 4);
                sc.ball_cnt++;
            }
            break;
        case 'M':
            if(in.f32s(v, 6)){
                Material m; m.base_color = vec3(v[0], v[1], v[2]); m.roughness = v[3]; m.metallic = v[4]; m.eta = v[5];
                sc.materials.push_back(m);
                material = (uint32_t) sc.materials.size() - 1;
            }
            break;
        case 'G': { int g; if(in.i32(g)) group = g; break; }
        case 'E': if(in.f32s(v, 3)) sc.camera.eye = vec3(v[0], v[1], v[2]); break;
        case 'V':
            if(in.f32s(v, 6)){ sc.camera.look_at = vec3(v[0], v[1], v[2]); sc.camera.view_up = vec3(v[3], v[4], v[5]); }
            break;
        case 'F': { float f; if(in.f32(f)) sc.camera.fov = f; break; }        // stored; the CLI renders with 50 like the reference
        case 'R': { int w, h; if(in.i32(w) && in.i32(h)) sc.resolution = { w, h }; break; }
        case 'L': {
            int parallel = 0; float ball_r = 0.0f;
            if(in.f32s(v, 10) && in.i32(parallel) && in.f32(ball_r)){
                CudaLight L; memset(&L, 0, sizeof L);
                L.pos = float3{ v[0], v[1], v[2] }; L.dir = float3{ v[3], v[4], v[5] }; L.illum = float3{ v[6], v[7], v[8] };
                L.cutoff = v[9] * kRadiansPerDegree;
                L.is_parallel = parallel;
                L.light_ball.center = L.pos; L.light_ball.r = ball_r;
                L.light_ball.mtl_old.Kd = L.illum;                  // the only material field of a light ball the kernels read
                sc.lights.push_back(L);
            }
            break;
        }
        case '/': { char second; if(in.tag(second) && second == '/') in.skip_line(); break; }     // "//" comment; a lone '/' eats one character
        default: break;
        }
    }
    sc.parse_ms = now_ms() - t0;
}

bool parse_scene_file(const std::string &path, SceneFile &out){
    MappedFile f;
    if(!f.open(path)) return false;
    parse_scene_text(f.begin(), f.end(), out);
    return true;
}

// ---- Wavefront OBJ -----------------------------------------------------------------------------------

int append_obj(const std::string &path, const Material &mtl, int group_id, SceneFile &sc, std::string *err){
    const double t0 = now_ms();
    MappedFile f;
    if(!f.open(path)){ if(err) *err = "cannot open " + path; return -1; }
    TextCursor in(f.begin(), f.end());
    sc.materials.push_back(mtl);
    const uint32_t material = (uint32_t) sc.materials.size() - 1;
    int next_id = 0;
    for(const SceneItem &it : sc.items) next_id = std::max(next_id, it.obj_id + 1);
    std::vector<float> pos;                     // xyz per 'v'
    pos.reserve(f.size() / 24);
    std::vector<uint32_t> corner;
    int added = 0;
    while(!in.at_end()){
        std::string_view key = in.word();
        if(key == "v"){
            float x, y, z;
            if(!(in.number_on_line(x) && in.number_on_line(y) && in.number_on_line(z))){ if(err) *err = "malformed vertex in " + path; return -1; }
            pos.push_back(x); pos.push_back(y); pos.push_back(z);
        } else if(key == "f"){
            corner.clear();
            const long long nv = (long long) (pos.size() / 3);
            long long idx;
            while(in.number_on_line(idx)){
                in.skip_word();                                  // "/vt/vn" behind the position index
                if(idx < 0) idx = nv + idx + 1;
                if(idx < 1 || idx > nv){ if(err) *err = "face index out of range in " + path; return -1; }
                corner.push_back((uint32_t) (idx - 1));
            }
            for(size_t k = 1; k + 1 < corner.size(); ++k){
                const uint32_t tri[3] = { corner[0], corner[k], corner[k + 1] };
                sc.items.push_back(SceneItem{ (uint8_t) Object::kTriangle, group_id, material, (uint32_t) (sc.tri_geom.size() / 9), next_id++ });
                for(uint32_t c : tri) sc.tri_geom.insert(sc.tri_geom.end(), &pos[(size_t) c * 3], &pos[(size_t) c * 3] + 3);
                sc.tri_cnt++; added++;
            }
        }
        in.skip_line();                                          // vt, vn, o, g, s, usemtl, mtllib, '#', trailing fields
    }
    sc.parse_ms = now_ms() - t0;
    return added;
}

// ---- flattening into the boundary records ------------------------------------------------------------

namespace {

CudaMaterial boundary_material(const Material &m){
    CudaMaterial cm;
    cm.base_color = float3{ m.base_color.x, m.base_color.y, m.base_color.z };
    cm.roughness = m.roughness; cm.metallic = m.metallic; cm.eta = m.eta;
    cm.type = m.eta > 0.0f ? 1 : (m.metallic > 0.0f ? 2 : 3);     // dielectric / conductor / uber tag (src/geometric.cu:41-49; no kernel reads it)
    return cm;
}

struct Flattener {
    FlatScene &out;
    void sphere(const float *g, const Material &m, int id, int group){
        CudaSphere c; memset(&c, 0, sizeof c);
        c.center = float3{ g[0], g[1], g[2] }; c.r = g[3]; c.mtl = boundary_material(m); c.id = id;
        out.kind.push_back(0); out.index.push_back((int32_t) out.spheres.size()); out.group.push_back(group);
        out.spheres.push_back(c);
    }
    void triangle(const float *g, const Material &m, int id, int group){
        CudaTriangle c; memset(&c, 0, sizeof c);
        c.v0 = float3{ g[0], g[1], g[2] }; c.v1 = float3{ g[3], g[4], g[5] }; c.v2 = float3{ g[6], g[7], g[8] };
        c.mtl = boundary_material(m); c.id = id;
        out.kind.push_back(1); out.index.push_back((int32_t) out.triangles.size()); out.group.push_back(group);
        out.triangles.push_back(c);
    }
    void lights(const std::vector<CudaLight> &src){
        for(CudaLight l : src){
            float len = std::sqrt(l.dir.x * l.dir.x + l.dir.y * l.dir.y + l.dir.z * l.dir.z);
            l.dir = float3{ l.dir.x / len, l.dir.y / len, l.dir.z / len };
            out.lights.push_back(l);
        }
    }
};

void flatten_groups(const std::map<int, AABB> &groups, const std::vector<CudaLight> &lights, FlatScene &out){
    out = FlatScene();
    Flattener fl{ out };
    for(const auto &g : groups){
        for(const Object *obj : g.second.objs){
            if(obj->kind() == Object::kSphere){
                const Sphere *s = static_cast<const Sphere *>(obj);
                const float geom[4] = { s->center.x, s->center.y, s->center.z, s->r };
                fl.sphere(geom, s->mtl, s->obj_id, g.first);
            } else {
                const Triangle *t = static_cast<const Triangle *>(obj);
                float geom[9];
                for(int k = 0; k < 3; ++k){ geom[3 * k] = t->vert[k].x; geom[3 * k + 1] = t->vert[k].y; geom[3 * k + 2] = t->vert[k].z; }
                fl.triangle(geom, t->mtl, t->obj_id, g.first);
            }
        }
    }
    fl.lights(lights);
}

} // namespace

void flatten_scene(const SceneFile &sc, FlatScene &out){
    out = FlatScene();
    Flattener fl{ out };
    // group id order, file order inside a group: a stable sort of the item numbers by group
    std::vector<uint32_t> order(sc.items.size());
    for(uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b){ return sc.items[a].group < sc.items[b].group; });
    out.triangles.reserve((size_t) sc.tri_cnt); out.spheres.reserve((size_t) sc.ball_cnt);
    for(uint32_t i : order){
        const SceneItem &it = sc.items[i];
        if(it.kind == Object::kSphere) fl.sphere(&sc.sphere_geom[(size_t) it.geom * 4], sc.materials[it.material], it.obj_id, it.group);
        else fl.triangle(&sc.tri_geom[(size_t) it.geom * 9], sc.materials[it.material], it.obj_id, it.group);
    }
    fl.lights(sc.lights);
}

// ---- camera --------------------------------------------------------------------------------------------
// Pinhole frame of the reference CLI: the image plane sits one unit in front of the eye, UL is its top-left corner,
// dx / dy step one pixel right / down.  Every float operation is in the reference's order (the camera record is
// compared byte for byte with the Python mirror): unit vectors are v * (1 / sqrt(v.v)) as glm::normalize computes
// them, the half-height tangent is taken in double precision from a float angle.

CudaCamera make_cuda_camera(const Camera &camera, float fov_deg, int W, int H){
    struct V { float x, y, z; };
    auto sub = [](V a, V b){ return V{ a.x - b.x, a.y - b.y, a.z - b.z }; };
    auto add = [](V a, V b){ return V{ a.x + b.x, a.y + b.y, a.z + b.z }; };
    auto scale = [](float s, V a){ return V{ a.x * s, a.y * s, a.z * s }; };
    auto over = [](V a, float s){ return V{ a.x / s, a.y / s, a.z / s }; };
    auto cross = [](V a, V b){ return V{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; };
    auto unit = [&](V a){ float inv = 1.0f / std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); return scale(inv, a); };
    const V eye{ camera.eye.x, camera.eye.y, camera.eye.z }, target{ camera.look_at.x, camera.look_at.y, camera.look_at.z },
            up{ camera.view_up.x, camera.view_up.y, camera.view_up.z };

    const float aspect = float(W) / float(H);
    const float angle = fov_deg * kPi / 180.0f;
    const float half_h = (float) std::tan((double) (angle / 2));
    const float half_w = aspect * half_h;
    const V back = unit(sub(eye, target));            // from the target towards the eye
    const V right = unit(cross(up, back));
    const V upward = cross(back, right);
    const V corner = sub(add(sub(eye, scale(half_w, right)), scale(half_h, upward)), back);
    const V step_x = over(scale(2 * half_w, right), float(W));
    const V step_y = over(scale(-2 * half_h, upward), float(H));

    CudaCamera cam; memset(&cam, 0, sizeof cam);       // U, V, W stay zero: no kernel reads them
    cam.eye = float3{ eye.x, eye.y, eye.z };
    cam.UL = float3{ corner.x, corner.y, corner.z };
    cam.dx = float3{ step_x.x, step_x.y, step_x.z };
    cam.dy = float3{ step_y.x, step_y.y, step_y.z };
    return cam;
}

// ---- output stage --------------------------------------------------------------------------------------

namespace {

void be32(std::vector<unsigned char> &v, uint32_t x){ for(int s = 24; s >= 0; s -= 8) v.push_back((unsigned char) (x >> s)); }

void png_chunk(std::vector<unsigned char> &png, const char *type, const unsigned char *body, size_t n){
    be32(png, (uint32_t) n);
    const size_t at = png.size();
    png.insert(png.end(), type, type + 4);
    if(n) png.insert(png.end(), body, body + n);
    be32(png, (uint32_t) crc32(0L, png.data() + at, (uInt) (png.size() - at)));
}

} // namespace

// 8-bit RGB, rows top to bottom -> PNG (colour type 2, no interlace, filter 0 on every row, one IDAT)
bool write_png_rgb8(const std::string &path, const unsigned char *rgb, int W, int H, std::string *err){
    const size_t stride = (size_t) W * 3;
    std::vector<unsigned char> scan((stride + 1) * (size_t) H);
    for(int y = 0; y < H; ++y){
        scan[(stride + 1) * y] = 0;
        memcpy(&scan[(stride + 1) * y + 1], rgb + stride * y, stride);
    }
    uLongf zlen = compressBound((uLong) scan.size());
    std::vector<unsigned char> z(zlen);
    if(compress2(z.data(), &zlen, scan.data(), (uLong) scan.size(), 6) != Z_OK){ if(err) *err = "zlib failure"; return false; }
    std::vector<unsigned char> png = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n' };
    std::vector<unsigned char> head;
    be32(head, (uint32_t) W); be32(head, (uint32_t) H);
    const unsigned char tail[5] = { 8, 2, 0, 0, 0 };             // bit depth, colour type RGB, deflate, adaptive filtering, no interlace
    head.insert(head.end(), tail, tail + 5);
    png_chunk(png, "IHDR", head.data(), head.size());
    png_chunk(png, "IDAT", z.data(), zlen);
    png_chunk(png, "IEND", nullptr, 0);
    FILE *f = fopen(path.c_str(), "wb");
    if(!f){ if(err) *err = "cannot open " + path; return false; }
    const bool ok = fwrite(png.data(), 1, png.size(), f) == png.size();
    fclose(f);
    if(!ok && err) *err = "short write to " + path;
    return ok;
}

bool write_image(const std::string &path, const float3 *img, int W, int H, std::string *err){
    const bool pfm = path.size() > 4 && path.compare(path.size() - 4, 4, ".pfm") == 0;
    if(pfm){
        FILE *f = fopen(path.c_str(), "wb");
        if(!f){ if(err) *err = "cannot open " + path; return false; }
        fprintf(f, "PF\n%d %d\n-1.0\n", W, H);
        for(int y = H - 1; y >= 0; --y) fwrite(&img[(size_t) y * W], sizeof(float3), (size_t) W, f);
        fclose(f);
        return true;
    }
    std::vector<unsigned char> rgb((size_t) W * H * 3);
    if(hpt_tonemap_host(&img->x, rgb.data(), (int64_t) W * H, 0) != HPT_OK){ if(err) *err = hpt_last_error(); return false; }
    return write_png_rgb8(path, rgb.data(), W, H, err);
}

hpt_params g_run_params = { 1, 0, 0, 0, 0, 0, 0, 0, 0 };
bool g_seed_from_clock = true;
int g_devices = 1;            // devices the helper API renders on (pt_cli --gpus): > 1 = fan-out inside the blocking call

} // namespace hpt_host

// ---- the reference's helper API --------------------------------------------------------------------------
// The reference keeps what it moved in namespace-level vectors and re-uploads them on every render
// (src/pt_cu_helper.cpp:3-10,66-77); here the moved scene lives on the device behind an hpt_scene handle.

namespace {

struct MovedScene {
    hpt_host::FlatScene flat;
    hpt_scene *device = nullptr;          // one device ...
    hpt_multi *fan_out = nullptr;         // ... or the fan-out over hpt_host::g_devices devices (hpt_multi_*, RCCL gather)
    int light_sample = 0;
    void drop(){
        if(device){ hpt_scene_destroy(device); device = nullptr; }
        if(fan_out){ hpt_multi_destroy(fan_out); fan_out = nullptr; }
    }
    bool ready() const { return device || fan_out; }
    bool upload(const char *who, bool with_groups){
        drop();
        int rc;
        if(hpt_host::g_devices > 1){
            rc = hpt_multi_create(flat.lights.data(), (int) flat.lights.size(), flat.spheres.data(), (int) flat.spheres.size(),
                                  flat.triangles.data(), (int) flat.triangles.size(), nullptr, hpt_host::g_devices, 0, &fan_out);
            if(rc == HPT_OK && with_groups)
                rc = hpt_multi_set_groups(fan_out, flat.kind.data(), flat.index.data(), flat.group.data(), (int) flat.kind.size());
        } else {
            rc = hpt_scene_create(flat.lights.data(), (int) flat.lights.size(), flat.spheres.data(), (int) flat.spheres.size(),
                                  flat.triangles.data(), (int) flat.triangles.size(), &device);
            if(rc == HPT_OK && with_groups)
                rc = hpt_scene_set_groups(device, flat.kind.data(), flat.index.data(), flat.group.data(), (int) flat.kind.size());
        }
        if(rc != HPT_OK){ std::cerr << who << ": " << hpt_last_error() << std::endl; drop(); return false; }
        return true;
    }
};

MovedScene g_pt, g_bdpt;

hpt_params run_params(){
    hpt_params p = hpt_host::g_run_params;
    if(hpt_host::g_seed_from_clock) p.seed = (uint64_t) time(nullptr);            // the reference seeds from time(NULL), src/pt_cu.cu:282
    return p;
}

} // namespace

void move_data_to_cuda_pt(std::map<int, hpt_host::AABB> groups, std::vector<CudaLight> &lights, int light_sample){
    hpt_host::flatten_groups(groups, lights, g_pt.flat);
    g_pt.light_sample = light_sample;
    if(g_pt.upload("move_data_to_cuda_pt", false)) std::cout << "moved" << std::endl;
}

void run_cuda_pt(CudaCamera cam, float3 *image_buffer, int light_depth, int eye_depth, int W, int H, int spp){
    (void) light_depth;
    if(!g_pt.ready()){ std::cerr << "run_cuda_pt: no scene moved to the device" << std::endl; return; }
    hpt_params p = run_params();
    int rc = g_pt.fan_out ? hpt_multi_render_pt(g_pt.fan_out, &cam, W, H, eye_depth, spp, &p, &image_buffer->x)
                          : hpt_render_pt(g_pt.device, &cam, W, H, eye_depth, spp, &p, &image_buffer->x);
    if(rc != HPT_OK) std::cerr << "run_cuda_pt: " << hpt_last_error() << std::endl;
}

void move_data_to_cuda_bdpt(std::map<int, hpt_host::AABB> groups, std::vector<CudaLight> &lights, int light_sample){
    // the reference divides illum by light_sample here for its CUDA kernel's larger vertex pool (src/bdpt_cu_helper.cpp:60-62);
    // the estimator behind run_cuda_bdpt is run_cpu_bdpt's, which takes the undivided flux (src/cpu_bdpt.cpp:255)
    hpt_host::flatten_groups(groups, lights, g_bdpt.flat);
    g_bdpt.light_sample = light_sample;
    g_bdpt.upload("move_data_to_cuda_bdpt", true);
}

void run_cuda_bdpt(CudaCamera cam, float3 *image_buffer, int light_depth, int eye_depth, int W, int H, int spp, int spl){
    if(!g_bdpt.ready()){ std::cerr << "run_cuda_bdpt: no scene moved to the device" << std::endl; return; }
    hpt_params p = run_params();
    int rc = g_bdpt.fan_out ? hpt_multi_render_bdpt(g_bdpt.fan_out, &cam, W, H, eye_depth, light_depth, spp, spl, &p, &image_buffer->x)
                            : hpt_render_bdpt(g_bdpt.device, &cam, W, H, eye_depth, light_depth, spp, spl, &p, &image_buffer->x);
    if(rc != HPT_OK) std::cerr << "run_cuda_bdpt: " << hpt_last_error() << std::endl;
}

// ---- C entry points (tests, Python: scene_io.load_scene_fast / load_obj) ---------------------------------------
extern "C" {

// Parses a scene file (or, kind = 1, an OBJ appended to an empty scene with a 0.7 grey diffuse material in group 0)
// and flattens it without touching the device.  The arrays stay valid until the next call.
int hpt_host_flatten_file(const char *path, int kind, int *nl, int *ns, int *nt, const void **lights, const void **spheres,
                          const void **tris, const int32_t **okind, const int32_t **oindex, const int32_t **ogroup,
                          float *camera21 /* CudaCamera for W x H, fov 50 */, int W, int H, int *res_wh, double *parse_ms, float *eye_look_up_fov10){
    using namespace hpt_host;
    static FlatScene flat;
    SceneFile sc;
    if(kind == 1){
        Material grey; grey.base_color = vec3(0.7f, 0.7f, 0.7f); grey.roughness = 1.0f;
        std::string err;
        if(append_obj(path, grey, 0, sc, &err) < 0) return 1;
    } else if(!parse_scene_file(path, sc)) return 1;
    flatten_scene(sc, flat);
    *nl = (int) flat.lights.size(); *ns = (int) flat.spheres.size(); *nt = (int) flat.triangles.size();
    *lights = flat.lights.data(); *spheres = flat.spheres.data(); *tris = flat.triangles.data();
    if(okind){ *okind = flat.kind.data(); *oindex = flat.index.data(); *ogroup = flat.group.data(); }
    if(camera21 && W > 0 && H > 0){ CudaCamera cam = make_cuda_camera(sc.camera, 50.0f, W, H); memcpy(camera21, &cam, sizeof cam); }
    if(res_wh){ res_wh[0] = sc.resolution.first; res_wh[1] = sc.resolution.second; }
    if(parse_ms) *parse_ms = sc.parse_ms;
    if(eye_look_up_fov10){
        const float v[10] = { sc.camera.eye.x, sc.camera.eye.y, sc.camera.eye.z, sc.camera.look_at.x, sc.camera.look_at.y, sc.camera.look_at.z,
                              sc.camera.view_up.x, sc.camera.view_up.y, sc.camera.view_up.z, sc.camera.fov };
        memcpy(eye_look_up_fov10, v, sizeof v);
    }
    return 0;
}

// older entry point of the mirror tests: scene text only, no ordering arrays
int hpt_host_flatten_scene_file(const char *path, int *nl, int *ns, int *nt, const void **lights, const void **spheres,
                                const void **tris, float *camera21, int W, int H, int *res_wh){
    return hpt_host_flatten_file(path, 0, nl, ns, nt, lights, spheres, tris, nullptr, nullptr, nullptr, camera21, W, H, res_wh, nullptr, nullptr);
}

// the same scene through the reference-shaped object model (std::map<int, AABB> of Object*): must flatten to the same bytes
int hpt_host_flatten_via_groups(const char *path, int *nl, int *ns, int *nt, const void **lights, const void **spheres, const void **tris){
    using namespace hpt_host;
    static FlatScene flat;
    SceneFile sc;
    if(!parse_scene_file(path, sc)) return 1;
    flatten_groups(sc.groups(), sc.lights, flat);
    *nl = (int) flat.lights.size(); *ns = (int) flat.spheres.size(); *nt = (int) flat.triangles.size();
    *lights = flat.lights.data(); *spheres = flat.spheres.data(); *tris = flat.triangles.data();
    return 0;
}

int hpt_host_write_image(const char *path, const float *rgb, int W, int H){       // tone-maps on the device
    std::string err;
    return hpt_host::write_image(path, (const float3 *) rgb, W, H, &err) ? 0 : 1;
}

int hpt_host_write_png_rgb8(const char *path, const unsigned char *rgb, int W, int H){
    std::string err;
    return hpt_host::write_png_rgb8(path, rgb, W, H, &err) ? 0 : 1;
}

}
