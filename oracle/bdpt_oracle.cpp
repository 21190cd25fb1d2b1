// TEST INFRASTRUCTURE ONLY -- CPU oracle: the bidirectional estimator of the reference's
// CPU renderer (reference: src/cpu_bdpt.cpp:30-488, run_cpu_bdpt), including the CPU scene
// model it traces (reference: src/object.cpp:16-121: virtual sphere/triangle intersection,
// one-level group boxes).  It is the checker / CPU baseline for the BDPT-estimator rows of
// SURVEY.md section 8; nothing in path_tracing_amd/ links or calls it.
//
// PARITY STATUS: parity unpinned (see ref_math.hpp).  Advisory: in replay mode (one
// std::mt19937 stream per pass, seeds 1337 / 9999, one thread) it is compared with an image the
// survey stage produced from the reference's own cpu_bdpt.cpp (tests/golden/README.md).
//
// Host arithmetic notes: the reference's CPU model uses glm: normalize(v) = v * (1/sqrt(dot)),
// while its float3 helpers divide by the length (ref_math.hpp); each is kept where the
// reference has it.  Triangle determinant test compares in double (ESP is a double literal,
// include/object.h:7).
#include "ref_math.hpp"
#include <algorithm>
#include <map>
#include <random>
#include <vector>
#include <omp.h>

using namespace orc;

extern "C" {

struct OracleBdptOpts {
    uint64_t seed;        // counter mode stream key
    int rng_mode;         // 0 = PCG counter streams (any thread count), 1 = std::mt19937 replay (reference seeds)
    int threads;          // OpenMP threads (0 = default); replay is bit-reproducible only with 1
    int x0, y0, x1, y1;   // pixel window (counter mode); replay mode renders rows [0, y1) in full
    int max_delta;        // cap on free delta bounces (reference: uncapped)
};

struct OracleBdptStats { uint64_t samples, closest_rays, shadow_rays, connections, tri_tests, sphere_tests; };

} // extern "C"

namespace {

// ---- glm-style vec3 helpers (multiply by inverse sqrt) --------------------------------------
inline V3 gnormalize(V3 a){ float inv = 1.0f / sqrtf(dot(a, a)); return a * inv; }
inline V3 smul(float s, V3 a){ return v3(a.x * s, a.y * s, a.z * s); }

struct Obj { int kind; int index; };                 // 0 sphere, 1 triangle
struct Group { V3 mn, mx; std::vector<Obj> objs; };

struct BScene {
    const RLight *lights; int nl;
    const RSphere *spheres; const RTriangle *tris;
    std::vector<Group> groups;                        // map order
};

struct Rng2 {
    int mode; Pcg pcg; std::mt19937 mt; std::uniform_real_distribution<float> U{0.0f, 1.0f};
    float next(){ return mode == 0 ? pcg.next() : U(mt); }
};

// light / eye vertices (reference include/bdpt_cu.cuh:6-27, fields used by the CPU path)
struct LightVertex { V3 pos, normal, throughput; RMat mtl; bool is_light_source, is_parallel; float source_cutoff, pdf_fwd, pdf_rev; };
struct EyeVertex { V3 pos, normal, throughput; RMat mtl; float pdf_fwd, pdf_rev; };

// AABB::intersectAABB, src/object.cpp:104-121 (widens degenerate boxes in place, like the reference)
bool group_box_hit(Group &g, V3 ro, V3 rd, float tmin, float tmax){
    float *mn = &g.mn.x, *mx = &g.mx.x; const float *o = &ro.x, *d = &rd.x;
    for(int a = 0; a < 3; ++a){
        if(mx[a] - mn[a] < 1e-6f){ mn[a] -= 0.5f * 1e-6f; mx[a] += 0.5f * 1e-6f; }
        float invD = 1.0f / d[a];
        float t0 = (mn[a] - o[a]) * invD;
        float t1 = (mx[a] - o[a]) * invD;
        if(invD < 0) std::swap(t0, t1);
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if(tmax <= tmin) return false;
    }
    return true;
}

// Sphere::check_intersect, src/object.cpp:16-44 (scale = 1; u,v outputs are unused by the caller)
bool cpu_sphere(const RSphere &s, V3 O, V3 vec, float tMin, float tMax, float &t){
    V3 D = gnormalize(vec);
    V3 OC = O - s.center;
    const float a = 1.0f * D.x * D.x + 1.0f * D.y * D.y + 1.0f * D.z * D.z;
    const float b = 2.0f * (1.0f * D.x * OC.x + 1.0f * D.y * OC.y + 1.0f * D.z * OC.z);
    const float c = 1.0f * OC.x * OC.x + 1.0f * OC.y * OC.y + 1.0f * OC.z * OC.z - s.r * s.r;
    if(c <= 1e-6f) return false;                     // origin inside / on the sphere
    const float disc = b * b - 4.0f * a * c;
    if(disc < 0.0f) return false;
    const float sdisc = std::sqrt(std::max(0.0f, disc));
    float t0 = (-b - sdisc) / (2.0f * a);
    float t1 = (-b + sdisc) / (2.0f * a);
    if(t0 > t1) std::swap(t0, t1);
    float tc = (t0 >= tMin) ? t0 : t1;
    if(tc < tMin || tc > tMax) return false;
    t = tc;
    return true;
}
// Triangle::check_intersect, src/object.cpp:72-95
bool cpu_triangle(const RTriangle &tr, V3 O, V3 vec, float tMin, float tMax, float &t){
    const V3 e1 = tr.v1 - tr.v0, e2 = tr.v2 - tr.v0;
    const V3 pvec = cross(vec, e2);
    const float det = dot(e1, pvec);
    if((double) std::fabs(det) < 1e-6) return false;
    const float invDet = 1.0f / det;
    const V3 tvec = O - tr.v0;
    float u = dot(tvec, pvec) * invDet;
    if(u < 0.0f || u > 1.0f) return false;
    const V3 qvec = cross(tvec, e1);
    float v = dot(vec, qvec) * invDet;
    if(v < 0.0f || (u + v) > 1.0f) return false;
    float tt = dot(e2, qvec) * invDet;
    if(tt < tMin || tt > tMax) return false;
    t = tt;
    return true;
}

// cpu_find_closest_hit, src/cpu_bdpt.cpp:30-80
Hit bd_closest(BScene &sc, V3 ro, V3 rd, OracleBdptStats &st){
    Hit best; best.hit = false; best.t = 1e20f; best.is_light = false; best.prim = -1;
    best.mtl.base_color = v3(0, 0, 0); best.mtl.roughness = 0; best.mtl.metallic = 0; best.mtl.eta = 0; best.mtl.type = 0;
    best.pos = v3(0, 0, 0); best.normal = v3(0, 0, 0);
    st.closest_rays++;
    for(Group &g : sc.groups){
        if(!group_box_hit(g, ro, rd, 1e-4f, best.t)) continue;
        for(const Obj &ob : g.objs){
            float t;
            if(ob.kind == 0){
                st.sphere_tests++;
                const RSphere &s = sc.spheres[ob.index];
                if(!cpu_sphere(s, ro, rd, 1e-4f, best.t, t)) continue;
                best.hit = true; best.t = t; best.mtl = s.mtl;
                V3 P = ro + rd * t;
                V3 n = gnormalize(v3(1.0f * (P.x - s.center.x), 1.0f * (P.y - s.center.y), 1.0f * (P.z - s.center.z)));
                if(dot(n, rd) > 0.0f) n = v3(-n.x, -n.y, -n.z);
                best.pos = P; best.normal = n; best.is_light = false;
            } else {
                st.tri_tests++;
                const RTriangle &tr = sc.tris[ob.index];
                if(!cpu_triangle(tr, ro, rd, 1e-4f, best.t, t)) continue;
                best.hit = true; best.t = t; best.mtl = tr.mtl;
                V3 P = ro + rd * t;
                V3 n = gnormalize(cross(tr.v1 - tr.v0, tr.v2 - tr.v0));
                if(dot(n, rd) > 0.0f) n = v3(-n.x, -n.y, -n.z);
                best.pos = P; best.normal = n; best.is_light = false;
            }
        }
    }
    for(int i = 0; i < sc.nl; ++i){
        const RLight &L = sc.lights[i];
        float t;
        st.sphere_tests++;
        if(intersect_sphere(ro, rd, L.light_ball.center, L.light_ball.r, t, best.t)){
            best.hit = true; best.t = t;
            best.mtl.base_color = L.illum; best.mtl.eta = 0.0f; best.mtl.roughness = 1.0f; best.mtl.metallic = 0.0f;
            best.pos = ro + rd * t;
            best.normal = normalize(best.pos - L.light_ball.center);
            best.is_light = true;
            if(dot(best.normal, rd) > 0.0f) best.normal = best.normal * -1.0f;
        }
    }
    return best;
}

// cpu_check_visibility, src/cpu_bdpt.cpp:82-107 (transparent objects never block)
bool bd_visible(BScene &sc, V3 p1, V3 p2, OracleBdptStats &st){
    V3 diff = p2 - p1;
    float dist = length(diff);
    V3 dir = diff / dist;
    float max_dist = dist - 1e-3f;
    st.shadow_rays++;
    for(Group &g : sc.groups){
        float tMin = 1e-3f, tMax = max_dist;
        if(!group_box_hit(g, p1, dir, tMin, tMax)) continue;
        for(const Obj &ob : g.objs){
            float t;
            if(ob.kind == 0){
                st.sphere_tests++;
                const RSphere &s = sc.spheres[ob.index];
                if(cpu_sphere(s, p1, dir, tMin, tMax, t) && s.mtl.eta <= 0.0f) return false;
            } else {
                st.tri_tests++;
                const RTriangle &tr = sc.tris[ob.index];
                if(cpu_triangle(tr, p1, dir, tMin, tMax, t) && tr.mtl.eta <= 0.0f) return false;
            }
        }
    }
    return true;
}

// cpu_calculate_mis_weight, src/cpu_bdpt.cpp:112-167
float bd_mis_weight(const EyeVertex *eye_path, int s_idx, const LightVertex *light_path, int t_idx,
                    V3 dir_e_to_l, float dist2, V3 camera_pos){
    if(s_idx < 0 || t_idx < 0) return 0.0f;
    const EyeVertex &ev = eye_path[s_idx];
    const LightVertex &lv = light_path[t_idx];
    V3 ns = normalize(ev.normal);
    V3 nt = normalize(lv.normal);
    float cos_s = fmaxf(0.0f, dot(ns, dir_e_to_l));
    float cos_t = fmaxf(0.0f, dot(nt, dir_e_to_l * -1.0f));
    if(cos_s <= 0.0f || cos_t <= 0.0f || dist2 < 1e-6f) return 0.0f;
    V3 wo_s = (s_idx == 0) ? normalize(camera_pos - ev.pos) : normalize(eye_path[s_idx - 1].pos - ev.pos);
    V3 wo_t = (t_idx == 0) ? normalize(lv.normal) : normalize(light_path[t_idx - 1].pos - lv.pos);
    float pdf_omega_s = fmaxf(bsdf_pdf(ev.mtl, wo_s, dir_e_to_l, ns), 1e-6f);
    float pdf_omega_t = fmaxf(bsdf_pdf(lv.mtl, wo_t, dir_e_to_l * -1.0f, nt), 1e-6f);
    float pdf_s_to_t = pdf_omega_s * cos_t / dist2;
    float pdf_t_to_s = pdf_omega_t * cos_s / dist2;
    float sum_ratios = 1.0f;
    float current_ratio = 1.0f;
    float prev_pdf_rev = pdf_t_to_s;
    for(int i = s_idx; i > 0; --i){
        if(eye_path[i].mtl.eta > 0.0f) break;
        current_ratio *= prev_pdf_rev / fmaxf(eye_path[i].pdf_fwd, 1e-8f);
        sum_ratios += current_ratio;
        prev_pdf_rev = eye_path[i].pdf_rev;
    }
    current_ratio = 1.0f;
    prev_pdf_rev = pdf_s_to_t;
    for(int i = t_idx; i > 0; --i){
        if(light_path[i].is_light_source){
            current_ratio *= prev_pdf_rev / fmaxf(light_path[i].pdf_fwd, 1e-8f);
            sum_ratios += current_ratio;
            break;
        }
        if(light_path[i].mtl.eta > 0.0f) break;
        current_ratio *= prev_pdf_rev / fmaxf(light_path[i].pdf_fwd, 1e-8f);
        sum_ratios += current_ratio;
        prev_pdf_rev = light_path[i].pdf_rev;
    }
    if(std::isnan(sum_ratios) || std::isinf(sum_ratios) || sum_ratios <= 0.0f) return 0.0f;
    return 1.0f / sum_ratios;
}

inline V3 clamp15(V3 c){                                // cpu_clamp_radiance, src/cpu_bdpt.cpp:18-25
    float m = std::max({ c.x, c.y, c.z });
    if(m > 15.0f){ float s = 15.0f / m; return v3(c.x * s, c.y * s, c.z * s); }
    return c;
}

} // namespace

extern "C" {

// camera4 = eye(3), look_at(3), view_up(3), fov_deg(1).  objects: kind (0 sphere, 1 triangle), index into
// the sphere/triangle arrays, group id -- in the scene file's insertion order.
int oracle_bdpt_render(const void *lights_v, int nl, const void *spheres_v, int ns, const void *tris_v, int nt,
                       const int *obj_kind, const int *obj_index, const int *obj_group, int nobj,
                       const float *camera10, float *image, int W, int H, int eye_depth, int light_depth,
                       int spp, int spl, const OracleBdptOpts *opts, OracleBdptStats *stats_out){
    (void) ns; (void) nt;
    if(!camera10 || !image || !opts || W <= 0 || H <= 0 || spp <= 0 || spl <= 0 || eye_depth <= 0 || light_depth <= 0) return 1;
    if(nl == 0) return 0;                                             // src/cpu_bdpt.cpp:178
    OracleBdptOpts o = *opts;
    if(o.max_delta <= 0) o.max_delta = o.rng_mode == 0 ? 64 : 10000;      // counter mode shares the GPU path's cap
    const int trig = o.rng_mode == 0 ? 0 : 1;                              // polynomial sincos with the counter streams
    BScene sc;
    sc.lights = (const RLight *) lights_v; sc.nl = nl;
    sc.spheres = (const RSphere *) spheres_v; sc.tris = (const RTriangle *) tris_v;
    {
        std::map<int, Group> gm;                                      // AABB::add_obj, src/object.cpp:123-146
        for(int i = 0; i < nobj; ++i){
            auto it = gm.find(obj_group[i]);
            if(it == gm.end()){ Group g; g.mn = v3(99999.f, 99999.f, 99999.f); g.mx = v3(-99999.f, -99999.f, -99999.f); it = gm.emplace(obj_group[i], g).first; }
            Group &g = it->second;
            if(obj_kind[i] == 0){
                const RSphere &s = sc.spheres[obj_index[i]];
                g.mn.x = std::min({ g.mn.x, s.center.x + s.r, s.center.x - s.r }); g.mx.x = std::max({ g.mx.x, s.center.x + s.r, s.center.x - s.r });
                g.mn.y = std::min({ g.mn.y, s.center.y + s.r, s.center.y - s.r }); g.mx.y = std::max({ g.mx.y, s.center.y + s.r, s.center.y - s.r });
                g.mn.z = std::min({ g.mn.z, s.center.z + s.r, s.center.z - s.r }); g.mx.z = std::max({ g.mx.z, s.center.z + s.r, s.center.z - s.r });
            } else {
                const RTriangle &t = sc.tris[obj_index[i]];
                const V3 *vv[3] = { &t.v0, &t.v1, &t.v2 };
                for(int k = 0; k < 3; ++k){
                    g.mn.x = std::min(g.mn.x, vv[k]->x); g.mn.y = std::min(g.mn.y, vv[k]->y); g.mn.z = std::min(g.mn.z, vv[k]->z);
                    g.mx.x = std::max(g.mx.x, vv[k]->x); g.mx.y = std::max(g.mx.y, vv[k]->y); g.mx.z = std::max(g.mx.z, vv[k]->z);
                }
            }
            Obj ob; ob.kind = obj_kind[i]; ob.index = obj_index[i];
            g.objs.push_back(ob);
        }
        for(auto &kv : gm) sc.groups.push_back(kv.second);
        // intersectAABB widens a degenerate axis every time it runs (src/object.cpp:108-111); start from the box it settles on
        for(Group &g : sc.groups){ float *mn = &g.mn.x, *mx = &g.mx.x;
            for(int a = 0; a < 3; ++a) for(int guard = 0; guard < 64 && mx[a] - mn[a] < 1e-6f; ++guard){ mn[a] -= 0.5f * 1e-6f; mx[a] += 0.5f * 1e-6f; } }
    }
    // scene bounds (parallel-light emission), src/cpu_bdpt.cpp:181-187
    V3 c_min = v3(1e9f, 1e9f, 1e9f), c_max = v3(-1e9f, -1e9f, -1e9f);
    for(const Group &g : sc.groups){
        c_min = v3(std::min(c_min.x, g.mn.x), std::min(c_min.y, g.mn.y), std::min(c_min.z, g.mn.z));
        c_max = v3(std::max(c_max.x, g.mx.x), std::max(c_max.y, g.mx.y), std::max(c_max.z, g.mx.z));
    }
    V3 min_bound = c_min, max_bound = c_max;
    // camera, src/cpu_bdpt.cpp:189-200 (uses the scene's own fov)
    V3 eye = v3(camera10[0], camera10[1], camera10[2]), look_at = v3(camera10[3], camera10[4], camera10[5]);
    V3 view_up = v3(camera10[6], camera10[7], camera10[8]);
    float aspect = float(W) / float(H);
    float theta = camera10[9] * kPi / 180.0f;
    float half_height = std::tan(theta / 2.0f);
    float half_width = aspect * half_height;
    V3 cw = gnormalize(eye - look_at);
    V3 cu = gnormalize(cross(view_up, cw));
    V3 cv = cross(cw, cu);
    V3 cUL = eye - smul(half_width, cu) + smul(half_height, cv) - cw;
    V3 cdx = smul(2.0f * half_width, cu) / float(W);
    V3 cdy = smul(-2.0f * half_height, cv) / float(H);
    V3 cam_eye = eye;

    int total_lights = nl;
    int total_light_paths = total_lights * spl;
    std::vector<LightVertex> light_vertices((size_t) total_light_paths * light_depth);
    for(auto &lv : light_vertices){ memset(&lv, 0, sizeof lv); }
    int nthreads = o.threads > 0 ? o.threads : omp_get_max_threads();
    std::vector<OracleBdptStats> tstats(nthreads);
    for(auto &s : tstats) memset(&s, 0, sizeof s);

    // ---- 1. light subpaths, src/cpu_bdpt.cpp:211-325 ----
#pragma omp parallel num_threads(nthreads)
    {
        OracleBdptStats &st = tstats[omp_get_thread_num()];
        Rng2 rng; rng.mode = o.rng_mode;
        if(o.rng_mode == 1) rng.mt.seed(1337 + omp_get_thread_num());
#pragma omp for schedule(dynamic, 64)
        for(int idx = 0; idx < total_light_paths; ++idx){
            if(o.rng_mode == 0) rng.pcg.seed(o.seed ^ 0x4C49474854ull, (uint32_t) idx, 0u);
            int light_idx = idx % total_lights;
            const RLight light = sc.lights[light_idx];
            int base = idx * light_depth;
            V3 ray_o, ray_d;
            float ray_eta = 1.0f;
            if(light.is_parallel){
                ray_d = normalize(light.dir);
                V3 scene_center = (min_bound + max_bound) * 0.5f;
                float scene_radius = length(max_bound - min_bound) * 0.5f;
                V3 w = ray_d;
                V3 u_vec = (fabsf(w.x) > 0.9f) ? v3(0, 1, 0) : v3(1, 0, 0);
                V3 v_vec = normalize(cross(w, u_vec));
                u_vec = normalize(cross(v_vec, w));
                float r1 = rng.next(), r2 = rng.next();
                float offset_u = (r1 - 0.5f) * scene_radius * 2.0f;
                float offset_v = (r2 - 0.5f) * scene_radius * 2.0f;
                ray_o = scene_center - ray_d * (scene_radius * 2.0f) + u_vec * offset_u + v_vec * offset_v;
            } else {
                ray_o = light.pos;
                V3 w = normalize(light.dir);
                V3 u_vec = (fabsf(w.x) > 0.9f) ? v3(0, 1, 0) : v3(1, 0, 0);
                V3 v_vec = normalize(cross(w, u_vec));
                u_vec = normalize(cross(v_vec, w));
                float u1 = rng.next(), u2 = rng.next();
                V3 local_dir;
                if(trig == 1){
                    float th = acosf(1.0f - u1 * (1.0f - cosf(light.cutoff)));
                    float phi = 2.0f * kPi * u2;
                    local_dir = v3(sinf(th) * cosf(phi), sinf(th) * sinf(phi), cosf(th));
                } else {
                    // counter mode (shared with the HIP kernels): cos(theta) directly, sin from cos,
                    // phi through the polynomial sincos -- no acosf/sinf/cosf
                    float cos_t = 1.0f - u1 * (1.0f - cosf(light.cutoff));
                    float sin_t = sqrtf(fmaxf(0.0f, 1.0f - cos_t * cos_t));
                    float sp, cp; sincos_2pi_poly(u2, sp, cp);
                    local_dir = v3(sin_t * cp, sin_t * sp, cos_t);
                }
                ray_d = normalize(u_vec * local_dir.x + v_vec * local_dir.y + w * local_dir.z);
                ray_o = ray_o + ray_d * light.light_ball.r;
            }
            V3 throughput = light.illum / fmaxf((float) spl, 1.0f);
            LightVertex &v0 = light_vertices[base];
            v0.pos = ray_o; v0.normal = ray_d; v0.throughput = throughput;
            v0.is_light_source = true; v0.source_cutoff = light.cutoff; v0.is_parallel = light.is_parallel != 0;
            V3 last_normal = ray_d, last_pos = ray_o;
            float last_pdf_omega = 1.0f / kPi;
            int deltas = 0;
            for(int depth = 1; depth < light_depth; depth++){
                LightVertex &vx = light_vertices[base + depth];
                vx.throughput = v3(0, 0, 0);
                Hit hit = bd_closest(sc, ray_o, ray_d, st);
                if(!hit.hit) break;
                if(hit.is_light){
                    vx.pos = hit.pos; vx.normal = hit.normal; vx.throughput = throughput; vx.mtl = hit.mtl;
                    vx.is_light_source = true; vx.source_cutoff = 0.0f; vx.is_parallel = false;
                    break;
                }
                if(length(throughput) < 1e-4f) break;
                float dist2 = dot(hit.pos - last_pos, hit.pos - last_pos);
                if(dist2 < 1e-6f) break;
                float cos_at_hit = fabsf(dot(hit.normal, ray_d * -1.0f));
                float cos_at_prev = fabsf(dot(last_normal, ray_d));
                float pdf_fwd = last_pdf_omega * cos_at_hit / dist2;
                V3 wo = ray_d * -1.0f;
                V3 wi, bsdf_val; float pdf_omega, new_eta; bool is_delta;
                float u_rr = rng.next(), u1 = rng.next(), u2 = rng.next();
                bsdf_sample(trig, hit.mtl, wo, hit.normal, u_rr, u1, u2, ray_eta, wi, bsdf_val, pdf_omega, is_delta, new_eta);
                if(pdf_omega <= 0.0f) break;            // non-delta: :299; delta: the TIR return (defined: terminate)
                if(is_delta){
                    throughput = throughput * bsdf_val;
                    ray_d = wi; ray_eta = new_eta;
                    ray_o = hit.pos + hit.normal * (dot(wi, hit.normal) < 0.0f ? -kEps : kEps);
                    if(++deltas > o.max_delta) break;
                    depth--;
                    continue;
                }
                vx.pos = hit.pos; vx.normal = hit.normal; vx.throughput = throughput; vx.mtl = hit.mtl; vx.is_light_source = false;
                float pdf_rev_omega = bsdf_pdf(hit.mtl, wi, wo, hit.normal);
                float pdf_rev = pdf_rev_omega * cos_at_prev / dist2;
                vx.pdf_fwd = pdf_fwd; vx.pdf_rev = pdf_rev;
                throughput = throughput * bsdf_val * fabsf(dot(hit.normal, wi)) / pdf_omega;
                if(!is_valid_color(throughput)) break;
                ray_d = wi;
                ray_o = hit.pos + hit.normal * kEps;
                last_pdf_omega = pdf_omega; last_normal = hit.normal; last_pos = hit.pos;
            }
        }
    }

    // ---- 2. eye subpaths + connections, src/cpu_bdpt.cpp:336-480 ----
    int x0 = 0, y0 = 0, x1 = W, y1 = H;
    if(o.rng_mode == 0){ x0 = std::max(o.x0, 0); y0 = std::max(o.y0, 0); if(o.x1 > 0) x1 = std::min(o.x1, W); if(o.y1 > 0) y1 = std::min(o.y1, H); }
    else if(o.y1 > 0) y1 = std::min(o.y1, H);
    const int ww = x1 - x0, wh = y1 - y0;
    const int n_lv = total_light_paths * light_depth;
#pragma omp parallel num_threads(nthreads)
    {
        OracleBdptStats &st = tstats[omp_get_thread_num()];
        Rng2 rng; rng.mode = o.rng_mode;
        if(o.rng_mode == 1) rng.mt.seed(9999 + omp_get_thread_num());
        std::vector<EyeVertex> eye_path(eye_depth);
#pragma omp for schedule(dynamic, 16)
        for(int wi_ = 0; wi_ < ww * wh; ++wi_){
            int px = x0 + wi_ % ww, py = y0 + wi_ / ww;
            int idx = py * W + px;
            V3 accum = v3(0, 0, 0);
            for(int s = 0; s < spp; ++s){
                if(o.rng_mode == 0) rng.pcg.seed(o.seed, (uint32_t) idx, (uint32_t) s);
                st.samples++;
                float pixel_x = (float) px + rng.next();
                float pixel_y = (float) py + rng.next();
                V3 pixel_pos = cUL + cdx * pixel_x + cdy * pixel_y;
                V3 ray_o = cam_eye;
                V3 ray_d = normalize(pixel_pos - cam_eye);
                float ray_eta = 1.0f;
                V3 throughput = v3(1.0f, 1.0f, 1.0f);
                V3 last_normal = ray_d, last_pos = cam_eye;
                float last_pdf_omega = 1.0f;
                V3 final_color = v3(0, 0, 0);
                int deltas = 0;
                for(int depth = 0; depth < eye_depth; depth++){
                    EyeVertex &vx = eye_path[depth];
                    vx.throughput = v3(0, 0, 0);
                    Hit hit = bd_closest(sc, ray_o, ray_d, st);
                    if(!hit.hit) break;
                    if(hit.is_light && depth == 0){ final_color = final_color + hit.mtl.base_color; break; }
                    float pdf_fwd = 1.0f;
                    if(depth > 0){
                        float d2 = dot(hit.pos - last_pos, hit.pos - last_pos);
                        float cos_at_hit = fabsf(dot(hit.normal, ray_d * -1.0f));
                        pdf_fwd = last_pdf_omega * cos_at_hit / fmaxf(d2, 1e-6f);
                    }
                    vx.pos = hit.pos; vx.normal = hit.normal; vx.throughput = throughput; vx.mtl = hit.mtl; vx.pdf_fwd = 0.0f; vx.pdf_rev = 1.0f;

                    V3 total_L = v3(0, 0, 0);
                    for(int li = 0; li < n_lv; li++){
                        const LightVertex &lv = light_vertices[li];
                        if(length(lv.throughput) < 1e-6f) continue;
                        V3 d_vec = lv.pos - vx.pos;
                        float dist2 = dot(d_vec, d_vec);
                        if(dist2 < 1e-6f) continue;
                        float dist = sqrtf(dist2);
                        V3 wi = d_vec / dist;
                        float cosE = fmaxf(0.0f, dot(vx.normal, wi));
                        float cosL = fmaxf(0.0f, dot(lv.normal, wi * -1.0f));
                        if(cosE <= 0.0f || cosL <= 0.0f) continue;
                        int t_idx = li % light_depth;
                        if(lv.is_light_source && lv.source_cutoff > 0.0f && !lv.is_parallel){
                            int real_light = (li / light_depth) % total_lights;
                            V3 light_dir = normalize(sc.lights[real_light].dir);
                            if(dot(light_dir, wi * -1.0f) < cosf(lv.source_cutoff)) continue;
                        }
                        V3 wo_e = ray_d * -1.0f;
                        V3 fE = bsdf_evaluate(vx.mtl, wo_e, wi, vx.normal);
                        V3 fL = v3(1.0f, 1.0f, 1.0f);
                        if(!lv.is_light_source && t_idx > 0){
                            V3 prev = light_vertices[li - 1].pos;
                            V3 wo_l = normalize(prev - lv.pos);
                            fL = bsdf_evaluate(lv.mtl, wo_l, wi * -1.0f, lv.normal);
                        }
                        if((fE.x <= 0.0f && fE.y <= 0.0f && fE.z <= 0.0f) || (fL.x <= 0.0f && fL.y <= 0.0f && fL.z <= 0.0f)) continue;
                        st.connections++;
                        if(!bd_visible(sc, vx.pos + vx.normal * kEps, lv.pos + lv.normal * kEps, st)) continue;
                        float G = (cosE * cosL) / fmaxf(dist2, 1e-4f);
                        const LightVertex *lp_base = &light_vertices[(size_t) (li / light_depth) * light_depth];
                        float mis_w = bd_mis_weight(eye_path.data(), depth, lp_base, t_idx, d_vec, dist2, cam_eye);
                        V3 contrib = vx.throughput * fE * G * fL * lv.throughput * v3(1.0f, 1.0f, 1.0f) * mis_w;
                        if(is_valid_color(contrib)) total_L = total_L + clamp15(contrib);
                    }
                    final_color = final_color + total_L;

                    V3 wo = ray_d * -1.0f;
                    V3 wi, bsdf_val; float pdf_omega, new_eta; bool is_delta;
                    float u_rr = rng.next(), u1 = rng.next(), u2 = rng.next();
                    bsdf_sample(trig, hit.mtl, wo, hit.normal, u_rr, u1, u2, ray_eta, wi, bsdf_val, pdf_omega, is_delta, new_eta);
                    if(pdf_omega <= 0.0f) break;
                    if(is_delta){
                        throughput = throughput * bsdf_val;
                        ray_d = wi; ray_eta = new_eta;
                        ray_o = hit.pos + hit.normal * (dot(wi, hit.normal) < 0.0f ? -kEps : kEps);
                        last_pos = hit.pos; last_normal = hit.normal; last_pdf_omega = 1.0f;
                        if(++deltas > o.max_delta) break;
                        depth--; continue;
                    }
                    float pdf_rev_omega = bsdf_pdf(hit.mtl, wi, wo, hit.normal);
                    float d2 = dot(hit.pos - last_pos, hit.pos - last_pos);
                    float cos_at_prev = fabsf(dot(last_normal, ray_d));
                    vx.pdf_fwd = pdf_fwd;
                    vx.pdf_rev = pdf_rev_omega * cos_at_prev / fmaxf(d2, 1e-6f);
                    throughput = throughput * bsdf_val * fabsf(dot(hit.normal, wi)) / pdf_omega;
                    if(!is_valid_color(throughput)) break;
                    ray_d = wi;
                    ray_o = hit.pos + hit.normal * kEps;
                    last_pdf_omega = pdf_omega; last_normal = hit.normal; last_pos = hit.pos;
                }
                if(!is_valid_color(final_color)) final_color = v3(0, 0, 0);
                accum = accum + final_color;
            }
            V3 out = accum / (float) spp;
            image[3 * (size_t) idx + 0] = out.x; image[3 * (size_t) idx + 1] = out.y; image[3 * (size_t) idx + 2] = out.z;
        }
    }
    if(stats_out){
        OracleBdptStats t; memset(&t, 0, sizeof t);
        for(auto &s : tstats){
            t.samples += s.samples; t.closest_rays += s.closest_rays; t.shadow_rays += s.shadow_rays;
            t.connections += s.connections; t.tri_tests += s.tri_tests; t.sphere_tests += s.sphere_tests;
        }
        *stats_out = t;
    }
    return 0;
}

} // extern "C"
