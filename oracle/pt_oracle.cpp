// TEST INFRASTRUCTURE ONLY -- CPU oracle: unidirectional path tracer with NEE.
//
// Restates the per-sample loop of the reference's CUDA kernel
// (reference: src/pt_cu.cu:20-250, cuda_path_trace_kernel) on the CPU, with the
// reference's brute-force scene scans (include/geometric.cuh:293-388).  It is the
// checker for the HIP kernels; nothing in path_tracing_amd/ links or calls it.
//
// PARITY STATUS: parity unpinned (see ref_math.hpp header and tests/golden/README.md).
//
// Behaviour the reference leaves undefined, defined here exactly as in DESIGN.md:
//  * shadow rays (check_visibility reads uninitialised mtl_old, geometric.cuh:310-311,
//    319-320): a blocker is opaque iff its mtl.eta <= 0, otherwise its
//    transmittance is 1 (what src/cpu_bdpt.cpp:102 does).  opts.glass_shadow_opaque=1
//    makes every blocker opaque (the "zeroed mtl_old" reading).
//  * delta bounces are free and uncapped (pt_cu.cu:228): capped at opts.max_delta.
//  * bsdf_sample's total-internal-reflection return (geometric.cuh:514) ends the path.
//  * the three bsdf_sample uniforms are drawn in the order u_rr, u1, u2 (pt_cu.cu:210-212).
//  * uniforms are in [0,1) (cuRAND's are in (0,1]; the stream is unseeded there anyway).
#include "ref_math.hpp"
#include <algorithm>
#include <random>
#include <vector>
#include <omp.h>

using namespace orc;

extern "C" {

struct OraclePtOpts {
    uint64_t seed;          // counter mode: stream key; mt mode: per-pixel seed base (seed + pixel index)
    int rng_mode;           // 0 = PCG counter streams (GPU-compatible), 1 = std::mt19937 per pixel
    int trig_mode;          // 0 = shared polynomial sincos, 1 = glibc sinf/cosf
    int sample_offset;      // first global sample index (counter mode)
    int x0, y0, x1, y1;     // render window [x0,x1) x [y0,y1) inside the W x H image
    int threads;            // OpenMP threads (0 = default)
    int glass_shadow_opaque;
    int max_delta;          // cap on free delta bounces per sample
    int output_sum;         // 1: write the per-pixel sum over samples instead of the mean
    int russian_roulette;   // 1: unbiased roulette after every non-delta bounce (NOT in the reference; default off)
    int ball_draw_reversed; // 1: the three uniforms of a unit-ball try fill z,y,x (argument evaluation
                            //    order of make_float3(u(),u(),u()) is unspecified, geometric.cuh:410)
};

struct OraclePtStats {
    uint64_t samples, closest_rays, shadow_rays, bounces, delta_bounces, tri_tests, sphere_tests;
    // host walk of the library's exported tree (oracle_pt_render_bvh only): child boxes slab-tested (2 per inner-node
    // visit) and triangle tests, by closest-hit and by shadow rays -- what hpt_stats.boxes_* / tris_* must equal
    uint64_t boxes_closest, tris_closest, boxes_shadow, tris_shadow;
};

// The tree libhpt.so built for the scene (include/hpt.h, hpt_bvh_info / hpt_bvh_export_host): the reference has no
// acceleration structure, so this is not restated from it -- the walk below is the independent count SURVEY 8(d) asks for.
struct OracleBvh {
    const uint32_t *qnodes; int num_nodes;     // 8 words per node
    const uint32_t *tris; int num_tris;        // 12 words per triangle, leaf order; word 3 = scan ordinal
    int num_rounds;                            // spheres + light balls = ordinal of input triangle 0
    float qorigin[3], qscale[3];
};

} // extern "C"

namespace {

struct Rng {
    int mode;
    Pcg pcg;
    std::mt19937 mt;
    std::uniform_real_distribution<float> U{0.0f, 1.0f};
    float next(){ return mode == 0 ? pcg.next() : U(mt); }
};

struct Scene {
    const RLight *lights; int nl;
    const RSphere *spheres; int ns;
    const RTriangle *tris; int nt;
    std::vector<float> cos_cutoff;     // cosf(light.cutoff), hoisted (pt_cu.cu:73,79,168)
    const OracleBvh *bvh = nullptr;    // non-null: triangles are found by walking the library's tree instead of the scan
};

// ---- host walk of the exported tree ----------------------------------------------------------
// Visits what one ray of the device's plain traversal visits, in its order (path_tracing_amd/csrc/pt_kernels.hip,
// trace_chunk without a budget): a node step slab-tests both children's quantised boxes, descends into the nearer hit
// child and stacks the other; a leaf tests all of its triangles; the walk ends on an empty stack.  The primitive test is
// the oracle's own intersect_triangle on the INPUT triangle the leaf slot's ordinal names, so the hit found is what the
// brute-force scan finds as long as the boxes are conservative -- and the scan's strict '<' tie rule is kept by
// preferring the lower ordinal at equal t.
constexpr uint32_t kLeaf = 0x80000000u, kNoChild = 0xFFFFFFFFu, kDone = 0xFFFFFFFEu;

struct WalkRay {
    float ix, iy, iz, ox, oy, oz;
    WalkRay(const OracleBvh &b, V3 ro, V3 rd){
        float dx = fabsf(rd.x) > 1e-20f ? rd.x : copysignf(1e-20f, rd.x);
        float dy = fabsf(rd.y) > 1e-20f ? rd.y : copysignf(1e-20f, rd.y);
        float dz = fabsf(rd.z) > 1e-20f ? rd.z : copysignf(1e-20f, rd.z);
        ix = 1.0f / dx; iy = 1.0f / dy; iz = 1.0f / dz;
        ox = (b.qorigin[0] - ro.x) * ix; oy = (b.qorigin[1] - ro.y) * iy; oz = (b.qorigin[2] - ro.z) * iz;
        ix *= b.qscale[0]; iy *= b.qscale[1]; iz *= b.qscale[2];
    }
};

// one inner-node step: returns the next code and updates the stack
static inline uint32_t node_step(const OracleBvh &b, const WalkRay &r, uint32_t cur, float limit, uint32_t *stk, int &sp){
    const uint32_t *w = b.qnodes + (size_t) cur * 8;
    // words 0-5: per axis the lower planes of (left | right << 16), then the upper planes (include/hpt.h, hpt_bvh_export_host)
    float a0 = fmaf((float) (w[0] & 0xFFFFu), r.ix, r.ox), a1 = fmaf((float) (w[1] & 0xFFFFu), r.ix, r.ox);
    float b0 = fmaf((float) (w[2] & 0xFFFFu), r.iy, r.oy), b1 = fmaf((float) (w[3] & 0xFFFFu), r.iy, r.oy);
    float c0 = fmaf((float) (w[4] & 0xFFFFu), r.iz, r.oz), c1 = fmaf((float) (w[5] & 0xFFFFu), r.iz, r.oz);
    float ln = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fmaxf(fminf(c0, c1), 0.0f));
    float lf = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fminf(fmaxf(c0, c1), limit));
    a0 = fmaf((float) (w[0] >> 16), r.ix, r.ox); a1 = fmaf((float) (w[1] >> 16), r.ix, r.ox);
    b0 = fmaf((float) (w[2] >> 16), r.iy, r.oy); b1 = fmaf((float) (w[3] >> 16), r.iy, r.oy);
    c0 = fmaf((float) (w[4] >> 16), r.iz, r.oz); c1 = fmaf((float) (w[5] >> 16), r.iz, r.oz);
    float rn = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fmaxf(fminf(c0, c1), 0.0f));
    float rf = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fminf(fmaxf(c0, c1), limit));
    uint32_t lc = w[6], rc = w[7];
    bool hl = (ln <= lf * 1.000002f) && (lc != kNoChild);
    bool hr = (rn <= rf * 1.000002f) && (rc != kNoChild);
    bool any = hl || hr, both = hl && hr;
    bool left_first = hl && (!hr || ln <= rn);
    if(both){ stk[sp++] = left_first ? rc : lc; return left_first ? lc : rc; }
    if(any) return left_first ? lc : rc;
    if(sp > 0) return stk[--sp];
    return kDone;
}

// closest triangle hit below best_t (ties: lower ordinal); returns the INPUT index of the triangle or -1
int walk_closest(const Scene &sc, V3 ro, V3 rd, float &best_t, uint32_t best_ord, uint64_t &boxes, uint64_t &tris){
    const OracleBvh &b = *sc.bvh;
    WalkRay r(b, ro, rd);
    uint32_t stk[128]; int sp = 0;
    uint32_t cur = 0u;
    int found = -1;
    for(;;){
        while(!(cur & kLeaf)){ boxes += 2; cur = node_step(b, r, cur, best_t, stk, sp); }
        if(cur == kDone) break;
        uint32_t first = (cur & 0x7FFFFFFFu) >> 3, cnt = (cur & 7u) + 1u;
        for(uint32_t k = 0; k < cnt; ++k){
            tris += 1;
            uint32_t ord = b.tris[(size_t) (first + k) * 12 + 3];
            const RTriangle &tr = sc.tris[ord - (uint32_t) b.num_rounds];
            float t;
            if(intersect_triangle(ro, rd, tr.v0, tr.v1, tr.v2, t, 1e20f) && (t < best_t || (t == best_t && ord < best_ord))){
                best_t = t; best_ord = ord; found = (int) (ord - (uint32_t) b.num_rounds);
            }
        }
        if(sp > 0) cur = stk[--sp]; else break;
    }
    return found;
}

// true when an opaque triangle blocks (1e-3, max_d)
bool walk_any(const Scene &sc, V3 p1, V3 dir, float max_d, bool glass_opaque, uint64_t &boxes, uint64_t &tris){
    const OracleBvh &b = *sc.bvh;
    WalkRay r(b, p1, dir);
    uint32_t stk[128]; int sp = 0;
    uint32_t cur = 0u;
    for(;;){
        while(!(cur & kLeaf)){ boxes += 2; cur = node_step(b, r, cur, max_d, stk, sp); }
        if(cur == kDone) break;
        uint32_t first = (cur & 0x7FFFFFFFu) >> 3, cnt = (cur & 7u) + 1u;
        bool blocked = false;
        for(uint32_t k = 0; k < cnt; ++k){            // the device tests every triangle of the leaf before it looks
            tris += 1;
            uint32_t ord = b.tris[(size_t) (first + k) * 12 + 3];
            const RTriangle &tr = sc.tris[ord - (uint32_t) b.num_rounds];
            float t;
            if(intersect_triangle(p1, dir, tr.v0, tr.v1, tr.v2, t, max_d) && t > 1e-3f && (glass_opaque || tr.mtl.eta <= 0.0f)) blocked = true;
        }
        if(blocked) return true;
        if(sp > 0) cur = stk[--sp]; else break;
    }
    return false;
}

// geometric.cuh:327-388 -- spheres, then light balls, then triangles; strict '<'
Hit closest_hit(const Scene &sc, V3 ro, V3 rd, OraclePtStats &st){
    Hit best; best.hit = false; best.t = 1e20f; best.is_light = false; best.prim = -1;
    best.mtl.base_color = v3(0, 0, 0); best.mtl.roughness = 0; best.mtl.metallic = 0; best.mtl.eta = 0; best.mtl.type = 0;
    best.pos = v3(0, 0, 0); best.normal = v3(0, 0, 0);
    float t; const float max_dist = 1e20f;
    st.sphere_tests += (uint64_t) (sc.ns + sc.nl);
    if(!sc.bvh) st.tri_tests += (uint64_t) sc.nt;
    for(int i = 0; i < sc.ns; ++i){
        const RSphere &s = sc.spheres[i];
        if(intersect_sphere(ro, rd, s.center, s.r, t, max_dist) && t < best.t){
            best.hit = true; best.t = t; best.mtl = s.mtl;
            best.pos = ro + rd * t;
            best.normal = normalize(best.pos - s.center);
            best.is_light = false; best.prim = i;
            if(dot(best.normal, rd) > 0.0f) best.normal = best.normal * -1.0f;
        }
    }
    for(int i = 0; i < sc.nl; ++i){
        const RSphere &s = sc.lights[i].light_ball;
        if(intersect_sphere(ro, rd, s.center, s.r, t, max_dist) && t < best.t){
            best.hit = true; best.t = t;
            best.mtl.base_color = sc.lights[i].illum;          // only field set (geometric.cuh:360-361)
            best.pos = ro + rd * t;
            best.normal = normalize(best.pos - s.center);
            best.is_light = true; best.prim = sc.ns + i;
            if(dot(best.normal, rd) > 0.0f) best.normal = best.normal * -1.0f;
        }
    }
    if(sc.bvh){
        // same answer through the library's tree (a sphere or light ball at equal t keeps the hit: lower ordinal)
        float bt = best.t;
        int i = walk_closest(sc, ro, rd, bt, best.hit ? (uint32_t) best.prim : 0xFFFFFFFFu, st.boxes_closest, st.tris_closest);
        if(i >= 0){
            const RTriangle &tr = sc.tris[i];
            best.hit = true; best.t = bt; best.mtl = tr.mtl;
            best.pos = ro + rd * bt;
            best.normal = normalize(cross(tr.v1 - tr.v0, tr.v2 - tr.v0));
            best.is_light = false; best.prim = sc.ns + sc.nl + i;
            if(dot(best.normal, rd) > 0.0f) best.normal = best.normal * -1.0f;
        }
        return best;
    }
    for(int i = 0; i < sc.nt; ++i){
        const RTriangle &tr = sc.tris[i];
        if(intersect_triangle(ro, rd, tr.v0, tr.v1, tr.v2, t, max_dist) && t < best.t){
            best.hit = true; best.t = t; best.mtl = tr.mtl;
            best.pos = ro + rd * t;
            best.normal = normalize(cross(tr.v1 - tr.v0, tr.v2 - tr.v0));
            best.is_light = false; best.prim = sc.ns + sc.nl + i;
            if(dot(best.normal, rd) > 0.0f) best.normal = best.normal * -1.0f;
        }
    }
    return best;
}

// geometric.cuh:293-325 -- triangles then spheres (not light balls); range (1e-3, dist-1e-3)
bool visible(const Scene &sc, V3 p1, V3 p2, bool glass_opaque, OraclePtStats &st){
    V3 diff = p2 - p1;
    float dist = length(diff);
    V3 dir = diff / dist;
    float max_d = dist - 1e-3f;
    const float min_d = 1e-3f;
    float t;
    if(sc.bvh){
        // the device's order: spheres first (a blocked ray never enters the tree), then the any-hit walk
        st.sphere_tests += (uint64_t) sc.ns;
        bool blocked = false;
        for(int i = 0; i < sc.ns; ++i){
            const RSphere &s = sc.spheres[i];
            if(intersect_sphere(p1, dir, s.center, s.r, t, max_d) && t > min_d && (glass_opaque || s.mtl.eta <= 0.0f)) blocked = true;
        }
        if(blocked) return false;
        return !walk_any(sc, p1, dir, max_d, glass_opaque, st.boxes_shadow, st.tris_shadow);
    }
    st.tri_tests += (uint64_t) sc.nt;
    st.sphere_tests += (uint64_t) sc.ns;
    for(int i = 0; i < sc.nt; ++i){
        const RTriangle &tr = sc.tris[i];
        if(intersect_triangle(p1, dir, tr.v0, tr.v1, tr.v2, t, max_d) && t > min_d){
            if(glass_opaque || tr.mtl.eta <= 0.0f) return false;
        }
    }
    for(int i = 0; i < sc.ns; ++i){
        const RSphere &s = sc.spheres[i];
        if(intersect_sphere(p1, dir, s.center, s.r, t, max_d) && t > min_d){
            if(glass_opaque || s.mtl.eta <= 0.0f) return false;
        }
    }
    return true;
}

// geometric.cuh:407-413
V3 random_in_unit_sphere(Rng &rng, bool reversed){
    V3 p;
    do {
        float a = rng.next(), b = rng.next(), c = rng.next();
        p = (reversed ? v3(c, b, a) : v3(a, b, c)) * 2.0f - v3(1.0f, 1.0f, 1.0f);
    } while(dot(p, p) >= 1.0f);
    return p;
}

// one camera sample, pt_cu.cu:36-245
V3 trace_sample(const Scene &sc, const RCamera &cam, int px, int py, int max_depth,
                const OraclePtOpts &o, Rng &rng, OraclePtStats &st){
    V3 final_color = v3(0, 0, 0);
    float pixel_x = (float) px + rng.next();
    float pixel_y = (float) py + rng.next();
    V3 ray_o = cam.eye;
    V3 pixel_pos = cam.UL + cam.dx * pixel_x + cam.dy * pixel_y;
    V3 ray_d = normalize(pixel_pos - ray_o);
    float ray_eta = 1.0f;
    V3 throughput = v3(1.0f, 1.0f, 1.0f);
    bool last_is_delta = true;
    int delta_count = 0;
    st.samples++;

    for(int depth = 0; depth < max_depth; ++depth){
        Hit hit = closest_hit(sc, ray_o, ray_d, st);
        st.closest_rays++;
        if(!hit.hit) break;
        V3 wo = ray_d * -1.0f;

        if(hit.is_light){                                   // pt_cu.cu:59-122
            V3 emission = hit.mtl.base_color;
            float area = 1.0f, cone_ratio = 1.0f;
            bool valid_light = false;
            for(int i = 0; i < sc.nl; ++i){
                const RLight &L = sc.lights[i];
                V3 c2h = hit.pos - L.pos;
                if(fabsf(length(c2h) - L.light_ball.r) < 1e-2f){
                    valid_light = true;
                    area = 4.0f * kPi * L.light_ball.r * L.light_ball.r;
                    if(L.cutoff > 0.0f && !L.is_parallel){
                        cone_ratio = (1.0f - sc.cos_cutoff[i]) / 2.0f;
                        V3 main_dir = normalize(L.dir);
                        if(depth == 0) cone_ratio = 1.f;
                        else if(dot(main_dir, normalize(c2h)) < sc.cos_cutoff[i]) cone_ratio = 0.0f;
                    }
                    break;
                }
            }
            if(valid_light && cone_ratio > 0.0f) emission = emission / (area * cone_ratio);
            else emission = v3(0, 0, 0);
            if(emission.x > 0.0f || emission.y > 0.0f || emission.z > 0.0f){
                if(last_is_delta){
                    V3 contrib = throughput * emission;
                    if(is_valid_color(contrib)) final_color = final_color + clamp_radiance(contrib, 15.0f);
                }
                // else: the reference's MIS branch is a stub with pdf_light_dir = 0 (pt_cu.cu:103-118)
            }
            break;
        }

        // next-event estimation, pt_cu.cu:125-202
        if(hit.mtl.eta <= 0.0f && (hit.mtl.metallic < 0.99f || hit.mtl.roughness > 0.01f) && sc.nl > 0){
            int l_idx = std::min((int) (rng.next() * sc.nl), sc.nl - 1);
            const RLight &light = sc.lights[l_idx];
            if(light.is_parallel){
                V3 light_dir = normalize(light.dir * -1.0f);
                float cos_surface = fmaxf(0.0f, dot(hit.normal, light_dir));
                if(cos_surface > 0.0f){
                    st.shadow_rays++;
                    if(visible(sc, hit.pos + hit.normal * kEps, hit.pos + light_dir * 1e4f, o.glass_shadow_opaque != 0, st)){
                        V3 brdf = bsdf_evaluate(hit.mtl, wo, light_dir, hit.normal);
                        V3 transmittance = v3(1.0f, 1.0f, 1.0f);
                        V3 contrib = throughput * brdf * light.illum * transmittance * cos_surface * (float) sc.nl;
                        if(is_valid_color(contrib)) final_color = final_color + clamp_radiance(contrib, 15.0f);
                    }
                }
            } else {
                V3 d_local = random_in_unit_sphere(rng, o.ball_draw_reversed != 0);
                if(length(d_local) > 0.001f) d_local = normalize(d_local);
                else d_local = v3(0, 1, 0);
                V3 light_pos = light.pos + d_local * light.light_ball.r;
                V3 wi_light = light_pos - hit.pos;
                float dist2 = dot(wi_light, wi_light);
                float dist = sqrtf(dist2);
                wi_light = wi_light / dist;
                float cos_surface = fmaxf(0.0f, dot(hit.normal, wi_light));
                float cos_light = fmaxf(0.0f, dot(d_local, wi_light * -1.0f));
                if(cos_surface > 0.0f && cos_light > 0.0f){
                    bool inside_cone = true;
                    if(light.cutoff > 0.0f && !light.is_parallel){
                        V3 main_dir = normalize(light.dir);
                        if(dot(main_dir, wi_light * -1.0f) < sc.cos_cutoff[l_idx]) inside_cone = false;
                    }
                    if(inside_cone){
                        st.shadow_rays++;
                        if(visible(sc, hit.pos + hit.normal * kEps, light_pos + d_local * kEps, o.glass_shadow_opaque != 0, st)){
                            float area = 4.0f * kPi * light.light_ball.r * light.light_ball.r;
                            float pdf_light_area = 1.0f / (sc.nl * area);
                            float pdf_light_dir = pdf_light_area * dist2 / fmaxf(cos_light, 1e-6f);
                            float pdf_bsdf = bsdf_pdf(hit.mtl, wo, wi_light, hit.normal);
                            float p_l = pdf_light_dir * pdf_light_dir;
                            float p_b = pdf_bsdf * pdf_bsdf;
                            float mis_w = p_l / fmaxf(p_l + p_b, 1e-8f);
                            V3 brdf = bsdf_evaluate(hit.mtl, wo, wi_light, hit.normal);
                            V3 transmittance = v3(1.0f, 1.0f, 1.0f);
                            V3 contrib = throughput * brdf * light.illum * transmittance * cos_surface / pdf_light_dir * mis_w;
                            if(is_valid_color(contrib)) final_color = final_color + clamp_radiance(contrib, 15.0f);
                        }
                    }
                }
            }
        }

        // BSDF sampling, pt_cu.cu:204-241
        V3 wi, bsdf_val; float pdf_omega, new_eta; bool is_delta;
        float u_rr = rng.next(), u1 = rng.next(), u2 = rng.next();
        bsdf_sample(o.trig_mode, hit.mtl, wo, hit.normal, u_rr, u1, u2, ray_eta, wi, bsdf_val, pdf_omega, is_delta, new_eta);
        if(pdf_omega <= 0.0f) break;        // non-delta: pt_cu.cu:214; delta: the TIR return (defined: terminate)
        st.bounces++;
        if(is_delta){
            throughput = throughput * bsdf_val;
            ray_d = wi;
            ray_eta = new_eta;
            if(dot(wi, hit.normal) < 0.0f) ray_o = hit.pos - hit.normal * kEps;
            else ray_o = hit.pos + hit.normal * kEps;
            last_is_delta = true;
            if(!is_valid_color(throughput)) break;
            st.delta_bounces++;
            if(++delta_count > o.max_delta) break;
            depth--;
            continue;
        }
        float cos_wi = fabsf(dot(hit.normal, wi));
        throughput = throughput * bsdf_val * cos_wi / pdf_omega;
        if(!is_valid_color(throughput)) break;
        if(o.russian_roulette){
            // survive with q = clamp(max throughput channel, 0.05, 1); survivors are scaled by 1/q
            float q = fminf(1.0f, fmaxf(0.05f, fmaxf(throughput.x, fmaxf(throughput.y, throughput.z))));
            float u = rng.next();
            if(!(u < q)) break;
            throughput = throughput / q;
        }
        ray_d = wi;
        ray_o = hit.pos + hit.normal * kEps;
        last_is_delta = false;
    }
    if(!is_valid_color(final_color)) final_color = v3(0, 0, 0);
    return final_color;
}

} // namespace

extern "C" {

// Renders the window [x0,x1)x[y0,y1) of a W x H image; pixels outside are left untouched.
// Arguments mirror pt_render_wrapper (reference: include/pt_cu.cuh:6-13).
int oracle_pt_render_bvh(const void *lights, int nl, const void *spheres, int ns, const void *tris, int nt,
                         const void *camera, float *image, int W, int H, int max_depth, int spp,
                         const OraclePtOpts *opts, OraclePtStats *stats_out, const OracleBvh *bvh);

int oracle_pt_render(const void *lights, int nl, const void *spheres, int ns, const void *tris, int nt,
                     const void *camera, float *image, int W, int H, int max_depth, int spp,
                     const OraclePtOpts *opts, OraclePtStats *stats_out){
    return oracle_pt_render_bvh(lights, nl, spheres, ns, tris, nt, camera, image, W, H, max_depth, spp, opts, stats_out, nullptr);
}

// bvh != null: the same estimator with the triangles found through the library's exported tree, counting the walk
int oracle_pt_render_bvh(const void *lights, int nl, const void *spheres, int ns, const void *tris, int nt,
                         const void *camera, float *image, int W, int H, int max_depth, int spp,
                         const OraclePtOpts *opts, OraclePtStats *stats_out, const OracleBvh *bvh){
    if(!camera || !image || !opts || W <= 0 || H <= 0 || spp <= 0) return 1;
    if(bvh && (bvh->num_tris != nt || bvh->num_rounds != ns + nl || bvh->num_nodes < 1 || !bvh->qnodes || (nt > 0 && !bvh->tris))) return 2;
    Scene sc;
    sc.bvh = bvh;
    sc.lights = (const RLight *) lights; sc.nl = nl;
    sc.spheres = (const RSphere *) spheres; sc.ns = ns;
    sc.tris = (const RTriangle *) tris; sc.nt = nt;
    sc.cos_cutoff.resize(nl > 0 ? nl : 0);
    for(int i = 0; i < nl; ++i) sc.cos_cutoff[i] = cosf(sc.lights[i].cutoff);
    RCamera cam; memcpy(&cam, camera, sizeof cam);
    OraclePtOpts o = *opts;
    int x1 = o.x1 > 0 ? std::min(o.x1, W) : W, y1 = o.y1 > 0 ? std::min(o.y1, H) : H;
    int x0 = std::max(o.x0, 0), y0 = std::max(o.y0, 0);
    if(o.max_delta <= 0) o.max_delta = 64;
    int nthreads = o.threads > 0 ? o.threads : omp_get_max_threads();
    int ww = x1 - x0, wh = y1 - y0;
    if(ww <= 0 || wh <= 0) return 1;
    std::vector<OraclePtStats> tstats(nthreads);
    for(auto &s : tstats) memset(&s, 0, sizeof s);

#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
    for(int wi = 0; wi < ww * wh; ++wi){
        int px = x0 + wi % ww, py = y0 + wi / ww;
        int idx = py * W + px;
        OraclePtStats &st = tstats[omp_get_thread_num()];
        Rng rng; rng.mode = o.rng_mode;
        if(o.rng_mode == 1) rng.mt.seed((uint32_t) (o.seed + (uint64_t) idx));
        V3 color_sum = v3(0, 0, 0);
        for(int s = 0; s < spp; ++s){
            if(o.rng_mode == 0) rng.pcg.seed(o.seed, (uint32_t) idx, (uint32_t) (o.sample_offset + s));
            V3 c = trace_sample(sc, cam, px, py, max_depth, o, rng, st);
            color_sum = color_sum + c;                       // pt_cu.cu:245
        }
        V3 out = o.output_sum ? color_sum : color_sum / (float) spp;     // pt_cu.cu:248
        image[3 * (size_t) idx + 0] = out.x;
        image[3 * (size_t) idx + 1] = out.y;
        image[3 * (size_t) idx + 2] = out.z;
    }
    if(stats_out){
        OraclePtStats t; memset(&t, 0, sizeof t);
        for(auto &s : tstats){
            t.samples += s.samples; t.closest_rays += s.closest_rays; t.shadow_rays += s.shadow_rays;
            t.bounces += s.bounces; t.delta_bounces += s.delta_bounces; t.tri_tests += s.tri_tests;
            t.sphere_tests += s.sphere_tests;
            t.boxes_closest += s.boxes_closest; t.tris_closest += s.tris_closest;
            t.boxes_shadow += s.boxes_shadow; t.tris_shadow += s.tris_shadow;
        }
        *stats_out = t;
    }
    return 0;
}

// ---- function-level probes (known-answer style checks of the shared math) --------------
void oracle_sincos_2pi(const float *u, int n, float *s, float *c){
    for(int i = 0; i < n; ++i) sincos_2pi_poly(u[i], s[i], c[i]);
}
void oracle_pcg_uniforms(uint64_t seed, uint32_t pixel, uint32_t sample, int n, float *out){
    Pcg p; p.seed(seed, pixel, sample);
    for(int i = 0; i < n; ++i) out[i] = p.next();
}
// out[0..2] = f, out[3] = pdf   (geometric.cuh:419-484)
void oracle_bsdf_eval_pdf(const float *mat7, const float *wo, const float *wi, const float *n, float *out){
    RMat m; m.base_color = v3(mat7[0], mat7[1], mat7[2]); m.roughness = mat7[3]; m.metallic = mat7[4]; m.eta = mat7[5]; m.type = 0;
    V3 f = bsdf_evaluate(m, v3(wo[0], wo[1], wo[2]), v3(wi[0], wi[1], wi[2]), v3(n[0], n[1], n[2]));
    out[0] = f.x; out[1] = f.y; out[2] = f.z;
    out[3] = bsdf_pdf(m, v3(wo[0], wo[1], wo[2]), v3(wi[0], wi[1], wi[2]), v3(n[0], n[1], n[2]));
}
// out[0..2] = wi, out[3..5] = f, out[6] = pdf, out[7] = is_delta, out[8] = new_eta   (geometric.cuh:486-562)
void oracle_bsdf_sample(int trig_mode, const float *mat7, const float *wo, const float *n, const float *u3, float cur_eta, float *out){
    RMat m; m.base_color = v3(mat7[0], mat7[1], mat7[2]); m.roughness = mat7[3]; m.metallic = mat7[4]; m.eta = mat7[5]; m.type = 0;
    V3 wi, f; float pdf, ne; bool d;
    bsdf_sample(trig_mode, m, v3(wo[0], wo[1], wo[2]), v3(n[0], n[1], n[2]), u3[0], u3[1], u3[2], cur_eta, wi, f, pdf, d, ne);
    out[0] = wi.x; out[1] = wi.y; out[2] = wi.z; out[3] = f.x; out[4] = f.y; out[5] = f.z; out[6] = pdf; out[7] = d ? 1.0f : 0.0f; out[8] = ne;
}
// Function-level known-answer table: the restated reference functions (ref_math.hpp: separate bsdf_evaluate / bsdf_pdf /
// bsdf_sample, each rebuilding its own frame like geometric.cuh:419-562) on n records.  Same record layout as the
// device probe k_probe_functions (24 floats in, 40 floats out).
void oracle_function_kats(const float *in, int n, float *out){
#pragma omp parallel for schedule(static)
    for(int i = 0; i < n; ++i){
        const float *r = in + (size_t) i * 24;
        float *o = out + (size_t) i * 40;
        RMat m; m.base_color = v3(r[0], r[1], r[2]); m.roughness = r[3]; m.metallic = r[4]; m.eta = r[5]; m.type = 0;
        V3 N = v3(r[6], r[7], r[8]), wo_w = v3(r[9], r[10], r[11]), wi_w = v3(r[12], r[13], r[14]);
        V3 f = bsdf_evaluate(m, wo_w, wi_w, N);
        o[0] = f.x; o[1] = f.y; o[2] = f.z; o[3] = bsdf_pdf(m, wo_w, wi_w, N);
        V3 swi, sf; float spdf, ne; bool d;
        bsdf_sample(0, m, wo_w, N, r[15], r[16], r[17], r[18], swi, sf, spdf, d, ne);
        o[4] = swi.x; o[5] = swi.y; o[6] = swi.z; o[7] = sf.x; o[8] = sf.y; o[9] = sf.z; o[10] = spdf; o[11] = d ? 1.0f : 0.0f; o[12] = ne;
        o[13] = fr_dielectric(r[19], r[20], r[21]);
        V3 sch = fr_schlick(r[19], m.base_color);
        o[14] = sch.x; o[15] = sch.y; o[16] = sch.z;
        V3 T, B; build_local_frame(N, T, B);
        V3 wo_l = world_to_local(wo_w, T, B, N), wi_l = world_to_local(wi_w, T, B, N);
        const float alpha = roughness_to_alpha(m.roughness);
        o[17] = tr_D(wi_l, alpha); o[18] = tr_lambda(wi_l, alpha); o[19] = tr_G(wo_l, wi_l, alpha);
        float sn, cs; sincos_2pi_poly(r[17], sn, cs);
        o[20] = sn; o[21] = cs;
        V3 vn = sample_vndf(0, wo_l.z > 0 ? wo_l : wo_l * -1.0f, alpha, r[16], r[17]);
        o[22] = vn.x; o[23] = vn.y; o[24] = vn.z;
        o[25] = is_valid_color(f) ? 1.0f : 0.0f;
        V3 cl = clamp_radiance(f * 20.0f, 15.0f);
        o[26] = cl.x; o[27] = cl.y; o[28] = cl.z;
        o[29] = T.x; o[30] = T.y; o[31] = T.z; o[32] = B.x; o[33] = B.y; o[34] = B.z;
        o[35] = wo_l.x; o[36] = wo_l.y; o[37] = wo_l.z;
        o[38] = 0.0f; o[39] = 0.0f;
    }
}
// brute-force closest hit for a batch of rays: out t (1e20 = miss) and prim ordinal (-1 = miss)
void oracle_closest_hits(const void *lights, int nl, const void *spheres, int ns, const void *tris, int nt,
                         const float *ro, const float *rd, int nrays, float *t_out, int *prim_out){
    Scene sc; sc.lights = (const RLight *) lights; sc.nl = nl; sc.spheres = (const RSphere *) spheres; sc.ns = ns;
    sc.tris = (const RTriangle *) tris; sc.nt = nt;
#pragma omp parallel for schedule(dynamic, 64)
    for(int i = 0; i < nrays; ++i){
        OraclePtStats st; memset(&st, 0, sizeof st);
        Hit h = closest_hit(sc, v3(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]), v3(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]), st);
        t_out[i] = h.hit ? h.t : 1e20f; prim_out[i] = h.hit ? h.prim : -1;
    }
}
void oracle_visibility(const void *spheres, int ns, const void *tris, int nt,
                       const float *p1, const float *p2, int nrays, int glass_opaque, int *vis_out){
    Scene sc; sc.lights = nullptr; sc.nl = 0; sc.spheres = (const RSphere *) spheres; sc.ns = ns;
    sc.tris = (const RTriangle *) tris; sc.nt = nt;
#pragma omp parallel for schedule(dynamic, 64)
    for(int i = 0; i < nrays; ++i){
        OraclePtStats st; memset(&st, 0, sizeof st);
        vis_out[i] = visible(sc, v3(p1[3 * i], p1[3 * i + 1], p1[3 * i + 2]), v3(p2[3 * i], p2[3 * i + 1], p2[3 * i + 2]), glass_opaque != 0, st) ? 1 : 0;
    }
}

} // extern "C"
