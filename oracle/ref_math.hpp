// TEST INFRASTRUCTURE ONLY -- CPU oracle for the MI355X path-tracing hot path.
//
// Independent CPU restatement of the reference's shared geometry + BSDF math
// (reference: include/geometric.cuh).  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may build, link or call anything under oracle/.
// The product (path_tracing_amd/) never includes this header.
//
// PARITY STATUS: *parity unpinned*.  The reference ships no tests, golden
// vectors or fixtures for this path, and its sources cannot be compiled in this
// image (they need glm and <curand_kernel.h>, which are absent; no stand-in
// headers are written).  Advisory cross-checks only: see tests/golden/README.md.
//
// Arithmetic contract (shared, by construction, with the HIP kernels):
//  * IEEE binary32, every expression evaluated exactly in the order written,
//    no FMA contraction (-ffp-contract=off on both sides), correctly rounded
//    '/' and sqrtf on both sides.
//  * trig_mode 0 ("poly"): sin/cos of 2*pi*u come from sincos_2pi_poly() below,
//    which the HIP kernels restate op for op, so CPU and GPU agree bit for bit.
//    trig_mode 1 ("libm"): cosf/sinf from glibc, used only to replay the
//    reference's host arithmetic (mt19937 replay checks).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

constexpr float kEps = 1e-4f;                       // geometric.cuh:6
constexpr float kPi = 3.14159265358979323846f;      // geometric.cuh:7

struct V3 { float x, y, z; };

static inline V3 v3(float x, float y, float z){ V3 r; r.x = x; r.y = y; r.z = z; return r; }
// float3 algebra, geometric.cuh:90-99 (same operand order, left-to-right sums)
static inline V3 operator+(V3 a, V3 b){ return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b){ return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator*(V3 a, float s){ return v3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator*(V3 a, V3 b){ return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 operator/(V3 a, float s){ return v3(a.x / s, a.y / s, a.z / s); }
static inline float dot(V3 a, V3 b){ return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b){
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float length(V3 a){ return sqrtf(dot(a, a)); }
static inline V3 normalize(V3 a){ return a / length(a); }      // geometric.cuh:98 (divide, not rsqrt)
static inline V3 reflect(V3 I, V3 N){ return I - N * 2.0f * dot(N, I); }   // geometric.cuh:99

// ---- reference POD layouts (measured) ---------------------
struct RMatOld { V3 Kd, Kg, Ks; float glossy, exp, refract, reflect; };   // 52 B, geometric.cuh:15-18
struct RMat { V3 base_color; float roughness, metallic, eta; int type; };   // 28 B, geometric.cuh:21-27
struct RSphere { V3 center; float r; RMatOld mtl_old; RMat mtl; int id; };  // 100 B, geometric.cuh:29-35
struct RTriangle { V3 v0, v1, v2; RMatOld mtl_old; RMat mtl; int id; };    // 120 B, geometric.cuh:37-42
struct RLight { V3 pos, dir, illum; RSphere light_ball; float cutoff; int is_parallel; }; // 144 B, :73-78
struct RCamera { V3 eye, U, V, W, UL, dx, dy; };                             // 84 B, geometric.cuh:67-69
static_assert(sizeof(RMatOld) == 52 && sizeof(RMat) == 28, "layout");
static_assert(sizeof(RSphere) == 100 && sizeof(RTriangle) == 120, "layout");
static_assert(sizeof(RLight) == 144 && sizeof(RCamera) == 84, "layout");

struct Hit {                   // subset of CudaHit, geometric.cuh:44-51
    bool hit; float t; V3 pos, normal; RMat mtl; bool is_light;
    int prim;                  // global ordinal (spheres, then light balls, then triangles); oracle extra
};

// ---- deterministic sin/cos of 2*pi*u for u in [0,1) ---------------------------
// Quadrant reduction + cephes-style minimax polynomials on [-pi/4, pi/4].
// Every operation is a single IEEE binary32 op in the written order.
static inline void sincos_2pi_poly(float u, float &s, float &c){
    float q = floorf(u * 4.0f + 0.5f);          // nearest quarter turn, 0..4
    float r = u - q * 0.25f;                    // exact: |r| <= 1/8
    float x = r * 6.28318530717958647692f;      // |x| <= pi/4
    float x2 = x * x;
    float sp = -1.9515295891e-4f;
    sp = sp * x2 + 8.3321608736e-3f;
    sp = sp * x2 + -1.6666654611e-1f;
    float sn = x + x * x2 * sp;
    float cp = 2.443315711809948e-5f;
    cp = cp * x2 + -1.388731625493765e-3f;
    cp = cp * x2 + 4.166664568298827e-2f;
    float cs = 1.0f - 0.5f * x2 + x2 * x2 * cp;
    int qi = ((int) q) & 3;
    if(qi == 0){ s = sn; c = cs; }
    else if(qi == 1){ s = cs; c = -sn; }
    else if(qi == 2){ s = -sn; c = -cs; }
    else { s = -cs; c = sn; }
}

// sin/cos of phi = 2*pi*u2 as the reference writes it (geometric.cuh:210-212, 554-555)
static inline void sincos_phi(int trig_mode, float u2, float &s, float &c){
    if(trig_mode == 1){
        float phi = 2.0f * kPi * u2;
        c = cosf(phi); s = sinf(phi);
    } else {
        sincos_2pi_poly(u2, s, c);
    }
}

// ---- local frame, geometric.cuh:119-133 ---------------------------------------
static inline void build_local_frame(V3 N, V3 &T, V3 &B){
    if(fabsf(N.z) < 0.999f) T = normalize(cross(v3(0, 0, 1), N));
    else T = normalize(cross(v3(0, 1, 0), N));
    B = cross(N, T);
}
static inline V3 world_to_local(V3 v, V3 T, V3 B, V3 N){ return v3(dot(v, T), dot(v, B), dot(v, N)); }
static inline V3 local_to_world(V3 v, V3 T, V3 B, V3 N){
    return v3(T.x * v.x + B.x * v.y + N.x * v.z,
              T.y * v.x + B.y * v.y + N.y * v.z,
              T.z * v.x + B.z * v.y + N.z * v.z);
}
// local trig, geometric.cuh:136-142
static inline float cos2_theta(V3 w){ return w.z * w.z; }
static inline float abs_cos_theta(V3 w){ return fabsf(w.z); }
static inline float sin2_theta(V3 w){ return fmaxf(0.0f, 1.0f - cos2_theta(w)); }
static inline float sin_theta(V3 w){ return sqrtf(sin2_theta(w)); }
static inline float tan_theta(V3 w){ return sin_theta(w) / (w.z + 1e-7f); }
static inline float tan2_theta(V3 w){ return sin2_theta(w) / (cos2_theta(w) + 1e-7f); }

// exact dielectric Fresnel, geometric.cuh:145-160
static inline float fr_dielectric(float cosThetaI, float etaI, float etaT){
    cosThetaI = fmaxf(-1.0f, fminf(1.0f, cosThetaI));
    if(!(cosThetaI > 0.0f)){
        float tmp = etaI; etaI = etaT; etaT = tmp;
        cosThetaI = fabsf(cosThetaI);
    }
    float sinThetaI = sqrtf(fmaxf(0.0f, 1.0f - cosThetaI * cosThetaI));
    float sinThetaT = etaI / etaT * sinThetaI;
    if(sinThetaT >= 1.0f) return 1.0f;
    float cosThetaT = sqrtf(fmaxf(0.0f, 1.0f - sinThetaT * sinThetaT));
    float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return (Rparl * Rparl + Rperp * Rperp) / 2.0f;
}
// Schlick, geometric.cuh:163-167
static inline V3 fr_schlick(float cosThetaI, V3 R0){
    float c = fmaxf(0.0f, 1.0f - cosThetaI);
    float c5 = c * c * c * c * c;
    return R0 + (v3(1.0f, 1.0f, 1.0f) - R0) * c5;
}
// GGX, geometric.cuh:173-197
static inline float roughness_to_alpha(float roughness){ float x = fmaxf(roughness, 1e-3f); return x * x; }
static inline float tr_D(V3 wh, float alpha){
    float t2 = tan2_theta(wh);
    if(std::isinf(t2)) return 0.0f;
    float cos4 = cos2_theta(wh) * cos2_theta(wh);
    float e = (cos4 * (alpha * alpha + t2 * t2));
    if(e < 1e-12f) return 0.0f;
    return (alpha * alpha) / (kPi * e);
}
static inline float tr_lambda(V3 w, float alpha){
    float at = fabsf(tan_theta(w));
    if(std::isinf(at)) return 0.0f;
    float a2t2 = (alpha * at) * (alpha * at);
    return (-1.0f + sqrtf(1.0f + a2t2)) / 2.0f;
}
static inline float tr_G(V3 wo, V3 wi, float alpha){
    return 1.0f / (1.0f + tr_lambda(wo, alpha) + tr_lambda(wi, alpha));
}
// visible-normal sampling, geometric.cuh:200-221
static inline V3 sample_vndf(int trig_mode, V3 wo, float alpha, float u1, float u2){
    V3 V = normalize(v3(alpha * wo.x, alpha * wo.y, wo.z));
    V3 T1 = (V.z < 0.9999f) ? normalize(cross(v3(0, 0, 1), V)) : v3(1, 0, 0);
    V3 T2 = cross(V, T1);
    float r = sqrtf(u1);
    float sn, cs; sincos_phi(trig_mode, u2, sn, cs);
    float t1 = r * cs;
    float t2 = r * sn;
    float s = 0.5f * (1.0f + V.z);
    t2 = (1.0f - s) * sqrtf(fmaxf(0.0f, 1.0f - t1 * t1)) + s * t2;
    V3 Nh = T1 * t1 + T2 * t2 + V * sqrtf(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2));
    return normalize(v3(alpha * Nh.x, alpha * Nh.y, fmaxf(0.0f, Nh.z)));
}

// guards, geometric.cuh:223-235
static inline bool is_valid_color(V3 c){
    return !(std::isnan(c.x) || std::isnan(c.y) || std::isnan(c.z) ||
             std::isinf(c.x) || std::isinf(c.y) || std::isinf(c.z) ||
             c.x < 0.0f || c.y < 0.0f || c.z < 0.0f);
}
static inline V3 clamp_radiance(V3 c, float max_val){
    float m = fmaxf(c.x, fmaxf(c.y, c.z));
    if(m > max_val) return c * (max_val / m);
    return c;
}

// sphere, geometric.cuh:240-259 (unit direction assumed; near root then far root)
static inline bool intersect_sphere(V3 ro, V3 rd, V3 center, float radius, float &t, float max_dist){
    V3 oc = ro - center;
    float b = dot(oc, rd);
    float c = dot(oc, oc) - radius * radius;
    float h = b * b - c;
    if(h < 0.0f) return false;
    h = sqrtf(h);
    float t_hit = -b - h;
    if(t_hit > kEps && t_hit < max_dist){ t = t_hit; return true; }
    t_hit = -b + h;
    if(t_hit > kEps && t_hit < max_dist){ t = t_hit; return true; }
    return false;
}
// Moeller-Trumbore, geometric.cuh:261-291
static inline bool intersect_triangle(V3 ro, V3 rd, V3 v0, V3 v1, V3 v2, float &t, float max_dist){
    V3 e1 = v1 - v0;
    V3 e2 = v2 - v0;
    V3 h = cross(rd, e2);
    float a = dot(e1, h);
    if(a > -1e-6f && a < 1e-6f) return false;
    float f = 1.0f / a;
    V3 s = ro - v0;
    float u = f * dot(s, h);
    if(u < 0.0f || u > 1.0f) return false;
    V3 q = cross(s, e1);
    float v = f * dot(rd, q);
    if(v < 0.0f || u + v > 1.0f) return false;
    float t_hit = f * dot(e2, q);
    if(t_hit > kEps && t_hit < max_dist){ t = t_hit; return true; }
    return false;
}

// BSDF value, geometric.cuh:419-456
static inline V3 bsdf_evaluate(const RMat &m, V3 wo_w, V3 wi_w, V3 N){
    V3 T, B; build_local_frame(N, T, B);
    V3 wo = world_to_local(wo_w, T, B, N);
    V3 wi = world_to_local(wi_w, T, B, N);
    if(wo.z == 0.0f || wi.z == 0.0f) return v3(0, 0, 0);
    if(m.eta > 0.0f && m.roughness < 0.001f) return v3(0, 0, 0);
    float alpha = roughness_to_alpha(m.roughness);
    V3 whv = wo + wi;
    if(length(whv) < 1e-6f) return v3(0, 0, 0);
    V3 wh = normalize(whv);
    if(wh.z < 0.0f) wh = wh * -1.0f;
    V3 diffuse = m.base_color / kPi * (1.0f - m.metallic);
    if(wo.z * wi.z < 0.0f) diffuse = v3(0, 0, 0);
    float D = tr_D(wh, alpha);
    float G = tr_G(wo, wi, alpha);
    V3 F;
    if(m.metallic > 0.0f) F = fr_schlick(abs_cos_theta(wo), m.base_color);
    else { float fr = fr_dielectric(dot(wo, wh), 1.0f, m.eta); F = v3(fr, fr, fr); }
    V3 specular = (F * D * G) / fmaxf(4.0f * abs_cos_theta(wo) * abs_cos_theta(wi), 1e-4f);
    if(wo.z * wi.z > 0.0f) return diffuse + specular;
    return diffuse;
}
// BSDF pdf, geometric.cuh:458-484
static inline float bsdf_pdf(const RMat &m, V3 wo_w, V3 wi_w, V3 N){
    V3 T, B; build_local_frame(N, T, B);
    V3 wo = world_to_local(wo_w, T, B, N);
    V3 wi = world_to_local(wi_w, T, B, N);
    if(wo.z * wi.z <= 0.0f) return 0.0f;
    if(m.eta > 0.0f && m.roughness < 0.001f) return 0.0f;
    float alpha = roughness_to_alpha(m.roughness);
    V3 whv = wo + wi;
    if(length(whv) < 1e-6f) return 0.0f;
    V3 wh = normalize(whv);
    if(wh.z < 0.0f) wh = wh * -1.0f;
    float pdf_diffuse = abs_cos_theta(wi) / kPi;
    float G1 = 1.0f / (1.0f + tr_lambda(wo, alpha));
    float pdf_wh = tr_D(wh, alpha) * G1 * fmaxf(0.0f, dot(wo, wh)) / abs_cos_theta(wo);
    float pdf_specular = pdf_wh / (4.0f * dot(wo, wh) + 1e-7f);
    float spec_weight = m.metallic > 0.0f ? 1.0f : 0.5f;
    float diff_weight = 1.0f - spec_weight;
    return diff_weight * pdf_diffuse + spec_weight * pdf_specular;
}
// BSDF sampling, geometric.cuh:486-562.  pdf<=0 with is_delta set is the
// reference's uninitialised total-internal-reflection return:
// callers here terminate the path on it.
static inline void bsdf_sample(int trig_mode, const RMat &m, V3 wo_w, V3 N,
                               float u_rr, float u1, float u2, float current_eta,
                               V3 &wi_w, V3 &f, float &pdf, bool &is_delta, float &new_eta){
    is_delta = false;
    new_eta = current_eta;
    wi_w = v3(0, 0, 0); f = v3(0, 0, 0); pdf = 0.0f;
    V3 T, B; build_local_frame(N, T, B);
    V3 wo = world_to_local(wo_w, T, B, N);
    V3 wi;
    if(m.eta > 0.0f && m.roughness < 0.001f && m.metallic < 0.01f){
        is_delta = true;
        float F = fr_dielectric(wo.z, current_eta, m.eta);
        if(u_rr < F){
            wi = v3(-wo.x, -wo.y, wo.z);
            pdf = F;
            f = v3(F, F, F) / abs_cos_theta(wi);
        } else {
            float eta = wo.z > 0.0f ? (current_eta / m.eta) : (m.eta / current_eta);
            float sin2I = fmaxf(0.0f, 1.0f - cos2_theta(wo));
            float sin2T = eta * eta * sin2I;
            if(sin2T >= 1.0f){ pdf = 0.0f; return; }
            float cosT = sqrtf(1.0f - sin2T);
            if(wo.z > 0.0f) cosT = -cosT;
            wi = v3(-eta * wo.x, -eta * wo.y, cosT);
            new_eta = (wo.z > 0.0f) ? m.eta : 1.0f;
            pdf = 1.0f - F;
            f = m.base_color * (1.0f - F) / abs_cos_theta(wi);
        }
        wi_w = local_to_world(wi, T, B, N);
        return;
    }
    if(m.metallic > 0.99f && m.roughness < 0.001f){
        is_delta = true;
        wi = v3(-wo.x, -wo.y, wo.z);
        pdf = 1.0f;
        f = fr_schlick(abs_cos_theta(wo), m.base_color) / abs_cos_theta(wi);
        wi_w = local_to_world(wi, T, B, N);
        return;
    }
    float alpha = roughness_to_alpha(m.roughness);
    float spec_weight = m.metallic > 0.0f ? 1.0f : 0.5f;
    if(u_rr < spec_weight){
        V3 wh = sample_vndf(trig_mode, wo.z > 0 ? wo : wo * -1.0f, alpha, u1, u2);
        if(wo.z < 0.0f) wh = wh * -1.0f;
        wi = reflect(wo * -1.0f, wh);
        if(wo.z * wi.z <= 0.0f){ pdf = 0.0f; return; }
    } else {
        float r = sqrtf(u1);
        float sn, cs; sincos_phi(trig_mode, u2, sn, cs);
        wi = v3(r * cs, r * sn, sqrtf(fmaxf(0.0f, 1.0f - u1)));
        if(wo.z < 0.0f) wi.z *= -1.0f;
    }
    wi_w = local_to_world(wi, T, B, N);
    pdf = bsdf_pdf(m, wo_w, wi_w, N);
    f = bsdf_evaluate(m, wo_w, wi_w, N);
}

// ---- random numbers ------------------------------------------------------------
// Counter mode (shared with the HIP kernels): one PCG32 (XSH-RR 64/32) stream per
// (seed, global pixel index, global sample index); uniforms are 24-bit, in [0,1).
static inline uint64_t mix64(uint64_t z){            // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct Pcg {
    uint64_t state;
    void seed(uint64_t seed, uint32_t pixel, uint32_t sample){
        state = mix64(mix64(seed + (uint64_t) pixel) + (uint64_t) sample);
    }
    uint32_t next_u32(){
        uint64_t old = state;
        state = old * 6364136223846793005ull + 1442695040888963407ull;
        uint32_t xs = (uint32_t) (((old >> 18) ^ old) >> 27);
        uint32_t rot = (uint32_t) (old >> 59);
        return (xs >> rot) | (xs << ((32u - rot) & 31u));
    }
    float next(){ return (float) (next_u32() >> 8) * (1.0f / 16777216.0f); }
};

} // namespace orc
