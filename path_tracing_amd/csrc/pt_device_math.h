// Device math of the path-tracing hot path for gfx950: float3 algebra, intersection
// tests, GGX/Fresnel BSDF, guards and the per-path random stream.
//
// What it computes follows the reference's shared geometry/BSDF header
// (reference include/geometric.cuh:90-99, 119-235, 240-291, 419-562).  Arithmetic
// contract: IEEE binary32, every expression evaluated in the order written, no FMA
// contraction (this translation unit is built with -ffp-contract=off; hipcc's '/'
// and sqrtf are correctly rounded by default), so a CPU evaluating the same
// expressions agrees bit for bit.  Box tests in the traversal are exempt: they only
// have to be conservative and use explicit fmaf.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace hpt {

#define HPT_DEV __device__ __forceinline__

// development (-DHPT_SHADE_PROFILE=3|4|5, counting renders only): wave trips and active lanes at probe points inside the BSDF
// functions, counted through a pointer k_shade publishes (scripts/shade_sections.py); nothing in a product build
#ifdef HPT_SHADE_PROFILE
static __device__ unsigned long long *g_hpt_probe;
#define HPT_PROBE(set, idx) do { unsigned long long *p_ = g_hpt_probe; if(p_ && HPT_SHADE_PROFILE == (set)){ unsigned long long m_ = __ballot(true); \
    if((int) (threadIdx.x & 63u) == __ffsll((long long) m_) - 1){ atomicAdd(p_ + 2 * (idx), 1ull); atomicAdd(p_ + 2 * (idx) + 1, (unsigned long long) __popcll(m_)); } } } while(0)
#else
#define HPT_PROBE(set, idx) do {} while(0)
#endif

constexpr float kEps = 1e-4f;
constexpr float kPi = 3.14159265358979323846f;

struct f3 { float x, y, z; };

HPT_DEV f3 mk3(float x, float y, float z){ f3 r; r.x = x; r.y = y; r.z = z; return r; }
HPT_DEV f3 operator+(f3 a, f3 b){ return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
HPT_DEV f3 operator-(f3 a, f3 b){ return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
HPT_DEV f3 operator*(f3 a, float s){ return mk3(a.x * s, a.y * s, a.z * s); }
HPT_DEV f3 operator*(f3 a, f3 b){ return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
HPT_DEV f3 operator/(f3 a, float s){ return mk3(a.x / s, a.y / s, a.z / s); }
HPT_DEV float dot3(f3 a, f3 b){ return a.x * b.x + a.y * b.y + a.z * b.z; }
HPT_DEV f3 cross3(f3 a, f3 b){
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
HPT_DEV float length3(f3 a){ return sqrtf(dot3(a, a)); }
HPT_DEV f3 normalize3(f3 a){ return a / length3(a); }
HPT_DEV f3 reflect3(f3 I, f3 N){ return I - N * 2.0f * dot3(N, I); }
HPT_DEV f3 neg3(f3 a){ return a * -1.0f; }

HPT_DEV bool is_inf(float x){ return __builtin_isinf(x); }
HPT_DEV bool is_nan(float x){ return __builtin_isnan(x); }

HPT_DEV bool is_valid_color(f3 c){
    return !(is_nan(c.x) || is_nan(c.y) || is_nan(c.z) || is_inf(c.x) || is_inf(c.y) || is_inf(c.z) ||
             c.x < 0.0f || c.y < 0.0f || c.z < 0.0f);
}
HPT_DEV f3 clamp_radiance(f3 c, float max_val){
    float m = fmaxf(c.x, fmaxf(c.y, c.z));
    if(m > max_val) return c * (max_val / m);
    return c;
}

// sin/cos of 2*pi*u, u in [0,1): quadrant reduction + minimax polynomials on [-pi/4, pi/4].
// One IEEE op per step in the written order (the CPU checker restates this op for op).
HPT_DEV void sincos_2pi(float u, float &s, float &c){
    float q = floorf(u * 4.0f + 0.5f);
    float r = u - q * 0.25f;
    float x = r * 6.28318530717958647692f;
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

// ---- per-path random stream: PCG32 (XSH-RR 64/32), 24-bit uniforms in [0,1) -------------
HPT_DEV uint64_t mix64(uint64_t z){
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
HPT_DEV uint64_t rng_seed(uint64_t seed, uint32_t pixel, uint32_t sample){
    return mix64(mix64(seed + (uint64_t) pixel) + (uint64_t) sample);
}
HPT_DEV float rng_next(uint64_t &state){
    uint64_t old = state;
    state = old * 6364136223846793005ull + 1442695040888963407ull;
    uint32_t xs = (uint32_t) (((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t) (old >> 59);
    uint32_t v = (xs >> rot) | (xs << ((32u - rot) & 31u));
    return (float) (v >> 8) * (1.0f / 16777216.0f);
}

// ---- material record as the BSDF sees it ------------------------------------------------
struct Mat { f3 base; float roughness, metallic, eta; };

// local shading frame
HPT_DEV void build_local_frame(f3 N, f3 &T, f3 &B){
    if(fabsf(N.z) < 0.999f) T = normalize3(cross3(mk3(0, 0, 1), N));
    else T = normalize3(cross3(mk3(0, 1, 0), N));
    B = cross3(N, T);
}
HPT_DEV f3 to_local(f3 v, f3 T, f3 B, f3 N){ return mk3(dot3(v, T), dot3(v, B), dot3(v, N)); }
HPT_DEV f3 to_world(f3 v, f3 T, f3 B, f3 N){
    return mk3(T.x * v.x + B.x * v.y + N.x * v.z,
               T.y * v.x + B.y * v.y + N.y * v.z,
               T.z * v.x + B.z * v.y + N.z * v.z);
}
HPT_DEV float cos2_theta(f3 w){ return w.z * w.z; }
HPT_DEV float sin2_theta(f3 w){ return fmaxf(0.0f, 1.0f - cos2_theta(w)); }
HPT_DEV float tan_theta(f3 w){ return sqrtf(sin2_theta(w)) / (w.z + 1e-7f); }
HPT_DEV float tan2_theta(f3 w){ return sin2_theta(w) / (cos2_theta(w) + 1e-7f); }

HPT_DEV float fr_dielectric(float cosI, float etaI, float etaT){
    cosI = fmaxf(-1.0f, fminf(1.0f, cosI));
    if(!(cosI > 0.0f)){
        float tmp = etaI; etaI = etaT; etaT = tmp;
        cosI = fabsf(cosI);
    }
    float sinI = sqrtf(fmaxf(0.0f, 1.0f - cosI * cosI));
    float sinT = etaI / etaT * sinI;
    if(sinT >= 1.0f) return 1.0f;
    float cosT = sqrtf(fmaxf(0.0f, 1.0f - sinT * sinT));
    float Rparl = ((etaT * cosI) - (etaI * cosT)) / ((etaT * cosI) + (etaI * cosT));
    float Rperp = ((etaI * cosI) - (etaT * cosT)) / ((etaI * cosI) + (etaT * cosT));
    return (Rparl * Rparl + Rperp * Rperp) / 2.0f;
}
HPT_DEV f3 fr_schlick(float cosI, f3 R0){
    float c = fmaxf(0.0f, 1.0f - cosI);
    float c5 = c * c * c * c * c;
    return R0 + (mk3(1.0f, 1.0f, 1.0f) - R0) * c5;
}
HPT_DEV float roughness_to_alpha(float roughness){ float x = fmaxf(roughness, 1e-3f); return x * x; }
HPT_DEV float ggx_D(f3 wh, float alpha){
    float t2 = tan2_theta(wh);
    if(is_inf(t2)) return 0.0f;
    float cos4 = cos2_theta(wh) * cos2_theta(wh);
    float e = (cos4 * (alpha * alpha + t2 * t2));
    if(e < 1e-12f) return 0.0f;
    return (alpha * alpha) / (kPi * e);
}
HPT_DEV float ggx_lambda(f3 w, float alpha){
    float at = fabsf(tan_theta(w));
    if(is_inf(at)) return 0.0f;
    float a2t2 = (alpha * at) * (alpha * at);
    return (-1.0f + sqrtf(1.0f + a2t2)) / 2.0f;
}
HPT_DEV float ggx_G(f3 wo, f3 wi, float alpha){
    return 1.0f / (1.0f + ggx_lambda(wo, alpha) + ggx_lambda(wi, alpha));
}
// r = sqrt(u1) and (sn, cs) = sincos(2 pi u2) come from the caller: the cosine-lobe sample uses the same two
HPT_DEV f3 sample_visible_normal(f3 wo, float alpha, float r, float sn, float cs){
    f3 V = normalize3(mk3(alpha * wo.x, alpha * wo.y, wo.z));
    f3 T1 = (V.z < 0.9999f) ? normalize3(cross3(mk3(0, 0, 1), V)) : mk3(1, 0, 0);
    f3 T2 = cross3(V, T1);
    float t1 = r * cs;
    float t2 = r * sn;
    float s = 0.5f * (1.0f + V.z);
    t2 = (1.0f - s) * sqrtf(fmaxf(0.0f, 1.0f - t1 * t1)) + s * t2;
    f3 Nh = T1 * t1 + T2 * t2 + V * sqrtf(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2));
    return normalize3(mk3(alpha * Nh.x, alpha * Nh.y, fmaxf(0.0f, Nh.z)));
}

// Shading context shared by every BSDF query at one hit point: the reference's evaluate / pdf /
// sample each rebuild the same local frame from N and re-project the same wo
// (geometric.cuh:421-424, 460-462, 495-497); building them once gives identical values.
struct ShadeCtx { f3 T, B, N; f3 wo; };

HPT_DEV ShadeCtx make_shade_ctx(f3 N, f3 wo_w){
    ShadeCtx c; c.N = N;
    build_local_frame(N, c.T, c.B);
    c.wo = to_local(wo_w, c.T, c.B, N);
    return c;
}

// BSDF value and pdf in one pass over the shared terms (the two reference functions,
// geometric.cuh:419-456 and 458-484, build the same half vector, D and Lambda(wo)).
// Terms of a BSDF query that depend on the hit only (material, outgoing direction), for callers that make
// several queries at one hit: the diffuse lobe (a per-material constant) and Lambda(wo).
struct ShadePre { f3 diffuse; float lam_o; };

// (wo, wi in the local frame of the hit; bsdf_eval_pdf below projects a world direction first)
template <bool WANT_F = true, bool WANT_PDF = true>
HPT_DEV void bsdf_eval_pdf_local(const Mat &m, f3 wo, f3 wi, f3 &f_out, float &pdf_out, const ShadePre *pre = nullptr){
    f_out = mk3(0, 0, 0); pdf_out = 0.0f;
    bool eval_zero = !WANT_F || (wo.z == 0.0f || wi.z == 0.0f);
    bool pdf_zero = !WANT_PDF || (wo.z * wi.z <= 0.0f);
    if(eval_zero && pdf_zero) return;
    if(m.eta > 0.0f && m.roughness < 0.001f) return;
    HPT_PROBE(4, 0);                                               // past the early returns: a value and/or a pdf is wanted
    float alpha = roughness_to_alpha(m.roughness);
    f3 whv = wo + wi;
    if(length3(whv) < 1e-6f) return;
    f3 wh = normalize3(whv);
    if(wh.z < 0.0f) wh = wh * -1.0f;
    float D = ggx_D(wh, alpha);
    float lam_o = pre ? pre->lam_o : ggx_lambda(wo, alpha);
    float awo = fabsf(wo.z), awi = fabsf(wi.z);
    if(WANT_F && !eval_zero){
        HPT_PROBE(4, 1);                                           // the value: G, Fresnel, specular term
        f3 diffuse = pre ? pre->diffuse : m.base / kPi * (1.0f - m.metallic);
        if(wo.z * wi.z < 0.0f) diffuse = mk3(0, 0, 0);
        float G = 1.0f / (1.0f + lam_o + ggx_lambda(wi, alpha));
        f3 F;
        if(m.metallic > 0.0f) F = fr_schlick(awo, m.base);
        else { HPT_PROBE(4, 2); float fr = fr_dielectric(dot3(wo, wh), 1.0f, m.eta); F = mk3(fr, fr, fr); }
        f3 specular = (F * D * G) / fmaxf(4.0f * awo * awi, 1e-4f);
        f_out = (wo.z * wi.z > 0.0f) ? diffuse + specular : diffuse;
    }
    if(WANT_PDF && !pdf_zero){
        HPT_PROBE(4, 3);                                           // the pdf
        float pdf_diffuse = awi / kPi;
        float G1 = 1.0f / (1.0f + lam_o);
        float pdf_wh = D * G1 * fmaxf(0.0f, dot3(wo, wh)) / awo;
        float pdf_specular = pdf_wh / (4.0f * dot3(wo, wh) + 1e-7f);
        float spec_weight = m.metallic > 0.0f ? 1.0f : 0.5f;
        float diff_weight = 1.0f - spec_weight;
        pdf_out = diff_weight * pdf_diffuse + spec_weight * pdf_specular;
    }
}

template <bool WANT_F = true, bool WANT_PDF = true>
HPT_DEV void bsdf_eval_pdf(const Mat &m, const ShadeCtx &c, f3 wi_w, f3 &f_out, float &pdf_out, const ShadePre *pre = nullptr){
    bsdf_eval_pdf_local<WANT_F, WANT_PDF>(m, c.wo, to_local(wi_w, c.T, c.B, c.N), f_out, pdf_out, pre);
}

// BSDF sampling (geometric.cuh:486-562).  pdf <= 0 means "terminate the path" for both the
// non-delta rejection and the reference's uninitialised total-internal-reflection return.
HPT_DEV void bsdf_sample(const Mat &m, const ShadeCtx &c, float u_rr, float u1, float u2, float cur_eta,
                         f3 &wi_w, f3 &f, float &pdf, bool &is_delta, float &new_eta, const ShadePre *pre = nullptr){
    is_delta = false;
    new_eta = cur_eta;
    wi_w = mk3(0, 0, 0); f = mk3(0, 0, 0); pdf = 0.0f;
    f3 wo = c.wo;
    f3 wi;
    if(m.eta > 0.0f && m.roughness < 0.001f && m.metallic < 0.01f){
        HPT_PROBE(3, 0);                                           // smooth dielectric
        is_delta = true;
        float F = fr_dielectric(wo.z, cur_eta, m.eta);
        if(u_rr < F){
            wi = mk3(-wo.x, -wo.y, wo.z);
            pdf = F;
            f = mk3(F, F, F) / fabsf(wi.z);
        } else {
            float eta = wo.z > 0.0f ? (cur_eta / m.eta) : (m.eta / cur_eta);
            float sin2I = fmaxf(0.0f, 1.0f - cos2_theta(wo));
            float sin2T = eta * eta * sin2I;
            if(sin2T >= 1.0f){ pdf = 0.0f; return; }
            float cosT = sqrtf(1.0f - sin2T);
            if(wo.z > 0.0f) cosT = -cosT;
            wi = mk3(-eta * wo.x, -eta * wo.y, cosT);
            new_eta = (wo.z > 0.0f) ? m.eta : 1.0f;
            pdf = 1.0f - F;
            f = m.base * (1.0f - F) / fabsf(wi.z);
        }
        wi_w = to_world(wi, c.T, c.B, c.N);
        return;
    }
    if(m.metallic > 0.99f && m.roughness < 0.001f){
        HPT_PROBE(3, 1);                                           // mirror
        is_delta = true;
        wi = mk3(-wo.x, -wo.y, wo.z);
        pdf = 1.0f;
        f = fr_schlick(fabsf(wo.z), m.base) / fabsf(wi.z);
        wi_w = to_world(wi, c.T, c.B, c.N);
        return;
    }
    float alpha = roughness_to_alpha(m.roughness);
    float spec_weight = m.metallic > 0.0f ? 1.0f : 0.5f;
    // both lobes start from the same disk point (geometric.cuh:205-207 and :393-396); computed once, ahead of
    // the branch the lanes of a wave split on
    float r = sqrtf(u1);
    float sn, cs; sincos_2pi(u2, sn, cs);
    if(u_rr < spec_weight){
        HPT_PROBE(3, 2);                                           // specular lobe: visible-normal sample
        f3 wh = sample_visible_normal(wo.z > 0 ? wo : wo * -1.0f, alpha, r, sn, cs);
        if(wo.z < 0.0f) wh = wh * -1.0f;
        wi = reflect3(wo * -1.0f, wh);
        if(wo.z * wi.z <= 0.0f){ pdf = 0.0f; return; }
    } else {
        wi = mk3(r * cs, r * sn, sqrtf(fmaxf(0.0f, 1.0f - u1)));
        if(wo.z < 0.0f) wi.z *= -1.0f;
    }
    HPT_PROBE(3, 3);                                               // both lobes: value and pdf of the sampled direction
    wi_w = to_world(wi, c.T, c.B, c.N);
    bsdf_eval_pdf(m, c, wi_w, f, pdf, pre);
}

// ---- primitive tests --------------------------------------------------------------------
// unit-direction sphere test, near root then far root (geometric.cuh:240-259)
HPT_DEV bool hit_sphere(f3 ro, f3 rd, f3 center, float radius, float max_dist, float &t){
    f3 oc = ro - center;
    float b = dot3(oc, rd);
    float c = dot3(oc, oc) - radius * radius;
    float h = b * b - c;
    if(h < 0.0f) return false;
    h = sqrtf(h);
    float th = -b - h;
    if(th > kEps && th < max_dist){ t = th; return true; }
    th = -b + h;
    if(th > kEps && th < max_dist){ t = th; return true; }
    return false;
}
// Moeller-Trumbore with precomputed edges, two-sided (geometric.cuh:261-291)
HPT_DEV bool hit_triangle(f3 ro, f3 rd, f3 v0, f3 e1, f3 e2, float max_dist, float &t){
    f3 h = cross3(rd, e2);
    float a = dot3(e1, h);
    if(a > -1e-6f && a < 1e-6f) return false;
    float f = 1.0f / a;
    f3 s = ro - v0;
    float u = f * dot3(s, h);
    if(u < 0.0f || u > 1.0f) return false;
    f3 q = cross3(s, e1);
    float v = f * dot3(rd, q);
    if(v < 0.0f || u + v > 1.0f) return false;
    float th = f * dot3(e2, q);
    if(th > kEps && th < max_dist){ t = th; return true; }
    return false;
}
} // namespace hpt
