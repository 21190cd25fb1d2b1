// HIP kernels of the bidirectional estimator of the reference's CPU renderer
// (reference src/cpu_bdpt.cpp:173-488), for gfx950.  What is computed follows cpu_bdpt.cpp where
// it and the CUDA BDPT kernel disagree: light-vertex pool nl*spl, illum / spl,
// zero-throughput vertices skipped, CPU scene model (reference src/object.cpp:16-121: sphere test
// that rejects origins inside, inclusive ranges, last object wins ties, one box per group).
//
//   light_trace   one lane per light subpath (a few dozen paths): emission + bounces -> vertex pool
//   generate      primary rays (identity queue)
//   extend        closest hit, CPU scene model, one BVH per group
//   vertex        eye vertex: depth-0 light hit, pdf bookkeeping, BSDF sample, compaction
//   connect       one WAVE per (eye vertex, 64 light vertices): culls, 2 x BSDF value, shadow ray,
//                 MIS weight -> contribution table [path][light vertex]
//   reduce        per eye vertex, sums the table in light-vertex order (the reference's loop order,
//                 so the float sums match the CPU bit for bit) into the sample's radiance
//
// Arithmetic contract as in pt_device_math.h (-ffp-contract=off, expressions in reference order).
#include "bdpt_kernels.h"
#include "pt_device_math.h"

namespace hpt {

namespace {

HPT_DEV uint32_t f2u(float f){ return __float_as_uint(f); }
HPT_DEV float u2f(uint32_t u){ return __uint_as_float(u); }
HPT_DEV f3 xyz(float4 v){ return mk3(v.x, v.y, v.z); }
HPT_DEV f3 ld3(const float *p){ return mk3(p[0], p[1], p[2]); }

constexpr uint32_t kBdMiss = 0xFFFFFFFFu;
constexpr uint32_t kBdSphere = 0x80000000u;      // | sphere index
constexpr uint32_t kBdLight = 0xC0000000u;       // | light index

// glm::normalize: v * inversesqrt(dot(v, v)) -- the CPU scene model's normalisation
HPT_DEV f3 gnormalize(f3 a){ float inv = 1.0f / sqrtf(dot3(a, a)); return a * inv; }

HPT_DEV uint32_t lds_push(bool want, uint32_t *lds_counter){
    unsigned long long mask = __ballot(want);
    if(mask == 0ull) return 0u;
    uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
    uint32_t base = 0u;
    int leader = __ffsll((long long) mask) - 1;
    if((int) (threadIdx.x & 63u) == leader) base = atomicAdd(lds_counter, (uint32_t) __popcll(mask));
    base = (uint32_t) __shfl((int) base, leader, 64);
    return base + prefix;
}

HPT_DEV bool tile_to_pixel(const Tiling &tl, uint32_t p, int &x, int &y){
    uint32_t ts2 = (uint32_t) (tl.tile * tl.tile);
    uint32_t lt = p / ts2, q = p % ts2;
    uint32_t gt = lt * (uint32_t) tl.world + (uint32_t) tl.rank;
    if(gt >= (uint32_t) tl.ntiles) return false;
    uint32_t ty = gt / (uint32_t) tl.tiles_x, tx = (gt % (uint32_t) tl.tiles_x + ty) % (uint32_t) tl.tiles_x;   // rows rotated, as in pt_kernels.hip
    uint32_t sub = q >> 6, l = q & 63u;
    uint32_t spr = (uint32_t) tl.tile >> 3;
    uint32_t bx = sub % spr, by = sub / spr;
    x = (int) (tx * (uint32_t) tl.tile + bx * 8u + (l & 7u));
    y = (int) (ty * (uint32_t) tl.tile + by * 8u + (l >> 3));
    return x < tl.W && y < tl.H;
}

// ---- CPU scene model: primitive tests (reference src/object.cpp) --------------------------------
// AABB::intersectAABB, src/object.cpp:104-121 (degenerate boxes are pre-widened on the host)
HPT_DEV bool group_box_hit(const DevGroup &g, f3 ro, f3 rd, float tmin, float tmax){
    float o[3] = { ro.x, ro.y, ro.z }, d[3] = { rd.x, rd.y, rd.z };
#pragma unroll
    for(int a = 0; a < 3; ++a){
        float invD = 1.0f / d[a];
        float t0 = (g.mn[a] - o[a]) * invD;
        float t1 = (g.mx[a] - o[a]) * invD;
        if(invD < 0){ float tmp = t0; t0 = t1; t1 = tmp; }
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if(tmax <= tmin) return false;
    }
    return true;
}
// Sphere::check_intersect, src/object.cpp:16-44 (scale 1)
HPT_DEV bool cpu_sphere(f3 center, float r, f3 O, f3 vec, float tMin, float tMax, float &t){
    f3 D = gnormalize(vec);
    f3 OC = O - center;
    float a = 1.0f * D.x * D.x + 1.0f * D.y * D.y + 1.0f * D.z * D.z;
    float b = 2.0f * (1.0f * D.x * OC.x + 1.0f * D.y * OC.y + 1.0f * D.z * OC.z);
    float c = 1.0f * OC.x * OC.x + 1.0f * OC.y * OC.y + 1.0f * OC.z * OC.z - r * r;
    if(c <= 1e-6f) return false;
    float disc = b * b - 4.0f * a * c;
    if(disc < 0.0f) return false;
    float sdisc = sqrtf(fmaxf(0.0f, disc));
    float t0 = (-b - sdisc) / (2.0f * a);
    float t1 = (-b + sdisc) / (2.0f * a);
    if(t0 > t1){ float tmp = t0; t0 = t1; t1 = tmp; }
    float tc = (t0 >= tMin) ? t0 : t1;
    if(tc < tMin || tc > tMax) return false;
    t = tc;
    return true;
}
// Triangle::check_intersect, src/object.cpp:72-95 (determinant compared in double like the reference)
HPT_DEV bool cpu_triangle(f3 v0, f3 e1, f3 e2, f3 O, f3 vec, float tMin, float tMax, float &t){
    f3 pvec = cross3(vec, e2);
    float det = dot3(e1, pvec);
    if((double) fabsf(det) < 1e-6) return false;
    float invDet = 1.0f / det;
    f3 tvec = O - v0;
    float u = dot3(tvec, pvec) * invDet;
    if(u < 0.0f || u > 1.0f) return false;
    f3 qvec = cross3(tvec, e1);
    float v = dot3(vec, qvec) * invDet;
    if(v < 0.0f || (u + v) > 1.0f) return false;
    float tt = dot3(e2, qvec) * invDet;
    if(tt < tMin || tt > tMax) return false;
    t = tt;
    return true;
}

// Walks one group's BVH.  Closest: candidates with t <= best_t, ties to the HIGHER iteration order
// (the CPU loop accepts t <= best.t in insertion order).  ANY: true at the first opaque hit.
// work of the connection shadow rays of one lane (HPT_FLAG_COUNT_WORK renders of the bidirectional path)
struct BdTally { uint32_t nodes, tris, spheres, group_boxes; };

template <bool ANY, bool COUNT = false>
HPT_DEV bool bd_walk(const BdptSceneDev &sc, uint32_t root, f3 ro, f3 rd, float tMin, float tmax, uint32_t *stk,
                     float &best_t, uint32_t &best_code, uint32_t &best_seq, BdTally *tally = nullptr){
    if(root == kEmptyChild) return false;
    float dx = fabsf(rd.x) > 1e-20f ? rd.x : copysignf(1e-20f, rd.x);
    float dy = fabsf(rd.y) > 1e-20f ? rd.y : copysignf(1e-20f, rd.y);
    float dz = fabsf(rd.z) > 1e-20f ? rd.z : copysignf(1e-20f, rd.z);
    float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    float ox = ro.x * ix, oy = ro.y * iy, oz = ro.z * iz;
    float limit = ANY ? tmax : best_t;
    uint32_t cur = root;
    int sp = 0;
    for(;;){
        bool descend = false;
        if(!(cur & kLeafFlag)){
            const float4 *n = sc.nodes + (size_t) cur * 4;
            float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
            if(COUNT) tally->nodes += 1u;
            float a0 = fmaf(n0.x, ix, -ox), a1 = fmaf(n1.x, ix, -ox);
            float b0 = fmaf(n0.y, iy, -oy), b1 = fmaf(n1.y, iy, -oy);
            float c0 = fmaf(n0.z, iz, -oz), c1 = fmaf(n1.z, iz, -oz);
            float ln = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fmaxf(fminf(c0, c1), 0.0f));
            float lf = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fminf(fmaxf(c0, c1), limit));
            a0 = fmaf(n2.x, ix, -ox); a1 = fmaf(n3.x, ix, -ox);
            b0 = fmaf(n2.y, iy, -oy); b1 = fmaf(n3.y, iy, -oy);
            c0 = fmaf(n2.z, iz, -oz); c1 = fmaf(n3.z, iz, -oz);
            float rn = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fmaxf(fminf(c0, c1), 0.0f));
            float rf = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fminf(fmaxf(c0, c1), limit));
            uint32_t lc = f2u(n0.w), rc = f2u(n1.w);
            bool hl = (ln <= lf * 1.000002f) && (lc != kEmptyChild);
            bool hr = (rn <= rf * 1.000002f) && (rc != kEmptyChild);
            if(hl && hr){
                bool left_first = ln <= rn;
                stk[sp * kBlock] = left_first ? rc : lc;
                ++sp;
                cur = left_first ? lc : rc;
                descend = true;
            } else if(hl){ cur = lc; descend = true; }
            else if(hr){ cur = rc; descend = true; }
        } else {
            uint32_t first = (cur & 0x7FFFFFFFu) >> 3;
            uint32_t cnt = (cur & 7u) + 1u;
            for(uint32_t k = 0; k < cnt; ++k){
                const float4 *tp = sc.tris + (size_t) (first + k) * 3;
                float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                float t;
                if(COUNT) tally->tris += 1u;
                if(cpu_triangle(xyz(t0), xyz(t1), xyz(t2), ro, rd, tMin, ANY ? tmax : best_t, t)){
                    if(ANY){
                        if(f2u(t2.w) & 1u) return true;
                    } else {
                        uint32_t seq = f2u(t0.w);
                        if(t < best_t || seq > best_seq || best_code == kBdMiss){
                            best_t = t; best_code = first + k; best_seq = seq; limit = t;
                        }
                    }
                }
            }
        }
        if(!descend){
            if(sp == 0) break;
            --sp;
            cur = stk[sp * kBlock];
        }
    }
    return false;
}

// cpu_find_closest_hit, src/cpu_bdpt.cpp:30-80
HPT_DEV void bd_closest(const BdptSceneDev &sc, f3 ro, f3 rd, uint32_t *stk, float &t_out, uint32_t &code_out){
    float best_t = 1e20f;
    uint32_t best = kBdMiss, best_seq = 0u;
    for(int gi = 0; gi < sc.num_groups; ++gi){
        DevGroup g = sc.groups[gi];
        if(!group_box_hit(g, ro, rd, 1e-4f, best_t)) continue;
        for(uint32_t k = 0; k < g.sphere_count; ++k){
            DevRound s = sc.spheres[g.sphere_first + k];
            float t;
            if(cpu_sphere(mk3(s.c[0], s.c[1], s.c[2]), s.r, ro, rd, 1e-4f, best_t, t)){
                uint32_t seq = s.pad[0];
                if(t < best_t || seq > best_seq || best == kBdMiss){ best_t = t; best = kBdSphere | (g.sphere_first + k); best_seq = seq; }
            }
        }
        bd_walk<false>(sc, g.root, ro, rd, 1e-4f, best_t, stk, best_t, best, best_seq);
    }
    for(int i = 0; i < sc.num_lights; ++i){
        const DevLight &L = sc.lights[i];
        float t;
        if(hit_sphere(ro, rd, ld3(L.ball_c), L.r, best_t, t)){ best_t = t; best = kBdLight | (uint32_t) i; }
    }
    t_out = best_t; code_out = best;
}

// cpu_check_visibility, src/cpu_bdpt.cpp:82-107
template <bool COUNT = false>
HPT_DEV bool bd_visible(const BdptSceneDev &sc, f3 p1, f3 p2, uint32_t *stk, BdTally *tally = nullptr){
    f3 diff = p2 - p1;
    float dist = length3(diff);
    f3 dir = diff / dist;
    float max_dist = dist - 1e-3f;
    for(int gi = 0; gi < sc.num_groups; ++gi){
        DevGroup g = sc.groups[gi];
        if(COUNT) tally->group_boxes += 1u;
        if(!group_box_hit(g, p1, dir, 1e-3f, max_dist)) continue;
        for(uint32_t k = 0; k < g.sphere_count; ++k){
            DevRound s = sc.spheres[g.sphere_first + k];
            float t;
            if(COUNT) tally->spheres += 1u;
            if(cpu_sphere(mk3(s.c[0], s.c[1], s.c[2]), s.r, p1, dir, 1e-3f, max_dist, t) && (s.flags & 1u)) return false;
        }
        float bt = max_dist; uint32_t code = 0, seq = 0;
        if(bd_walk<true, COUNT>(sc, g.root, p1, dir, 1e-3f, max_dist, stk, bt, code, seq, tally)) return false;
    }
    return true;
}

struct BdHit { f3 pos, normal; Mat m; bool is_light; };

// position, flipped normal and material of a hit (cpu_bdpt.cpp:47-57, 69-76)
HPT_DEV BdHit bd_resolve(const BdptSceneDev &sc, f3 ro, f3 rd, float t, uint32_t code){
    BdHit h;
    h.pos = ro + rd * t;
    h.is_light = false;
    if((code & kBdLight) == kBdLight){
        const DevLight &L = sc.lights[code & 0x3FFFFFFFu];
        h.m.base = ld3(L.illum); h.m.eta = 0.0f; h.m.roughness = 1.0f; h.m.metallic = 0.0f;
        h.normal = normalize3(h.pos - ld3(L.ball_c));
        h.is_light = true;
        if(dot3(h.normal, rd) > 0.0f) h.normal = h.normal * -1.0f;
        return h;
    }
    uint32_t mat;
    if(code & kBdSphere){
        DevRound s = sc.spheres[code & 0x3FFFFFFFu];
        f3 n = gnormalize(mk3(1.0f * (h.pos.x - s.c[0]), 1.0f * (h.pos.y - s.c[1]), 1.0f * (h.pos.z - s.c[2])));
        if(dot3(n, rd) > 0.0f) n = mk3(-n.x, -n.y, -n.z);
        h.normal = n; mat = s.material;
    } else {
        const float4 *tp = sc.tris + (size_t) code * 3;
        float4 t1 = tp[1], t2 = tp[2];
        f3 n = gnormalize(cross3(xyz(t1), xyz(t2)));
        if(dot3(n, rd) > 0.0f) n = mk3(-n.x, -n.y, -n.z);
        h.normal = n; mat = f2u(t1.w);
    }
    DevMaterial dm = sc.mats[mat];
    h.m.base = ld3(dm.base); h.m.roughness = dm.roughness; h.m.metallic = dm.metallic; h.m.eta = dm.eta;
    return h;
}

HPT_DEV f3 bsdf_value(const Mat &m, f3 wo_w, f3 wi_w, f3 N){
    ShadeCtx c = make_shade_ctx(N, wo_w);
    f3 f; float pdf;
    bsdf_eval_pdf<true, false>(m, c, wi_w, f, pdf);
    return f;
}
HPT_DEV float bsdf_pdf_only(const Mat &m, f3 wo_w, f3 wi_w, f3 N){
    ShadeCtx c = make_shade_ctx(N, wo_w);
    f3 f; float pdf;
    bsdf_eval_pdf<false, true>(m, c, wi_w, f, pdf);
    return pdf;
}

HPT_DEV void store_lv(LightVertexDev *lv, f3 pos, f3 normal, f3 thr, const Mat &m, uint32_t flags, float cutoff, float pf, float pr){
    lv->pos[0] = pos.x; lv->pos[1] = pos.y; lv->pos[2] = pos.z; lv->pdf_fwd = pf;
    lv->normal[0] = normal.x; lv->normal[1] = normal.y; lv->normal[2] = normal.z; lv->pdf_rev = pr;
    lv->thr[0] = thr.x; lv->thr[1] = thr.y; lv->thr[2] = thr.z; lv->source_cutoff = cutoff;
    lv->base[0] = m.base.x; lv->base[1] = m.base.y; lv->base[2] = m.base.z; lv->roughness = m.roughness;
    lv->metallic = m.metallic; lv->eta = m.eta; lv->flags = flags; lv->pad = 0u;
}

// ---- 1. light subpaths, src/cpu_bdpt.cpp:217-325 ------------------------------------------------
__global__ __launch_bounds__(kBlock)
void k_bdpt_light_trace(BdptSceneDev sc, LightVertexDev *lvs, int total_paths, int light_depth, int spl, uint64_t seed, int max_delta){
    __shared__ uint32_t s_stack[kStackDepth * kBlock];
    int idx = blockIdx.x * kBlock + threadIdx.x;
    if(idx >= total_paths) return;
    uint32_t *stk = s_stack + threadIdx.x;
    uint64_t rs = rng_seed(seed ^ 0x4C49474854ull, (uint32_t) idx, 0u);
    int light_idx = idx % sc.num_lights;
    const DevLight &L = sc.lights[light_idx];
    LightVertexDev *base = lvs + (size_t) idx * light_depth;
    Mat none; none.base = mk3(0, 0, 0); none.roughness = 0; none.metallic = 0; none.eta = 0;
    for(int d = 0; d < light_depth; ++d) store_lv(base + d, mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0), none, 0u, 0.0f, 0.0f, 0.0f);
    f3 ray_o, ray_d;
    float ray_eta = 1.0f;
    f3 w = normalize3(ld3(L.raw_dir));
    f3 u_vec = (fabsf(w.x) > 0.9f) ? mk3(0, 1, 0) : mk3(1, 0, 0);
    f3 v_vec = normalize3(cross3(w, u_vec));
    u_vec = normalize3(cross3(v_vec, w));
    if(L.is_parallel){
        ray_d = w;
        f3 mn = ld3(sc.scene_min), mx = ld3(sc.scene_max);
        f3 center = (mn + mx) * 0.5f;
        float radius = length3(mx - mn) * 0.5f;
        float r1 = rng_next(rs), r2 = rng_next(rs);
        float offset_u = (r1 - 0.5f) * radius * 2.0f;
        float offset_v = (r2 - 0.5f) * radius * 2.0f;
        ray_o = center - ray_d * (radius * 2.0f) + u_vec * offset_u + v_vec * offset_v;
    } else {
        float u1 = rng_next(rs), u2 = rng_next(rs);
        // cone sample: cos(theta) = 1 - u1 (1 - cos cutoff); sin from cos, phi by the shared polynomial
        // (the CPU reference goes through acosf/sinf/cosf; its streams are not reproducible on a GPU anyway)
        float cos_t = 1.0f - u1 * (1.0f - L.cos_cutoff);
        float sin_t = sqrtf(fmaxf(0.0f, 1.0f - cos_t * cos_t));
        float sp, cp; sincos_2pi(u2, sp, cp);
        f3 local_dir = mk3(sin_t * cp, sin_t * sp, cos_t);
        ray_d = normalize3(u_vec * local_dir.x + v_vec * local_dir.y + w * local_dir.z);
        ray_o = ld3(L.pos) + ray_d * L.r;
    }
    f3 throughput = ld3(L.illum) / fmaxf((float) spl, 1.0f);
    store_lv(base, ray_o, ray_d, throughput, none, 1u | (L.is_parallel ? 2u : 0u), L.cutoff, 0.0f, 0.0f);
    f3 last_normal = ray_d, last_pos = ray_o;
    float last_pdf_omega = 1.0f / kPi;
    int deltas = 0;
    for(int depth = 1; depth < light_depth; depth++){
        LightVertexDev *vx = base + depth;
        vx->thr[0] = vx->thr[1] = vx->thr[2] = 0.0f;
        float t; uint32_t code;
        bd_closest(sc, ray_o, ray_d, stk, t, code);
        if(code == kBdMiss) break;
        BdHit hit = bd_resolve(sc, ray_o, ray_d, t, code);
        if(hit.is_light){
            store_lv(vx, hit.pos, hit.normal, throughput, hit.m, 1u, 0.0f, vx->pdf_fwd, vx->pdf_rev);
            break;
        }
        if(length3(throughput) < 1e-4f) break;
        float dist2 = dot3(hit.pos - last_pos, hit.pos - last_pos);
        if(dist2 < 1e-6f) break;
        float cos_at_hit = fabsf(dot3(hit.normal, ray_d * -1.0f));
        float cos_at_prev = fabsf(dot3(last_normal, ray_d));
        float pdf_fwd = last_pdf_omega * cos_at_hit / dist2;
        f3 wo = ray_d * -1.0f;
        f3 wi, bsdf_val; float pdf_omega, new_eta; bool is_delta;
        float u_rr = rng_next(rs), u1 = rng_next(rs), u2 = rng_next(rs);
        ShadeCtx ctx = make_shade_ctx(hit.normal, wo);
        bsdf_sample(hit.m, ctx, u_rr, u1, u2, ray_eta, wi, bsdf_val, pdf_omega, is_delta, new_eta);
        if(pdf_omega <= 0.0f) break;
        if(is_delta){
            throughput = throughput * bsdf_val;
            ray_d = wi; ray_eta = new_eta;
            ray_o = hit.pos + hit.normal * (dot3(wi, hit.normal) < 0.0f ? -kEps : kEps);
            if(++deltas > max_delta) break;
            depth--;
            continue;
        }
        float pdf_rev_omega = bsdf_pdf_only(hit.m, wi, wo, hit.normal);
        float pdf_rev = pdf_rev_omega * cos_at_prev / dist2;
        store_lv(vx, hit.pos, hit.normal, throughput, hit.m, 0u, vx->source_cutoff, pdf_fwd, pdf_rev);
        throughput = throughput * bsdf_val * fabsf(dot3(hit.normal, wi)) / pdf_omega;
        if(!is_valid_color(throughput)) break;
        ray_d = wi;
        ray_o = hit.pos + hit.normal * kEps;
        last_pdf_omega = pdf_omega; last_normal = hit.normal; last_pos = hit.pos;
    }
}

// ---- 2. eye paths -------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock)
void k_bdpt_generate(Tiling tl, CameraDev cam, PathBuf pb, BdptPathBuf bp, uint32_t *qcount, uint32_t total,
                     uint32_t first_sample, uint64_t seed){
    if(blockIdx.x == 0 && threadIdx.x == 0) *qcount = total;
    uint32_t stride = gridDim.x * kBlock;
    for(uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += stride){
        uint32_t p = i % (uint32_t) tl.n_local, j = i / (uint32_t) tl.n_local;
        int px, py;
        bool active = tile_to_pixel(tl, p, px, py);
        if(active){
            uint64_t rs = rng_seed(seed, (uint32_t) (py * tl.W + px), first_sample + j);
            float pixel_x = (float) px + rng_next(rs);
            float pixel_y = (float) py + rng_next(rs);
            f3 eye = mk3(cam.eye[0], cam.eye[1], cam.eye[2]);
            f3 pixel_pos = mk3(cam.UL[0], cam.UL[1], cam.UL[2]) + mk3(cam.dx[0], cam.dx[1], cam.dx[2]) * pixel_x
                           + mk3(cam.dy[0], cam.dy[1], cam.dy[2]) * pixel_y;
            f3 dir = normalize3(pixel_pos - eye);
            pb.org_eta[i] = make_float4(eye.x, eye.y, eye.z, 1.0f);
            pb.dir_flags[i] = make_float4(dir.x, dir.y, dir.z, u2f(0u));                  // depth 0, no delta bounces
            pb.thr[i] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
            pb.rng[i] = make_uint2((uint32_t) rs, (uint32_t) (rs >> 32));
            bp.last_pos_pdf[i] = make_float4(eye.x, eye.y, eye.z, 1.0f);                  // cpu_bdpt.cpp:359-361
            bp.last_normal[i] = make_float4(dir.x, dir.y, dir.z, 0.0f);
        } else {
            pb.dir_flags[i] = make_float4(0.0f, 0.0f, 1.0f, u2f(2u));
            pb.hit[i] = make_uint2(f2u(1e20f), kBdMiss);
        }
        pb.col[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

__global__ __launch_bounds__(kBlock)
void k_bdpt_extend(BdptSceneDev sc, PathBuf pb, const uint32_t *queue, const uint32_t *qcount){
    __shared__ uint32_t s_stack[kStackDepth * kBlock];
    uint32_t count = *qcount;
    uint32_t *stk = s_stack + threadIdx.x;
    for(uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock){
        uint32_t path = queue ? queue[i] : i;
        float4 o = pb.org_eta[path], d = pb.dir_flags[path];
        if(f2u(d.w) & 2u) continue;
        float t; uint32_t code;
        bd_closest(sc, xyz(o), xyz(d), stk, t, code);
        pb.hit[path] = make_uint2(f2u(t), code);
    }
}

constexpr int kVtxChunk = 2048;
constexpr int kVtxTargetGroups = 1024;

// one eye-loop iteration minus the connection loop, src/cpu_bdpt.cpp:362-384 and 443-472
__global__ __launch_bounds__(kBlock)
void k_bdpt_vertex(BdptSceneDev sc, PathBuf pb, BdptPathBuf bp, const uint32_t *queue, const uint32_t *qcount,
                   uint32_t *next_queue, uint32_t *next_count, uint32_t *cqueue, uint32_t *ccount,
                   int eye_depth, int max_delta, uint32_t slots, float ex, float ey, float ez){
    __shared__ uint32_t s_next[kVtxChunk];
    __shared__ uint32_t s_conn[kVtxChunk];
    __shared__ uint32_t s_cnt[4];
    if(threadIdx.x < 4) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t count = *qcount;
    uint32_t chunk = (count + kVtxTargetGroups - 1) / kVtxTargetGroups;
    chunk = (chunk + kBlock - 1) / kBlock * kBlock;
    chunk = chunk < (uint32_t) kBlock ? (uint32_t) kBlock : (chunk > (uint32_t) kVtxChunk ? (uint32_t) kVtxChunk : chunk);
    uint32_t begin = blockIdx.x * chunk;
    uint32_t end = begin + chunk < count ? begin + chunk : count;
    for(uint32_t base = begin; base < end; base += kBlock){
        uint32_t i = base + threadIdx.x;
        bool alive = false, connect = false;
        uint32_t path = 0;
        if(i < count){
            path = queue ? queue[i] : i;
            uint2 h = pb.hit[path];
            if(h.y != kBdMiss){
                float4 o4 = pb.org_eta[path], d4 = pb.dir_flags[path], th4 = pb.thr[path];
                f3 ro = xyz(o4), rd = xyz(d4), throughput = xyz(th4);
                float ray_eta = o4.w;
                uint32_t flags = f2u(d4.w);
                int depth = (int) ((flags >> 8) & 0xFFu);
                int deltas = (int) ((flags >> 16) & 0xFFFFu);
                BdHit hit = bd_resolve(sc, ro, rd, u2f(h.x), h.y);
                if(hit.is_light && depth == 0){                      // cpu_bdpt.cpp:372-375
                    float4 col = pb.col[path];
                    col.x = col.x + hit.m.base.x; col.y = col.y + hit.m.base.y; col.z = col.z + hit.m.base.z;
                    pb.col[path] = col;
                } else {
                    float4 lp = bp.last_pos_pdf[path], ln4 = bp.last_normal[path];
                    f3 last_pos = xyz(lp), last_normal = xyz(ln4);
                    float last_pdf_omega = lp.w;
                    float pdf_fwd = 1.0f;
                    if(depth > 0){
                        float d2 = dot3(hit.pos - last_pos, hit.pos - last_pos);
                        float cos_at_hit = fabsf(dot3(hit.normal, rd * -1.0f));
                        pdf_fwd = last_pdf_omega * cos_at_hit / fmaxf(d2, 1e-6f);
                    }
                    f3 wo = rd * -1.0f;
                    // the vertex the connect kernel will join to every light vertex
                    connect = true;
                    bp.vtx_pos[path] = make_float4(hit.pos.x, hit.pos.y, hit.pos.z, hit.m.roughness);
                    bp.vtx_nrm[path] = make_float4(hit.normal.x, hit.normal.y, hit.normal.z, hit.m.metallic);
                    bp.vtx_thr[path] = make_float4(throughput.x, throughput.y, throughput.z, hit.m.eta);
                    bp.vtx_wo[path] = make_float4(wo.x, wo.y, wo.z, u2f((uint32_t) depth));
                    bp.hist_pos_eta[(size_t) depth * slots + path] = make_float4(hit.pos.x, hit.pos.y, hit.pos.z, hit.m.eta);
                    bp.vtx_base[path] = make_float4(hit.m.base.x, hit.m.base.y, hit.m.base.z, 0.0f);
                    {   // shading contexts of this vertex for the connection kernel: what every one of its connections
                        // would otherwise rebuild (cpu_bdpt.cpp:401 evaluates the BSDF with the vertex normal,
                        // cpu_calculate_mis_weight :119-131 with the re-normalised one and the direction to the previous vertex)
                        const float alpha = roughness_to_alpha(hit.m.roughness);
                        ShadeCtx ce = make_shade_ctx(hit.normal, wo);
                        float lam_e = ggx_lambda(ce.wo, alpha);
                        f3 diffuse = hit.m.base / kPi * (1.0f - hit.m.metallic);
                        f3 ns = normalize3(hit.normal);
                        f3 wo_s_w = (depth == 0) ? normalize3(mk3(ex, ey, ez) - hit.pos)
                                                 : normalize3(xyz(bp.hist_pos_eta[(size_t) (depth - 1) * slots + path]) - hit.pos);
                        ShadeCtx cs = make_shade_ctx(ns, wo_s_w);
                        float lam_s = ggx_lambda(cs.wo, alpha);
                        float4 *e = bp.ectx + path;
                        e[0 * (size_t) slots] = make_float4(ce.T.x, ce.T.y, ce.T.z, ce.B.x);
                        e[1 * (size_t) slots] = make_float4(ce.B.y, ce.B.z, ce.wo.x, ce.wo.y);
                        e[2 * (size_t) slots] = make_float4(ce.wo.z, lam_e, diffuse.x, diffuse.y);
                        e[3 * (size_t) slots] = make_float4(diffuse.z, ns.x, ns.y, ns.z);
                        e[4 * (size_t) slots] = make_float4(cs.T.x, cs.T.y, cs.T.z, cs.B.x);
                        e[5 * (size_t) slots] = make_float4(cs.B.y, cs.B.z, cs.wo.x, cs.wo.y);
                        e[6 * (size_t) slots] = make_float4(cs.wo.z, lam_s, 0.0f, 0.0f);
                    }

                    uint2 r2 = pb.rng[path];
                    uint64_t rs = ((uint64_t) r2.y << 32) | (uint64_t) r2.x;
                    float u_rr = rng_next(rs), u1 = rng_next(rs), u2 = rng_next(rs);
                    ShadeCtx ctx = make_shade_ctx(hit.normal, wo);
                    f3 wi, bsdf_val; float pdf_omega, new_eta; bool is_delta;
                    bsdf_sample(hit.m, ctx, u_rr, u1, u2, ray_eta, wi, bsdf_val, pdf_omega, is_delta, new_eta);
                    if(!(pdf_omega <= 0.0f)){
                        f3 new_o;
                        if(is_delta){
                            throughput = throughput * bsdf_val;
                            ray_eta = new_eta;
                            new_o = hit.pos + hit.normal * (dot3(wi, hit.normal) < 0.0f ? -kEps : kEps);
                            last_pos = hit.pos; last_normal = hit.normal; last_pdf_omega = 1.0f;
                            ++deltas;
                            alive = deltas <= max_delta;
                        } else {
                            float pdf_rev_omega = bsdf_pdf_only(hit.m, wi, wo, hit.normal);
                            float d2 = dot3(hit.pos - last_pos, hit.pos - last_pos);
                            float cos_at_prev = fabsf(dot3(last_normal, rd));
                            float pdf_rev = pdf_rev_omega * cos_at_prev / fmaxf(d2, 1e-6f);
                            bp.hist_pdf[(size_t) depth * slots + path] = make_float2(pdf_fwd, pdf_rev);
                            throughput = throughput * bsdf_val * fabsf(dot3(hit.normal, wi)) / pdf_omega;
                            new_o = hit.pos + hit.normal * kEps;
                            last_pdf_omega = pdf_omega; last_normal = hit.normal; last_pos = hit.pos;
                            ++depth;
                            alive = is_valid_color(throughput) && depth < eye_depth;
                        }
                        if(alive){
                            uint32_t nf = ((uint32_t) depth << 8) | ((uint32_t) deltas << 16);
                            pb.org_eta[path] = make_float4(new_o.x, new_o.y, new_o.z, ray_eta);
                            pb.dir_flags[path] = make_float4(wi.x, wi.y, wi.z, u2f(nf));
                            pb.thr[path] = make_float4(throughput.x, throughput.y, throughput.z, 0.0f);
                            pb.rng[path] = make_uint2((uint32_t) rs, (uint32_t) (rs >> 32));
                            bp.last_pos_pdf[path] = make_float4(last_pos.x, last_pos.y, last_pos.z, last_pdf_omega);
                            bp.last_normal[path] = make_float4(last_normal.x, last_normal.y, last_normal.z, 0.0f);
                        }
                    }
                }
            }
        }
        uint32_t cpos = lds_push(connect, &s_cnt[1]);
        if(connect) s_conn[cpos] = path;
        uint32_t qpos = lds_push(alive, &s_cnt[0]);
        if(alive) s_next[qpos] = path;
    }
    __syncthreads();
    if(threadIdx.x == 0){
        s_cnt[2] = s_cnt[0] ? atomicAdd(next_count, s_cnt[0]) : 0u;
        s_cnt[3] = s_cnt[1] ? atomicAdd(ccount, s_cnt[1]) : 0u;
    }
    __syncthreads();
    for(uint32_t k = threadIdx.x; k < s_cnt[0]; k += kBlock) next_queue[s_cnt[2] + k] = s_next[k];
    for(uint32_t k = threadIdx.x; k < s_cnt[1]; k += kBlock) cqueue[s_cnt[3] + k] = s_conn[k];
}

HPT_DEV Mat lv_mat(const LightVertexDev &lv){
    Mat m; m.base = ld3(lv.base); m.roughness = lv.roughness; m.metallic = lv.metallic; m.eta = lv.eta;
    return m;
}

// per light vertex, once per render: see LightVertexCtx
__global__ __launch_bounds__(kBlock)
void k_bdpt_light_ctx(const LightVertexDev *lvs, LightVertexCtx *out, int n_lv, int light_depth){
    int j = blockIdx.x * kBlock + threadIdx.x;
    if(j >= n_lv) return;
    const LightVertexDev lv = lvs[j];
    const int t_idx = j % light_depth;
    const Mat m = lv_mat(lv);
    const float alpha = roughness_to_alpha(m.roughness);
    f3 N = ld3(lv.normal), pos = ld3(lv.pos);
    f3 to_prev = t_idx > 0 ? normalize3(ld3(lvs[j - 1].pos) - pos) : mk3(0, 0, 0);
    ShadeCtx cl = make_shade_ctx(N, to_prev);
    f3 nt = normalize3(N);
    f3 wo_t_w = (t_idx == 0) ? normalize3(N) : to_prev;
    ShadeCtx ct = make_shade_ctx(nt, wo_t_w);
    LightVertexCtx c;
    c.T[0] = cl.T.x; c.T[1] = cl.T.y; c.T[2] = cl.T.z; c.B[0] = cl.B.x; c.B[1] = cl.B.y; c.B[2] = cl.B.z;
    c.wo_l[0] = cl.wo.x; c.wo_l[1] = cl.wo.y; c.wo_l[2] = cl.wo.z;
    c.nt[0] = nt.x; c.nt[1] = nt.y; c.nt[2] = nt.z;
    c.Tn[0] = ct.T.x; c.Tn[1] = ct.T.y; c.Tn[2] = ct.T.z; c.Bn[0] = ct.B.x; c.Bn[1] = ct.B.y; c.Bn[2] = ct.B.z;
    c.wo_t[0] = ct.wo.x; c.wo_t[1] = ct.wo.y; c.wo_t[2] = ct.wo.z;
    c.lam_l = ggx_lambda(cl.wo, alpha); c.lam_t = ggx_lambda(ct.wo, alpha);
    f3 diffuse = m.base / kPi * (1.0f - m.metallic);
    c.diffuse[0] = diffuse.x; c.diffuse[1] = diffuse.y; c.diffuse[2] = diffuse.z;
    c.pad[0] = c.pad[1] = 0.0f;
    out[j] = c;
}

// the two shading contexts of an eye vertex as k_bdpt_vertex stored them; each is loaded where it is used (loading
// both up front costs the connection kernel a wave of occupancy: 133 instead of 114 VGPRs)
HPT_DEV void load_eye_value_ctx(const BdptPathBuf &bp, uint32_t path, uint32_t slots, f3 v_n, ShadeCtx &c, ShadePre &pre){
    const float4 *p = bp.ectx + path;
    float4 a = p[0], b = p[(size_t) slots], d = p[2 * (size_t) slots], e = p[3 * (size_t) slots];
    c.T = mk3(a.x, a.y, a.z); c.B = mk3(a.w, b.x, b.y); c.N = v_n; c.wo = mk3(b.z, b.w, d.x);
    pre.lam_o = d.y; pre.diffuse = mk3(d.z, d.w, e.x);
}
HPT_DEV void load_eye_mis_ctx(const BdptPathBuf &bp, uint32_t path, uint32_t slots, ShadeCtx &c, ShadePre &pre){
    const float4 *p = bp.ectx + path;
    float4 d = p[3 * (size_t) slots], e = p[4 * (size_t) slots], f = p[5 * (size_t) slots], g = p[6 * (size_t) slots];
    c.N = mk3(d.y, d.z, d.w); c.T = mk3(e.x, e.y, e.z); c.B = mk3(e.w, f.x, f.y); c.wo = mk3(f.z, f.w, g.x);
    pre.lam_o = g.y; pre.diffuse = mk3(0, 0, 0);          // the pdf does not use the diffuse lobe
}

// cpu_calculate_mis_weight, src/cpu_bdpt.cpp:112-167.  The current eye vertex still carries the
// placeholder pdfs (0, 1) when the CPU connects it (cpu_bdpt.cpp:385); earlier ones their final values.
// The two solid-angle pdfs are evaluated in the per-vertex contexts (re-normalised normals, directions to the
// previous vertices) that k_bdpt_vertex / k_bdpt_light_ctx prepared.
HPT_DEV float bd_mis_weight(const BdptPathBuf &bp, uint32_t path, uint32_t slots, int s_idx, const Mat &ev_m,
                            const LightVertexDev *light_path, const LightVertexCtx *lcp, int t_idx, f3 dir_e_to_l, float dist2){
    const LightVertexDev &lv = light_path[t_idx];
    ShadeCtx cs; ShadePre ps;
    load_eye_mis_ctx(bp, path, slots, cs, ps);
    f3 ns = cs.N;
    f3 nt = ld3(lcp->nt);
    float cos_s = fmaxf(0.0f, dot3(ns, dir_e_to_l));
    float cos_t = fmaxf(0.0f, dot3(nt, dir_e_to_l * -1.0f));
    if(cos_s <= 0.0f || cos_t <= 0.0f || dist2 < 1e-6f) return 0.0f;
    f3 unused; float p_s, p_t;
    bsdf_eval_pdf<false, true>(ev_m, cs, dir_e_to_l, unused, p_s, &ps);
    ShadeCtx ct; ct.T = ld3(lcp->Tn); ct.B = ld3(lcp->Bn); ct.N = nt; ct.wo = ld3(lcp->wo_t);
    ShadePre pt; pt.lam_o = lcp->lam_t; pt.diffuse = mk3(0, 0, 0);
    bsdf_eval_pdf<false, true>(lv_mat(lv), ct, dir_e_to_l * -1.0f, unused, p_t, &pt);
    float pdf_omega_s = fmaxf(p_s, 1e-6f);
    float pdf_omega_t = fmaxf(p_t, 1e-6f);
    float pdf_s_to_t = pdf_omega_s * cos_t / dist2;
    float pdf_t_to_s = pdf_omega_t * cos_s / dist2;
    float sum_ratios = 1.0f;
    float current_ratio = 1.0f;
    float prev_pdf_rev = pdf_t_to_s;
    for(int i = s_idx; i > 0; --i){
        float eta_i, pf, pr;
        if(i == s_idx){ eta_i = ev_m.eta; pf = 0.0f; pr = 1.0f; }
        else {
            eta_i = bp.hist_pos_eta[(size_t) i * slots + path].w;
            float2 pp = bp.hist_pdf[(size_t) i * slots + path]; pf = pp.x; pr = pp.y;
        }
        if(eta_i > 0.0f) break;
        current_ratio *= prev_pdf_rev / fmaxf(pf, 1e-8f);
        sum_ratios += current_ratio;
        prev_pdf_rev = pr;
    }
    current_ratio = 1.0f;
    prev_pdf_rev = pdf_s_to_t;
    for(int i = t_idx; i > 0; --i){
        const LightVertexDev &li = light_path[i];
        if(li.flags & 1u){
            current_ratio *= prev_pdf_rev / fmaxf(li.pdf_fwd, 1e-8f);
            sum_ratios += current_ratio;
            break;
        }
        if(li.eta > 0.0f) break;
        current_ratio *= prev_pdf_rev / fmaxf(li.pdf_fwd, 1e-8f);
        sum_ratios += current_ratio;
        prev_pdf_rev = li.pdf_rev;
    }
    if(is_nan(sum_ratios) || is_inf(sum_ratios) || sum_ratios <= 0.0f) return 0.0f;
    return 1.0f / sum_ratios;
}

// connection loop body, src/cpu_bdpt.cpp:389-439.  A workgroup takes 4 (eye vertex, 64 light vertices)
// tiles = 256 candidate pairs per trip.  Phase 1: every lane runs the cheap culls of its pair (zero throughput,
// distance, both cosines, emission cone); culled pairs get their zero written at once, the survivors go onto an
// LDS list (ballot + mbcnt).  Phase 2: the expensive part -- two BSDF values, the shadow ray, the MIS weight --
// runs over that list 256 pairs at a time: the list is kept ACROSS trips and evaluated only when it holds a full
// workgroup of pairs (and once at the end), so every wave of phase 2 has all its lanes busy whatever fraction of
// the candidates survives.  The table is indexed by (vertex, light vertex), so the order of evaluation is free.
// Frames, local directions, Lambda terms and diffuse lobes come from the per-vertex contexts.
template <bool COUNT>
__global__ __launch_bounds__(kBlock)
void k_bdpt_connect(BdptSceneDev sc, PathBuf pb, BdptPathBuf bp, const LightVertexDev *lvs, const LightVertexCtx *lctx, int n_lv,
                    int light_depth, const uint32_t *cqueue, const uint32_t *ccount, uint32_t slots, WorkCounters *wc){
    BdTally tally; tally.nodes = tally.tris = tally.spheres = tally.group_boxes = 0u;
    uint32_t n_pairs = 0u, n_survivors = 0u, n_shadow = 0u, n_lit = 0u;
    extern __shared__ uint32_t s_dyn_stack[];          // [stack level][lane], sized by the scene's deepest group tree
    __shared__ uint32_t s_pair_path[2 * kBlock];
    __shared__ uint32_t s_pair_j[2 * kBlock];
    __shared__ uint32_t s_n;
    uint32_t *stk = s_dyn_stack + threadIdx.x;
    uint32_t count = *ccount;
    uint32_t chunks = ((uint32_t) n_lv + 63u) / 64u;
    unsigned long long items = (unsigned long long) count * chunks;
    uint32_t wave_in_block = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    unsigned long long groups = (items + 3ull) / 4ull;
    // phase 2 for one surviving pair
    auto evaluate = [&](uint32_t path, int j){
        float4 vp = bp.vtx_pos[path], vn = bp.vtx_nrm[path], vt = bp.vtx_thr[path], vw = bp.vtx_wo[path], vb = bp.vtx_base[path];
        const LightVertexDev lv = lvs[j];
        const LightVertexCtx *lcp = lctx + j;
        f3 v_pos = xyz(vp), v_n = xyz(vn), v_thr = xyz(vt);
        Mat vm; vm.base = xyz(vb); vm.roughness = vp.w; vm.metallic = vn.w; vm.eta = vt.w;
        int depth = (int) f2u(vw.w);
        f3 lthr = ld3(lv.thr);
        f3 d_vec = ld3(lv.pos) - v_pos;
        float dist2 = dot3(d_vec, d_vec);
        float dist = sqrtf(dist2);
        f3 wi = d_vec / dist;
        float cosE = fmaxf(0.0f, dot3(v_n, wi));
        float cosL = fmaxf(0.0f, dot3(ld3(lv.normal), wi * -1.0f));
        int t_idx = j % light_depth;
        f3 contrib = mk3(0, 0, 0);
        f3 fE; float pdf_unused;
        { ShadeCtx ce; ShadePre pe;
          load_eye_value_ctx(bp, path, slots, v_n, ce, pe);
          bsdf_eval_pdf<true, false>(vm, ce, wi, fE, pdf_unused, &pe); }
        f3 fL = mk3(1.0f, 1.0f, 1.0f);
        if(!(lv.flags & 1u) && t_idx > 0){
            ShadeCtx cl; cl.T = ld3(lcp->T); cl.B = ld3(lcp->B); cl.N = ld3(lv.normal); cl.wo = ld3(lcp->wo_l);
            ShadePre pl; pl.lam_o = lcp->lam_l; pl.diffuse = ld3(lcp->diffuse);
            bsdf_eval_pdf<true, false>(lv_mat(lv), cl, wi * -1.0f, fL, pdf_unused, &pl);
        }
        bool ok = !((fE.x <= 0.0f && fE.y <= 0.0f && fE.z <= 0.0f) || (fL.x <= 0.0f && fL.y <= 0.0f && fL.z <= 0.0f));
        if(COUNT){ n_survivors += 1u; if(ok) n_shadow += 1u; }
        if(ok && bd_visible<COUNT>(sc, v_pos + v_n * kEps, ld3(lv.pos) + ld3(lv.normal) * kEps, stk, &tally)){
            if(COUNT) n_lit += 1u;
            float G = (cosE * cosL) / fmaxf(dist2, 1e-4f);
            const size_t first = (size_t) (j / light_depth) * light_depth;
            float mis_w = bd_mis_weight(bp, path, slots, depth, vm, lvs + first, lcp, t_idx, d_vec, dist2);
            f3 c = v_thr * fE * G * fL * lthr * mk3(1.0f, 1.0f, 1.0f) * mis_w;
            if(is_valid_color(c)) contrib = clamp_radiance(c, 15.0f);
        }
        bp.contrib[(size_t) path * n_lv + j] = make_float4(contrib.x, contrib.y, contrib.z, 0.0f);
    };
    if(threadIdx.x == 0) s_n = 0u;
    __syncthreads();
    for(unsigned long long gidx = blockIdx.x; gidx < groups; gidx += gridDim.x){
        // ---- phase 1: culls ----
        unsigned long long w = gidx * 4ull + wave_in_block;
        bool survive = false;
        uint32_t path = 0u; int j = 0;
        if(w < items){
            path = cqueue[(uint32_t) (w / chunks)];
            j = (int) ((uint32_t) (w % chunks) * 64u + lane);
            if(COUNT && j < n_lv) n_pairs += 1u;
            if(j < n_lv){
                float4 vp = bp.vtx_pos[path], vn = bp.vtx_nrm[path];
                const LightVertexDev *lv = lvs + j;
                f3 lthr = ld3(lv->thr);
                bool ok = !(length3(lthr) < 1e-6f);
                f3 d_vec = ld3(lv->pos) - xyz(vp);
                float dist2 = dot3(d_vec, d_vec);
                ok = ok && !(dist2 < 1e-6f);
                float dist = sqrtf(dist2);
                f3 wi = d_vec / dist;
                float cosE = fmaxf(0.0f, dot3(xyz(vn), wi));
                float cosL = fmaxf(0.0f, dot3(ld3(lv->normal), wi * -1.0f));
                ok = ok && !(cosE <= 0.0f || cosL <= 0.0f);
                uint32_t lflags = lv->flags;
                if(ok && (lflags & 1u) && lv->source_cutoff > 0.0f && !(lflags & 2u)){
                    int real_light = (j / light_depth) % sc.num_lights;
                    const DevLight &L = sc.lights[real_light];
                    f3 light_dir = normalize3(ld3(L.raw_dir));
                    if(dot3(light_dir, wi * -1.0f) < L.cos_cutoff) ok = false;
                }
                survive = ok;
            }
            // one 64-bit word per (vertex, chunk) says which pairs get a table entry; a culled pair gets none (its
            // contribution is zero and k_bdpt_reduce never reads it)
            unsigned long long vm = __ballot(survive);
            if(lane == 0u) bp.valid[(size_t) path * chunks + (uint32_t) (w % chunks)] = vm;
        }
        // the list holds fewer than kBlock pairs here and gains at most kBlock
        uint32_t pos = lds_push(survive, &s_n);
        if(survive){ s_pair_path[pos] = path; s_pair_j[pos] = (uint32_t) j; }
        __syncthreads();
        // ---- phase 2: a full workgroup of survivors, taken from the top of the list ----
        uint32_t n = s_n;
        // every wave has its snapshot of the count before any wave can push again: without this barrier a wave that
        // runs ahead into the next trip's lds_push could change s_n under a slower wave's read, and the waves of one
        // workgroup would then disagree about taking the branch below (which holds a barrier)
        __syncthreads();
        if(n >= (uint32_t) kBlock){
            uint32_t e = n - (uint32_t) kBlock + threadIdx.x;
            evaluate(s_pair_path[e], (int) s_pair_j[e]);
            if(threadIdx.x == 0) s_n = n - (uint32_t) kBlock;       // nobody reads s_n or pushes before the barrier below
            __syncthreads();                                        // the list entries above n - kBlock have been read
        }
    }
    uint32_t n = s_n;
    if(threadIdx.x < n) evaluate(s_pair_path[threadIdx.x], (int) s_pair_j[threadIdx.x]);
    if(COUNT){
        unsigned long long v[8] = { n_pairs, n_survivors, n_shadow, n_lit, tally.nodes, tally.tris, tally.spheres, tally.group_boxes };
        for(int k = 0; k < 8; ++k) for(int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
        if((threadIdx.x & 63u) == 0u){
            unsigned long long *dst[8] = { &wc->bd_pairs, &wc->bd_survivors, &wc->bd_shadow_rays, &wc->bd_unoccluded, &wc->bd_nodes, &wc->bd_tris,
                                           &wc->bd_spheres, &wc->bd_group_boxes };
            for(int k = 0; k < 8; ++k) if(v[k]) atomicAdd(dst[k], v[k]);
        }
    }
}

// total_L of one eye vertex: the table row summed in light-vertex order (the CPU loop's order, so the
// float sums match bit for bit), then added to the sample.  One lane owns one vertex and adds serially;
// the rows are fetched through LDS in tiles of 64 vertices x 8 entries so that every global load reads
// whole 128-B lines (8 lanes x 16 B per row) instead of one 16-B piece per row.
constexpr int kRedEntries = 8;
__global__ __launch_bounds__(kBlock)
void k_bdpt_reduce(PathBuf pb, BdptPathBuf bp, int n_lv, const uint32_t *cqueue, const uint32_t *ccount){
    __shared__ float4 s_tile[kBlock / 64][64][kRedEntries + 1];          // +1: rows 144 B apart, conflict-free ds_read_b128
    uint32_t count = *ccount;
    uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t waves_total = gridDim.x * (kBlock / 64);
    for(uint32_t base = (blockIdx.x * (kBlock / 64) + wave) * 64u; base < count; base += waves_total * 64u){
        uint32_t mine = base + lane;
        uint32_t my_path = mine < count ? cqueue[mine] : 0u;
        f3 total = mk3(0, 0, 0);
        const uint32_t chunks = ((uint32_t) n_lv + 63u) / 64u;
        unsigned long long my_valid = 0ull;
        for(int e0 = 0; e0 < n_lv; e0 += kRedEntries){
            // which of this vertex's pairs have a table entry (the others were culled: exact zeros, adding them changes nothing)
            if((e0 & 63) == 0) my_valid = mine < count ? bp.valid[(size_t) my_path * chunks + (uint32_t) (e0 >> 6)] : 0ull;
            const uint32_t mb = (uint32_t) (my_valid >> (e0 & 63)) & 0xFFu;
            if(__ballot(mb != 0u) == 0ull) continue;                    // none of the wave's 64 vertices has an entry in this group
            // 8 loads: lanes (v = it * 8 + lane / 8, e = lane % 8)
#pragma unroll
            for(int it = 0; it < 8; ++it){
                uint32_t v = (uint32_t) it * 8u + (lane >> 3);
                int e = e0 + (int) (lane & 7u);
                uint32_t vp = (uint32_t) __shfl((int) my_path, (int) v, 64);
                uint32_t vb = (uint32_t) __shfl((int) mb, (int) v, 64);
                float4 val = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if((vb >> (lane & 7u)) & 1u) val = bp.contrib[(size_t) vp * n_lv + e];
                s_tile[wave][v][lane & 7u] = val;
            }
            __builtin_amdgcn_wave_barrier();
            int lim = n_lv - e0 < kRedEntries ? n_lv - e0 : kRedEntries;
            for(int e = 0; e < lim; ++e){ float4 c = s_tile[wave][lane][e]; total = total + xyz(c); }
            __builtin_amdgcn_wave_barrier();
        }
        if(mine < count){
            float4 col = pb.col[my_path];
            col.x = col.x + total.x; col.y = col.y + total.y; col.z = col.z + total.z;
            pb.col[my_path] = col;
        }
    }
}

uint32_t grid_for(uint32_t items){
    uint32_t g = (items + kBlock - 1) / kBlock;
    if(g < 1u) g = 1u;
    return g > 4096u ? 4096u : g;
}

} // namespace

void launch_bdpt_light_trace(hipStream_t s, const BdptSceneDev &sc, LightVertexDev *lv, int total_paths, int light_depth,
                             int spl, uint64_t seed, int max_delta){
    if(total_paths <= 0) return;
    hipLaunchKernelGGL(k_bdpt_light_trace, dim3((total_paths + kBlock - 1) / kBlock), dim3(kBlock), 0, s, sc, lv, total_paths,
                       light_depth, spl, seed, max_delta);
}
void launch_bdpt_generate(hipStream_t s, const Tiling &tl, const CameraDev &cam, PathBuf pb, BdptPathBuf bp, uint32_t *qcount,
                          int samples_this_pass, uint32_t first_sample, uint64_t seed){
    uint32_t total = (uint32_t) tl.n_local * (uint32_t) samples_this_pass;
    hipLaunchKernelGGL(k_bdpt_generate, dim3(grid_for(total)), dim3(kBlock), 0, s, tl, cam, pb, bp, qcount, total, first_sample, seed);
}
// max_groups != 0 caps the grid (the blind tail iterations of HPT_FLAG_NO_HOST_WAIT): extend, connect and reduce walk
// their queues with a stride, so a small grid is only slower when the queue is long -- which it is not there
static uint32_t capped(uint32_t g, uint32_t max_groups){ return max_groups != 0u && g > max_groups ? max_groups : g; }

void launch_bdpt_extend(hipStream_t s, const BdptSceneDev &sc, PathBuf pb, const uint32_t *queue, const uint32_t *qcount,
                        uint32_t max_items, uint32_t max_groups){
    hipLaunchKernelGGL(k_bdpt_extend, dim3(capped(grid_for(max_items), max_groups)), dim3(kBlock), 0, s, sc, pb, queue, qcount);
}
void launch_bdpt_vertex(hipStream_t s, const BdptSceneDev &sc, PathBuf pb, BdptPathBuf bp, const uint32_t *queue,
                        const uint32_t *qcount, uint32_t max_items, uint32_t *next_queue, uint32_t *next_count,
                        uint32_t *cqueue, uint32_t *ccount, int eye_depth, int max_delta, uint32_t slots, const float eye[3]){
    uint32_t g = (max_items + kVtxChunk - 1) / kVtxChunk;
    if(g < (uint32_t) kVtxTargetGroups) g = (uint32_t) kVtxTargetGroups;
    uint32_t small = (max_items + kBlock - 1) / kBlock;
    if(small < g) g = small < 1u ? 1u : small;
    hipLaunchKernelGGL(k_bdpt_vertex, dim3(g), dim3(kBlock), 0, s, sc, pb, bp, queue, qcount, next_queue, next_count, cqueue,
                       ccount, eye_depth, max_delta, slots, eye[0], eye[1], eye[2]);
}
void launch_bdpt_light_ctx(hipStream_t s, const LightVertexDev *lv, LightVertexCtx *ctx, int n_lv, int light_depth){
    if(n_lv <= 0) return;
    hipLaunchKernelGGL(k_bdpt_light_ctx, dim3((n_lv + kBlock - 1) / kBlock), dim3(kBlock), 0, s, lv, ctx, n_lv, light_depth);
}
void launch_bdpt_connect(hipStream_t s, const BdptSceneDev &sc, PathBuf pb, BdptPathBuf bp, const LightVertexDev *lv,
                         const LightVertexCtx *lctx, int n_lv, int light_depth, const uint32_t *cqueue, const uint32_t *ccount,
                         uint32_t max_items, uint32_t slots, uint32_t max_groups, WorkCounters *wc){
    unsigned long long waves = (unsigned long long) max_items * (((unsigned) n_lv + 63u) / 64u);
    unsigned long long g = (waves + (kBlock / 64) - 1) / (kBlock / 64);
    if(g < 1ull) g = 1ull;
    if(g > 4096ull) g = 4096ull;            // several trips per workgroup: the survivor list fills up across them
    const size_t stack_bytes = (size_t) (sc.stack_levels > 0 ? sc.stack_levels : kStackDepth) * kBlock * sizeof(uint32_t);
    if(wc) hipLaunchKernelGGL((k_bdpt_connect<true>), dim3(capped((uint32_t) g, max_groups)), dim3(kBlock), stack_bytes, s, sc, pb, bp, lv, lctx, n_lv, light_depth,
                              cqueue, ccount, slots, wc);
    else hipLaunchKernelGGL((k_bdpt_connect<false>), dim3(capped((uint32_t) g, max_groups)), dim3(kBlock), stack_bytes, s, sc, pb, bp, lv, lctx, n_lv, light_depth,
                            cqueue, ccount, slots, wc);
}
void launch_bdpt_reduce(hipStream_t s, PathBuf pb, BdptPathBuf bp, int n_lv, const uint32_t *cqueue, const uint32_t *ccount,
                        uint32_t max_items, uint32_t max_groups){
    hipLaunchKernelGGL(k_bdpt_reduce, dim3(capped(grid_for(max_items), max_groups)), dim3(kBlock), 0, s, pb, bp, n_lv, cqueue, ccount);
}

} // namespace hpt
