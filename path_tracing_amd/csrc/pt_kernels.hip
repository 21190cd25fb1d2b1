// HIP kernels of the unidirectional path-tracing hot path, written for gfx950 (MI355X):
// a wavefront pipeline over queues of path slots
//
//   generate -> [ trace -> shade ]* -> trace -> resolve -> finalize
//
// generate  primary rays for (pixel, sample) slots                 (reference src/pt_cu.cu:36-46)
// trace     one merged launch per iteration: closest hit of every queued path (spheres + light balls
//           by scan, triangles by BVH; what it must return: reference include/geometric.cuh:327-388)
//           and the any-hit shadow rays of the previous iteration, whose unoccluded contributions are
//           added to the sample's radiance (include/geometric.cuh:293-325, src/pt_cu.cu:136-146,
//           174-196).  A trace step is two launches: every ray gets a budget of node steps, the few
//           that need more are set aside and finished by the resume launch.
// shade     light-hit emission, next-event estimation set-up, BSDF sampling, throughput update;
//           survivors and shadow requests are compacted in LDS (wave64 ballot + mbcnt prefix) and
//           flushed with one global atomic per workgroup            (reference src/pt_cu.cu:54-241)
// resolve   per pixel, adds this pass's samples in sample order (src/pt_cu.cu:243-245)
// finalize  mean over samples into the packed local framebuffer      (src/pt_cu.cu:248)
// k_extend / k_connect are the one-ray-per-lane forms kept for the brute-force scan variants (tests).
//
// One lane owns one path for the whole launch, every (pixel, sample) owns one slot for the
// whole pass, and sums run in a fixed order, so results do not depend on scheduling, queue
// order, pass size or the number of devices.  Built with -ffp-contract=off (pt_device_math.h).
#include "pt_kernels.h"
#include "pt_device_math.h"

namespace hpt {

namespace {

HPT_DEV uint32_t f2u(float f){ return __float_as_uint(f); }
HPT_DEV float u2f(uint32_t u){ return __uint_as_float(u); }
HPT_DEV f3 xyz(float4 v){ return mk3(v.x, v.y, v.z); }

// (m & a) | (~m & b) in one instruction, and the two 16-bit halves of a word as floats (SDWA conversions): the compiler
// expands the first into three and masks before converting the low half
HPT_DEV uint32_t bit_select(uint32_t m, uint32_t a, uint32_t b){ uint32_t r; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b)); return r; }
HPT_DEV float lo16f(uint32_t w){ float r; asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(r) : "v"(w)); return r; }
HPT_DEV float hi16f(uint32_t w){ float r; asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(w)); return r; }

// ---- wave64 queue push: ballot, mbcnt prefix, one atomic per wave on a workgroup-local LDS counter ----
HPT_DEV uint32_t lds_push(bool want, uint32_t *lds_counter){
    unsigned long long mask = __ballot(want);
    if(mask == 0ull) return 0u;
    uint32_t lo = (uint32_t) mask, hi = (uint32_t) (mask >> 32);
    uint32_t prefix = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
    uint32_t base = 0u;
    int leader = __ffsll((long long) mask) - 1;
    if((int) (threadIdx.x & 63u) == leader) base = atomicAdd(lds_counter, (uint32_t) __popcll(mask));
    base = (uint32_t) __shfl((int) base, leader, 64);
    return base + prefix;
}

// ---- tiling -----------------------------------------------------------------------------
// local slot p -> global pixel; false when the slot lies outside the image
HPT_DEV bool tile_to_pixel(const Tiling &tl, uint32_t p, int &x, int &y){
    uint32_t ts2 = (uint32_t) (tl.tile * tl.tile);
    uint32_t lt = p / ts2, q = p % ts2;
    uint32_t gt = lt * (uint32_t) tl.world + (uint32_t) tl.rank;
    if(gt >= (uint32_t) tl.ntiles) return false;
    // tile gt sits in row gt / tiles_x at column (gt % tiles_x + row) % tiles_x: every row is rotated by its
    // index, so the tiles of one rank run down the image diagonally instead of in columns (an object that
    // fills a few tile columns would otherwise load only the ranks owning them)
    uint32_t ty = gt / (uint32_t) tl.tiles_x, tx = (gt % (uint32_t) tl.tiles_x + ty) % (uint32_t) tl.tiles_x;
    uint32_t sub = q >> 6, l = q & 63u;
    uint32_t spr = (uint32_t) tl.tile >> 3;
    uint32_t bx = sub % spr, by = sub / spr;
    x = (int) (tx * (uint32_t) tl.tile + bx * 8u + (l & 7u));
    y = (int) (ty * (uint32_t) tl.tile + by * 8u + (l >> 3));
    return x < tl.W && y < tl.H;
}

// ---- traversal --------------------------------------------------------------------------
struct Tally { uint32_t boxes, tris, steps, wave_steps; };   // steps: traversal loop trips of this lane;
                                                              // wave_steps: per ray, the wave's longest lane

// Walks the triangle BVH.  ANY: returns true at the first opaque blocker in (1e-3, tmax).
// Closest: refines (best_t, best_slot, best_ord); exact ties go to the lower scan ordinal,
// which is what the reference's in-order scan with a strict '<' keeps.
template <bool ANY, bool COUNT>
HPT_DEV bool walk_bvh(const SceneDev &sc, f3 ro, f3 rd, float tmax, uint32_t *stk,
                      float &best_t, uint32_t &best_slot, uint32_t &best_ord, Tally &tally){
    float dx = fabsf(rd.x) > 1e-20f ? rd.x : copysignf(1e-20f, rd.x);
    float dy = fabsf(rd.y) > 1e-20f ? rd.y : copysignf(1e-20f, rd.y);
    float dz = fabsf(rd.z) > 1e-20f ? rd.z : copysignf(1e-20f, rd.z);
    float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    float ox = ro.x * ix, oy = ro.y * iy, oz = ro.z * iz;
    float limit = ANY ? tmax : best_t;
    uint32_t cur = 0u;
    int sp = 0;
    for(;;){
        bool descend = false;
        if(COUNT) tally.steps += 1;
        if(!(cur & kLeafFlag)){
            const float4 *n = sc.nodes + (size_t) cur * 4;
            float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
            if(COUNT) tally.boxes += 2;
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
                if(COUNT) tally.tris += 1;
                float t;
                if(hit_triangle(ro, rd, xyz(t0), xyz(t1), xyz(t2), ANY ? tmax : 1e20f, t)){
                    if(ANY){
                        if(t > 1e-3f && (f2u(t2.w) & 1u)) return true;
                    } else {
                        uint32_t ord = f2u(t0.w);
                        if(t < best_t || (t == best_t && ord < best_ord)){
                            best_t = t; best_slot = first + k; best_ord = ord; limit = t;
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

// Reference-order scan over every triangle (tests: BVH == scan).
template <bool ANY, bool COUNT>
HPT_DEV bool scan_tris(const SceneDev &sc, f3 ro, f3 rd, float tmax,
                       float &best_t, uint32_t &best_slot, uint32_t &best_ord, Tally &tally){
    for(int s = 0; s < sc.num_tris; ++s){
        const float4 *tp = sc.tris + (size_t) s * 3;
        float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
        if(COUNT) tally.tris += 1;
        float t;
        if(hit_triangle(ro, rd, xyz(t0), xyz(t1), xyz(t2), ANY ? tmax : 1e20f, t)){
            if(ANY){
                if(t > 1e-3f && (f2u(t2.w) & 1u)) return true;
            } else {
                uint32_t ord = f2u(t0.w);
                if(t < best_t || (t == best_t && ord < best_ord)){ best_t = t; best_slot = (uint32_t) s; best_ord = ord; }
            }
        }
    }
    return false;
}

// closest hit over spheres, light balls (scan order) and triangles; returns t and primitive code
template <bool BRUTE, bool COUNT>
HPT_DEV void closest_hit(const SceneDev &sc, f3 ro, f3 rd, uint32_t *stk, float &t_out, uint32_t &prim_out, Tally &tally){
    float best_t = 1e20f;
    uint32_t best_prim = kHitMiss, best_ord = 0xFFFFFFFFu;
    for(int i = 0; i < sc.num_rounds; ++i){
        DevRound r = sc.rounds[i];
        float t;
        if(hit_sphere(ro, rd, mk3(r.c[0], r.c[1], r.c[2]), r.r, 1e20f, t) && t < best_t){
            best_t = t; best_prim = kHitRoundFlag | (uint32_t) i; best_ord = (uint32_t) i;
        }
    }
    uint32_t slot = 0xFFFFFFFFu;
    float bt = best_t;
    if(BRUTE) scan_tris<false, COUNT>(sc, ro, rd, 1e20f, bt, slot, best_ord, tally);
    else walk_bvh<false, COUNT>(sc, ro, rd, 1e20f, stk, bt, slot, best_ord, tally);
    if(slot != 0xFFFFFFFFu){ best_t = bt; best_prim = slot; }
    t_out = best_t; prim_out = best_prim;
}

// shadow segment origin p1, unit direction, range (1e-3, max_d): true when unoccluded
template <bool BRUTE, bool COUNT>
HPT_DEV bool segment_visible(const SceneDev &sc, f3 p1, f3 dir, float max_d, uint32_t *stk, Tally &tally){
    for(int i = 0; i < sc.num_spheres; ++i){
        DevRound r = sc.rounds[i];
        float t;
        if(hit_sphere(p1, dir, mk3(r.c[0], r.c[1], r.c[2]), r.r, max_d, t) && t > 1e-3f && (r.flags & 1u)) return false;
    }
    float bt = max_d; uint32_t slot = 0, ord = 0;
    bool blocked = BRUTE ? scan_tris<true, COUNT>(sc, p1, dir, max_d, bt, slot, ord, tally)
                         : walk_bvh<true, COUNT>(sc, p1, dir, max_d, stk, bt, slot, ord, tally);
    return !blocked;
}

HPT_DEV uint32_t wave_max_u32(uint32_t v){
    for(int off = 32; off > 0; off >>= 1){ uint32_t o = (uint32_t) __shfl_xor((int) v, off, 64); v = o > v ? o : v; }
    return v;
}

HPT_DEV void flush_tally(const Tally &tally, uint32_t rays, WorkCounters *wc, bool shadow){
    // wave reduction, then one atomic per wave and counter
    unsigned long long b = tally.boxes, t = tally.tris, r = rays, st = tally.steps;
    for(int off = 32; off > 0; off >>= 1){
        b += __shfl_down(b, off, 64); t += __shfl_down(t, off, 64); r += __shfl_down(r, off, 64);
        st += __shfl_down(st, off, 64);
    }
    if((threadIdx.x & 63u) == 0u){
        if(st) atomicAdd(shadow ? &wc->lane_steps_shadow : &wc->lane_steps_closest, st);
        if(tally.wave_steps) atomicAdd(shadow ? &wc->wave_steps_shadow : &wc->wave_steps_closest, 64ull * tally.wave_steps);
        if(b) atomicAdd(shadow ? &wc->boxes_shadow : &wc->boxes_closest, b);
        if(t) atomicAdd(shadow ? &wc->tris_shadow : &wc->tris_closest, t);
        if(r) atomicAdd(shadow ? &wc->shadow_rays : &wc->closest_rays, r);
    }
}

// ---- kernels ----------------------------------------------------------------------------

// The primary ray of path slot i (reference src/pt_cu.cu:36-46): sample j = i / n_local of local pixel i % n_local,
// jittered with the first two uniforms of the path's stream.  False for a slot outside the image.  k_generate stores
// the result; the PRIMARY variants of k_trace / k_shade recompute it instead, which removes the generate launch and the
// 72 B per path it writes (and iteration 0 reads back) from every pass.
HPT_DEV bool primary_ray(const PrimaryGen &g, uint32_t i, f3 &eye, f3 &dir, uint64_t &rs){
    uint32_t p = i % (uint32_t) g.tl.n_local, j = i / (uint32_t) g.tl.n_local;
    int px, py;
    if(!tile_to_pixel(g.tl, p, px, py)) return false;
    rs = rng_seed(g.seed, (uint32_t) (py * g.tl.W + px), g.first_sample + j);
    float pixel_x = (float) px + rng_next(rs);
    float pixel_y = (float) py + rng_next(rs);
    eye = mk3(g.cam.eye[0], g.cam.eye[1], g.cam.eye[2]);
    f3 pixel_pos = mk3(g.cam.UL[0], g.cam.UL[1], g.cam.UL[2]) + mk3(g.cam.dx[0], g.cam.dx[1], g.cam.dx[2]) * pixel_x
                   + mk3(g.cam.dy[0], g.cam.dy[1], g.cam.dy[2]) * pixel_y;
    dir = normalize3(pixel_pos - eye);
    return true;
}

__global__ __launch_bounds__(kBlock)
void k_generate(Tiling tl, CameraDev cam, PathBuf pb, uint32_t *qcount,
                uint32_t total, uint32_t first_sample, uint64_t seed, WorkCounters *wc){
    // Slot i is path i: the first iteration's queue is the identity, so no compaction (and no
    // atomics) here.  Slots outside the image are marked dead (flag bit 1) and never traced.
    if(blockIdx.x == 0 && threadIdx.x == 0) *qcount = total;
    uint32_t stride = gridDim.x * kBlock;
    for(uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += stride){
        PrimaryGen g; g.tl = tl; g.cam = cam; g.first_sample = first_sample; g.pad = 0u; g.seed = seed;
        f3 eye = mk3(0, 0, 0), dir = mk3(0, 0, 1); uint64_t rs = 0ull;
        bool active = primary_ray(g, i, eye, dir, rs);
        if(active){
            pb.org_eta[i] = make_float4(eye.x, eye.y, eye.z, 1.0f);
            pb.dir_flags[i] = make_float4(dir.x, dir.y, dir.z, u2f(1u));      // last_is_delta = true, depth 0
            pb.thr[i] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
            pb.rng[i] = make_uint2((uint32_t) rs, (uint32_t) (rs >> 32));
        } else {
            pb.dir_flags[i] = make_float4(0.0f, 0.0f, 1.0f, u2f(2u));         // dead slot
            pb.hit[i] = make_uint2(f2u(1e20f), kHitMiss);
        }
        pb.col[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if(wc){
            unsigned long long m = __ballot(active);
            if((threadIdx.x & 63u) == 0u && m) atomicAdd(&wc->samples, (unsigned long long) __popcll(m));
        }
    }
}

template <bool BRUTE, bool COUNT>
__global__ __launch_bounds__(kBlock)
void k_extend(SceneDev sc, PathBuf pb, const uint32_t *queue, const uint32_t *qcount, WorkCounters *wc){
    __shared__ uint32_t s_stack[kStackDepth * kBlock];
    uint32_t count = *qcount;
    uint32_t *stk = s_stack + threadIdx.x;
    Tally tally; tally.boxes = 0; tally.tris = 0; tally.steps = 0; tally.wave_steps = 0;
    uint32_t rays = 0;
    for(uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock){
        uint32_t path = queue ? queue[i] : i;              // null queue: identity (first iteration)
        float4 o = pb.org_eta[path], d = pb.dir_flags[path];
        if(f2u(d.w) & 2u) continue;                        // slot outside the image
        float t; uint32_t prim;
        uint32_t s0 = tally.steps;
        closest_hit<BRUTE, COUNT>(sc, xyz(o), xyz(d), stk, t, prim, tally);
        pb.hit[path] = make_uint2(f2u(t), prim);
        ++rays;
        if(COUNT) tally.wave_steps += wave_max_u32(tally.steps - s0);     // lanes that skipped the trip are not here
    }
    if(COUNT) flush_tally(tally, rays, wc, false);
}

template <bool BRUTE, bool COUNT>
__global__ __launch_bounds__(kBlock)
void k_connect(SceneDev sc, PathBuf pb, ShadowBuf sb, const uint32_t *squeue, const uint32_t *scount, WorkCounters *wc){
    __shared__ uint32_t s_stack[kStackDepth * kBlock];
    uint32_t count = *scount;
    uint32_t *stk = s_stack + threadIdx.x;
    Tally tally; tally.boxes = 0; tally.tris = 0; tally.steps = 0; tally.wave_steps = 0;
    uint32_t rays = 0;
    for(uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock){
        uint32_t path = squeue[i];
        float4 a = sb.org_max[path], b = sb.dir[path];
        uint32_t s0 = tally.steps;
        bool vis = segment_visible<BRUTE, COUNT>(sc, xyz(a), xyz(b), a.w, stk, tally);
        ++rays;
        if(COUNT) tally.wave_steps += wave_max_u32(tally.steps - s0);
        if(vis){
            float4 c = sb.contrib[path];
            float4 col = pb.col[path];
            col.x = col.x + c.x; col.y = col.y + c.y; col.z = col.z + c.z;
            pb.col[path] = col;
        }
    }
    if(COUNT) flush_tally(tally, rays, wc, true);
}

#ifndef HPT_LDS_MATS
#define HPT_LDS_MATS 128
#endif
#ifndef HPT_LDS_LIGHTS
#define HPT_LDS_LIGHTS 32
#endif
#ifndef HPT_SHADE_CHUNK
#define HPT_SHADE_CHUNK 1024
#endif
#ifdef HPT_SHADE_WAVES
#define HPT_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(HPT_SHADE_WAVES, 8)))
#else
#define HPT_SHADE_ATTR
#endif
constexpr int kLdsMats = HPT_LDS_MATS;    // material records staged in LDS (6 KiB)
constexpr int kLdsLights = HPT_LDS_LIGHTS;   // light records staged in LDS (3.5 KiB)
constexpr int kShadeChunk = HPT_SHADE_CHUNK;        // paths per workgroup at most (LDS staging capacity)
constexpr int kShadeTargetGroups = 1024; // workgroups a short queue is spread over
// Next-event candidates of a wave are staged in LDS and evaluated 64 at a time (see k_shade): words per record
constexpr int kNeeWords = 20;
// 0-2 wo (local) | 3-5 wi (local) | 6 Lambda(wo) | 7 material index | 8-10 throughput | 11 pdf_light_dir |
// 12-14 shadow segment start | 15-17 shadow segment end | 18 path slot | 19 light index | parallel << 31

// development (-DHPT_SHADE_PROFILE=1|2|5, counting renders only; 3 and 4: probes inside the BSDF functions, pt_device_math.h): how many wave trips enter a section of k_shade and with how many
// lanes, through the eight bd_* counters (four sections per build): scripts/shade_sections.py
#ifdef HPT_SHADE_PROFILE
#define HPT_SECTION(set, idx) do { if(wc && HPT_SHADE_PROFILE == (set)){ unsigned long long m_ = __ballot(true); \
    if((int) (threadIdx.x & 63u) == __ffsll((long long) m_) - 1){ atomicAdd(&wc->bd_pairs + 2 * (idx), 1ull); atomicAdd(&wc->bd_pairs + 2 * (idx) + 1, (unsigned long long) __popcll(m_)); } } } while(0)
#else
#define HPT_SECTION(set, idx) do {} while(0)
#endif

template <bool PRIMARY, bool STRIDED>
__global__ __launch_bounds__(kBlock) HPT_SHADE_ATTR
void k_shade(SceneDev sc, PathBuf pb, const uint32_t *queue, const uint32_t *qcount,
             uint32_t *next_queue, uint32_t *next_count, ShadowBuf sb, uint32_t *squeue, uint32_t *scount,
             int max_depth, int max_delta, int roulette, WorkCounters *wc, PrimaryGen pg){
    __shared__ DevMaterial s_mats[kLdsMats];
    __shared__ DevLight s_lights[kLdsLights];
    // survivors and shadow requests of this workgroup's chunk are compacted in LDS (wave64
    // ballot + mbcnt prefix, one LDS atomic per wave) and flushed with ONE global atomic per
    // queue: same-address global atomics saturate near 88 per microsecond on this chip.
    __shared__ uint32_t s_next[kShadeChunk];
    __shared__ uint32_t s_shadow[kShadeChunk];
    __shared__ uint32_t s_cnt[4];          // [0] survivors, [1] shadow requests, [2],[3] global bases
    // Next-event estimation is evaluated DENSELY.  Only about a third of the lanes of a trip get past the cosine and
    // cone tests, and the BSDF value + pdf + MIS + shadow-record code behind them is ~30 % of this kernel's time
    // (knock-out: 16.3 -> 11.2 ms per pass).  So a lane that passes leaves a 20-word record in its wave's LDS staging
    // area instead of evaluating; once 64 records are waiting (about three trips) the wave evaluates them with every
    // lane busy.  Same expressions on the same operands, so the contributions are bit-identical; only the order of the
    // shadow queue changes, which nothing depends on.  Wave-private: no barrier, the count lives in a register.
    __shared__ uint32_t s_stage[kBlock / 64][kNeeWords][64];
    uint32_t (*stage)[64] = s_stage[threadIdx.x >> 6];
    uint32_t staged = 0u;                  // records waiting in this wave's staging area (wave-uniform)
    const uint32_t lane = threadIdx.x & 63u;
    // evaluates the first n staged records, one per lane (pt_cu.cu:136-146 parallel lights, :179-196 ball lights)
    auto flush_nee = [&](uint32_t n){
        bool want = false; uint32_t spath = 0u;
        if(lane < n){
            HPT_SECTION(2, 3);                                         // section h: staged next-event evaluation
            f3 wo_l = mk3(u2f(stage[0][lane]), u2f(stage[1][lane]), u2f(stage[2][lane]));
            f3 wi_l = mk3(u2f(stage[3][lane]), u2f(stage[4][lane]), u2f(stage[5][lane]));
            ShadePre pre; pre.lam_o = u2f(stage[6][lane]);
            const uint32_t mat_i = stage[7][lane], light_i = stage[19][lane];
            f3 thr = mk3(u2f(stage[8][lane]), u2f(stage[9][lane]), u2f(stage[10][lane]));
            const float pdf_light_dir = u2f(stage[11][lane]);
            f3 p1 = mk3(u2f(stage[12][lane]), u2f(stage[13][lane]), u2f(stage[14][lane]));
            f3 p2 = mk3(u2f(stage[15][lane]), u2f(stage[16][lane]), u2f(stage[17][lane]));
            spath = stage[18][lane];
            const DevMaterial dm = (sc.num_mats <= kLdsMats ? s_mats : sc.mats)[mat_i];
            const DevLight &L = (sc.num_lights <= kLdsLights ? s_lights : sc.lights)[light_i & 0x7FFFFFFFu];
            Mat m; m.base = mk3(dm.base[0], dm.base[1], dm.base[2]); m.roughness = dm.roughness; m.metallic = dm.metallic; m.eta = dm.eta;
            pre.diffuse = mk3(dm.diffuse[0], dm.diffuse[1], dm.diffuse[2]);
            f3 illum = mk3(L.illum[0], L.illum[1], L.illum[2]);
            f3 brdf; float pdf_bsdf;
            bsdf_eval_pdf_local(m, wo_l, wi_l, brdf, pdf_bsdf, &pre);
            const float cos_surface = fmaxf(0.0f, wi_l.z);           // = max(0, dot(normal, wi)): the frame's third axis is the normal
            f3 contrib;
            if(light_i >> 31) contrib = thr * brdf * illum * mk3(1.0f, 1.0f, 1.0f) * cos_surface * (float) sc.num_lights;
            else {
                float p_l = pdf_light_dir * pdf_light_dir;
                float p_b = pdf_bsdf * pdf_bsdf;
                float mis_w = p_l / fmaxf(p_l + p_b, 1e-8f);
                contrib = thr * brdf * illum * mk3(1.0f, 1.0f, 1.0f) * cos_surface / pdf_light_dir * mis_w;
            }
            if(is_valid_color(contrib)){
                HPT_SECTION(5, 3);                                     // next-event contribution kept: shadow record written
                want = true;
                f3 c = clamp_radiance(contrib, 15.0f);
                f3 diff = p2 - p1;                              // geometric.cuh:298-303
                float dist = length3(diff);
                f3 dir = diff / dist;
                sb.org_max[spath] = make_float4(p1.x, p1.y, p1.z, dist - 1e-3f);
                sb.dir[spath] = make_float4(dir.x, dir.y, dir.z, 0.0f);
                sb.contrib[spath] = make_float4(c.x, c.y, c.z, 0.0f);
            }
        }
        uint32_t spos = lds_push(want, &s_cnt[1]);
        if(want) s_shadow[spos] = spath;
    };
#ifdef HPT_SHADE_PROFILE
    g_hpt_probe = wc ? &wc->bd_pairs : nullptr;            // every thread stores the same value before it reads it
#endif
    if(threadIdx.x < 4) s_cnt[threadIdx.x] = 0u;
    const bool mats_in_lds = sc.num_mats <= kLdsMats;
    const bool lights_in_lds = sc.num_lights <= kLdsLights;
    if(mats_in_lds){
        const uint32_t *src = (const uint32_t *) sc.mats; uint32_t *dst = (uint32_t *) s_mats;
        for(int w = threadIdx.x; w < sc.num_mats * (int) (sizeof(DevMaterial) / 4); w += kBlock) dst[w] = src[w];
    }
    if(lights_in_lds){
        const uint32_t *src = (const uint32_t *) sc.lights; uint32_t *dst = (uint32_t *) s_lights;
        for(int w = threadIdx.x; w < sc.num_lights * (int) (sizeof(DevLight) / 4); w += kBlock) dst[w] = src[w];
    }
    __syncthreads();
    const DevMaterial *mats = mats_in_lds ? s_mats : sc.mats;
    const DevLight *lights = lights_in_lds ? s_lights : sc.lights;

    uint32_t count = *qcount;
    // contiguous chunk per workgroup: small enough to spread a short queue over the chip,
    // never larger than the LDS staging buffers
    uint32_t chunk = (count + kShadeTargetGroups - 1) / kShadeTargetGroups;
    chunk = (chunk + kBlock - 1) / kBlock * kBlock;
    chunk = chunk < (uint32_t) kBlock ? (uint32_t) kBlock : (chunk > (uint32_t) kShadeChunk ? (uint32_t) kShadeChunk : chunk);
    uint32_t iters = 0;
    // one chunk per workgroup when the grid covers the queue (the usual launch); STRIDED: a smaller grid (the unseen
    // tail iterations of HPT_FLAG_NO_HOST_WAIT) walks the chunks with a stride
    for(uint32_t cb = blockIdx.x; (unsigned long long) cb * chunk < count; cb += gridDim.x){
    if(STRIDED && cb != blockIdx.x){
        __syncthreads();                                   // the previous chunk's lists have been copied out
        if(threadIdx.x < 4) s_cnt[threadIdx.x] = 0u;
        __syncthreads();
    }
    uint32_t begin = cb * chunk;
    uint32_t end = begin + chunk < count ? begin + chunk : count;
    for(uint32_t base = begin; base < end; base += kBlock){
        uint32_t i = base + threadIdx.x;
        bool alive = false, nee = false;
        uint32_t path = 0;
        // the next-event candidate of this lane's path, if it gets one (record layout: kNeeWords)
        f3 n_wo = mk3(0, 0, 0), n_wi = mk3(0, 0, 0), n_thr = mk3(0, 0, 0), s_p1 = mk3(0, 0, 0), s_p2 = mk3(0, 0, 0);
        float n_lam = 0.0f, n_pdf_light = 0.0f; uint32_t n_mat = 0u, n_light = 0u;
        if(i < count){
            path = queue ? queue[i] : i;
            ++iters;
            HPT_SECTION(1, 0);                                         // section a: a queue entry
            uint2 h = pb.hit[path];
            uint32_t prim = h.y;
            // PRIMARY (iteration 0 without a generate launch): the path state is not in memory yet -- the ray is recomputed
            // from the slot number, throughput 1, depth 0, "last bounce was delta" so a light seen directly counts
            // (pt_cu.cu:41-46); every slot's radiance record is initialised here (emission or zero)
            f3 p_eye = mk3(0, 0, 0), p_dir = mk3(0, 0, 1), first_col = mk3(0, 0, 0); uint64_t p_rs = 0ull;
            if(PRIMARY && !primary_ray(pg, path, p_eye, p_dir, p_rs)) prim = kHitMiss;
            if(prim != kHitMiss){                                     // miss ends the path (pt_cu.cu:54)
                float t = u2f(h.x);
                float4 o4, d4, th4;
                if(PRIMARY){
                    o4 = make_float4(p_eye.x, p_eye.y, p_eye.z, 1.0f); d4 = make_float4(p_dir.x, p_dir.y, p_dir.z, u2f(1u));
                    th4 = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
                } else { o4 = pb.org_eta[path]; d4 = pb.dir_flags[path]; th4 = pb.thr[path]; }
                f3 ro = xyz(o4), rd = xyz(d4), throughput = xyz(th4);
                float ray_eta = o4.w;
                uint32_t flags = f2u(d4.w);
                bool last_is_delta = (flags & 1u) != 0u;
                int depth = (int) ((flags >> 8) & 0xFFu);
                int delta_count = (int) ((flags >> 16) & 0xFFu);
                f3 pos = ro + rd * t;
                f3 wo = rd * -1.0f;
                f3 normal; uint32_t mat_idx = 0; bool is_light = false; uint32_t light_idx = 0;
                bool have_frame = false; f3 frame_T = mk3(0, 0, 0), frame_B = mk3(0, 0, 0);
                if(prim & kHitRoundFlag){
                    DevRound r = sc.rounds[prim & 0x7FFFFFFFu];
                    normal = normalize3(pos - mk3(r.c[0], r.c[1], r.c[2]));
                    is_light = (r.flags & 2u) != 0u;
                    mat_idx = r.material; light_idx = r.material;
                    if(dot3(normal, rd) > 0.0f) normal = normal * -1.0f;
                } else {
                    // the triangle's unit normal and the local frames of both orientations were computed once
                    // per scene by k_tri_frames with these same expressions
                    const float4 *fp = sc.tri_frames + (size_t) prim * 4;
                    float4 q0 = fp[0], q1 = fp[1], q2 = fp[2], q3 = fp[3];
                    normal = mk3(q0.x, q0.y, q0.z);
                    mat_idx = f2u(q3.w);
                    have_frame = true;
                    if(dot3(normal, rd) > 0.0f){
                        normal = normal * -1.0f;
                        frame_T = mk3(q2.y, q2.z, q2.w); frame_B = mk3(q3.x, q3.y, q3.z);
                    } else {
                        frame_T = mk3(q0.w, q1.x, q1.y); frame_B = mk3(q1.z, q1.w, q2.x);
                    }
                }

                if(is_light){                                          // pt_cu.cu:59-122
                    HPT_SECTION(1, 1);                                 // section b: light hit
                    const DevLight &hl = lights[light_idx];
                    f3 emission = mk3(hl.illum[0], hl.illum[1], hl.illum[2]);
                    float area = 1.0f, cone_ratio = 1.0f;
                    bool valid_light = false;
                    for(int li = 0; li < sc.num_lights; ++li){
                        const DevLight &L = lights[li];
                        f3 c2h = pos - mk3(L.pos[0], L.pos[1], L.pos[2]);
                        if(fabsf(length3(c2h) - L.r) < 1e-2f){
                            valid_light = true;
                            area = L.area;
                            if(L.cutoff > 0.0f && !L.is_parallel){
                                cone_ratio = L.cone_ratio;
                                if(depth == 0) cone_ratio = 1.f;
                                else if(dot3(mk3(L.main_dir[0], L.main_dir[1], L.main_dir[2]), normalize3(c2h)) < L.cos_cutoff) cone_ratio = 0.0f;
                            }
                            break;
                        }
                    }
                    if(valid_light && cone_ratio > 0.0f) emission = emission / (area * cone_ratio);
                    else emission = mk3(0, 0, 0);
                    if((emission.x > 0.0f || emission.y > 0.0f || emission.z > 0.0f) && last_is_delta){
                        f3 contrib = throughput * emission;
                        if(is_valid_color(contrib)){
                            f3 c = clamp_radiance(contrib, 15.0f);
                            if(PRIMARY) first_col = mk3(0.0f + c.x, 0.0f + c.y, 0.0f + c.z);       // the sum starts at zero (pt_cu.cu:39)
                            else {
                                float4 col = pb.col[path];
                                col.x = col.x + c.x; col.y = col.y + c.y; col.z = col.z + c.z;
                                pb.col[path] = col;
                            }
                        }
                    }
                    // BSDF-sampled light hits after a non-delta bounce add nothing (the reference's
                    // MIS branch is a stub with pdf_light_dir = 0, pt_cu.cu:103-118); path ends here.
                } else {
                    HPT_SECTION(1, 2);                                 // section c: surface hit
                    DevMaterial dm = mats[mat_idx];
                    Mat m; m.base = mk3(dm.base[0], dm.base[1], dm.base[2]);
                    m.roughness = dm.roughness; m.metallic = dm.metallic; m.eta = dm.eta;
                    ShadePre pre; pre.diffuse = mk3(dm.diffuse[0], dm.diffuse[1], dm.diffuse[2]);     // uploaded with the scene
                    // a delta lobe does not count as a bounce; any other material's path ends at max_depth, so on
                    // its last bounce the sampled direction would never be used (pt_cu.cu:37, 228-241)
                    const bool delta_mat = (m.eta > 0.0f && m.roughness < 0.001f && m.metallic < 0.01f) || (m.metallic > 0.99f && m.roughness < 0.001f);
                    const bool last_bounce = !delta_mat && depth + 1 >= max_depth;
                    uint64_t rs;
                    if(PRIMARY) rs = p_rs;
                    else { uint2 r2 = pb.rng[path]; rs = ((uint64_t) r2.y << 32) | (uint64_t) r2.x; }
                    ShadeCtx ctx;
                    if(have_frame){ ctx.N = normal; ctx.T = frame_T; ctx.B = frame_B; ctx.wo = to_local(wo, frame_T, frame_B, normal); }
                    else ctx = make_shade_ctx(normal, wo);
                    pre.lam_o = ggx_lambda(ctx.wo, roughness_to_alpha(m.roughness));     // shared by the NEE and the sampled query
                    // the hit-level words of a next-event record are set here, outside the nested branches behind the rejection
                    // loop.  (With them inside both branches hipcc 7.2's SLP vectorizer dropped the y component of the packed
                    // x/y pairs -- green = 0 in every contribution, caught by tests/test_gpu_parity.py on input.txt; the build
                    // now passes -fno-slp-vectorize, see the Makefile.)
                    n_wo = ctx.wo; n_lam = pre.lam_o; n_thr = throughput; n_mat = mat_idx;
                    s_p1 = pos + normal * kEps;

                    // next-event estimation, pt_cu.cu:125-202
                    if(m.eta <= 0.0f && (m.metallic < 0.99f || m.roughness > 0.01f) && sc.num_lights > 0){
                        int l_idx = min((int) (rng_next(rs) * sc.num_lights), sc.num_lights - 1);
                        const DevLight &L = lights[l_idx];
                        if(L.is_parallel){
                            f3 light_dir = mk3(L.neg_dir[0], L.neg_dir[1], L.neg_dir[2]);
                            float cos_surface = fmaxf(0.0f, dot3(normal, light_dir));
                            if(cos_surface > 0.0f){
                                nee = true;
                                n_wi = to_local(light_dir, ctx.T, ctx.B, ctx.N);
                                n_light = (uint32_t) l_idx | 0x80000000u;
                                s_p2 = pos + light_dir * 1e4f;
                            }
                        } else {
                            f3 d_local;
                            do {
                                HPT_SECTION(1, 3);                     // section d: one try of the unit-ball rejection loop
                                float a = rng_next(rs), b = rng_next(rs), c = rng_next(rs);
                                d_local = mk3(a, b, c) * 2.0f - mk3(1.0f, 1.0f, 1.0f);
#ifdef HPT_KNOCK_BALL_TAIL       // development: upper bound of what any restructuring of the loop's tail could save (wrong image)
                            } while(false);
#else
                            } while(dot3(d_local, d_local) >= 1.0f);
#endif
                            HPT_SECTION(2, 0);                         // section e: next-event geometry behind the loop
                            if(length3(d_local) > 0.001f) d_local = normalize3(d_local);
                            else d_local = mk3(0, 1, 0);
                            f3 light_pos = mk3(L.pos[0], L.pos[1], L.pos[2]) + d_local * L.r;
                            f3 wi_light = light_pos - pos;
                            float dist2 = dot3(wi_light, wi_light);
                            float dist = sqrtf(dist2);
                            wi_light = wi_light / dist;
                            float cos_surface = fmaxf(0.0f, dot3(normal, wi_light));
                            float cos_light = fmaxf(0.0f, dot3(d_local, wi_light * -1.0f));
                            if(cos_surface > 0.0f && cos_light > 0.0f){
                                bool inside_cone = true;
                                if(L.cutoff > 0.0f){
                                    if(dot3(mk3(L.main_dir[0], L.main_dir[1], L.main_dir[2]), wi_light * -1.0f) < L.cos_cutoff) inside_cone = false;
                                }
                                if(inside_cone){
                                    float pdf_light_area = 1.0f / (sc.num_lights * L.area);
                                    nee = true;
                                    n_pdf_light = pdf_light_area * dist2 / fmaxf(cos_light, 1e-6f);
                                    n_wi = to_local(wi_light, ctx.T, ctx.B, ctx.N);
                                    n_light = (uint32_t) l_idx;
                                    s_p2 = light_pos + d_local * kEps;
                                }
                            }
                        }
                    }

                    // BSDF sampling and path continuation, pt_cu.cu:204-241
                    f3 wi = mk3(0, 0, 0), bsdf_val = mk3(0, 0, 0); float pdf_omega = 0.0f, new_eta = ray_eta; bool is_delta = false;
                    if(!last_bounce){
                        HPT_SECTION(2, 1);                             // section f: direction sampling
                        float u_rr = rng_next(rs), u1 = rng_next(rs), u2 = rng_next(rs);
                        bsdf_sample(m, ctx, u_rr, u1, u2, ray_eta, wi, bsdf_val, pdf_omega, is_delta, new_eta, &pre);
                    }
                    if(!(pdf_omega <= 0.0f)){          // pt_cu.cu:214 (and the TIR return, defined: terminate)
                        HPT_SECTION(2, 2);                             // section g: throughput update and continuation
                        f3 new_o;
                        if(is_delta){
                            HPT_SECTION(5, 0);                         // delta continuation
                            throughput = throughput * bsdf_val;
                            ray_eta = new_eta;
                            if(dot3(wi, normal) < 0.0f) new_o = pos - normal * kEps;
                            else new_o = pos + normal * kEps;
                            last_is_delta = true;
                            ++delta_count;
                            alive = is_valid_color(throughput) && delta_count <= max_delta;
                        } else {
                            HPT_SECTION(5, 1);                         // non-delta continuation
                            float cos_wi = fabsf(dot3(normal, wi));
                            throughput = throughput * bsdf_val * cos_wi / pdf_omega;
                            new_o = pos + normal * kEps;
                            last_is_delta = false;
                            ++depth;
                            alive = is_valid_color(throughput);
                            if(alive && roulette){
                                float q = fminf(1.0f, fmaxf(0.05f, fmaxf(throughput.x, fmaxf(throughput.y, throughput.z))));
                                float u = rng_next(rs);
                                alive = u < q;
                                throughput = throughput / q;
                            }
                            alive = alive && depth < max_depth;
                        }
                        if(alive){
                            HPT_SECTION(5, 2);                         // the path goes on: state written back
                            uint32_t nf = (last_is_delta ? 1u : 0u) | ((uint32_t) depth << 8) | ((uint32_t) delta_count << 16);
                            pb.org_eta[path] = make_float4(new_o.x, new_o.y, new_o.z, ray_eta);
                            pb.dir_flags[path] = make_float4(wi.x, wi.y, wi.z, u2f(nf));
                            pb.thr[path] = make_float4(throughput.x, throughput.y, throughput.z, 0.0f);
                            pb.rng[path] = make_uint2((uint32_t) rs, (uint32_t) (rs >> 32));
                        }
                    }
                }
            }
            if(PRIMARY) pb.col[path] = make_float4(first_col.x, first_col.y, first_col.z, 0.0f);
        }
        // stage this trip's next-event candidates; evaluate when the wave's staging area would overflow
        {
            const unsigned long long cm = __ballot(nee);
            if(cm != 0ull){
                const uint32_t k = (uint32_t) __popcll(cm);
                if(staged + k > 64u){ flush_nee(staged); staged = 0u; }
                if(nee){
                    const uint32_t slot = staged + __builtin_amdgcn_mbcnt_hi((uint32_t) (cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) cm, 0u));
                    stage[0][slot] = f2u(n_wo.x); stage[1][slot] = f2u(n_wo.y); stage[2][slot] = f2u(n_wo.z);
                    stage[3][slot] = f2u(n_wi.x); stage[4][slot] = f2u(n_wi.y); stage[5][slot] = f2u(n_wi.z);
                    stage[6][slot] = f2u(n_lam); stage[7][slot] = n_mat;
                    stage[8][slot] = f2u(n_thr.x); stage[9][slot] = f2u(n_thr.y); stage[10][slot] = f2u(n_thr.z);
                    stage[11][slot] = f2u(n_pdf_light);
                    stage[12][slot] = f2u(s_p1.x); stage[13][slot] = f2u(s_p1.y); stage[14][slot] = f2u(s_p1.z);
                    stage[15][slot] = f2u(s_p2.x); stage[16][slot] = f2u(s_p2.y); stage[17][slot] = f2u(s_p2.z);
                    stage[18][slot] = path; stage[19][slot] = n_light;
                }
                staged += k;
            }
        }
        uint32_t qpos = lds_push(alive, &s_cnt[0]);
        if(alive) s_next[qpos] = path;
    }
    if(staged != 0u){ flush_nee(staged); staged = 0u; }
    __syncthreads();
    if(threadIdx.x == 0){
        s_cnt[2] = s_cnt[0] ? atomicAdd(next_count, s_cnt[0]) : 0u;
        s_cnt[3] = s_cnt[1] ? atomicAdd(scount, s_cnt[1]) : 0u;
    }
    __syncthreads();
    for(uint32_t k = threadIdx.x; k < s_cnt[0]; k += kBlock) next_queue[s_cnt[2] + k] = s_next[k];
    for(uint32_t k = threadIdx.x; k < s_cnt[1]; k += kBlock) squeue[s_cnt[3] + k] = s_shadow[k];
    if(!STRIDED) break;
    }
    if(wc){
        unsigned long long v = iters;
        for(int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if((threadIdx.x & 63u) == 0u && v) atomicAdd(&wc->path_iters, v);
    }
}

// accum[p] += sum over this pass's samples, in sample order, after the per-sample guard
__global__ __launch_bounds__(kBlock)
void k_resolve(Tiling tl, PathBuf pb, float4 *accum, int samples){
    uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if(p >= (uint32_t) tl.n_local) return;
    float4 a = accum[p];
    f3 sum = mk3(a.x, a.y, a.z);
    for(int j = 0; j < samples; ++j){
        float4 c4 = pb.col[(size_t) j * tl.n_local + p];
        f3 c = xyz(c4);
        if(!is_valid_color(c)) c = mk3(0, 0, 0);            // pt_cu.cu:243
        sum = sum + c;                                      // pt_cu.cu:245
    }
    accum[p] = make_float4(sum.x, sum.y, sum.z, 0.0f);
}

// Per triangle, once per scene: the unit normal of pt_cu.cu's hit record (normalize(cross(e1, e2)),
// geometric.cuh:286) and the local shading frame of geometric.cuh:421-424 for both orientations of
// that normal (the shading normal is flipped against the ray).  Computed here, on the device, with the
// functions k_shade would otherwise run per hit, so the values are the same bit for bit.
//   q0 = N.xyz T.x | q1 = T.yz B.xy | q2 = B.z T'.xyz | q3 = B'.xyz material   (T', B': frame of -N)
__global__ __launch_bounds__(kBlock)
void k_tri_frames(const float4 *tris, int num_tris, float4 *frames){
    int i = blockIdx.x * kBlock + threadIdx.x;
    if(i >= num_tris) return;
    float4 t1 = tris[(size_t) i * 3 + 1], t2 = tris[(size_t) i * 3 + 2];
    f3 n = normalize3(cross3(xyz(t1), xyz(t2)));
    f3 T, B, T2, B2;
    build_local_frame(n, T, B);
    build_local_frame(n * -1.0f, T2, B2);
    frames[(size_t) i * 4 + 0] = make_float4(n.x, n.y, n.z, T.x);
    frames[(size_t) i * 4 + 1] = make_float4(T.y, T.z, B.x, B.y);
    frames[(size_t) i * 4 + 2] = make_float4(B.z, T2.x, T2.y, T2.z);
    frames[(size_t) i * 4 + 3] = make_float4(B2.x, B2.y, B2.z, t1.w);
}

// d_local[p] = accum[p] / spp (scale_is_div) -- pt_cu.cu:248
__global__ __launch_bounds__(kBlock)
void k_finalize(Tiling tl, const float4 *accum, float *d_local, float divisor){
    uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if(p >= (uint32_t) tl.n_local) return;
    float4 a = accum[p];
    f3 v = mk3(a.x, a.y, a.z);
    if(divisor != 1.0f) v = v / divisor;
    d_local[(size_t) p * 3 + 0] = v.x;
    d_local[(size_t) p * 3 + 1] = v.y;
    d_local[(size_t) p * 3 + 2] = v.z;
}

// row-major image <- [rank][local slot] packed buffers
__global__ __launch_bounds__(kBlock)
void k_untile(Tiling tl, const float *gathered, float *image){
    uint32_t idx = blockIdx.x * kBlock + threadIdx.x;
    if(idx >= (uint32_t) (tl.W * tl.H)) return;
    uint32_t x = idx % (uint32_t) tl.W, y = idx / (uint32_t) tl.W;
    uint32_t ts = (uint32_t) tl.tile;
    uint32_t tx = x / ts, ty = y / ts;
    uint32_t gt = ty * (uint32_t) tl.tiles_x + (tx + (uint32_t) tl.tiles_x - ty % (uint32_t) tl.tiles_x) % (uint32_t) tl.tiles_x;   // inverse of tile_to_pixel's row rotation
    uint32_t r = gt % (uint32_t) tl.world, lt = gt / (uint32_t) tl.world;
    uint32_t bx = (x % ts) >> 3, by = (y % ts) >> 3;
    uint32_t l = ((y & 7u) << 3) | (x & 7u);
    uint32_t q = (by * (ts >> 3) + bx) * 64u + l;
    size_t src = ((size_t) r * tl.n_local + (size_t) lt * ts * ts + q) * 3;
    size_t dst = (size_t) idx * 3;
    image[dst + 0] = gathered[src + 0];
    image[dst + 1] = gathered[src + 1];
    image[dst + 2] = gathered[src + 2];
}

// 8-bit output stage of the reference CLI (src/main_cli.cpp:227-242): per channel clamp to [0, 1], pow(x, 1/2.2),
// x 255, truncate.  byte(x) is a non-decreasing step function of x, so the device does not evaluate powf at all:
// thr[k] (k = 1..255) is the smallest float whose byte is >= k, found on the HOST with the host's own powf
// (hpt_api.cpp, tonemap_thresholds), and the byte is the number of thresholds <= x -- eight steps of a binary
// search in LDS.  The bytes are therefore exactly what the reference's host loop produces with the same libm;
// NaN compares false everywhere and lands on 0, which is where std::max(0.0f, std::min(NaN, 1.0f)) sends it.
// One thread packs four consecutive output bytes; bgr = 1 writes the reference's cv::Vec3b channel order.
__global__ __launch_bounds__(kBlock)
void k_tonemap(const float *linear, uint32_t *out_words, unsigned long long num_values, int bgr, const float *thr_table){
    __shared__ float s_thr[256];
    s_thr[threadIdx.x] = thr_table[threadIdx.x];
    __syncthreads();
    unsigned long long w = (unsigned long long) blockIdx.x * kBlock + threadIdx.x;
    unsigned long long first = w * 4ull;
    if(first >= num_values) return;
    uint32_t packed = 0u;
    for(int k = 0; k < 4; ++k){
        unsigned long long j = first + (unsigned long long) k;
        if(j >= num_values) break;
        unsigned long long src = j;
        if(bgr){ unsigned long long px = j / 3ull; src = px * 3ull + (2ull - (j - px * 3ull)); }
        float x = linear[src];
        uint32_t lo = 0u;
        for(uint32_t step = 128u; step > 0u; step >>= 1) if(x >= s_thr[lo + step]) lo += step;   // s_thr[0] unused (byte >= 0 always)
        packed |= lo << (8 * k);
    }
    if(first + 4ull <= num_values) out_words[w] = packed;
    else {
        unsigned char *tail = (unsigned char *) out_words + first;
        for(unsigned long long k = 0; first + k < num_values; ++k) tail[k] = (unsigned char) (packed >> (8 * k));
    }
}

// ---- merged trace kernel: closest-hit and any-hit rays with dynamic lane refill -----------
// One launch serves the extension rays of iteration i+1 and the shadow rays of iteration i
// (they are independent).  A workgroup owns a contiguous chunk of one queue; its lanes pull
// rays from the chunk through an LDS counter, and a lane whose ray is finished pulls the next
// one as soon as enough lanes of its wave are idle.  Inside, the classic while-while shape:
// lanes walk inner nodes until each holds a leaf, then the (much longer) triangle code runs
// once for all of them.
//
// Measured alternatives on config 3 (1024^2, 32 spp, ms per render; this version 42.3):
//   one ray per lane, separate extend/connect launches, 32 KiB static stack ........ 69.5
//   single loop doing a node OR a leaf step per trip (both code paths every trip) ... 45.5
//   per-trip vote between refill / node step / leaf step ............................. 44.4
//   persistent waves pulling 256-ray batches from one global cursor .................. 55.9
//   persistent waves, 64-ray batches, 32 cursors on ONE 128-B line ................... 207.9
//   same with one cursor per 128-B line (82 VGPRs) ................................... 47.5
// Chunk size: 256 -> 47.3, 512 -> 45.5, 1024 -> 49.6, 2048 -> 59.3 (long chunks leave the tail of
// a launch to a few workgroups); refill threshold 8 -> 51.0, 16 -> 49.6, 32 -> 48.6, 48 -> 48.0.
// k_trace's registers are fitted to eight waves per SIMD -- vector AND scalar: left alone the compiler takes 106 SGPRs, and
// more than 96 cap a CU at six 256-thread workgroups whatever the vector registers allow (80 with this attribute, the
// rest spilled to lanes of a VGPR).  Measured against the unconstrained build: first launch 11.37 -> 11.00 ms, resume
// 8.14 -> 7.70 per 64-spp pass of config 3, 14.00 -> 13.46 on the 1 M-triangle shape.  `make variant EXTRA=-DHPT_TRACE_WAVES=0`
// builds without it, =7 with seven.
#ifndef HPT_TRACE_WAVES
#define HPT_TRACE_WAVES 8
#endif
#if HPT_TRACE_WAVES > 0
#define HPT_TRACE_ATTR __attribute__((amdgpu_waves_per_eu(HPT_TRACE_WAVES, 8)))
#else
#define HPT_TRACE_ATTR
#endif
constexpr int kTraceChunk = 2048;     // rays per workgroup of the first launch.  Round 1, unsplit step: 512 -> 37.6 ms, 768 -> 36.5, 1024 -> 36.5, 1280 -> 36.1,
                                      // 1536 -> 36.7, 2048 -> 38.4.  Round 3, split step with the four-wide resume, two passes in flight (whose
                                      // kernels fill each other's tails): 1024 -> 126.7 ms per 256-spp render of config 3, 1536 -> 125.8, 2048 -> 125.3,
                                      // 3072 -> 126.1, 4096 -> 125.9; with refill at 56 idle lanes 2048 -> 124.5 (1 M triangles, 4096^2 x 8 spp: 74.2 -> 73.1)
constexpr uint32_t kDoneCode = 0xFFFFFFFEu;   // leaf-flagged code a finished walk parks in cur
constexpr int kRefillMin = 56;        // idle lanes that trigger a refill (24 -> 34.4 ms per 64-spp pass, 32 -> 34.1, 40 -> 33.8, 48 -> 33.8, 56 -> 33.7)
constexpr int kNodeMin = 4;           // fewer lanes than this still walking nodes (while others hold leaves): do the leaves first
                                      // (A/B: off 42.1 ms, 2: 39.9, 4: 39.4, 8: 40.2, 16: 41.3, 32: 43.2)
constexpr uint32_t kTraceShortQueue = 1u << 20;   // below this many rays a workgroup takes 256 instead of kTraceChunk

// LDSK > 0: only the first LDSK stack levels of a lane live in LDS, deeper ones in its column of `deep` (global memory,
// one column per lane of the grid, `deep_stride` words apart): the resume launch walks the long rays with the whole tree
// depth as its stack bound (22 KB of LDS per workgroup at depth 21, 27 KB at 26: five or six workgroups per CU), while a
// ray rarely holds more than a dozen pending subtrees
template <bool ANY, bool COUNT, bool RESUME, bool TOP, bool PRIMARY, int LDSK, bool WIDE>
HPT_DEV void trace_chunk(const SceneDev &sc, PathBuf pb, ShadowBuf sb, const uint32_t *queue,
                         uint32_t end, uint32_t *stk, uint32_t *s_next, int refill_min, int node_min, WorkCounters *wc,
                         uint32_t budget, uint32_t *s_long, uint32_t *s_nlong, const uint4 *top, const PrimaryGen &pg,
                         uint32_t *deep, uint32_t deep_stride){
    bool active = false, exhausted = false;
    uint32_t path = 0u, cur = 0u, steps = 0u;
    int sp = 0;
    f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 1);
    float ix = 0, iy = 0, iz = 0, ox = 0, oy = 0, oz = 0;
    uint32_t mx = 0u, my = 0u, mz = 0u;          // all ones where the ray runs against the axis (its near plane is a box's upper one)
    float tmax = 0.0f, limit = 0.0f, best_t = 1e20f;
    uint32_t best_prim = kHitMiss, best_ord = 0xFFFFFFFFu;
    unsigned long long n_lane_steps = 0, n_wave_steps = 0, n_boxes = 0, n_tris = 0, n_rays = 0, n_leaf_lane = 0, n_leaf_wave = 0;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t step_limit = budget != 0u ? budget : 0xFFFFFFFFu;
    for(;;){
        if(!RESUME && budget != 0u){
            // a ray that has used its node-step budget is set aside for the second launch (it restarts
            // there with what it found so far as its limit), so this lane goes back to the short rays
            bool defer = active && steps >= budget;
            unsigned long long dm = __ballot(defer);
            if(dm != 0ull){
                uint32_t dn = (uint32_t) __popcll(dm);
                uint32_t dprefix = __builtin_amdgcn_mbcnt_hi((uint32_t) (dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) dm, 0u));
                int dleader = __ffsll((long long) dm) - 1;
                uint32_t dbase = 0u;
                if((int) lane == dleader) dbase = atomicAdd(s_nlong, dn);
                dbase = (uint32_t) __shfl((int) dbase, dleader, 64);
                if(defer){
                    s_long[dbase + dprefix] = path;
                    if(!ANY) pb.hit[path] = make_uint2(f2u(best_t), best_prim);
                    active = false;
                }
            }
        }
        unsigned long long idle = __ballot(!active);
        if(idle != 0ull && !exhausted && (idle == ~0ull || __popcll(idle) >= refill_min)){
            // ---- refill: idle lanes pull the next rays of this workgroup's chunk ----
            uint32_t n = (uint32_t) __popcll(idle);
            uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t) (idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) idle, 0u));
            int leader = __ffsll((long long) idle) - 1;
            uint32_t base = 0u;
            if((int) lane == leader) base = atomicAdd(s_next, n);
            base = (uint32_t) __shfl((int) base, leader, 64);
            exhausted = base + n >= end;
            if(!active){
                uint32_t i = base + prefix;
                if(i < end){
                    path = queue ? queue[i] : i;
                    bool start = true;
                    if(ANY){
                        float4 a = sb.org_max[path], b = sb.dir[path];
                        ro = xyz(a); rd = xyz(b); tmax = a.w;
                        if(COUNT && !RESUME) n_rays += 1;
                        // spheres first (the reference scans them after the triangles; the result is a boolean);
                        // a resumed ray has passed them already
                        for(int r = 0; r < (RESUME ? 0 : sc.num_spheres); ++r){
                            DevRound rr = sc.rounds[r];
                            float t;
                            if(hit_sphere(ro, rd, mk3(rr.c[0], rr.c[1], rr.c[2]), rr.r, tmax, t) && t > 1e-3f && (rr.flags & 1u)) start = false;
                        }
                        limit = tmax;
                    } else {
                        bool outside;
                        if(PRIMARY){
                            // iteration 0 without a generate launch: the camera ray of this slot, recomputed (the first
                            // launch reads the identity queue, the resume launch the slots it was handed)
                            uint64_t rs_unused;
                            outside = !primary_ray(pg, path, ro, rd, rs_unused);
                            if(outside && !RESUME) pb.hit[path] = make_uint2(f2u(1e20f), kHitMiss);
                        } else {
                            float4 o = pb.org_eta[path], d = pb.dir_flags[path];
                            ro = xyz(o); rd = xyz(d);
                            outside = (f2u(d.w) & 2u) != 0u;
                        }
                        if(outside) start = false;                          // slot outside the image
                        else if(RESUME){
                            // restart with the closest hit of the first launch as the limit
                            uint2 h = pb.hit[path];
                            best_t = u2f(h.x); best_prim = h.y; best_ord = 0xFFFFFFFFu;
                            if(best_prim != kHitMiss)
                                best_ord = (best_prim & kHitRoundFlag) ? (best_prim & ~kHitRoundFlag) : f2u(sc.tris[(size_t) best_prim * 3].w);
                            limit = best_t;
                        } else {
                            if(COUNT) n_rays += 1;
                            best_t = 1e20f; best_prim = kHitMiss; best_ord = 0xFFFFFFFFu;
                            for(int r = 0; r < sc.num_rounds; ++r){
                                DevRound rr = sc.rounds[r];
                                float t;
                                if(hit_sphere(ro, rd, mk3(rr.c[0], rr.c[1], rr.c[2]), rr.r, 1e20f, t) && t < best_t){
                                    best_t = t; best_prim = kHitRoundFlag | (uint32_t) r; best_ord = (uint32_t) r;
                                }
                            }
                            limit = best_t;
                        }
                    }
                    if(start){
                        float dx = fabsf(rd.x) > 1e-20f ? rd.x : copysignf(1e-20f, rd.x);
                        float dy = fabsf(rd.y) > 1e-20f ? rd.y : copysignf(1e-20f, rd.y);
                        float dz = fabsf(rd.z) > 1e-20f ? rd.z : copysignf(1e-20f, rd.z);
                        // box tests only have to be conservative: the 1-ulp hardware reciprocal is inside the
                        // 2e-6 slack of the slab test (the triangle tests below use exact arithmetic)
                        // (the counting render divides exactly: its node and triangle tallies are then a function of the ray
                        // and the tree alone, which the oracle's host walk of the exported tree reproduces to the last count)
                        if(COUNT){ ix = 1.0f / dx; iy = 1.0f / dy; iz = 1.0f / dz; }
                        else { ix = __builtin_amdgcn_rcpf(dx); iy = __builtin_amdgcn_rcpf(dy); iz = __builtin_amdgcn_rcpf(dz); }
                        // quantised boxes: plane = qorigin + q * qscale, so t = q * (qscale * inv) + (qorigin - o) * inv
                        ox = (sc.qorigin[0] - ro.x) * ix; oy = (sc.qorigin[1] - ro.y) * iy; oz = (sc.qorigin[2] - ro.z) * iz;
                        ix *= sc.qscale[0]; iy *= sc.qscale[1]; iz *= sc.qscale[2];
                        mx = ix < 0.0f ? 0xFFFFFFFFu : 0u; my = iy < 0.0f ? 0xFFFFFFFFu : 0u; mz = iz < 0.0f ? 0xFFFFFFFFu : 0u;
                        cur = 0u; sp = 0; steps = 0u;
                        active = true;
                    }
                }
            }
            if(!__any(active)){
                if(exhausted) break;
                continue;
            }
        } else if(idle == ~0ull){
            break;                                      // chunk exhausted and every ray finished
        }
        const unsigned long long idle_at_entry = __ballot(!active);
        // phase 1: walk inner nodes until this lane holds a leaf (or its ray is finished); when only a
        // few lanes are still descending while others wait with a leaf, go and do the leaves first
        while(active && !(cur & kLeafFlag) && (RESUME || steps < step_limit)){
            if(node_min > 0){
                unsigned long long walking = __ballot(true);
                if(__popcll(walking) < node_min && __popcll(walking) < 64 - (int) __popcll(idle_at_entry)) break;
            }
            if(COUNT){
                n_lane_steps += 1; n_boxes += 2;
                if((int) lane == __ffsll((long long) __ballot(true)) - 1) n_wave_steps += 64;   // one wave trip
            }
            if(!RESUME) steps += 1u;
            if(WIDE){
                // four-wide step (development A/B): four child boxes from one 64-B node; the nearest hit child is entered, the
                // other hit children are stacked in slot order (ordering them too buys 0.3 % fewer steps on the benchmark
                // meshes: scripts/micro/bvh4_steps.cpp).  t = q * i + o is monotone in q, so the near plane of an axis is the
                // lower one when i >= 0 and the upper one otherwise: one bit-select per word picks it for two children.
                const uint4 *n = sc.wnodes + (size_t) cur * 4;
                const uint4 qx = n[0], qy = n[1], qz = n[2], qc = n[3];
                const uint32_t nx01 = bit_select(mx, qx.z, qx.x), nx23 = bit_select(mx, qx.w, qx.y), fx01 = bit_select(mx, qx.x, qx.z), fx23 = bit_select(mx, qx.y, qx.w);
                const uint32_t ny01 = bit_select(my, qy.z, qy.x), ny23 = bit_select(my, qy.w, qy.y), fy01 = bit_select(my, qy.x, qy.z), fy23 = bit_select(my, qy.y, qy.w);
                const uint32_t nz01 = bit_select(mz, qz.z, qz.x), nz23 = bit_select(mz, qz.w, qz.y), fz01 = bit_select(mz, qz.x, qz.z), fz23 = bit_select(mz, qz.y, qz.w);
                const float tn0 = fmaxf(fmaxf(fmaf(lo16f(nx01), ix, ox), fmaf(lo16f(ny01), iy, oy)), fmaxf(fmaf(lo16f(nz01), iz, oz), 0.0f));
                const float tf0 = fminf(fminf(fmaf(lo16f(fx01), ix, ox), fmaf(lo16f(fy01), iy, oy)), fminf(fmaf(lo16f(fz01), iz, oz), limit));
                const float tn1 = fmaxf(fmaxf(fmaf(hi16f(nx01), ix, ox), fmaf(hi16f(ny01), iy, oy)), fmaxf(fmaf(hi16f(nz01), iz, oz), 0.0f));
                const float tf1 = fminf(fminf(fmaf(hi16f(fx01), ix, ox), fmaf(hi16f(fy01), iy, oy)), fminf(fmaf(hi16f(fz01), iz, oz), limit));
                const float tn2 = fmaxf(fmaxf(fmaf(lo16f(nx23), ix, ox), fmaf(lo16f(ny23), iy, oy)), fmaxf(fmaf(lo16f(nz23), iz, oz), 0.0f));
                const float tf2 = fminf(fminf(fmaf(lo16f(fx23), ix, ox), fmaf(lo16f(fy23), iy, oy)), fminf(fmaf(lo16f(fz23), iz, oz), limit));
                const float tn3 = fmaxf(fmaxf(fmaf(hi16f(nx23), ix, ox), fmaf(hi16f(ny23), iy, oy)), fmaxf(fmaf(hi16f(nz23), iz, oz), 0.0f));
                const float tf3 = fminf(fminf(fmaf(hi16f(fx23), ix, ox), fmaf(hi16f(fy23), iy, oy)), fminf(fmaf(hi16f(fz23), iz, oz), limit));
                // key = entry distance with the slot number in its two lowest bits (distances are >= 0: their bit patterns order
                // like the values), all ones for a child that is missed or absent
                const uint32_t k0 = (tn0 <= tf0 * 1.000002f && qc.x != kEmptyChild) ? (f2u(tn0) & ~3u) : 0xFFFFFFFFu;
                const uint32_t k1 = (tn1 <= tf1 * 1.000002f && qc.y != kEmptyChild) ? ((f2u(tn1) & ~3u) | 1u) : 0xFFFFFFFFu;
                const uint32_t k2 = (tn2 <= tf2 * 1.000002f && qc.z != kEmptyChild) ? ((f2u(tn2) & ~3u) | 2u) : 0xFFFFFFFFu;
                const uint32_t k3 = (tn3 <= tf3 * 1.000002f && qc.w != kEmptyChild) ? ((f2u(tn3) & ~3u) | 3u) : 0xFFFFFFFFu;
                const uint32_t kmin = min(min(k0, k1), min(k2, k3));
                const bool any = kmin != 0xFFFFFFFFu;
                const uint32_t slot = kmin & 3u;
                uint32_t near = qc.x;
                near = slot == 1u ? qc.y : near; near = slot == 2u ? qc.z : near; near = slot == 3u ? qc.w : near;
                // stacked: hit and not the nearest, i.e. kmin < key < all ones -- one add and one unsigned compare per child
                const uint32_t c1 = ~kmin, c2 = c1 - 1u;
                const int e0 = (k0 + c1 < c2) ? 1 : 0, e1 = (k1 + c1 < c2) ? 1 : 0, e2 = (k2 + c1 < c2) ? 1 : 0, e3 = (k3 + c1 < c2) ? 1 : 0;
                uint32_t top;
                int p = sp;
                if(sp + 3 < LDSK){
                    // every level this step can touch is in LDS; a child that is not stacked is written to the spare level
                    top = stk[(sp > 0 ? sp - 1 : 0) * kBlock];
                    stk[(e0 ? p : LDSK) * kBlock] = qc.x; p += e0;
                    stk[(e1 ? p : LDSK) * kBlock] = qc.y; p += e1;
                    stk[(e2 ? p : LDSK) * kBlock] = qc.z; p += e2;
                    stk[(e3 ? p : LDSK) * kBlock] = qc.w; p += e3;
                } else {
                    auto level = [&](int l) -> uint32_t * { return l < LDSK ? stk + l * kBlock : deep + (size_t) (l - LDSK) * deep_stride; };
                    top = *level(sp > 0 ? sp - 1 : 0);
                    if(e0){ *level(p) = qc.x; ++p; }
                    if(e1){ *level(p) = qc.y; ++p; }
                    if(e2){ *level(p) = qc.z; ++p; }
                    if(e3){ *level(p) = qc.w; ++p; }
                }
                cur = any ? near : (sp > 0 ? top : kDoneCode);
                sp = any ? p : (sp > 0 ? sp - 1 : 0);
                continue;
            }
            // TOP: a ray with a budget of kTopLevels node steps only ever reaches nodes of depth < kTopLevels, which the
            // breadth-first node order puts among the first kTopNodes: those sit in LDS (staged once per workgroup)
            const uint4 *n = TOP ? top + cur * 2u : sc.qnodes + (size_t) cur * 2;
            uint4 w0 = n[0], w1 = n[1];
            // per axis one word of lower planes (left | right << 16) and one of upper planes, then the two child codes.  t = q * i + o
            // is monotone in q: the near plane of an axis is the lower one when i >= 0 and the upper one otherwise, so one
            // bit-select per word picks the near (far) planes of both children -- the same values min / max of the pair give
            const uint32_t nx = bit_select(mx, w0.y, w0.x), fx = bit_select(mx, w0.x, w0.y);
            const uint32_t ny = bit_select(my, w0.w, w0.z), fy = bit_select(my, w0.z, w0.w);
            const uint32_t nz = bit_select(mz, w1.y, w1.x), fz = bit_select(mz, w1.x, w1.y);
            float ln = fmaxf(fmaxf(fmaf(lo16f(nx), ix, ox), fmaf(lo16f(ny), iy, oy)), fmaxf(fmaf(lo16f(nz), iz, oz), 0.0f));
            float lf = fminf(fminf(fmaf(lo16f(fx), ix, ox), fmaf(lo16f(fy), iy, oy)), fminf(fmaf(lo16f(fz), iz, oz), limit));
            float rn = fmaxf(fmaxf(fmaf(hi16f(nx), ix, ox), fmaf(hi16f(ny), iy, oy)), fmaxf(fmaf(hi16f(nz), iz, oz), 0.0f));
            float rf = fminf(fminf(fmaf(hi16f(fx), ix, ox), fmaf(hi16f(fy), iy, oy)), fminf(fmaf(hi16f(fz), iz, oz), limit));
            uint32_t lc = w1.z, rc = w1.w;
            bool hl = (ln <= lf * 1.000002f) && (lc != kEmptyChild);
            bool hr = (rn <= rf * 1.000002f) && (rc != kEmptyChild);
            // branch-free step (the exec-mask bookkeeping of nested ifs costs more scalar issue than
            // the selects): always store the far child at the stack top, always fetch the entry below
            // it, then pick by selects.  The stack has one spare level for the unconditional store.
            bool any = hl || hr, both = hl && hr;
            bool left_first = hl && (!hr || ln <= rn);
            uint32_t top;
            if(LDSK > 0 && sp >= LDSK){                  // rare: this lane's stack has grown past its LDS share
                top = sp > LDSK ? deep[(size_t) (sp - 1 - LDSK) * deep_stride] : stk[(LDSK - 1) * kBlock];
                deep[(size_t) (sp - LDSK) * deep_stride] = left_first ? rc : lc;
            } else {
                top = stk[(sp > 0 ? sp - 1 : 0) * kBlock];
                stk[sp * kBlock] = left_first ? rc : lc;
            }
            uint32_t near = left_first ? lc : rc;
            cur = any ? near : (sp > 0 ? top : kDoneCode);
            sp = any ? sp + (both ? 1 : 0) : (sp > 0 ? sp - 1 : 0);
        }
        // phase 2: the leaf (lanes that left phase 1 early still hold an inner node and skip it);
        // kDoneCode = the walk found its stack empty
        if(active && (cur & kLeafFlag)){
            bool blocked = false, finished = cur == kDoneCode;
            if(!finished){
                if(COUNT){
                    n_leaf_lane += 1;
                    if((int) lane == __ffsll((long long) __ballot(true)) - 1) n_leaf_wave += 64;
                }
                uint32_t first = (cur & 0x7FFFFFFFu) >> 3;
                uint32_t cnt = (cur & 7u) + 1u;
                for(uint32_t k = 0; k < cnt; ++k){
                    const float4 *tp = sc.tris + (size_t) (first + k) * 3;
                    float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                    if(COUNT) n_tris += 1;
                    float t;
                    if(hit_triangle(ro, rd, xyz(t0), xyz(t1), xyz(t2), ANY ? tmax : 1e20f, t)){
                        if(ANY){
                            if(t > 1e-3f && (f2u(t2.w) & 1u)) blocked = true;
                        } else {
                            uint32_t ord = f2u(t0.w);
                            if(t < best_t || (t == best_t && ord < best_ord)){
                                best_t = t; best_prim = first + k; best_ord = ord; limit = t;
                            }
                        }
                    }
                }
                if(ANY && blocked) finished = true;                   // occluded: no contribution
                else if(sp > 0){
                    --sp;
                    cur = (LDSK > 0 && sp >= LDSK) ? deep[(size_t) (sp - LDSK) * deep_stride] : stk[sp * kBlock];
                }
                else finished = true;
            }
            if(finished){
                active = false;
                if(!ANY) pb.hit[path] = make_uint2(f2u(best_t), best_prim);
                else if(!blocked){
                    float4 c = sb.contrib[path];
                    float4 col = pb.col[path];
                    col.x = col.x + c.x; col.y = col.y + c.y; col.z = col.z + c.z;
                    pb.col[path] = col;
                }
            }
        }
    }
    if(COUNT){
        for(int off = 32; off > 0; off >>= 1){
            n_boxes += __shfl_down(n_boxes, off, 64); n_tris += __shfl_down(n_tris, off, 64);
            n_rays += __shfl_down(n_rays, off, 64); n_lane_steps += __shfl_down(n_lane_steps, off, 64);
            n_wave_steps += __shfl_down(n_wave_steps, off, 64); n_leaf_lane += __shfl_down(n_leaf_lane, off, 64);
            n_leaf_wave += __shfl_down(n_leaf_wave, off, 64);
        }
        if(lane == 0u){
            if(n_boxes) atomicAdd(ANY ? &wc->boxes_shadow : &wc->boxes_closest, n_boxes);
            if(n_tris) atomicAdd(ANY ? &wc->tris_shadow : &wc->tris_closest, n_tris);
            if(n_rays) atomicAdd(ANY ? &wc->shadow_rays : &wc->closest_rays, n_rays);
            if(n_lane_steps) atomicAdd(ANY ? &wc->lane_steps_shadow : &wc->lane_steps_closest, n_lane_steps);
            if(n_wave_steps) atomicAdd(ANY ? &wc->wave_steps_shadow : &wc->wave_steps_closest, n_wave_steps);
            if(n_leaf_lane) atomicAdd(ANY ? &wc->leaf_lane_shadow : &wc->leaf_lane_closest, n_leaf_lane);
            if(n_leaf_wave) atomicAdd(ANY ? &wc->leaf_wave_shadow : &wc->leaf_wave_closest, n_leaf_wave);
        }
    }
}

// Two-launch split of one trace step (budget != 0): the first launch gives every ray `budget` node
// steps; rays that need more (the few that run deep into a dense mesh) are set aside -- their slots
// collected in LDS and appended to the long queue with one global atomic per workgroup -- and a second
// launch (RESUME) restarts them with the partial result as the limit.  Lanes of the first launch are
// therefore never parked on a long ray while the short rays around them wait for a refill; the result
// is the same (closest hit with the ordinal tie-break, or the occlusion boolean, of the same ray).
struct LongQueues { uint32_t *equeue, *ecount, *squeue, *scount; uint32_t budget; };

template <bool COUNT, bool RESUME, bool TOP, bool PRIMARY, int LDSK, bool WIDE>
__global__ __launch_bounds__(kBlock) HPT_TRACE_ATTR
void k_trace(SceneDev sc, PathBuf pb, ShadowBuf sb, const uint32_t *equeue, const uint32_t *ecount_ptr,
             const uint32_t *squeue, const uint32_t *scount_ptr, uint32_t chunk_rays, int refill_min, int node_min,
             int stack_words, LongQueues lq, WorkCounters *wc, PrimaryGen pg, uint32_t *deep){
    extern __shared__ uint32_t s_dyn_stack[];        // [stack level][lane], sized by the BVH depth; then the long-ray list
    __shared__ uint32_t s_next, s_nlong, s_gbase;
    __shared__ uint4 s_top[TOP ? kTopNodes * 2 : 1];  // the top of the quantised tree (first launch of a split step)
    uint32_t *s_long = s_dyn_stack + stack_words;
    if(TOP){
        const uint32_t words = (uint32_t) (sc.num_nodes < kTopNodes ? sc.num_nodes : kTopNodes) * 2u;
        for(uint32_t i = threadIdx.x; i < words; i += kBlock) s_top[i] = sc.qnodes[i];
        // (the first __syncthreads of the chunk loop below orders these stores before any traversal)
    }
    uint32_t ecount = ecount_ptr ? *ecount_ptr : 0u, scount = scount_ptr ? *scount_ptr : 0u;
    // short queues (the delta-bounce tail) get one ray per lane so they spread over every CU
    uint32_t ce = ecount >= kTraceShortQueue ? chunk_rays : (uint32_t) kBlock;
    uint32_t cs = scount >= kTraceShortQueue ? chunk_rays : (uint32_t) kBlock;
    uint32_t ne = (ecount + ce - 1) / ce, ns = (scount + cs - 1) / cs;
    for(uint32_t bid = blockIdx.x; bid < ne + ns; bid += gridDim.x){
        bool shadow = bid >= ne;
        uint32_t chunk = shadow ? bid - ne : bid;
        uint32_t csize = shadow ? cs : ce;
        uint32_t begin = chunk * csize;
        uint32_t total = shadow ? scount : ecount;
        uint32_t end = begin + csize < total ? begin + csize : total;
        if(threadIdx.x == 0){ s_next = begin; s_nlong = 0u; }
        __syncthreads();
        uint32_t *deep_lane = LDSK > 0 ? deep + (size_t) blockIdx.x * kBlock + threadIdx.x : nullptr;
        const uint32_t deep_stride = gridDim.x * kBlock;
        if(shadow) trace_chunk<true, COUNT, RESUME, TOP, false, LDSK, WIDE>(sc, pb, sb, squeue, end, s_dyn_stack + threadIdx.x, &s_next, refill_min, node_min, wc,
                                                                      lq.budget, s_long, &s_nlong, s_top, pg, deep_lane, deep_stride);
        else trace_chunk<false, COUNT, RESUME, TOP, PRIMARY, LDSK, WIDE>(sc, pb, sb, equeue, end, s_dyn_stack + threadIdx.x, &s_next, refill_min, node_min, wc,
                                                                   lq.budget, s_long, &s_nlong, s_top, pg, deep_lane, deep_stride);
        __syncthreads();
        if(!RESUME && lq.budget != 0u){
            uint32_t n = s_nlong;
            if(n != 0u){
                if(threadIdx.x == 0) s_gbase = atomicAdd(shadow ? lq.scount : lq.ecount, n);
                __syncthreads();
                uint32_t *dst = (shadow ? lq.squeue : lq.equeue) + s_gbase;
                for(uint32_t i = threadIdx.x; i < n; i += kBlock) dst[i] = s_long[i];
            }
        }
        __syncthreads();
    }
}

template <bool BRUTE>
__global__ __launch_bounds__(kBlock)
void k_probe_closest(SceneDev sc, const float *org, const float *dir, int n, float *t_out, int32_t *prim_out){
    __shared__ uint32_t s_stack[kStackDepth * kBlock];
    int i = blockIdx.x * kBlock + threadIdx.x;
    if(i >= n) return;
    Tally tally; tally.boxes = 0; tally.tris = 0; tally.steps = 0; tally.wave_steps = 0;
    float t; uint32_t prim;
    closest_hit<BRUTE, false>(sc, mk3(org[3 * i], org[3 * i + 1], org[3 * i + 2]), mk3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]),
                              s_stack + threadIdx.x, t, prim, tally);
    int32_t ord = -1;
    if(prim != kHitMiss){
        if(prim & kHitRoundFlag) ord = (int32_t) (prim & 0x7FFFFFFFu);
        else ord = (int32_t) f2u(sc.tris[(size_t) prim * 3].w);
    }
    t_out[i] = t; prim_out[i] = ord;
}

template <bool BRUTE>
__global__ __launch_bounds__(kBlock)
void k_probe_visibility(SceneDev sc, const float *p1, const float *p2, int n, int32_t *vis_out){
    __shared__ uint32_t s_stack[kStackDepth * kBlock];
    int i = blockIdx.x * kBlock + threadIdx.x;
    if(i >= n) return;
    Tally tally; tally.boxes = 0; tally.tris = 0; tally.steps = 0; tally.wave_steps = 0;
    f3 a = mk3(p1[3 * i], p1[3 * i + 1], p1[3 * i + 2]), b = mk3(p2[3 * i], p2[3 * i + 1], p2[3 * i + 2]);
    f3 diff = b - a;
    float dist = length3(diff);
    f3 d = diff / dist;
    vis_out[i] = segment_visible<BRUTE, false>(sc, a, d, dist - 1e-3f, s_stack + threadIdx.x, tally) ? 1 : 0;
}

// Function-level probe (tests): the device versions of the reference's BSDF / Fresnel / GGX / frame
// functions (reference include/geometric.cuh:119-235, 419-562) on caller-given inputs, called the way k_shade calls
// them -- one shared ShadeCtx per hit, the merged value+pdf query, the precomputed diffuse lobe and Lambda(wo).
// in:  24 floats per record   0-5 material (base rgb, roughness, metallic, eta) | 6-8 N | 9-11 wo | 12-14 wi |
//                             15-17 u_rr u1 u2 | 18 current eta | 19 cosI | 20 etaI | 21 etaT
// out: 40 floats per record   0-2 f(wo, wi) 3 pdf | 4-6 sampled wi 7-9 f 10 pdf 11 is_delta 12 new_eta |
//                             13 FrDielectric(cosI, etaI, etaT) 14-16 FrSchlick(cosI, base) | 17 D(wi_l) 18 Lambda(wi_l)
//                             19 G(wo_l, wi_l) | 20 sin 21 cos of 2 pi u2 | 22-24 visible normal | 25 is_valid_color(f)
//                             26-28 clamp_radiance(20 f, 15) | 29-31 T 32-34 B of the local frame | 35-37 wo in that frame
__global__ __launch_bounds__(kBlock)
void k_probe_functions(const float *in, int n, float *out){
    int i = blockIdx.x * kBlock + threadIdx.x;
    if(i >= n) return;
    const float *r = in + (size_t) i * 24;
    float *o = out + (size_t) i * 40;
    Mat m; m.base = mk3(r[0], r[1], r[2]); m.roughness = r[3]; m.metallic = r[4]; m.eta = r[5];
    f3 N = mk3(r[6], r[7], r[8]), wo_w = mk3(r[9], r[10], r[11]), wi_w = mk3(r[12], r[13], r[14]);
    ShadeCtx ctx = make_shade_ctx(N, wo_w);
    ShadePre pre;
    pre.diffuse = m.base / kPi * (1.0f - m.metallic);
    const float alpha = roughness_to_alpha(m.roughness);
    pre.lam_o = ggx_lambda(ctx.wo, alpha);
    f3 f; float pdf;
    bsdf_eval_pdf(m, ctx, wi_w, f, pdf, &pre);
    o[0] = f.x; o[1] = f.y; o[2] = f.z; o[3] = pdf;
    f3 swi, sf; float spdf, new_eta; bool is_delta;
    bsdf_sample(m, ctx, r[15], r[16], r[17], r[18], swi, sf, spdf, is_delta, new_eta, &pre);
    o[4] = swi.x; o[5] = swi.y; o[6] = swi.z; o[7] = sf.x; o[8] = sf.y; o[9] = sf.z; o[10] = spdf;
    o[11] = is_delta ? 1.0f : 0.0f; o[12] = new_eta;
    o[13] = fr_dielectric(r[19], r[20], r[21]);
    f3 sch = fr_schlick(r[19], m.base);
    o[14] = sch.x; o[15] = sch.y; o[16] = sch.z;
    f3 wi_l = to_local(wi_w, ctx.T, ctx.B, ctx.N);
    o[17] = ggx_D(wi_l, alpha); o[18] = ggx_lambda(wi_l, alpha); o[19] = ggx_G(ctx.wo, wi_l, alpha);
    float sn, cs; sincos_2pi(r[17], sn, cs);
    o[20] = sn; o[21] = cs;
    f3 vn = sample_visible_normal(ctx.wo.z > 0 ? ctx.wo : ctx.wo * -1.0f, alpha, sqrtf(r[16]), sn, cs);
    o[22] = vn.x; o[23] = vn.y; o[24] = vn.z;
    o[25] = is_valid_color(f) ? 1.0f : 0.0f;
    f3 cl = clamp_radiance(f * 20.0f, 15.0f);
    o[26] = cl.x; o[27] = cl.y; o[28] = cl.z;
    o[29] = ctx.T.x; o[30] = ctx.T.y; o[31] = ctx.T.z; o[32] = ctx.B.x; o[33] = ctx.B.y; o[34] = ctx.B.z;
    o[35] = ctx.wo.x; o[36] = ctx.wo.y; o[37] = ctx.wo.z;
    o[38] = 0.0f; o[39] = 0.0f;
}

uint32_t grid_for(uint32_t items){
    uint32_t g = (items + kBlock - 1) / kBlock;
    if(g < 1u) g = 1u;
    return g > 2048u ? 2048u : g;       // persistent grid-stride loops above 2048 workgroups
}

} // namespace

// ---- launchers --------------------------------------------------------------------------

void launch_generate(hipStream_t s, const Tiling &tl, const CameraDev &cam, PathBuf pb,
                     uint32_t *qcount, int samples_this_pass, uint32_t first_sample, uint64_t seed,
                     WorkCounters *wc){
    uint32_t total = (uint32_t) tl.n_local * (uint32_t) samples_this_pass;
    hipLaunchKernelGGL(k_generate, dim3(grid_for(total)), dim3(kBlock), 0, s, tl, cam, pb, qcount, total,
                       first_sample, seed, wc);
}

void launch_extend(hipStream_t s, const SceneDev &sc, PathBuf pb, const uint32_t *queue, const uint32_t *qcount,
                   uint32_t max_items, int flags, WorkCounters *wc){
    dim3 g(grid_for(max_items)), b(kBlock);
    bool brute = flags & 1, count = flags & 2;
    if(brute && count) hipLaunchKernelGGL((k_extend<true, true>), g, b, 0, s, sc, pb, queue, qcount, wc);
    else if(brute) hipLaunchKernelGGL((k_extend<true, false>), g, b, 0, s, sc, pb, queue, qcount, wc);
    else if(count) hipLaunchKernelGGL((k_extend<false, true>), g, b, 0, s, sc, pb, queue, qcount, wc);
    else hipLaunchKernelGGL((k_extend<false, false>), g, b, 0, s, sc, pb, queue, qcount, wc);
}

void launch_shade(hipStream_t s, const SceneDev &sc, PathBuf pb, const uint32_t *queue, const uint32_t *qcount,
                  uint32_t max_items, uint32_t *next_queue, uint32_t *next_count, ShadowBuf sb, uint32_t *squeue,
                  uint32_t *scount, int max_depth, int max_delta, int roulette, WorkCounters *wc, const PrimaryGen *primary,
                  uint32_t max_groups){
    // enough workgroups for either chunking regime (see k_shade)
    uint32_t g = (max_items + kShadeChunk - 1) / kShadeChunk;
    if(g < (uint32_t) kShadeTargetGroups) g = (uint32_t) kShadeTargetGroups;
    uint32_t small = (max_items + kBlock - 1) / kBlock;
    if(small < g) g = small < 1u ? 1u : small;
    const bool strided = max_groups != 0u && g > max_groups;        // the kernel then walks the chunks with a stride
    if(strided) g = max_groups;
    PrimaryGen none{};
    if(primary) hipLaunchKernelGGL((k_shade<true, false>), dim3(g), dim3(kBlock), 0, s, sc, pb, queue, qcount, next_queue,
                                   next_count, sb, squeue, scount, max_depth, max_delta, roulette, wc, *primary);
    else if(strided) hipLaunchKernelGGL((k_shade<false, true>), dim3(g), dim3(kBlock), 0, s, sc, pb, queue, qcount, next_queue,
                                        next_count, sb, squeue, scount, max_depth, max_delta, roulette, wc, none);
    else hipLaunchKernelGGL((k_shade<false, false>), dim3(g), dim3(kBlock), 0, s, sc, pb, queue, qcount, next_queue,
                            next_count, sb, squeue, scount, max_depth, max_delta, roulette, wc, none);
}

void launch_connect(hipStream_t s, const SceneDev &sc, PathBuf pb, ShadowBuf sb, const uint32_t *squeue,
                    const uint32_t *scount, uint32_t max_items, int flags, WorkCounters *wc){
    dim3 g(grid_for(max_items)), b(kBlock);
    bool brute = flags & 1, count = flags & 2;
    if(brute && count) hipLaunchKernelGGL((k_connect<true, true>), g, b, 0, s, sc, pb, sb, squeue, scount, wc);
    else if(brute) hipLaunchKernelGGL((k_connect<true, false>), g, b, 0, s, sc, pb, sb, squeue, scount, wc);
    else if(count) hipLaunchKernelGGL((k_connect<false, true>), g, b, 0, s, sc, pb, sb, squeue, scount, wc);
    else hipLaunchKernelGGL((k_connect<false, false>), g, b, 0, s, sc, pb, sb, squeue, scount, wc);
}

void launch_trace(hipStream_t s, const SceneDev &sc, PathBuf pb, ShadowBuf sb, const uint32_t *equeue,
                  const uint32_t *ecount, uint32_t max_extend, const uint32_t *squeue, const uint32_t *scount,
                  uint32_t max_shadow, int stack_levels, int flags, int tuning, WorkCounters *wc, const TraceSplit *split,
                  const PrimaryGen *primary, uint32_t max_groups){
    uint32_t chunk = ((tuning >> 16) & 0xFF) ? (uint32_t) ((tuning >> 16) & 0xFF) * 256u : (uint32_t) kTraceChunk;
    int refill_min = ((tuning >> 8) & 0xFF) ? ((tuning >> 8) & 0xFF) : kRefillMin;
    int node_min = ((tuning >> 24) & 0x7F) ? ((tuning >> 24) & 0x7F) : kNodeMin;
    if(node_min == 0x7F) node_min = 0;                 // tuning: switch the early leaf break off
    // worst-case grid for either chunking regime of k_trace (long queues: `chunk` rays per workgroup,
    // queues shorter than kTraceShortQueue: kBlock rays per workgroup)
    auto groups = [&](uint32_t items){
        uint32_t a = (items + chunk - 1) / chunk;
        uint32_t shortest = items < kTraceShortQueue ? items : kTraceShortQueue;
        uint32_t b = (shortest + kBlock - 1) / kBlock;
        return a > b ? a : b;
    };
    uint32_t g = (ecount ? groups(max_extend) : 0u) + (scount ? groups(max_shadow) : 0u);
    if(g == 0u) return;
    if(max_groups != 0u && g > max_groups) g = max_groups;          // k_trace walks the chunks with a stride
    if(stack_levels < 1) stack_levels = 1;
    if(stack_levels > kStackDepth) stack_levels = kStackDepth;
    int stack_words = (stack_levels + 1) * kBlock;                            // + the spare level of the branch-free step
    const bool count = (flags & 2) != 0;
    LongQueues lq{};
    if(split && split->budget > 0 && !count){
        lq.equeue = split->equeue; lq.ecount = split->ecount; lq.squeue = split->squeue; lq.scount = split->scount;
        lq.budget = (uint32_t) split->budget;
        // a ray of this launch is set aside after `budget` node steps, so its stack never grows past that many entries
        if(stack_levels > split->budget && !(flags & 4)) stack_words = (split->budget + 1) * kBlock;      // flags bit 2: development A/B
    }
    size_t lds = (size_t) stack_words * sizeof(uint32_t) + (lq.budget ? (size_t) chunk * sizeof(uint32_t) : 0);
    const bool top = lq.budget != 0u && lq.budget <= (uint32_t) kTopLevels && !(tuning & 0x80);       // tuning bit 7: node fetches from global memory (A/B)
    PrimaryGen none{};
    const PrimaryGen &pg = primary ? *primary : none;
#define HPT_LAUNCH_TRACE(C, T, P) hipLaunchKernelGGL((k_trace<C, false, T, P, 0, false>), dim3(g), dim3(kBlock), lds, s, sc, pb, sb, equeue, ecount, squeue, \
                                                     scount, chunk, refill_min, node_min, stack_words, lq, wc, pg, (uint32_t *) nullptr)
    if(count) HPT_LAUNCH_TRACE(true, false, false);
    else if(top && primary) HPT_LAUNCH_TRACE(false, true, true);
    else if(top) HPT_LAUNCH_TRACE(false, true, false);
    else if(primary) HPT_LAUNCH_TRACE(false, false, true);
    else HPT_LAUNCH_TRACE(false, false, false);
#undef HPT_LAUNCH_TRACE
}

// second launch of a split trace step: the rays launch_trace set aside.  Their number is only known on
// the device, so a fixed grid walks the chunks.
void launch_trace_resume(hipStream_t s, const SceneDev &sc, PathBuf pb, ShadowBuf sb, bool extend, bool shadow,
                         uint32_t max_items, int stack_levels, WorkCounters *wc, const TraceSplit &split, const PrimaryGen *primary,
                         uint32_t max_groups, uint32_t *deep_stack, bool tiny_lds_share, bool wide, int dev_tuning){
    if(!extend && !shadow) return;
    if(stack_levels < 1) stack_levels = 1;
    if(stack_levels > kStackDepth) stack_levels = kStackDepth;
    // deep_stack: the lanes' stack levels past kResumeLdsLevels live in global memory (resume_deep_stack_words()); a tree
    // that is no deeper keeps the plain LDS stack
    // (tiny_lds_share: two levels in LDS, so that the tests walk the global-memory levels on every ray)
    const int lds_levels = tiny_lds_share ? 2 : kResumeLdsLevels;
    const bool spill = deep_stack != nullptr && stack_levels > lds_levels;
    int stack_words = ((spill ? lds_levels : stack_levels) + 1) * kBlock;
    // every ray of this launch is a long one: lanes refill sooner, a workgroup takes more rays (a lane
    // gets ~8 rays, which evens out their lengths) and the leaf phase waits for fewer stragglers
    uint32_t chunk2 = kLongChunk; int refill2 = kLongRefillMin, node_min2 = kLongNodeMin;
    {   // development sweeps (hpt_params.flags bits 21-28): refill threshold, early-leaf-break threshold, chunk
        static const int refill_tab[8] = { 0, 8, 16, 24, 32, 40, 48, 56 }, node_tab[8] = { 0, 1, 2, 4, 12, 16, 24, 32 };
        static const uint32_t chunk_tab[4] = { 0u, 1024u, 2048u, 512u };
        if(dev_tuning & 7) refill2 = refill_tab[dev_tuning & 7];
        if((dev_tuning >> 3) & 7) node_min2 = node_tab[(dev_tuning >> 3) & 7];
        if((dev_tuning >> 6) & 3) chunk2 = chunk_tab[(dev_tuning >> 6) & 3];
    }
    uint32_t per = (max_items + kBlock - 1) / kBlock;
    uint64_t g = (uint64_t) per * ((extend ? 1u : 0u) + (shadow ? 1u : 0u));
    uint32_t g2 = g < kResumeMaxGroups ? (uint32_t) (g < 1u ? 1u : g) : kResumeMaxGroups;
    if(max_groups != 0u && g2 > max_groups) g2 = max_groups;          // blind tail iterations (HPT_FLAG_NO_HOST_WAIT)
    LongQueues none{};
    PrimaryGen no_primary{};
    const size_t lds = (size_t) stack_words * sizeof(uint32_t);
    const uint32_t *eq = extend ? split.equeue : nullptr, *ec = extend ? split.ecount : nullptr;
    const uint32_t *sq = shadow ? split.squeue : nullptr, *scn = shadow ? split.scount : nullptr;
#define HPT_LAUNCH_RESUME(P, K, W, PG) hipLaunchKernelGGL((k_trace<false, true, false, P, K, W>), dim3(g2), dim3(kBlock), lds, s, sc, pb, sb, eq, ec, sq, scn, \
                                                          chunk2, refill2, node_min2, stack_words, none, wc, PG, deep_stack)
    // the long rays walk the four-wide twin of the tree (up to three children stacked per step: needs the deep-stack buffer);
    // `wide` false = the binary walk (development A/B, and scenes whose wide tree would outgrow the stack)
    if(wide && deep_stack && sc.wnodes && 3 * sc.wide_depth + 2 <= lds_levels + kDeepLevels){
        stack_words = (lds_levels + 1) * kBlock;
        const size_t lds = (size_t) stack_words * sizeof(uint32_t);
        if(!(dev_tuning & 7)) refill2 = kWideRefillMin;
        if(!((dev_tuning >> 3) & 7)) node_min2 = kWideNodeMin;
        if(primary && extend){ if(tiny_lds_share) HPT_LAUNCH_RESUME(true, 2, true, *primary); else HPT_LAUNCH_RESUME(true, kResumeLdsLevels, true, *primary); }
        else { if(tiny_lds_share) HPT_LAUNCH_RESUME(false, 2, true, no_primary); else HPT_LAUNCH_RESUME(false, kResumeLdsLevels, true, no_primary); }
        return;
    }
    if(primary && extend){
        if(spill && tiny_lds_share) HPT_LAUNCH_RESUME(true, 2, false, *primary);
        else if(spill) HPT_LAUNCH_RESUME(true, kResumeLdsLevels, false, *primary);
        else HPT_LAUNCH_RESUME(true, 0, false, *primary);
    } else {
        if(spill && tiny_lds_share) HPT_LAUNCH_RESUME(false, 2, false, no_primary);
        else if(spill) HPT_LAUNCH_RESUME(false, kResumeLdsLevels, false, no_primary);
        else HPT_LAUNCH_RESUME(false, 0, false, no_primary);
    }
#undef HPT_LAUNCH_RESUME
}

size_t resume_deep_stack_words(){ return (size_t) kDeepLevels * kResumeMaxGroups * kBlock; }

void launch_tri_frames(hipStream_t s, const float4 *tris, int num_tris, float4 *frames){
    if(num_tris <= 0) return;
    hipLaunchKernelGGL(k_tri_frames, dim3((num_tris + kBlock - 1) / kBlock), dim3(kBlock), 0, s, tris, num_tris, frames);
}

void launch_resolve(hipStream_t s, const Tiling &tl, PathBuf pb, float4 *accum, int samples_this_pass){
    uint32_t g = ((uint32_t) tl.n_local + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_resolve, dim3(g), dim3(kBlock), 0, s, tl, pb, accum, samples_this_pass);
}

void launch_finalize(hipStream_t s, const Tiling &tl, const float4 *accum, float *d_local, float divisor){
    uint32_t g = ((uint32_t) tl.n_local + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_finalize, dim3(g), dim3(kBlock), 0, s, tl, accum, d_local, divisor);
}

void launch_untile(hipStream_t s, const Tiling &tl, const float *d_gathered, float *d_image){
    uint32_t g = ((uint32_t) (tl.W * tl.H) + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_untile, dim3(g), dim3(kBlock), 0, s, tl, d_gathered, d_image);
}

void launch_tonemap(hipStream_t s, const float *d_linear, void *d_bytes, unsigned long long num_values, int bgr, const float *d_thresholds){
    if(num_values == 0) return;
    unsigned long long words = (num_values + 3ull) / 4ull;
    hipLaunchKernelGGL(k_tonemap, dim3((unsigned) ((words + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, d_linear, (uint32_t *) d_bytes,
                       num_values, bgr, d_thresholds);
}

void launch_probe_functions(hipStream_t s, const float *d_in, int n, float *d_out){
    if(n <= 0) return;
    hipLaunchKernelGGL(k_probe_functions, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, d_in, n, d_out);
}

void launch_probe_closest(hipStream_t s, const SceneDev &sc, const float *org, const float *dir, int n, int flags,
                          float *t_out, int32_t *prim_out){
    dim3 g((n + kBlock - 1) / kBlock), b(kBlock);
    if(flags & 1) hipLaunchKernelGGL((k_probe_closest<true>), g, b, 0, s, sc, org, dir, n, t_out, prim_out);
    else hipLaunchKernelGGL((k_probe_closest<false>), g, b, 0, s, sc, org, dir, n, t_out, prim_out);
}

void launch_probe_visibility(hipStream_t s, const SceneDev &sc, const float *p1, const float *p2, int n, int flags,
                             int32_t *vis_out){
    dim3 g((n + kBlock - 1) / kBlock), b(kBlock);
    if(flags & 1) hipLaunchKernelGGL((k_probe_visibility<true>), g, b, 0, s, sc, p1, p2, n, vis_out);
    else hipLaunchKernelGGL((k_probe_visibility<false>), g, b, 0, s, sc, p1, p2, n, vis_out);
}

} // namespace hpt
