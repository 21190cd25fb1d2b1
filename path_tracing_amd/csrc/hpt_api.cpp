// C ABI of libhpt.so (include/hpt.h): scene upload, the wavefront render loop, tiling and
// the one-shot pt_render_wrapper equivalent.  Compiled with hipcc (host code only here).
//
// Render loop per pass (S samples of every local pixel in flight):
//   generate -> repeat { trace, shade } until the queue drains -> trace -> resolve
// with two passes in flight at a time on two streams (render_local).
// Queue counters live in device memory, one slot per iteration, so the host issues the first
// eye_depth iterations without ever reading the device back; only scenes whose paths are still
// alive after that (chains of free delta bounces, reference src/pt_cu.cu:228) cost one
// counter read-back per extra iteration.
#include "../../include/hpt.h"
#include "hpt_scene.h"
#include "pt_kernels.h"
#include "bdpt_kernels.h"

#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <new>
#include <string>
#include <vector>

using namespace hpt;

namespace hpt {   // hpt_multi.cpp
bool multi_matches(const hpt_multi *m, int n_devices, const void *lights, int nl, const void *spheres, int ns, const void *tris, int nt);
}

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg){ g_err = msg; return code; }

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if(e_ != hipSuccess) \
    return fail(e_ == hipErrorOutOfMemory ? HPT_ERR_NOMEM : HPT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } while(0)

template <typename T, typename A>
hipError_t upload(const std::vector<T, A> &v, T **dptr){
    *dptr = nullptr;
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    hipError_t e = hipMalloc((void **) dptr, bytes);
    if(e != hipSuccess) return e;
    if(!v.empty()) e = hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

struct DevBuf {              // device allocation released on every return path
    void *p = nullptr;
    ~DevBuf(){ if(p) hipFree(p); }
    hipError_t alloc(size_t bytes){ return hipMalloc(&p, bytes ? bytes : 1); }
    template <typename T> T *as() const { return (T *) p; }
};

struct TimedLaunch { hipEvent_t a, b; int cls; };
// Frames a scene whose rays are nearly all long used to render without the split (round 2: the binary resume launch cost
// more than the single-launch walk there).  With the four-wide resume launch the split wins on that scene too (100 k random
// triangles, 16 spp: 29.9 ms split, 33.1 unsplit binary, 30.6 unsplit four-wide), so the hold-off is switched off; the
// asynchronous read-back of the set-aside share stays (hpt_stats.long_rays_last_pass).
constexpr int kSplitHold = 0;
constexpr int kMaxPipes = 4;             // passes of a render in flight at a time, at most (default 2; development: flags bits 29-30)
// device memory per path slot of one pipeline (ensure_pass): path state 80 B, pending shadow ray 48 B, five queues of 4 B
constexpr double kBytesPerPathSlot = 148.0;

static_assert(sizeof(hpt_stats) == 312 && sizeof(hpt_params) == 40, "ABI records: keep path_tracing_amd/__init__.py and tests/test_boundary.py in step");

} // namespace

struct hpt_scene {
    SceneDev sd{};
    BvhNode *d_nodes = nullptr; QBvhNode *d_qnodes = nullptr; WideNode *d_wnodes = nullptr; DevTriangle *d_tris = nullptr; DevRound *d_rounds = nullptr;
    DevMaterial *d_mats = nullptr; DevLight *d_lights = nullptr;
    float4 *d_tri_frames = nullptr;
    int device = 0;
    int stack_levels = kStackDepth;       // traversal stack entries per lane
    int last_counter_stride = 0, last_budget = 0;   // layout of `counters` after the last PT render (0: not a PT render)
    // default budget only: the share of rays the last split render set aside is read back asynchronously; a
    // scene whose rays are nearly all long (every ray restarts: the split only costs) renders the next
    // kSplitHold frames without the split, then is probed again
    uint32_t *h_split = nullptr; int h_split_words = 0; hipEvent_t ev_split = nullptr;
    bool split_probe_pending = false; int split_probe_stride = 0, split_hold = 0;
    int num_cus = 256;

    // workspace, grown on demand.  Two passes of a PT render are in flight at a time (render_local), each with its own
    // path state, queues and counters (pass[0] on the caller's stream, pass[1] on p2_stream) -- the kernels of one
    // fill the issue slots the other leaves idle; everything else (BDPT, probes) uses pass[0]
    struct PassBuffers {
        size_t cap_paths = 0;
        PathBuf pb{}; ShadowBuf sb{};
        uint32_t *queue[2] = { nullptr, nullptr };   // path queues (ping-pong)
        uint32_t *squeue = nullptr;                  // shadow queue (path slots)
        uint32_t *lqueue[2] = { nullptr, nullptr };  // rays set aside by the first trace launch: closest-hit, shadow
        uint32_t *deep_stack = nullptr;              // stack levels of the resume launch past its LDS share (launch_trace_resume)
        uint32_t *counters = nullptr; int n_counters = 0;
        uint32_t *h_count = nullptr;                 // pinned read-back word
    } pass[kMaxPipes];
    size_t cap_local = 0;
    hipStream_t px_stream[kMaxPipes] = {}; int px_priority[kMaxPipes] = {}; hipEvent_t px_fork = nullptr, px_done[kMaxPipes] = {};   // pipelines 1..: own streams
    const uint32_t *last_counters = nullptr;     // counters of the last pass rendered (either pipeline)
    float4 *accum = nullptr;
    WorkCounters *d_wc = nullptr;
    float *d_local_own = nullptr; size_t cap_local_own = 0;
    float *d_image_own = nullptr; size_t cap_image_own = 0;

    // bidirectional (cpu_bdpt-estimator) path: host copy of the records + grouping, device scene built on first use
    std::vector<unsigned char> h_lights, h_spheres, h_tris;
    int nl = 0, ns = 0, nt = 0;
    std::vector<int32_t> g_kind, g_index, g_group;
    bool bd_ready = false;
    BdptSceneDev bd{};
    BvhNode *bd_nodes = nullptr; DevTriangle *bd_tris = nullptr; DevRound *bd_spheres = nullptr; DevGroup *bd_groups = nullptr;
    DevMaterial *bd_mats = nullptr; DevLight *bd_lights = nullptr;
    BdptPathBuf bp{}; size_t bd_cap_slots = 0, bd_cap_hist = 0, bd_cap_contrib = 0, bd_cap_valid = 0;
    LightVertexDev *d_lv = nullptr; size_t bd_cap_lv = 0;
    LightVertexCtx *d_lctx = nullptr; size_t bd_cap_lctx = 0;
    uint32_t *cqueue = nullptr; size_t bd_cap_cqueue = 0;

    hipEvent_t ev_start = nullptr, ev_stop = nullptr; bool ev_valid = false;
    std::vector<TimedLaunch> timed; std::vector<hipEvent_t> event_pool; size_t event_next = 0;
    hpt_stats stats{};
    bool stats_pending = false; int last_flags = 0;
};

namespace {

using PassBuffers = hpt_scene::PassBuffers;

void free_pass(PassBuffers &w){
    hipFree(w.pb.org_eta); hipFree(w.pb.dir_flags); hipFree(w.pb.thr); hipFree(w.pb.col); hipFree(w.pb.rng); hipFree(w.pb.hit);
    hipFree(w.sb.org_max); hipFree(w.sb.dir); hipFree(w.sb.contrib);
    hipFree(w.queue[0]); hipFree(w.queue[1]); hipFree(w.squeue); hipFree(w.lqueue[0]); hipFree(w.lqueue[1]); hipFree(w.deep_stack);
    w.pb = PathBuf{}; w.sb = ShadowBuf{};
    w.queue[0] = w.queue[1] = w.squeue = w.lqueue[0] = w.lqueue[1] = w.deep_stack = nullptr;
    w.cap_paths = 0;
}

int ensure_pass(PassBuffers &w, size_t paths, int n_counters){
    if(paths > w.cap_paths){
        free_pass(w);
        HIP_TRY(hipMalloc((void **) &w.pb.org_eta, paths * sizeof(float4)));
        HIP_TRY(hipMalloc((void **) &w.pb.dir_flags, paths * sizeof(float4)));
        HIP_TRY(hipMalloc((void **) &w.pb.thr, paths * sizeof(float4)));
        HIP_TRY(hipMalloc((void **) &w.pb.col, paths * sizeof(float4)));
        HIP_TRY(hipMalloc((void **) &w.pb.rng, paths * sizeof(uint2)));
        HIP_TRY(hipMalloc((void **) &w.pb.hit, paths * sizeof(uint2)));
        HIP_TRY(hipMalloc((void **) &w.sb.org_max, paths * sizeof(float4)));
        HIP_TRY(hipMalloc((void **) &w.sb.dir, paths * sizeof(float4)));
        HIP_TRY(hipMalloc((void **) &w.sb.contrib, paths * sizeof(float4)));
        HIP_TRY(hipMalloc((void **) &w.queue[0], paths * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void **) &w.queue[1], paths * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void **) &w.squeue, paths * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void **) &w.lqueue[0], paths * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void **) &w.lqueue[1], paths * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void **) &w.deep_stack, resume_deep_stack_words() * sizeof(uint32_t)));
        w.cap_paths = paths;
    }
    if(n_counters > w.n_counters){
        hipFree(w.counters); w.counters = nullptr; w.n_counters = 0;
        HIP_TRY(hipMalloc((void **) &w.counters, (size_t) n_counters * sizeof(uint32_t)));
        w.n_counters = n_counters;
    }
    if(!w.h_count) HIP_TRY(hipHostMalloc((void **) &w.h_count, 64));
    return HPT_OK;
}

// streams and events of the pipelines past the first (which runs on the caller's stream)
int ensure_pipes(hpt_scene *s, hipStream_t caller, int npipes){
    // The pipelines only overlap if their streams sit on different hardware queues.  The runtime maps streams
    // of one priority onto a small pool of queues (GPU_MAX_HW_QUEUES, 4 by default) by reference count, so once a
    // process holds a few more streams -- RCCL's, after a communicator exists -- a second stream of the caller's
    // priority can land on the caller's queue and the passes serialise (measured: 169 ms per config-3 render
    // instead of 161).  Streams of another priority come from another pool: the second pipeline takes the highest
    // priority unless the caller's stream already has it, then the default one; further pipelines take the remaining levels in turn.
    int pr_least = 0, pr_greatest = 0, pr_caller = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest));
    if(hipStreamGetPriority(caller, &pr_caller) != hipSuccess){ (void) hipGetLastError(); pr_caller = 0; }
    std::vector<int> levels;                                   // every level but the caller's, highest first
    for(int l = pr_greatest; l <= pr_least; ++l) if(l != pr_caller) levels.push_back(l);
    if(levels.empty()) levels.push_back(pr_caller);
    for(int k = 1; k < npipes; ++k){
        const int want = levels[(size_t) (k - 1) % levels.size()];
        if(s->px_stream[k] && s->px_priority[k] != want){ hipStreamSynchronize(s->px_stream[k]); hipStreamDestroy(s->px_stream[k]); s->px_stream[k] = nullptr; }
        if(!s->px_stream[k]){
            HIP_TRY(hipStreamCreateWithPriority(&s->px_stream[k], hipStreamNonBlocking, want));
            s->px_priority[k] = want;
        }
        if(!s->px_done[k]) HIP_TRY(hipEventCreateWithFlags(&s->px_done[k], hipEventDisableTiming));
    }
    if(!s->px_fork) HIP_TRY(hipEventCreateWithFlags(&s->px_fork, hipEventDisableTiming));
    return HPT_OK;
}

int ensure_workspace(hpt_scene *s, size_t paths, size_t n_local, int n_counters){
    int rc = ensure_pass(s->pass[0], paths, n_counters);
    if(rc) return rc;
    if(n_local > s->cap_local){
        hipFree(s->accum); s->accum = nullptr; s->cap_local = 0;
        HIP_TRY(hipMalloc((void **) &s->accum, n_local * sizeof(float4)));
        s->cap_local = n_local;
    }
    if(!s->d_wc) HIP_TRY(hipMalloc((void **) &s->d_wc, sizeof(WorkCounters)));
    if(!s->ev_start){ HIP_TRY(hipEventCreate(&s->ev_start)); HIP_TRY(hipEventCreate(&s->ev_stop)); }
    return HPT_OK;
}

int make_tiling(int W, int H, const hpt_params *p, Tiling &tl){
    if(W <= 0 || H <= 0) return fail(HPT_ERR_INVALID, "image size must be positive");
    int world = (p && p->world > 1) ? p->world : 1;
    int rank = (p && p->world > 1) ? p->rank : 0;
    int tile = (p && p->tile > 0) ? p->tile : 32;
    if(tile % 8 != 0 || tile > 1024) return fail(HPT_ERR_INVALID, "tile must be a multiple of 8 (<= 1024)");
    if(rank < 0 || rank >= world) return fail(HPT_ERR_INVALID, "rank outside [0, world)");
    tl.W = W; tl.H = H; tl.tile = tile;
    tl.tiles_x = (W + tile - 1) / tile; tl.tiles_y = (H + tile - 1) / tile;
    tl.ntiles = tl.tiles_x * tl.tiles_y;
    tl.rank = rank; tl.world = world;
    long long per_rank = (tl.ntiles + world - 1) / world;
    long long n_local = per_rank * tile * tile;
    if(n_local > 0x7FFFFFFFll) return fail(HPT_ERR_INVALID, "local framebuffer too large");
    tl.n_local = (int) n_local;
    return HPT_OK;
}

hipEvent_t pool_event(hpt_scene *s){
    if(s->event_next == s->event_pool.size()){
        hipEvent_t e; hipEventCreate(&e); s->event_pool.push_back(e);
    }
    return s->event_pool[s->event_next++];
}

struct LaunchTimer {        // brackets one launch with events when TIME_KERNELS is set
    hpt_scene *s; hipStream_t st; bool on; TimedLaunch tl;
    LaunchTimer(hpt_scene *s_, hipStream_t st_, bool on_, int cls) : s(s_), st(st_), on(on_) {
        if(on){ tl.a = pool_event(s); tl.b = pool_event(s); tl.cls = cls; hipEventRecord(tl.a, st); }
    }
    ~LaunchTimer(){ if(on){ hipEventRecord(tl.b, st); s->timed.push_back(tl); } }
};

// The scene's buffers live on the device that was current when it was created; launching from a thread whose current
// device is another one would hand those pointers to the wrong GPU (a fault, not an error code).
int on_scene_device(const hpt_scene *s){
    int dev = -1;
    if(hipGetDevice(&dev) != hipSuccess || dev != s->device)
        return fail(HPT_ERR_INVALID, "the scene lives on another device than the calling thread's current one (hipSetDevice first)");
    return HPT_OK;
}

// the wavefront render loop; everything is enqueued on `stream`
int render_local(hpt_scene *s, const void *camera, int W, int H, int eye_depth, int spp,
                 const hpt_params *params, float *d_local, hipStream_t stream){
    if(!s) return fail(HPT_ERR_INVALID, "null scene");
    if(!camera || !d_local) return fail(HPT_ERR_INVALID, "null camera or output");
    if(spp <= 0 || eye_depth <= 0 || eye_depth > 255) return fail(HPT_ERR_INVALID, "spp must be > 0 and eye_depth in [1, 255]");
    if(int rcd = on_scene_device(s)) return rcd;
    hpt_params P; memset(&P, 0, sizeof P);
    if(params) P = *params;
    if(P.max_delta <= 0) P.max_delta = 64;
    if(P.max_delta > 250) P.max_delta = 250;
    Tiling tl;
    int rc = make_tiling(W, H, &P, tl);
    if(rc) return rc;

    const float *cf = (const float *) camera;       // CudaCamera: eye, U, V, W, UL, dx, dy (12 B each)
    CameraDev cam;
    memcpy(cam.eye, cf + 0, 12); memcpy(cam.UL, cf + 12, 12); memcpy(cam.dx, cf + 15, 12); memcpy(cam.dy, cf + 18, 12);

    const int flags = P.flags;
    const bool count = (flags & HPT_FLAG_COUNT_WORK) != 0;
    const bool timek = (flags & HPT_FLAG_TIME_KERNELS) != 0;
    const bool brute = (flags & HPT_FLAG_BRUTE_FORCE) != 0;
    const bool legacy = brute || (P.reserved & 1);          // separate extend/connect kernels (the scan variants)
    const int kflags = (brute ? 1 : 0) | (count ? 2 : 0) | ((flags >> 16) & 1 ? 4 : 0);      // flags bits 16-31: development switches

    // Samples in flight per pass: about 128 Mi path slots (19 GiB of path state, queues and shadow records per pipeline:
    // little on a 288 GB device).  Fewer, larger passes amortise the low-occupancy tail iterations of every pass
    // (config 3, ms per 256-spp render, one pipeline: 4 Mi slots 291, 16 Mi 219, 64 Mi 146, 128 Mi 142, 256 Mi 139).
    // Two passes are in flight at a time, on two streams with a workspace each: while one pipeline's kernel drains or
    // waits on memory the other's waves take the issue slots (64 Mi slots each: 138.1 ms, 128 Mi each: 135.7).  A render
    // that fits one pass is cut into two half passes for the same reason -- the share of one rank of a multi-GPU render
    // is such a render (rank 0 of 4 at config 3, 64 Mi slots: one pass 37.2-37.4 ms, two half passes 35.3-36.5; of 2:
    // 72.1 -> 68.2; of 8: 19.5 -> 18.7; on another box 35.4 against 35.6: never a loss beyond the noise) -- unless it is
    // so small (< 1 Mi slots) that launch latencies are what it costs.
    int npipes = (!(flags & HPT_FLAG_SINGLE_PIPELINE) && !count && !legacy) ? 2 + ((flags >> 29) & 3) : 1;     // flags bits 29-30 (development): 3 or 4 pipelines
    if(npipes > kMaxPipes) npipes = kMaxPipes;
    int spass = P.samples_per_pass;
    if(spass <= 0){
        long long target = 128ll << 20;
        // ... on a device that has the memory for it: when the workspace would have to grow, the pass is sized so that
        // both pipelines' state fits in 70 % of what is free now plus what the scene already holds (a smaller device,
        // or several scenes on one device, get smaller passes instead of HPT_ERR_NOMEM; the image does not depend on it)
        size_t have = 0; for(const PassBuffers &w : s->pass) have += w.cap_paths;
        if((size_t) std::min<long long>(target, (long long) tl.n_local * spp) > s->pass[0].cap_paths){
            size_t free_b = 0, total_b = 0;
            if(hipMemGetInfo(&free_b, &total_b) == hipSuccess){
                const double usable = 0.7 * ((double) free_b + (double) have * kBytesPerPathSlot);
                const long long fit = (long long) (usable / (kBytesPerPathSlot * (double) npipes));
                if(fit < target){
                    target = std::max<long long>(fit, tl.n_local);
                    // a workspace sized this way earlier is kept (no reallocation for a few per cent more)
                    if((long long) s->pass[0].cap_paths >= target * 3 / 4) target = (long long) s->pass[0].cap_paths;
                }
            } else (void) hipGetLastError();
        }
        spass = (int) std::max<long long>(1, target / tl.n_local);
        spass = std::min(spass, spp);
        // every round of the render keeps all pipelines busy: the passes of the render are cut to a multiple of their number
        if(npipes > 1 && spp >= npipes && (long long) tl.n_local * spp >= (1ll << 20)){
            const int rounds = (spp + spass * npipes - 1) / (spass * npipes);
            spass = (spp + rounds * npipes - 1) / (rounds * npipes);
        }
    }
    spass = std::min(spass, spp);
    const int npass = (spp + spass - 1) / spass;
    if(npass < npipes) npipes = npass;
    size_t paths = (size_t) tl.n_local * spass;
    if(paths > 0x7FFFFFF0ull) return fail(HPT_ERR_INVALID, "too many path slots per pass");
    int max_iters = eye_depth + P.max_delta + 1;
    int n_counters = 4 * (max_iters + 2);
    rc = ensure_workspace(s, paths, tl.n_local, n_counters);
    if(rc) return rc;
    for(int k = 1; k < npipes; ++k){ rc = ensure_pass(s->pass[k], paths, n_counters); if(rc) return rc; }
    if(npipes > 1){ rc = ensure_pipes(s, stream, npipes); if(rc) return rc; }

    WorkCounters *wc = count ? s->d_wc : nullptr;
    s->timed.clear(); s->event_next = 0;
    s->last_flags = flags;
    s->stats.ms_total = s->stats.ms_extend = s->stats.ms_shade = s->stats.ms_connect = s->stats.ms_other = 0.0;
    s->stats.n_extend = s->stats.n_shade = s->stats.n_connect = s->stats.n_other = 0;

    // split of the trace step: decided from the previous render of this scene (see hpt_scene::h_split)
    const bool auto_budget = ((P.reserved >> 1) & 0x3F) == 0;
    bool split_off = false;
    if(auto_budget){
        if(s->split_probe_pending && hipEventQuery(s->ev_split) == hipSuccess){
            s->split_probe_pending = false;
            uint64_t traced = 0, set_aside = 0;
            for(int i = 0; i < s->split_probe_stride; ++i){
                traced += (uint64_t) s->h_split[i] + s->h_split[(size_t) s->split_probe_stride + i];
                set_aside += (uint64_t) s->h_split[(size_t) 2 * s->split_probe_stride + i] + s->h_split[(size_t) 3 * s->split_probe_stride + i];
            }
            if(traced > 0 && set_aside * 2 > traced) s->split_hold = kSplitHold;
        }
        if(s->split_hold > 0){ split_off = true; --s->split_hold; }
    }
    // node-step budget of the first trace launch (tuning bits 1..6: 0 = default, 0x3F = no split)
    int budget = (P.reserved >> 1) & 0x3F;
    budget = budget == 0 ? kTraceBudget : (budget == 0x3F ? 0 : budget);
    if(count || legacy) budget = 0;                     // work counts are those of the plain single-launch traversal
    int tuning = P.reserved;
    if(auto_budget && split_off){
        budget = 0;                                     // all rays long: single launches with the long-ray tuning
        if(((tuning >> 16) & 0xFF) == 0) tuning |= (int) (kLongChunk / 256u) << 16;
        if(((tuning >> 8) & 0xFF) == 0) tuning |= kLongRefillMin << 8;
        if(((tuning >> 24) & 0x7F) == 0) tuning |= kLongNodeMin << 24;
    }
    s->last_budget = budget;
    const int roulette = (flags & HPT_FLAG_RUSSIAN_ROULETTE) ? 1 : 0;

    HIP_TRY(hipMemsetAsync(s->d_wc, 0, sizeof(WorkCounters), stream));
    HIP_TRY(hipMemsetAsync(s->accum, 0, (size_t) tl.n_local * sizeof(float4), stream));
    HIP_TRY(hipEventRecord(s->ev_start, stream));

    // one pass in flight on one pipeline
    struct Pass {
        PathBuf pb; ShadowBuf sb; uint32_t *queue[2], *squeue, *lqueue[2], *deep_stack; uint32_t *counters, *h_count; hipStream_t st;
        int sthis = 0, cur = 0, pending_shadow = -1; uint32_t slots = 0; PrimaryGen primary{};
        uint32_t *qcnt = nullptr, *scnt = nullptr, *lecnt = nullptr, *lscnt = nullptr;
    };
    Pass pipe[kMaxPipes]{};
    for(int k = 0; k < npipes; ++k){
        const PassBuffers &w = s->pass[k];
        Pass &q = pipe[k];
        q.pb = w.pb; q.sb = w.sb; q.queue[0] = w.queue[0]; q.queue[1] = w.queue[1]; q.squeue = w.squeue;
        q.lqueue[0] = w.lqueue[0]; q.lqueue[1] = w.lqueue[1];
        q.deep_stack = ((flags >> 18) & 1) ? nullptr : w.deep_stack;      // flags bits 16-31: development switches (bit 18: whole stack in LDS; bit 19: two levels in LDS; bit 20: binary resume launch; bits 21-28: its tuning)
        q.counters = w.counters; q.h_count = w.h_count; q.st = k == 0 ? stream : s->px_stream[k];
    }
    for(Pass &q : pipe){
        if(!q.counters) continue;
        q.qcnt = q.counters;                              // qcnt[i]: paths entering iteration i
        q.scnt = q.counters + (max_iters + 2);            // scnt[i]: shadow rays of iteration i
        q.lecnt = q.counters + 2 * (max_iters + 2);       // lecnt[i] / lscnt[i]: rays the trace launch of
        q.lscnt = q.counters + 3 * (max_iters + 2);       // iteration i set aside for its second launch
    }

    // Iteration 0 needs no generate launch: its trace and shade kernels recompute the camera ray of a slot from the slot
    // number (PRIMARY variants; -3 % per render: the launch and the 72 B per path it writes and iteration 0 reads back).
    // The counting and the scan variants keep the stored form.
    const bool in_flight_primaries = !count && !legacy && !((flags >> 17) & 1);      // flags bits 16-31: development switches
    auto begin_pass = [&](Pass &q, int done) -> int {
        q.sthis = std::min(spass, spp - done);
        q.slots = (uint32_t) tl.n_local * (uint32_t) q.sthis;
        q.cur = 0; q.pending_shadow = -1;
        HIP_TRY(hipMemsetAsync(q.counters, 0, (size_t) n_counters * sizeof(uint32_t), q.st));
        q.primary.tl = tl; q.primary.cam = cam; q.primary.first_sample = (uint32_t) (P.sample_offset + done); q.primary.pad = 0u; q.primary.seed = P.seed;
        if(in_flight_primaries){
            HIP_TRY(hipMemsetD32Async((hipDeviceptr_t) &q.qcnt[0], (int) q.slots, 1, q.st));
        } else {
            LaunchTimer t(s, q.st, timek, 3);
            launch_generate(q.st, tl, cam, q.pb, &q.qcnt[0], q.sthis, q.primary.first_sample, P.seed, wc);
        }
        return HPT_OK;
    };
    // HPT_FLAG_NO_HOST_WAIT enqueues the iterations past eye_depth without knowing whether a path is left: those launches
    // get a grid of 8 workgroups per CU instead of one sized for a full queue (an empty launch then costs a few
    // microseconds instead of ~80), and the kernels walk the chunks of whatever the queue holds with a stride
    const uint32_t blind_groups = (flags & HPT_FLAG_NO_HOST_WAIT) ? (uint32_t) std::max(s->num_cus, 1) * 8u : 0u;
    auto iteration = [&](Pass &q, int it){
        const uint32_t *eq = it == 0 ? nullptr : q.queue[q.cur];
        const PrimaryGen *primary = (it == 0 && in_flight_primaries) ? &q.primary : nullptr;
        const uint32_t cap = it >= eye_depth ? blind_groups : 0u;
        if(legacy){
            LaunchTimer t(s, q.st, timek, 0);
            launch_extend(q.st, s->sd, q.pb, eq, &q.qcnt[it], q.slots, kflags, wc);
        } else {
            // extension rays of this iteration + shadow rays of the previous one, one launch
            TraceSplit split{ q.lqueue[0], &q.lecnt[it], q.lqueue[1], &q.lscnt[it], budget };
            { LaunchTimer t(s, q.st, timek, 0);
              launch_trace(q.st, s->sd, q.pb, q.sb, eq, &q.qcnt[it], q.slots, q.squeue,
                           q.pending_shadow >= 0 ? &q.scnt[q.pending_shadow] : nullptr, q.slots, s->stack_levels, kflags, tuning, wc, &split, primary, cap); }
            if(split.budget > 0){
                LaunchTimer t(s, q.st, timek, 4);
                launch_trace_resume(q.st, s->sd, q.pb, q.sb, true, q.pending_shadow >= 0, q.slots, s->stack_levels, wc, split, primary, cap, q.deep_stack, ((flags >> 19) & 1) != 0, ((flags >> 20) & 1) == 0, (flags >> 21) & 0xFF);
            }
            q.pending_shadow = -1;
        }
        { LaunchTimer t(s, q.st, timek, 1);
          launch_shade(q.st, s->sd, q.pb, eq, &q.qcnt[it], q.slots, q.queue[q.cur ^ 1],
                       &q.qcnt[it + 1], q.sb, q.squeue, &q.scnt[it], eye_depth, P.max_delta, roulette, wc, primary, cap); }
        if(legacy){
            LaunchTimer t(s, q.st, timek, 2);
            launch_connect(q.st, s->sd, q.pb, q.sb, q.squeue, &q.scnt[it], q.slots, kflags, wc);
        } else q.pending_shadow = it;
        q.cur ^= 1;
    };
    // Iterations past eye_depth: only paths that took free delta bounces are still alive (reference src/pt_cu.cu:228),
    // and how many more iterations they need is known on the device only.  The host looks before it launches, every
    // other iteration (an empty launch costs less than a read-back): the counter of each pipeline in flight is read
    // back on that pipeline's stream FIRST, then the host waits for one after the other, so the pipelines keep running
    // side by side while it does.  This is the one place where hpt_render_pt_device blocks the calling thread
    // (include/hpt.h); a scene without delta materials never gets here with a non-empty queue and pays one read-back.
    const bool no_host_wait = (flags & HPT_FLAG_NO_HOST_WAIT) != 0;     // enqueue every tail iteration unseen
    auto tails = [&](int npipes) -> int {
        bool live[kMaxPipes]; int nlive = npipes;
        for(int k = 0; k < kMaxPipes; ++k) live[k] = k < npipes;
        for(int it = eye_depth; it < max_iters && nlive > 0; ++it){
            const bool look = !no_host_wait && ((it - eye_depth) & 1) == 0;
            if(look) for(int k = 0; k < npipes; ++k) if(live[k])
                HIP_TRY(hipMemcpyAsync(pipe[k].h_count, &pipe[k].qcnt[it], sizeof(uint32_t), hipMemcpyDeviceToHost, pipe[k].st));
            for(int k = 0; k < npipes; ++k){
                if(!live[k]) continue;
                if(look){
                    HIP_TRY(hipStreamSynchronize(pipe[k].st));
                    if(*pipe[k].h_count == 0u){ live[k] = false; --nlive; continue; }
                }
                iteration(pipe[k], it);
            }
        }
        for(int k = 0; k < npipes; ++k){
            Pass &q = pipe[k];
            if(q.pending_shadow < 0) continue;
            TraceSplit split{ q.lqueue[0], &q.lecnt[max_iters], q.lqueue[1], &q.lscnt[max_iters], budget };
            { LaunchTimer t(s, q.st, timek, 2);
              launch_trace(q.st, s->sd, q.pb, q.sb, nullptr, nullptr, 0, q.squeue, &q.scnt[q.pending_shadow], q.slots,
                           s->stack_levels, kflags, tuning, wc, &split, nullptr, blind_groups); }
            if(split.budget > 0){
                LaunchTimer t(s, q.st, timek, 4);
                launch_trace_resume(q.st, s->sd, q.pb, q.sb, false, true, q.slots, s->stack_levels, wc, split, nullptr, blind_groups, q.deep_stack, ((flags >> 19) & 1) != 0, ((flags >> 20) & 1) == 0, (flags >> 21) & 0xFF);
            }
        }
        return HPT_OK;
    };

    for(int done = 0; done < spp; done += spass * npipes){
        int active = 0;                                     // pipelines with a pass in this round
        while(active < npipes && done + active * spass < spp) ++active;
        if(active > 1){
            // the other pipelines start after everything already queued on the caller's stream (the previous
            // resolve of their radiance buffers included)
            HIP_TRY(hipEventRecord(s->px_fork, stream));
            for(int k = 1; k < active; ++k) HIP_TRY(hipStreamWaitEvent(s->px_stream[k], s->px_fork, 0));
        }
        for(int k = 0; k < active; ++k){ rc = begin_pass(pipe[k], done + k * spass); if(rc) return rc; }
        for(int it = 0; it < eye_depth && it < max_iters; ++it)
            for(int k = 0; k < active; ++k) iteration(pipe[k], it);
        rc = tails(active); if(rc) return rc;
        // the per-pixel sums are added in sample order: pipeline 0's pass, then the next one's, ...
        for(int k = 0; k < active; ++k){
            if(k > 0){
                HIP_TRY(hipEventRecord(s->px_done[k], s->px_stream[k]));
                HIP_TRY(hipStreamWaitEvent(stream, s->px_done[k], 0));
            }
            LaunchTimer t(s, stream, timek, 3);
            launch_resolve(stream, tl, pipe[k].pb, s->accum, pipe[k].sthis);
            s->last_counters = pipe[k].counters;
        }
    }
    float divisor = (flags & HPT_FLAG_OUTPUT_SUM) ? 1.0f : (float) spp;
    { LaunchTimer t(s, stream, timek, 3);
      launch_finalize(stream, tl, s->accum, d_local, divisor); }
    HIP_TRY(hipEventRecord(s->ev_stop, stream));
    if(auto_budget && s->last_budget > 0 && !s->split_probe_pending){
        int words = 4 * (max_iters + 2);
        if(words > s->h_split_words){
            if(s->h_split) hipHostFree(s->h_split);
            s->h_split = nullptr; s->h_split_words = 0;
            HIP_TRY(hipHostMalloc((void **) &s->h_split, (size_t) words * sizeof(uint32_t)));
            s->h_split_words = words;
        }
        if(!s->ev_split) HIP_TRY(hipEventCreateWithFlags(&s->ev_split, hipEventDisableTiming));
        HIP_TRY(hipMemcpyAsync(s->h_split, s->last_counters, (size_t) words * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipEventRecord(s->ev_split, stream));
        s->split_probe_pending = true; s->split_probe_stride = max_iters + 2;
    }
    HIP_TRY(hipGetLastError());
    s->ev_valid = true;
    s->stats_pending = true;
    s->last_counter_stride = max_iters + 2;
    return HPT_OK;
}

int collect_stats(hpt_scene *s){
    if(!s->stats_pending) return HPT_OK;
    HIP_TRY(hipEventSynchronize(s->ev_stop));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev_start, s->ev_stop));
    s->stats.ms_total = ms;
    double sum[5] = { 0, 0, 0, 0, 0 }; uint32_t cnt[5] = { 0, 0, 0, 0, 0 };
    for(const TimedLaunch &t : s->timed){
        float e = 0.f;
        if(hipEventElapsedTime(&e, t.a, t.b) == hipSuccess){ sum[t.cls] += e; cnt[t.cls]++; }
    }
    s->stats.ms_extend = sum[0]; s->stats.ms_shade = sum[1]; s->stats.ms_connect = sum[2]; s->stats.ms_other = sum[3];
    s->stats.n_extend = cnt[0]; s->stats.n_shade = cnt[1]; s->stats.n_connect = cnt[2]; s->stats.n_other = cnt[3];
    s->stats.ms_resume = sum[4]; s->stats.n_resume = cnt[4];
    s->stats.split_budget = (uint32_t) s->last_budget;
    s->stats.traced_rays_last_pass = s->stats.long_rays_last_pass = 0;
    if(s->last_counter_stride > 0 && s->last_counters){
        std::vector<uint32_t> h((size_t) 4 * s->last_counter_stride);
        HIP_TRY(hipMemcpy(h.data(), s->last_counters, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for(int i = 0; i < s->last_counter_stride; ++i){
            s->stats.traced_rays_last_pass += (uint64_t) h[i] + h[(size_t) s->last_counter_stride + i];
            s->stats.long_rays_last_pass += (uint64_t) h[(size_t) 2 * s->last_counter_stride + i] + h[(size_t) 3 * s->last_counter_stride + i];
        }
    }
    WorkCounters wc;
    HIP_TRY(hipMemcpy(&wc, s->d_wc, sizeof wc, hipMemcpyDeviceToHost));
    s->stats.samples = wc.samples; s->stats.closest_rays = wc.closest_rays; s->stats.shadow_rays = wc.shadow_rays;
    s->stats.boxes_closest = wc.boxes_closest; s->stats.tris_closest = wc.tris_closest;
    s->stats.boxes_shadow = wc.boxes_shadow; s->stats.tris_shadow = wc.tris_shadow; s->stats.path_iters = wc.path_iters;
    s->stats.lane_steps_closest = wc.lane_steps_closest; s->stats.wave_steps_closest = wc.wave_steps_closest;
    s->stats.lane_steps_shadow = wc.lane_steps_shadow; s->stats.wave_steps_shadow = wc.wave_steps_shadow;
    s->stats.leaf_lane_closest = wc.leaf_lane_closest; s->stats.leaf_wave_closest = wc.leaf_wave_closest;
    s->stats.leaf_lane_shadow = wc.leaf_lane_shadow; s->stats.leaf_wave_shadow = wc.leaf_wave_shadow;
    s->stats.bd_pairs = wc.bd_pairs; s->stats.bd_survivors = wc.bd_survivors; s->stats.bd_shadow_rays = wc.bd_shadow_rays;
    s->stats.bd_unoccluded = wc.bd_unoccluded; s->stats.bd_nodes = wc.bd_nodes; s->stats.bd_tris = wc.bd_tris;
    s->stats.bd_spheres = wc.bd_spheres; s->stats.bd_group_boxes = wc.bd_group_boxes;
    s->stats_pending = false;
    return HPT_OK;
}


void free_bdpt_scene(hpt_scene *s){
    hipFree(s->bd_nodes); hipFree(s->bd_tris); hipFree(s->bd_spheres); hipFree(s->bd_groups); hipFree(s->bd_mats); hipFree(s->bd_lights);
    s->bd_nodes = nullptr; s->bd_tris = nullptr; s->bd_spheres = nullptr; s->bd_groups = nullptr; s->bd_mats = nullptr; s->bd_lights = nullptr;
    s->bd_ready = false;
}

void free_bdpt(hpt_scene *s){
    free_bdpt_scene(s);
    hipFree(s->bp.last_pos_pdf); hipFree(s->bp.last_normal); hipFree(s->bp.vtx_pos); hipFree(s->bp.vtx_nrm); hipFree(s->bp.vtx_thr);
    hipFree(s->bp.vtx_wo); hipFree(s->bp.vtx_base); hipFree(s->bp.hist_pos_eta); hipFree(s->bp.hist_pdf); hipFree(s->bp.contrib); hipFree(s->bp.valid);
    hipFree(s->bp.ectx);
    hipFree(s->d_lv); hipFree(s->d_lctx); hipFree(s->cqueue);
    s->bp = BdptPathBuf{}; s->d_lv = nullptr; s->d_lctx = nullptr; s->cqueue = nullptr;
    s->bd_cap_slots = s->bd_cap_hist = s->bd_cap_contrib = s->bd_cap_valid = s->bd_cap_lv = s->bd_cap_lctx = s->bd_cap_cqueue = 0;
}

int ensure_bdpt_scene(hpt_scene *s){
    if(s->bd_ready) return HPT_OK;
    HostBdptScene hb;
    bool grouped = !s->g_kind.empty();
    const char *err = build_bdpt_host_scene(s->h_lights.data(), s->nl, s->h_spheres.data(), s->ns, s->h_tris.data(), s->nt,
                                            grouped ? s->g_kind.data() : nullptr, grouped ? s->g_index.data() : nullptr,
                                            grouped ? s->g_group.data() : nullptr, (int) s->g_kind.size(), hb);
    if(err && *err) return fail(HPT_ERR_INVALID, err);
    free_bdpt_scene(s);
    hipError_t e = upload(hb.nodes, &s->bd_nodes);
    if(e == hipSuccess) e = upload(hb.tris, &s->bd_tris);
    if(e == hipSuccess) e = upload(hb.spheres, &s->bd_spheres);
    if(e == hipSuccess) e = upload(hb.groups, &s->bd_groups);
    if(e == hipSuccess) e = upload(hb.materials, &s->bd_mats);
    if(e == hipSuccess) e = upload(hb.lights, &s->bd_lights);
    if(e != hipSuccess) return fail(HPT_ERR_DEVICE, std::string("bdpt scene upload: ") + hipGetErrorString(e));
    s->bd.nodes = (const float4 *) s->bd_nodes; s->bd.tris = (const float4 *) s->bd_tris; s->bd.spheres = s->bd_spheres;
    s->bd.groups = s->bd_groups; s->bd.mats = s->bd_mats; s->bd.lights = s->bd_lights;
    s->bd.num_groups = (int) hb.groups.size(); s->bd.num_lights = s->nl; s->bd.num_mats = (int) hb.materials.size();
    s->bd.stack_levels = std::min(std::max(hb.bvh_depth, 1) + 1, kStackDepth);
    for(int a = 0; a < 3; ++a){ s->bd.scene_min[a] = hb.scene_min[a]; s->bd.scene_max[a] = hb.scene_max[a]; }
    s->bd_ready = true;
    return HPT_OK;
}

template <typename T>
int grow(T **p, size_t &cap, size_t need){
    if(need <= cap) return HPT_OK;
    hipFree(*p); *p = nullptr; cap = 0;
    HIP_TRY(hipMalloc((void **) p, need * sizeof(T)));
    cap = need;
    return HPT_OK;
}

// the bidirectional render loop (reference src/cpu_bdpt.cpp:173-488), enqueued on `stream`
int render_bdpt_local(hpt_scene *s, const void *camera, int W, int H, int eye_depth, int light_depth, int spp, int spl,
                      const hpt_params *params, float *d_local, hipStream_t stream){
    if(!s) return fail(HPT_ERR_INVALID, "null scene");
    if(!camera || !d_local) return fail(HPT_ERR_INVALID, "null camera or output");
    if(spp <= 0 || spl <= 0 || eye_depth <= 0 || eye_depth > 255 || light_depth <= 0 || light_depth > 255)
        return fail(HPT_ERR_INVALID, "spp, spl must be > 0 and depths in [1, 255]");
    if(int rcd = on_scene_device(s)) return rcd;
    hpt_params P; memset(&P, 0, sizeof P);
    if(params) P = *params;
    if(P.max_delta <= 0) P.max_delta = 64;            // the CPU renderer has no cap (cpu_bdpt.cpp:458)
    if(P.max_delta > 250) P.max_delta = 250;
    Tiling tl;
    int rc = make_tiling(W, H, &P, tl);
    if(rc) return rc;
    rc = ensure_bdpt_scene(s);
    if(rc) return rc;
    s->last_counter_stride = 0; s->last_budget = 0;

    // light vertices: nl * spl subpaths of light_depth vertices each; the contribution table holds one 16-B entry
    // per (path slot, light vertex) pair and at least one image's worth of slots, so it is bounded here
    const long long n_lv64 = (long long) s->nl * spl * light_depth;
    if(n_lv64 > (1ll << 24)) return fail(HPT_ERR_INVALID, "too many light vertices (num_lights * spl * light_depth > 2^24)");
    const int total_light_paths = s->nl * spl;
    const int n_lv = (int) n_lv64;
    // slots per pass: bound the contribution table (16 B per pair) to about 1 GiB
    int spass = P.samples_per_pass;
    if(spass <= 0){
        long long pairs = 64ll << 20;
        long long slots = std::max<long long>(tl.n_local, std::min<long long>(4ll << 20, pairs / std::max(n_lv, 1)));
        spass = (int) std::max<long long>(1, slots / tl.n_local);
    }
    spass = std::min(spass, spp);
    size_t slots = (size_t) tl.n_local * spass;
    if(slots > 0x7FFFFFF0ull) return fail(HPT_ERR_INVALID, "too many path slots per pass");
    if((double) slots * (double) std::max(n_lv, 1) * 16.0 > 64.0 * 1073741824.0)
        return fail(HPT_ERR_INVALID, "contribution table (path slots x light vertices x 16 B) would exceed 64 GiB: lower spl, light_depth, "
                                     "the image size per rank or samples_per_pass");
    const int max_iters = eye_depth + P.max_delta + 1;
    int n_counters = 2 * (max_iters + 2);
    rc = ensure_workspace(s, slots, tl.n_local, n_counters);
    if(rc) return rc;
    if(slots > s->bd_cap_slots){
        size_t c;
        c = s->bd_cap_slots; rc = grow(&s->bp.last_pos_pdf, c, slots); if(rc) return rc;
        c = s->bd_cap_slots; rc = grow(&s->bp.last_normal, c, slots); if(rc) return rc;
        c = s->bd_cap_slots; rc = grow(&s->bp.vtx_pos, c, slots); if(rc) return rc;
        c = s->bd_cap_slots; rc = grow(&s->bp.vtx_nrm, c, slots); if(rc) return rc;
        c = s->bd_cap_slots; rc = grow(&s->bp.vtx_thr, c, slots); if(rc) return rc;
        c = s->bd_cap_slots; rc = grow(&s->bp.vtx_wo, c, slots); if(rc) return rc;
        c = s->bd_cap_slots; rc = grow(&s->bp.vtx_base, c, slots); if(rc) return rc;
        c = s->bd_cap_slots * 7; rc = grow(&s->bp.ectx, c, slots * 7); if(rc) return rc;
        s->bd_cap_slots = slots;
        s->bd_cap_hist = 0; s->bd_cap_contrib = 0; s->bd_cap_valid = 0;
    }
    rc = grow(&s->cqueue, s->bd_cap_cqueue, slots); if(rc) return rc;
    { size_t need = slots * (size_t) eye_depth;
      if(need > s->bd_cap_hist){
          size_t c1 = s->bd_cap_hist, c2 = s->bd_cap_hist;
          rc = grow(&s->bp.hist_pos_eta, c1, need); if(rc) return rc;
          rc = grow(&s->bp.hist_pdf, c2, need); if(rc) return rc;
          s->bd_cap_hist = need;
      } }
    rc = grow(&s->bp.contrib, s->bd_cap_contrib, slots * (size_t) std::max(n_lv, 1)); if(rc) return rc;
    rc = grow(&s->bp.valid, s->bd_cap_valid, slots * (size_t) ((std::max(n_lv, 1) + 63) / 64)); if(rc) return rc;
    rc = grow(&s->d_lv, s->bd_cap_lv, (size_t) std::max(n_lv, 1)); if(rc) return rc;
    rc = grow(&s->d_lctx, s->bd_cap_lctx, (size_t) std::max(n_lv, 1)); if(rc) return rc;

    const float *cf = (const float *) camera;
    CameraDev cam;
    memcpy(cam.eye, cf + 0, 12); memcpy(cam.UL, cf + 12, 12); memcpy(cam.dx, cf + 15, 12); memcpy(cam.dy, cf + 18, 12);
    s->timed.clear(); s->event_next = 0; s->last_flags = P.flags;
    s->stats.ms_total = s->stats.ms_extend = s->stats.ms_shade = s->stats.ms_connect = s->stats.ms_other = 0.0;
    s->stats.n_extend = s->stats.n_shade = s->stats.n_connect = s->stats.n_other = 0;
    const bool timek = (P.flags & HPT_FLAG_TIME_KERNELS) != 0;

    HIP_TRY(hipMemsetAsync(s->d_wc, 0, sizeof(WorkCounters), stream));
    HIP_TRY(hipMemsetAsync(s->accum, 0, (size_t) tl.n_local * sizeof(float4), stream));
    HIP_TRY(hipEventRecord(s->ev_start, stream));
    if(s->nl > 0){                                       // no lights: the CPU renderer returns at once (cpu_bdpt.cpp:178)
        { LaunchTimer t(s, stream, timek, 3);
          launch_bdpt_light_trace(stream, s->bd, s->d_lv, total_light_paths, light_depth, spl, P.seed, P.max_delta);
          launch_bdpt_light_ctx(stream, s->d_lv, s->d_lctx, n_lv, light_depth); }
        for(int done = 0; done < spp; done += spass){
            int sthis = std::min(spass, spp - done);
            uint32_t nslots = (uint32_t) tl.n_local * (uint32_t) sthis;
            HIP_TRY(hipMemsetAsync(s->pass[0].counters, 0, (size_t) n_counters * sizeof(uint32_t), stream));
            uint32_t *qcnt = s->pass[0].counters, *ccnt = s->pass[0].counters + (max_iters + 2);
            { LaunchTimer t(s, stream, timek, 3);
              launch_bdpt_generate(stream, tl, cam, s->pass[0].pb, s->bp, &qcnt[0], sthis, (uint32_t) (P.sample_offset + done), P.seed); }
            int cur = 0;
            for(int it = 0; it < max_iters; ++it){
                const int ci = it;                             // counter slot of this iteration
                // past eye_depth only paths on free delta bounces are alive: the host looks every other iteration (an
                // iteration on an empty queue costs four empty launches), or never with HPT_FLAG_NO_HOST_WAIT
                if(it >= eye_depth && !(P.flags & HPT_FLAG_NO_HOST_WAIT) && ((it - eye_depth) & 1) == 0){
                    HIP_TRY(hipMemcpyAsync(s->pass[0].h_count, &qcnt[ci], sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                    HIP_TRY(hipStreamSynchronize(stream));
                    if(*s->pass[0].h_count == 0u) break;
                }
                const uint32_t *eq = it == 0 ? nullptr : s->pass[0].queue[cur];
                // unseen tail iterations: small fixed grids for the three kernels that walk their queue with a stride
                // (k_bdpt_vertex keeps its grid: one chunk per workgroup, an empty one returns at once)
                const uint32_t cap = (it >= eye_depth && (P.flags & HPT_FLAG_NO_HOST_WAIT)) ? (uint32_t) std::max(s->num_cus, 1) * 8u : 0u;
                { LaunchTimer t(s, stream, timek, 0);
                  launch_bdpt_extend(stream, s->bd, s->pass[0].pb, eq, &qcnt[ci], nslots, cap); }
                { LaunchTimer t(s, stream, timek, 1);
                  launch_bdpt_vertex(stream, s->bd, s->pass[0].pb, s->bp, eq, &qcnt[ci], nslots, s->pass[0].queue[cur ^ 1], &qcnt[ci + 1],
                                     s->cqueue, &ccnt[ci], eye_depth, P.max_delta, (uint32_t) slots, cam.eye); }
                { LaunchTimer t(s, stream, timek, 2);
                  launch_bdpt_connect(stream, s->bd, s->pass[0].pb, s->bp, s->d_lv, s->d_lctx, n_lv, light_depth, s->cqueue, &ccnt[ci], nslots,
                                      (uint32_t) slots, cap, (P.flags & HPT_FLAG_COUNT_WORK) ? s->d_wc : nullptr); }
                { LaunchTimer t(s, stream, timek, 3);
                  launch_bdpt_reduce(stream, s->pass[0].pb, s->bp, n_lv, s->cqueue, &ccnt[ci], nslots, cap); }
                cur ^= 1;
            }
            { LaunchTimer t(s, stream, timek, 3);
              launch_resolve(stream, tl, s->pass[0].pb, s->accum, sthis); }
        }
    }
    float divisor = (P.flags & HPT_FLAG_OUTPUT_SUM) ? 1.0f : (float) spp;
    { LaunchTimer t(s, stream, timek, 3);
      launch_finalize(stream, tl, s->accum, d_local, divisor); }
    HIP_TRY(hipEventRecord(s->ev_stop, stream));
    HIP_TRY(hipGetLastError());
    s->ev_valid = true; s->stats_pending = true;
    return HPT_OK;
}

} // namespace

namespace {

// what the one-shot wrappers keep between calls (include/hpt.h, hpt_pt_render_wrapper): the scene of the current
// device, or -- when more than one device is configured (hpt_wrapper_set_devices / HPT_DEVICES) -- the fan-out
struct WrapperCache { std::mutex mu; hpt_scene *scene = nullptr; int device = -1; hpt_multi *multi = nullptr; int devices = 0; } g_wrap;

bool wrapper_cache_enabled(){
    static const bool on = [](){ const char *e = getenv("HPT_WRAPPER_CACHE"); return !(e && e[0] == '0'); }();
    return on;
}

// number of devices the one-shot wrappers render on: hpt_wrapper_set_devices(), else HPT_DEVICES, else 1
int wrapper_devices(){
    if(g_wrap.devices > 0) return g_wrap.devices;
    static const int from_env = [](){ const char *e = getenv("HPT_DEVICES"); int v = e ? atoi(e) : 1; return v > 0 ? v : 1; }();
    return from_env;
}

bool same_bytes(const std::vector<unsigned char> &kept, const void *given, size_t bytes){
    return kept.size() == bytes && (bytes == 0 || memcmp(kept.data(), given, bytes) == 0);
}

// g_wrap.mu held.  The kept scene when the arrays are byte-identical to the ones it was built from, else a new one.
int wrapper_scene(const void *lights, int nl, const void *spheres, int ns, const void *tris, int nt, hpt_scene **out){
    int dev = -1;
    if(hipGetDevice(&dev) != hipSuccess) dev = -1;
    hpt_scene *k = g_wrap.scene;
    if(k && wrapper_cache_enabled() && dev == g_wrap.device && nl == k->nl && ns == k->ns && nt == k->nt && nl >= 0 && ns >= 0 && nt >= 0 &&
       same_bytes(k->h_lights, lights, (size_t) nl * HPT_LIGHT_BYTES) && same_bytes(k->h_spheres, spheres, (size_t) ns * HPT_SPHERE_BYTES) &&
       same_bytes(k->h_tris, tris, (size_t) nt * HPT_TRIANGLE_BYTES)){
        *out = k;
        return HPT_OK;
    }
    if(k){ hpt_scene_destroy(k); g_wrap.scene = nullptr; g_wrap.device = -1; }
    int rc = hpt_scene_create(lights, nl, spheres, ns, tris, nt, out);
    if(rc) return rc;
    if(wrapper_cache_enabled()){ g_wrap.scene = *out; g_wrap.device = dev; }
    return HPT_OK;
}

// g_wrap.mu held: a scene that is not kept is destroyed (the error text of the render survives)
void wrapper_release(hpt_scene *s){
    if(s == g_wrap.scene) return;
    std::string keep = g_err;
    hpt_scene_destroy(s);
    g_err = keep;
}

// g_wrap.mu held: the kept fan-out over `devices` devices for these arrays, else a new one
int wrapper_multi(int devices, const void *lights, int nl, const void *spheres, int ns, const void *tris, int nt, hpt_multi **out){
    if(g_wrap.multi && wrapper_cache_enabled() && hpt::multi_matches(g_wrap.multi, devices, lights, nl, spheres, ns, tris, nt)){
        *out = g_wrap.multi;
        return HPT_OK;
    }
    if(g_wrap.multi){ hpt_multi_destroy(g_wrap.multi); g_wrap.multi = nullptr; }
    int rc = hpt_multi_create(lights, nl, spheres, ns, tris, nt, nullptr, devices, 0, out);
    if(rc) return rc;
    if(wrapper_cache_enabled()) g_wrap.multi = *out;
    return HPT_OK;
}

void wrapper_release_multi(hpt_multi *m){
    if(m == g_wrap.multi) return;
    std::string keep = g_err;
    hpt_multi_destroy(m);
    g_err = keep;
}

} // namespace

namespace hpt {

int fail_with(int code, const std::string &msg){ return fail(code, msg); }     // for hpt_multi.cpp

// Uploads a flattened scene to the current device (hpt_scene_create = build_host_scene + this; the multi-device
// fan-out builds once and uploads to every device).
int scene_upload(const HostScene &hs, const void *lights, int nl, const void *spheres, int ns, const void *tris, int nt,
                 hpt_scene **out){
    *out = nullptr;
    hpt_scene *s = new (std::nothrow) hpt_scene();
    if(!s) return fail(HPT_ERR_NOMEM, "out of host memory");
    auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipGetDevice(&s->device);
    if(e == hipSuccess){ hipDeviceProp_t prop; if(hipGetDeviceProperties(&prop, s->device) == hipSuccess && prop.multiProcessorCount > 0) s->num_cus = prop.multiProcessorCount; }
    if(e == hipSuccess) e = upload(hs.nodes, &s->d_nodes);
    if(e == hipSuccess) e = upload(hs.qnodes, &s->d_qnodes);
    if(e == hipSuccess) e = upload(hs.wnodes, &s->d_wnodes);
    if(e == hipSuccess) e = upload(hs.tris, &s->d_tris);
    if(e == hipSuccess) e = upload(hs.rounds, &s->d_rounds);
    if(e == hipSuccess) e = upload(hs.materials, &s->d_mats);
    if(e == hipSuccess) e = upload(hs.lights, &s->d_lights);
    if(e != hipSuccess){
        std::string msg = std::string("scene upload: ") + hipGetErrorString(e);
        hpt_scene_destroy(s);
        return fail(e == hipErrorOutOfMemory ? HPT_ERR_NOMEM : HPT_ERR_DEVICE, msg);
    }
    e = hipMalloc((void **) &s->d_tri_frames, std::max<size_t>((size_t) nt, 1) * 4 * sizeof(float4));
    if(e == hipSuccess){
        launch_tri_frames(nullptr, (const float4 *) s->d_tris, nt, s->d_tri_frames);
        e = hipDeviceSynchronize();
    }
    if(e != hipSuccess){
        std::string msg = std::string("triangle frames: ") + hipGetErrorString(e);
        hpt_scene_destroy(s);
        return fail(e == hipErrorOutOfMemory ? HPT_ERR_NOMEM : HPT_ERR_DEVICE, msg);
    }
    auto t1 = std::chrono::steady_clock::now();
    s->sd.nodes = (const float4 *) s->d_nodes; s->sd.tris = (const float4 *) s->d_tris;
    s->sd.tri_frames = s->d_tri_frames;
    s->sd.qnodes = (const uint4 *) s->d_qnodes;
    s->sd.wnodes = (const uint4 *) s->d_wnodes; s->sd.wide_depth = hs.wide_depth;
    for(int a = 0; a < 3; ++a){ s->sd.qorigin[a] = hs.qorigin[a]; s->sd.qscale[a] = hs.qscale[a]; }
    s->sd.rounds = s->d_rounds; s->sd.mats = s->d_mats; s->sd.lights = s->d_lights;
    s->sd.num_rounds = ns + nl; s->sd.num_spheres = ns; s->sd.num_lights = nl; s->sd.num_tris = nt;
    s->sd.num_mats = (int) hs.materials.size(); s->sd.num_nodes = (int) hs.qnodes.size();
    s->nl = nl; s->ns = ns; s->nt = nt;
    if(nl) s->h_lights.assign((const unsigned char *) lights, (const unsigned char *) lights + (size_t) nl * HPT_LIGHT_BYTES);
    if(ns) s->h_spheres.assign((const unsigned char *) spheres, (const unsigned char *) spheres + (size_t) ns * HPT_SPHERE_BYTES);
    if(nt) s->h_tris.assign((const unsigned char *) tris, (const unsigned char *) tris + (size_t) nt * HPT_TRIANGLE_BYTES);
    memset(&s->stats, 0, sizeof s->stats);
    s->stats.bvh_nodes = (uint32_t) hs.nodes.size(); s->stats.bvh_depth = (uint32_t) hs.bvh_depth;
    s->stack_levels = hs.bvh_depth > 0 ? hs.bvh_depth : 1;     // a leaf at depth d has d inner ancestors: at most d pushes
    s->stats.n_tris = (uint32_t) nt; s->stats.n_materials = (uint32_t) hs.materials.size();
    s->stats.ms_bvh_build = hs.ms_bvh_build;
    s->stats.ms_upload = std::chrono::duration<double, std::milli>(t1 - t0).count();
    *out = s;
    return HPT_OK;
}

} // namespace hpt

extern "C" {

const char *hpt_last_error(void){ return g_err.c_str(); }

int hpt_device_count(void){
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

int hpt_scene_create(const void *lights, int nl, const void *spheres, int ns, const void *tris, int nt,
                     hpt_scene **out){
    if(!out) return fail(HPT_ERR_INVALID, "null out_scene");
    *out = nullptr;
    HostScene hs;
    const char *err = build_host_scene(lights, nl, spheres, ns, tris, nt, hs);
    if(err && *err) return fail(HPT_ERR_INVALID, err);
    return hpt::scene_upload(hs, lights, nl, spheres, ns, tris, nt, out);
}

void hpt_scene_destroy(hpt_scene *s){
    if(!s) return;
    for(PassBuffers &w : s->pass){
        free_pass(w);
        hipFree(w.counters);
        if(w.h_count) hipHostFree(w.h_count);
    }
    hipFree(s->accum); hipFree(s->d_wc);
    if(s->px_fork) hipEventDestroy(s->px_fork);
    for(int k = 1; k < kMaxPipes; ++k){
        if(s->px_done[k]) hipEventDestroy(s->px_done[k]);
        if(s->px_stream[k]) hipStreamDestroy(s->px_stream[k]);
    }
    if(s->h_split) hipHostFree(s->h_split);
    if(s->ev_split) hipEventDestroy(s->ev_split);
    hipFree(s->d_local_own); hipFree(s->d_image_own);
    hipFree(s->d_nodes); hipFree(s->d_qnodes); hipFree(s->d_wnodes); hipFree(s->d_tris); hipFree(s->d_rounds); hipFree(s->d_mats); hipFree(s->d_lights);
    hipFree(s->d_tri_frames);
    free_bdpt(s);
    if(s->ev_start) hipEventDestroy(s->ev_start);
    if(s->ev_stop) hipEventDestroy(s->ev_stop);
    for(hipEvent_t e : s->event_pool) hipEventDestroy(e);
    delete s;
}

int64_t hpt_local_pixels(int W, int H, const hpt_params *params){
    Tiling tl;
    if(make_tiling(W, H, params, tl)) return -1;
    return tl.n_local;
}

int hpt_render_pt_device(hpt_scene *scene, const void *camera, int W, int H, int eye_depth, int spp,
                         const hpt_params *params, void *d_local, void *hip_stream){
    return render_local(scene, camera, W, H, eye_depth, spp, params, (float *) d_local, (hipStream_t) hip_stream);
}

int hpt_untile(const void *d_gathered, void *d_image, int W, int H, const hpt_params *params, void *hip_stream){
    Tiling tl;
    int rc = make_tiling(W, H, params, tl);
    if(rc) return rc;
    if(!d_gathered || !d_image) return fail(HPT_ERR_INVALID, "null buffer");
    launch_untile((hipStream_t) hip_stream, tl, (const float *) d_gathered, (float *) d_image);
    HIP_TRY(hipGetLastError());
    return HPT_OK;
}

int hpt_render_pt(hpt_scene *s, const void *camera, int W, int H, int eye_depth, int spp,
                  const hpt_params *params, float *host_image){
    if(!s) return fail(HPT_ERR_INVALID, "null scene");
    if(!host_image) return fail(HPT_ERR_INVALID, "null image");
    if(params && params->world > 1) return fail(HPT_ERR_INVALID, "hpt_render_pt renders the whole image: world must be 0 or 1");
    Tiling tl;
    int rc = make_tiling(W, H, params, tl);
    if(rc) return rc;
    size_t nloc = (size_t) tl.n_local * 3, nimg = (size_t) W * H * 3;
    if(nloc > s->cap_local_own){
        hipFree(s->d_local_own); s->d_local_own = nullptr; s->cap_local_own = 0;
        HIP_TRY(hipMalloc((void **) &s->d_local_own, nloc * sizeof(float)));
        s->cap_local_own = nloc;
    }
    if(nimg > s->cap_image_own){
        hipFree(s->d_image_own); s->d_image_own = nullptr; s->cap_image_own = 0;
        HIP_TRY(hipMalloc((void **) &s->d_image_own, nimg * sizeof(float)));
        s->cap_image_own = nimg;
    }
    hipStream_t st = nullptr;
    rc = render_local(s, camera, W, H, eye_depth, spp, params, s->d_local_own, st);
    if(rc) return rc;
    launch_untile(st, tl, s->d_local_own, s->d_image_own);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(host_image, s->d_image_own, nimg * sizeof(float), hipMemcpyDeviceToHost));
    return HPT_OK;
}

void hpt_wrapper_cache_clear(void){
    std::lock_guard<std::mutex> lock(g_wrap.mu);
    if(g_wrap.scene) hpt_scene_destroy(g_wrap.scene);
    g_wrap.scene = nullptr; g_wrap.device = -1;
    if(g_wrap.multi) hpt_multi_destroy(g_wrap.multi);
    g_wrap.multi = nullptr;
}

int hpt_wrapper_set_devices(int num_devices){
    if(num_devices < 0) return fail(HPT_ERR_INVALID, "negative device count");
    std::lock_guard<std::mutex> lock(g_wrap.mu);
    g_wrap.devices = num_devices;
    return HPT_OK;
}

int hpt_pt_render_wrapper(const void *lights, int nl, const void *spheres, int ns, const void *tris, int nt,
                          const float scene_min[3], const float scene_max[3], const void *camera, float *host_image,
                          int W, int H, int light_depth, int light_sample, int eye_depth, int spp, int64_t seed){
    (void) scene_min; (void) scene_max; (void) light_depth; (void) light_sample;   // ignored by the reference too
    std::lock_guard<std::mutex> lock(g_wrap.mu);
    if(wrapper_devices() > 1){
        // the blocking call fans out over the node's devices internally (image tiles, RCCL gather): hpt_multi.cpp
        hpt_multi *m = nullptr;
        int rc = wrapper_multi(wrapper_devices(), lights, nl, spheres, ns, tris, nt, &m);
        if(rc) return rc;
        hpt_params p; memset(&p, 0, sizeof p);
        p.seed = seed >= 0 ? (uint64_t) seed : (uint64_t) time(nullptr);
        rc = hpt_multi_render_pt(m, camera, W, H, eye_depth, spp, &p, host_image);
        wrapper_release_multi(m);
        return rc;
    }
    hpt_scene *s = nullptr;
    int rc = wrapper_scene(lights, nl, spheres, ns, tris, nt, &s);
    if(rc) return rc;
    hpt_params p; memset(&p, 0, sizeof p);
    p.seed = seed >= 0 ? (uint64_t) seed : (uint64_t) time(nullptr);      // reference: time(NULL), pt_cu.cu:282
    rc = hpt_render_pt(s, camera, W, H, eye_depth, spp, &p, host_image);
    wrapper_release(s);
    return rc;
}

// ---- 8-bit output stage ------------------------------------------------------------------------------------------
namespace {

// what the reference's host loop computes per channel (src/main_cli.cpp:233-241)
unsigned char tonemap_byte(float x){
    float c = std::max(0.0f, std::min(x, 1.0f));
    float g = std::pow(c, 1.0f / 2.2f);
    return (unsigned char) (g * 255.0f);
}

// thr[k], k = 1..255: the smallest float in [0, 1] whose byte is >= k (byte(x) is non-decreasing in x, so a binary
// search over the bit patterns of the non-negative floats finds it); thr[0] = -inf
const float *tonemap_thresholds(){
    static float table[256];
    static std::once_flag once;
    std::call_once(once, [](){
        table[0] = -INFINITY;
        for(int k = 1; k < 256; ++k){
            uint32_t lo = 0u, hi = 0x3F800000u;               // bits of 0.0f .. 1.0f; byte(1.0f) = 255 >= k
            while(lo < hi){
                uint32_t mid = lo + (hi - lo) / 2u;
                float x; memcpy(&x, &mid, 4);
                if((int) tonemap_byte(x) >= k) hi = mid; else lo = mid + 1u;
            }
            memcpy(&table[k], &lo, 4);
        }
    });
    return table;
}

struct TonemapDevice { std::mutex mu; float *d_table[64] = {}; } g_tonemap;

// the threshold table on the current device (uploaded once per device)
int tonemap_device_table(const float **out){
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if(dev < 0 || dev >= 64) return fail(HPT_ERR_INVALID, "device ordinal out of range");
    std::lock_guard<std::mutex> lock(g_tonemap.mu);
    if(!g_tonemap.d_table[dev]){
        float *d = nullptr;
        HIP_TRY(hipMalloc((void **) &d, 256 * sizeof(float)));
        hipError_t e = hipMemcpy(d, tonemap_thresholds(), 256 * sizeof(float), hipMemcpyHostToDevice);
        if(e != hipSuccess){ hipFree(d); return fail(HPT_ERR_DEVICE, std::string("tonemap table upload: ") + hipGetErrorString(e)); }
        g_tonemap.d_table[dev] = d;
    }
    *out = g_tonemap.d_table[dev];
    return HPT_OK;
}

} // namespace

void hpt_tonemap_table(float thresholds_out[256]){ memcpy(thresholds_out, tonemap_thresholds(), 256 * sizeof(float)); }

void hpt_tonemap_reference(const float *linear_rgb, unsigned char *rgb8, int64_t num_pixels, int bgr){
    for(int64_t p = 0; p < num_pixels; ++p)
        for(int c = 0; c < 3; ++c) rgb8[p * 3 + c] = tonemap_byte(linear_rgb[p * 3 + (bgr ? 2 - c : c)]);
}

int hpt_tonemap(const void *d_linear_rgb, void *d_rgb8, int64_t num_pixels, int bgr, void *hip_stream){
    if(num_pixels < 0 || (num_pixels > 0 && (!d_linear_rgb || !d_rgb8))) return fail(HPT_ERR_INVALID, "bad tonemap argument");
    if(((uintptr_t) d_rgb8 & 3u) != 0u) return fail(HPT_ERR_INVALID, "tonemap output must be 4-byte aligned");
    const float *table = nullptr;
    int rc = tonemap_device_table(&table);
    if(rc) return rc;
    launch_tonemap((hipStream_t) hip_stream, (const float *) d_linear_rgb, d_rgb8, (unsigned long long) num_pixels * 3ull, bgr ? 1 : 0, table);
    HIP_TRY(hipGetLastError());
    return HPT_OK;
}

int hpt_tonemap_host(const float *linear_rgb, unsigned char *rgb8, int64_t num_pixels, int bgr){
    if(num_pixels < 0 || (num_pixels > 0 && (!linear_rgb || !rgb8))) return fail(HPT_ERR_INVALID, "bad tonemap argument");
    if(num_pixels == 0) return HPT_OK;
    DevBuf d_in, d_out;
    size_t n = (size_t) num_pixels * 3;
    HIP_TRY(d_in.alloc(n * sizeof(float)));
    HIP_TRY(d_out.alloc((n + 3) / 4 * 4));
    HIP_TRY(hipMemcpy(d_in.p, linear_rgb, n * sizeof(float), hipMemcpyHostToDevice));
    int rc = hpt_tonemap(d_in.p, d_out.p, num_pixels, bgr, nullptr);
    if(rc) return rc;
    HIP_TRY(hipMemcpy(rgb8, d_out.p, n, hipMemcpyDeviceToHost));
    return HPT_OK;
}

int hpt_get_stats(const hpt_scene *scene, hpt_stats *out){
    if(!scene || !out) return fail(HPT_ERR_INVALID, "null argument");
    int rc = collect_stats(const_cast<hpt_scene *>(scene));
    if(rc) return rc;
    *out = scene->stats;
    return HPT_OK;
}

int hpt_trace_closest(hpt_scene *s, const float *origins, const float *dirs, int n, int flags,
                      float *t_out, int32_t *prim_out){
    if(!s || !origins || !dirs || !t_out || !prim_out || n < 0) return fail(HPT_ERR_INVALID, "bad argument");
    if(n == 0) return HPT_OK;
    DevBuf d_o, d_d, d_t, d_p;
    size_t b3 = (size_t) n * 3 * sizeof(float);
    HIP_TRY(d_o.alloc(b3)); HIP_TRY(d_d.alloc(b3));
    HIP_TRY(d_t.alloc((size_t) n * 4)); HIP_TRY(d_p.alloc((size_t) n * 4));
    HIP_TRY(hipMemcpy(d_o.p, origins, b3, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_d.p, dirs, b3, hipMemcpyHostToDevice));
    launch_probe_closest(nullptr, s->sd, d_o.as<float>(), d_d.as<float>(), n, (flags & HPT_FLAG_BRUTE_FORCE) ? 1 : 0, d_t.as<float>(), d_p.as<int32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(t_out, d_t.p, (size_t) n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(prim_out, d_p.p, (size_t) n * 4, hipMemcpyDeviceToHost));
    return HPT_OK;
}

int hpt_probe_functions(const float *records_in, int n, float *results_out){
    if(n < 0 || (n > 0 && (!records_in || !results_out))) return fail(HPT_ERR_INVALID, "bad argument");
    if(n == 0) return HPT_OK;
    DevBuf d_in, d_out;
    HIP_TRY(d_in.alloc((size_t) n * 24 * sizeof(float)));
    HIP_TRY(d_out.alloc((size_t) n * 40 * sizeof(float)));
    HIP_TRY(hipMemcpy(d_in.p, records_in, (size_t) n * 24 * sizeof(float), hipMemcpyHostToDevice));
    launch_probe_functions(nullptr, d_in.as<float>(), n, d_out.as<float>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(results_out, d_out.p, (size_t) n * 40 * sizeof(float), hipMemcpyDeviceToHost));
    return HPT_OK;
}

int hpt_scene_set_groups(hpt_scene *s, const int32_t *obj_kind, const int32_t *obj_index, const int32_t *obj_group, int nobj){
    if(!s) return fail(HPT_ERR_INVALID, "null scene");
    if(nobj < 0 || (nobj > 0 && (!obj_kind || !obj_index || !obj_group))) return fail(HPT_ERR_INVALID, "bad group arrays");
    if(nobj != 0 && nobj != s->ns + s->nt) return fail(HPT_ERR_INVALID, "group arrays must list every sphere and triangle once");
    s->g_kind.assign(obj_kind, obj_kind + nobj); s->g_index.assign(obj_index, obj_index + nobj); s->g_group.assign(obj_group, obj_group + nobj);
    s->bd_ready = false;
    return HPT_OK;
}

int hpt_render_bdpt_device(hpt_scene *scene, const void *camera, int W, int H, int eye_depth, int light_depth, int spp, int spl,
                           const hpt_params *params, void *d_local, void *hip_stream){
    return render_bdpt_local(scene, camera, W, H, eye_depth, light_depth, spp, spl, params, (float *) d_local, (hipStream_t) hip_stream);
}

int hpt_render_bdpt(hpt_scene *s, const void *camera, int W, int H, int eye_depth, int light_depth, int spp, int spl,
                    const hpt_params *params, float *host_image){
    if(!s) return fail(HPT_ERR_INVALID, "null scene");
    if(!host_image) return fail(HPT_ERR_INVALID, "null image");
    if(params && params->world > 1) return fail(HPT_ERR_INVALID, "hpt_render_bdpt renders the whole image: world must be 0 or 1");
    Tiling tl;
    int rc = make_tiling(W, H, params, tl);
    if(rc) return rc;
    size_t nloc = (size_t) tl.n_local * 3, nimg = (size_t) W * H * 3;
    if(nloc > s->cap_local_own){
        hipFree(s->d_local_own); s->d_local_own = nullptr; s->cap_local_own = 0;
        HIP_TRY(hipMalloc((void **) &s->d_local_own, nloc * sizeof(float)));
        s->cap_local_own = nloc;
    }
    if(nimg > s->cap_image_own){
        hipFree(s->d_image_own); s->d_image_own = nullptr; s->cap_image_own = 0;
        HIP_TRY(hipMalloc((void **) &s->d_image_own, nimg * sizeof(float)));
        s->cap_image_own = nimg;
    }
    hipStream_t st = nullptr;
    rc = render_bdpt_local(s, camera, W, H, eye_depth, light_depth, spp, spl, params, s->d_local_own, st);
    if(rc) return rc;
    launch_untile(st, tl, s->d_local_own, s->d_image_own);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(host_image, s->d_image_own, nimg * sizeof(float), hipMemcpyDeviceToHost));
    return HPT_OK;
}

int hpt_bdpt_render_wrapper(const void *lights, int nl, const void *spheres, int ns, const void *tris, int nt,
                            const float scene_min[3], const float scene_max[3], const void *camera, float *host_image,
                            int W, int H, int light_depth, int light_sample, int eye_depth, int spp, int spl, int64_t seed){
    (void) scene_min; (void) scene_max;
    // The reference's helper hands over illum / light_sample (src/bdpt_cu_helper.cpp:60-62); the
    // estimator built here is cpu_bdpt's, which divides by spl itself, so that pre-division is undone.
    std::vector<unsigned char> L((const unsigned char *) lights, (const unsigned char *) lights + (size_t) std::max(nl, 0) * HPT_LIGHT_BYTES);
    if(light_sample > 1) for(int i = 0; i < nl; ++i){
        float *illum = (float *) (L.data() + (size_t) i * HPT_LIGHT_BYTES + 24);
        for(int c = 0; c < 3; ++c) illum[c] = illum[c] * (float) light_sample;
    }
    std::lock_guard<std::mutex> lock(g_wrap.mu);
    if(wrapper_devices() > 1){
        hpt_multi *m = nullptr;
        int rc = wrapper_multi(wrapper_devices(), L.data(), nl, spheres, ns, tris, nt, &m);
        if(rc) return rc;
        hpt_params p; memset(&p, 0, sizeof p);
        p.seed = seed >= 0 ? (uint64_t) seed : (uint64_t) time(nullptr);
        rc = hpt_multi_render_bdpt(m, camera, W, H, eye_depth, light_depth, spp, spl, &p, host_image);
        wrapper_release_multi(m);
        return rc;
    }
    hpt_scene *s = nullptr;
    int rc = wrapper_scene(L.data(), nl, spheres, ns, tris, nt, &s);
    if(rc) return rc;
    hpt_params p; memset(&p, 0, sizeof p);
    p.seed = seed >= 0 ? (uint64_t) seed : (uint64_t) time(nullptr);
    rc = hpt_render_bdpt(s, camera, W, H, eye_depth, light_depth, spp, spl, &p, host_image);
    wrapper_release(s);
    return rc;
}

int hpt_bvh_export_host(const void *lights, int nl, const void *spheres, int ns, const void *tris, int nt, hpt_bvh_info *info,
                        void *qnodes_out, size_t qnodes_cap, void *tris_out, size_t tris_cap){
    if(!info) return fail(HPT_ERR_INVALID, "null info");
    HostScene hs;
    const char *err = build_host_scene(lights, nl, spheres, ns, tris, nt, hs);
    if(err && *err) return fail(HPT_ERR_INVALID, err);
    info->num_nodes = (int32_t) hs.qnodes.size(); info->num_tris = (int32_t) hs.tris.size();
    info->bvh_depth = hs.bvh_depth; info->num_rounds = ns + nl;
    for(int a = 0; a < 3; ++a){ info->qorigin[a] = hs.qorigin[a]; info->qscale[a] = hs.qscale[a]; }
    if(qnodes_out){
        if(qnodes_cap < hs.qnodes.size() * sizeof(QBvhNode)) return fail(HPT_ERR_INVALID, "qnodes_out too small");
        memcpy(qnodes_out, hs.qnodes.data(), hs.qnodes.size() * sizeof(QBvhNode));
    }
    if(tris_out){
        if(tris_cap < hs.tris.size() * sizeof(DevTriangle)) return fail(HPT_ERR_INVALID, "tris_out too small");
        memcpy(tris_out, hs.tris.data(), hs.tris.size() * sizeof(DevTriangle));
    }
    return HPT_OK;
}

int hpt_scene_export_bvh(const hpt_scene *s, hpt_bvh_info *info, void *qnodes_out, size_t qnodes_cap, void *tris_out, size_t tris_cap){
    if(!s || !info) return fail(HPT_ERR_INVALID, "null argument");
    if(int rcd = on_scene_device(s)) return rcd;
    info->num_nodes = s->sd.num_nodes; info->num_tris = s->sd.num_tris;
    info->bvh_depth = (int32_t) s->stats.bvh_depth; info->num_rounds = s->sd.num_rounds;
    for(int a = 0; a < 3; ++a){ info->qorigin[a] = s->sd.qorigin[a]; info->qscale[a] = s->sd.qscale[a]; }
    if(qnodes_out){
        const size_t bytes = (size_t) s->sd.num_nodes * sizeof(QBvhNode);
        if(qnodes_cap < bytes) return fail(HPT_ERR_INVALID, "qnodes_out too small");
        HIP_TRY(hipMemcpy(qnodes_out, s->d_qnodes, bytes, hipMemcpyDeviceToHost));
    }
    if(tris_out){
        const size_t bytes = (size_t) s->sd.num_tris * sizeof(DevTriangle);
        if(tris_cap < bytes) return fail(HPT_ERR_INVALID, "tris_out too small");
        if(bytes) HIP_TRY(hipMemcpy(tris_out, s->d_tris, bytes, hipMemcpyDeviceToHost));
    }
    return HPT_OK;
}

int hpt_trace_visibility(hpt_scene *s, const float *p1, const float *p2, int n, int flags, int32_t *vis_out){
    if(!s || !p1 || !p2 || !vis_out || n < 0) return fail(HPT_ERR_INVALID, "bad argument");
    if(n == 0) return HPT_OK;
    DevBuf d_a, d_b, d_v;
    size_t b3 = (size_t) n * 3 * sizeof(float);
    HIP_TRY(d_a.alloc(b3)); HIP_TRY(d_b.alloc(b3));
    HIP_TRY(d_v.alloc((size_t) n * 4));
    HIP_TRY(hipMemcpy(d_a.p, p1, b3, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_b.p, p2, b3, hipMemcpyHostToDevice));
    launch_probe_visibility(nullptr, s->sd, d_a.as<float>(), d_b.as<float>(), n, (flags & HPT_FLAG_BRUTE_FORCE) ? 1 : 0, d_v.as<int32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(vis_out, d_v.p, (size_t) n * 4, hipMemcpyDeviceToHost));
    return HPT_OK;
}

} // extern "C"
