// Host side of scene upload: flattens the reference's AoS records into the device layout
// (hpt_scene.h) and builds a binned-SAH BVH2 over the triangles.
//
// The reference has no acceleration structure: closest-hit and shadow rays scan every
// primitive (reference include/geometric.cuh:293-388).  The BVH must therefore return
// exactly what the scan returns: leaves keep the reference scan ordinal of every
// triangle so traversal can break exact ties the way the scan's strict '<' does, and
// node boxes are padded so no hit the scan accepts is culled.
//
// Float expressions here that feed the kernels (light invariants, triangle edges) are
// evaluated in the order written, without FMA contraction (-ffp-contract=off).
#include "hpt_scene.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <array>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#ifdef __linux__
#include <sched.h>
#endif

namespace hpt {
namespace {

// ---- reference record layouts (include/hpt.h; reference include/geometric.cuh:15-78) ----
struct RefMatOld { float Kd[3], Kg[3], Ks[3], glossy, exp, refract, reflect; };
struct RefMat { float base[3], roughness, metallic, eta; int32_t type; };
struct RefSphere { float c[3], r; RefMatOld o; RefMat m; int32_t id; };
struct RefTriangle { float v0[3], v1[3], v2[3]; RefMatOld o; RefMat m; int32_t id; };
struct RefLight { float pos[3], dir[3], illum[3]; RefSphere ball; float cutoff; int32_t is_parallel; };
static_assert(sizeof(RefSphere) == 100 && sizeof(RefTriangle) == 120 && sizeof(RefLight) == 144, "layout");

constexpr float kPi = 3.14159265358979323846f;

struct Box {
    float mn[3], mx[3];
    void reset(){ for(int a = 0; a < 3; ++a){ mn[a] = INFINITY; mx[a] = -INFINITY; } }
    void grow(const float *p){ for(int a = 0; a < 3; ++a){ mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); } }
    void grow(const Box &b){ for(int a = 0; a < 3; ++a){ mn[a] = std::min(mn[a], b.mn[a]); mx[a] = std::max(mx[a], b.mx[a]); } }
    float half_area() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if(dx < 0 || dy < 0 || dz < 0) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Prim { Box box; float cen[3]; uint32_t index; };

// Host threads for the O(n) passes of a large scene: [0, n) is cut into kChunks contiguous ranges -- always the same ones,
// whatever the machine, so that nothing a pass derives from the cut depends on where it runs -- and the ranges are handed
// to a small pool of helper threads (created on first use, at most kChunks - 1) and to the caller itself, which takes
// ranges like any helper and returns when all are done: a caller never waits for a helper to START, so a pool that is
// busy (several big subtrees at once), or gone (a forked child), only costs speed.  fn(chunk, begin, end).
constexpr int kChunks = 32;
constexpr size_t kParallelForMin = 65536;

// Threads worth starting: the hardware's count cut down to the scheduler affinity mask, 64 at most.  (A cgroup CPU quota is
// deliberately NOT applied: on the 16-CPU-quota GPU boxes the 1 M-triangle build measured 700 ms on 1 thread, 330-380 on
// 16, 270-310 on 32 and 260 on 64.)
int usable_cpus(){
    static const int n = [](){
        long best = (long) std::thread::hardware_concurrency();
        if(best < 1) best = 1;
#ifdef HPT_DEV_TUNING
        if(const char *e = getenv("HPT_BUILD_THREADS")){ int v = atoi(e); if(v >= 1) return v; }
#endif
#ifdef __linux__
        cpu_set_t set; CPU_ZERO(&set);
        if(sched_getaffinity(0, sizeof set, &set) == 0){ long c = CPU_COUNT(&set); if(c > 0 && c < best) best = c; }
#endif
        return (int) std::min<long>(best, 64);
    }();
    return n;
}

class HelperPool {
public:
    static HelperPool &get(){ static HelperPool p; return p; }
    void submit(const std::function<void()> &job, int copies){
        { std::lock_guard<std::mutex> lock(mu_);
          if(workers_.empty()){
              int n = std::min(usable_cpus() - 1, kChunks - 1);
              for(int i = 0; i < n; ++i) workers_.emplace_back([this](){ run(); });
          }
          for(int i = 0; i < copies && i < (int) workers_.size(); ++i) queue_.push_back(job); }
        cv_.notify_all();
    }
    ~HelperPool(){
        { std::lock_guard<std::mutex> lock(mu_); stop_ = true; }
        cv_.notify_all();
        for(std::thread &t : workers_) t.join();
    }
private:
    void run(){
        for(;;){
            std::function<void()> job;
            { std::unique_lock<std::mutex> lock(mu_);
              cv_.wait(lock, [this](){ return stop_ || !queue_.empty(); });
              if(stop_) return;
              job = std::move(queue_.front()); queue_.pop_front(); }
            job();
        }
    }
    std::mutex mu_; std::condition_variable cv_; std::deque<std::function<void()>> queue_; std::vector<std::thread> workers_; bool stop_ = false;
};

template <typename F>
void parallel_chunks(size_t n, F fn){
    auto range = [n](int c, size_t &b, size_t &e){ b = n * (size_t) c / kChunks; e = n * (size_t) (c + 1) / kChunks; };
    if(n < kParallelForMin || usable_cpus() <= 1){
        for(int c = 0; c < kChunks; ++c){ size_t b, e; range(c, b, e); if(b < e) fn(c, b, e); }
        return;
    }
    struct Job { std::atomic<int> next{0}, done{0}; };
    auto job = std::make_shared<Job>();
    // helpers hold the job (and a copy of fn's captures by reference: valid until done == kChunks, which the caller awaits)
    auto take = [job, range, &fn](){
        for(;;){
            int c = job->next.fetch_add(1);
            if(c >= kChunks) break;
            size_t b, e; range(c, b, e);
            if(b < e) fn(c, b, e);
            job->done.fetch_add(1, std::memory_order_release);
        }
    };
    HelperPool::get().submit(take, kChunks - 1);
    take();
    while(job->done.load(std::memory_order_acquire) < kChunks) std::this_thread::yield();
}

struct Builder {
    pod_vector<Prim> prims;
    pod_vector<BvhNode> nodes;        // pre-sized by the caller; slots are handed out by next_node
    pod_vector<Prim> scratch;         // the big nodes' partition buffer (sized once, by the root)
    std::atomic<uint32_t> next_node{0};
    std::atomic<int> max_depth_seen{0};
    float pad_abs = 0.0f;
    int max_leaf = kMaxLeafTris;
    uint32_t node_base = 0, tri_base = 0;     // offsets when several BVHs share one node / triangle array

    static constexpr int kBins = 16;
    // Subtrees of more than kParallelMin triangles are built on their own host thread while the parent goes on with the
    // sibling (at most as many threads at a time as the process may use CPUs).  The tree does not depend on it: a subtree over
    // prims[first, first + count) only reads and permutes that range, its triangles keep that range of leaf slots
    // (leaf order = the final order of prims), and node numbers -- the only thing the thread schedule decides -- are
    // replaced by the breadth-first renumbering afterwards.
    static constexpr int kParallelMin = 8192;
    static int parallel_levels(){ int l = 0; while((2 << l) <= usable_cpus() && l < 6) ++l; return l; }     // 2^levels subtree threads at most
    static constexpr size_t kBigNode = 131072;        // nodes this large bin and partition their range on several threads

    void write_box(float *mn, float *mx, const Box &b) const {
        // pad by an absolute slack plus two ulps so that rounding in the slab test and in
        // Moeller-Trumbore can never cull a hit the brute-force scan accepts
        for(int a = 0; a < 3; ++a){
            float lo = b.mn[a] - pad_abs, hi = b.mx[a] + pad_abs;
            lo = std::nextafter(std::nextafter(lo, -INFINITY), -INFINITY);
            hi = std::nextafter(std::nextafter(hi, INFINITY), INFINITY);
            mn[a] = lo; mx[a] = hi;
        }
    }

    uint32_t make_leaf(int first, int count) const {
        return kLeafFlag | (((uint32_t) first + tri_base) << 3) | (uint32_t) (count - 1);
    }

    // node slots are handed out in blocks of kSlotBlock per thread (one shared counter, touched once per block); the slots a
    // thread does not use stay empty and are dropped by the breadth-first renumbering.  The root takes slot 0.
    static constexpr uint32_t kSlotBlock = 256;
    static constexpr size_t kSlotSlack = (size_t) kSlotBlock * 160;      // > (threads that ever build a subtree) x block: 2 x 64 spawned at most
    const uint64_t id = [](){ static std::atomic<uint64_t> counter{0}; return ++counter; }();
    uint32_t alloc_node(){
        static thread_local uint64_t owner = 0;
        static thread_local uint32_t next = 0, end = 0;
        if(owner != id || next == end){ owner = id; next = next_node.fetch_add(kSlotBlock, std::memory_order_relaxed); end = next + kSlotBlock; }
        return next++;
    }

    void note_depth(int depth){
        int seen = max_depth_seen.load(std::memory_order_relaxed);
        while(depth > seen && !max_depth_seen.compare_exchange_weak(seen, depth, std::memory_order_relaxed)){}
    }

    // Bounds (primitive boxes, centroids) of prims[first, first+count): the root's only -- every other node gets its own from the
    // partition that made it.
    void range_bounds(int first, int count, Box &bb, Box &cb, bool parallel){
        bb.reset(); cb.reset();
        if(parallel && (size_t) count >= kBigNode){
            std::vector<Box> pb((size_t) kChunks), pc((size_t) kChunks);
            for(int c = 0; c < kChunks; ++c){ pb[(size_t) c].reset(); pc[(size_t) c].reset(); }
            parallel_chunks((size_t) count, [&](int c, size_t b0, size_t e0){
                for(size_t i = (size_t) first + b0; i < (size_t) first + e0; ++i){ pb[(size_t) c].grow(prims[i].box); pc[(size_t) c].grow(prims[i].cen); }
            });
            for(int c = 0; c < kChunks; ++c){ bb.grow(pb[(size_t) c]); cb.grow(pc[(size_t) c]); }
        } else
        for(int i = first; i < first + count; ++i){ bb.grow(prims[i].box); cb.grow(prims[i].cen); }
    }

    uint32_t build(int first, int count, int depth, Box &out_box, int par_levels = -1){
        if(par_levels < 0) par_levels = parallel_levels();
        if(par_levels > 0 && (size_t) count >= kBigNode && scratch.size() < prims.size()) scratch.resize(prims.size());    // before any thread is started
        Box bb, cb;
        range_bounds(first, count, bb, cb, par_levels > 0);
        out_box = bb;
        return build_node(first, count, depth, bb, cb, par_levels);
    }

    struct Bins { Box box[3][kBins]; int cnt[3][kBins]; void reset(){ for(int a = 0; a < 3; ++a) for(int b = 0; b < kBins; ++b){ box[a][b].reset(); cnt[a][b] = 0; } } };

    // Builds the subtree over prims[first, first+count) whose primitive boxes span bb and whose centroids span cb; returns its
    // child code.  One pass bins the range on all three axes, one pass partitions it and gathers the two children's bounds.
    uint32_t build_node(int first, int count, int depth, const Box &bb, const Box &cb, int par_levels){
        note_depth(depth);
        if(count <= max_leaf) return make_leaf(first, count);

        // levels still available below this node; force balanced splits when they run short
        int remaining = kMaxBvhDepth - depth;
        int need = 0;
        { int leaves = (count + max_leaf - 1) / max_leaf; while((1 << need) < leaves) ++need; }
        bool force_median = need >= remaining;

        int axis = 0;
        float ext[3] = { cb.mx[0] - cb.mn[0], cb.mx[1] - cb.mn[1], cb.mx[2] - cb.mn[2] };
        if(ext[1] > ext[axis]) axis = 1;
        if(ext[2] > ext[axis]) axis = 2;

        const bool big = par_levels > 0 && (size_t) count >= kBigNode;        // the few nodes at the top of a large tree
        int mid = -1;
        Box lbb, lcb, rbb, rcb; lbb.reset(); lcb.reset(); rbb.reset(); rcb.reset();
        if(!force_median && ext[axis] > 0.0f){
            float scale[3]; bool use[3];
            for(int ax = 0; ax < 3; ++ax){ use[ax] = ext[ax] > 0.0f; scale[ax] = use[ax] ? (float) kBins / ext[ax] : 0.0f; }
            auto bin_range = [&](Bins &bins, size_t b0, size_t e0){
                for(size_t i = b0; i < e0; ++i){
                    const Prim &p = prims[i];
                    for(int ax = 0; ax < 3; ++ax){
                        if(!use[ax]) continue;
                        int b = (int) ((p.cen[ax] - cb.mn[ax]) * scale[ax]);
                        b = std::min(std::max(b, 0), kBins - 1);
                        bins.box[ax][b].grow(p.box); bins.cnt[ax][b]++;
                    }
                }
            };
            Bins bins; bins.reset();
            if(big){
                // per-chunk bins merged afterwards: boxes grow by min / max and counts add, so the result is the serial one
                std::vector<Bins> part((size_t) kChunks);
                for(Bins &b : part) b.reset();
                parallel_chunks((size_t) count, [&](int c, size_t b0, size_t e0){ bin_range(part[(size_t) c], (size_t) first + b0, (size_t) first + e0); });
                for(const Bins &pb : part) for(int ax = 0; ax < 3; ++ax) for(int b = 0; b < kBins; ++b){ bins.box[ax][b].grow(pb.box[ax][b]); bins.cnt[ax][b] += pb.cnt[ax][b]; }
            } else bin_range(bins, (size_t) first, (size_t) first + (size_t) count);

            float best_cost = INFINITY; int best_axis = -1, best_bin = -1;
            for(int ax = 0; ax < 3; ++ax){
                if(!use[ax]) continue;
                float right_area[kBins]; int right_cnt[kBins];
                Box acc; acc.reset(); int c = 0;
                for(int b = kBins - 1; b > 0; --b){
                    acc.grow(bins.box[ax][b]); c += bins.cnt[ax][b];
                    right_area[b] = acc.half_area(); right_cnt[b] = c;
                }
                acc.reset(); c = 0;
                for(int b = 0; b < kBins - 1; ++b){
                    acc.grow(bins.box[ax][b]); c += bins.cnt[ax][b];
                    if(c == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = acc.half_area() * (float) c + right_area[b + 1] * (float) right_cnt[b + 1];
                    if(cost < best_cost){ best_cost = cost; best_axis = ax; best_bin = b; }
                }
            }
            if(best_axis >= 0){
                const float sc = scale[best_axis], lo = cb.mn[best_axis];
                auto goes_left = [&](const Prim &p){
                    int b = (int) ((p.cen[best_axis] - lo) * sc);
                    b = std::min(std::max(b, 0), kBins - 1);
                    return b <= best_bin;
                };
                if(big){
                    // stable partition through a scratch copy: every chunk counts (and gathers the children's bounds), then
                    // scatters to the offsets the counts give
                    std::vector<size_t> nleft((size_t) kChunks, 0), nall((size_t) kChunks, 0);
                    std::vector<Box> cl((size_t) kChunks), ccl((size_t) kChunks), cr((size_t) kChunks), ccr((size_t) kChunks);
                    for(int c = 0; c < kChunks; ++c){ cl[(size_t) c].reset(); ccl[(size_t) c].reset(); cr[(size_t) c].reset(); ccr[(size_t) c].reset(); }
                    parallel_chunks((size_t) count, [&](int c, size_t b0, size_t e0){
                        size_t l = 0;
                        for(size_t i = (size_t) first + b0; i < (size_t) first + e0; ++i){
                            const Prim &p = prims[i];
                            if(goes_left(p)){ ++l; cl[(size_t) c].grow(p.box); ccl[(size_t) c].grow(p.cen); } else { cr[(size_t) c].grow(p.box); ccr[(size_t) c].grow(p.cen); }
                        }
                        nleft[(size_t) c] = l; nall[(size_t) c] = e0 - b0;
                    });
                    size_t total_left = 0; for(size_t l : nleft) total_left += l;
                    for(int c = 0; c < kChunks; ++c){ lbb.grow(cl[(size_t) c]); lcb.grow(ccl[(size_t) c]); rbb.grow(cr[(size_t) c]); rcb.grow(ccr[(size_t) c]); }
                    std::vector<size_t> loff((size_t) kChunks), roff((size_t) kChunks);
                    { size_t l = 0, r = total_left; for(int c = 0; c < kChunks; ++c){ loff[(size_t) c] = l; roff[(size_t) c] = r; l += nleft[(size_t) c]; r += nall[(size_t) c] - nleft[(size_t) c]; } }
                    // (the scratch range [first, first + count) belongs to this node alone: big nodes working at once are disjoint)
                    Prim *tmp = scratch.data() + first;
                    parallel_chunks((size_t) count, [&](int c, size_t b0, size_t e0){
                        size_t l = loff[(size_t) c], r = roff[(size_t) c];
                        for(size_t i = (size_t) first + b0; i < (size_t) first + e0; ++i){ if(goes_left(prims[i])) tmp[l++] = prims[i]; else tmp[r++] = prims[i]; }
                    });
                    parallel_chunks((size_t) count, [&](int, size_t b0, size_t e0){ std::copy(tmp + b0, tmp + e0, prims.begin() + first + (long) b0); });
                    mid = first + (int) total_left;
                } else {
                    // std::partition applies the predicate to every element exactly once: the children's bounds are gathered on the way
                    auto it = std::partition(prims.begin() + first, prims.begin() + first + count, [&](const Prim &p){
                        const bool l = goes_left(p);
                        if(l){ lbb.grow(p.box); lcb.grow(p.cen); } else { rbb.grow(p.box); rcb.grow(p.cen); }
                        return l;
                    });
                    mid = (int) (it - prims.begin());
                }
                if(mid == first || mid == first + count) mid = -1;
            }
        }
        if(mid < 0){
            // object median along the widest centroid axis (also the degenerate-centroid case)
            mid = first + count / 2;
            std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + first + count,
                             [&](const Prim &a, const Prim &b){
                                 if(a.cen[axis] != b.cen[axis]) return a.cen[axis] < b.cen[axis];
                                 return a.index < b.index;
                             });
            range_bounds(first, mid - first, lbb, lcb, false);
            range_bounds(mid, first + count - mid, rbb, rcb, false);
        }

        uint32_t me = alloc_node();
        // (children return codes that already include node_base / tri_base)
        uint32_t lc, rc;
        if(par_levels > 0 && count > kParallelMin){
            std::thread left([&](){ lc = build_node(first, mid - first, depth + 1, lbb, lcb, par_levels - 1); });
            rc = build_node(mid, first + count - mid, depth + 1, rbb, rcb, par_levels - 1);
            left.join();
        } else {
            lc = build_node(first, mid - first, depth + 1, lbb, lcb, 0);
            rc = build_node(mid, first + count - mid, depth + 1, rbb, rcb, 0);
        }
        BvhNode &n = nodes[me];
        write_box(n.lmin, n.lmax, lbb); n.left = lc;
        write_box(n.rmin, n.rmax, rbb); n.right = rc;
        n.pad0 = n.pad1 = 0;
        return me + node_base;
    }
};

void set_empty_box(float *mn, float *mx){
    for(int a = 0; a < 3; ++a){ mn[a] = INFINITY; mx[a] = -INFINITY; }
}

struct MatKey {
    uint32_t w[7];
    bool operator<(const MatKey &o) const { return memcmp(w, o.w, sizeof w) < 0; }
};

uint32_t intern_material(const RefMat &m, std::map<MatKey, uint32_t> &table, std::vector<DevMaterial> &out){
    MatKey k; memcpy(k.w, &m, sizeof k.w);
    auto it = table.find(k);
    if(it != table.end()) return it->second;
    DevMaterial d;
    d.base[0] = m.base[0]; d.base[1] = m.base[1]; d.base[2] = m.base[2];
    d.roughness = m.roughness; d.metallic = m.metallic; d.eta = m.eta; d.type = (uint32_t) m.type; d.pad = 0;
    // same float operations as the kernel's `m.base / kPi * (1.0f - m.metallic)` (IEEE divide, no contraction)
    const float pi = 3.14159265358979323846f;
    for(int a = 0; a < 3; ++a){ float q = m.base[a] / pi; d.diffuse[a] = q * (1.0f - m.metallic); }
    d.pad2 = 0.0f;
    uint32_t idx = (uint32_t) out.size();
    out.push_back(d);
    table.emplace(k, idx);
    return idx;
}

void normalize3(const float *v, float s, float *out){
    // normalize(v * s) with the reference's float3 algebra (geometric.cuh:92,95,97,98)
    float x = v[0] * s, y = v[1] * s, z = v[2] * s;
    float len = sqrtf(x * x + y * y + z * z);
    out[0] = x / len; out[1] = y / len; out[2] = z / len;
}

} // namespace

const char *build_host_scene(const void *lights_v, int nl, const void *spheres_v, int ns,
                             const void *tris_v, int nt, HostScene &hs){
    if(nl < 0 || ns < 0 || nt < 0) return "negative primitive count";
    if((nl > 0 && !lights_v) || (ns > 0 && !spheres_v) || (nt > 0 && !tris_v)) return "null primitive array";
    if((uint64_t) nt >= (1ull << 28)) return "too many triangles (limit 2^28)";
    const RefLight *lights = (const RefLight *) lights_v;
    const RefSphere *spheres = (const RefSphere *) spheres_v;
    const RefTriangle *tris = (const RefTriangle *) tris_v;

    hs = HostScene();
    hs.num_spheres = ns; hs.num_lights = nl; hs.num_tris = nt;
    std::map<MatKey, uint32_t> mat_table;

    // spheres then light balls: the reference's closest-hit scan order (geometric.cuh:340-368)
    hs.rounds.reserve((size_t) ns + nl);
    for(int i = 0; i < ns; ++i){
        DevRound r; memset(&r, 0, sizeof r);
        r.c[0] = spheres[i].c[0]; r.c[1] = spheres[i].c[1]; r.c[2] = spheres[i].c[2]; r.r = spheres[i].r;
        r.material = intern_material(spheres[i].m, mat_table, hs.materials);
        r.flags = (spheres[i].m.eta <= 0.0f) ? 1u : 0u;
        hs.rounds.push_back(r);
    }
    for(int i = 0; i < nl; ++i){
        DevRound r; memset(&r, 0, sizeof r);
        const RefSphere &b = lights[i].ball;
        r.c[0] = b.c[0]; r.c[1] = b.c[1]; r.c[2] = b.c[2]; r.r = b.r;
        r.material = (uint32_t) i;
        r.flags = 2u;
        hs.rounds.push_back(r);
    }
    hs.lights.reserve(nl);
    for(int i = 0; i < nl; ++i){
        const RefLight &L = lights[i];
        DevLight d; memset(&d, 0, sizeof d);
        d.raw_dir[0] = L.dir[0]; d.raw_dir[1] = L.dir[1]; d.raw_dir[2] = L.dir[2];
        d.ball_c[0] = L.ball.c[0]; d.ball_c[1] = L.ball.c[1]; d.ball_c[2] = L.ball.c[2];
        d.pos[0] = L.pos[0]; d.pos[1] = L.pos[1]; d.pos[2] = L.pos[2];
        d.r = L.ball.r;
        normalize3(L.dir, 1.0f, d.main_dir);          // normalize(light.dir), pt_cu.cu:75,167
        normalize3(L.dir, -1.0f, d.neg_dir);          // normalize(light.dir * -1.0f), pt_cu.cu:131
        d.cos_cutoff = cosf(L.cutoff);                // pt_cu.cu:73,79,168
        d.cutoff = L.cutoff;
        d.illum[0] = L.illum[0]; d.illum[1] = L.illum[1]; d.illum[2] = L.illum[2];
        d.area = 4.0f * kPi * L.ball.r * L.ball.r;    // pt_cu.cu:70,179
        d.is_parallel = L.is_parallel ? 1u : 0u;
        d.cone_ratio = (1.0f - d.cos_cutoff) / 2.0f;  // pt_cu.cu:73
        hs.lights.push_back(d);
    }

    auto t0 = std::chrono::steady_clock::now();
#ifdef HPT_DEV_TUNING
    auto tp = t0;
    auto lap = [&](const char *what){ auto n = std::chrono::steady_clock::now(); fprintf(stderr, "  build phase %-12s %.1f ms\n", what, std::chrono::duration<double, std::milli>(n - tp).count()); tp = n; };
#else
    auto lap = [](const char *){};
#endif
    Builder B;
#ifdef HPT_DEV_TUNING      // development builds only (`make variant EXTRA=-DHPT_DEV_TUNING`, scripts/sweep_leaf.py): the product reads no environment here
    if(const char *e = getenv("HPT_MAX_LEAF")){ int v = atoi(e); if(v >= 1 && v <= 8) B.max_leaf = v; }
#endif
    B.prims.resize(nt);
    Box scene_box; scene_box.reset();
    {   std::vector<Box> part((size_t) kChunks);
        for(Box &b : part) b.reset();
        parallel_chunks((size_t) nt, [&](int c, size_t b0, size_t e0){
            for(size_t i = b0; i < e0; ++i){
                Prim &p = B.prims[i];
                p.box.reset();
                p.box.grow(tris[i].v0); p.box.grow(tris[i].v1); p.box.grow(tris[i].v2);
                for(int a = 0; a < 3; ++a) p.cen[a] = 0.5f * (p.box.mn[a] + p.box.mx[a]);
                p.index = (uint32_t) i;
                part[(size_t) c].grow(p.box);
            }
        });
        for(const Box &b : part) scene_box.grow(b);
    }
    float extent = 0.0f;
    if(nt > 0) for(int a = 0; a < 3; ++a){
        extent = std::max(extent, scene_box.mx[a] - scene_box.mn[a]);
        extent = std::max(extent, std::max(std::fabs(scene_box.mn[a]), std::fabs(scene_box.mx[a])));
    }
    B.pad_abs = 2e-6f * extent;
    B.nodes.resize((nt > 0 ? (size_t) nt : 1) + Builder::kSlotSlack);          // an inner node has two non-empty subtrees: fewer than nt of them
    lap("prims");

    if(nt == 0){
        BvhNode n; memset(&n, 0, sizeof n);
        set_empty_box(n.lmin, n.lmax); set_empty_box(n.rmin, n.rmax);
        n.left = n.right = kEmptyChild;
        B.nodes[0] = n; B.next_node = 1;
    } else if(nt <= B.max_leaf){
        BvhNode n; memset(&n, 0, sizeof n);
        B.nodes[0] = n; B.next_node = 1;
        Box lb;
        uint32_t lc = B.build(0, nt, 1, lb);
        B.write_box(B.nodes[0].lmin, B.nodes[0].lmax, lb); B.nodes[0].left = lc;
        set_empty_box(B.nodes[0].rmin, B.nodes[0].rmax); B.nodes[0].right = kEmptyChild;
    } else {
        Box rb;
        uint32_t root = B.build(0, nt, 0, rb);
        if(root != 0) return "internal error: BVH root is not node 0";
    }
    lap("tree");
    B.nodes.resize(B.next_node.load());
    hs.nodes.swap(B.nodes);
    hs.bvh_depth = B.max_depth_seen.load();
    {   // breadth-first renumbering: node 0 stays the root and every node of depth d precedes every node of depth d + 1,
        // so the nodes a ray can reach in its first k steps are the first 2^k - 1 at most (k_trace keeps those in LDS)
        const size_t n = hs.nodes.size();
        pod_vector<uint32_t> order(n);
        pod_vector<uint32_t> new_index(n);
        auto inner = [](uint32_t c){ return c != kEmptyChild && !(c & kLeafFlag); };
        // level by level: the inner children of a level, in order, are the next level; a level is cut into chunks that count
        // their children, a prefix sum places them, and the chunks fill their part
        size_t lvl_begin = 0, lvl_end = 1;
        order[0] = 0u;
        while(lvl_begin < lvl_end){
            const size_t cnt = lvl_end - lvl_begin;
            size_t per_chunk[kChunks] = {};
            parallel_chunks(cnt, [&](int c, size_t b0, size_t e0){
                size_t k = 0;
                for(size_t i = b0; i < e0; ++i){ const BvhNode &nd = hs.nodes[order[lvl_begin + i]]; k += (inner(nd.left) ? 1u : 0u) + (inner(nd.right) ? 1u : 0u); }
                per_chunk[c] = k;
            });
            size_t offs[kChunks], total = 0;
            for(int c = 0; c < kChunks; ++c){ offs[c] = total; total += per_chunk[c]; }
            if(lvl_end + total > n) return "internal error: BVH has more reachable nodes than slots";
            parallel_chunks(cnt, [&](int c, size_t b0, size_t e0){
                size_t o = lvl_end + offs[c];
                for(size_t i = b0; i < e0; ++i){
                    const BvhNode &nd = hs.nodes[order[lvl_begin + i]];
                    if(inner(nd.left)){ new_index[nd.left] = (uint32_t) o; order[o++] = nd.left; }
                    if(inner(nd.right)){ new_index[nd.right] = (uint32_t) o; order[o++] = nd.right; }
                }
            });
            lvl_begin = lvl_end; lvl_end += total;
        }
        order.resize(lvl_end);
        // (slots a build thread reserved and did not use are not reachable and drop out here)
        const size_t reachable = order.size();
        pod_vector<BvhNode> sorted(reachable);
        parallel_chunks(reachable, [&](int, size_t b0, size_t e0){
            for(size_t i = b0; i < e0; ++i){
                BvhNode nd = hs.nodes[order[i]];
                if(nd.left != kEmptyChild && !(nd.left & kLeafFlag)) nd.left = new_index[nd.left];
                if(nd.right != kEmptyChild && !(nd.right & kLeafFlag)) nd.right = new_index[nd.right];
                sorted[i] = nd;
            }
        });
        hs.nodes.swap(sorted);
    }
    lap("renumber");
    {   // quantised twin: 16-bit grid over the union of the (padded) node boxes
        float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        {   std::vector<Box> part((size_t) kChunks);
            for(Box &b : part) b.reset();
            parallel_chunks(hs.nodes.size(), [&](int c, size_t b0, size_t e0){
                Box &pb = part[(size_t) c];
                for(size_t i = b0; i < e0; ++i){
                    const BvhNode &n = hs.nodes[i];
                    if(n.left != kEmptyChild){ pb.grow(n.lmin); pb.grow(n.lmax); }
                    if(n.right != kEmptyChild){ pb.grow(n.rmin); pb.grow(n.rmax); }
                }
            });
            for(const Box &b : part) for(int a = 0; a < 3; ++a){ lo[a] = std::min(lo[a], b.mn[a]); hi[a] = std::max(hi[a], b.mx[a]); }
        }
        for(int a = 0; a < 3; ++a){
            if(!(lo[a] <= hi[a])){ lo[a] = 0.0f; hi[a] = 1.0f; }
            hs.qorigin[a] = lo[a];
            float ext = hi[a] - lo[a];
            hs.qscale[a] = ext > 0.0f ? ext / 65535.0f : 1.0f;
            hs.qscale[a] = std::nextafter(hs.qscale[a], INFINITY);          // never shorter than the box
        }
        auto qlo = [&](float v, int a){ double q = std::floor(((double) v - (double) hs.qorigin[a]) / (double) hs.qscale[a]) - 1.0; return (uint16_t) std::min(65535.0, std::max(0.0, q)); };
        auto qhi = [&](float v, int a){ double q = std::ceil(((double) v - (double) hs.qorigin[a]) / (double) hs.qscale[a]) + 1.0; return (uint16_t) std::min(65535.0, std::max(0.0, q)); };
        hs.qnodes.resize(hs.nodes.size());
        parallel_chunks(hs.nodes.size(), [&](int, size_t b0, size_t e0){
            for(size_t i = b0; i < e0; ++i){
                const BvhNode &n = hs.nodes[i]; QBvhNode &q = hs.qnodes[i];
                for(int a = 0; a < 3; ++a){
                    if(n.left != kEmptyChild){ q.plane[a][0][0] = qlo(n.lmin[a], a); q.plane[a][1][0] = qhi(n.lmax[a], a); } else { q.plane[a][0][0] = 65535; q.plane[a][1][0] = 0; }
                    if(n.right != kEmptyChild){ q.plane[a][0][1] = qlo(n.rmin[a], a); q.plane[a][1][1] = qhi(n.rmax[a], a); } else { q.plane[a][0][1] = 65535; q.plane[a][1][1] = 0; }
                }
                q.left = n.left; q.right = n.right;
            }
        });
    }
    if(hs.bvh_depth > kMaxBvhDepth) return "internal error: BVH deeper than the traversal stack";
    lap("quantise");
    {   // four-wide twin: greedy collapse of the binary tree, breadth-first numbering, same quantised child boxes
        struct Kid { const float *mn, *mx; uint32_t code; };
        auto half_area = [](const Kid &k){ float dx = k.mx[0] - k.mn[0], dy = k.mx[1] - k.mn[1], dz = k.mx[2] - k.mn[2]; return dx * dy + dy * dz + dz * dx; };
        auto qlo = [&](float v, int a){ double q = std::floor(((double) v - (double) hs.qorigin[a]) / (double) hs.qscale[a]) - 1.0; return (uint32_t) std::min(65535.0, std::max(0.0, q)); };
        auto qhi = [&](float v, int a){ double q = std::ceil(((double) v - (double) hs.qorigin[a]) / (double) hs.qscale[a]) + 1.0; return (uint32_t) std::min(65535.0, std::max(0.0, q)); };
        struct Kids { Kid k[4]; int n; uint32_t code[4]; };
        pod_vector<uint32_t> todo(hs.nodes.size() + 1);             // binary node each wide node was collapsed from (breadth-first)
        pod_vector<Kids> all(hs.nodes.size() + 1);
        todo[0] = 0u;
        // pass 1, level by level: which child boxes a wide node holds; the inner ones among them are the next level (a chunk of
        // the level counts its own, a prefix sum gives them their numbers)
        size_t lvl_begin = 0, lvl_end = 1;
        while(lvl_begin < lvl_end){
            const size_t cnt = lvl_end - lvl_begin;
            size_t per_chunk[kChunks] = {};
            parallel_chunks(cnt, [&](int c, size_t b0, size_t e0){
                size_t inner_kids = 0;
                for(size_t i = b0; i < e0; ++i){
                    Kids ks; ks.n = 0;
                    auto open = [&](uint32_t idx){
                        const BvhNode &b = hs.nodes[idx];
                        if(b.left != kEmptyChild) ks.k[ks.n++] = Kid{ b.lmin, b.lmax, b.left };
                        if(b.right != kEmptyChild) ks.k[ks.n++] = Kid{ b.rmin, b.rmax, b.right };
                    };
                    open(todo[lvl_begin + i]);
                    while(ks.n < 4){
                        int best = -1; float ba = -1.0f;
                        for(int k = 0; k < ks.n; ++k) if(!(ks.k[k].code & kLeafFlag)){ float a = half_area(ks.k[k]); if(a > ba){ ba = a; best = k; } }
                        if(best < 0) break;
                        // an inner binary node has two children at most: opening one replaces it by them (n grows by one at most)
                        uint32_t cc = ks.k[best].code;
                        ks.k[best] = ks.k[ks.n - 1]; --ks.n;
                        open(cc);
                    }
                    for(int k = 0; k < ks.n; ++k) if(!(ks.k[k].code & kLeafFlag)) ++inner_kids;
                    all[lvl_begin + i] = ks;
                }
                per_chunk[c] = inner_kids;
            });
            size_t offs[kChunks], total = 0;
            for(int c = 0; c < kChunks; ++c){ offs[c] = total; total += per_chunk[c]; }
            parallel_chunks(cnt, [&](int c, size_t b0, size_t e0){
                size_t o = lvl_end + offs[c];
                for(size_t i = b0; i < e0; ++i){
                    Kids &ks = all[lvl_begin + i];
                    for(int k = 0; k < 4; ++k){
                        if(k >= ks.n) ks.code[k] = kEmptyChild;
                        else if(ks.k[k].code & kLeafFlag) ks.code[k] = ks.k[k].code;
                        else { ks.code[k] = (uint32_t) o; todo[o++] = ks.k[k].code; }
                    }
                }
            });
            ++hs.wide_depth;
            lvl_begin = lvl_end; lvl_end += total;
        }
        all.resize(lvl_end);
        // pass 2: quantise and pack
        hs.wnodes.resize(all.size());
        parallel_chunks(all.size(), [&](int, size_t b0, size_t e0){
            for(size_t qi = b0; qi < e0; ++qi){
                const Kids &ks = all[qi];
                WideNode w;
                uint32_t lo[3][4], hi[3][4];
                for(int k = 0; k < 4; ++k){
                    if(k < ks.n){ for(int a = 0; a < 3; ++a){ lo[a][k] = qlo(ks.k[k].mn[a], a); hi[a][k] = qhi(ks.k[k].mx[a], a); } }
                    else { for(int a = 0; a < 3; ++a){ lo[a][k] = 65535u; hi[a][k] = 0u; } }
                }
                for(int a = 0; a < 3; ++a){
                    w.w[a * 4 + 0] = lo[a][0] | (lo[a][1] << 16); w.w[a * 4 + 1] = lo[a][2] | (lo[a][3] << 16);
                    w.w[a * 4 + 2] = hi[a][0] | (hi[a][1] << 16); w.w[a * 4 + 3] = hi[a][2] | (hi[a][3] << 16);
                }
                for(int k = 0; k < 4; ++k) w.w[12 + k] = ks.code[k];
                hs.wnodes[qi] = w;
            }
        });
    }

    lap("wide");
    hs.tris.resize(nt);
    // materials are numbered in order of first appearance in the INPUT (spheres, then triangles as handed over): one sequential
    // pass over the records that mostly hits its one-entry cache; the leaf-order fill below looks the numbers up
    pod_vector<uint32_t> mat_of((size_t) nt);
    {   RefMat last; memset(&last, 0xFF, sizeof last); uint32_t last_idx = 0; bool have = false;
        for(int i = 0; i < nt; ++i){
            const RefMat &m = tris[i].m;
            if(!have || memcmp(&m, &last, 28) != 0){ last_idx = intern_material(m, mat_table, hs.materials); last = m; have = true; }
            mat_of[(size_t) i] = last_idx;
        }
    }
    parallel_chunks((size_t) nt, [&](int, size_t b0, size_t e0){
        for(size_t s = b0; s < e0; ++s){
            const RefTriangle &t = tris[B.prims[s].index];        // leaf order = the order the build left the primitives in
            DevTriangle &d = hs.tris[s];
            for(int a = 0; a < 3; ++a){
                d.v0[a] = t.v0[a];
                d.e1[a] = t.v1[a] - t.v0[a];          // geometric.cuh:266-267
                d.e2[a] = t.v2[a] - t.v0[a];
            }
            d.ordinal = (uint32_t) (ns + nl) + B.prims[s].index;
            d.material = mat_of[B.prims[s].index];
            d.flags = (t.m.eta <= 0.0f) ? 1u : 0u;
        }
    });
    lap("triangles");
    auto t1 = std::chrono::steady_clock::now();
    hs.ms_bvh_build = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if(hs.materials.empty()){ DevMaterial m; memset(&m, 0, sizeof m); hs.materials.push_back(m); }
    return "";
}


const char *build_bdpt_host_scene(const void *lights_v, int nl, const void *spheres_v, int ns, const void *tris_v, int nt,
                                  const int32_t *obj_kind, const int32_t *obj_index, const int32_t *obj_group, int nobj,
                                  HostBdptScene &out){
    if(nl < 0 || ns < 0 || nt < 0) return "negative primitive count";
    if((nl > 0 && !lights_v) || (ns > 0 && !spheres_v) || (nt > 0 && !tris_v)) return "null primitive array";
    const RefSphere *spheres = (const RefSphere *) spheres_v;
    const RefTriangle *tris = (const RefTriangle *) tris_v;
    out = HostBdptScene();
    {   // the light records are the PT path's
        HostScene tmp;
        const char *e = build_host_scene(lights_v, nl, nullptr, 0, nullptr, 0, tmp);
        if(e && *e) return e;
        out.lights = tmp.lights;
    }
    struct Item { int kind, index; uint32_t seq; };
    std::map<int, std::vector<Item>> gm;
    uint32_t seq = 0;
    if(obj_kind && obj_index && obj_group){
        // sequence numbers follow the CPU renderer's iteration order: groups in map order, objects in insertion order
        for(int i = 0; i < nobj; ++i){
            if(obj_kind[i] == 0 ? (obj_index[i] < 0 || obj_index[i] >= ns) : (obj_index[i] < 0 || obj_index[i] >= nt)) return "object index out of range";
            gm[obj_group[i]].push_back(Item{ obj_kind[i], obj_index[i], 0u });
        }
    } else {
        for(int i = 0; i < ns; ++i) gm[0].push_back(Item{ 0, i, 0u });
        for(int i = 0; i < nt; ++i) gm[0].push_back(Item{ 1, i, 0u });
    }
    for(auto &kv : gm) for(Item &it : kv.second) it.seq = seq++;

    std::map<MatKey, uint32_t> mat_table;
    Box scene_box; scene_box.reset();
    std::vector<Box> gboxes;
    for(auto &kv : gm){
        Box gb; for(int a = 0; a < 3; ++a){ gb.mn[a] = 99999.f; gb.mx[a] = -99999.f; }       // AABB::add_obj, src/object.cpp:123-146
        for(const Item &it : kv.second){
            if(it.kind == 0){
                const RefSphere &s = spheres[it.index];
                for(int a = 0; a < 3; ++a){
                    gb.mn[a] = std::min({ gb.mn[a], s.c[a] + s.r, s.c[a] - s.r });
                    gb.mx[a] = std::max({ gb.mx[a], s.c[a] + s.r, s.c[a] - s.r });
                }
            } else {
                const RefTriangle &t = tris[it.index];
                const float *vv[3] = { t.v0, t.v1, t.v2 };
                for(int k = 0; k < 3; ++k) for(int a = 0; a < 3; ++a){ gb.mn[a] = std::min(gb.mn[a], vv[k][a]); gb.mx[a] = std::max(gb.mx[a], vv[k][a]); }
            }
        }
        // AABB::intersectAABB widens a degenerate axis in place every time it is called
        // (src/object.cpp:108-111); the box it settles on is used from the start here
        for(int a = 0; a < 3; ++a) for(int guard = 0; guard < 64 && gb.mx[a] - gb.mn[a] < 1e-6f; ++guard){ gb.mn[a] -= 0.5f * 1e-6f; gb.mx[a] += 0.5f * 1e-6f; }
        gboxes.push_back(gb);
        scene_box.grow(gb);
    }
    for(int a = 0; a < 3; ++a){ out.scene_min[a] = 1e9f; out.scene_max[a] = -1e9f; }
    for(const Box &gb : gboxes) for(int a = 0; a < 3; ++a){ out.scene_min[a] = std::min(out.scene_min[a], gb.mn[a]); out.scene_max[a] = std::max(out.scene_max[a], gb.mx[a]); }
    float extent = 0.0f;
    if(!gboxes.empty()) for(int a = 0; a < 3; ++a){
        extent = std::max(extent, scene_box.mx[a] - scene_box.mn[a]);
        extent = std::max(extent, std::max(std::fabs(scene_box.mn[a]), std::fabs(scene_box.mx[a])));
    }
    size_t gi = 0;
    for(auto &kv : gm){
        DevGroup g; memset(&g, 0, sizeof g);
        for(int a = 0; a < 3; ++a){ g.mn[a] = gboxes[gi].mn[a]; g.mx[a] = gboxes[gi].mx[a]; }
        g.sphere_first = (uint32_t) out.spheres.size();
        std::vector<Item> gtris;
        for(const Item &it : kv.second){
            if(it.kind == 0){
                const RefSphere &s = spheres[it.index];
                DevRound r; memset(&r, 0, sizeof r);
                r.c[0] = s.c[0]; r.c[1] = s.c[1]; r.c[2] = s.c[2]; r.r = s.r;
                r.material = intern_material(s.m, mat_table, out.materials);
                r.flags = (s.m.eta <= 0.0f) ? 1u : 0u;
                r.pad[0] = it.seq;
                out.spheres.push_back(r);
            } else gtris.push_back(it);
        }
        g.sphere_count = (uint32_t) out.spheres.size() - g.sphere_first;
        Builder B;
        B.node_base = (uint32_t) out.nodes.size(); B.tri_base = (uint32_t) out.tris.size();
        B.pad_abs = 2e-6f * extent;
        B.prims.resize(gtris.size());
        for(size_t i = 0; i < gtris.size(); ++i){
            const RefTriangle &t = tris[gtris[i].index];
            Prim &p = B.prims[i];
            p.box.reset(); p.box.grow(t.v0); p.box.grow(t.v1); p.box.grow(t.v2);
            for(int a = 0; a < 3; ++a) p.cen[a] = 0.5f * (p.box.mn[a] + p.box.mx[a]);
            p.index = (uint32_t) i;
        }
        B.nodes.resize(std::max<size_t>(gtris.size(), 1) + Builder::kSlotSlack);
        if(gtris.empty()) g.root = kEmptyChild;
        else { Box rb; g.root = B.build(0, (int) gtris.size(), 0, rb); }
        if(B.max_depth_seen.load() > kMaxBvhDepth) return "internal error: BVH deeper than the traversal stack";
        out.bvh_depth = std::max(out.bvh_depth, B.max_depth_seen.load());
        out.nodes.insert(out.nodes.end(), B.nodes.begin(), B.nodes.begin() + B.next_node.load());
        for(const Prim &pr : B.prims){
            const Item &it = gtris[pr.index];
            const RefTriangle &t = tris[it.index];
            DevTriangle d;
            for(int a = 0; a < 3; ++a){ d.v0[a] = t.v0[a]; d.e1[a] = t.v1[a] - t.v0[a]; d.e2[a] = t.v2[a] - t.v0[a]; }   // src/object.cpp:75
            d.ordinal = it.seq;
            d.material = intern_material(t.m, mat_table, out.materials);
            d.flags = (t.m.eta <= 0.0f) ? 1u : 0u;
            out.tris.push_back(d);
        }
        out.groups.push_back(g);
        ++gi;
    }
    if(out.materials.empty()){ DevMaterial m; memset(&m, 0, sizeof m); out.materials.push_back(m); }
    if(out.nodes.empty()){ BvhNode n; memset(&n, 0, sizeof n); out.nodes.push_back(n); }
    return "";
}

} // namespace hpt
