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
#include <cstdlib>
#include <cstring>
#include <map>
#include <array>

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

struct Builder {
    std::vector<Prim> prims;
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> order;      // leaf-order list of input triangle indices
    float pad_abs = 0.0f;
    int max_depth_seen = 0;
    int max_leaf = kMaxLeafTris;
    uint32_t node_base = 0, tri_base = 0;     // offsets when several BVHs share one node / triangle array

    static constexpr int kBins = 16;

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

    uint32_t make_leaf(int first, int count){
        uint32_t start = (uint32_t) order.size() + tri_base;
        for(int i = 0; i < count; ++i) order.push_back(prims[first + i].index);
        return kLeafFlag | (start << 3) | (uint32_t) (count - 1);
    }

    // Builds the subtree over prims[first, first+count); returns its child code and box.
    uint32_t build(int first, int count, int depth, Box &out_box){
        max_depth_seen = std::max(max_depth_seen, depth);
        Box bb; bb.reset();
        Box cb; cb.reset();
        for(int i = first; i < first + count; ++i){ bb.grow(prims[i].box); cb.grow(prims[i].cen); }
        out_box = bb;
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

        int mid = -1;
        if(!force_median && ext[axis] > 0.0f){
            float best_cost = INFINITY; int best_axis = -1, best_bin = -1;
            for(int ax = 0; ax < 3; ++ax){
                if(!(ext[ax] > 0.0f)) continue;
                Box bin_box[kBins]; int bin_cnt[kBins];
                for(int b = 0; b < kBins; ++b){ bin_box[b].reset(); bin_cnt[b] = 0; }
                float scale = (float) kBins / ext[ax];
                for(int i = first; i < first + count; ++i){
                    int b = (int) ((prims[i].cen[ax] - cb.mn[ax]) * scale);
                    b = std::min(std::max(b, 0), kBins - 1);
                    bin_box[b].grow(prims[i].box); bin_cnt[b]++;
                }
                float right_area[kBins]; int right_cnt[kBins];
                Box acc; acc.reset(); int c = 0;
                for(int b = kBins - 1; b > 0; --b){
                    acc.grow(bin_box[b]); c += bin_cnt[b];
                    right_area[b] = acc.half_area(); right_cnt[b] = c;
                }
                acc.reset(); c = 0;
                for(int b = 0; b < kBins - 1; ++b){
                    acc.grow(bin_box[b]); c += bin_cnt[b];
                    if(c == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = acc.half_area() * (float) c + right_area[b + 1] * (float) right_cnt[b + 1];
                    if(cost < best_cost){ best_cost = cost; best_axis = ax; best_bin = b; }
                }
            }
            if(best_axis >= 0){
                float scale = (float) kBins / ext[best_axis];
                float lo = cb.mn[best_axis];
                auto it = std::partition(prims.begin() + first, prims.begin() + first + count, [&](const Prim &p){
                    int b = (int) ((p.cen[best_axis] - lo) * scale);
                    b = std::min(std::max(b, 0), kBins - 1);
                    return b <= best_bin;
                });
                mid = (int) (it - prims.begin());
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
        }

        uint32_t me = (uint32_t) nodes.size();
        nodes.emplace_back();
        // (children return codes that already include node_base / tri_base)
        Box lb, rb;
        uint32_t lc = build(first, mid - first, depth + 1, lb);
        uint32_t rc = build(mid, first + count - mid, depth + 1, rb);
        BvhNode &n = nodes[me];
        write_box(n.lmin, n.lmax, lb); n.left = lc;
        write_box(n.rmin, n.rmax, rb); n.right = rc;
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
    Builder B;
#ifdef HPT_DEV_TUNING      // development builds only (`make variant EXTRA=-DHPT_DEV_TUNING`, scripts/sweep_leaf.py): the product reads no environment here
    if(const char *e = getenv("HPT_MAX_LEAF")){ int v = atoi(e); if(v >= 1 && v <= 8) B.max_leaf = v; }
#endif
    B.prims.resize(nt);
    Box scene_box; scene_box.reset();
    for(int i = 0; i < nt; ++i){
        Prim &p = B.prims[i];
        p.box.reset();
        p.box.grow(tris[i].v0); p.box.grow(tris[i].v1); p.box.grow(tris[i].v2);
        for(int a = 0; a < 3; ++a) p.cen[a] = 0.5f * (p.box.mn[a] + p.box.mx[a]);
        p.index = (uint32_t) i;
        scene_box.grow(p.box);
    }
    float extent = 0.0f;
    if(nt > 0) for(int a = 0; a < 3; ++a){
        extent = std::max(extent, scene_box.mx[a] - scene_box.mn[a]);
        extent = std::max(extent, std::max(std::fabs(scene_box.mn[a]), std::fabs(scene_box.mx[a])));
    }
    B.pad_abs = 2e-6f * extent;
    B.nodes.reserve(nt > 0 ? (size_t) nt : 1);
    B.order.reserve(nt);

    if(nt == 0){
        BvhNode n; memset(&n, 0, sizeof n);
        set_empty_box(n.lmin, n.lmax); set_empty_box(n.rmin, n.rmax);
        n.left = n.right = kEmptyChild;
        B.nodes.push_back(n);
    } else if(nt <= B.max_leaf){
        BvhNode n; memset(&n, 0, sizeof n);
        B.nodes.push_back(n);
        Box lb;
        uint32_t lc = B.build(0, nt, 1, lb);
        B.write_box(B.nodes[0].lmin, B.nodes[0].lmax, lb); B.nodes[0].left = lc;
        set_empty_box(B.nodes[0].rmin, B.nodes[0].rmax); B.nodes[0].right = kEmptyChild;
    } else {
        Box rb;
        uint32_t root = B.build(0, nt, 0, rb);
        if(root != 0) return "internal error: BVH root is not node 0";
    }
    hs.nodes.swap(B.nodes);
    hs.bvh_depth = B.max_depth_seen;
    {   // breadth-first renumbering: node 0 stays the root and every node of depth d precedes every node of depth d + 1,
        // so the nodes a ray can reach in its first k steps are the first 2^k - 1 at most (k_trace keeps those in LDS)
        const size_t n = hs.nodes.size();
        std::vector<uint32_t> order; order.reserve(n);
        std::vector<uint32_t> new_index(n, 0u);
        order.push_back(0u);
        for(size_t head = 0; head < order.size(); ++head){
            const BvhNode &nd = hs.nodes[order[head]];
            for(uint32_t c : { nd.left, nd.right })
                if(c != kEmptyChild && !(c & kLeafFlag)){ new_index[c] = (uint32_t) order.size(); order.push_back(c); }
        }
        if(order.size() != n) return "internal error: BVH nodes unreachable from the root";
        std::vector<BvhNode> sorted(n);
        for(size_t i = 0; i < n; ++i){
            BvhNode nd = hs.nodes[order[i]];
            if(nd.left != kEmptyChild && !(nd.left & kLeafFlag)) nd.left = new_index[nd.left];
            if(nd.right != kEmptyChild && !(nd.right & kLeafFlag)) nd.right = new_index[nd.right];
            sorted[i] = nd;
        }
        hs.nodes.swap(sorted);
    }
    {   // quantised twin: 16-bit grid over the union of the (padded) node boxes
        float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for(const BvhNode &n : hs.nodes){
            for(int a = 0; a < 3; ++a){
                if(n.left != kEmptyChild){ lo[a] = std::min(lo[a], n.lmin[a]); hi[a] = std::max(hi[a], n.lmax[a]); }
                if(n.right != kEmptyChild){ lo[a] = std::min(lo[a], n.rmin[a]); hi[a] = std::max(hi[a], n.rmax[a]); }
            }
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
        for(size_t i = 0; i < hs.nodes.size(); ++i){
            const BvhNode &n = hs.nodes[i]; QBvhNode &q = hs.qnodes[i];
            for(int a = 0; a < 3; ++a){
                if(n.left != kEmptyChild){ q.lmin[a] = qlo(n.lmin[a], a); q.lmax[a] = qhi(n.lmax[a], a); } else { q.lmin[a] = 65535; q.lmax[a] = 0; }
                if(n.right != kEmptyChild){ q.rmin[a] = qlo(n.rmin[a], a); q.rmax[a] = qhi(n.rmax[a], a); } else { q.rmin[a] = 65535; q.rmax[a] = 0; }
            }
            q.left = n.left; q.right = n.right;
        }
    }
    if(hs.bvh_depth > kMaxBvhDepth) return "internal error: BVH deeper than the traversal stack";
    {   // four-wide twin: greedy collapse of the binary tree, breadth-first numbering, same quantised child boxes
        struct Kid { const float *mn, *mx; uint32_t code; };
        auto half_area = [](const Kid &k){ float dx = k.mx[0] - k.mn[0], dy = k.mx[1] - k.mn[1], dz = k.mx[2] - k.mn[2]; return dx * dy + dy * dz + dz * dx; };
        auto qlo = [&](float v, int a){ double q = std::floor(((double) v - (double) hs.qorigin[a]) / (double) hs.qscale[a]) - 1.0; return (uint32_t) std::min(65535.0, std::max(0.0, q)); };
        auto qhi = [&](float v, int a){ double q = std::ceil(((double) v - (double) hs.qorigin[a]) / (double) hs.qscale[a]) + 1.0; return (uint32_t) std::min(65535.0, std::max(0.0, q)); };
        std::vector<uint32_t> todo{0u};            // binary node each wide node was collapsed from
        std::vector<int> depth{1};
        hs.wnodes.clear(); hs.wnodes.emplace_back();
        for(size_t qi = 0; qi < todo.size(); ++qi){
            Kid kids[4]; int nk = 0;
            auto open = [&](uint32_t idx){
                const BvhNode &b = hs.nodes[idx];
                if(b.left != kEmptyChild) kids[nk++] = Kid{ b.lmin, b.lmax, b.left };
                if(b.right != kEmptyChild) kids[nk++] = Kid{ b.rmin, b.rmax, b.right };
            };
            open(todo[qi]);
            while(nk < 4){
                int best = -1; float ba = -1.0f;
                for(int k = 0; k < nk; ++k) if(!(kids[k].code & kLeafFlag)){ float a = half_area(kids[k]); if(a > ba){ ba = a; best = k; } }
                if(best < 0) break;
                // an inner binary node has two children at most: opening one replaces it by them (nk grows by one at most)
                uint32_t c = kids[best].code;
                kids[best] = kids[nk - 1]; --nk;
                open(c);
            }
            WideNode w;
            uint32_t lo[3][4], hi[3][4], code[4];
            for(int k = 0; k < 4; ++k){
                if(k < nk){
                    for(int a = 0; a < 3; ++a){ lo[a][k] = qlo(kids[k].mn[a], a); hi[a][k] = qhi(kids[k].mx[a], a); }
                    if(kids[k].code & kLeafFlag) code[k] = kids[k].code;
                    else { code[k] = (uint32_t) todo.size(); todo.push_back(kids[k].code); depth.push_back(depth[qi] + 1); hs.wnodes.emplace_back(); }
                } else { for(int a = 0; a < 3; ++a){ lo[a][k] = 65535u; hi[a][k] = 0u; } code[k] = kEmptyChild; }
            }
            for(int a = 0; a < 3; ++a){
                w.w[a * 4 + 0] = lo[a][0] | (lo[a][1] << 16); w.w[a * 4 + 1] = lo[a][2] | (lo[a][3] << 16);
                w.w[a * 4 + 2] = hi[a][0] | (hi[a][1] << 16); w.w[a * 4 + 3] = hi[a][2] | (hi[a][3] << 16);
            }
            for(int k = 0; k < 4; ++k) w.w[12 + k] = code[k];
            hs.wnodes[qi] = w;
            hs.wide_depth = std::max(hs.wide_depth, depth[qi]);
        }
    }

    hs.tris.resize(nt);
    for(int s = 0; s < nt; ++s){
        const RefTriangle &t = tris[B.order[s]];
        DevTriangle &d = hs.tris[s];
        for(int a = 0; a < 3; ++a){
            d.v0[a] = t.v0[a];
            d.e1[a] = t.v1[a] - t.v0[a];          // geometric.cuh:266-267
            d.e2[a] = t.v2[a] - t.v0[a];
        }
        d.ordinal = (uint32_t) (ns + nl) + B.order[s];
        d.material = intern_material(t.m, mat_table, hs.materials);
        d.flags = (t.m.eta <= 0.0f) ? 1u : 0u;
    }
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
        if(gtris.empty()) g.root = kEmptyChild;
        else { Box rb; g.root = B.build(0, (int) gtris.size(), 0, rb); }
        if(B.max_depth_seen > kMaxBvhDepth) return "internal error: BVH deeper than the traversal stack";
        out.bvh_depth = std::max(out.bvh_depth, B.max_depth_seen);
        out.nodes.insert(out.nodes.end(), B.nodes.begin(), B.nodes.end());
        for(uint32_t oi : B.order){
            const Item &it = gtris[oi];
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
