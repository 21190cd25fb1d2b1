// How many traversal steps would a 4-wide BVH save?  Host-only count on the build's own tree (no GPU).
//   g++ -O2 -std=c++17 -I path_tracing_amd/csrc -o /tmp/bvh4_steps scripts/micro/bvh4_steps.cpp path_tracing_amd/csrc/scene_build.cpp
//   python scripts/micro/bvh4_steps.py      (writes the scene records, runs the tool)
// The binary tree is collapsed greedily (the child with the largest box is opened until a node has 4 children or only
// leaves are left); rays are diffuse bounce rays: origin on a random triangle (area-weighted), cosine-weighted direction.
// Closest-hit traversal, children visited near-first, stack without distances (as k_trace), leaves of <= 2 triangles.
#include "hpt_scene.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <vector>
using namespace hpt;
struct V { float x, y, z; };
static V sub(V a, V b){ return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V add(V a, V b){ return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V mul(V a, float s){ return {a.x * s, a.y * s, a.z * s}; }
static float dot(V a, V b){ return a.x * b.x + a.y * b.y + a.z * b.z; }
static V cross(V a, V b){ return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static V norm(V a){ return mul(a, 1.0f / std::sqrt(dot(a, a))); }
struct Box4 { float mn[3], mx[3]; uint32_t code; };
struct Node4 { Box4 c[4]; int n; };
static bool slab(const float *mn, const float *mx, V o, V inv, float limit, float &tn){
    float t0 = (mn[0] - o.x) * inv.x, t1 = (mx[0] - o.x) * inv.x; float lo = std::min(t0, t1), hi = std::max(t0, t1);
    t0 = (mn[1] - o.y) * inv.y; t1 = (mx[1] - o.y) * inv.y; lo = std::max(lo, std::min(t0, t1)); hi = std::min(hi, std::max(t0, t1));
    t0 = (mn[2] - o.z) * inv.z; t1 = (mx[2] - o.z) * inv.z; lo = std::max(lo, std::min(t0, t1)); hi = std::min(hi, std::max(t0, t1));
    lo = std::max(lo, 0.0f); hi = std::min(hi, limit);
    tn = lo; return lo <= hi * 1.000002f;
}
static bool tri_hit(const DevTriangle &t, V o, V d, float &tt){
    V v0{t.v0[0], t.v0[1], t.v0[2]}, e1{t.e1[0], t.e1[1], t.e1[2]}, e2{t.e2[0], t.e2[1], t.e2[2]};
    V p = cross(d, e2); float det = dot(e1, p); if(std::fabs(det) < 1e-8f) return false;
    float id = 1.0f / det; V s = sub(o, v0); float u = dot(s, p) * id; if(u < 0 || u > 1) return false;
    V q = cross(s, e1); float v = dot(d, q) * id; if(v < 0 || u + v > 1) return false;
    tt = dot(e2, q) * id; return tt > 1e-4f;
}
int main(int argc, char **argv){
    if(argc < 2){ fprintf(stderr, "usage: bvh4_steps <triangles.bin> [rays]\n"); return 2; }
    FILE *f = fopen(argv[1], "rb"); if(!f) return 2;
    fseek(f, 0, SEEK_END); long bytes = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> raw((size_t) bytes); if(fread(raw.data(), 1, (size_t) bytes, f) != (size_t) bytes) return 2; fclose(f);
    const int TRI = 120;                                // HPT_TRIANGLE_BYTES (include/hpt.h)
    int nt = (int) (bytes / TRI);
    HostScene hs;
    const char *err = build_host_scene(nullptr, 0, nullptr, 0, raw.data(), nt, hs);
    if(err && *err){ fprintf(stderr, "build: %s\n", err); return 1; }
    const auto &N = hs.nodes;
    // ---- collapse ----
    std::vector<Node4> N4; std::vector<int> map2to4(N.size(), -1);
    auto area = [](const float *mn, const float *mx){ float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2]; return dx * dy + dy * dz + dz * dx; };
    std::vector<uint32_t> todo{0u}; map2to4[0] = 0; N4.emplace_back();
    for(size_t qi = 0; qi < todo.size(); ++qi){
        uint32_t n2 = todo[qi]; int me = map2to4[n2];
        std::vector<Box4> kids;
        auto push_children = [&](uint32_t idx){
            const BvhNode &b = N[idx];
            if(b.left != kEmptyChild){ Box4 c; for(int a = 0; a < 3; ++a){ c.mn[a] = b.lmin[a]; c.mx[a] = b.lmax[a]; } c.code = b.left; kids.push_back(c); }
            if(b.right != kEmptyChild){ Box4 c; for(int a = 0; a < 3; ++a){ c.mn[a] = b.rmin[a]; c.mx[a] = b.rmax[a]; } c.code = b.right; kids.push_back(c); }
        };
        push_children(n2);
        for(;;){
            if(kids.size() >= 4) break;
            int best = -1; float ba = -1;
            for(size_t k = 0; k < kids.size(); ++k) if(!(kids[k].code & kLeafFlag)){ float a = area(kids[k].mn, kids[k].mx); if(a > ba){ ba = a; best = (int) k; } }
            if(best < 0) break;
            uint32_t open = kids[(size_t) best].code; kids.erase(kids.begin() + best); push_children(open);
        }
        Node4 out; out.n = (int) kids.size();
        for(int k = 0; k < out.n; ++k){
            out.c[k] = kids[(size_t) k];
            if(!(kids[(size_t) k].code & kLeafFlag)){ uint32_t c2 = kids[(size_t) k].code; map2to4[c2] = (int) N4.size(); N4.emplace_back(); todo.push_back(c2); out.c[k].code = (uint32_t) map2to4[c2]; }
        }
        N4[(size_t) me] = out;
    }
    double fill = 0; for(const Node4 &n : N4) fill += n.n;
    printf("triangles %d, binary nodes %zu (depth %d), 4-wide nodes %zu (%.2f children per node)\n", nt, N.size(), hs.bvh_depth, N4.size(), fill / N4.size());
    // ---- rays ----
    int nr = argc > 2 ? atoi(argv[2]) : 400000;
    std::mt19937 rng(7); std::uniform_real_distribution<float> U(0.0f, 1.0f);
    std::vector<double> cdf((size_t) hs.num_tris); double acc = 0;
    for(int i = 0; i < hs.num_tris; ++i){ const DevTriangle &t = hs.tris[(size_t) i]; V c = cross(V{t.e1[0], t.e1[1], t.e1[2]}, V{t.e2[0], t.e2[1], t.e2[2]}); acc += 0.5 * std::sqrt(dot(c, c)); cdf[(size_t) i] = acc; }
    unsigned long long s2 = 0, s4 = 0, b2 = 0, b4 = 0, t2 = 0, t4 = 0, long2 = 0, long4 = 0, nlong = 0, max2 = 0, max4 = 0, s4u = 0, t4u = 0, long4u = 0, s4d = 0, t4d = 0, culled = 0, c_long = 0, c_restart = 0, c_restart_t = 0, c_cont = 0, c_cont_t = 0, c_entries = 0, c_max_entries = 0;
    std::vector<uint32_t> stk(256);
    for(int r = 0; r < nr; ++r){
        double pick = U(rng) * acc; int ti = (int) (std::lower_bound(cdf.begin(), cdf.end(), pick) - cdf.begin()); ti = std::min(ti, hs.num_tris - 1);
        const DevTriangle &t = hs.tris[(size_t) ti];
        float a = U(rng), b = U(rng); if(a + b > 1){ a = 1 - a; b = 1 - b; }
        V v0{t.v0[0], t.v0[1], t.v0[2]}, e1{t.e1[0], t.e1[1], t.e1[2]}, e2{t.e2[0], t.e2[1], t.e2[2]};
        V n = norm(cross(e1, e2)); if(U(rng) < 0.5f) n = mul(n, -1.0f);
        V o = add(add(v0, add(mul(e1, a), mul(e2, b))), mul(n, 1e-3f));
        V tx = norm(std::fabs(n.z) < 0.999f ? cross(V{0, 0, 1}, n) : cross(V{0, 1, 0}, n)), ty = cross(n, tx);
        float u1 = U(rng), u2 = U(rng), rr = std::sqrt(u1), ph = 6.2831853f * u2;
        V d = norm(add(add(mul(tx, rr * std::cos(ph)), mul(ty, rr * std::sin(ph))), mul(n, std::sqrt(1 - u1))));
        V inv{1.0f / (std::fabs(d.x) > 1e-20f ? d.x : 1e-20f), 1.0f / (std::fabs(d.y) > 1e-20f ? d.y : 1e-20f), 1.0f / (std::fabs(d.z) > 1e-20f ? d.z : 1e-20f)};
        auto leaf = [&](uint32_t code, float &limit, unsigned long long &tc){
            uint32_t first = (code & 0x7FFFFFFFu) >> 3, cnt = (code & 7u) + 1u;
            for(uint32_t k = 0; k < cnt; ++k){ float tt; ++tc; if(tri_hit(hs.tris[first + k], o, d, tt) && tt < limit) limit = tt; }
        };
        // binary
        unsigned long long steps = 0; { float limit = 1e20f; int sp = 0; uint32_t cur = 0;
          for(;;){
              if(cur & kLeafFlag){ leaf(cur, limit, t2); if(sp == 0) break; cur = stk[(size_t) --sp]; continue; }
              ++steps; b2 += 2; const BvhNode &nd = N[cur]; float ln, rn;
              bool hl = nd.left != kEmptyChild && slab(nd.lmin, nd.lmax, o, inv, limit, ln), hr = nd.right != kEmptyChild && slab(nd.rmin, nd.rmax, o, inv, limit, rn);
              if(hl && hr){ bool lf = ln <= rn; stk[(size_t) sp++] = lf ? nd.right : nd.left; cur = lf ? nd.left : nd.right; }
              else if(hl) cur = nd.left; else if(hr) cur = nd.right; else { if(sp == 0) break; cur = stk[(size_t) --sp]; }
          } }
        s2 += steps; max2 = std::max(max2, steps);
        unsigned long long steps4 = 0; { float limit = 1e20f; int sp = 0; uint32_t cur = 0;
          for(;;){
              if(cur & kLeafFlag){ leaf(cur, limit, t4); if(sp == 0) break; cur = stk[(size_t) --sp]; continue; }
              ++steps4; const Node4 &nd = N4[cur]; b4 += (unsigned) nd.n;
              float tn[4]; int idx[4], nh = 0;
              for(int k = 0; k < nd.n; ++k){ float x; if(slab(nd.c[k].mn, nd.c[k].mx, o, inv, limit, x)){ tn[nh] = x; idx[nh] = k; ++nh; } }
              for(int i = 1; i < nh; ++i) for(int j = i; j > 0 && tn[j] < tn[j - 1]; --j){ std::swap(tn[j], tn[j - 1]); std::swap(idx[j], idx[j - 1]); }
              if(nh == 0){ if(sp == 0) break; cur = stk[(size_t) --sp]; continue; }
              for(int i = nh - 1; i >= 1; --i) stk[(size_t) sp++] = nd.c[idx[i]].code;
              cur = nd.c[idx[0]].code;
          } }
        s4 += steps4; max4 = std::max(max4, steps4);
        // 4-wide, cheaper ordering: the nearest hit child is entered, the other hit children are stacked in slot order
        unsigned long long steps4u = 0; { float limit = 1e20f; int sp = 0; uint32_t cur = 0;
          for(;;){
              if(cur & kLeafFlag){ leaf(cur, limit, t4u); if(sp == 0) break; cur = stk[(size_t) --sp]; continue; }
              ++steps4u; const Node4 &nd = N4[cur];
              float tn[4]; int idx[4], nh = 0, best = -1;
              for(int k = 0; k < nd.n; ++k){ float x; if(slab(nd.c[k].mn, nd.c[k].mx, o, inv, limit, x)){ tn[nh] = x; idx[nh] = k; if(best < 0 || x < tn[best]) best = nh; ++nh; } }
              if(nh == 0){ if(sp == 0) break; cur = stk[(size_t) --sp]; continue; }
              for(int i = 0; i < nh; ++i) if(i != best) stk[(size_t) sp++] = nd.c[idx[i]].code;
              cur = nd.c[idx[best]].code;
          } }
        s4u += steps4u;
        // the same walk with the entry distance kept beside every stacked child: a popped entry that lies beyond the hit found
        // meanwhile is dropped without being visited (inner node) or tested (leaf)
        { float limit = 1e20f; int sp = 0; uint32_t cur = 0; float dst[256];
          for(;;){
              if(cur & kLeafFlag){ leaf(cur, limit, t4d); bool got = false; while(sp > 0){ --sp; if(dst[sp] <= limit){ cur = stk[(size_t) sp]; got = true; break; } ++culled; } if(!got) break; continue; }
              ++s4d; const Node4 &nd = N4[cur];
              float tn[4]; int idx[4], nh = 0, best = -1;
              for(int k = 0; k < nd.n; ++k){ float x; if(slab(nd.c[k].mn, nd.c[k].mx, o, inv, limit, x)){ tn[nh] = x; idx[nh] = k; if(best < 0 || x < tn[best]) best = nh; ++nh; } }
              if(nh == 0){ bool got = false; while(sp > 0){ --sp; if(dst[sp] <= limit){ cur = stk[(size_t) sp]; got = true; break; } ++culled; } if(!got) break; continue; }
              for(int i = 0; i < nh; ++i) if(i != best){ dst[sp] = tn[i]; stk[(size_t) sp++] = nd.c[idx[i]].code; }
              cur = nd.c[idx[best]].code;
          } }
        if(steps > 6){ ++nlong; long2 += steps; long4 += steps4; long4u += steps4u; }
        // the split trace step: six binary node steps (leaves tested as they come), then the four-wide walk (nearest child ordered)
        // either RESTARTED at the root with the hit found so far as the limit (what k_trace does), or CONTINUED from the binary
        // walk's own position: its current node and stack entries translated to four-wide codes (a binary node without a
        // four-wide twin -- one that the collapse opened -- stands for its children, untested)
        { float limit = 1e20f; int sp = 0; uint32_t cur = 0; unsigned long long st = 0, tt_ = 0; bool done = false;
          for(;;){
              if(cur & kLeafFlag){ leaf(cur, limit, tt_); if(sp == 0){ done = true; break; } cur = stk[(size_t) --sp]; continue; }
              if(st >= 6) break;
              ++st; const BvhNode &nd = N[cur]; float ln, rn;
              bool hl = nd.left != kEmptyChild && slab(nd.lmin, nd.lmax, o, inv, limit, ln), hr = nd.right != kEmptyChild && slab(nd.rmin, nd.rmax, o, inv, limit, rn);
              if(hl && hr){ bool lf = ln <= rn; stk[(size_t) sp++] = lf ? nd.right : nd.left; cur = lf ? nd.left : nd.right; }
              else if(hl) cur = nd.left; else if(hr) cur = nd.right; else { if(sp == 0){ done = true; break; } cur = stk[(size_t) --sp]; }
          }
          if(!done){
              ++c_long;
              auto walk4 = [&](std::vector<uint32_t> &stack, float lim, unsigned long long &steps_out, unsigned long long &tris_out){
                  uint32_t c = stack.back(); stack.pop_back();
                  for(;;){
                      if(c & kLeafFlag){ leaf(c, lim, tris_out); if(stack.empty()) break; c = stack.back(); stack.pop_back(); continue; }
                      ++steps_out; const Node4 &nd = N4[c];
                      float tn[4]; int idx[4], nh = 0, best = -1;
                      for(int k = 0; k < nd.n; ++k){ float x; if(slab(nd.c[k].mn, nd.c[k].mx, o, inv, lim, x)){ tn[nh] = x; idx[nh] = k; if(best < 0 || x < tn[best]) best = nh; ++nh; } }
                      if(nh == 0){ if(stack.empty()) break; c = stack.back(); stack.pop_back(); continue; }
                      for(int i = 0; i < nh; ++i) if(i != best) stack.push_back(nd.c[idx[i]].code);
                      c = nd.c[idx[best]].code;
                  }
              };
              std::vector<uint32_t> a{0u}; walk4(a, limit, c_restart, c_restart_t);
              // translate: bottom of the binary stack first, the current node on top
              std::vector<uint32_t> b;
              std::function<void(uint32_t)> put = [&](uint32_t code){
                  if(code & kLeafFlag){ b.push_back(code); return; }
                  if(map2to4[code] >= 0){ b.push_back((uint32_t) map2to4[code]); return; }
                  const BvhNode &nd = N[code];
                  if(nd.right != kEmptyChild) put(nd.right);
                  if(nd.left != kEmptyChild) put(nd.left);
              };
              for(int i = 0; i < sp; ++i) put(stk[(size_t) i]);
              put(cur);
              c_entries += b.size(); c_max_entries = std::max<unsigned long long>(c_max_entries, b.size());
              walk4(b, limit, c_cont, c_cont_t);
          } }
    }
    printf("rays %d: binary %.2f node steps per ray (%.2f boxes, %.2f triangle tests, max %llu); 4-wide %.2f steps (%.2f boxes, %.2f triangle tests, max %llu)\n",
           nr, (double) s2 / nr, (double) b2 / nr, (double) t2 / nr, max2, (double) s4 / nr, (double) b4 / nr, (double) t4 / nr, max4);
    printf("steps ratio 4-wide / binary: %.3f; boxes ratio %.3f\n", (double) s4 / s2, (double) b4 / b2);
    printf("4-wide with only the nearest child ordered: %.2f steps per ray (x %.3f of the fully ordered walk), %.2f triangle tests; long rays %.2f steps\n",
           (double) s4u / nr, (double) s4u / std::max(1ull, s4), (double) t4u / nr, (double) long4u / std::max(1ull, nlong));
    printf("4-wide, nearest child ordered, distances kept on the stack: %.2f steps per ray (x %.3f), %.2f triangle tests (x %.3f), %.2f entries dropped per ray\n",
           (double) s4d / nr, (double) s4d / std::max(1ull, s4u), (double) t4d / nr, (double) t4d / std::max(1ull, t4u), (double) culled / nr);
    printf("rays with more than 6 binary steps: %.1f %%, binary %.2f steps, 4-wide %.2f (ratio %.3f)\n", 100.0 * nlong / nr, (double) long2 / std::max(1ull, nlong), (double) long4 / std::max(1ull, nlong), (double) long4 / std::max(1ull, long2));
    printf("split step, %.1f %% of the rays set aside after 6 binary steps: four-wide RESTART %.2f steps and %.2f triangle tests per such ray; CONTINUED from the binary walk's position %.2f steps (x %.3f) and %.2f tests (x %.3f), %.2f translated entries per ray (max %llu)\n",
           100.0 * c_long / nr, (double) c_restart / std::max(1ull, c_long), (double) c_restart_t / std::max(1ull, c_long), (double) c_cont / std::max(1ull, c_long), (double) c_cont / std::max(1ull, c_restart),
           (double) c_cont_t / std::max(1ull, c_long), (double) c_cont_t / std::max(1ull, c_restart_t), (double) c_entries / std::max(1ull, c_long), c_max_entries);
    return 0;
}
