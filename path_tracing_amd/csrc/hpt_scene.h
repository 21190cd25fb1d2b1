// Device-side scene layout of the path-tracing hot path, shared by the host flattening /
// BVH code (scene_build.cpp) and the HIP kernels (pt_kernels.hip).  Plain structs only.
//
// Everything is 16-byte records so one lane fetches a record with dwordx4 loads:
//   BVH node      64 B  both children's boxes + child codes (one fetch tests two boxes)
//   triangle      48 B  v0 | e1 = v1-v0 | e2 = v2-v0, in BVH leaf order
//   round prim    32 B  spheres followed by light balls (the reference's scan order)
//   material      32 B  de-duplicated CudaMaterial records
//   light        112 B  CudaLight with the per-light invariants hoisted
#pragma once
#include <cstdint>
#include <memory>
#include <utility>
#include <vector>

namespace hpt {

constexpr uint32_t kLeafFlag = 0x80000000u;   // child code: leaf | first_tri << 3 | (count - 1)
constexpr uint32_t kEmptyChild = 0xFFFFFFFFu; // child code of an absent child (box is inverted)
constexpr int kMaxLeafTris = 2;              // triangles per leaf (A/B on four scenes, DESIGN.md section 5: 2 beats 1, 3, 4, 6, 8)
constexpr int kMaxBvhDepth = 30;              // traversal stack holds kStackDepth entries
constexpr int kStackDepth = 32;

struct Float4 { float x, y, z, w; };

// std::vector whose resize() leaves trivially constructible elements uninitialised: the scene arrays are tens to hundreds of
// megabytes, every element is written by the (multi-threaded) build, and a zero fill by the one thread that resizes them --
// page faults included -- was a fifth of the build
template <typename T>
struct default_init_allocator : std::allocator<T> {
    template <typename U> struct rebind { using other = default_init_allocator<U>; };
    default_init_allocator() = default;
    template <typename U> default_init_allocator(const default_init_allocator<U> &){}
    template <typename U> void construct(U *p){ ::new((void *) p) U; }
    template <typename U, typename... A> void construct(U *p, A &&... a){ ::new((void *) p) U(std::forward<A>(a)...); }
};
template <typename T> using pod_vector = std::vector<T, default_init_allocator<T>>;

struct BvhNode {           // 64 B
    float lmin[3]; uint32_t left;
    float lmax[3]; uint32_t right;
    float rmin[3]; uint32_t pad0;
    float rmax[3]; uint32_t pad1;
};

// Quantised twin of BvhNode, 32 B: child boxes as 16-bit grid coordinates of the scene box
// (min rounded down, max rounded up, so the quantised box contains the float box); two dwordx4
// fetches per node visit instead of four.  Traversal only needs conservative boxes.
// plane[axis][0 = lower, 1 = upper][0 = left child, 1 = right child]: word 2 * axis holds the lower planes of both children, word
// 2 * axis + 1 the upper ones -- t = q * i + o is monotone in q, so a ray takes its near planes from one word and its far planes
// from the other by the sign of its direction (one bit-select per word, no min / max per plane pair)
struct QBvhNode {
    uint16_t plane[3][2][2];
    uint32_t left, right;
};

// Four-wide twin of the tree (development: the resume launch's A/B), 64 B: the binary tree
// collapsed greedily (the child with the largest box is opened until a node has four children or only leaves are left),
// child boxes on the same 16-bit grid.  Words 0-3 x, 4-7 y, 8-11 z: lo(c0)|lo(c1)<<16, lo(c2)|lo(c3)<<16, hi(c0)|hi(c1)<<16,
// hi(c2)|hi(c3)<<16 -- a ray picks the near and far planes of two children with one bit-select per word; words 12-15 the
// child codes (leaf codes as in the binary tree, inner = index into the wide node array, kEmptyChild).
struct WideNode { uint32_t w[16]; };

struct DevTriangle {       // 48 B
    float v0[3]; uint32_t ordinal;   // reference scan ordinal (num_spheres + num_lights + input index)
    float e1[3]; uint32_t material;  // index into the material table
    float e2[3]; uint32_t flags;     // bit 0: opaque to shadow rays (mtl.eta <= 0)
};

struct DevRound {          // 32 B: sphere or light ball
    float c[3]; float r;
    uint32_t material;     // spheres: material index; light balls: light index
    uint32_t flags;        // bit 0: opaque to shadow rays; bit 1: is a light ball
    uint32_t pad[2];
};

struct DevMaterial {       // 48 B
    float base[3]; float roughness;
    float metallic; float eta; uint32_t type; uint32_t pad;
    float diffuse[3]; float pad2;            // base / pi * (1 - metallic): the diffuse lobe of every BSDF value (geometric.cuh:433)
};

struct DevLight {          // 112 B
    float pos[3]; float r;
    float main_dir[3]; float cos_cutoff;     // normalize(dir); cosf(cutoff) from the host libm
    float neg_dir[3]; float cutoff;          // normalize(dir * -1) for parallel lights
    float illum[3]; float area;              // 4 * pi * r * r
    uint32_t is_parallel; float cone_ratio;  // (1 - cos_cutoff) / 2
    uint32_t pad[2];
    float raw_dir[3]; uint32_t pad2;         // light.dir as handed over (the BDPT path normalises it itself)
    float ball_c[3]; uint32_t pad3;          // light_ball.center (equals pos for parsed scenes)
};

static_assert(sizeof(BvhNode) == 64 && sizeof(DevTriangle) == 48 && sizeof(DevRound) == 32, "layout");
static_assert(sizeof(QBvhNode) == 32 && sizeof(WideNode) == 64, "layout");
static_assert(sizeof(DevMaterial) == 48 && sizeof(DevLight) == 112, "layout");

// Host-side flattened scene, ready to upload.
struct HostScene {
    pod_vector<BvhNode> nodes;         // nodes[0] is the root (always an inner node)
    pod_vector<QBvhNode> qnodes;       // same tree, quantised boxes
    pod_vector<WideNode> wnodes;       // four-wide collapse of it (the resume launch's tree)
    int wide_depth = 0;
    float qorigin[3] = {0, 0, 0}, qscale[3] = {1, 1, 1};   // grid: coordinate = qorigin + q * qscale
    pod_vector<DevTriangle> tris;      // leaf order
    std::vector<DevRound> rounds;      // spheres, then light balls
    std::vector<DevMaterial> materials;
    std::vector<DevLight> lights;
    int num_spheres = 0, num_lights = 0, num_tris = 0;
    int bvh_depth = 0;
    double ms_bvh_build = 0.0;
};

// ---- scene of the bidirectional (cpu_bdpt-estimator) path -------------------------------------
// The reference's CPU renderer traces a one-level list of groups (reference src/cpu_bdpt.cpp:43-61,
// src/object.cpp:104-146): a group's box is slab-tested, then its objects are tried in insertion
// order with an inclusive range, so among equal distances the LAST object wins.  Each group gets
// its own BVH here; `ordinal` of a triangle / `pad[0]` of a sphere hold that iteration order.
struct DevGroup {          // 48 B
    float mn[3]; uint32_t root;            // child code of the group's BVH (leaf, inner or kEmptyChild)
    float mx[3]; uint32_t sphere_first;
    uint32_t sphere_count; uint32_t pad[3];
};
static_assert(sizeof(DevGroup) == 48, "layout");

struct HostBdptScene {
    std::vector<BvhNode> nodes;
    std::vector<DevTriangle> tris;       // leaf order per group
    std::vector<DevRound> spheres;       // grouped, insertion order inside a group
    std::vector<DevGroup> groups;        // map order
    std::vector<DevMaterial> materials;
    std::vector<DevLight> lights;
    float scene_min[3], scene_max[3];    // union of the group boxes (parallel-light emission)
    int bvh_depth = 0;
};

// obj_kind[i] (0 sphere, 1 triangle), obj_index[i] (into spheres / tris), obj_group[i]: the scene
// file's objects in insertion order; null arrays = one group holding the spheres then the triangles.
const char *build_bdpt_host_scene(const void *lights, int nl, const void *spheres, int ns, const void *tris, int nt,
                                  const int32_t *obj_kind, const int32_t *obj_index, const int32_t *obj_group, int nobj,
                                  HostBdptScene &out);

// Flattens reference records (layouts in include/hpt.h) and builds the BVH.
// Returns an empty string on success, an error message otherwise.
const char *build_host_scene(const void *lights, int nl, const void *spheres, int ns,
                             const void *tris, int nt, HostScene &out);

} // namespace hpt

struct hpt_scene;
namespace hpt {
// uploads a flattened scene to the current HIP device (hpt_api.cpp); the records are kept for the bidirectional path
int scene_upload(const HostScene &hs, const void *lights, int nl, const void *spheres, int ns, const void *tris, int nt,
                 hpt_scene **out);
} // namespace hpt
