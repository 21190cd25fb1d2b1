// Single-process multi-device fan-out behind the blocking render call (include/hpt.h, hpt_multi_*).
//
// The reference's launch API is one blocking call per frame (run_cuda_pt -> pt_render_wrapper, reference
// src/pt_cu_helper.cpp:66-77, src/pt_cu.cu:255-297).  Pixels are independent (src/pt_cu.cu:27-35), so the call
// fans out internally: the scene is flattened and its BVH built once, uploaded to every device, each device
// renders its image tiles (one host thread per device drives hpt_render_pt_device on that device's stream),
// the packed local framebuffers are gathered on the first device -- one ncclGather per device inside one RCCL
// group, i.e. every sender uses its own xGMI link to the root, nothing is ringed -- and un-tiled there.
// The random streams are keyed by global pixel and sample index, so the image is the single-device image bit
// for bit whatever the number of devices.
//
// RCCL is loaded with dlopen at the first multi-device use, so libhpt.so itself has no link dependency on it;
// if it cannot be loaded or a communicator cannot be made the call fails (no silent substitute).  Exchange
// mode 1 (explicit, for boxes without RCCL and for tests that put several ranks on ONE device, which RCCL
// refuses) moves the buffers with hipMemcpyPeerAsync instead.
#include "../../include/hpt.h"
#include "hpt_scene.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace hpt { int fail_with(int code, const std::string &msg); }

namespace {

using hpt::fail_with;

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Gather)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl &rccl(){
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [](){
        const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for(const char *n : names){ r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL); if(r.handle) break; }
        if(!r.handle){ r.error = std::string("cannot load librccl.so: ") + dlerror(); return; }
        auto sym = [&](const char *name){ void *p = dlsym(r.handle, name); if(!p && r.error.empty()) r.error = std::string("librccl.so lacks ") + name; return p; };
        r.CommInitAll = (decltype(r.CommInitAll)) sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy)) sym("ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart)) sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd)) sym("ncclGroupEnd");
        r.Gather = (decltype(r.Gather)) sym("ncclGather");
        r.GetErrorString = (decltype(r.GetErrorString)) sym("ncclGetErrorString");
    });
    return r;
}

#define MHIP_TRY(expr) do { hipError_t e_ = (expr); if(e_ != hipSuccess) \
    return fail_with(e_ == hipErrorOutOfMemory ? HPT_ERR_NOMEM : HPT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } while(0)
#define NCCL_TRY(expr) do { ncclResult_t r_ = (expr); if(r_ != ncclSuccess) \
    return fail_with(HPT_ERR_DEVICE, std::string(#expr) + ": " + rccl().GetErrorString(r_)); } while(0)

} // namespace

struct hpt_multi {
    int n = 0, exchange = 0;
    std::vector<int> dev;
    std::vector<hpt_scene *> scene;
    std::vector<hipStream_t> stream;
    std::vector<float *> d_local; size_t cap_local = 0;       // floats per device
    float *d_gathered = nullptr; size_t cap_gathered = 0;     // on dev[0]
    float *d_image = nullptr; size_t cap_image = 0;           // on dev[0]
    std::vector<ncclComm_t> comm;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    std::vector<double> render_ms;
    double gather_ms = 0.0, total_ms = 0.0;
    // what the scene was built from (the one-shot wrappers reuse a fan-out for byte-identical arrays)
    std::vector<unsigned char> h_lights, h_spheres, h_tris;
};

namespace {

int ensure_buffers(hpt_multi *m, size_t n_local, size_t W, size_t H){
    size_t nloc = n_local * 3;
    if(nloc > m->cap_local){
        for(int d = 0; d < m->n; ++d){
            MHIP_TRY(hipSetDevice(m->dev[d]));
            hipFree(m->d_local[d]); m->d_local[d] = nullptr;
            MHIP_TRY(hipMalloc((void **) &m->d_local[d], nloc * sizeof(float)));
        }
        m->cap_local = nloc;
    }
    MHIP_TRY(hipSetDevice(m->dev[0]));
    if(nloc * m->n > m->cap_gathered){
        hipFree(m->d_gathered); m->d_gathered = nullptr; m->cap_gathered = 0;
        MHIP_TRY(hipMalloc((void **) &m->d_gathered, nloc * m->n * sizeof(float)));
        m->cap_gathered = nloc * m->n;
    }
    if(W * H * 3 > m->cap_image){
        hipFree(m->d_image); m->d_image = nullptr; m->cap_image = 0;
        MHIP_TRY(hipMalloc((void **) &m->d_image, W * H * 3 * sizeof(float)));
        m->cap_image = W * H * 3;
    }
    return HPT_OK;
}

// every device renders its tiles; `render` is called on the device's own host thread with the device current
template <typename F>
int fan_out(hpt_multi *m, F render){
    std::vector<int> rc(m->n, HPT_OK);
    std::vector<std::string> msg(m->n);
    std::vector<std::thread> th;
    th.reserve(m->n);
    for(int d = 0; d < m->n; ++d){
        th.emplace_back([&, d](){
            hipError_t e = hipSetDevice(m->dev[d]);
            if(e != hipSuccess){ rc[d] = HPT_ERR_DEVICE; msg[d] = std::string("hipSetDevice: ") + hipGetErrorString(e); return; }
            rc[d] = render(d);
            if(rc[d] == HPT_OK){
                e = hipStreamSynchronize(m->stream[d]);
                if(e != hipSuccess){ rc[d] = HPT_ERR_DEVICE; msg[d] = std::string("hipStreamSynchronize: ") + hipGetErrorString(e); return; }
                hpt_stats st;
                if(hpt_get_stats(m->scene[d], &st) == HPT_OK) m->render_ms[d] = st.ms_total;
            } else msg[d] = hpt_last_error();
        });
    }
    for(std::thread &t : th) t.join();
    for(int d = 0; d < m->n; ++d)
        if(rc[d] != HPT_OK) return fail_with(rc[d], "device " + std::to_string(m->dev[d]) + " (rank " + std::to_string(d) + "): " + msg[d]);
    return HPT_OK;
}

// packed local framebuffers -> [rank][local slot] on dev[0] -> row-major image -> host
int exchange_and_assemble(hpt_multi *m, size_t n_local, int W, int H, const hpt_params *params, float *host_image){
    const size_t count = n_local * 3;
    MHIP_TRY(hipSetDevice(m->dev[0]));
    MHIP_TRY(hipEventRecord(m->ev_a, m->stream[0]));
    if(m->exchange == 0){
        Rccl &R = rccl();
        NCCL_TRY(R.GroupStart());
        // the group is closed on every path: the first failure inside it is kept and returned after GroupEnd (an open
        // group would swallow every later RCCL call of this thread, ncclCommDestroy in hpt_multi_destroy included)
        std::string first_error;
        for(int d = 0; d < m->n && first_error.empty(); ++d){
            hipError_t e = hipSetDevice(m->dev[d]);
            if(e != hipSuccess){ first_error = std::string("hipSetDevice: ") + hipGetErrorString(e); break; }
            ncclResult_t r = R.Gather(m->d_local[d], d == 0 ? m->d_gathered : nullptr, count, ncclFloat, 0, m->comm[d], m->stream[d]);
            if(r != ncclSuccess) first_error = std::string("ncclGather (rank ") + std::to_string(d) + "): " + R.GetErrorString(r);
        }
        ncclResult_t ge = R.GroupEnd();
        if(!first_error.empty()){ (void) hipSetDevice(m->dev[0]); return fail_with(HPT_ERR_DEVICE, first_error); }
        if(ge != ncclSuccess) return fail_with(HPT_ERR_DEVICE, std::string("ncclGroupEnd: ") + R.GetErrorString(ge));
        MHIP_TRY(hipSetDevice(m->dev[0]));
    } else {
        for(int d = 0; d < m->n; ++d){
            if(m->dev[d] == m->dev[0])
                MHIP_TRY(hipMemcpyAsync(m->d_gathered + d * count, m->d_local[d], count * sizeof(float), hipMemcpyDeviceToDevice, m->stream[0]));
            else
                MHIP_TRY(hipMemcpyPeerAsync(m->d_gathered + d * count, m->dev[0], m->d_local[d], m->dev[d], count * sizeof(float), m->stream[0]));
        }
    }
    MHIP_TRY(hipEventRecord(m->ev_b, m->stream[0]));
    hpt_params p; memset(&p, 0, sizeof p);
    if(params) p = *params;
    p.rank = 0; p.world = m->n;
    int rc = hpt_untile(m->d_gathered, m->d_image, W, H, &p, m->stream[0]);
    if(rc) return rc;
    MHIP_TRY(hipMemcpyAsync(host_image, m->d_image, (size_t) W * H * 3 * sizeof(float), hipMemcpyDeviceToHost, m->stream[0]));
    MHIP_TRY(hipStreamSynchronize(m->stream[0]));
    float ms = 0.f;
    if(hipEventElapsedTime(&ms, m->ev_a, m->ev_b) == hipSuccess) m->gather_ms = ms;
    return HPT_OK;
}

template <typename F>
int multi_render(hpt_multi *m, int W, int H, const hpt_params *params, float *host_image, F render_rank){
    if(!m) return fail_with(HPT_ERR_INVALID, "null fan-out handle");
    if(!host_image) return fail_with(HPT_ERR_INVALID, "null image");
    int restore = 0; hipGetDevice(&restore);
    hpt_params p; memset(&p, 0, sizeof p);
    if(params) p = *params;
    p.rank = 0; p.world = m->n;
    int64_t n_local = hpt_local_pixels(W, H, &p);
    if(n_local <= 0) return HPT_ERR_INVALID;
    auto t0 = std::chrono::steady_clock::now();
    int rc = ensure_buffers(m, (size_t) n_local, (size_t) W, (size_t) H);
    if(rc == HPT_OK) rc = fan_out(m, [&](int d){ hpt_params q = p; q.rank = d; return render_rank(d, q); });
    if(rc == HPT_OK) rc = exchange_and_assemble(m, (size_t) n_local, W, H, &p, host_image);
    m->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    hipSetDevice(restore);
    return rc;
}

} // namespace

extern "C" {

void hpt_multi_destroy(hpt_multi *m){
    if(!m) return;
    int restore = 0; hipGetDevice(&restore);
    for(size_t d = 0; d < m->comm.size(); ++d) if(m->comm[d]) rccl().CommDestroy(m->comm[d]);
    for(int d = 0; d < (int) m->dev.size(); ++d){
        hipSetDevice(m->dev[d]);
        if(d < (int) m->scene.size() && m->scene[d]) hpt_scene_destroy(m->scene[d]);
        if(d < (int) m->d_local.size()) hipFree(m->d_local[d]);
        if(d < (int) m->stream.size() && m->stream[d]) hipStreamDestroy(m->stream[d]);
    }
    if(!m->dev.empty()){
        hipSetDevice(m->dev[0]);
        hipFree(m->d_gathered); hipFree(m->d_image);
        if(m->ev_a) hipEventDestroy(m->ev_a);
        if(m->ev_b) hipEventDestroy(m->ev_b);
    }
    hipSetDevice(restore);
    delete m;
}

int hpt_multi_create(const void *lights, int nl, const void *spheres, int ns, const void *tris, int nt,
                     const int *device_ids, int num_devices, int exchange, hpt_multi **out){
    if(!out) return fail_with(HPT_ERR_INVALID, "null out handle");
    *out = nullptr;
    if(exchange != 0 && exchange != 1) return fail_with(HPT_ERR_INVALID, "exchange must be 0 (RCCL) or 1 (peer copies)");
    int visible = 0;
    MHIP_TRY(hipGetDeviceCount(&visible));
    if(num_devices <= 0) num_devices = visible;
    if(num_devices <= 0 || num_devices > 64) return fail_with(HPT_ERR_INVALID, "no HIP device (or more than 64 ranks)");
    hpt_multi *m = new hpt_multi();
    m->n = num_devices; m->exchange = exchange;
    for(int d = 0; d < num_devices; ++d){
        int id = device_ids ? device_ids[d] : d;
        if(id < 0 || id >= visible){ delete m; return fail_with(HPT_ERR_INVALID, "device id outside [0, hipGetDeviceCount)"); }
        m->dev.push_back(id);
    }
    if(exchange == 0){
        for(int a = 0; a < num_devices; ++a) for(int b = a + 1; b < num_devices; ++b)
            if(m->dev[a] == m->dev[b]){ delete m; return fail_with(HPT_ERR_INVALID, "RCCL exchange needs distinct devices (several ranks on one device: exchange = 1)"); }
        if(!rccl().error.empty()){ delete m; return fail_with(HPT_ERR_DEVICE, rccl().error); }
    }
    int restore = 0; hipGetDevice(&restore);
    m->scene.assign(num_devices, nullptr); m->stream.assign(num_devices, nullptr); m->d_local.assign(num_devices, nullptr);
    m->render_ms.assign(num_devices, 0.0);
    // flatten + BVH once, upload everywhere
    hpt::HostScene hs;
    const char *err = hpt::build_host_scene(lights, nl, spheres, ns, tris, nt, hs);
    int rc = HPT_OK;
    if(err && *err) rc = fail_with(HPT_ERR_INVALID, err);
    for(int d = 0; d < num_devices && rc == HPT_OK; ++d){
        hipError_t e = hipSetDevice(m->dev[d]);
        if(e == hipSuccess) e = hipStreamCreateWithFlags(&m->stream[d], hipStreamNonBlocking);
        if(e != hipSuccess){ rc = fail_with(HPT_ERR_DEVICE, std::string("device set-up: ") + hipGetErrorString(e)); break; }
        rc = hpt::scene_upload(hs, lights, nl, spheres, ns, tris, nt, &m->scene[d]);
    }
    if(rc == HPT_OK){
        hipError_t e = hipSetDevice(m->dev[0]);
        if(e == hipSuccess) e = hipEventCreate(&m->ev_a);
        if(e == hipSuccess) e = hipEventCreate(&m->ev_b);
        if(e != hipSuccess) rc = fail_with(HPT_ERR_DEVICE, std::string("event set-up: ") + hipGetErrorString(e));
    }
    if(rc == HPT_OK && exchange == 0){
        m->comm.assign(num_devices, nullptr);
        ncclResult_t r = rccl().CommInitAll(m->comm.data(), num_devices, m->dev.data());
        if(r != ncclSuccess){ m->comm.clear(); rc = fail_with(HPT_ERR_DEVICE, std::string("ncclCommInitAll: ") + rccl().GetErrorString(r)); }
    }
    if(rc == HPT_OK && exchange == 1){
        // peer access for the copies into the root's buffer (already-enabled is fine)
        for(int d = 1; d < num_devices; ++d) if(m->dev[d] != m->dev[0]){
            int can = 0;
            hipDeviceCanAccessPeer(&can, m->dev[0], m->dev[d]);
            if(can){ hipSetDevice(m->dev[0]); hipError_t e = hipDeviceEnablePeerAccess(m->dev[d], 0); if(e != hipSuccess) (void) hipGetLastError(); }
        }
    }
    if(rc == HPT_OK){
        if(nl) m->h_lights.assign((const unsigned char *) lights, (const unsigned char *) lights + (size_t) nl * HPT_LIGHT_BYTES);
        if(ns) m->h_spheres.assign((const unsigned char *) spheres, (const unsigned char *) spheres + (size_t) ns * HPT_SPHERE_BYTES);
        if(nt) m->h_tris.assign((const unsigned char *) tris, (const unsigned char *) tris + (size_t) nt * HPT_TRIANGLE_BYTES);
    }
    hipSetDevice(restore);
    if(rc != HPT_OK){ std::string keep = hpt_last_error(); hpt_multi_destroy(m); return fail_with(rc, keep); }
    *out = m;
    return HPT_OK;
}

int hpt_multi_num_devices(const hpt_multi *m){ return m ? m->n : 0; }

int hpt_multi_set_groups(hpt_multi *m, const int32_t *obj_kind, const int32_t *obj_index, const int32_t *obj_group, int num_objects){
    if(!m) return fail_with(HPT_ERR_INVALID, "null fan-out handle");
    for(int d = 0; d < m->n; ++d){
        int rc = hpt_scene_set_groups(m->scene[d], obj_kind, obj_index, obj_group, num_objects);
        if(rc) return rc;
    }
    return HPT_OK;
}

int hpt_multi_render_pt(hpt_multi *m, const void *camera, int W, int H, int eye_depth, int spp,
                        const hpt_params *params, float *host_image){
    return multi_render(m, W, H, params, host_image, [&](int d, const hpt_params &q){
        return hpt_render_pt_device(m->scene[d], camera, W, H, eye_depth, spp, &q, m->d_local[d], m->stream[d]);
    });
}

int hpt_multi_render_bdpt(hpt_multi *m, const void *camera, int W, int H, int eye_depth, int light_depth, int spp, int spl,
                          const hpt_params *params, float *host_image){
    return multi_render(m, W, H, params, host_image, [&](int d, const hpt_params &q){
        return hpt_render_bdpt_device(m->scene[d], camera, W, H, eye_depth, light_depth, spp, spl, &q, m->d_local[d], m->stream[d]);
    });
}

int hpt_multi_get_timing(const hpt_multi *m, double *render_ms_per_device, double *gather_ms, double *total_ms){
    if(!m) return fail_with(HPT_ERR_INVALID, "null fan-out handle");
    if(render_ms_per_device) for(int d = 0; d < m->n; ++d) render_ms_per_device[d] = m->render_ms[d];
    if(gather_ms) *gather_ms = m->gather_ms;
    if(total_ms) *total_ms = m->total_ms;
    return HPT_OK;
}

} // extern "C"

namespace hpt {

// the fan-out kept by the one-shot wrappers: reused when the next call hands over byte-identical arrays
bool multi_matches(const hpt_multi *m, int n_devices, const void *lights, int nl, const void *spheres, int ns, const void *tris, int nt){
    auto same = [](const std::vector<unsigned char> &kept, const void *given, size_t bytes){
        return kept.size() == bytes && (bytes == 0 || memcmp(kept.data(), given, bytes) == 0);
    };
    return m && m->n == n_devices && nl >= 0 && ns >= 0 && nt >= 0 &&
           same(m->h_lights, lights, (size_t) nl * HPT_LIGHT_BYTES) && same(m->h_spheres, spheres, (size_t) ns * HPT_SPHERE_BYTES) &&
           same(m->h_tris, tris, (size_t) nt * HPT_TRIANGLE_BYTES);
}

} // namespace hpt
