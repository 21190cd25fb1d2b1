// Attempt at a reduced form of the k_shade miscompile.  It does NOT trigger it (both builds agree with the host on gfx950 / hipcc 7.2.26015); the
// reproducer that does is scripts/micro/slp_repro.py, which rebuilds the full kernel in its miscompiled form.  Kept as the record of a shape that is safe:
// float3 values assigned inside two nested branches behind a rejection loop, then written word by word to a staging
// area in LDS.  Built with and without -fno-slp-vectorize by slp_reduced.sh; prints how many y components differ
// from the host's evaluation of the same expressions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
struct f3 { float x, y, z; };
__host__ __device__ inline f3 mk(float x, float y, float z){ f3 r; r.x = x; r.y = y; r.z = z; return r; }
__host__ __device__ inline f3 operator+(f3 a, f3 b){ return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__host__ __device__ inline f3 operator-(f3 a, f3 b){ return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ inline f3 operator*(f3 a, float s){ return mk(a.x * s, a.y * s, a.z * s); }
__host__ __device__ inline f3 operator*(f3 a, f3 b){ return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__host__ __device__ inline float dot(f3 a, f3 b){ return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ inline float rnd(uint64_t &s){ uint64_t o = s; s = o * 6364136223846793005ull + 1442695040888963407ull;
    uint32_t xs = (uint32_t) (((o >> 18) ^ o) >> 27), rot = (uint32_t) (o >> 59); uint32_t v = (xs >> rot) | (xs << ((32u - rot) & 31u)); return (float) (v >> 8) * (1.0f / 16777216.0f); }
__host__ __device__ inline void body(int i, const float *in, bool &nee, f3 &wo, f3 &wi, f3 &thr, f3 &p1, f3 &p2, float &lam){
    f3 pos = mk(in[9 * i], in[9 * i + 1], in[9 * i + 2]), nrm = mk(in[9 * i + 3], in[9 * i + 4], in[9 * i + 5]), t = mk(in[9 * i + 6], in[9 * i + 7], in[9 * i + 8]);
    f3 T = mk(nrm.y, nrm.z, nrm.x), B = mk(nrm.z, nrm.x, nrm.y);
    f3 ctxwo = mk(dot(t, T), dot(t, B), dot(t, nrm));
    uint64_t rs = 0x9E3779B97F4A7C15ull * (uint64_t) (i + 1);
    nee = false; wo = wi = thr = p1 = p2 = mk(0, 0, 0); lam = 0;
    if(rnd(rs) < 0.25f){
        f3 ld = mk(0.3f, 0.8f, 0.52f);
        float c = fmaxf(0.0f, dot(nrm, ld));
        if(c > 0.0f){ nee = true; wo = ctxwo; lam = ctxwo.z * 0.5f; thr = t; p1 = pos + nrm * 1e-4f; wi = mk(dot(ld, T), dot(ld, B), dot(ld, nrm)); p2 = pos + ld * 1e4f; }
    } else {
        f3 d;
        do { float a = rnd(rs), b = rnd(rs), c = rnd(rs); d = mk(a, b, c) * 2.0f - mk(1.0f, 1.0f, 1.0f); } while(dot(d, d) >= 1.0f);
        f3 lp = mk(0.1f, 0.9f, 0.2f) + d * 0.05f;
        f3 w = lp - pos; float d2 = dot(w, w), dist = sqrtf(d2); w = mk(w.x / dist, w.y / dist, w.z / dist);
        float cs = fmaxf(0.0f, dot(nrm, w)), cl = fmaxf(0.0f, dot(d, w * -1.0f));
        if(cs > 0.0f && cl > 0.0f){
            if(dot(mk(0, -1, 0), w * -1.0f) >= 0.2f){ nee = true; wo = ctxwo; lam = ctxwo.z * 0.5f; thr = t; p1 = pos + nrm * 1e-4f; wi = mk(dot(w, T), dot(w, B), dot(w, nrm)); p2 = lp + d * 1e-4f; }
        }
    }
}
__global__ void k(const float *in, float *out, int n){
    __shared__ uint32_t stage[16][64];
    int i = blockIdx.x * 64 + threadIdx.x;
    bool nee; f3 wo, wi, thr, p1, p2; float lam;
    body(i < n ? i : 0, in, nee, wo, wi, thr, p1, p2, lam);
    if(i >= n) nee = false;
    unsigned long long m = __ballot(nee);
    if(nee){
        uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
        float v[16] = { wo.x, wo.y, wo.z, wi.x, wi.y, wi.z, lam, thr.x, thr.y, thr.z, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z };
        for(int k2 = 0; k2 < 16; ++k2) stage[k2][slot] = __float_as_uint(v[k2]);
        stage[15][slot] = __float_as_uint(p2.z);
    }
    __syncthreads();
    int cnt = __popcll(m);
    if((int) threadIdx.x < cnt) for(int k2 = 0; k2 < 16; ++k2) out[((size_t) blockIdx.x * 64 + threadIdx.x) * 16 + k2] = __uint_as_float(stage[k2][threadIdx.x]);
}
int main(){
    const int n = 64 * 256;
    std::vector<float> in((size_t) n * 9), out((size_t) n * 16, -1.0f), ref((size_t) n * 16, -1.0f);
    uint64_t s = 12345; for(float &x : in) x = rnd(s) * 2.0f - 1.0f;
    for(int b = 0; b < n / 64; ++b){ int slot = 0; for(int l = 0; l < 64; ++l){ int i = b * 64 + l; bool nee; f3 wo, wi, thr, p1, p2; float lam; body(i, in.data(), nee, wo, wi, thr, p1, p2, lam);
        if(nee){ float v[16] = { wo.x, wo.y, wo.z, wi.x, wi.y, wi.z, lam, thr.x, thr.y, thr.z, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z }; for(int k2 = 0; k2 < 16; ++k2) ref[((size_t) b * 64 + slot) * 16 + k2] = v[k2]; ++slot; } } }
    float *din, *dout; hipMalloc(&din, in.size() * 4); hipMalloc(&dout, out.size() * 4);
    hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dout, out.data(), out.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 64), dim3(64), 0, 0, din, dout, n);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    long bad[16] = {0}, tot = 0; for(size_t r = 0; r < (size_t) n; ++r) if(ref[r * 16] != -1.0f || ref[r * 16 + 1] != -1.0f){ ++tot; for(int k2 = 0; k2 < 16; ++k2) if(fabsf(out[r * 16 + k2] - ref[r * 16 + k2]) > 1e-3f * (1.0f + fabsf(ref[r * 16 + k2]))) ++bad[k2]; }
    printf("records %ld; wrong per word:", tot); for(int k2 = 0; k2 < 16; ++k2) printf(" %ld", bad[k2]); printf("\n");
    return 0;
}
