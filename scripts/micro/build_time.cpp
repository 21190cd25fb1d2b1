// Host-only timing of build_host_scene with its phases (development):
//   g++ -O2 -std=c++17 -pthread -ffp-contract=off -DHPT_DEV_TUNING -I path_tracing_amd/csrc -o /tmp/build_time scripts/micro/build_time.cpp path_tracing_amd/csrc/scene_build.cpp
//   /tmp/build_time <triangles.bin of 120-B records>
#include "hpt_scene.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace hpt;
int main(int argc, char **argv){
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long bytes = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> raw((size_t) bytes); if(fread(raw.data(), 1, (size_t) bytes, f) != (size_t) bytes) return 2; fclose(f);
    int nt = (int) (bytes / 120);
    for(int r = 0; r < 3; ++r){
        HostScene hs;
        auto t0 = std::chrono::steady_clock::now();
        const char *err = build_host_scene(nullptr, 0, nullptr, 0, raw.data(), nt, hs);
        auto t1 = std::chrono::steady_clock::now();
        printf("%d tris: total %.1f ms, ms_bvh_build %.1f, nodes %zu depth %d wide %zu %s\n", nt, std::chrono::duration<double, std::milli>(t1 - t0).count(), hs.ms_bvh_build, hs.nodes.size(), hs.bvh_depth, hs.wnodes.size(), err);
    }
}
