// pt_cli -- dependency-free clone of the reference's headless front-end for the PT path
// (reference src/main_cli.cpp:42-256): same flags, same banner, same scene grammar, same
// camera (FOV hard-coded to 50 like the reference, main_cli.cpp:158), same 8-bit output stage;
// PNG through zlib instead of OpenCV.  Extra flags the reference lacks:
//   --width/--height  override the scene's R line        --seed N   reproducible streams
//   --max-depth N     eye depth (reference: EYE_DEPTH 4)  --obj FILE append an OBJ's faces (current material: 0.7 grey diffuse)
//   --rr              optional unbiased Russian roulette (pt)
//   --gpus N          render on N devices of this node inside the blocking call (image tiles, RCCL gather)
// --mode pt and --mode bdpt are built (ppm is outside this library); bdpt renders the reference's CPU
// estimator (run_cpu_bdpt) on the GPU.
#include "scene_model.hpp"
#include "../../../include/hpt.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#define LIGHT_DEPTH 4
#define EYE_DEPTH 4

namespace hpt_host { extern hpt_params g_run_params; extern bool g_seed_from_clock; extern int g_devices; }

int main(int argc, char **argv){
    int spp = 8, spl = 8;
    std::string mode = "pt", output_file = "output.png", input_file = "../../input.txt", device = "gpu", obj_file;
    int width = 0, height = 0, max_depth = EYE_DEPTH;
    long long seed = -1;
    for(int i = 1; i < argc; ++i){
        std::string arg = argv[i];
        if(arg == "--spp" && i + 1 < argc) spp = std::stoi(argv[++i]);
        else if(arg == "--spl" && i + 1 < argc) spl = std::stoi(argv[++i]);
        else if(arg == "--mode" && i + 1 < argc) mode = argv[++i];
        else if(arg == "--device" && i + 1 < argc) device = argv[++i];
        else if(arg == "--output" && i + 1 < argc) output_file = argv[++i];
        else if(arg == "--input" && i + 1 < argc) input_file = argv[++i];
        else if(arg == "--width" && i + 1 < argc) width = std::stoi(argv[++i]);
        else if(arg == "--height" && i + 1 < argc) height = std::stoi(argv[++i]);
        else if(arg == "--seed" && i + 1 < argc) seed = std::stoll(argv[++i]);
        else if(arg == "--max-depth" && i + 1 < argc) max_depth = std::stoi(argv[++i]);
        else if(arg == "--obj" && i + 1 < argc) obj_file = argv[++i];
        else if(arg == "--rr") hpt_host::g_run_params.flags |= HPT_FLAG_RUSSIAN_ROULETTE;
        else if(arg == "--gpus" && i + 1 < argc) hpt_host::g_devices = std::max(1, std::stoi(argv[++i]));
        else if(arg == "--help" || arg == "-h"){
            std::cout << "Usage: pt_cli [options]\n"
                      << "Options:\n"
                      << "  --spp <int>       Samples per pixel (default: 8)\n"
                      << "  --spl <int>       Samples per light (default: 8)\n"
                      << "  --mode <string>   Render mode: pt, bdpt (default: pt)\n"
                      << "  --device <string> Compute device: gpu (default: gpu)\n"
                      << "  --output <string> Output image path (.png or .pfm)\n"
                      << "  --input <string>  Input scene file\n"
                      << "  --width/--height <int>  override the scene's R line\n"
                      << "  --seed <int>      reproducible random streams (default: clock)\n"
                      << "  --max-depth <int> eye depth (default: 4)\n"
                      << "  --obj <file>      append the faces of a Wavefront OBJ\n"
                      << "  --rr              unbiased Russian roulette (pt mode; not in the reference, off by default)\n"
                      << "  --gpus <int>      devices of this node to render on (image tiles, RCCL gather; default: 1)\n";
            return 0;
        }
    }
    std::cout << "====================================\n";
    std::cout << " Device : " << device << "\n";
    std::cout << " Mode   : " << mode << "\n";
    std::cout << " SPP    : " << spp << "\n";
    std::cout << " SPL    : " << spl << " (used in BDPT/PPM)\n";
    std::cout << " Input  : " << input_file << "\n";
    std::cout << " Output : " << output_file << "\n";
    std::cout << "====================================\n";
    if(mode != "pt" && mode != "bdpt"){ std::cerr << "[Error] this build provides --mode pt and --mode bdpt (ppm is outside this library).\n"; return -1; }

    hpt_host::SceneFile scene;
    if(!hpt_host::parse_scene_file(input_file, scene)){
        std::cerr << "[Error] Cannot open input file: " << input_file << "\n";
        return -1;
    }
    if(!obj_file.empty()){
        hpt_host::Material grey; grey.base_color = {0.7f, 0.7f, 0.7f}; grey.roughness = 1.0f;
        std::string err;
        int n = hpt_host::append_obj(obj_file, grey, 2, scene, &err);
        if(n < 0){ std::cerr << "[Error] " << err << "\n"; return -1; }
        std::cout << "OBJ triangles: " << n << std::endl;
    }
    std::cout << "[Parse] " << scene.parse_ms << " ms\n";
    std::cout << "Ball:" << std::endl << scene.ball_cnt << std::endl;
    std::cout << "Triangle:" << std::endl << scene.tri_cnt << std::endl;
    std::cout << "Light:" << std::endl << scene.lights.size() << std::endl;

    const int W = width > 0 ? width : scene.resolution.first;
    const int H = height > 0 ? height : scene.resolution.second;
    float F = 50;
    CudaCamera cam = hpt_host::make_cuda_camera(scene.camera, F, W, H);
    std::vector<float3> frame_results((size_t) W * H);

    std::cout << "[Init] Transferring Data to the GPU...\n";
    if(mode == "bdpt") move_data_to_cuda_bdpt(scene.groups(), scene.lights, spl);
    else move_data_to_cuda_pt(scene.groups(), scene.lights, spl);
    if(seed >= 0){ hpt_host::g_seed_from_clock = false; hpt_host::g_run_params.seed = (uint64_t) seed; }

    std::cout << "[Render] Starting Render...\n";
    auto start_time = std::chrono::steady_clock::now();
    if(mode == "bdpt") run_cuda_bdpt(cam, frame_results.data(), LIGHT_DEPTH, max_depth, W, H, spp, spl);
    else run_cuda_pt(cam, frame_results.data(), LIGHT_DEPTH, max_depth, W, H, spp);
    std::cout << "\n";
    auto diff = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start_time);
    std::cout << "[Render] Finished in " << diff.count() << " ms.\n";

    std::cout << "[Save] Writing to " << output_file << "...\n";
    std::string err;
    if(hpt_host::write_image(output_file, frame_results.data(), W, H, &err)) std::cout << "[Success] Image saved!\n";
    else std::cerr << "[Error] Failed to save image.\n";
    return 0;
}
