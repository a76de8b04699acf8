// main.cpp -- headless render-loop entry point with the shape of the reference's main
// (src/main.cpp:9-31): parse arguments -> Window() -> Scene::parse(-i file) -> Window::mainloop().
// Extra flags drive what the reference hard-codes or leaves to the shader defaults.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "scene.h"
#include "window.h"

using namespace glrt;

static void usage(const char *exe) {
    std::printf("usage: %s -i scene.json [-s N] [--max-depth D] [--spp N] [--frames F] [--frames-in-flight B] [--bvh sah|sah-reinsert|sah-gpu|lbvh|sah-levels-cpu|lbvh-cpu|reference] [--order-by-hits] [--out file.png] [--save-every-frame] [--device G | --gpus N | --devices a,b,..] [--extensions] [--whitted]\n"
                "  -i, --input             scene description (JSON; schema: SURVEY.md Appendix C)            [required]\n"
                "  -s, --sample-per-cycle  accepted for compatibility; like the reference (main.cpp:13) it is not read\n"
                "      --max-depth D       u_maxDepth (default 16, the reference shader's default)\n"
                "      --spp N             samples per pixel per frame, u_nSamples (default 1 as window.cpp:239)\n"
                "      --frames F          frames to accumulate before exiting (default 16)\n"
                "      --frames-in-flight B frames per launch of the render kernel (default 16; same pixels as 1)\n"
                "      --bvh KIND          sah (CPU, default) | sah-reinsert (sah + insertion-based optimisation: seconds to build, config 5 renders 3 percent faster) | sah-gpu (binned SAH built on the GPU) | lbvh (linear BVH built on the GPU) | sah-levels-cpu | lbvh-cpu | reference (the reference host's own tree, its builder restated: exact ties and grazing-ray box misses as under the reference host; a worse tree, never re-ordered)\n"
                "      --order-by-hits     before the first frame, order every fork's children by the closest hits of one calibration frame (config 5: -1 percent per frame)\n"
                "      --out file.png      tonemapped output (default output.png, written after the last frame)\n"
                "      --save-every-frame  write the image after every frame, one frame per launch (the reference's cadence, window.cpp:164)\n"
                "      --device G          HIP device ordinal (default: current)\n"
                "      --gpus N            render on HIP devices 0..N-1: interleaved 8-row stripes, gathered when the image is written\n"
                "      --devices a,b,..    the same with an explicit device list (an ordinal may repeat)\n"
                "      --extensions        accept what the reference does not have: shapes of type \"sphere\" (center, radius) and the material\n"
                "                          \"dielectric\" (ior, tint); parity with the reference is not defined for such scenes\n"
                "      --whitted           with --extensions: Whitted-style transport (direct light at diffuse surfaces, specular bounces only)\n", exe);
}

int main(int argc, char **argv) {
    std::string input, out = "output.png";
    int depth = 16, spp = 1, frames = 16, device = -1, in_flight = 0;
    bool every_frame = false, extensions = false, whitted = false, order_by_hits = false;
    std::vector<int> devices;
    std::string bvh;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&](const char *name) -> const char * {
            if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", name); std::exit(1); }
            return argv[++i];
        };
        if (a == "-i" || a == "--input") input = next("--input");
        else if (a == "-s" || a == "--sample-per-cycle") (void)next("--sample-per-cycle");
        else if (a == "--max-depth") depth = std::atoi(next("--max-depth"));
        else if (a == "--spp") spp = std::atoi(next("--spp"));
        else if (a == "--frames") frames = std::atoi(next("--frames"));
        else if (a == "--frames-in-flight") in_flight = std::atoi(next("--frames-in-flight"));
        else if (a == "--out") out = next("--out");
        else if (a == "--bvh") bvh = next("--bvh");
        else if (a == "--device") device = std::atoi(next("--device"));
        else if (a == "--save-every-frame") every_frame = true;
        else if (a == "--order-by-hits") order_by_hits = true;
        else if (a == "--extensions") extensions = true;
        else if (a == "--whitted") { extensions = true; whitted = true; }
        else if (a == "--gpus") { const int n = std::atoi(next("--gpus")); devices.clear(); for (int k = 0; k < n; k++) devices.push_back(k); }
        else if (a == "--devices") {
            devices.clear();
            for (const char *p = next("--devices"); *p;) { devices.push_back(std::atoi(p)); while (*p && *p != ',') p++; if (*p == ',') p++; }
        }
        else { usage(argv[0]); return 1; }
    }
    if (input.empty()) { usage(argv[0]); return 1; }

    auto window = std::make_unique<Window>();
    if (devices.empty()) window->setDevice(device);
    else window->setDevices(devices);
    window->setMaxDepth(depth);
    window->setSamplesPerFrame(spp);
    window->setFrameLimit(frames);
    if (in_flight > 0) window->setFramesInFlight(in_flight);
    window->setOutput(out, every_frame);
    window->setOrderChildrenByHits(order_by_hits);

    auto scene = std::make_shared<Scene>();
    if (!bvh.empty()) scene->setBvhBuilder(bvh);
    scene->enableExtensions(extensions);
    scene->setWhitted(whitted);
    scene->parse(input);

    window->mainloop(scene);
    std::printf("[INFO] %d frames, %.3f ms per frame (wall time between the last two waits for the device, averaged over the frames issued in between), %llu rays\n", frames, window->lastFrameMs(), window->raysTraced());
    return 0;
}
