// window.cpp -- headless glrt::Window over the C-ABI HIP layer.  Structure follows the reference's
// Window (src/core/window.cpp): mainloop :105-182, render :213-318, resetBuffer :366-381,
// saveCurrentFrame :383-414; every GL call is replaced by its glrtx_* counterpart.
#include "window.h"

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#include "glrt_host.h"
#include "glrtx.h"

namespace glrt {

#define GLRTX_CHECK(call)                                                                        \
    do {                                                                                         \
        if ((call) != GLRTX_OK) GLRT_FatalError("%s: %s", #call, glrtx_group_last_error(grp_));  \
    } while (0)

Window::Window() {
    if (const char *e = std::getenv("GLRT_FRAMES")) frameLimit_ = std::atoi(e);
    if (const char *e = std::getenv("GLRT_MAX_DEPTH")) maxDepth_ = std::atoi(e);
    if (const char *e = std::getenv("GLRT_FRAMES_IN_FLIGHT")) setFramesInFlight(std::atoi(e));
    if (const char *e = std::getenv("GLRT_GPUS")) {
        const int n = std::atoi(e);
        if (n > 1) { devices_.clear(); for (int i = 0; i < n; i++) devices_.push_back(i); }
    }
}

Window::~Window() {
    if (grp_) glrtx_group_destroy(grp_);
}

unsigned long long Window::raysTraced() const {
    glrtx_stats st;
    if (!grp_ || glrtx_group_get_stats(grp_, &st) != GLRTX_OK) return 0;
    return st.rays;
}

void Window::mainloop(const std::shared_ptr<Scene> &scene_, double fps) {
    (void)fps;  // the reference's default fps = -1 renders every iteration (window.cpp:126); so do we
    scene = scene_;
    if (!grp_ && glrtx_group_create(&grp_, devices_.data(), (int)devices_.size()) != GLRTX_OK)
        GLRT_FatalError("glrtx_group_create: %s", glrtx_group_last_error(nullptr));
    // scene upload: the five buffers Scene::parse handed to TextureBuffer::setData (scene.cpp:254-269)
    GLRTX_CHECK(glrtx_group_upload_scene(
        grp_, scene->vertices.empty() ? nullptr : &scene->vertices[0].pos[0], scene->vertices.size(),
        scene->triangles.empty() ? nullptr : &scene->triangles[0].indices[0], scene->triangles.size(),
        scene->materials.empty() ? nullptr : &scene->materials[0].type[0], scene->materials.size(),
        scene->lights.empty() ? nullptr : &scene->lights[0].indices[0], scene->lights.size(),
        scene->nodes.empty() ? nullptr : &scene->nodes[0].bboxMin[0], scene->nodes.size()));
    // extensions the scene asked for (Scene::enableExtensions; none in a reference scene): analytic spheres, dielectric, Whitted
    if (!scene->spheres.empty() || scene->hasDielectric_ || scene->whitted_) {
        const int flags = (scene->hasDielectric_ ? GLRTX_EXT_DIELECTRIC : 0) | (scene->whitted_ ? GLRTX_EXT_WHITTED : 0);
        for (int i = 0; i < glrtx_group_size(grp_); i++) {
            glrtx_ctx *c = glrtx_group_ctx(grp_, i);
            if (glrtx_upload_spheres(c, scene->spheres.empty() ? nullptr : scene->spheres.data(), scene->spheres.size() / 5) != GLRTX_OK ||
                glrtx_set_extensions(c, flags) != GLRTX_OK)
                GLRT_FatalError("extensions: %s", glrtx_last_error(c));
        }
        GLRT_Info("extensions: %zu analytic spheres%s%s (not part of the reference)", scene->spheres.size() / 5,
                  scene->hasDielectric_ ? ", dielectric" : "", scene->whitted_ ? ", Whitted termination" : "");
    }
    resize(scene->width, scene->height);
    if (const char *e = std::getenv("GLRT_BVH_ORDER")) orderByHits_ = std::string(e) == "hits";
    if (orderByHits_ && !scene->nodes.empty() && scene->spheres.empty() && !scene->hasDielectric_ && !scene->whitted_) {
        // one calibration frame on the first member's share of the rows (interleaved stripes: a fair sample of the image), counted by the render kernel itself
        glrtx_params p;
        frameParams(p);
        glrt_frame_seed(0x9e3779b9u, p.seed);
        std::vector<uint32_t> hist(scene->triangles.size(), 0u);
        glrtx_ctx *c0 = glrtx_group_ctx(grp_, 0);
        if (glrtx_hit_histogram(c0, &p, hist.data(), hist.size()) != GLRTX_OK) GLRT_FatalError("glrtx_hit_histogram: %s", glrtx_last_error(c0));
        if (glrt_bvh_add_shadow_hits(hist.data(), hist.size(), &scene->triangles[0].indices[0], &scene->materials[0].type[0], scene->materials.size()) < 0)
            GLRT_FatalError("glrt_bvh_add_shadow_hits failed");
        const int exchanged = glrt_bvh_order_by_hits(&scene->nodes[0].bboxMin[0], scene->nodes.size(), hist.data(), hist.size());
        if (exchanged < 0) GLRT_FatalError("glrt_bvh_order_by_hits failed (%d)", exchanged);
        GLRT_Info("BVH: children ordered by the hits of a calibration frame, %d forks exchanged", exchanged);
        GLRTX_CHECK(glrtx_group_upload_scene(
            grp_, &scene->vertices[0].pos[0], scene->vertices.size(), &scene->triangles[0].indices[0], scene->triangles.size(),
            &scene->materials[0].type[0], scene->materials.size(), scene->lights.empty() ? nullptr : &scene->lights[0].indices[0], scene->lights.size(),
            &scene->nodes[0].bboxMin[0], scene->nodes.size()));
    }
    initialize();
    // The reference presents (and saves) every frame; when only the final image is wanted the frames of a static
    // camera go to the device several at a time -- same pixels, bit for bit, fewer and fuller launches.
    const bool every = saveEveryFrame_ && !output_.empty();
    const int step = every ? 1 : framesInFlight_;
    // Launches are issued back to back and the loop waits for the device only where it needs the pixels (a frame that is saved, the end of the run): the calls are
    // asynchronous, and a launch issued behind idle time runs longer -- 2 % behind 1 ms, 5 % behind 3 ms (profiles/r04_ab_launch_warmth.txt).  lastFrameMs() is the wall
    // time per frame between two such waits.
    auto t0 = std::chrono::steady_clock::now();
    int since = 0;
    for (int i = 0; i < frameLimit_; i += step) {
        const int n = frameLimit_ - i < step ? frameLimit_ - i : step;
        if (n == 1) render();
        else renderFrames(n);
        since += n;
        if (every || i + step >= frameLimit_) {
            GLRTX_CHECK(glrtx_group_sync(grp_));
            lastMs_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / since;
            if (every) saveCurrentFrame(output_, true);  // window.cpp:164
            t0 = std::chrono::steady_clock::now();
            since = 0;
        }
    }
    if (!every && !output_.empty() && frameLimit_ > 0) saveCurrentFrame(output_, true);
}

void Window::initialize() {
    for (int i = 0; i < glrtx_group_size(grp_); i++)
        if (glrtx_count_rays(glrtx_group_ctx(grp_, i), 1) != GLRTX_OK) GLRT_FatalError("glrtx_count_rays failed");
}

void Window::frameParams(glrtx_params &p) const {
    // window.cpp:230-243: the per-frame uniforms (u_seed is filled in by the caller)
    float cam[16];
    glrt_mat4_mul(scene->viewM, scene->modelM, cam);
    if (glrt_mat4_inverse(cam, p.c2w) != GLRT_HOST_OK || glrt_mat4_inverse(scene->projM, p.s2c) != GLRT_HOST_OK)
        GLRT_FatalError("camera matrix is singular");
    p.aperture = scene->apertureRadius;
    p.focal = scene->focalLength;
    p.seed[0] = p.seed[1] = 0.0f;
    p.n_samples = samplesPerFrame_;
    p.max_depth = maxDepth_;
}

void Window::render() {
    glrtx_params p;
    frameParams(p);
    glrt_frame_seed(frame_++, p.seed);
    GLRTX_CHECK(glrtx_group_render(grp_, &p));  // window.cpp:290, the draw that runs the path tracer
    noteFallback();
}

// Once per run: say so when launches leave the wavefront kernel for the (slower) persistent megakernel (glrtx_stats.fallback_last).
void Window::noteFallback() {
    if (fallbackNoted_) return;
    glrtx_stats st;
    if (glrtx_group_get_stats(grp_, &st) != GLRTX_OK || st.fallback_launches == 0) return;
    fallbackNoted_ = true;
    GLRT_Info("Render kernel: persistent megakernel instead of the wavefront kernel (%s%s%s): about half the rays per second, no frames in flight",
              (st.fallback_last & GLRTX_FALLBACK_DEPTH) ? "u_maxDepth > 255 " : "", (st.fallback_last & GLRTX_FALLBACK_SAMPLES) ? "u_nSamples >= 2^20 " : "",
              (st.fallback_last & GLRTX_FALLBACK_EXTENSIONS) ? "extension scene" : "");
}

void Window::renderFrames(int n) {
    glrtx_params p;
    frameParams(p);
    std::vector<float> seeds(2 * (size_t)n);
    for (int f = 0; f < n; f++) glrt_frame_seed(frame_++, &seeds[2 * (size_t)f]);
    GLRTX_CHECK(glrtx_group_render_frames(grp_, &p, seeds.data(), n));
    noteFallback();
}

void Window::resizeDefault(int w, int h) {
    width_ = w;
    height_ = h;
    resetBuffer();
}

void Window::resetBuffer() { GLRTX_CHECK(glrtx_group_resize(grp_, width_, height_)); }  // window.cpp:366-381

void Window::saveCurrentFrame(const std::string &filename, bool overwrite) const {
    std::vector<unsigned char> bytes((size_t)width_ * height_ * 4);
    // resolve = screen.frag (rgb/count, clamp, gamma 2.2) + the vertical flip of window.cpp:391-398
    // (with several GPUs the stripes are first gathered on the first one)
    if (glrtx_group_resolve_rgba8(grp_, bytes.data(), (size_t)width_ * 4, 2.2f, 1) != GLRTX_OK)
        GLRT_FatalError("glrtx_group_resolve_rgba8: %s", glrtx_group_last_error(grp_));
    std::string path = filename;
    if (!overwrite) {
        int count = 0;
        const size_t dot = filename.find_last_of('.');
        const std::string base = filename.substr(0, dot), ext = dot == std::string::npos ? "" : filename.substr(dot);
        while (std::ifstream(path).good()) path = base + "_" + std::to_string(count++) + ext;
    }
    if (!writePng(path, width_, height_, bytes.data())) GLRT_Warn("Failed to save: %s", path.c_str());
    else GLRT_Info("Save: %s", path.c_str());
}

// ---------------------------------------------------------------------------------------------- PNG
namespace {
uint32_t crc32(const unsigned char *d, size_t n, uint32_t c = 0) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t r = i;
            for (int k = 0; k < 8; k++) r = (r & 1) ? 0xEDB88320u ^ (r >> 1) : r >> 1;
            table[i] = r;
        }
        init = true;
    }
    c = ~c;
    for (size_t i = 0; i < n; i++) c = table[(c ^ d[i]) & 0xFF] ^ (c >> 8);
    return ~c;
}
void be32(std::vector<unsigned char> &v, uint32_t x) {
    v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16));
    v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}
void chunk(std::vector<unsigned char> &out, const char *type, const std::vector<unsigned char> &data) {
    be32(out, (uint32_t)data.size());
    std::vector<unsigned char> td(type, type + 4);
    td.insert(td.end(), data.begin(), data.end());
    out.insert(out.end(), td.begin(), td.end());
    be32(out, crc32(td.data(), td.size()));
}
}  // namespace

bool writePng(const std::string &filename, int w, int h, const unsigned char *rgba) {
    std::vector<unsigned char> raw;
    raw.reserve((size_t)h * ((size_t)w * 4 + 1));
    for (int y = 0; y < h; y++) {
        raw.push_back(0);  // filter: none
        raw.insert(raw.end(), rgba + (size_t)y * w * 4, rgba + (size_t)(y + 1) * w * 4);
    }
    std::vector<unsigned char> z = {0x78, 0x01};  // zlib header, stored blocks
    uint32_t a = 1, b = 0;
    for (unsigned char c : raw) { a = (a + c) % 65521u; b = (b + a) % 65521u; }
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n >= raw.size() ? 1 : 0);
        z.push_back((unsigned char)(n & 0xFF)); z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xFF)); z.push_back((unsigned char)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + (long)off, raw.begin() + (long)(off + n));
        if (raw.empty()) break;
    }
    be32(z, (b << 16) | a);
    std::vector<unsigned char> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<unsigned char> ihdr;
    be32(ihdr, (uint32_t)w); be32(ihdr, (uint32_t)h);
    ihdr.insert(ihdr.end(), {8, 6, 0, 0, 0});  // 8-bit RGBA
    chunk(png, "IHDR", ihdr);
    chunk(png, "IDAT", z);
    chunk(png, "IEND", {});
    std::ofstream f(filename.c_str(), std::ios::binary);
    if (f.fail()) return false;
    f.write((const char *)png.data(), (std::streamsize)png.size());
    return f.good();
}

}  // namespace glrt
