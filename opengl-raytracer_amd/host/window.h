// window.h -- glrt::Window, the drop-in render-loop class (reference: src/core/window.h:14-58).
// Same public surface -- Window(), mainloop(scene, fps = -1), width(), height() -- and the same
// protected virtual hooks (initialize/render/resize/mouse/keyboard), but headless: no GLFW window,
// no GL context, no ImGui.  render() forwards to the C-ABI HIP layer (include/glrtx.h) where the
// reference's render() issued GL calls (window.cpp:213-318).  Because nothing ever closes a headless
// window, the loop runs for a fixed number of frames (setFrameLimit / GLRT_FRAMES, default 16).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "common.h"
#include "scene.h"

struct glrtx_group;
struct glrtx_params;

namespace glrt {

struct MouseEvent { int button = 0, action = 0, mods = 0; double x = 0, y = 0; };  // event.h stand-in (never raised)

class GLRT_API Window {
public:
    Window();
    virtual ~Window();
    void mainloop(const std::shared_ptr<Scene> &scene, double fps = -1.0);
    int width() const { return width_; }
    int height() const { return height_; }

    // Headless controls (the reference hard-codes these: u_nSamples = 1 at window.cpp:239, u_maxDepth
    // left at the shader default 16, a fresh random u_seed per frame at :226-238, output.png every frame).
    void setFrameLimit(int frames) { frameLimit_ = frames; }
    void setMaxDepth(int depth) { maxDepth_ = depth; }
    void setSamplesPerFrame(int spp) { samplesPerFrame_ = spp; }
    void setOutput(const std::string &file, bool everyFrame = false) { output_ = file; saveEveryFrame_ = everyFrame; }
    void setDevice(int hipDevice) { devices_.assign(1, hipDevice); }
    // Several GPUs of the node (no reference counterpart): the image rows are split into interleaved 8-row stripes, one share per
    // listed HIP device, rendered concurrently and gathered on the first one when a frame is saved (glrtx_group, include/glrtx.h).
    // The same ordinal may be listed more than once.  The image is bit-identical to the single-GPU one.  GLRT_GPUS=N = devices 0..N-1.
    void setDevices(const std::vector<int> &hipDevices) { if (!hipDevices.empty()) devices_ = hipDevices; }
    void setFirstFrame(unsigned f) { frame_ = f; }
    // Profile-guided child order (no reference counterpart; off by default; GLRT_BVH_ORDER=hits): before the first frame one calibration frame of the scene's camera is
    // rendered, the closest hits per triangle are counted by the render kernel (glrtx_hit_histogram) and at every fork of the BVH the child that is hit more often goes
    // into the slot the traversal visits first (glrt_bvh_order_by_hits): config 5 -1 %, config 4 -1.8 %, the Cornell-box scenes +-0.3 % (profiles/r06_hit_order.txt).
    // Exact ties between two triangles may resolve to the other one.
    void setOrderChildrenByHits(bool on) { orderByHits_ = on; }
    // Frames issued per launch of the render kernel (glrtx_render_frames; bit-identical to one launch per frame).
    // Used when no image is written between frames and render() is not overridden per frame; GLRT_FRAMES_IN_FLIGHT.
    void setFramesInFlight(int n) { framesInFlight_ = n < 1 ? 1 : n; }
    // wall-clock ms PER FRAME between the last two waits for the device, averaged over the frames issued in between (with one PNG per run: the whole run, cold first
    // launches included; with --save-every-frame: the last frame).  The device's own time of the last launch is glrtx_stats.kernel_ms_last.
    double lastFrameMs() const { return lastMs_; }
    unsigned long long raysTraced() const;

protected:
    virtual void initialize();
    virtual void render();
    virtual void renderFrames(int n);  // n consecutive frames with a static camera, one launch
    virtual void resize(int width, int height) { resizeDefault(width, height); }
    virtual void mouse(const MouseEvent &) {}
    virtual void keyboard(int, int, int, int) {}

private:
    void frameParams(struct ::glrtx_params &p) const;
    void resizeDefault(int width, int height);
    void resetBuffer();
    void saveCurrentFrame(const std::string &filename, bool overwrite = true) const;
    void noteFallback();

    glrtx_group *grp_ = nullptr;
    std::vector<int> devices_ = {-1};  // -1: the current HIP device
    int width_ = 0, height_ = 0;
    int frameLimit_ = 16, maxDepth_ = 16, samplesPerFrame_ = 1, framesInFlight_ = 16;
    unsigned frame_ = 0;
    bool saveEveryFrame_ = false;
    bool orderByHits_ = false;
    bool fallbackNoted_ = false;
    std::string output_ = "output.png";
    double lastMs_ = 0.0;
    std::shared_ptr<Scene> scene = nullptr;
};

// RGBA8 -> PNG (stored deflate blocks; own encoder, the reference uses stb_image_write).
bool writePng(const std::string &filename, int w, int h, const unsigned char *rgba);

}  // namespace glrt
