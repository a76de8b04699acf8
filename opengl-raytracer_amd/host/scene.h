// scene.h -- glrt::Scene, the drop-in scene class (reference: src/core/scene.h:45-74).
// Same public surface: Scene(), Scene(filename), parse(filename); Window reads the private state as
// a friend exactly like the reference's Window does (scene.h:73).  Where the reference held GL
// TextureBuffer objects (scene.h:63-67) this class holds the flat host buffers in the identical
// byte layout and hands them to glrtx_upload_scene (include/glrtx.h).
#pragma once
#include <string>
#include <vector>

#include "common.h"

namespace glrt {

// Wire-format records, byte-identical to the reference's (scene.h:16-35, trimesh.h:15-25, bvh.h:84-100).
struct Vertex { float pos[3], normal[3], uv[3], tangent[3], binormal[3]; };
struct Triangle { float indices[4]; };  // i, j, k, materialId
enum class MaterialType : int { Emitter = 1, Diffuse = 2, Conductor = 3, Dielectric = 4, Media = 5 };
struct Material { float type[3], emission[3], param0[3], param1[3], param2[3], texIds[3]; };
struct BVHNode { float bboxMin[3], bboxMax[3], children[3]; };

class GLRT_API Scene {
public:
    Scene();
    explicit Scene(const std::string &filename);
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;

    // JSON scene description (schema: SURVEY.md Appendix C, derived from scene.cpp:57-250).
    void parse(const std::string &filename);

    // Programmatic alternative to parse(): adopt already-flat buffers (synthetic scenes, tests).
    void setBuffers(int width, int height, const float viewM[16], const float projM[16], float apertureRadius,
                    float focalLength, std::vector<Vertex> vertices, std::vector<Triangle> triangles,
                    std::vector<Material> materials, std::vector<BVHNode> nodes = {});

    int filmWidth() const { return width; }
    int filmHeight() const { return height; }
    size_t numTriangles() const { return triangles.size(); }
    size_t numLights() const { return lights.size(); }
    size_t numNodes() const { return nodes.size(); }
    int bvhDepth() const { return bvhDepth_; }
    // "sah" (default) | "sah-gpu" / "lbvh" (built on the GPU) | "sah-levels-cpu" / "lbvh-cpu" | "reference" (the reference host's own tree, never re-ordered); call before parse().  GLRT_BVH overrides.
    void setBvhBuilder(const std::string &kind) { bvhBuilder_ = kind; }
    // EXTENSIONS beyond the reference (parity unpinned; include/glrtx.h).  Off, parse() is the reference's: "dielectric" is an
    // unsupported material (FatalError, scene.cpp:216-218) and a shape that is not "obj" contributes no geometry (scene.cpp:222).
    // On (call before parse(); GLRT_EXTENSIONS=1): material "dielectric" {"ior": n, "tint": [r,g,b]} and shape
    // {"type": "sphere", "center": [x,y,z], "radius": r} are accepted; Window then uploads the spheres and enables
    // GLRTX_EXT_DIELECTRIC (and GLRTX_EXT_WHITTED if setWhitted(true)).
    void enableExtensions(bool on) { extensions_ = on; }
    void setWhitted(bool on) { whitted_ = on; }
    size_t numSpheres() const { return spheres.size() / 5; }

private:
    void finalize();  // lights list + BVH (scene.cpp:246-256)

    int width = 0, height = 0;
    float apertureRadius = 0.0f, focalLength = 1.0f;
    float modelM[16], viewM[16], projM[16];  // column-major

    std::vector<Vertex> vertices;
    std::vector<Triangle> triangles;
    std::vector<Triangle> lights;
    std::vector<Material> materials;
    std::vector<BVHNode> nodes;
    int bvhDepth_ = 0;
    std::string bvhBuilder_ = "sah";
    std::vector<float> spheres;  // extension: 5 floats per sphere {cx, cy, cz, radius, material}
    bool extensions_ = false, whitted_ = false, hasDielectric_ = false;

    friend class Window;
    friend struct SceneProbe;
};

// OBJ triangles the way the reference's loader yields them (trimesh.cpp:113-191): three fresh
// vertices per triangle, file normals normalised, otherwise per-vertex face normals (:38-64).
bool loadObj(const std::string &filename, std::vector<Vertex> &out, std::string &err);

}  // namespace glrt
