// scene.cpp -- glrt::Scene::parse and helpers.  Behaviour follows the reference's parser
// (src/core/scene.cpp:31-276) key by key: same required keys (abort on absence), same defaults and
// warnings for optional ones, one material per shape, material id = shape order, light list =
// triangles of shapes with non-zero emission.  Parsers are own code (json.h, loadObj below).
#include "scene.h"

#include <cmath>
#include <cstring>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "glrt_host.h"
#include "glrtx.h"
#include "json.h"

namespace glrt {

namespace {
std::string dirOf(const std::string &path) {
    const size_t p = path.find_last_of("/\\");
    return p == std::string::npos ? std::string(".") : path.substr(0, p);
}
void vec3(const Json &j, float out[3]) {
    for (int k = 0; k < 3; k++) out[k] = (float)j[(size_t)k].number_value();
}
void fill3(float out[3], float v) { out[0] = out[1] = out[2] = v; }
}  // namespace

Scene::Scene() {
    std::memset(modelM, 0, sizeof modelM);
    std::memset(viewM, 0, sizeof viewM);
    std::memset(projM, 0, sizeof projM);
    for (int i = 0; i < 4; i++) modelM[i * 5] = viewM[i * 5] = projM[i * 5] = 1.0f;
}

Scene::Scene(const std::string &filename) : Scene() { parse(filename); }

void Scene::parse(const std::string &filename) {
    std::ifstream reader(filename.c_str(), std::ios::in);
    if (reader.fail()) GLRT_FatalError("Failed to open file: %s", filename.c_str());
    std::stringstream ss;
    ss << reader.rdbuf();
    std::string err;
    const Json json = Json::parse(ss.str(), err);
    if (!err.empty()) GLRT_Warn("%s", err.c_str());
    const std::string baseDir = dirOf(filename);
    if (const char *e = std::getenv("GLRT_EXTENSIONS")) extensions_ = extensions_ || std::atoi(e) != 0;
    spheres.clear();
    hasDielectric_ = false;

    // film (scene.cpp:57-60)
    width = json["film"]["width"].int_value();
    height = json["film"]["height"].int_value();
    GLRT_Info("window: %d x %d", width, height);

    // camera (scene.cpp:62-114)
    const std::string type = json["camera"]["type"].string_value();
    GLRT_Info("Camera type: %s", type.c_str());
    if (type == "perspective") {
        const Json &cam = json["camera"];
        apertureRadius = (float)cam["apertureRadius"].number_value();  // absent -> 0
        focalLength = (float)cam["focalLength"].number_value();        // absent -> 0 (scene.cpp:71-74)
        if (cam["lookAt"].is_null()) GLRT_FatalError("perspective camera node does not have \"lookAt\" key!");
        float origin[3], target[3], up[3];
        vec3(cam["lookAt"]["origin"], origin);
        vec3(cam["lookAt"]["target"], target);
        vec3(cam["lookAt"]["up"], up);
        glrt_look_at(origin, target, up, viewM);
        if (cam["fov"].is_null()) GLRT_FatalError("perspective camera node does not have \"fov\" key!");
        if (cam["nearClip"].is_null()) GLRT_FatalError("perspective camera node does not have \"nearClip\" key!");
        if (cam["farClip"].is_null()) GLRT_FatalError("perspective camera node does not have \"farClip\" key!");
        glrt_perspective((float)cam["fov"].number_value(), (float)width / (float)height,
                         (float)cam["nearClip"].number_value(), (float)cam["farClip"].number_value(), projM);
    }

    // shapes (scene.cpp:116-250)
    vertices.clear(); triangles.clear(); lights.clear(); materials.clear(); nodes.clear();
    const auto &shapes = json["scene"].array_items();
    for (size_t i = 0; i < shapes.size(); i++) {
        const Json &sh = shapes[i];
        const std::string material = sh["material"].string_value();
        Material m;
        std::memset(&m, 0, sizeof m);
        if (material == "diffuse") {
            fill3(m.type, (float)MaterialType::Diffuse);
            if (sh["reflectance"].is_null()) { GLRT_Warn("diffuse node does not have \"reflectance\" key!"); fill3(m.param0, 0.5f); }
            else vec3(sh["reflectance"], m.param0);
        } else if (material == "conductor") {
            fill3(m.type, (float)MaterialType::Conductor);
            if (sh["kappa"].is_null()) { GLRT_Warn("conductor node does not have \"kappa\" key!"); fill3(m.param0, 1.0f); }
            else vec3(sh["kappa"], m.param0);
            if (sh["eta"].is_null()) { GLRT_Warn("conductor node does not have \"eta\" key!"); fill3(m.param1, 1.0f); }
            else vec3(sh["eta"], m.param1);
            if (sh["alpha"].is_null()) { GLRT_Warn("conductor node does not have \"alpha\" key!"); fill3(m.param2, 0.0f); }
            else fill3(m.param2, (float)sh["alpha"].number_value());
        } else if (material == "emitter") {
            fill3(m.type, (float)MaterialType::Emitter);
            if (sh["emission"].is_null()) GLRT_Warn("emitter node does not have \"emission\" key!");
            else vec3(sh["emission"], m.emission);
        } else if (material == "media") {
            // The reference uploads two 3D textures here, but its shader's volume branch is compiled
            // out (raytrace.frag:4, :424-487): the material only marks the surface as pass-through.
            fill3(m.type, (float)MaterialType::Media);
        } else if (material == "dielectric" && extensions_) {
            // EXTENSION (no reference counterpart): MTRL_DIELECTRIC = 4 (raytrace.frag:32); param0 = tint, param1.x = index of refraction
            fill3(m.type, 4.0f);
            if (sh["tint"].is_null()) fill3(m.param0, 1.0f);
            else vec3(sh["tint"], m.param0);
            fill3(m.param1, 0.0f);
            m.param1[0] = sh["ior"].is_null() ? 1.5f : (float)sh["ior"].number_value();
            hasDielectric_ = true;
        } else {
            GLRT_FatalError("Unsupported material: %s", material.c_str());
        }
        materials.push_back(m);

        if (sh["type"].string_value() == "obj") {
            const std::string file = baseDir + "/" + sh["filename"].string_value();
            std::vector<Vertex> mesh;
            std::string oerr;
            if (!loadObj(file, mesh, oerr)) GLRT_FatalError("Failed to load *.obj file: %s (%s)", file.c_str(), oerr.c_str());
            const size_t base = vertices.size();
            vertices.insert(vertices.end(), mesh.begin(), mesh.end());
            for (size_t t = 0; t + 2 < mesh.size(); t += 3) {
                Triangle tri;
                tri.indices[0] = (float)(base + t);
                tri.indices[1] = (float)(base + t + 1);
                tri.indices[2] = (float)(base + t + 2);
                tri.indices[3] = (float)(materials.size() - 1);
                triangles.push_back(tri);
            }
        } else if (sh["type"].string_value() == "sphere" && extensions_) {  // EXTENSION: analytic sphere
            float ctr[3] = {0.f, 0.f, 0.f};
            if (!sh["center"].is_null()) vec3(sh["center"], ctr);
            const float radius = sh["radius"].is_null() ? 1.0f : (float)sh["radius"].number_value();
            if (!(radius > 0.0f)) GLRT_FatalError("sphere radius must be positive");
            spheres.insert(spheres.end(), {ctr[0], ctr[1], ctr[2], radius, (float)(materials.size() - 1)});
        }
    }
    finalize();
    GLRT_Info("Scene setup OK!");
    GLRT_Info("#vertex: %d", (int)vertices.size());
    GLRT_Info("#triangle: %d", (int)triangles.size());
    GLRT_Info("#BVH node: %d", (int)nodes.size());
}

void Scene::setBuffers(int w, int h, const float view[16], const float proj[16], float aperture, float focal,
                       std::vector<Vertex> v, std::vector<Triangle> t, std::vector<Material> m, std::vector<BVHNode> n) {
    width = w; height = h; apertureRadius = aperture; focalLength = focal;
    std::memcpy(viewM, view, sizeof viewM);
    std::memcpy(projM, proj, sizeof projM);
    vertices = std::move(v); triangles = std::move(t); materials = std::move(m); nodes = std::move(n);
    finalize();
}

void Scene::finalize() {
    lights.clear();
    for (const Triangle &t : triangles) {
        const size_t m = (size_t)t.indices[3];
        if (m < materials.size()) {
            const float *e = materials[m].emission;
            if (std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) != 0.0f) lights.push_back(t);  // scene.cpp:246-248
        }
    }
    if (nodes.empty() && !triangles.empty()) {
        // BVH::construct (scene.cpp:251-253 -> bvh.cpp:59-160).  Builders: "sah" (CPU, default), "sah-gpu" (binned SAH by levels + exact sweep below, built on the
        // GPU through glrtx_build_bvh_sah: the CPU SAH tree's quality in ~1.3 ms per 100 k triangles), "lbvh" (linear BVH
        // built on the GPU through glrtx_build_lbvh: 20 % faster to build, ~3 % more traversal steps), "sah-levels-cpu" / "lbvh-cpu" (the same trees from the host library),
        // "sah-reinsert" ("sah" + glrt_bvh_reinsert: config 5 renders 3 % faster, the build takes seconds), "reference" (the reference host's OWN tree, its builder
        // restated rule for rule -- glrt_host.h: glrt_bvh_build_reference -- and left in its own child order: exact ties and grazing-ray box misses then fall where the
        // reference host's do).  Any of them renders the same image except at those two kinds of pixels.
        std::string kind = bvhBuilder_;
        if (const char *e = std::getenv("GLRT_BVH")) kind = e;
        nodes.resize(glrt_bvh_node_count(triangles.size()));
        const float *v = &vertices[0].pos[0], *t = &triangles[0].indices[0];
        if (kind == "sah" || kind == "sah-reinsert" || kind == "lbvh-cpu" || kind == "sah-levels-cpu" || kind == "reference") {
            const int rc = kind == "sah" || kind == "sah-reinsert" ? glrt_bvh_build_sah(v, vertices.size(), t, triangles.size(), &nodes[0].bboxMin[0], &bvhDepth_)
                         : kind == "reference" ? glrt_bvh_build_reference(v, vertices.size(), t, triangles.size(), &nodes[0].bboxMin[0], &bvhDepth_)
                         : kind == "lbvh-cpu" ? glrt_bvh_build_lbvh(v, vertices.size(), t, triangles.size(), &nodes[0].bboxMin[0], &bvhDepth_)
                                              : glrt_bvh_build_sah_levels(v, vertices.size(), t, triangles.size(), &nodes[0].bboxMin[0], &bvhDepth_);
            if (rc != GLRT_HOST_OK) GLRT_FatalError("BVH construction (%s) failed (%d)", kind.c_str(), rc);
            if (kind == "sah-reinsert") {  // insertion-based optimisation of the finished tree (glrt_host.h: glrt_bvh_reinsert): ~3 % fewer box visits on config 5, seconds of CPU
                double cost[2] = {0.0, 0.0};
                int depth = -1;
                const int moved = glrt_bvh_reinsert(&nodes[0].bboxMin[0], nodes.size(), 8, &depth, cost);
                if (moved == GLRT_HOST_EDEPTH) {  // the optimised tree would not fit the traversal stack: the pass has put the SAH tree back (glrt_host.h)
                    GLRT_Info("BVH: reinsertion would deepen the tree to %d levels (the traversal stack holds 64 entries): keeping the SAH tree", depth);
                    depth = -1;
                } else if (moved < 0) GLRT_FatalError("glrt_bvh_reinsert failed (%d)", moved);
                if (depth >= 0) bvhDepth_ = depth;
                if (moved >= 0) GLRT_Info("BVH: %d subtrees reinserted, summed fork area %.2f -> %.2f root areas (depth %d)", moved, cost[0], cost[1], bvhDepth_);
            }
        } else if (kind == "lbvh" || kind == "sah-gpu") {
            glrtx_ctx *ctx = nullptr;
            if (glrtx_create(&ctx, -1) != GLRTX_OK) GLRT_FatalError("GPU BVH builder: %s", glrtx_last_error(nullptr));
            float ms = 0.0f;
            const int rc = kind == "lbvh" ? glrtx_build_lbvh(ctx, v, vertices.size(), t, triangles.size(), &nodes[0].bboxMin[0], &bvhDepth_, &ms)
                                          : glrtx_build_bvh_sah(ctx, v, vertices.size(), t, triangles.size(), &nodes[0].bboxMin[0], &bvhDepth_, &ms);
            if (rc != GLRTX_OK) GLRT_FatalError("GPU BVH builder (%s): %s", kind.c_str(), glrtx_last_error(ctx));
            GLRT_Info("%s over %zu triangles built on the GPU in %.3f ms (depth %d)", kind == "lbvh" ? "LBVH" : "SAH tree", triangles.size(), ms, bvhDepth_);
            glrtx_destroy(ctx);
        } else {
            GLRT_FatalError("unknown BVH builder '%s' (sah | sah-reinsert | sah-gpu | lbvh | sah-levels-cpu | lbvh-cpu | reference)", kind.c_str());
        }
        // the light side first: at a fork where only one child holds emitting triangles that child is visited first (glrt_host.h: glrt_bvh_lights_first)
        const char *lf = std::getenv("GLRT_BVH_LIGHTS_FIRST");
        if (kind == "reference") {
            GLRT_Info("BVH: the reference host's own tree (bvh.cpp:72-160 restated), left in its own child order (depth %d)", bvhDepth_);
        } else if (!(lf && lf[0] == '0' && lf[1] == 0) && !materials.empty()) {
            const int sw = glrt_bvh_lights_first(&nodes[0].bboxMin[0], nodes.size(), t, triangles.size(), &materials[0].type[0], materials.size());
            if (sw < 0) GLRT_FatalError("glrt_bvh_lights_first failed (%d)", sw);
            if (sw > 0) GLRT_Info("BVH: the light side first at %d forks", sw);
        }
    }
}

// ------------------------------------------------------------------------------------------------ OBJ
bool loadObj(const std::string &filename, std::vector<Vertex> &out, std::string &err) {
    std::ifstream in(filename.c_str());
    if (in.fail()) { err = "cannot open"; return false; }
    std::vector<float> P, N, T;
    struct Corner { int v, t, n; };
    std::vector<Corner> corners;  // 3 per triangle
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag) || tag[0] == '#') continue;
        if (tag == "v") { float x, y, z; ls >> x >> y >> z; P.insert(P.end(), {x, y, z}); }
        else if (tag == "vn") { float x, y, z; ls >> x >> y >> z; N.insert(N.end(), {x, y, z}); }
        else if (tag == "vt") { float u = 0, v = 0; ls >> u >> v; T.insert(T.end(), {u, v}); }
        else if (tag == "f") {
            std::vector<Corner> poly;
            std::string tok;
            while (ls >> tok) {
                Corner c{-1, -1, -1};
                int *slot[3] = {&c.v, &c.t, &c.n};
                size_t s = 0;
                for (int k = 0; k < 3 && s <= tok.size(); k++) {
                    const size_t e = tok.find('/', s);
                    const std::string part = tok.substr(s, e == std::string::npos ? std::string::npos : e - s);
                    if (!part.empty()) {
                        const int idx = std::atoi(part.c_str());
                        const int count = (int)((k == 0 ? P.size() / 3 : k == 1 ? T.size() / 2 : N.size() / 3));
                        *slot[k] = idx > 0 ? idx - 1 : count + idx;  // negative = relative to the end
                    }
                    if (e == std::string::npos) break;
                    s = e + 1;
                }
                poly.push_back(c);
            }
            for (size_t k = 1; k + 1 < poly.size(); k++) {  // fan triangulation
                corners.push_back(poly[0]); corners.push_back(poly[k]); corners.push_back(poly[k + 1]);
            }
        }
    }
    bool hasNorm = !corners.empty(), hasUV = !corners.empty();
    out.clear();
    out.reserve(corners.size());
    for (const Corner &c : corners) {
        Vertex v;
        std::memset(&v, 0, sizeof v);
        if (c.v < 0 || (size_t)c.v * 3 + 2 >= P.size()) { err = "vertex index out of range"; return false; }
        std::memcpy(v.pos, &P[(size_t)c.v * 3], 12);
        if (c.n >= 0 && (size_t)c.n * 3 + 2 < N.size()) {
            const float *n = &N[(size_t)c.n * 3];
            const float l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            for (int k = 0; k < 3; k++) v.normal[k] = n[k] / l;  // normalised on load (trimesh.cpp:162)
        } else {
            hasNorm = false;
        }
        if (c.t >= 0 && (size_t)c.t * 2 + 1 < T.size()) { v.uv[0] = T[(size_t)c.t * 2]; v.uv[1] = T[(size_t)c.t * 2 + 1]; }
        else hasUV = false;
        out.push_back(v);
    }
    if (!hasNorm) {  // face normals accumulated per vertex (trimesh.cpp:38-64); vertices are not shared
        for (size_t t = 0; t + 2 < out.size(); t += 3) {
            const float *a = out[t].pos, *b = out[t + 1].pos, *c = out[t + 2].pos;
            const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
            float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            const float l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            for (int k = 0; k < 3; k++) n[k] = l != 0.0f ? n[k] / l : 0.0f;
            for (int q = 0; q < 3; q++) std::memcpy(out[t + q].normal, n, 12);
        }
    }
    if (hasUV) {
        // Tangent frame from the texture coordinates (trimesh.cpp:67-110), carried in texels 3 and 4 of the vertex record.  The path tracer never reads them
        // (raytrace.frag:316-321 fetches texels 0 and 1 only); they are filled so that the vertex buffer a Scene hands over is the reference's record for record.
        // Per face: the reference's (-dP1 dv2 + dP2 dv1) / det and (-dP2 du1 + dP1 du2) / det -- the NEGATED derivatives of the position along u and v, as written there --, each normalised; a vertex sums its faces' (here: its one face's -- vertices are not shared) and
        // normalises the sums where both are non-zero.  A face whose texture coordinates are collinear (zero determinant) contributes nothing.
        for (size_t t = 0; t + 2 < out.size(); t += 3) {
            const Vertex &a = out[t], &b = out[t + 1], &c = out[t + 2];
            const float p1[3] = {b.pos[0] - a.pos[0], b.pos[1] - a.pos[1], b.pos[2] - a.pos[2]}, p2[3] = {c.pos[0] - a.pos[0], c.pos[1] - a.pos[1], c.pos[2] - a.pos[2]};
            const float du1 = b.uv[0] - a.uv[0], dv1 = b.uv[1] - a.uv[1], du2 = c.uv[0] - a.uv[0], dv2 = c.uv[1] - a.uv[1];
            const float det = du1 * dv2 - dv1 * du2;
            if (det == 0.0f) continue;
            float tg[3], bn[3];
            for (int k = 0; k < 3; k++) {
                tg[k] = (-p1[k] * dv2 + p2[k] * dv1) / det;
                bn[k] = (-p2[k] * du1 + p1[k] * du2) / det;
            }
            const float lt = std::sqrt(tg[0] * tg[0] + tg[1] * tg[1] + tg[2] * tg[2]), lb = std::sqrt(bn[0] * bn[0] + bn[1] * bn[1] + bn[2] * bn[2]);
            for (int q = 0; q < 3; q++)
                for (int k = 0; k < 3; k++) { out[t + q].tangent[k] += tg[k] / lt; out[t + q].binormal[k] += bn[k] / lb; }
        }
        for (Vertex &v : out) {
            const float lt = std::sqrt(v.tangent[0] * v.tangent[0] + v.tangent[1] * v.tangent[1] + v.tangent[2] * v.tangent[2]);
            const float lb = std::sqrt(v.binormal[0] * v.binormal[0] + v.binormal[1] * v.binormal[1] + v.binormal[2] * v.binormal[2]);
            if (lt > 0.0f && lb > 0.0f)
                for (int k = 0; k < 3; k++) { v.tangent[k] /= lt; v.binormal[k] /= lb; }
        }
    }
    return true;
}

}  // namespace glrt

// ---------------------------------------------------------------------------------------------- test hook
// Host-only probe for the parser tests (no GPU): parses a JSON scene with the CPU builders and reports what Window
// would upload.  counts = {width, height, vertices, triangles, materials, lights, nodes, bvh depth};
// buffers (any may be NULL) receive the flat arrays in the wire format.
namespace glrt {
struct SceneProbe {
    static int run(const char *json, const char *bvh_kind, long long counts[8], float view[16], float proj[16], float lens[2], float *vert,
                   float *tri, float *mat, float *light, float *nodes) {
        Scene sc;
        if (bvh_kind && *bvh_kind) sc.setBvhBuilder(bvh_kind);
        sc.parse(json);
        const long long c[8] = {sc.width, sc.height, (long long)sc.vertices.size(), (long long)sc.triangles.size(),
                                (long long)sc.materials.size(), (long long)sc.lights.size(), (long long)sc.nodes.size(), sc.bvhDepth_};
        for (int i = 0; i < 8; i++) counts[i] = c[i];
        if (view) std::memcpy(view, sc.viewM, sizeof sc.viewM);
        if (proj) std::memcpy(proj, sc.projM, sizeof sc.projM);
        if (lens) { lens[0] = sc.apertureRadius; lens[1] = sc.focalLength; }
        if (vert && !sc.vertices.empty()) std::memcpy(vert, sc.vertices.data(), sc.vertices.size() * sizeof(Vertex));
        if (tri && !sc.triangles.empty()) std::memcpy(tri, sc.triangles.data(), sc.triangles.size() * sizeof(Triangle));
        if (mat && !sc.materials.empty()) std::memcpy(mat, sc.materials.data(), sc.materials.size() * sizeof(Material));
        if (light && !sc.lights.empty()) std::memcpy(light, sc.lights.data(), sc.lights.size() * sizeof(Triangle));
        if (nodes && !sc.nodes.empty()) std::memcpy(nodes, sc.nodes.data(), sc.nodes.size() * sizeof(BVHNode));
        return 0;
    }
};
}  // namespace glrt

extern "C" GLRT_API int glrt_scene_probe(const char *json, const char *bvh_kind, long long counts[8], float view[16], float proj[16],
                                         float lens[2], float *vert, float *tri, float *mat, float *light, float *nodes) {
    return glrt::SceneProbe::run(json, bvh_kind, counts, view, proj, lens, vert, tri, mat, light, nodes);
}
