/*
 * glrt_host.h -- C ABI of libglrt_host.so: host-side (CPU) helpers that sit
 * next to the device path: BVH construction into the reference's flat
 * 'u_bvhBuffer' format and the camera matrices the reference computes with GLM.
 * No GPU, no HIP; loadable on any box.
 *
 * Reference interfaces replaced:
 *   glrt_bvh_build_sah      BVH::construct / constructRec   src/core/bvh.cpp:59-160
 *   glrt_bvh_build_reference  the same pair, restated rule for rule: the reference host's OWN tree (opt-in: exact ties and
 *                            grazing-ray box misses then fall where the reference host's would; parity unpinned, see below)
 *   glrt_bvh_build_lbvh     same role, linear BVH for large scenes (BASELINE config 5)
 *   glrt_bvh_build_chain    (no counterpart: expresses BASELINE config "brute force, no BVH"
 *                            in the same node format)
 *   glrt_look_at            glm::lookAt                     src/core/scene.cpp:93
 *   glrt_perspective        glm::perspective(radians(fov))  src/core/scene.cpp:113
 *   glrt_mat4_inverse/_mul  glm::inverse, operator*         src/core/window.cpp:230-233
 *   glrt_frame_seed         per-frame u_seed draw           src/core/window.cpp:226-238
 *                            (the reference draws from mt19937(random_device); this is the
 *                             deterministic sequence SURVEY.md section 8(d) fixes)
 * All matrices are column-major float[16] exactly as uploaded by
 * glUniformMatrix4fv(..., GL_FALSE, ...) (shader_program.cpp:151-156).
 */
#ifndef GLRT_HOST_H
#define GLRT_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLRT_HOST_OK 0
#define GLRT_HOST_EINVAL (-1) /* null pointer / empty input */
#define GLRT_HOST_EINDEX (-2) /* triangle references a vertex out of range */
#define GLRT_HOST_EDEPTH (-3) /* tree deeper than the 64-entry traversal stack allows */

/* floats per Vertex record in u_vertBuffer: pos, normal, uv, tangent, binormal (trimesh.h:15-25) */
#define GLRT_VERTEX_FLOATS 15
/* floats per Material record in u_matBuffer: 6 x vec3 (scene.h:28-35) */
#define GLRT_MATERIAL_FLOATS 18
/* floats per BVH node in u_bvhBuffer: 3 x vec3 (bvh.h:84-100) */
#define GLRT_BVHNODE_FLOATS 9

/* 2*n_tri-1 (one triangle per leaf), 0 for an empty scene. */
size_t glrt_bvh_node_count(size_t n_tri);

/* vert: n_vert * GLRT_VERTEX_FLOATS, tri: n_tri * 4 (i, j, k, material) as floats.
 * nodes_out: glrt_bvh_node_count(n_tri) * 9 floats.  max_depth_out may be NULL.
 * glrt_bvh_build_sah: 16 bins on each of the three axes; a subtree of n triangles is charged n^0.8 in the split cost (every subtree ends in one-triangle leaves, where the
 * surface-area heuristic's n overstates what a ray pays: headline -0.6 ... -0.8 % per frame, profiles/r06_sah_count_weight.txt; GLRT_SAH_ALPHA=1 restores the heuristic). */
int glrt_bvh_build_sah(const float *vert, size_t n_vert, const float *tri, size_t n_tri, float *nodes_out,
                       int *max_depth_out);
int glrt_bvh_build_chain(const float *vert, size_t n_vert, const float *tri, size_t n_tri, float *nodes_out);
/* The tree the reference's own host builds (BVH::constructRec, src/core/bvh.cpp:72-160 with bvh.h:11-82), restated rule for rule: longest centroid axis only,
 * std::nth_element at the middle for <= 8 triangles, 16 buckets + std::partition above, the cut at the unsorted middle when no bucket split pays, boxes accumulated from
 * +-1e8, nodes in pre-order.  A closest hit depends on the tree only at exact ties and at grazing-ray box misses: under this tree both fall where
 * the reference host's do -- to the extent that libstdc++'s nth_element / partition order is the reference build's (same toolchain: yes).  It is a worse tree than
 * glrt_bvh_build_sah's (one axis binned) and is never improved behind the caller's back: glrt::Scene applies neither glrt_bvh_lights_first nor any other re-ordering to it.
 * PARITY UNPINNED: the reference host is unbuildable here, no tree of its making exists to compare with.  Same arguments and return codes as
 * glrt_bvh_build_sah. */
int glrt_bvh_build_reference(const float *vert, size_t n_vert, const float *tri, size_t n_tri, float *nodes_out, int *max_depth_out);
/* Linear BVH (30-bit Morton order + Karras hierarchy), improved by GLRT_LBVH_ROTATION_PASSES bottom-up sweeps of tree
 * rotations (child <-> grandchild and grandchild <-> grandchild exchanges) and by rebuilding every maximal subtree of at most
 * GLRT_LBVH_REBUILD_LEAVES leaves with the exact sweep SAH.  Same output, bit for bit, as the GPU builder glrtx_build_lbvh (include/glrtx.h); internal node i at index i,
 * leaves after them in Morton order. */
#ifndef GLRT_LBVH_ROTATION_PASSES
#define GLRT_LBVH_ROTATION_PASSES 4
#endif
#ifndef GLRT_LBVH_REBUILD_LEAVES
#define GLRT_LBVH_REBUILD_LEAVES 64
#endif
int glrt_bvh_build_lbvh(const float *vert, size_t n_vert, const float *tri, size_t n_tri, float *nodes_out,
                        int *max_depth_out);
/* Binned SAH built level by level from the top (16 bins, 3 axes) down to segments of at most GLRT_LBVH_REBUILD_LEAVES triangles, which the exact sweep SAH of the
 * LBVH pass then builds from their leaves (host/bvh.cpp: "SAH by levels").  Same node layout as glrt_bvh_build_lbvh; same output, bit for bit, as the GPU builder
 * glrtx_build_bvh_sah (include/glrtx.h).  Quality: the CPU binned-SAH tree's or better (profiles/r05_tree_study.txt). */
int glrt_bvh_build_sah_levels(const float *vert, size_t n_vert, const float *tri, size_t n_tri, float *nodes_out, int *max_depth_out);
/* Post-pass over a tree of ANY of the builders above (or of the device builders, include/glrtx.h): at every fork where exactly one child's subtree contains emitting
 * triangles (material emission != 0: the rule of the light list, scene.cpp:246-248) that child is put into the slot the reference's traversal visits FIRST (children.y,
 * raytrace.frag:299-307), so that shadow rays -- half of all rays -- meet their light before anything else and cull the rest by its distance.  Two child references per
 * exchanged fork change, nothing else.  mat: n_mat records of 18 floats (scene.h:28-35).  Returns the number of forks exchanged (>= 0) or GLRT_HOST_E*.
 * glrt::Scene::parse and the Python scene builder apply it after their builder (GLRT_BVH_LIGHTS_FIRST=0 leaves the builder's order). */
int glrt_bvh_lights_first(float *nodes, size_t n_nodes, const float *tri, size_t n_tri, const float *mat, size_t n_mat);
/* The order of a fork's children from measured hits: tri_hits[t] = closest hits triangle t collected in a calibration frame (glrtx_hit_histogram, glrtx.h); at every
 * fork the child whose subtree collected more hits PER UNIT COST (a subtree of n triangles is charged n^0.5; GLRT_HITS_COST_EXP overrides, 0 = hits alone: round 6's first
 * form, which sent every ray through the bigger child first on unbalanced trees) goes into the slot the traversal visits first (raytrace.frag:299-307).  No box and no closest hit changes; exact ties between
 * two triangles may resolve to the other one.  Apply it last.  Returns the forks exchanged, or GLRT_HOST_E*. */
int glrt_bvh_order_by_hits(float *nodes, size_t n_nodes, const uint32_t *tri_hits, size_t n_tri);
/* The shadow rays' share of a calibration frame's hits, by the reference's sampling rule (raytrace.frag:341-343: a light triangle is drawn uniformly for every shaded
 * hit): (sum of tri_hits) / (number of emitting triangles) is added to every emitting triangle.  Call it on glrtx_hit_histogram's output before glrt_bvh_order_by_hits.
 * Returns the number of emitting triangles, or GLRT_HOST_E*. */
int glrt_bvh_add_shadow_hits(uint32_t *tri_hits, size_t n_tri, const float *tri, const float *mat, size_t n_mat);
/* Optimisation pass over a finished tree of any builder: every subtree is taken out and put back where the summed area of the forks' boxes grows least (insertion-based
 * optimisation, Bittner et al. 2013; host/bvh.cpp).  At most max_passes passes, stopping when one gains < 0.1 %.  The tree is renumbered in DFS pre-order.
 * cost_out (may be NULL): summed fork area / root area before [0] and after [1].  Returns the number of subtrees moved (>= 0; 0 and max_depth -1 for trees it leaves
 * alone: < 4 leaves, absent children, non-finite boxes) or GLRT_HOST_E*
 * -- GLRT_HOST_EDEPTH when the optimised tree would be deeper than the 64-entry traversal stack allows: `nodes` is then left exactly as it came in.  Apply glrt_bvh_lights_first AFTER it. */
int glrt_bvh_reinsert(float *nodes, size_t n_nodes, int max_passes, int *max_depth_out, double *cost_out);

void glrt_look_at(const float eye[3], const float center[3], const float up[3], float out[16]);
void glrt_perspective(float fovy_deg, float aspect, float z_near, float z_far, float out[16]);
void glrt_mat4_mul(const float a[16], const float b[16], float out[16]);
/* returns 0, or GLRT_HOST_EINVAL for a singular matrix */
int glrt_mat4_inverse(const float m[16], float out[16]);
/* frame f -> (fract(0.137 + 0.6180340 f), fract(0.731 + 0.3819660 f)) */
void glrt_frame_seed(uint32_t frame, float out[2]);

#ifdef __cplusplus
}
#endif
#endif
