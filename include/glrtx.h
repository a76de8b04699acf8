/*
 * glrtx.h -- C ABI of libglrtx.so, the MI355X (gfx950) device layer that replaces the
 * GL side of tatsy/opengl-raytracer's per-pixel path-tracing pass.
 *
 * Drop-in boundary.  What each entry point replaces in the reference:
 *
 *   glrtx_create / glrtx_destroy     GL context + program setup: Window::Window, Window::initialize
 *                                    src/core/window.cpp:30-81, :185-211
 *   glrtx_upload_scene               the five TextureBuffer(size, fmt, usage) + setData(ptr) uploads
 *                                    src/core/scene.cpp:254-269, src/core/texture_buffer.h:8-12
 *                                    (byte layouts unchanged: scene.h:16-35, trimesh.h:15-25, bvh.h:84-100)
 *   glrtx_build_lbvh                 BVH::construct, src/core/bvh.cpp:59-160 (device-side linear BVH instead)
 *   glrtx_resize                     Window::resize -> resetBuffer (accumulators re-created, cleared)
 *                                    src/core/window.cpp:324-335, :366-381
 *   glrtx_clear                      glClear of the accumulation targets on reset (same lines)
 *   glrtx_render                     first half of Window::render(): uniform upload + the single
 *                                    glDrawArrays(GL_TRIANGLES, 0, 6) that runs raytrace.frag on every pixel
 *                                    src/core/window.cpp:213-295; shader src/shaders/raytrace.frag:565-614
 *   glrtx_render_frames              n consecutive iterations of the accumulation loop in Window::mainloop
 *                                    (src/core/window.cpp:121-169 calling render(), with the fresh u_seed of
 *                                    :226-238 per frame and a static camera), issued as one launch
 *   glrtx_params                     the uniforms set per frame, window.cpp:230-243, plus u_maxDepth which
 *                                    the reference leaves at its shader default 16 (raytrace.frag:47)
 *   glrtx_read_accum                 reading fbo[select] colour attachments 0/1 (RGB32F + R32F)
 *                                    src/core/window.cpp:366-381; fused here into float4(L.rgb, count)
 *   glrtx_resolve_rgba8              second half of Window::render() (screen.frag: rgb/count, clamp, gamma)
 *                                    + saveCurrentFrame's read-back and vertical flip
 *                                    src/shaders/screen.frag:15-25, src/core/window.cpp:297-317, :383-414
 *   glrtx_set_partition / glrtx_bind_accum / glrtx_set_stream
 *                                    no counterpart (reference is single-GPU): row-stripe sharding across
 *                                    one-process-per-GPU ranks, SURVEY.md section 8(e)
 *   glrtx_stats / glrtx_timer_*      replaces the whole-frame Timer, src/core/timer.h:7-36, window.cpp:119-168
 *   glrtx_last_error                 replaces FatalError's stderr + abort(), src/core/common.h:88-94
 *
 * Conventions: plain C, no torch / HIP types in signatures.  Host pointers are borrowed for the
 * duration of a call and copied; the ctx owns all device memory it allocates; the caller owns
 * output buffers.  Return value 0 = ok, negative = GLRTX_E*; the message is available from
 * glrtx_last_error(ctx) (ctx may be NULL for create-time failures).  One host thread drives a ctx.
 * Image rows use GL orientation: row 0 is gl_FragCoord.y = 0.5 (bottom), like the reference's FBOs.
 */
#ifndef GLRTX_H
#define GLRTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLRTX_OK 0
#define GLRTX_EINVAL (-1)   /* bad argument / call order */
#define GLRTX_EDEVICE (-2)  /* HIP runtime error or no usable gfx950 device */
#define GLRTX_ESCENE (-3)   /* scene buffers inconsistent (index out of range, BVH not a tree) */
#define GLRTX_EDEPTH (-4)   /* BVH needs more than the 64-entry traversal stack (raytrace.frag:284) */
#define GLRTX_ENOMEM (-5)

#define GLRTX_ABI_VERSION 10

typedef struct glrtx_ctx glrtx_ctx;

/* Per-frame uniforms (window.cpp:230-243).  Matrices are column-major, untransposed, exactly the
 * 16 floats glUniformMatrix4fv(..., GL_FALSE, ...) receives (shader_program.cpp:151-156). */
typedef struct glrtx_params {
    float c2w[16];   /* u_c2wMat = inverse(viewM * modelM) */
    float s2c[16];   /* u_s2cMat = inverse(projM) */
    float aperture;  /* u_apertureRadius */
    float focal;     /* u_focalLength */
    float seed[2];   /* u_seed */
    int32_t n_samples; /* u_nSamples (reference host sends 1, window.cpp:239) */
    int32_t max_depth; /* u_maxDepth ("k bounces" := k) */
} glrtx_params;

typedef struct glrtx_stats {
    uint64_t rays;          /* executions of intersect(Ray, out Intersection) since last clear/reset_stats;
                               only counted by launches made while ray counting is enabled */
    uint64_t rays_untraced; /* of `rays`: shadow rays the reference traces although both outcomes of its light test give the
                               same radiance bit for bit (a cosine <= 0, or a contribution too small to register); they are
                               counted above but resolved without a traversal */
    uint64_t paths;         /* pixel samples traced (owned pixels * n_samples per launch) */
    uint64_t launches;      /* frames rendered: glrtx_render calls + frames of glrtx_render_frames calls */
    uint64_t kernel_launches; /* launches of the render kernel (one per glrtx_render / glrtx_render_frames call) */
    double kernel_ms_total; /* sum over launches of the RENDER kernel's device time (HIP events on the ctx stream) */
    double accumulate_ms_total; /* sum of the plane-accumulation passes that follow glrtx_render_frames launches */
    float kernel_ms_last;   /* render kernel of the last launch */
    int32_t frames_last;    /* frames covered by the last launch */
    int32_t width, height;  /* full image */
    int32_t owned_rows;     /* rows of this ctx's partition */
    int32_t stack_entries;  /* traversal stack entries the uploaded BVH needs */
    int32_t lds_bytes;      /* dynamic LDS per workgroup of the render kernel */
    int32_t n_tri, n_fork, n_mat, n_light;
    int32_t variant_last;   /* kernel the last launch actually ran: 2 workgroup-local wavefront, 1 persistent megakernel, 0 tile megakernel */
    int32_t fallback_last;  /* 0, or why the last launch left the selected wavefront kernel (GLRTX_FALLBACK_* bits) */
    float resolve_ms_last;  /* device time of the last resolve kernel (glrtx_resolve_rgba8), without the copy to the host */
    int32_t node_fetch_last; /* wavefront kernel, last launch: 0 = one record per lane, 1 = pair-cooperative node fetch (large trees), 2 = the two in alternate
                                steps (small trees) -- all bit-identical; GLRTX_PAIR_FETCH=0/1/2 forces one (was reserved0) */
    uint64_t fallback_launches; /* launches since reset_stats that ran on the persistent megakernel although variant 2 was selected:
                                   ~2x slower per ray and without frames in flight -- visible here instead of silent */
    int32_t pipe_slots;         /* overlapped single-frame launches (glrtx_render): internal slots the last such launch could choose from -- GLRTX_PIPE_SLOTS
                                   (default 6), less when the memory budget or a failed allocation says so, 0 when such launches run un-piped */
    int32_t pipe_resident_max;  /* ... and the most of them that had a render kernel on the device at once (counted when a launch is issued) since
                                   reset_stats: 1 means consecutive launches did not overlap, whatever the reason (queue mapping, a caller that syncs) */
    int32_t device_error_pending; /* ABI 9: 1 while a launch that FAILED on the device sits unreported in the context's launch ring: glrtx_get_stats never blocks and
                                     never consumes such a record (its rc stays GLRTX_OK and the timing fields stop advancing); the error itself -- kernel, size, frame
                                     count, device -- is returned by the next glrtx_sync, or by the launch that needs the record's slot */
    int32_t wf_state_mib;       /* (was reserved1) MiB of path state the last wavefront launch ran on: an entry per workgroup OF THAT LAUNCH and path-queue position, two
                                   sets of six float4 planes -- 768 for a full grid on a 256-CU device whatever the frames in flight (384 for an overlapped single-frame
                                   launch at 1080p), 1 for a 9x9 image (ABI 10: sized by the launch, not by the device) */
    /* ABI 10 */
    int32_t shadow_limited;     /* which shadow-ray search the context's launches run: 0 the reference's own closest-hit search (default: bit-exact contract), 1 the
                                   range-limited one (glrtx_set_shadow_range_limit / GLRTX_SHADOW_LIMIT=1: outside the bit-exact contract).  bench.py prints it */
    int32_t reserved2;
    uint64_t feed_launches;     /* fed launches since reset_stats: launches that stayed open for the calls behind them (see glrtx_render) */
    uint64_t feed_appended;     /* frames of glrtx_render / glrtx_render_frames calls that a launch already running took by itself instead of a launch of their own */
} glrtx_stats;

/* glrtx_stats.fallback_last: the wavefront kernel packs depth and sample index into one word of its path state */
#define GLRTX_FALLBACK_DEPTH 1      /* u_maxDepth > 255 */
#define GLRTX_FALLBACK_SAMPLES 2    /* u_nSamples >= 2^20 */
#define GLRTX_FALLBACK_EXTENSIONS 4 /* analytic spheres uploaded or extension flags set (the extension kernel is a megakernel) */

int glrtx_abi_version(void);

/* device_id: HIP ordinal, or -1 for the current device. */
int glrtx_create(glrtx_ctx **out, int device_id);
void glrtx_destroy(glrtx_ctx *ctx);
const char *glrtx_last_error(const glrtx_ctx *ctx);

/* Buffers in the reference wire format; counts are in records (vertices, triangles, materials,
 * light triangles, BVH nodes).  n_light may be 0.  The scene is validated and repacked for the
 * device; it replaces any previous scene. */
int glrtx_upload_scene(glrtx_ctx *ctx, const float *vert, size_t n_vert, const float *tri, size_t n_tri,
                       const float *mat, size_t n_mat, const float *light, size_t n_light, const float *bvh,
                       size_t n_nodes);

/* Host-only (no device, no ctx): run the validation and repacking glrtx_upload_scene performs and
 * report the interior-node count and the traversal stack entries the BVH needs.  On failure the
 * message is available from glrtx_last_error(NULL). */
int glrtx_check_scene(const float *vert, size_t n_vert, const float *tri, size_t n_tri, const float *mat, size_t n_mat,
                      const float *light, size_t n_light, const float *bvh, size_t n_nodes, int *n_fork_out,
                      int *stack_entries_out);

/* Host-only test hook (no device, no ctx): the repacked fork records as the kernels read them -- 16 floats per fork
 * {child L box min, ref L} {child L box max, ref R} {child R box min, -} {child R box max, -}, refs as int bit patterns
 * (>= 0 fork index; < 0 the record ~id of a LEAF of the tree, ids numbered from 1 in the order the traversal meets the leaves -- a hit carries this id,
 * not the wire triangle index; ref -1 = id 0 = the all-zero never-hit record that stands for an absent child).  A fork whose two children are both leaves
 * has no fork record: its triangle records are chained (children.y's names children.x's) and its parent refers to the first of them, so n_fork_out counts
 * only the forks that remain -- so that a test can replay
 * the traversal step's push/pop rules on the packed tree and check stack_entries against the deepest stack it reaches.
 * forks_out may be NULL (counts only).  No reference counterpart (the reference's stack is a fixed int[64], raytrace.frag:284). */
int glrtx_debug_pack_forks(const float *vert, size_t n_vert, const float *tri, size_t n_tri, const float *mat, size_t n_mat,
                           const float *light, size_t n_light, const float *bvh, size_t n_nodes, float *forks_out,
                           size_t capacity_forks, int *n_fork_out, int *root_ref_out, int *stack_entries_out);

/* Linear BVH built on the device (30-bit Morton order, Karras hierarchy, bottom-up fit), returned in the wire format
 * glrtx_upload_scene takes: nodes_out = (2*n_tri-1)*9 floats, root = node 0.  Takes the place of the reference's CPU
 * builder BVH::construct (src/core/bvh.cpp:59-160) for large scenes; identical, bit for bit, to glrt_bvh_build_lbvh
 * (glrt_host.h).  build_ms_out (may be NULL): device time of the build without the host<->device copies. */
int glrtx_build_lbvh(glrtx_ctx *ctx, const float *vert, size_t n_vert, const float *tri, size_t n_tri, float *nodes_out,
                     int *max_depth_out, float *build_ms_out);
/* Binned SAH built on the device, level by level from the top (16 bins, 3 axes), with the exact sweep SAH of the LBVH pass for the subtrees of <= 64 triangles
 * (csrc/sahl.hip.h; round 5).  Same contract and node layout as glrtx_build_lbvh; identical, bit for bit, to glrt_bvh_build_sah_levels (glrt_host.h).  The tree is as good
 * as the CPU binned-SAH builder's -- config 5 takes 80.5 instead of the LBVH's 84.4 traversal steps per ray (profiles/r05_tree_study.txt) -- for about twice the LBVH's
 * build time.  Replaces BVH::construct (src/core/bvh.cpp:59-160) like glrtx_build_lbvh does.
 * Neither device builder knows the materials: glrt_bvh_lights_first (glrt_host.h) on the returned nodes puts the child that holds the emitting triangles into the slot the
 * reference's traversal visits first -- worth 4 % per frame on the headline scene (profiles/r05_lights_first.txt); glrtx_upload_scene itself never reorders a tree. */
int glrtx_build_bvh_sah(glrtx_ctx *ctx, const float *vert, size_t n_vert, const float *tri, size_t n_tri, float *nodes_out,
                     int *max_depth_out, float *build_ms_out);

/* Full image size; (re)allocates and clears this ctx's accumulator rows. */
int glrtx_resize(glrtx_ctx *ctx, int width, int height);
int glrtx_clear(glrtx_ctx *ctx);

/* Row-stripe partition for multi-GPU: this ctx owns stripes s (of stripe_rows rows) with
 * s % world == rank; its accumulator holds only those rows, in increasing y, pixel coordinates
 * stay global.  stripe_rows: a multiple of 8 (whole 8x8 work tiles).  Default (rank 0, world 1) owns everything.
 * Must precede glrtx_resize. */
int glrtx_set_partition(glrtx_ctx *ctx, int rank, int world, int stripe_rows);
/* Global y of local accumulator row r (r in [0, owned_rows)), or -1. */
int glrtx_local_row_to_y(const glrtx_ctx *ctx, int local_row);

/* Optional: render into caller-owned device memory (e.g. a torch tensor that RCCL gathers)
 * instead of the ctx's own buffer: capacity_rows (>= owned_rows) rows of pitch_bytes, width float4 each.
 * While bound, glrtx_resize / glrtx_set_partition fail with GLRTX_EINVAL for a shape that does not fit
 * the buffer (the ctx cannot grow memory it does not own).  NULL unbinds. */
int glrtx_bind_accum(glrtx_ctx *ctx, void *device_ptr, size_t pitch_bytes, int capacity_rows);
/* Optional: launch on a caller-owned hipStream_t (passed as void*); NULL restores the ctx stream. */
int glrtx_set_stream(glrtx_ctx *ctx, void *hip_stream);

/* Kernel variant (tuning knob, no reference counterpart): 2 = workgroup-local wavefront (default: one
 * persistent launch; each workgroup runs traverse/shade trips over its own pixel blocks), 1 = persistent
 * megakernel with path regeneration, 0 = megakernel, one 16x16 tile per workgroup.  All three produce
 * bit-identical images.  The default can also be set with the environment variable GLRTX_VARIANT. */
int glrtx_set_variant(glrtx_ctx *ctx, int variant);

/* Shadow rays (sampleDirect, raytrace.frag:337-403).  Default (0): every light sample is resolved by the reference's own closest-hit search -- tHit
 * starts at INFTY, boxes are culled by the hits found, in the reference's visiting order -- and only ends early once an occluder in front of the
 * light is known (exact: tHit can only shrink).  1 opts in to the RANGE LIMIT of rounds 1-4: the search starts with tHit just beyond the light
 * sample's distance, which also culls boxes beyond the light (3-4 % fewer node visits per frame).  The limit gives the reference's verdict unless a
 * triangle test's computed t falls below the limit while the computed entry distance of one of its ancestors' boxes lies above it -- an
 * ill-conditioned (grazing) triangle test near the edge of a light lying flush in its box; no margin in terms of the distance bounds that error,
 * so the mode is NOT part of the bit-exact contract (csrc/pt_kernel.hip.h: shadow_limit).  Also: environment variable GLRTX_SHADOW_LIMIT=1 at
 * glrtx_create.  Takes effect with the next launch.  Group members: through glrtx_group_ctx. */
int glrtx_set_shadow_range_limit(glrtx_ctx *ctx, int enable);

/* Enable/disable per-launch ray counting (one atomic per wavefront); default off. */
int glrtx_count_rays(glrtx_ctx *ctx, int enable);

/* Asynchronous: accumulates n_samples new samples per owned pixel.  Returns as soon as the launch is enqueued (up to 16 launches may be
 * outstanding per context).  Consecutive calls overlap on the device: the render kernel of a call runs on one of six internal streams with
 * buffers of its own and hands its samples over in planes; only the pass that adds them to the accumulator runs on the context's stream
 * (glrtx_set_stream), in call order -- so everything a caller orders behind the call on that stream (resolve, read-back, a collective on the
 * rows) sees the finished accumulator, and per pixel the additions happen in the order of the calls.
 * FED LAUNCHES (ABI 10; the context's own stream only): a call that follows another render call directly -- same camera, samples and depth, nothing in between
 * that reads the accumulator or changes what a launch depends on -- does not become a launch of its own while that launch is still running: its frame is published to
 * the running kernel in host-coherent memory and rendered by it (glrtx_stats.feed_appended).  The pixels, and what every later call sees, are the same; only the launch
 * boundaries go away (one ramp and one drain per burst).  glrtx_sync, glrtx_resolve_rgba8, glrtx_read_accum, glrtx_clear, ... seal the open launch: nothing issued
 * after them is appended to a launch queued in front of them.  GLRTX_NO_FEED=1 turns this off.  See INTEGRATION.md. */
int glrtx_render(glrtx_ctx *ctx, const glrtx_params *params);
/* Frames in flight.  Same result, bit for bit, as n_frames consecutive glrtx_render calls whose params differ only
 * in `seed` (seeds_xy = n_frames pairs; params->seed is ignored) -- the reference's accumulation loop with a static
 * camera, window.cpp:226-252 -- but issued as ONE launch, so that a small image (or one rank's share of it) still fills
 * the GPU and the tail of one frame overlaps the next.  Every sample is kept in its own plane and the planes are added
 * to the accumulator in frame order.  stats.launches counts n_frames.  Asynchronous; seeds_xy is copied. */
int glrtx_render_frames(glrtx_ctx *ctx, const glrtx_params *params, const float *seeds_xy, int n_frames);
int glrtx_sync(glrtx_ctx *ctx);

/* Copy the owned rows (owned_rows x width float4) to host memory; implies a sync. */
int glrtx_read_accum(glrtx_ctx *ctx, float *dst_rgba, size_t dst_pitch_bytes);
/* Device address / pitch of the accumulator currently rendered into. */
int glrtx_accum_device_ptr(const glrtx_ctx *ctx, void **ptr_out, size_t *pitch_bytes_out);

/* Tonemap the owned rows to RGBA8: clamp(rgb/count, 0, 1)^(1/gamma), alpha 255.  If flip_y, row 0 of
 * dst is the top image row (as written by the reference's saveCurrentFrame).  Implies a sync. */
int glrtx_resolve_rgba8(glrtx_ctx *ctx, uint8_t *dst, size_t dst_pitch_bytes, float gamma, int flip_y);

/* Profile of a calibration frame (ABI 10): hist_out[t] = how often triangle t of the uploaded scene was the closest hit of a path ray -- camera rays and bounces -- in
 * ONE frame of `p`, rendered by the wavefront kernel into a scratch accumulator (the context's accumulator and statistics are left as they were; the call waits for the
 * device).  n_tri must be the uploaded scene's triangle count.  Input of glrt_bvh_order_by_hits (glrt_host.h): the child that is hit more often goes into the slot the
 * reference's traversal visits first (raytrace.frag:299-307; which child is which is the builder's choice, bvh.cpp:72-160). */
int glrtx_hit_histogram(glrtx_ctx *ctx, const glrtx_params *p, uint32_t *hist_out, size_t n_tri);

/* Measurement aid (bench.py's roofline_aux): device time of ONE launch of the resolve kernel -- screen.frag:15-25 over the owned rows -- from `reps` launches back to
 * back between one pair of events (a single launch between two events also measures the command processor's latency on both sides). */
int glrtx_debug_resolve_burst(glrtx_ctx *ctx, float gamma, int reps, float *ms_per_launch);

int glrtx_get_stats(const glrtx_ctx *ctx, glrtx_stats *out);
int glrtx_reset_stats(glrtx_ctx *ctx);

/* HIP-event stopwatch on the stream launches go to: begin, N x render, end -> elapsed device ms. */
int glrtx_timer_begin(glrtx_ctx *ctx);
int glrtx_timer_end(glrtx_ctx *ctx, float *elapsed_ms_out);

/* ---- Extensions beyond the reference.  PARITY UNPINNED: the reference has no analytic primitive
 * (scenes are triangle meshes, raytrace.frag:226-257) and never branches on MTRL_DIELECTRIC (raytrace.frag:32), so there is no
 * reference output for any of this; it is checked against this build's own CPU restatement (oracle/pt_oracle.c, *_ext) and
 * against the tessellation limit of the pinned triangle path.  Off unless asked for; while active, launches run on the
 * persistent megakernel (no frames in flight) and everything the reference does define keeps its pinned arithmetic.
 *   glrtx_upload_spheres   n x {cx, cy, cz, radius, materialId}: analytic spheres next to the triangle BVH, tested one by
 *                          one (at most 1024); call after glrtx_upload_scene (which drops them); n = 0 removes them.
 *   glrtx_set_extensions   GLRTX_EXT_DIELECTRIC: materials of type 4 (MTRL_DIELECTRIC; param0 = tint, param1.x = index of
 *                          refraction) reflect / refract with the Fresnel reflectance as probability instead of being black.
 *                          GLRTX_EXT_WHITTED: Whitted-style transport -- a diffuse surface gathers its direct light and the
 *                          path ends; only specular bounces continue. */
#define GLRTX_EXT_DIELECTRIC 1
#define GLRTX_EXT_WHITTED 2
int glrtx_upload_spheres(glrtx_ctx *ctx, const float *spheres, size_t n_spheres);
int glrtx_set_extensions(glrtx_ctx *ctx, int flags);

/* ---- Groups: the same device layer on several GPUs of one node, behind one handle and one host thread.
 * No reference counterpart (the reference is single-GPU); SURVEY.md 8(b) sketches glrtx_create(ctx**, device_ids, n) with a
 * gathering read_accum -- this is that, kept apart from the single-context calls.  Member i owns the 8-row stripes s with
 * s % n == i (global pixel coordinates, resident accumulator rows, its own stream); rendering exchanges nothing; read_accum
 * and resolve_rgba8 first copy the stripes device-to-device into a full frame on member 0's GPU (one strided xGMI peer copy per member,
 * issued on the member's own stream so that the links work side by side), i.e. they
 * return the FULL image.  device_ids may name the same GPU more than once (partition emulation, used by the tests).
 * glrtx_group_ctx borrows a member for the per-context knobs (glrtx_set_variant, glrtx_count_rays, glrtx_get_stats);
 * do not resize, partition or destroy a member directly.  glrtx_group_get_stats sums rays / paths / rows and takes the
 * maximum of the kernel times (the members run concurrently). */
typedef struct glrtx_group glrtx_group;
int glrtx_group_create(glrtx_group **out, const int *device_ids, int n_devices);
void glrtx_group_destroy(glrtx_group *grp);
const char *glrtx_group_last_error(const glrtx_group *grp);
int glrtx_group_size(const glrtx_group *grp);
glrtx_ctx *glrtx_group_ctx(glrtx_group *grp, int i);
int glrtx_group_upload_scene(glrtx_group *grp, const float *vert, size_t n_vert, const float *tri, size_t n_tri, const float *mat,
                             size_t n_mat, const float *light, size_t n_light, const float *bvh, size_t n_nodes);
int glrtx_group_resize(glrtx_group *grp, int width, int height);
int glrtx_group_clear(glrtx_group *grp);
int glrtx_group_render(glrtx_group *grp, const glrtx_params *params);
int glrtx_group_render_frames(glrtx_group *grp, const glrtx_params *params, const float *seeds_xy, int n_frames);
int glrtx_group_sync(glrtx_group *grp);
int glrtx_group_read_accum(glrtx_group *grp, float *dst_rgba, size_t dst_pitch_bytes);
int glrtx_group_resolve_rgba8(glrtx_group *grp, uint8_t *dst, size_t dst_pitch_bytes, float gamma, int flip_y);
int glrtx_group_get_stats(const glrtx_group *grp, glrtx_stats *out);
/* Device-to-device copies the last gather (read_accum / resolve_rgba8) issued: one strided copy per member, on the member's own
 * stream, plus one for a partial last stripe -- at most 2 x members (diagnostic; the tests pin it). */
int glrtx_group_gather_copies(const glrtx_group *grp);

#ifdef __cplusplus
}
#endif
#endif
