// glrtx.hip -- C-ABI host layer of libglrtx.so (include/glrtx.h): context, scene validation and
// repacking, accumulator management, kernel launch, timing.  This is the "thin C-ABI HIP host
// layer" that stands where GLFW/GL stood in the reference (src/core/window.cpp,
// src/core/texture_buffer.cpp, src/core/framebuffer_object.cpp).  gfx950 only.
#include "glrtx.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <limits>
#include <string>
#include <vector>

#include "glrt_host.h"
#include "pt_kernel.hip.h"
#include "lbvh.hip.h"
#include "sahl.hip.h"
static_assert(glrtx::lbvh::kRotationPasses == GLRT_LBVH_ROTATION_PASSES, "device and CPU LBVH statements must run the same rotation sweeps");
static_assert(glrtx::lbvh::kRebuildLeaves == GLRT_LBVH_REBUILD_LEAVES, "device and CPU LBVH statements must rebuild the same subtrees");

using namespace glrtx;

namespace {

constexpr int kStripeQuantum = 8;   // stripe heights are multiples of the 8x8 work tile
constexpr int kGroupStripe = 8;     // stripe height glrtx_group uses: 1080 rows over 8 members = 136 / 128 rows (16-row stripes: 144 / 128)
constexpr int kPairFetchMinRecords = 32768;  // forks + triangles (2 MiB of 64-byte records) from which the wavefront kernel fetches nodes pair-cooperatively
constexpr int kFramesBudgetGiB = 32;  // device memory one glrtx_render_frames launch may use for its sample planes (and the single-frame pipe slots for their buffers)
constexpr float kInf = std::numeric_limits<float>::infinity();
constexpr unsigned kPipeSlots = 8;    // most single-frame launches that may be in flight at once (each with its own stream, state, queues, planes); pipe_slots are used
constexpr unsigned kLaunchRing = 16;  // launches that may be outstanding per context before a new one waits for the oldest

thread_local std::string g_create_error;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct glrtx_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // Per launch: start, render kernel done, all done.  A RING of triples: a launch takes the next free one and never waits for an
    // earlier launch; finished launches are folded into the stats lazily (glrtx_sync, glrtx_get_stats, or -- without blocking -- at the
    // next launch).  Only with kLaunchRing launches outstanding does a new launch wait, for the oldest one.
    struct LaunchRec {
        hipEvent_t ev0 = nullptr, evm = nullptr, eva = nullptr, ev1 = nullptr;  // render start / end, accumulation start / end (eva recorded only by overlapped launches)
        bool has_eva = false;
        const char *kernel = "";
        int frames = 0;
    };
    LaunchRec ring[kLaunchRing];
    unsigned ring_head = 0, ring_tail = 0;  // launches [tail, head) are recorded and not yet folded
    hipEvent_t tm0 = nullptr, tm1 = nullptr;      // glrtx_timer_*
    hipEvent_t rs0 = nullptr, rs1 = nullptr;      // around the resolve kernel
    std::string err;

    DevBuf forks, nrms, mats, lights, vine, accum_own, counter, rgba8, work;
    DevBuf spheres, sphereMat;  // extension kernel: analytic spheres
    int n_spheres = 0, ext_flags = 0;
    DevBuf wfState, wfQ;      // wavefront path state (kWfStatePlanes = 6 planes of float4 x ids) + per-workgroup queues (variant 2)
    DevBuf wfSeeds, wfPlanes;       // frames in flight: per-frame seeds, per-sample planes
    // Single-frame launches that overlap (launch_wgwf): pipe_slots (<= kPipeSlots) slots used in turn, each with its own stream, path state, queues, tile
    // counter and sample planes; the accumulator is only touched by the plane-accumulation pass, on the context's stream, in launch order.
    struct PipeSlot {
        hipStream_t stream = nullptr;
        hipEvent_t render_done = nullptr, acc_done = nullptr;
        bool used = false;
        DevBuf state, queues, planes, work;
        // fed launches (pt_kernel.hip.h: FeedHost / FeedDev): the host-coherent block the host publishes frames in, its device address, the launch's device mirror,
        // and the sample planes in chunks of kFeedChunkFrames frames, allocated as a burst grows and kept for the next one (chunk_bytes: what each was allocated with)
        FeedHost *feed_h = nullptr, *feed_h_dev = nullptr;
        DevBuf feed_d;
        DevBuf chunks[kFeedChunks];
        size_t chunk_bytes = 0;
    } pipe[kPipeSlots];
    // The launch that is still OPEN: a fed launch of this context that may still be running and that the next glrtx_render / glrtx_render_frames call with the same
    // camera appends its frames to (feed_append) instead of starting a launch of its own.  Every other entry point that orders something against the accumulator
    // or changes what a launch depends on SEALS it (seal_feed): nothing is appended any more, the launch finishes what it has.
    struct OpenFeed {
        PipeSlot *slot = nullptr;
        glrtx_params p{};       // the camera / sampling parameters the launch was started with (seed ignored)
        int frames = 0, cap = 0;  // frames published so far / the most this launch can take
        LaunchRec *rec = nullptr;
        size_t frame_bytes = 0;   // sample planes of one frame
    } open;
    bool feed_ok = true;            // cleared when the host-coherent blocks cannot be had; GLRTX_NO_FEED=1 (read at every launch, like the other switches) turns fed launches off (A/B, tests)
    bool last_was_render = false;   // the previous call on this context was glrtx_render / glrtx_render_frames (nothing has observed the accumulator since)
    glrtx_params last_p{};          // ... with these parameters
    hipEvent_t last_render_done = nullptr;  // ... and this is its render kernel's completion event (a slot's; not owned)
    unsigned pipe_next = 0;
    bool pipeline = true;           // GLRTX_NO_PIPELINE=1 switches it off (A/B)
    int pipe_share = 1;             // an overlapped launch issued while others still render takes 1 / pipe_share of the workgroup slots (GLRTX_PIPE_SHARE; 1: all)
    unsigned pipe_slots = 6;        // slots used in turn (GLRTX_PIPE_SLOTS, <= kPipeSlots)
    DevBuf bvhVert, bvhTri, bvhNodes;  // glrtx_build_lbvh staging
    lbvh::Workspace bvhWs;
    int variant = 2;          // 0 = tile megakernel, 1 = persistent megakernel with path regeneration, 2 = workgroup-local wavefront
    int n_cu = 256;
    DevScene sc{};
    bool have_scene = false;
    int n_tri = 0, n_fork = 0, n_mat = 0, n_light = 0;
    const float *frames_seeds = nullptr;  // set only inside glrtx_render_frames
    int frames_n = 1;
    int budget_share = 1;     // contexts of one group on this device: each takes 1 / budget_share of the frames-in-flight memory budget

    int width = 0, height = 0;
    int rank = 0, world = 1, stripe = 16;
    int owned_rows = 0;
    float4 *accum = nullptr;  // own or bound
    size_t pitch_bytes = 0;
    bool bound = false;
    int bound_rows = 0;       // rows the caller's buffer holds (glrtx_bind_accum)

    // Four words of host-coherent memory the persistent kernel's trip guards report through ({code, workgroup, trips without a tile, live paths}; pt_render_wgwf):
    // read -- and cleared -- when a launch is folded.
    unsigned *guard_host = nullptr, *guard_dev = nullptr;
    hipEvent_t state_done = nullptr;  // the last render kernel that used wfState (plain launches on the context's stream, fed launches on their slot's) has ended
    bool state_used = false;

    std::vector<int> leaf_tri;  // leaf record k of the uploaded scene -> wire triangle (glrtx_hit_histogram)
    unsigned *hit_hist_dev = nullptr;  // set only inside glrtx_hit_histogram

    bool count_rays = false;
    const char *last_kernel = "";  // name of the last render kernel launched (error reports)
    mutable bool counters_stale = false;          // a counting launch was issued since the device counters were last read
    mutable unsigned long long counters_host[2] = {0, 0};
    glrtx_stats st{};
};

namespace {

int fail(glrtx_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(c, call)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(c, GLRTX_EDEVICE, "%s failed: %s", #call, hipGetErrorString(e_));     \
    } while (0)

int dev_upload(glrtx_ctx *c, DevBuf &b, const void *src, size_t bytes) {
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }
    const size_t alloc = std::max<size_t>(bytes, 64);
    HIP_TRY(c, hipMalloc(&b.p, alloc));
    b.bytes = alloc;
    if (bytes) HIP_TRY(c, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return GLRTX_OK;
}

void dev_free(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.bytes = 0;
}

int owned_rows_of(int height, int rank, int world, int stripe) {
    int rows = 0;
    const int n_stripes = (height + stripe - 1) / stripe;
    for (int s = rank; s < n_stripes; s += world) rows += std::min(stripe, height - s * stripe);
    return rows;
}

inline float as_float(int v) { float f; std::memcpy(&f, &v, 4); return f; }

// Fold finished launches' event triples into the stats, oldest first.  block = false: only those whose events have already
// completed (hipEventQuery; nothing on the launch path ever waits for the device); block = true: all of them (the caller has
// synchronised the stream, or wants to).  A fault inside a kernel surfaces here, not at the launch call: say which launch it was.
int fold_launches(glrtx_ctx *c, bool block, unsigned keep = 0) {
    while (c->ring_head - c->ring_tail > keep) {
        glrtx_ctx::LaunchRec &r = c->ring[c->ring_tail % kLaunchRing];
        hipError_t e = block ? hipEventSynchronize(r.ev1) : hipEventQuery(r.ev1);
        if (e == hipErrorNotReady) { (void)hipGetLastError(); return GLRTX_OK; }
        // a polling caller (glrtx_get_stats) must not swallow a device error: the record stays in the ring, and the next blocking fold -- glrtx_sync,
        // or a launch that needs the slot -- reports it with the launch it belongs to
        // -- but the poller is TOLD: glrtx_stats.device_error_pending stays set until that blocking fold has consumed the record
        const bool guard = e == hipSuccess && c->guard_host && __atomic_load_n(&c->guard_host[0], __ATOMIC_ACQUIRE) != 0u;  // (a trip guard's report: below)
        if ((e != hipSuccess || guard) && !block) { (void)hipGetLastError(); c->st.device_error_pending = 1; return GLRTX_OK; }
        c->ring_tail++;
        if (e != hipSuccess || guard) c->st.device_error_pending = 0;
        if (e != hipSuccess)
            return fail(c, GLRTX_EDEVICE, "render launch failed on the device (%s): %s, %dx%d (%d owned rows), %d frame(s), device %d", hipGetErrorString(e),
                        r.kernel, c->width, c->height, c->owned_rows, r.frames, c->device);
        if (guard) {  // a trip guard of the persistent kernel fired (pt_render_wgwf): the launch left paths unfinished
            const unsigned code = c->guard_host[0], wg = c->guard_host[1], trips = c->guard_host[2], alive = c->guard_host[3];
            std::memset(c->guard_host, 0, 16);
            return fail(c, GLRTX_EDEVICE, "render launch aborted by a trip guard (%s): workgroup %u, %u trips since its last tile, %u paths alive; %s, %dx%d (%d owned rows), %d frame(s), device %d",
                        code == 1u ? "two trips in which no ray was dealt and no path moved" : (code == 3u ? "a fed launch could neither claim a tile nor close its feed" : "trip limit exceeded"), wg, trips, alive, r.kernel, c->width, c->height,
                        c->owned_rows, r.frames, c->device);
        }
        float ms = 0.f, ms2 = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, r.ev0, r.evm));   // render kernel
        HIP_TRY(c, hipEventElapsedTime(&ms2, r.has_eva ? r.eva : r.evm, r.ev1));  // plane accumulation (frames in flight, overlapped single frames), else ~0
        c->st.kernel_ms_last = ms;
        c->st.kernel_ms_total += ms;
        c->st.accumulate_ms_total += ms2;
        c->st.kernel_launches++;
    }
    return GLRTX_OK;
}

// The event triple of the launch being issued.  Never blocks unless kLaunchRing launches are outstanding.
int next_launch_rec(glrtx_ctx *c, glrtx_ctx::LaunchRec *&out) {
    if (int rc = fold_launches(c, false)) return rc;
    if (c->ring_head - c->ring_tail >= kLaunchRing)
        if (int rc = fold_launches(c, true, kLaunchRing - 1)) return rc;
    out = &c->ring[c->ring_head % kLaunchRing];
    return GLRTX_OK;
}

int lds_bytes_for(const DevScene &sc) {
    return sc.lds_head_f4 * (int)sizeof(float4) + 2 * sc.stack_entries * kBlockThreads * (int)sizeof(int);  // materials and lights | traversal stacks
}

int pfail(glrtx_ctx *c, std::string *err_out, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (err_out) *err_out = buf;
    else if (c) c->err = buf;
    else g_create_error = buf;
    return code;
}

// Validation + repacking of the wire-format scene; pure host code (no HIP calls), so it can be
// exercised without a GPU through glrtx_check_scene.
struct Packed {
    std::vector<float4> forks, tris, nrms, mats, lights, vine;
    int vine_uniform = 0;
    int n_vine = 0, vine_main = 0;  // records of the list (= triangles), index of the last one (the list is padded: see pack_scene)
    float4 root_lo = make_float4(0.f, 0.f, 0.f, 0.f), root_hi = make_float4(0.f, 0.f, 0.f, 0.f);
    int root_ref = REF_ABSENT;
    int root_boxed = 0;   // the wire root is a fork: its own box (root_lo / root_hi) is tested before anything else
    int stack_need = 0;
    std::vector<int> leaf_tri;  // leaf record k (id k + 1) -> wire triangle
};

int pack_scene(glrtx_ctx *c, std::string *err_out, Packed &P, const float *vert, size_t n_vert, const float *tri, size_t n_tri,
               const float *mat, size_t n_mat, const float *light, size_t n_light, const float *bvh, size_t n_nodes) {
    if ((n_vert && !vert) || (n_tri && !tri) || (n_mat && !mat) || (n_light && !light) || (n_nodes && !bvh))
        return pfail(c, err_out, GLRTX_EINVAL, "glrtx_upload_scene: NULL buffer with non-zero count");
    if (n_tri > (size_t)INT32_MAX / 4 || n_nodes > (size_t)INT32_MAX / 4 || n_vert > (size_t)INT32_MAX / 16)
        return pfail(c, err_out, GLRTX_EINVAL, "glrtx_upload_scene: scene too large for 32-bit indices");

    auto vidx = [&](float f, int &out) {
        if (!(f >= 0.0f) || (size_t)f >= n_vert) return false;
        out = (int)f;
        return true;
    };

    // ---- triangles: {v0, material} {v1-v0} {v2-v0}; normals {n0} {n1} {n2} -- staged per wire triangle here; the device gets one
    // record per LEAF of the tree (P.tris / P.nrms, filled behind the tree walk below)
    std::vector<float4> tris, nrms;
    tris.assign(4 * n_tri, make_float4(0.f, 0.f, 0.f, 0.f));  // 64-byte records, same shape as a fork record
    nrms.assign(3 * n_tri, make_float4(0.f, 0.f, 0.f, 0.f));
    for (size_t t = 0; t < n_tri; t++) {
        int i[3];
        for (int k = 0; k < 3; k++)
            if (!vidx(tri[4 * t + k], i[k])) return pfail(c, err_out, GLRTX_ESCENE, "triangle %zu: vertex index %g out of range", t, tri[4 * t + k]);
        const float mf = tri[4 * t + 3];
        if (!(mf >= 0.0f) || (size_t)mf >= n_mat) return pfail(c, err_out, GLRTX_ESCENE, "triangle %zu: material %g out of range", t, mf);
        const float *p0 = vert + 15 * (size_t)i[0], *p1 = vert + 15 * (size_t)i[1], *p2 = vert + 15 * (size_t)i[2];
        tris[4 * t + 0] = make_float4(p0[0], p0[1], p0[2], as_float((int)mf));
        tris[4 * t + 1] = make_float4(p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2], 0.f);
        tris[4 * t + 2] = make_float4(p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2], 0.f);
        nrms[3 * t + 0] = make_float4(p0[3], p0[4], p0[5], 0.f);
        nrms[3 * t + 1] = make_float4(p1[3], p1[4], p1[5], 0.f);
        nrms[3 * t + 2] = make_float4(p2[3], p2[4], p2[5], 0.f);
    }

    // ---- materials: {emission, type} {param0, alpha.x} {param1, alpha.y}
    std::vector<float4> &mats = P.mats;
    mats.assign(3 * n_mat, make_float4(0.f, 0.f, 0.f, 0.f));
    for (size_t m = 0; m < n_mat; m++) {
        const float *r = mat + 18 * m;
        mats[3 * m + 0] = make_float4(r[3], r[4], r[5], as_float((int)r[0]));
        mats[3 * m + 1] = make_float4(r[6], r[7], r[8], r[12]);
        mats[3 * m + 2] = make_float4(r[9], r[10], r[11], r[13]);
    }

    // ---- lights: {v0, material} {v1} {v2} {n0} {n1} {n2}
    std::vector<float4> &lights = P.lights;
    lights.assign(6 * n_light, make_float4(0.f, 0.f, 0.f, 0.f));
    for (size_t l = 0; l < n_light; l++) {
        int i[3];
        for (int k = 0; k < 3; k++)
            if (!vidx(light[4 * l + k], i[k])) return pfail(c, err_out, GLRTX_ESCENE, "light %zu: vertex index out of range", l);
        const float mf = light[4 * l + 3];
        if (!(mf >= 0.0f) || (size_t)mf >= n_mat) return pfail(c, err_out, GLRTX_ESCENE, "light %zu: material out of range", l);
        for (int k = 0; k < 3; k++) {
            const float *p = vert + 15 * (size_t)i[k];
            lights[6 * l + k] = make_float4(p[0], p[1], p[2], k == 0 ? as_float((int)mf) : 0.f);
            lights[6 * l + 3 + k] = make_float4(p[3], p[4], p[5], 0.f);
        }
    }

    // ---- BVH: walk the wire-format tree from node 0, fold leaves into refs, renumber forks in DFS order.
    // Leaf records.  Every leaf the walk reaches gets its own triangle record, numbered in the order the traversal meets them:
    // leaf record k has triangle id k + 1 and lies at record index ~(k + 1) (id 0 / index -1 is the all-zero never-hit record that
    // stands for an absent child).  The id -- not the wire triangle index -- is what the kernels carry in a hit.
    // Leaf pairs.  A fork whose two children are both leaves gets NO fork record: nothing of it is ever tested but its own box, and
    // that sits in its parent's record.  Its two triangle records are chained instead -- the first-visited child's record names the
    // other one's index (its predecessor in memory) in the spare word of its second float4, REF_FIN otherwise -- and the parent refers
    // to the first of them: the traversal step goes from one triangle to the next without a fork fetch, a push and a pop in
    // between (raytrace.frag:299-331: children.y is popped and tested first, then children.x; neither is box-tested).  With the
    // reference's builder 40 % of the forks are such pairs and 94 % of the headline config's triangle tests come through one.
    std::vector<float4> &forks = P.forks;
    forks.clear();
    std::vector<int> leaf_tri;   // leaf record k -> wire triangle
    std::vector<int> leaf_next;  // leaf record k -> ref of the record chained behind it, or REF_FIN
    int root_ref = REF_ABSENT;
    int stack_need = 0;
    const bool chain_pairs = std::getenv("GLRTX_NO_LEAF_CHAINS") == nullptr;  // (A/B switch)
    if (n_nodes > 0 && n_tri > 0) {
        std::vector<int> ref_of(n_nodes, 0);          // ref assigned to each wire node
        std::vector<unsigned char> seen(n_nodes, 0);
        struct Frame { int node; int stage; };
        std::vector<Frame> st;
        std::vector<int> need(n_nodes, 0);
        auto is_fork = [&](int n) { return bvh[9 * (size_t)n + 8] < 0.0f; };
        auto child = [&](int n, int k, int &out) {  // k = 0 left (children.x), 1 right (children.y)
            const float f = bvh[9 * (size_t)n + 6 + k];
            if (!(f >= 0.0f)) { out = -1; return true; }
            if ((size_t)f >= n_nodes) return false;
            out = (int)f;
            return true;
        };
        auto add_leaf = [&](int n, int &rc) {  // a leaf record for wire node n; returns its ref
            rc = GLRTX_OK;
            if (seen[n]) { rc = pfail(c, err_out, GLRTX_ESCENE, "BVH node %d is referenced more than once (not a tree)", n); return 0; }
            seen[n] = 1;
            const float tf = bvh[9 * (size_t)n + 8];
            if ((size_t)tf >= n_tri) { rc = pfail(c, err_out, GLRTX_ESCENE, "BVH leaf %d: triangle %g out of range", n, tf); return 0; }
            leaf_tri.push_back((int)tf);
            leaf_next.push_back(REF_FIN);
            ref_of[n] = ~(int)leaf_tri.size();  // id = record count so far (ids start at 1)
            need[n] = 0;
            return ref_of[n];
        };
        st.push_back({0, 0});
        while (!st.empty()) {
            Frame &f = st.back();
            const int n = f.node;
            if (f.stage == 0) {
                if (!is_fork(n)) {
                    int rc;
                    add_leaf(n, rc);
                    if (rc) return rc;
                    st.pop_back();
                    continue;
                }
                if (seen[n]) return pfail(c, err_out, GLRTX_ESCENE, "BVH node %d is referenced more than once (not a tree)", n);
                seen[n] = 1;
                int l, r;
                if (!child(n, 0, l) || !child(n, 1, r)) return pfail(c, err_out, GLRTX_ESCENE, "BVH node %d: child index out of range", n);
                if (chain_pairs && l >= 0 && r >= 0 && l != r && !is_fork(l) && !is_fork(r)) {  // a leaf pair: children.y first, then children.x
                    int rc;
                    const int first = add_leaf(r, rc);
                    if (rc) return rc;
                    const int second = add_leaf(l, rc);
                    if (rc) return rc;
                    leaf_next[(size_t)~first - 1] = second;  // == first - 1: the record in front of it in memory
                    ref_of[n] = first;
                    need[n] = 0;
                    st.pop_back();
                    continue;
                }
                ref_of[n] = (int)(forks.size() / 4);
                // both children start out absent: the never-hit record (index -1) behind an infinite box (see put_box)
                for (int k = 0; k < 4; k++) { const float e = (k & 1) ? kInf : -kInf; forks.push_back(make_float4(e, e, e, as_float(-1))); }
                // forks are numbered in the order the traversal meets them (children.y first, raytrace.frag:299-307):
                // the first-visited child's record directly follows its parent's
                f.stage = 1;
                if (r >= 0) { st.push_back({r, 0}); }
                continue;
            }
            if (f.stage == 1) {
                f.stage = 2;
                int l;
                child(n, 0, l);
                if (l >= 0) { st.push_back({l, 0}); }
                continue;
            }
            int l, r;
            child(n, 0, l);
            child(n, 1, r);
            const int fi = ref_of[n];
            // child box into the parent's record.  A LEAF child is never box-tested (raytrace.frag:310-331): it gets the box
            // (-inf, +inf), for which the slab test passes by itself with entry distance -inf -- the traversal step then needs no
            // test of the ref's sign (an absent child, the never-hit record, is treated the same way).  A leaf PAIR is a fork in the
            // wire format: its own box goes here like any fork's.
            auto put_box = [&](int slot, int child) {
                if (!is_fork(child)) return;
                const float *b = bvh + 9 * (size_t)child;
                forks[4 * fi + slot].x = b[0]; forks[4 * fi + slot].y = b[1]; forks[4 * fi + slot].z = b[2];
                forks[4 * fi + slot + 1].x = b[3]; forks[4 * fi + slot + 1].y = b[4]; forks[4 * fi + slot + 1].z = b[5];
            };
            // Traversal continues with the right child while the left one waits on the stack.  trav_step treats an absent child like a
            // leaf -- always "passed", never hit -- so a fork with one child still pushes one entry: its only child is stored in the
            // LEFT slot whichever wire slot it came from.  The step then continues with the never-hit record and pops the child right
            // after it, with tHit unchanged in between: the same visits in the same order, and the absent entry never sits on the
            // stack underneath a whole subtree (a chain of one-child forks needs ONE entry, like the reference's own stack).
            if (l >= 0 && r >= 0) {
                put_box(0, l); forks[4 * fi].w = as_float(ref_of[l]);
                put_box(2, r); forks[4 * fi + 1].w = as_float(ref_of[r]);
                need[n] = std::max(1 + need[r], need[l]);
            } else if (l >= 0 || r >= 0) {
                const int only = l >= 0 ? l : r;
                put_box(0, only); forks[4 * fi].w = as_float(ref_of[only]);
                need[n] = std::max(1, need[only]);  // the push of the child itself, then the child's own subtree from an empty stack
            } else need[n] = 1;  // both absent: the left never-hit record is pushed, the right one "visited"
            st.pop_back();
        }
        root_ref = ref_of[0];
        stack_need = need[0];
        P.root_boxed = is_fork(0) ? 1 : 0;  // the wire root's own box is tested first (:296-298) -- also when it is packed as a leaf pair

        // the per-leaf triangle records: id 0 the never-hit record, id k + 1 leaf record k
        P.tris.assign(4 * (leaf_tri.size() + 1), make_float4(0.f, 0.f, 0.f, 0.f));
        P.nrms.assign(3 * (leaf_tri.size() + 1), make_float4(0.f, 0.f, 0.f, 0.f));
        for (size_t k = 0; k < leaf_tri.size(); k++) {
            for (int j = 0; j < 4; j++) P.tris[4 * (k + 1) + j] = tris[4 * (size_t)leaf_tri[k] + j];
            for (int j = 0; j < 3; j++) P.nrms[3 * (k + 1) + j] = nrms[3 * (size_t)leaf_tri[k] + j];
            P.nrms[3 * (k + 1)].w = P.tris[4 * (k + 1)].w;  // the material id rides with the normals (round 6): the shade phase reads it there and no longer fetches the leaf record for one word
            P.tris[4 * (k + 1) + 1].w = as_float(leaf_next[k]);
        }
        P.tris[1].w = as_float(REF_FIN);

        // ---- vine: every fork has a leaf as children.y and the chain continues through children.x (what
        // glrt_bvh_build_chain emits for "brute force, no BVH").  Stored additionally as the list the traversal visits:
        // record i = {box of fork i, its children.y triangle}; the last record = the last fork's children.x leaf.
        if (is_fork(0) && leaf_tri.size() >= 2) {
            std::vector<float4> v;
            v.reserve(4 * leaf_tri.size());
            bool is_vine = true, uniform = true;
            int n = 0;
            auto rec = [&](const float *lo, const float *hi, int t) {  // t: triangle id of a leaf record
                const float4 *T = &P.tris[4 * (size_t)t];  // {v0, mat} {e1} {e2}
                v.push_back(make_float4(lo[0], lo[1], lo[2], T[0].x));
                v.push_back(make_float4(hi[0], hi[1], hi[2], T[0].y));
                v.push_back(make_float4(T[0].z, T[1].x, T[1].y, T[1].z));
                v.push_back(make_float4(T[2].x, T[2].y, T[2].z, as_float(t)));
            };
            const float inf = std::numeric_limits<float>::infinity();
            const float all_lo[3] = {-inf, -inf, -inf}, all_hi[3] = {inf, inf, inf};
            while (true) {
                int l, r;
                child(n, 0, l);
                child(n, 1, r);
                if (l < 0 || r < 0 || is_fork(r)) { is_vine = false; break; }
                const float *b = bvh + 9 * (size_t)n;
                if (std::memcmp(b, bvh, 6 * sizeof(float)) != 0) uniform = false;
                rec(b, b + 3, ~ref_of[r]);
                if (!is_fork(l)) { rec(all_lo, all_hi, ~ref_of[l]); break; }
                n = l;
            }
            if (is_vine && v.size() == 4 * leaf_tri.size()) {
                // Device layout (trav_scan): the n - 1 fork records in groups of four -- the last group filled up with never-hit records (an infinite box, an
                // all-zero triangle: det = 0) --, then the last leaf's record at index vine_main = roundup4(n - 1), then three more never-hit records: the scan
                // works through whole groups while the next ones are being fetched and reads up to three records past the one it needs.
                const size_t n = leaf_tri.size(), n_main = (n - 1 + 3) / 4 * 4;
                std::vector<float4> d(4 * (n_main + 4));
                const float4 pad[4] = {make_float4(-inf, -inf, -inf, 0.f), make_float4(inf, inf, inf, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, as_float(0))};
                for (size_t i = 0; i < n_main + 4; i++) {
                    const float4 *src = i < n - 1 ? &v[4 * i] : (i == n_main ? &v[4 * (n - 1)] : pad);
                    for (int j = 0; j < 4; j++) d[4 * i + j] = src[j];
                }
                P.vine.swap(d);
                P.n_vine = (int)n;
                P.vine_main = (int)n_main;
                P.vine_uniform = uniform ? 1 : 0;
            }
        }
    }
    // Renumbering: the forks of the tree's top levels (breadth-first from the root, kTopForks of them) take the first indices, the
    // others follow in the order the traversal meets them.  The top of the tree is then one contiguous, hot 8 KiB block.
    if (root_ref >= 0 && forks.size() / 4 > 1) {
        const int nf = (int)(forks.size() / 4);
        std::vector<int> newid(nf, -1), order;
        order.reserve(nf);
        order.push_back(root_ref);
        newid[root_ref] = 0;
        for (size_t q = 0; q < order.size() && (int)order.size() < kTopForks; q++) {
            const int f = order[q];
            int refs[2]; std::memcpy(&refs[0], &forks[4 * f + 0].w, 4); std::memcpy(&refs[1], &forks[4 * f + 1].w, 4);
            for (int k = 1; k >= 0; k--)  // children.y first, as the traversal
                if (refs[k] >= 0 && refs[k] != REF_ABSENT && newid[refs[k]] < 0 && (int)order.size() < kTopForks) { newid[refs[k]] = (int)order.size(); order.push_back(refs[k]); }
        }
        for (int f = 0; f < nf; f++) if (newid[f] < 0) { newid[f] = (int)order.size(); order.push_back(f); }
        std::vector<float4> re(forks.size());
        for (int f = 0; f < nf; f++) {
            for (int k = 0; k < 4; k++) re[4 * (size_t)newid[f] + k] = forks[4 * (size_t)f + k];
            for (int k = 0; k < 2; k++) {
                int r; std::memcpy(&r, &forks[4 * (size_t)f + k].w, 4);
                if (r >= 0) { r = newid[r]; std::memcpy(&re[4 * (size_t)newid[f] + k].w, &r, 4); }
            }
        }
        forks.swap(re);
        root_ref = newid[root_ref];
    }
    if (root_ref == REF_ABSENT) {  // empty scene: one childless fork, every ray misses
        forks.clear();
        for (int k = 0; k < 4; k++) { const float e = (k & 1) ? kInf : -kInf; forks.push_back(make_float4(e, e, e, as_float(-1))); }
        root_ref = 0;
        stack_need = 1;  // a ray that passes the (degenerate) root box pushes the left never-hit record
        P.root_boxed = 1;
        P.root_lo = make_float4(0.f, 0.f, 0.f, 0.f);
        P.root_hi = make_float4(0.f, 0.f, 0.f, 0.f);
        P.tris.assign(4, make_float4(0.f, 0.f, 0.f, 0.f));  // the never-hit record alone
        P.tris[1].w = as_float(REF_FIN);
        P.nrms.assign(3, make_float4(0.f, 0.f, 0.f, 0.f));
    } else if (P.root_boxed) {
        P.root_lo = make_float4(bvh[0], bvh[1], bvh[2], 0.f);
        P.root_hi = make_float4(bvh[3], bvh[4], bvh[5], 0.f);
    }
    if (stack_need > 63)
        return pfail(c, err_out, GLRTX_EDEPTH, "BVH needs %d traversal stack entries; the reference shader's stack holds 64", stack_need + 1);

    P.root_ref = root_ref;
    P.stack_need = stack_need;
    P.leaf_tri.swap(leaf_tri);
    return GLRTX_OK;
}


int ensure(glrtx_ctx *c, DevBuf &b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return GLRTX_OK;
    dev_free(b);
    HIP_TRY(c, hipMalloc(&b.p, std::max<size_t>(bytes, 64)));
    b.bytes = std::max<size_t>(bytes, 64);
    return GLRTX_OK;
}

#ifdef GLRTX_RAY_LOG
struct DbgLastLaunch { KernelArgs a; WfArgs w; int lds, grid; float4 *queues; int fetch; } g_dbg_last;  // what the last pt_render_wgwf launch was given (replay)
DevBuf g_dbg_log_rays, g_dbg_log_trips;
#endif

// Frames one launch may cover: bounded by a FIXED memory budget for what a launch needs PER FRAME -- one float4 plane per sample and owned pixel -- not by what happens
// to be free on the device, so that the launch shapes, and with them the timing, do not depend on what else runs there (path state does not count: it is addressed by
// workgroup and queue position, 0.8 GB whatever the frames in flight), and by the 31-bit path ids (frame * pixels + pixel).  GLRTX_FRAMES_BUDGET_MB overrides the default (tests).
// `slots`: the budget is shared by that many launches that may be in flight at once (fed launches: kFedSlots).
int frames_cap(const glrtx_ctx *c, const glrtx_params *p, int slots) {
    const size_t px = (size_t)((c->width + 7) / 8) * (size_t)((c->owned_rows + 7) / 8) * 64;
    const size_t per_frame = (size_t)std::max(p->n_samples, 1) * c->pitch_bytes * (size_t)std::max(c->owned_rows, 1);
    size_t budget = (size_t)kFramesBudgetGiB << 30;
    if (const char *v = std::getenv("GLRTX_FRAMES_BUDGET_MB")) budget = (size_t)std::max(1, std::atoi(v)) << 20;
    budget /= (size_t)std::max(c->budget_share, 1) * (size_t)std::max(slots, 1);
    const size_t id_cap = (((size_t)1 << 31) - 2) / std::max<size_t>(px, 1);  // path ids (frame * pixels + pixel) stay below 2^31
    return (int)std::max<size_t>(1, std::min<size_t>(id_cap, budget / std::max<size_t>(per_frame, 1)));
}

constexpr unsigned kFedSlots = 3;  // fed launches alternate between this many of the context's pipe slots (one renders, one drains into its accumulation pass, one is being fed)

// Nothing is appended to the open launch any more (glrtx_ctx::OpenFeed).  Host side only: the launch itself closes its feed when it runs dry.
void seal_feed(glrtx_ctx *c) {
    c->open.slot = nullptr;
    c->last_was_render = false;
}

bool same_camera(const glrtx_params &a, const glrtx_params &b) {
    return std::memcmp(a.c2w, b.c2w, sizeof a.c2w) == 0 && std::memcmp(a.s2c, b.s2c, sizeof a.s2c) == 0 && std::memcmp(&a.aperture, &b.aperture, 4) == 0 &&
           std::memcmp(&a.focal, &b.focal, 4) == 0 && a.n_samples == b.n_samples && a.max_depth == b.max_depth;
}

// The sample planes of frames [first, first + n) of the open launch: chunks of kFeedChunkFrames frames, allocated on demand -- also while the launch is running (an
// allocation does not wait for the device; nothing is ever freed here) -- and published through the feed block.
int feed_ensure_chunks(glrtx_ctx *c, glrtx_ctx::PipeSlot &sl, int first, int n, int cap, size_t frame_bytes) {
    for (int k = first / kFeedChunkFrames; k <= (first + n - 1) / kFeedChunkFrames; k++) {
        const int frames_in_chunk = std::min(kFeedChunkFrames, cap - k * kFeedChunkFrames);
        const size_t need = (size_t)std::max(frames_in_chunk, 1) * frame_bytes;
        if (!sl.chunks[k].p || sl.chunks[k].bytes < need) {
            if (sl.chunks[k].p) return fail(c, GLRTX_EDEVICE, "internal: plane chunk %d of a fed launch is too small (%zu < %zu bytes)", k, sl.chunks[k].bytes, need);
            HIP_TRY(c, hipMalloc(&sl.chunks[k].p, std::max<size_t>(need, 64)));
            sl.chunks[k].bytes = std::max<size_t>(need, 64);
        }
        sl.feed_h->chunks[k] = (float4 *)sl.chunks[k].p;
    }
    return GLRTX_OK;
}

// Append n frames to the open launch.  Returns 1 if the launch has taken them (nothing else to do), 0 if the caller must start a launch of its own (no open launch, another
// camera, the launch is full -- or it has closed its feed: the compare-and-swap on the published count fails).  Never an error: whatever goes wrong here, a launch can still be made.
int feed_append(glrtx_ctx *c, const glrtx_params *p, const float *seeds_xy, int n) {
    glrtx_ctx::OpenFeed &o = c->open;
    if (!o.slot) return 0;
    if (!same_camera(o.p, *p) || o.frames + n > o.cap) { o.slot = nullptr; return 0; }
    if (feed_ensure_chunks(c, *o.slot, o.frames, n, o.cap, o.frame_bytes) != GLRTX_OK) { (void)hipGetLastError(); c->err.clear(); o.slot = nullptr; return 0; }
    FeedHost *fh = o.slot->feed_h;
    for (int i = 0; i < n; i++) fh->seeds[o.frames + i] = make_float2(seeds_xy[2 * i], seeds_xy[2 * i + 1]);
    unsigned expect = (unsigned)o.frames;  // (release: seeds and chunk pointers before the count that announces them)
    if (!__atomic_compare_exchange_n(&fh->frames_pub, &expect, (unsigned)(o.frames + n), false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) { o.slot = nullptr; return 0; }
    o.frames += n;
    o.rec->frames = o.frames;
    c->st.frames_last = o.frames;
    c->st.feed_appended += (uint64_t)n;
    c->st.launches += (uint64_t)n;
    c->st.paths += (uint64_t)c->owned_rows * (uint64_t)c->width * (uint64_t)p->n_samples * (uint64_t)n;
    c->counters_stale = c->counters_stale || c->count_rays;
    return 1;
}

// Variant 2: one persistent launch; every workgroup runs the wavefront trips of the pixels it takes from the frame's tile counter.
// n_frames > 1 ("frames in flight"): the launch covers n_frames consecutive frames that differ only in u_seed (seeds_xy);
// the per-sample planes are added to the accumulator in frame order afterwards, so the result is bit-identical to
// n_frames separate launches.  Three forms (chosen here):
//   fed        -- on one of the context's pipe slots (own stream, state, queues, tile counter, plane chunks); the launch stays OPEN: later calls with the same camera
//                 append their frames to it while it runs (feed_append).  Every multi-frame launch on the context's own stream, and a single-frame launch that
//                 follows another one directly (a burst)
//   overlapped -- a single-frame launch on a pipe slot, the frame spread over all workgroup slots in one helping, slots handed over progressively (round 3): what a
//                 caller gets that renders, resolves and saves every frame -- the reference's loop, window.cpp:121-169 -- and any single frame on a caller's stream
//   plain      -- on the context's stream, nothing overlaps: multi-frame launches on a caller's stream (stream order is the caller's), GLRTX_NO_PIPELINE=1
int launch_wgwf(glrtx_ctx *c, const KernelArgs &a_in, const glrtx_params *p, const float *seeds_xy, int n_frames) {
    KernelArgs a = a_in;
    const int tiles8_x = (c->width + 7) / 8, tiles8_y = (c->owned_rows + 7) / 8;
    const size_t total = (size_t)tiles8_x * tiles8_y * 64;
    if (total * 2 >= (size_t)INT32_MAX) return fail(c, GLRTX_EINVAL, "image too large for the wgwf variant");
    const size_t plane_bytes = (size_t)a.pitch_f4 * (size_t)c->owned_rows * sizeof(float4);
    const size_t frame_bytes = (size_t)std::max(p->n_samples, 1) * plane_bytes;
    int rc;
    // ---- which form
    // (busy: the context's last render kernel -- a slot's, or one on the context's stream -- has not ended yet)
    const bool busy = c->last_render_done && hipEventQuery(c->last_render_done) == hipErrorNotReady;
    (void)hipGetLastError();
    const bool burst = c->last_was_render && busy && same_camera(c->last_p, *p);
    bool fed = c->feed_ok && std::getenv("GLRTX_NO_FEED") == nullptr && c->pipeline && c->stream == c->own_stream && p->n_samples >= 1 && seeds_xy != nullptr && (n_frames > 1 || burst);
    int fed_cap = fed ? std::min(frames_cap(c, p, (int)kFedSlots), kFeedMaxFrames) : 0;
    if (const char *v = std::getenv("GLRTX_FEED_CAP")) fed_cap = std::max(1, std::min(fed_cap, std::atoi(v)));  // (tests: launches that fill up)
    if (fed && n_frames > fed_cap) fed = false;
    bool piped = fed || (n_frames == 1 && c->pipeline && p->n_samples >= 1 && (size_t)p->n_samples * plane_bytes <= ((size_t)1 << 30));
    const size_t ids = total * (size_t)(fed ? fed_cap : n_frames);  // path id = frame * total + pixel
    if (ids + 1 >= (size_t)UINT32_MAX)
        return fail(c, GLRTX_EINVAL, "glrtx_render_frames: %d frames of %zu pixels exceed the 32-bit ray id space", n_frames, total);

    // ---- the kernel and how many of its workgroups fit the device.  Eight instantiations: ray counting on/off x (list scan of a vine (brute-force) tree | tree traversal
    // with one record per lane | with the pair-cooperative node fetch | with the two in alternate steps).  The pair fetch trades 25 vector-ALU instructions per step for a
    // third less time in the CU's vector-memory pipe (trav_asm.hip.h): it pays where a wave's lanes are spread over many records -- large trees, incoherent rays: config 5
    // (100 k triangles) -9 % per frame -- and costs ~1 % where they share the top of a small tree (headline, 10 k triangles: the step is paced by instruction issue there),
    // profiles/r04_ab_pair_fetch.txt.  Picked by the size of the record array; GLRTX_PAIR_FETCH=0/1 overrides.  All forms are bit-identical (same IEEE operations on the same record).
    // On small trees the two forms in alternate steps (FETCH 2: the pipe is the busier unit in one step, the SIMDs in the next) beat both by 1.0 .. 1.3 % in round 4.
    // Since the path state is read and written by queue position (WfArgs::state) the pipe has a fifth less to do and the plain form wins there: headline -0.7 %,
    // config 2 -0.4 %, config 4 -1.0 % against the alternating one, the pure pair form +1.9 % (profiles/r05_state_by_position.txt).  The alternating form stays
    // compiled in (GLRTX_PAIR_FETCH=2).  Random triangle soups prefer the pair form from 10 k triangles on (-2 %; 20 k: -4.5 %, 70 k: -7 %): what decides is how
    // far apart a wave's rays are in the tree, which the record count only approximates.
    using Kernel = void (*)(const KernelArgs, const WfArgs, unsigned *, float4 *);
    const bool vine = c->sc.n_vine > 0;
    int fetch = vine ? 0 : ((size_t)c->n_fork + (size_t)c->st.n_tri >= (size_t)kPairFetchMinRecords ? 1 : 0);
    if (const char *v = std::getenv("GLRTX_PAIR_FETCH")) fetch = vine ? 0 : std::max(0, std::min(2, std::atoi(v)));
    c->st.node_fetch_last = fetch;
    const bool cr = c->count_rays;
    const Kernel kernel = vine ? (cr ? (Kernel)pt_render_wgwf<true, true> : (Kernel)pt_render_wgwf<false, true>)
                        : fetch == 2 ? (cr ? (Kernel)pt_render_wgwf<true, false, 2> : (Kernel)pt_render_wgwf<false, false, 2>)
                        : fetch == 1 ? (cr ? (Kernel)pt_render_wgwf<true, false, 1> : (Kernel)pt_render_wgwf<false, false, 1>)
                                     : (cr ? (Kernel)pt_render_wgwf<true, false, 0> : (Kernel)pt_render_wgwf<false, false, 0>);
    // (north_star's "primitives staged into LDS": materials, camera block, root box and the per-lane stacks are; the top tree levels were built, measured worth
    // nothing -- profiles/r02_lds_top.json -- and removed.)
    const int lds = c->sc.lds_head_f4 * (int)sizeof(float4) + 2 * c->sc.stack_entries * kBlockThreads * (int)sizeof(int) +
                    kWgCtlWords * (int)sizeof(unsigned) + 2 * (int)sizeof(float4) + kCamFloatsPadded * (int)sizeof(float) + kLdsSeeds * (int)sizeof(float2) +  // ctl | root box | camera block | seeds |
                    kWgPathsMax / 8;  // light-test bits, one per path-queue position
    if (lds > 160 * 1024) return fail(c, GLRTX_EDEVICE, "wgwf kernel needs %d B of LDS (> 160 KiB)", lds);
    if (lds > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int per_cu = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlockThreads, lds));
    // (launch_bounds' second argument is a minimum, not a cap: an instantiation with little LDS may be reported with more workgroups per CU than the buffers below are
    //  sized for -- ADVICE round 5)
    per_cu = std::max(1, std::min(per_cu, GLRTX_WGWF_WAVES));
    if (const char *v = std::getenv("GLRTX_WGS_PER_CU")) per_cu = std::max(1, std::min(per_cu, std::atoi(v)));  // occupancy experiments
    const size_t wg_slots = (size_t)c->n_cu * (size_t)per_cu;
    const size_t queue_bytes_max = wg_slots * kWgQueueF4 * sizeof(float4);

    // ---- a pipe slot (fed and overlapped launches).  A slot owns path state, per-workgroup queues and sample planes of its own (~0.8 GB at 1080p for an overlapped
    // single frame, 1.4 GB + planes for a fed launch): the slots are charged to the same memory budget as the frames in flight (kFramesBudgetGiB, split between the members
    // of a group that share this GPU), a slot whose buffers cannot be allocated is given up -- the launch then runs plain on the context's stream instead of failing -- and
    // among the slots in use the next one whose previous launch has completed is taken (round robin only when none has), so that a launch never queues behind a busy slot
    // while an idle one exists.
    // Shapes.  plain / fed: every workgroup keeps up to kWgPathsMax paths alive, less when the launch is short (below; a fed launch is sized as if it were to take
    // 16 frames: it may).  overlapped, issued while the context's last launch is still rendering: the frame in ONE helping per workgroup, spread over all the slots (the smallest power of two that holds
    // work / resident paths: 2048 at 1080p; with 1024 the workgroups come back for a second helping, 1.31 instead of 1.24 ms per frame, with 4096 half of them get
    // nothing, 1.67), every launch has the full grid and takes its tiles without guided self-scheduling, so its workgroups do not finish together -- one that finds the
    // tile counter exhausted and its paths dead leaves, and a workgroup of the next launch (queued on another stream) takes the slot (profiles/r03_ab_pipeline.txt).
    auto shape = [&](bool overlapped, bool is_fed, int share, bool device_busy, int &block_paths, int &grid) {
        const int resident = std::max(1, per_cu / std::max(share, 1)) * c->n_cu;
        const size_t work = total * (size_t)(is_fed ? std::max(n_frames, std::min(16, fed_cap)) : n_frames);
        block_paths = kWgPathsMax;
        if (overlapped && device_busy) { block_paths = 256; while (block_paths < kWgPathsMax && (size_t)resident * block_paths < work) block_paths *= 2; }
        else {
            // A launch ends with a tail in which its last paths run out -- about one path item's lifetime, and an item is a pixel's n_samples samples in sequence
            // (their random numbers are one chain) -- and nothing overlaps that tail when the launch is alone on the device (plain, fed, or a single frame issued
            // to an idle device).  Fewer paths per workgroup mean more, shorter helpings and a shorter tail, but slower trips: the most paths that still give the
            // launch two helpings per workgroup (several frames of one sample per pixel) or four (more samples: longer items; a single frame: its tail is as long
            // as the rest of it).  Lone launches, profiles/r06_launch_shapes.txt: 1080p 1 spp, one frame 2.4 -> 1.67 ms (config 2: 2.1 -> 1.39, config 3: 17.0 -> 11.4,
            // config 5: 4.6 -> 3.1); 16 spp, one frame 39 -> 24 ms, four frames 19.4 -> 18.5; config 4 (4K, 16 spp), one frame 51 -> 39 ms.
            const size_t tenths = (p->n_samples > 1 || (n_frames == 1 && !is_fed)) ? 39 : 19;
            while (block_paths > 512 && work * 10 < tenths * (size_t)resident * block_paths) block_paths /= 2;
            while (block_paths > 256 && work < (size_t)resident * block_paths) block_paths /= 2;  // (small images: every workgroup of the grid has a full helping)
        }
        if (const char *v = std::getenv("GLRTX_BLOCK_PATHS")) { const int x = std::atoi(v); if (x >= 256 && x <= kWgPathsMax && (x & (x - 1)) == 0) block_paths = x; }
        grid = std::max(1, (int)std::min<size_t>((size_t)resident, (work + block_paths - 1) / block_paths));
    };
    glrtx_ctx::PipeSlot *slot = nullptr;
    int pipe_busy = 0, block_paths = 0, grid = 0;
    if (piped) {
        shape(!fed, fed, 1, busy, block_paths, grid);
        // (an overlapped slot's path state holds either of a single frame's two shapes -- issued to an idle or to a busy device: growing it later would free memory, and
        //  freeing waits for the device)
        size_t slot_state_entries = (size_t)grid * block_paths;
        if (!fed) { int b2 = 0, g2 = 0; shape(true, false, 1, !busy, b2, g2); slot_state_entries = std::max(slot_state_entries, (size_t)g2 * b2); }
        const size_t per_slot = (fed ? 0 : (size_t)kWfStatePlanes * slot_state_entries * sizeof(float4)) + (size_t)grid * kWgQueueF4 * sizeof(float4) + (fed ? 0 : frame_bytes);
        size_t budget = (size_t)kFramesBudgetGiB << 30;
        if (const char *v = std::getenv("GLRTX_FRAMES_BUDGET_MB")) budget = (size_t)std::max(1, std::atoi(v)) << 20;
        budget /= (size_t)std::max(c->budget_share, 1);
        unsigned allowed = (unsigned)std::min<size_t>(c->pipe_slots, budget / std::max<size_t>(per_slot, 1));
        if (fed) allowed = std::min(allowed, kFedSlots);
        c->st.pipe_slots = (int32_t)allowed;
        if (allowed >= 1) {
            unsigned pick = c->pipe_next % allowed;
            for (unsigned k = 0; k < allowed; k++) {
                glrtx_ctx::PipeSlot &o = c->pipe[(c->pipe_next + k) % allowed];
                if (!o.used || hipEventQuery(o.acc_done) == hipSuccess) { pick = (c->pipe_next + k) % allowed; break; }
            }
            (void)hipGetLastError();
            c->pipe_next = pick + 1;
            slot = &c->pipe[pick];
            for (unsigned k = 0; k < kPipeSlots; k++)
                if (k != pick && c->pipe[k].used && hipEventQuery(c->pipe[k].render_done) == hipErrorNotReady) pipe_busy++;
            (void)hipGetLastError();
            if (c->pipe_share > 1 && pipe_busy > 0 && !fed) shape(true, false, c->pipe_share, busy, block_paths, grid);  // (GLRTX_PIPE_SHARE: the round-3 form, A/B only)
            // the slot's buffers, before anything depends on them.  A fed launch's plane chunks hold a given number of bytes per frame: another size starts afresh
            // (the slot's last launch is waited for first: freeing memory waits for the device anyway)
            bool ok = true;
            // a fed launch writes the slot's feed block from the host: the slot's previous launch, which reads it, must have ended (with every fed slot busy the host
            // waits here: it is three launches ahead of the device)
            if (fed && slot->used) (void)hipEventSynchronize(slot->render_done);
            if (fed && slot->chunk_bytes != frame_bytes) {
                if (slot->used) (void)hipEventSynchronize(slot->acc_done);
                for (auto &ch : slot->chunks) dev_free(ch);
                std::memset(slot->feed_h->chunks, 0, sizeof slot->feed_h->chunks);
                slot->chunk_bytes = frame_bytes;
            }
            if (!fed) ok = ok && ensure(c, slot->state, (size_t)kWfStatePlanes * slot_state_entries * sizeof(float4)) == GLRTX_OK;
            ok = ok && ensure(c, slot->queues, (size_t)grid * kWgQueueF4 * sizeof(float4)) == GLRTX_OK;
            if (fed) ok = ok && ensure(c, slot->feed_d, sizeof(FeedDev)) == GLRTX_OK && feed_ensure_chunks(c, *slot, 0, n_frames, fed_cap, frame_bytes) == GLRTX_OK;
            else ok = ok && ensure(c, slot->planes, frame_bytes) == GLRTX_OK;
            if (!ok) {
                (void)hipGetLastError();
                c->err.clear();
                if (slot->used) (void)hipEventSynchronize(slot->acc_done);
                // memory is what ran out: this slot and the ones behind it give theirs back (they would never be picked again: pipe_slots shrinks to `pick`)
                for (unsigned k = pick; k < kPipeSlots; k++) {
                    glrtx_ctx::PipeSlot &o = c->pipe[k];
                    if (o.used) (void)hipEventSynchronize(o.acc_done);
                    dev_free(o.state); dev_free(o.planes); dev_free(o.queues); dev_free(o.feed_d);
                    for (auto &ch : o.chunks) dev_free(ch);
                    if (o.feed_h) std::memset(o.feed_h->chunks, 0, sizeof o.feed_h->chunks);
                    o.chunk_bytes = 0;
                    o.used = false;
                }
                (void)hipGetLastError();
                c->pipe_slots = pick;            // the slots in front of this one keep their buffers and stay in use
                if (pick == 0) c->pipeline = false;
                c->st.pipe_slots = (int32_t)pick;
                slot = nullptr;
                pipe_busy = 0;
            }
        }
        if (!slot) { piped = false; fed = false; }
    }
    if (!slot) {
        if (n_frames > frames_cap(c, p, 1)) return fail(c, GLRTX_ENOMEM, "glrtx_render_frames: %d frames exceed the frames-in-flight memory budget", n_frames);
        shape(false, false, 1, busy, block_paths, grid);
    }
    if (slot) c->st.pipe_resident_max = std::max<int32_t>(c->st.pipe_resident_max, pipe_busy + 1);
    // A fed launch runs on its slot's stream and queues but on the CONTEXT's path state, like a plain launch: the same launch is 0.8 % faster there than on a slot's state
    // buffer (same-context A/B, two boxes, profiles/r06_ab_feed_cadence.txt -- where a 0.8-GB buffer lands physically decides what its stream of stores costs:
    // profiles/r04_context_regimes.txt).  Launches that share that buffer run one after the other (state_done below): fed launches end all their workgroups together,
    // so there was nothing for them to overlap with anyway.
    const bool ctx_state = !slot || fed;
    DevBuf &stateBuf = ctx_state ? c->wfState : slot->state;
    DevBuf &queueBuf = slot ? slot->queues : c->wfQ;
    DevBuf &planeBuf = slot ? slot->planes : c->wfPlanes;
    unsigned *const workPtr = (unsigned *)(slot ? slot->work.p : c->work.p);
    const hipStream_t rstream = slot ? slot->stream : c->stream;  // the render kernel's stream
    // path state: two sets of six planes, an entry per workgroup of THIS launch and path-queue position (WfArgs::state): 0.8 GB for a full grid with kWgPathsMax paths
    // each, kilobytes for a small image (it used to be sized for the largest launch the device can hold, whatever the launch)
    const size_t state_entries = (size_t)grid * (size_t)block_paths;
    const size_t state_bytes = (size_t)kWfStatePlanes * state_entries * sizeof(float4);
    if ((rc = ensure(c, stateBuf, state_bytes))) return rc;
    if ((rc = ensure(c, queueBuf, (size_t)grid * kWgQueueF4 * sizeof(float4)))) return rc;  // per-workgroup queues
    WfArgs w;
    std::memset(&w, 0, sizeof w);
    w.state = (float4 *)stateBuf.p;
    w.ids = state_entries;  // (the plane stride)
    c->st.wf_state_mib = (int32_t)((state_bytes + ((size_t)1 << 20) - 1) >> 20);
    w.total = (int)total;
    w.tiles8_x = tiles8_x;
    w.refill_min = kRefillMin;
    if (const char *v = std::getenv("GLRTX_REFILL_MIN")) w.refill_min = std::max(1, std::min(64, std::atoi(v)));
    w.n_frames = fed ? std::max(fed_cap, 2) : n_frames;  // (fed: the most frames the launch can take -- and > 1, so that path ids are split into frame and pixel)
    w.tiles_per_frame = (int)(total >> 6);
    const size_t plane_f4 = (size_t)a.pitch_f4 * (size_t)c->owned_rows;
    const int n_planes = n_frames * p->n_samples;
    if (fed) {
        FeedHost *fh = slot->feed_h;
        for (int i = 0; i < n_frames; i++) fh->seeds[i] = make_float2(seeds_xy[2 * i], seeds_xy[2 * i + 1]);
        __atomic_store_n(&fh->frames_pub, (unsigned)n_frames, __ATOMIC_RELEASE);
        w.feed_host = slot->feed_h_dev;
        w.feed_dev = (FeedDev *)slot->feed_d.p;
        w.feed_plane_f4 = plane_f4;
        w.feed_margin = 64 * 4 * grid;  // where guided self-scheduling starts to taper the helpings (gss_div x 64 tiles)
        if (const char *v = std::getenv("GLRTX_FEED_MARGIN")) w.feed_margin = std::max(0, std::atoi(v));
    } else {
        if (n_frames > 1) {
            if ((rc = ensure(c, c->wfSeeds, (size_t)n_frames * sizeof(float2)))) return rc;
            HIP_TRY(c, hipMemcpyAsync(c->wfSeeds.p, seeds_xy, (size_t)n_frames * sizeof(float2), hipMemcpyHostToDevice, c->stream));
            w.seeds = (const float2 *)c->wfSeeds.p;
            w.seeds_in_lds = n_frames <= kLdsSeeds ? 1 : 0;
        }
        if (n_frames > 1 || slot) {
            if ((rc = ensure(c, planeBuf, (size_t)std::max(n_planes, 1) * plane_f4 * sizeof(float4)))) return rc;
            w.planes = (float4 *)planeBuf.p;
        }
    }
    w.block_paths = block_paths;
    w.gss_div = (slot && !fed) ? 0 : 4 * grid;  // (overlapped single-frame launches: no guided self-scheduling, see `shape` above)
    if (const char *v = std::getenv("GLRTX_GSS_DIV")) w.gss_div = std::max(0, std::atoi(v));
    w.suspend_max = kSuspendMax;
    if (const char *v = std::getenv("GLRTX_SUSPEND_MAX")) w.suspend_max = std::max(0, std::min(64, std::atoi(v)));
    // trip guards (pt_render_wgwf): a path is alive for at most n_samples x (max_depth + 2) shaded trips; 64 times that (parked trips, the other paths' rounds) and a
    // constant are allowed between two tiles a workgroup is given
    w.trip_limit = (int)std::min<long long>(INT32_MAX, 64ll * std::max(p->n_samples, 1) * (p->max_depth + 2) + 64);
    if (const char *v = std::getenv("GLRTX_TRIP_LIMIT")) w.trip_limit = std::max(1, std::atoi(v));
    w.err = c->guard_dev;
    // Shape invariants of the hand-written kernel, checked on the host before every launch (an access past one of these
    // buffers is a GPU memory fault, not an error code): every path id the launch can form indexes inside the state arrays;
    // every workgroup of the grid has its own queue slice; a slice holds both halves of the double-buffered ray queue
    // (2 rays per live path: the next ray and the shadow ray) and of the path-id queue; every (frame, sample) has its plane.
    {
        const size_t max_id = ids - 1;
        const size_t max_sidx = (size_t)kWfSetPlanes * state_entries + (size_t)grid * block_paths - 1;  // the highest state index a ray record can carry (set 1)
        const size_t slice_f4 = 2 * (size_t)2 * 2 * block_paths /* ray records */ + (2 * (size_t)block_paths * sizeof(unsigned) + 15) / 16 /* path ids [2] */;
        static_assert(kWgSuspendAt + (size_t)kSuspendF4 * kBlockThreads == kWgQueueF4, "the suspend area closes a workgroup's slice");
        bool ok = slice_f4 <= kWgSuspendAt && max_id < (size_t)WF_INVALID && 2 * max_sidx + 1 < (size_t)WF_INVALID && (size_t)grid * block_paths <= state_entries &&
                  (size_t)kWfStatePlanes * state_entries * sizeof(float4) < ((size_t)1 << 32) && (block_paths & (block_paths - 1)) == 0 && block_paths >= 256 &&
                  block_paths <= kWgPathsMax && slice_f4 <= kWgQueueF4 && queueBuf.bytes >= (size_t)grid * kWgQueueF4 * sizeof(float4) &&
                  stateBuf.bytes >= state_bytes && w.ids == state_entries && grid >= 1 && (size_t)grid <= wg_slots &&
                  p->max_depth <= kWfDepthMax && p->n_samples <= kWfSampleMax && workPtr != nullptr;
        if (!fed && n_frames > 1) ok = ok && c->wfSeeds.bytes >= (size_t)n_frames * sizeof(float2);
        if (w.planes) ok = ok && planeBuf.bytes >= (size_t)std::max(n_planes, 1) * plane_f4 * sizeof(float4);
        if (fed) ok = ok && n_frames >= 1 && n_frames <= fed_cap && fed_cap <= kFeedMaxFrames && slot->feed_d.bytes >= sizeof(FeedDev) && w.feed_host != nullptr &&
                      slot->chunks[(n_frames - 1) / kFeedChunkFrames].p != nullptr && slot->chunk_bytes == frame_bytes;
        if (!ok) return fail(c, GLRTX_EDEVICE, "internal: wgwf launch shapes inconsistent (ids %zu, max id %zu, grid %d of %zu slots (%d per CU), block_paths %d, frames %d, fed %d)", ids, max_id,
                             grid, wg_slots, per_cu, block_paths, n_frames, (int)fed);
        (void)queue_bytes_max;
    }
    glrtx_ctx::LaunchRec *rec = nullptr;
    if ((rc = next_launch_rec(c, rec))) return rc;
    // a slot's render kernel overwrites the planes its previous plane-accumulation pass (two launches ago, on the context's stream) reads
    if (slot && slot->used) HIP_TRY(c, hipStreamWaitEvent(rstream, slot->acc_done, 0));
    // ... and a kernel on the context's path state starts behind the last one that used it (another stream's, possibly)
    if (ctx_state && c->state_used) HIP_TRY(c, hipStreamWaitEvent(rstream, c->state_done, 0));
    HIP_TRY(c, hipMemsetAsync(workPtr, 0, sizeof(unsigned), rstream));
    if (fed) {  // the device mirror starts with the frames published so far (kernel boundary: visible to every workgroup of the render kernel)
        hipLaunchKernelGGL(feed_prefill_kernel, dim3(1), dim3(64), 0, rstream, (FeedDev *)slot->feed_d.p, (const FeedHost *)slot->feed_h_dev, n_frames);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipEventRecord(rec->ev0, rstream));
    c->last_kernel = vine ? "pt_render_wgwf (list scan)" : "pt_render_wgwf";
    c->counters_stale = c->counters_stale || c->count_rays;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlockThreads), lds, rstream, a, w, workPtr, (float4 *)queueBuf.p);
    HIP_TRY(c, hipGetLastError());
#ifdef GLRTX_RAY_LOG
    g_dbg_last = {a, w, lds, grid, (float4 *)queueBuf.p, fetch};
#endif
    HIP_TRY(c, hipEventRecord(rec->evm, rstream));
    if (ctx_state) { HIP_TRY(c, hipEventRecord(c->state_done, rstream)); c->state_used = true; }
    if (slot) {  // the context's stream -- where the caller's own work, the resolve pass and the next accumulation are ordered -- takes over
        HIP_TRY(c, hipEventRecord(slot->render_done, rstream));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, slot->render_done, 0));
        HIP_TRY(c, hipEventRecord(rec->eva, c->stream));  // (the interval evm..ev1 would include the wait for the passes queued before this one)
    }
    rec->has_eva = slot != nullptr;
    {
        const dim3 g((c->width + 63) / 64, (c->owned_rows + 3) / 4);
        if (fed)
            hipLaunchKernelGGL(accumulate_feed_kernel, g, dim3(256), 0, c->stream, a.accum, a.pitch_f4, c->width, c->owned_rows, (const FeedDev *)slot->feed_d.p, p->n_samples);
        else if (w.planes && n_planes > 0)
            hipLaunchKernelGGL(accumulate_planes_kernel, g, dim3(256), 0, c->stream, a.accum, a.pitch_f4, c->width, c->owned_rows, (const float4 *)planeBuf.p, n_planes);
        HIP_TRY(c, hipGetLastError());
    }
    if (slot) { HIP_TRY(c, hipEventRecord(slot->acc_done, c->stream)); slot->used = true; }
    HIP_TRY(c, hipEventRecord(rec->ev1, c->stream));
    rec->kernel = c->last_kernel; rec->frames = n_frames;
    c->ring_head++;
    c->st.frames_last = n_frames;
    c->st.paths += (uint64_t)c->owned_rows * (uint64_t)c->width * (uint64_t)p->n_samples * (uint64_t)n_frames;
    // what the next call may build on
    c->last_render_done = slot ? slot->render_done : c->state_done;  // (a plain launch runs on the context's path state: ctx_state above)
    if (fed) {
        c->open.slot = slot; c->open.p = *p; c->open.frames = n_frames; c->open.cap = fed_cap; c->open.rec = rec; c->open.frame_bytes = frame_bytes;
        c->st.feed_launches++;
    } else c->open.slot = nullptr;
    return GLRTX_OK;
}

// Whether the wavefront variant's packed path state can represent this launch (meta = depth | sample << 8 | flags << 28).
bool wgwf_can_hold(const glrtx_params *p) { return p->max_depth <= kWfDepthMax && p->n_samples <= kWfSampleMax; }

}  // namespace

extern "C" {

int glrtx_abi_version(void) { return GLRTX_ABI_VERSION; }

const char *glrtx_last_error(const glrtx_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int glrtx_create(glrtx_ctx **out, int device_id) {
    if (!out) return fail(nullptr, GLRTX_EINVAL, "glrtx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(nullptr, GLRTX_EDEVICE, "no HIP device available (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (device_id < 0) {
        if (hipGetDevice(&device_id) != hipSuccess) device_id = 0;
    }
    if (device_id >= n) return fail(nullptr, GLRTX_EINVAL, "device %d out of range (%d devices)", device_id, n);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess)
        return fail(nullptr, GLRTX_EDEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, GLRTX_EDEVICE, "device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
    glrtx_ctx *c = new (std::nothrow) glrtx_ctx;
    if (!c) return fail(nullptr, GLRTX_ENOMEM, "out of host memory");
    c->device = device_id;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&c->tm0)) != hipSuccess || (e = hipEventCreate(&c->tm1)) != hipSuccess ||
        (e = hipEventCreate(&c->rs0)) != hipSuccess || (e = hipEventCreate(&c->rs1)) != hipSuccess ||
        (e = hipMalloc(&c->counter.p, 2 * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMemset(c->counter.p, 0, 2 * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMalloc(&c->work.p, 64)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->state_done, hipEventDisableTiming)) != hipSuccess ||
        (e = hipHostMalloc((void **)&c->guard_host, 64, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess ||
        (e = hipHostGetDevicePointer((void **)&c->guard_dev, c->guard_host, 0)) != hipSuccess) {
        fail(nullptr, GLRTX_EDEVICE, "context setup failed: %s", hipGetErrorString(e));
        glrtx_destroy(c);
        return GLRTX_EDEVICE;
    }
    for (auto &r : c->ring)
        if ((e = hipEventCreate(&r.ev0)) != hipSuccess || (e = hipEventCreate(&r.evm)) != hipSuccess || (e = hipEventCreate(&r.eva)) != hipSuccess ||
            (e = hipEventCreate(&r.ev1)) != hipSuccess) {
            fail(nullptr, GLRTX_EDEVICE, "context setup failed: %s", hipGetErrorString(e));
            glrtx_destroy(c);
            return GLRTX_EDEVICE;
        }
    std::memset(c->guard_host, 0, 64);
    c->stream = c->own_stream;
    // The slots' streams are created at a priority of their own (GLRTX_PIPE_PRIORITY: high (default) | normal | low).  The runtime maps the streams of a process onto a
    // few hardware queues PER PRIORITY LEVEL, and streams that share a queue run one after the other: at the default priority the slots share those queues with
    // whatever streams the host application creates, and two caller streams kept busy with small kernels took one launch per frame from 1.25 to 2.7 ms
    // (tests/test_gpu_parity.py::test_overlapped_launches_next_to_a_callers_own_streams).  At a priority of their own the slots are mapped among themselves.
    // NOTE for embedders: at "high" the library's render kernels are scheduled ahead of the host application's own normal-priority work on this GPU.
    int prio_least = 0, prio_greatest = 0, prio = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) { (void)hipGetLastError(); prio_least = prio_greatest = 0; }
    {
        const char *v = std::getenv("GLRTX_PIPE_PRIORITY");
        const std::string m = v ? v : "high";
        prio = m == "low" ? prio_least : (m == "normal" ? 0 : prio_greatest);
        prio = std::max(std::min(prio, prio_least), prio_greatest);  // numerically: greatest <= prio <= least
    }
    for (auto &sl : c->pipe)
        if ((e = hipStreamCreateWithPriority(&sl.stream, hipStreamNonBlocking, prio)) != hipSuccess || (e = hipEventCreateWithFlags(&sl.render_done, hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&sl.acc_done, hipEventDisableTiming)) != hipSuccess || (e = hipMalloc(&sl.work.p, 64)) != hipSuccess) {
            fail(nullptr, GLRTX_EDEVICE, "context setup failed: %s", hipGetErrorString(e));
            glrtx_destroy(c);
            return GLRTX_EDEVICE;
        }
    for (auto &sl : c->pipe) {
        if (hipHostMalloc((void **)&sl.feed_h, sizeof(FeedHost), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostGetDevicePointer((void **)&sl.feed_h_dev, sl.feed_h, 0) != hipSuccess) {
            (void)hipGetLastError();
            c->feed_ok = false;  // no host-coherent memory: launches are never fed (they still overlap as in round 5)
            break;
        }
        std::memset(sl.feed_h, 0, sizeof(FeedHost));
    }
    c->pipeline = std::getenv("GLRTX_NO_PIPELINE") == nullptr;
    if (const char *v = std::getenv("GLRTX_PIPE_SHARE")) c->pipe_share = std::max(1, std::min(4, std::atoi(v)));
    if (const char *v = std::getenv("GLRTX_PIPE_SLOTS")) c->pipe_slots = (unsigned)std::max(1, std::min((int)kPipeSlots, std::atoi(v)));
    c->n_cu = prop.multiProcessorCount;
    if (const char *v = std::getenv("GLRTX_VARIANT")) { const int x = std::atoi(v); if (x >= 0 && x <= 2) c->variant = x; }
    if (const char *v = std::getenv("GLRTX_SHADOW_LIMIT")) c->sc.shadow_limited = std::atoi(v) != 0 ? 1 : 0;  // default 0: the reference's own search (pt_kernel.hip.h: shadow_limit)
    *out = c;
    return GLRTX_OK;
}

void glrtx_destroy(glrtx_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    for (auto &sl : c->pipe) {
        if (sl.stream) { (void)hipStreamSynchronize(sl.stream); (void)hipStreamDestroy(sl.stream); }
        if (sl.render_done) (void)hipEventDestroy(sl.render_done);
        if (sl.acc_done) (void)hipEventDestroy(sl.acc_done);
        dev_free(sl.state); dev_free(sl.queues); dev_free(sl.planes); dev_free(sl.work); dev_free(sl.feed_d);
        for (auto &ch : sl.chunks) dev_free(ch);
        if (sl.feed_h) (void)hipHostFree(sl.feed_h);
    }
    dev_free(c->spheres); dev_free(c->sphereMat); dev_free(c->forks); dev_free(c->nrms); dev_free(c->mats); dev_free(c->lights); dev_free(c->vine);
    dev_free(c->accum_own); dev_free(c->counter); dev_free(c->rgba8); dev_free(c->work);
    dev_free(c->wfState); dev_free(c->wfQ); dev_free(c->wfSeeds); dev_free(c->wfPlanes);
    dev_free(c->bvhVert); dev_free(c->bvhTri); dev_free(c->bvhNodes);
    if (c->bvhWs.p) { (void)hipFree(c->bvhWs.p); c->bvhWs.p = nullptr; c->bvhWs.bytes = 0; }
    for (auto &r : c->ring) {
        if (r.ev0) (void)hipEventDestroy(r.ev0);
        if (r.evm) (void)hipEventDestroy(r.evm);
        if (r.eva) (void)hipEventDestroy(r.eva);
        if (r.ev1) (void)hipEventDestroy(r.ev1);
    }
    if (c->tm0) (void)hipEventDestroy(c->tm0);
    if (c->tm1) (void)hipEventDestroy(c->tm1);
    if (c->rs0) (void)hipEventDestroy(c->rs0);
    if (c->rs1) (void)hipEventDestroy(c->rs1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->guard_host) (void)hipHostFree(c->guard_host);
    if (c->state_done) (void)hipEventDestroy(c->state_done);
    delete c;
}

int glrtx_upload_scene(glrtx_ctx *c, const float *vert, size_t n_vert, const float *tri, size_t n_tri, const float *mat,
                       size_t n_mat, const float *light, size_t n_light, const float *bvh, size_t n_nodes) {
    if (!c) return GLRTX_EINVAL;
    seal_feed(c);  // (nothing is appended to an open launch across this call: glrtx_ctx::OpenFeed)
    Packed P;
    if (int prc = pack_scene(c, nullptr, P, vert, n_vert, tri, n_tri, mat, n_mat, light, n_light, bvh, n_nodes)) return prc;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<float4> &forks = P.forks, &tris = P.tris, &nrms = P.nrms, &mats = P.mats, &lights = P.lights;
    const int root_ref = P.root_ref, stack_need = P.stack_need;

    int rc;
    // one node array: the leaf records in reverse order of their ids (id k at record index ~k; id 0, directly in front of fork 0, is the
    // all-zero record an absent child refers to: never hit), then the forks (DevScene::forks points at fork 0)
    const size_t tri_f4 = tris.size(), n_ids = tris.size() / 4;
    if ((tri_f4 + forks.size()) * sizeof(float4) >= ((size_t)1 << 32)) return fail(c, GLRTX_EINVAL, "glrtx_upload_scene: node array exceeds 4 GiB");
    std::vector<float4> nodes(tri_f4 + forks.size(), make_float4(0.f, 0.f, 0.f, 0.f));
    for (size_t id = 0; id < n_ids; id++) std::memcpy(&nodes[4 * (n_ids - 1 - id)], &tris[4 * id], 4 * sizeof(float4));
    std::memcpy(nodes.data() + tri_f4, forks.data(), forks.size() * sizeof(float4));
    if ((rc = dev_upload(c, c->forks, nodes.data(), nodes.size() * sizeof(float4)))) return rc;
    if ((rc = dev_upload(c, c->nrms, nrms.data(), nrms.size() * sizeof(float4)))) return rc;
    if ((rc = dev_upload(c, c->mats, mats.data(), mats.size() * sizeof(float4)))) return rc;
    if ((rc = dev_upload(c, c->lights, lights.data(), lights.size() * sizeof(float4)))) return rc;
    if (!P.vine.empty() && (rc = dev_upload(c, c->vine, P.vine.data(), P.vine.size() * sizeof(float4)))) return rc;

    DevScene &sc = c->sc;
    sc.forks = (const float4 *)c->forks.p + tri_f4;
    sc.nodes0 = (const float4 *)c->forks.p;
    sc.node_bias = (unsigned)(tri_f4 * sizeof(float4));
    sc.nrms = (const float4 *)c->nrms.p;
    sc.mats = (const float4 *)c->mats.p;
    sc.lights = (const float4 *)c->lights.p;
    sc.root_ref = root_ref;
    sc.root_boxed = P.root_boxed;
    sc.root_lo = P.root_lo; sc.root_hi = P.root_hi;
    sc.n_light = (int)n_light;
    sc.n_mat = (int)n_mat;
    sc.n_fork = (int)(forks.size() / 4);
    sc.stack_entries = stack_need;
    sc.mats_in_lds = (n_mat > 0 && n_mat <= (size_t)kMaxLdsMaterials) ? 1 : 0;
    sc.lights_in_lds = (n_light > 0 && n_light <= (size_t)kMaxLdsLights && std::getenv("GLRTX_NO_LDS_LIGHTS") == nullptr) ? 1 : 0;
    sc.lds_head_f4 = (sc.mats_in_lds ? 3 * (int)n_mat : 0) + (sc.lights_in_lds ? 6 * (int)n_light : 0);
    sc.vine = P.vine.empty() ? nullptr : (const float4 *)c->vine.p;
    sc.n_vine = P.n_vine;
    sc.vine_main = P.vine_main;
    sc.vine_uniform = P.vine_uniform;
    if (std::getenv("GLRTX_NO_VINE_SCAN")) sc.n_vine = 0;  // A/B: force the generic tree traversal
    c->n_tri = (int)n_tri; c->n_fork = (int)(forks.size() / 4); c->n_mat = (int)n_mat; c->n_light = (int)n_light;
    c->have_scene = true;
    c->leaf_tri.swap(P.leaf_tri);
    c->n_spheres = 0;  // spheres reference this scene's materials: upload them again after a new scene
    c->st.stack_entries = stack_need;
    c->st.lds_bytes = lds_bytes_for(sc);
    c->st.n_tri = c->n_tri; c->st.n_fork = c->n_fork; c->st.n_mat = c->n_mat; c->st.n_light = c->n_light;
    return GLRTX_OK;
}

// Host-only validation of a wire-format scene (no device needed): the same checks and repacking
// glrtx_upload_scene performs.  Outputs may be NULL.
int glrtx_check_scene(const float *vert, size_t n_vert, const float *tri, size_t n_tri, const float *mat, size_t n_mat,
                      const float *light, size_t n_light, const float *bvh, size_t n_nodes, int *n_fork_out,
                      int *stack_entries_out) {
    Packed P;
    if (int rc = pack_scene(nullptr, &g_create_error, P, vert, n_vert, tri, n_tri, mat, n_mat, light, n_light, bvh, n_nodes)) return rc;
    if (n_fork_out) *n_fork_out = (int)(P.forks.size() / 4);
    if (stack_entries_out) *stack_entries_out = P.stack_need;
    return GLRTX_OK;
}

// Host-only: the fork records as the device will see them (16 floats each: {minL, refL} {maxL, refR} {minR, -} {maxR, -}; refs as
// int bit patterns: a ref < 0 is the record ~id of a leaf, ids from 1 in traversal order, ref -1 = id 0 the never-hit record an absent child refers to;
// leaf pairs have no fork record -- include/glrtx.h), for tests that replay
// trav_step's push / pop rules on the packed tree and compare the deepest stack they reach with stack_entries.
int glrtx_debug_pack_forks(const float *vert, size_t n_vert, const float *tri, size_t n_tri, const float *mat, size_t n_mat, const float *light,
                           size_t n_light, const float *bvh, size_t n_nodes, float *forks_out, size_t capacity_forks, int *n_fork_out,
                           int *root_ref_out, int *stack_entries_out) {
    Packed P;
    if (int rc = pack_scene(nullptr, &g_create_error, P, vert, n_vert, tri, n_tri, mat, n_mat, light, n_light, bvh, n_nodes)) return rc;
    const size_t nf = P.forks.size() / 4;
    if (n_fork_out) *n_fork_out = (int)nf;
    if (root_ref_out) *root_ref_out = P.root_ref;
    if (stack_entries_out) *stack_entries_out = P.stack_need;
    if (forks_out) {
        if (capacity_forks < nf) return pfail(nullptr, &g_create_error, GLRTX_EINVAL, "glrtx_debug_pack_forks: %zu forks, room for %zu", nf, capacity_forks);
        std::memcpy(forks_out, P.forks.data(), nf * 16 * sizeof(float));
    }
    return GLRTX_OK;
}

int glrtx_build_lbvh(glrtx_ctx *c, const float *vert, size_t n_vert, const float *tri, size_t n_tri, float *nodes_out, int *max_depth_out,
                     float *build_ms_out) {
    if (!c) return GLRTX_EINVAL;
    seal_feed(c);  // (nothing is appended to an open launch across this call: glrtx_ctx::OpenFeed)
    if (!vert || !tri || !nodes_out || n_tri == 0 || n_vert == 0) return fail(c, GLRTX_EINVAL, "glrtx_build_lbvh: empty input");
    if (2 * n_tri - 1 > ((size_t)1 << 24) || n_vert > (size_t)INT32_MAX / 16)
        return fail(c, GLRTX_EINVAL, "glrtx_build_lbvh: %zu triangles: node indices must fit a float (2^24)", n_tri);
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n_nodes = 2 * n_tri - 1;
    int rc;
    if ((rc = dev_upload(c, c->bvhVert, vert, n_vert * 15 * sizeof(float)))) return rc;
    if ((rc = dev_upload(c, c->bvhTri, tri, n_tri * 4 * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->bvhNodes, n_nodes * 9 * sizeof(float)))) return rc;
    int depth = 0, bad = 0;
    HIP_TRY(c, hipEventRecord(c->tm0, c->stream));
    HIP_TRY(c, lbvh::build(c->stream, (const float *)c->bvhVert.p, (unsigned)n_vert, (const float *)c->bvhTri.p, (unsigned)n_tri,
                           (float *)c->bvhNodes.p, c->bvhWs, &depth, &bad));
    HIP_TRY(c, hipEventRecord(c->tm1, c->stream));
    HIP_TRY(c, hipEventSynchronize(c->tm1));
    if (bad) return fail(c, GLRTX_ESCENE, "glrtx_build_lbvh: a triangle references a vertex out of range");
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->tm0, c->tm1));
    HIP_TRY(c, hipMemcpy(nodes_out, c->bvhNodes.p, n_nodes * 9 * sizeof(float), hipMemcpyDeviceToHost));
    if (max_depth_out) *max_depth_out = depth;
    if (build_ms_out) *build_ms_out = ms;
    return depth < 63 ? GLRTX_OK : fail(c, GLRTX_EDEPTH, "glrtx_build_lbvh: tree depth %d exceeds the 64-entry traversal stack", depth);
}

int glrtx_build_bvh_sah(glrtx_ctx *c, const float *vert, size_t n_vert, const float *tri, size_t n_tri, float *nodes_out, int *max_depth_out,
                     float *build_ms_out) {
    if (!c) return GLRTX_EINVAL;
    seal_feed(c);  // (nothing is appended to an open launch across this call: glrtx_ctx::OpenFeed)
    if (!vert || !tri || !nodes_out || n_tri == 0 || n_vert == 0) return fail(c, GLRTX_EINVAL, "glrtx_build_bvh_sah: empty input");
    if (2 * n_tri - 1 > ((size_t)1 << 24) || n_vert > (size_t)INT32_MAX / 16)
        return fail(c, GLRTX_EINVAL, "glrtx_build_bvh_sah: %zu triangles: node indices must fit a float (2^24)", n_tri);
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n_nodes = 2 * n_tri - 1;
    int rc;
    if ((rc = dev_upload(c, c->bvhVert, vert, n_vert * 15 * sizeof(float)))) return rc;
    if ((rc = dev_upload(c, c->bvhTri, tri, n_tri * 4 * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->bvhNodes, n_nodes * 9 * sizeof(float)))) return rc;
    int depth = 0, bad = 0;
    HIP_TRY(c, hipEventRecord(c->tm0, c->stream));
    HIP_TRY(c, sahl::build(c->stream, (const float *)c->bvhVert.p, (unsigned)n_vert, (const float *)c->bvhTri.p, (unsigned)n_tri,
                           (float *)c->bvhNodes.p, c->bvhWs, &depth, &bad, nullptr));
    HIP_TRY(c, hipEventRecord(c->tm1, c->stream));
    HIP_TRY(c, hipEventSynchronize(c->tm1));
    if (bad) return fail(c, GLRTX_ESCENE, "glrtx_build_bvh_sah: a triangle references a vertex out of range");
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->tm0, c->tm1));
    HIP_TRY(c, hipMemcpy(nodes_out, c->bvhNodes.p, n_nodes * 9 * sizeof(float), hipMemcpyDeviceToHost));
    if (max_depth_out) *max_depth_out = depth;
    if (build_ms_out) *build_ms_out = ms;
    return depth < 63 ? GLRTX_OK : fail(c, GLRTX_EDEPTH, "glrtx_build_bvh_sah: tree depth %d exceeds the 64-entry traversal stack", depth);
}

int glrtx_upload_spheres(glrtx_ctx *c, const float *spheres, size_t n_spheres) {
    if (!c) return GLRTX_EINVAL;
    seal_feed(c);  // (nothing is appended to an open launch across this call: glrtx_ctx::OpenFeed)
    if (n_spheres && !spheres) return fail(c, GLRTX_EINVAL, "glrtx_upload_spheres: NULL buffer with non-zero count");
    if (n_spheres > (size_t)kMaxSpheres) return fail(c, GLRTX_EINVAL, "glrtx_upload_spheres: at most %d spheres (they are tested one by one)", kMaxSpheres);
    if (!c->have_scene) return fail(c, GLRTX_EINVAL, "glrtx_upload_spheres: upload the scene (materials) first");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<float4> sp(n_spheres);
    std::vector<int> mt(n_spheres);
    for (size_t k = 0; k < n_spheres; k++) {
        const float *r = spheres + 5 * k;
        if (!(r[3] > 0.0f)) return fail(c, GLRTX_ESCENE, "sphere %zu: radius %g is not positive", k, r[3]);
        if (!(r[4] >= 0.0f) || (size_t)r[4] >= (size_t)c->n_mat) return fail(c, GLRTX_ESCENE, "sphere %zu: material %g out of range", k, r[4]);
        sp[k] = make_float4(r[0], r[1], r[2], r[3]);
        mt[k] = (int)r[4];
    }
    int rc;
    if ((rc = dev_upload(c, c->spheres, sp.data(), n_spheres * sizeof(float4)))) return rc;
    if ((rc = dev_upload(c, c->sphereMat, mt.data(), n_spheres * sizeof(int)))) return rc;
    c->n_spheres = (int)n_spheres;
    return GLRTX_OK;
}

int glrtx_set_extensions(glrtx_ctx *c, int flags) {
    if (!c) return GLRTX_EINVAL;
    seal_feed(c);  // (nothing is appended to an open launch across this call: glrtx_ctx::OpenFeed)
    if (flags & ~(GLRTX_EXT_DIELECTRIC | GLRTX_EXT_WHITTED)) return fail(c, GLRTX_EINVAL, "glrtx_set_extensions: unknown flag bits 0x%x", flags);
    static_assert(GLRTX_EXT_DIELECTRIC == EXT_DIELECTRIC && GLRTX_EXT_WHITTED == EXT_WHITTED, "extension flag values");
    c->ext_flags = flags;
    return GLRTX_OK;
}

int glrtx_set_partition(glrtx_ctx *c, int rank, int world, int stripe_rows) {
    if (!c) return GLRTX_EINVAL;
    seal_feed(c);  // (nothing is appended to an open launch across this call: glrtx_ctx::OpenFeed)
    if (world < 1 || rank < 0 || rank >= world || stripe_rows < 1)
        return fail(c, GLRTX_EINVAL, "glrtx_set_partition: bad rank/world/stripe %d/%d/%d", rank, world, stripe_rows);
    if (stripe_rows % kStripeQuantum != 0)  // whole 8x8 work tiles per stripe (pixel -> row mapping itself works for any stripe height)
        return fail(c, GLRTX_EINVAL, "glrtx_set_partition: stripe_rows must be a multiple of %d", kStripeQuantum);
    if (c->bound && c->width > 0 && owned_rows_of(c->height, rank, world, stripe_rows) > c->bound_rows)
        return fail(c, GLRTX_EINVAL, "glrtx_set_partition: the new partition does not fit the bound accumulator (%d rows); unbind it first", c->bound_rows);
    c->rank = rank; c->world = world; c->stripe = stripe_rows;
    if (c->width > 0) return glrtx_resize(c, c->width, c->height);
    return GLRTX_OK;
}

int glrtx_local_row_to_y(const glrtx_ctx *c, int r) {
    if (!c || r < 0 || r >= c->owned_rows) return -1;
    return ((r / c->stripe) * c->world + c->rank) * c->stripe + r % c->stripe;
}

int glrtx_resize(glrtx_ctx *c, int width, int height) {
    if (!c) return GLRTX_EINVAL;
    seal_feed(c);  // (nothing is appended to an open launch across this call: glrtx_ctx::OpenFeed)
    if (width < 1 || height < 1 || width > 65536 || height > 65536) return fail(c, GLRTX_EINVAL, "glrtx_resize: bad size %dx%d", width, height);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->bound) {  // a caller-owned accumulator is bound: the new shape must fit it (nothing here can grow it)
        const int rows = owned_rows_of(height, c->rank, c->world, c->stripe);
        if ((size_t)width * sizeof(float4) > c->pitch_bytes || rows > c->bound_rows)
            return fail(c, GLRTX_EINVAL, "glrtx_resize: %dx%d (%d owned rows) does not fit the bound accumulator (%zu-byte rows, %d rows); unbind it first",
                        width, height, rows, c->pitch_bytes, c->bound_rows);
    }
    c->width = width; c->height = height;
    c->owned_rows = owned_rows_of(height, c->rank, c->world, c->stripe);
    c->st.width = width; c->st.height = height; c->st.owned_rows = c->owned_rows;
    if (!c->bound) {
        // pitch: rows padded to 256 B so every 8-pixel tile row is one aligned 128 B segment
        const size_t pitch = ((size_t)width * sizeof(float4) + 255) / 256 * 256;
        const size_t bytes = pitch * (size_t)std::max(c->owned_rows, 1);
        if (c->accum_own.bytes < bytes) {
            dev_free(c->accum_own);
            HIP_TRY(c, hipMalloc(&c->accum_own.p, bytes));
            c->accum_own.bytes = bytes;
        }
        c->accum = (float4 *)c->accum_own.p;
        c->pitch_bytes = pitch;
    }
    return glrtx_clear(c);
}

int glrtx_clear(glrtx_ctx *c) {
    if (!c) return GLRTX_EINVAL;
    seal_feed(c);  // (nothing is appended to an open launch across this call: glrtx_ctx::OpenFeed)
    if (!c->accum) return fail(c, GLRTX_EINVAL, "glrtx_clear: no accumulator (call glrtx_resize first)");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemsetAsync(c->accum, 0, c->pitch_bytes * (size_t)c->owned_rows, c->stream));
    return GLRTX_OK;
}

int glrtx_bind_accum(glrtx_ctx *c, void *device_ptr, size_t pitch_bytes, int capacity_rows) {
    if (!c) return GLRTX_EINVAL;
    seal_feed(c);  // (nothing is appended to an open launch across this call: glrtx_ctx::OpenFeed)
    if (c->width < 1) return fail(c, GLRTX_EINVAL, "glrtx_bind_accum: call glrtx_resize first");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!device_ptr) {
        c->bound = false;
        c->bound_rows = 0;
        return glrtx_resize(c, c->width, c->height);
    }
    if (pitch_bytes < (size_t)c->width * sizeof(float4) || pitch_bytes % sizeof(float4) != 0 || ((uintptr_t)device_ptr & 15) != 0)
        return fail(c, GLRTX_EINVAL, "glrtx_bind_accum: pitch/alignment invalid");
    if (capacity_rows < c->owned_rows)
        return fail(c, GLRTX_EINVAL, "glrtx_bind_accum: the buffer holds %d rows, this partition owns %d", capacity_rows, c->owned_rows);
    c->bound = true;
    c->bound_rows = capacity_rows;
    c->accum = (float4 *)device_ptr;
    c->pitch_bytes = pitch_bytes;
    return GLRTX_OK;
}

int glrtx_set_stream(glrtx_ctx *c, void *hip_stream) {
    if (!c) return GLRTX_EINVAL;
    seal_feed(c);  // (nothing is appended to an open launch across this call: glrtx_ctx::OpenFeed)
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (int rc = fold_launches(c, true)) return rc;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return GLRTX_OK;
}

int glrtx_set_variant(glrtx_ctx *c, int variant) {
    if (!c) return GLRTX_EINVAL;
    seal_feed(c);  // (nothing is appended to an open launch across this call: glrtx_ctx::OpenFeed)
    if (variant < 0 || variant > 2) return fail(c, GLRTX_EINVAL, "glrtx_set_variant: unknown variant %d", variant);
    c->variant = variant;
    return GLRTX_OK;
}

int glrtx_set_shadow_range_limit(glrtx_ctx *c, int enable) {
    if (!c) return GLRTX_EINVAL;
    seal_feed(c);  // (nothing is appended to an open launch across this call: glrtx_ctx::OpenFeed)
    c->sc.shadow_limited = enable != 0 ? 1 : 0;  // read by the next launch (the scene block is copied into every launch's arguments)
    return GLRTX_OK;
}

int glrtx_count_rays(glrtx_ctx *c, int enable) {
    if (!c) return GLRTX_EINVAL;
    seal_feed(c);  // (nothing is appended to an open launch across this call: glrtx_ctx::OpenFeed)
    c->count_rays = enable != 0;
    return GLRTX_OK;
}

int glrtx_render_frames(glrtx_ctx *c, const glrtx_params *p, const float *seeds_xy, int n_frames) {
    if (!c || !p) return GLRTX_EINVAL;
    if (n_frames < 0 || (n_frames > 0 && !seeds_xy)) return fail(c, GLRTX_EINVAL, "glrtx_render_frames: bad seeds/n_frames");
    if (n_frames == 0) return GLRTX_OK;
    if (n_frames == 1 || c->variant != 2 || !wgwf_can_hold(p) || c->n_spheres > 0 || c->ext_flags != 0) {  // the megakernels have no frames-in-flight form: one launch per frame
        for (int f = 0; f < n_frames; f++) {
            glrtx_params q = *p;
            q.seed[0] = seeds_xy[2 * f]; q.seed[1] = seeds_xy[2 * f + 1];
            if (int rc = glrtx_render(c, &q)) return rc;
        }
        return GLRTX_OK;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    // A launch of the same camera that is still open takes the frames itself (feed_append); otherwise they are issued as launches of at most `most` frames -- what the
    // frames-in-flight memory budget allows (frames_cap) -- of equal size, in order; each of them stays open for the next call.  An allocation that fails is reported,
    // not worked around.
    if (p->n_samples >= 0 && p->max_depth >= 0 && c->have_scene && c->accum && c->owned_rows > 0 && feed_append(c, p, seeds_xy, n_frames) == 1) {
        c->last_was_render = true; c->last_p = *p;
        return GLRTX_OK;
    }
    const bool may_feed = c->feed_ok && std::getenv("GLRTX_NO_FEED") == nullptr && c->pipeline && c->stream == c->own_stream;
    int most = std::min(n_frames, may_feed ? std::min(frames_cap(c, p, (int)kFedSlots), kFeedMaxFrames) : frames_cap(c, p, 1));
    if (const char *v = std::getenv("GLRTX_FEED_CAP")) { if (may_feed) most = std::max(1, std::min(most, std::atoi(v))); }
    const int n_launches = (n_frames + most - 1) / most;
    const int chunk = (n_frames + n_launches - 1) / n_launches;  // equal helpings: 16 frames under a limit of 14 are 8 + 8, not 14 + 2
    for (int f0 = 0; f0 < n_frames; f0 += chunk) {
        const int n = std::min(chunk, n_frames - f0);
        int rc;
        if (n == 1) {
            glrtx_params q = *p;
            q.seed[0] = seeds_xy[2 * f0]; q.seed[1] = seeds_xy[2 * f0 + 1];
            rc = glrtx_render(c, &q);
        } else {
            c->frames_seeds = seeds_xy + 2 * (size_t)f0;
            c->frames_n = n;
            rc = glrtx_render(c, p);
            c->frames_seeds = nullptr;
            c->frames_n = 1;
            if (rc == GLRTX_OK) c->st.launches += (uint64_t)(n - 1);
        }
        if (rc != GLRTX_OK) return rc;
    }
    return GLRTX_OK;
}

int glrtx_render(glrtx_ctx *c, const glrtx_params *p) {
    if (!c || !p) return GLRTX_EINVAL;
    if (!c->have_scene) return fail(c, GLRTX_EINVAL, "glrtx_render: no scene uploaded");
    if (!c->accum || c->width < 1) return fail(c, GLRTX_EINVAL, "glrtx_render: no accumulator (call glrtx_resize)");
    if (p->n_samples < 0 || p->max_depth < 0) return fail(c, GLRTX_EINVAL, "glrtx_render: negative n_samples/max_depth");
    HIP_TRY(c, hipSetDevice(c->device));
    const bool wavefront = c->variant == 2 && wgwf_can_hold(p) && c->n_spheres == 0 && c->ext_flags == 0;
    if (c->owned_rows > 0 && wavefront && c->frames_n == 1 && feed_append(c, p, p->seed, 1) == 1) {  // a launch of the same camera that is still open takes the frame itself
        c->last_was_render = true; c->last_p = *p;
        return GLRTX_OK;
    }
    if (!wavefront) seal_feed(c);
    c->st.launches++;
    if (c->owned_rows == 0) return GLRTX_OK;

    KernelArgs a;
    a.sc = c->sc;
    static_assert(sizeof a.cam == kCamFloats * sizeof(float), "camera block layout");
    std::memcpy(a.cam, p->c2w, 16 * sizeof(float));
    std::memcpy(a.cam + 16, p->s2c, 16 * sizeof(float));
    a.cam[32] = p->aperture; a.cam[33] = p->focal;
    a.seed_x = p->seed[0]; a.seed_y = p->seed[1];
    a.n_samples = p->n_samples; a.max_depth = p->max_depth;
    a.width = c->width; a.height = c->height;
    a.owned_rows = c->owned_rows;
    a.rank = c->rank; a.world = c->world; a.stripe = c->stripe;
    a.accum = c->accum;
    a.pitch_f4 = (int)(c->pitch_bytes / sizeof(float4));
    a.ray_counter = (unsigned long long *)c->counter.p;
    a.hit_hist = c->hit_hist_dev;  // (null outside glrtx_hit_histogram)
    a.tiles_x = (c->width + kTile - 1) / kTile;
    const int tiles_y = (c->owned_rows + kTile - 1) / kTile;
    a.n_tiles = a.tiles_x * tiles_y;

    // Host-side shape checks before launching a hand-written kernel.
    const int lds = lds_bytes_for(c->sc);
    if (lds > 160 * 1024) return fail(c, GLRTX_EDEVICE, "render kernel needs %d B of LDS (> 160 KiB)", lds);
    if (a.pitch_f4 < c->width) return fail(c, GLRTX_EINVAL, "accumulator pitch smaller than a row");

    if ((size_t)a.pitch_f4 * (size_t)c->owned_rows >= (size_t)INT32_MAX)
        return fail(c, GLRTX_EINVAL, "accumulator too large for 32-bit pixel offsets");

    // The wavefront variant packs depth and sample index into one word of the path state (kWfDepthMax, kWfSampleMax);
    // a launch beyond those ranges runs on the persistent megakernel instead (bit-identical, no packed state).
    // Extensions (analytic spheres, dielectric, Whitted termination) exist only in the persistent megakernel's EXT instantiation.
    const bool ext = c->n_spheres > 0 || c->ext_flags != 0;
    const int variant = (ext || (c->variant == 2 && !wgwf_can_hold(p))) ? 1 : c->variant;
    c->st.variant_last = variant;
    c->st.fallback_last = 0;
    if (c->variant == 2 && variant != 2) {  // not silently: the reason and a count are in the stats
        c->st.fallback_last = (ext ? GLRTX_FALLBACK_EXTENSIONS : 0) | (p->max_depth > kWfDepthMax ? GLRTX_FALLBACK_DEPTH : 0) |
                              (p->n_samples > kWfSampleMax ? GLRTX_FALLBACK_SAMPLES : 0);
        c->st.fallback_launches++;
    }
    if (variant == 2) {
        const int rc = launch_wgwf(c, a, p, c->frames_n == 1 ? p->seed : c->frames_seeds, c->frames_n);
        if (rc == GLRTX_OK) { c->last_was_render = true; c->last_p = *p; } else seal_feed(c);
        return rc;
    }
    glrtx_ctx::LaunchRec *rec = nullptr;
    if (int rc = next_launch_rec(c, rec)) return rc;
    if (variant == 1) {
        // persistent kernel: grid = what is resident at once (occupancy x CUs), capped by the work available
        using PKernel = void (*)(const KernelArgs, unsigned *, const ExtArgs);
        const PKernel pk = ext ? (c->count_rays ? (PKernel)pt_render_persistent<true, true> : (PKernel)pt_render_persistent<false, true>)
                               : (c->count_rays ? (PKernel)pt_render_persistent<true, false> : (PKernel)pt_render_persistent<false, false>);
        ExtArgs ex{};
        ex.spheres = (const float4 *)c->spheres.p; ex.sphere_mat = (const int *)c->sphereMat.p;
        ex.n_spheres = c->n_spheres; ex.flags = c->ext_flags;
        const int lds_p = lds + (ext ? c->n_spheres * (int)sizeof(float4) : 0);  // extension kernel: the spheres are staged behind the stacks
        if (lds_p > 160 * 1024) return fail(c, GLRTX_EDEVICE, "render kernel needs %d B of LDS (> 160 KiB)", lds_p);
        if (lds_p > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void *)pk, hipFuncAttributeMaxDynamicSharedMemorySize, lds_p));
        int per_cu = 0;
        HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pk, kBlockThreads, lds_p));
        if (per_cu < 1) per_cu = 1;
        const int tiles8 = ((c->width + 7) / 8) * ((c->owned_rows + 7) / 8);
        const int n_chunks = (tiles8 * 64 + kChunk - 1) / kChunk;
        const int waves_per_wg = kBlockThreads / 64;
        int grid = std::min(per_cu * c->n_cu, (n_chunks + waves_per_wg - 1) / waves_per_wg);
        if (grid < 1) grid = 1;
        HIP_TRY(c, hipMemsetAsync(c->work.p, 0, sizeof(unsigned), c->stream));
        HIP_TRY(c, hipEventRecord(rec->ev0, c->stream));
        c->last_kernel = ext ? "pt_render_persistent (extensions)" : "pt_render_persistent";
        hipLaunchKernelGGL(pk, dim3(grid), dim3(kBlockThreads), lds_p, c->stream, a, (unsigned *)c->work.p, ex);
    } else {
    HIP_TRY(c, hipEventRecord(rec->ev0, c->stream));
    c->last_kernel = "pt_render_kernel";
    if (c->count_rays) {
        if (lds > 64 * 1024)
            HIP_TRY(c, hipFuncSetAttribute((const void *)pt_render_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(pt_render_kernel<true>, dim3(a.n_tiles), dim3(kBlockThreads), lds, c->stream, a);
    } else {
        if (lds > 64 * 1024)
            HIP_TRY(c, hipFuncSetAttribute((const void *)pt_render_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(pt_render_kernel<false>, dim3(a.n_tiles), dim3(kBlockThreads), lds, c->stream, a);
    }
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(rec->evm, c->stream));
    HIP_TRY(c, hipEventRecord(rec->ev1, c->stream));
    rec->kernel = c->last_kernel; rec->frames = 1; rec->has_eva = false;
    c->ring_head++;
    c->counters_stale = c->counters_stale || c->count_rays;
    c->st.frames_last = 1;
    c->st.paths += (uint64_t)c->owned_rows * (uint64_t)c->width * (uint64_t)p->n_samples;
    return GLRTX_OK;
}

int glrtx_sync(glrtx_ctx *c) {
    if (!c) return GLRTX_EINVAL;
    seal_feed(c);  // (nothing is appended to an open launch across this call: glrtx_ctx::OpenFeed)
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->counters_stale && c->counter.p &&  // the device is idle here: bring the ray counters over now, so that glrtx_get_stats after a sync touches nothing
        hipMemcpy(c->counters_host, c->counter.p, sizeof c->counters_host, hipMemcpyDeviceToHost) == hipSuccess)
        c->counters_stale = false;
    return fold_launches(c, true);
}

int glrtx_read_accum(glrtx_ctx *c, float *dst, size_t dst_pitch_bytes) {
    if (!c || !dst) return GLRTX_EINVAL;
    seal_feed(c);  // (nothing is appended to an open launch across this call: glrtx_ctx::OpenFeed)
    if (!c->accum) return fail(c, GLRTX_EINVAL, "glrtx_read_accum: no accumulator");
    const size_t row = (size_t)c->width * sizeof(float4);
    if (dst_pitch_bytes < row) return fail(c, GLRTX_EINVAL, "glrtx_read_accum: dst pitch too small");
    if (int rc = glrtx_sync(c)) return rc;
    if (c->owned_rows == 0) return GLRTX_OK;
    HIP_TRY(c, hipMemcpy2D(dst, dst_pitch_bytes, c->accum, c->pitch_bytes, row, (size_t)c->owned_rows, hipMemcpyDeviceToHost));
    return GLRTX_OK;
}

int glrtx_accum_device_ptr(const glrtx_ctx *c, void **ptr_out, size_t *pitch_out) {
    if (!c || !ptr_out || !pitch_out) return GLRTX_EINVAL;
    seal_feed(const_cast<glrtx_ctx *>(c));  // (the caller is about to look at the accumulator: nothing more goes into a launch that is already queued in front of that)
    *ptr_out = c->accum;
    *pitch_out = c->pitch_bytes;
    return GLRTX_OK;
}

int glrtx_resolve_rgba8(glrtx_ctx *c, uint8_t *dst, size_t dst_pitch_bytes, float gamma, int flip_y) {
    if (!c || !dst) return GLRTX_EINVAL;
    seal_feed(c);  // (nothing is appended to an open launch across this call: glrtx_ctx::OpenFeed)
    if (!c->accum) return fail(c, GLRTX_EINVAL, "glrtx_resolve_rgba8: no accumulator");
    if (!(gamma > 0.f)) return fail(c, GLRTX_EINVAL, "glrtx_resolve_rgba8: gamma must be positive");
    if (dst_pitch_bytes < (size_t)c->width * 4) return fail(c, GLRTX_EINVAL, "glrtx_resolve_rgba8: dst pitch too small");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->owned_rows == 0) return glrtx_sync(c);
    const size_t bytes = (size_t)c->width * 4 * (size_t)c->owned_rows;
    if (c->rgba8.bytes < bytes) {
        dev_free(c->rgba8);
        HIP_TRY(c, hipMalloc(&c->rgba8.p, bytes));
        c->rgba8.bytes = bytes;
    }
    int per = kResolvePer;
    if (const char *v = std::getenv("GLRTX_RESOLVE_PER")) per = std::atoi(v);  // (A/B: pixels per lane)
    per = per == 1 || per == 4 ? per : 2;
    const dim3 grid = resolve_grid(c->width, c->owned_rows, per);
    HIP_TRY(c, hipEventRecord(c->rs0, c->stream));
    const auto rk = per == 1 ? resolve_kernel<1> : (per == 4 ? resolve_kernel<4> : resolve_kernel<2>);
    hipLaunchKernelGGL(rk, grid, dim3(256), 0, c->stream, (const float4 *)c->accum, (int)(c->pitch_bytes / sizeof(float4)),
                       c->width, c->owned_rows, (uchar4 *)c->rgba8.p, c->width, 1.0f / gamma, flip_y ? 1 : 0);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->rs1, c->stream));
    if (int rc = glrtx_sync(c)) return rc;
    HIP_TRY(c, hipEventElapsedTime(&c->st.resolve_ms_last, c->rs0, c->rs1));
    HIP_TRY(c, hipMemcpy2D(dst, dst_pitch_bytes, c->rgba8.p, (size_t)c->width * 4, (size_t)c->width * 4, (size_t)c->owned_rows,
                           hipMemcpyDeviceToHost));
    return GLRTX_OK;
}

// Device time of the resolve kernel by itself: `reps` launches back to back between one pair of events, per launch.  A single launch between two events (what
// glrtx_stats.resolve_ms_last reports) carries the command processor's latency on both sides -- ~6 us of a 15-us measurement at 1080p -- which says nothing about the kernel.
int glrtx_debug_resolve_burst(glrtx_ctx *c, float gamma, int reps, float *ms_per_launch) {
    if (!c || !ms_per_launch || reps < 1 || !(gamma > 0.f)) return GLRTX_EINVAL;
    seal_feed(c);
    if (!c->accum || c->owned_rows == 0) return fail(c, GLRTX_EINVAL, "glrtx_debug_resolve_burst: no accumulator");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = ensure(c, c->rgba8, (size_t)c->width * 4 * (size_t)c->owned_rows)) return rc;
    int per = kResolvePer;
    if (const char *v = std::getenv("GLRTX_RESOLVE_PER")) per = std::atoi(v);
    per = per == 1 || per == 4 ? per : 2;
    const dim3 grid = resolve_grid(c->width, c->owned_rows, per);
    const auto rk = per == 1 ? resolve_kernel<1> : (per == 4 ? resolve_kernel<4> : resolve_kernel<2>);
    for (int pass = 0; pass < 2; pass++) {  // (the first pass warms the device up)
        HIP_TRY(c, hipEventRecord(c->rs0, c->stream));
        for (int i = 0; i < reps; i++)
            hipLaunchKernelGGL(rk, grid, dim3(256), 0, c->stream, (const float4 *)c->accum, (int)(c->pitch_bytes / sizeof(float4)), c->width, c->owned_rows,
                               (uchar4 *)c->rgba8.p, c->width, 1.0f / gamma, 1);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(c->rs1, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->rs0, c->rs1));
    *ms_per_launch = ms / (float)reps;
    return GLRTX_OK;
}

// Profile of a calibration frame: how often every triangle of the uploaded scene is the CLOSEST hit of a path ray (camera rays and bounces), counted by the render
// kernel itself while it shades one frame of `p` into a scratch accumulator -- the context's own accumulator, its statistics and its open launches are left alone.
// What it is for: glrt_bvh_order_by_hits (glrt_host.h) puts, at every fork, the child whose subtree is hit more often into the slot the reference's traversal visits
// first (raytrace.frag:299-307) -- a hit found early culls the sibling.  The order of a fork's children is the builder's choice (bvh.cpp:72-160); images stay the
// reference's for the re-ordered tree (exact ties between two triangles may resolve to the other one: INTEGRATION.md).
int glrtx_hit_histogram(glrtx_ctx *c, const glrtx_params *p, uint32_t *hist_out, size_t n_tri) {
    if (!c || !p || !hist_out) return GLRTX_EINVAL;
    seal_feed(c);
    if (!c->have_scene) return fail(c, GLRTX_EINVAL, "glrtx_hit_histogram: no scene uploaded");
    if (!c->accum || c->width < 1 || c->owned_rows < 1) return fail(c, GLRTX_EINVAL, "glrtx_hit_histogram: no image size (call glrtx_resize)");
    if (n_tri != (size_t)c->n_tri) return fail(c, GLRTX_EINVAL, "glrtx_hit_histogram: the scene has %d triangles, room for %zu", c->n_tri, n_tri);
    if (c->variant != 2 || !wgwf_can_hold(p) || c->n_spheres > 0 || c->ext_flags != 0) return fail(c, GLRTX_EINVAL, "glrtx_hit_histogram: wavefront kernel only");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = glrtx_sync(c)) return rc;
    const size_t n_rec = c->leaf_tri.size() + 1;
    DevBuf hist, scratch;
    int rc = GLRTX_OK;
    float4 *const accum_was = c->accum;
    const bool pipeline_was = c->pipeline, count_was = c->count_rays;
    const glrtx_stats st_was = c->st;
    if ((rc = ensure(c, hist, n_rec * sizeof(unsigned))) == GLRTX_OK && (rc = ensure(c, scratch, c->pitch_bytes * (size_t)c->owned_rows)) == GLRTX_OK) {
        hipError_t e = hipMemsetAsync(hist.p, 0, n_rec * sizeof(unsigned), c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(scratch.p, 0, c->pitch_bytes * (size_t)c->owned_rows, c->stream);
        if (e != hipSuccess) rc = fail(c, GLRTX_EDEVICE, "glrtx_hit_histogram: %s", hipGetErrorString(e));
        if (rc == GLRTX_OK) {
            c->accum = (float4 *)scratch.p; c->pipeline = false; c->count_rays = false; c->hit_hist_dev = (unsigned *)hist.p;  // one plain launch on the context's stream
            rc = glrtx_render(c, p);
            c->hit_hist_dev = nullptr; c->accum = accum_was; c->pipeline = pipeline_was; c->count_rays = count_was;
            seal_feed(c);
            if (rc == GLRTX_OK) rc = glrtx_sync(c);
        }
        if (rc == GLRTX_OK) {
            std::vector<unsigned> h(n_rec);
            if (hipMemcpy(h.data(), hist.p, n_rec * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(c, GLRTX_EDEVICE, "glrtx_hit_histogram: read-back failed");
            else {
                std::fill(hist_out, hist_out + n_tri, 0u);
                for (size_t k = 0; k + 1 < n_rec; k++) hist_out[c->leaf_tri[k]] += h[k + 1];
            }
        }
    }
    dev_free(hist); dev_free(scratch);
    const double kms = c->st.kernel_ms_total; (void)kms;
    c->st = st_was;  // (the calibration frame is not part of the caller's statistics)
    return rc;
}

int glrtx_get_stats(const glrtx_ctx *c, glrtx_stats *out) {
    if (!c || !out) return GLRTX_EINVAL;
    // a getter leaves the calling thread's current device as it found it
    int dev_before = -1;
    const bool have_dev = hipGetDevice(&dev_before) == hipSuccess;
    if (hipSetDevice(c->device) == hipSuccess) (void)fold_launches(const_cast<glrtx_ctx *>(c), false);  // launches that have finished; never waits, never consumes a failed one
    *out = c->st;
    // the ray counters live on the device; glrtx_sync brings them over, so after a sync this call touches nothing.  Only when a
    // counting launch was issued since the last sync / read are they fetched here (a blocking copy); polling the stats of a
    // non-counting render loop costs no device round trip
    if (c->counters_stale && c->counter.p && hipSetDevice(c->device) == hipSuccess &&
        hipMemcpy(c->counters_host, c->counter.p, sizeof c->counters_host, hipMemcpyDeviceToHost) == hipSuccess)
        c->counters_stale = false;
    out->rays = c->counters_host[0]; out->rays_untraced = c->counters_host[1];
    out->shadow_limited = c->sc.shadow_limited;
    if (have_dev && dev_before != c->device) (void)hipSetDevice(dev_before);
    return GLRTX_OK;
}

int glrtx_reset_stats(glrtx_ctx *c) {
    if (!c) return GLRTX_EINVAL;
    seal_feed(c);  // (nothing is appended to an open launch across this call: glrtx_ctx::OpenFeed)
    if (int rc = glrtx_sync(c)) return rc;
    HIP_TRY(c, hipMemset(c->counter.p, 0, 2 * sizeof(unsigned long long)));
    c->counters_host[0] = c->counters_host[1] = 0; c->counters_stale = false;
    c->st.rays = 0; c->st.rays_untraced = 0; c->st.paths = 0; c->st.launches = 0; c->st.kernel_launches = 0; c->st.kernel_ms_total = 0.0; c->st.accumulate_ms_total = 0.0; c->st.kernel_ms_last = 0.f; c->st.fallback_launches = 0; c->st.pipe_resident_max = 0; c->st.feed_launches = 0; c->st.feed_appended = 0;
    return GLRTX_OK;
}

#ifdef GLRTX_TRAV_STATS
// diagnostic build only: read and clear the traversal statistics
int glrtx_debug_trav_stats(unsigned long long out[8]) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trav_stats), 8 * sizeof(unsigned long long)) != hipSuccess) return GLRTX_EDEVICE;
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_trav_stats), z, sizeof z) != hipSuccess) return GLRTX_EDEVICE;
    return GLRTX_OK;
}
int glrtx_debug_trav_trips(unsigned long long out[4]) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trav_trips), 4 * sizeof(unsigned long long)) != hipSuccess) return GLRTX_EDEVICE;
    unsigned long long z[4] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_trav_trips), z, sizeof z) != hipSuccess) return GLRTX_EDEVICE;
    return GLRTX_OK;
}
int glrtx_debug_trav_sp_hist(unsigned long long out[16]) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trav_sp_hist), 16 * sizeof(unsigned long long)) != hipSuccess) return GLRTX_EDEVICE;
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_trav_sp_hist), z, sizeof z) != hipSuccess) return GLRTX_EDEVICE;
    return GLRTX_OK;
}
int glrtx_debug_trav_hist(unsigned long long out[16]) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trav_hist), 16 * sizeof(unsigned long long)) != hipSuccess) return GLRTX_EDEVICE;
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_trav_hist), z, sizeof z) != hipSuccess) return GLRTX_EDEVICE;
    return GLRTX_OK;
}
#endif

#ifdef GLRTX_RAY_LOG
// diagnostic build only (tools/gpu_replay.py): record the ray queues of the following launches / replay their traverse phases alone
int glrtx_debug_ray_log_begin(glrtx_ctx *c, unsigned long long max_rays, unsigned max_trips) {
    if (!c) return GLRTX_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, g_dbg_log_rays, (size_t)max_rays * 2 * sizeof(float4)))) return rc;
    if ((rc = ensure(c, g_dbg_log_trips, (size_t)max_trips * sizeof(uint2)))) return rc;
    RayLog lg{(float4 *)g_dbg_log_rays.p, (uint2 *)g_dbg_log_trips.p, max_rays, max_trips, max_rays > 0 ? 1u : 0u};
    unsigned long long z = 0; unsigned zt = 0;
    HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(g_ray_log), &lg, sizeof lg));
    HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(g_ray_log_n), &z, sizeof z));
    HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(g_ray_log_trips), &zt, sizeof zt));
    return GLRTX_OK;
}
// the log's ray records (2 float4 each) and trips to / from the host: reordering experiments (tools/gpu_replay.py)
int glrtx_debug_ray_log_copy(glrtx_ctx *c, float *rays_host, unsigned long long n_records, unsigned *trips_host, unsigned n_trips, int to_device) {
    if (!c) return GLRTX_EINVAL;
    if (int rc = glrtx_sync(c)) return rc;
    if ((size_t)n_records * 32 > g_dbg_log_rays.bytes || (size_t)n_trips * 8 > g_dbg_log_trips.bytes) return fail(c, GLRTX_EINVAL, "ray log: copy larger than the log");
    if (rays_host) HIP_TRY(c, to_device ? hipMemcpy(g_dbg_log_rays.p, rays_host, (size_t)n_records * 32, hipMemcpyHostToDevice)
                                        : hipMemcpy(rays_host, g_dbg_log_rays.p, (size_t)n_records * 32, hipMemcpyDeviceToHost));
    if (trips_host) HIP_TRY(c, to_device ? hipMemcpy(g_dbg_log_trips.p, trips_host, (size_t)n_trips * 8, hipMemcpyHostToDevice)
                                         : hipMemcpy(trips_host, g_dbg_log_trips.p, (size_t)n_trips * 8, hipMemcpyDeviceToHost));
    return GLRTX_OK;
}
// stops recording and runs the traverse phase alone over the log `reps` times; out: {ms of the last replay, rays logged, trips logged, rays offered}
int glrtx_debug_ray_log_replay(glrtx_ctx *c, int reps, double out[4]) {
    if (!c) return GLRTX_EINVAL;
    if (int rc = glrtx_sync(c)) return rc;
    RayLog lg{};
    unsigned long long n = 0; unsigned nt = 0;
    HIP_TRY(c, hipMemcpyFromSymbol(&lg, HIP_SYMBOL(g_ray_log), sizeof lg));
    HIP_TRY(c, hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_ray_log_n), sizeof n));
    HIP_TRY(c, hipMemcpyFromSymbol(&nt, HIP_SYMBOL(g_ray_log_trips), sizeof nt));
    lg.on = 0u;
    HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(g_ray_log), &lg, sizeof lg));
    nt = std::min(nt, lg.cap_trips);
    std::vector<uint2> tr(nt);
    HIP_TRY(c, hipMemcpy(tr.data(), lg.trips, (size_t)nt * sizeof(uint2), hipMemcpyDeviceToHost));
    unsigned long long logged = 0;
    for (const uint2 &t : tr) logged += t.y;
    DbgLastLaunch L = g_dbg_last;
    L.w.suspend_max = 0;
    using RKernel = void (*)(const KernelArgs, const WfArgs, unsigned *, float4 *, const float4 *, const uint2 *, int);
    int fetch = L.fetch;
    if (const char *v = std::getenv("GLRTX_REPLAY_PAIR_FETCH")) fetch = std::max(0, std::min(2, std::atoi(v)));  // the same log through another form of the node fetch
    const RKernel rk = fetch == 2 ? (RKernel)pt_replay_traverse<2> : fetch == 1 ? (RKernel)pt_replay_traverse<1> : (RKernel)pt_replay_traverse<0>;
    if (L.lds > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void *)rk, hipFuncAttributeMaxDynamicSharedMemorySize, L.lds));
    int per_cu = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rk, kBlockThreads, L.lds));
    if (const char *v = std::getenv("GLRTX_WGS_PER_CU")) per_cu = std::max(1, std::min(per_cu, std::atoi(v)));
    const int grid = std::max(1, std::min(L.grid, per_cu * c->n_cu));
    float ms = 0.f;
    for (int r = 0; r < reps; r++) {
        HIP_TRY(c, hipMemsetAsync(c->work.p, 0, sizeof(unsigned), c->stream));
        HIP_TRY(c, hipEventRecord(c->tm0, c->stream));
        hipLaunchKernelGGL(rk, dim3(grid), dim3(kBlockThreads), L.lds, c->stream, L.a, L.w, (unsigned *)c->work.p, L.queues,
                           (const float4 *)lg.rays, (const uint2 *)lg.trips, (int)nt);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(c->tm1, c->stream));
        HIP_TRY(c, hipEventSynchronize(c->tm1));
        HIP_TRY(c, hipEventElapsedTime(&ms, c->tm0, c->tm1));
    }
    out[0] = ms; out[1] = (double)logged; out[2] = (double)nt; out[3] = (double)n;
    return GLRTX_OK;
}
#endif

#ifdef GLRTX_STEP_TIMING
int glrtx_debug_step_timing(unsigned long long out[4]) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_step_timing), 4 * sizeof(unsigned long long)) != hipSuccess) return GLRTX_EDEVICE;
    unsigned long long z[4] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_step_timing), z, sizeof z) != hipSuccess) return GLRTX_EDEVICE;
    return GLRTX_OK;
}
#endif

#ifdef GLRTX_PHASE_STATS
int glrtx_debug_trip_log(unsigned out[16 * 64 * 4]) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trip_log), 16 * 64 * 16) != hipSuccess) return GLRTX_EDEVICE;
    static unsigned z[16 * 64 * 4];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_trip_log), z, sizeof z) != hipSuccess) return GLRTX_EDEVICE;
    return GLRTX_OK;
}
int glrtx_debug_phase_cycles(unsigned long long out[8]) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_cycles), 8 * sizeof(unsigned long long)) != hipSuccess) return GLRTX_EDEVICE;
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof z) != hipSuccess) return GLRTX_EDEVICE;
    return GLRTX_OK;
}
#endif

int glrtx_timer_begin(glrtx_ctx *c) {
    if (!c) return GLRTX_EINVAL;
    seal_feed(c);  // (nothing is appended to an open launch across this call: glrtx_ctx::OpenFeed)
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventRecord(c->tm0, c->stream));
    return GLRTX_OK;
}

int glrtx_timer_end(glrtx_ctx *c, float *ms) {
    if (!c || !ms) return GLRTX_EINVAL;
    seal_feed(c);  // (nothing is appended to an open launch across this call: glrtx_ctx::OpenFeed)
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventRecord(c->tm1, c->stream));
    HIP_TRY(c, hipEventSynchronize(c->tm1));
    HIP_TRY(c, hipEventElapsedTime(ms, c->tm0, c->tm1));
    return GLRTX_OK;
}

// ---------------------------------------------------------------------------------------------------------------- groups
// Several contexts -- one per GPU of the node -- behind one handle, driven by one host thread: the multi-GPU form of the
// same entry points.  Context i owns the 8-row stripes s with s % n == i (glrtx_set_partition), renders
// them on its own stream with global pixel coordinates, and keeps its accumulator rows resident.  Nothing is exchanged while
// rendering.  Only when an image is wanted (read_accum / resolve) are the stripes copied, device to device, into a full-frame
// buffer on the first context's GPU (peer copies over xGMI; hipMemcpyPeerAsync also serves the same-device case the tests use).
}  // extern "C"

struct glrtx_group {
    std::vector<glrtx_ctx *> ctx;
    std::vector<hipEvent_t> done;  // per context: "its render stream has reached this point"
    DevBuf full, full8;            // on ctx[0]'s device: gathered accumulator (pitch = ctx[0]'s), resolved RGBA8
    std::vector<char> peer_ok;     // per context: its device can write the root's memory directly (same device, or peer access enabled)
    int gather_copies = 0;         // copies the last gather issued (tests)
    int width = 0, height = 0;
    std::string err;
};

namespace {

int gfail(glrtx_group *g, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g) g->err = buf; else g_create_error = buf;
    return code;
}
// forward a member context's failure
int gsub(glrtx_group *g, int i, int rc) {
    if (rc != GLRTX_OK) g->err = "context " + std::to_string(i) + " (device " + std::to_string(g->ctx[i]->device) + "): " + g->ctx[i]->err;
    return rc;
}
#define GHIP_TRY(g, call)                                                                       \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) return gfail(g, GLRTX_EDEVICE, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// All stripes -> g->full on ctx[0]'s device, ordered behind every context's outstanding work.  ONE strided copy per member
// (hipMemcpy2DAsync: a row of the copy is a whole stripe -- stripe x pitch bytes, contiguous on both sides -- the source steps by a
// stripe, the destination by world stripes), plus one for a partial last stripe, issued on the MEMBER's own stream right behind its
// render work: the members' copies run side by side over their own xGMI links, and the root stream only waits for their events.
// (Round 2 issued every stripe as its own hipMemcpyPeerAsync on the root's stream: 135 serialised copies at 1080p / 8 members.)
// Members without peer access to the root's memory fall back to hipMemcpyPeerAsync per stripe on the root's stream.
int group_gather(glrtx_group *g) {
    for (glrtx_ctx *m : g->ctx) seal_feed(m);  // (the copies read the members' accumulators in stream order: nothing is appended to a launch queued in front of them)
    glrtx_ctx *r = g->ctx[0];
    if (g->width < 1) return gfail(g, GLRTX_EINVAL, "glrtx_group: call glrtx_group_resize first");
    const size_t pitch = r->pitch_bytes;
    GHIP_TRY(g, hipSetDevice(r->device));
    if (g->full.bytes < pitch * (size_t)g->height) {
        dev_free(g->full);
        GHIP_TRY(g, hipMalloc(&g->full.p, pitch * (size_t)g->height));
        g->full.bytes = pitch * (size_t)g->height;
    }
    const int n = (int)g->ctx.size();
    g->gather_copies = 0;
    for (int i = 0; i < n; i++) {
        glrtx_ctx *c = g->ctx[i];
        if (c->pitch_bytes != pitch) return gfail(g, GLRTX_EINVAL, "glrtx_group: member accumulators have different pitches");
        if (c->owned_rows == 0) continue;
        const size_t stripe_b = (size_t)c->stripe * pitch;
        const int full_stripes = c->owned_rows / c->stripe, tail_rows = c->owned_rows % c->stripe;
        char *dst0 = (char *)g->full.p + (size_t)c->rank * stripe_b;  // stripe s of this member lies at global stripe s * world + rank
        if (g->peer_ok[i]) {
            GHIP_TRY(g, hipSetDevice(c->device));
            if (full_stripes > 0) {
                GHIP_TRY(g, hipMemcpy2DAsync(dst0, (size_t)c->world * stripe_b, c->accum, stripe_b, stripe_b, (size_t)full_stripes, hipMemcpyDeviceToDevice, c->stream));
                g->gather_copies++;
            }
            if (tail_rows > 0) {
                GHIP_TRY(g, hipMemcpyAsync(dst0 + (size_t)full_stripes * c->world * stripe_b, (const char *)c->accum + (size_t)full_stripes * stripe_b,
                                           (size_t)tail_rows * pitch, hipMemcpyDeviceToDevice, c->stream));
                g->gather_copies++;
            }
            GHIP_TRY(g, hipEventRecord(g->done[i], c->stream));
            GHIP_TRY(g, hipSetDevice(r->device));
            GHIP_TRY(g, hipStreamWaitEvent(r->stream, g->done[i], 0));
        } else {
            GHIP_TRY(g, hipSetDevice(c->device));
            GHIP_TRY(g, hipEventRecord(g->done[i], c->stream));
            GHIP_TRY(g, hipSetDevice(r->device));
            GHIP_TRY(g, hipStreamWaitEvent(r->stream, g->done[i], 0));
            for (int row = 0; row < c->owned_rows; row += c->stripe) {  // one contiguous block per stripe
                const int rows = std::min(c->stripe, c->owned_rows - row);
                const int y = ((row / c->stripe) * c->world + c->rank) * c->stripe;
                GHIP_TRY(g, hipMemcpyPeerAsync((char *)g->full.p + (size_t)y * pitch, r->device, (const char *)c->accum + (size_t)row * pitch, c->device,
                                               (size_t)rows * pitch, r->stream));
                g->gather_copies++;
            }
        }
    }
    return GLRTX_OK;
}

}  // namespace

extern "C" {

int glrtx_group_create(glrtx_group **out, const int *device_ids, int n_devices) {
    if (!out) return gfail(nullptr, GLRTX_EINVAL, "glrtx_group_create: out is NULL");
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64) return gfail(nullptr, GLRTX_EINVAL, "glrtx_group_create: need 1..64 device ids");
    glrtx_group *g = new (std::nothrow) glrtx_group;
    if (!g) return gfail(nullptr, GLRTX_ENOMEM, "out of host memory");
    for (int i = 0; i < n_devices; i++) {
        glrtx_ctx *c = nullptr;
        const int rc = glrtx_create(&c, device_ids[i]);
        if (rc != GLRTX_OK) { glrtx_group_destroy(g); return rc; }  // message already in the create-error slot
        g->ctx.push_back(c);
        hipEvent_t ev = nullptr;
        if (hipSetDevice(c->device) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            glrtx_group_destroy(g);
            return gfail(nullptr, GLRTX_EDEVICE, "glrtx_group_create: event creation failed on device %d", device_ids[i]);
        }
        g->done.push_back(ev);
    }
    // every member's device gets direct access to the root's memory where the platform offers it (its gather copy then runs on its own
    // stream over its own link); without it that member's stripes are copied by the root with hipMemcpyPeerAsync
    g->peer_ok.assign((size_t)n_devices, 1);
    for (int i = 1; i < n_devices; i++) {
        if (g->ctx[i]->device == g->ctx[0]->device) continue;
        int can = 0;
        g->peer_ok[i] = 0;
        if (hipDeviceCanAccessPeer(&can, g->ctx[i]->device, g->ctx[0]->device) == hipSuccess && can) {
            (void)hipSetDevice(g->ctx[i]->device);
            const hipError_t e = hipDeviceEnablePeerAccess(g->ctx[0]->device, 0);
            (void)hipGetLastError();  // "already enabled" is fine
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) g->peer_ok[i] = 1;
        }
    }
    // members that share a GPU also share its memory: the frames-in-flight budget is split between them
    for (int i = 0; i < n_devices; i++) {
        int same = 0;
        for (int j = 0; j < n_devices; j++) same += g->ctx[j]->device == g->ctx[i]->device;
        g->ctx[i]->budget_share = same;
    }
    *out = g;
    return GLRTX_OK;
}

void glrtx_group_destroy(glrtx_group *g) {
    if (!g) return;
    for (size_t i = 0; i < g->done.size(); i++) { (void)hipSetDevice(g->ctx[i]->device); (void)hipEventDestroy(g->done[i]); }
    if (!g->ctx.empty()) { (void)hipSetDevice(g->ctx[0]->device); (void)hipStreamSynchronize(g->ctx[0]->stream); dev_free(g->full); dev_free(g->full8); }
    for (glrtx_ctx *c : g->ctx) glrtx_destroy(c);
    delete g;
}

const char *glrtx_group_last_error(const glrtx_group *g) { return g ? g->err.c_str() : g_create_error.c_str(); }
int glrtx_group_size(const glrtx_group *g) { return g ? (int)g->ctx.size() : 0; }
glrtx_ctx *glrtx_group_ctx(glrtx_group *g, int i) { return (g && i >= 0 && i < (int)g->ctx.size()) ? g->ctx[i] : nullptr; }

int glrtx_group_upload_scene(glrtx_group *g, const float *vert, size_t n_vert, const float *tri, size_t n_tri, const float *mat, size_t n_mat,
                             const float *light, size_t n_light, const float *bvh, size_t n_nodes) {
    if (!g) return GLRTX_EINVAL;
    for (size_t i = 0; i < g->ctx.size(); i++)  // the scene is replicated: <= 27 MB even for 100k triangles
        if (int rc = gsub(g, (int)i, glrtx_upload_scene(g->ctx[i], vert, n_vert, tri, n_tri, mat, n_mat, light, n_light, bvh, n_nodes))) return rc;
    return GLRTX_OK;
}

int glrtx_group_resize(glrtx_group *g, int width, int height) {
    if (!g) return GLRTX_EINVAL;
    const int n = (int)g->ctx.size();
    for (int i = 0; i < n; i++) {
        glrtx_ctx *c = g->ctx[i];
        c->rank = i; c->world = n; c->stripe = kGroupStripe;  // == glrtx_set_partition(c, i, n, 8) without the intermediate resize
        if (int rc = gsub(g, i, glrtx_resize(c, width, height))) return rc;
    }
    g->width = width; g->height = height;
    return GLRTX_OK;
}

int glrtx_group_clear(glrtx_group *g) {
    if (!g) return GLRTX_EINVAL;
    for (size_t i = 0; i < g->ctx.size(); i++)
        if (int rc = gsub(g, (int)i, glrtx_clear(g->ctx[i]))) return rc;
    return GLRTX_OK;
}

int glrtx_group_render(glrtx_group *g, const glrtx_params *p) {
    if (!g) return GLRTX_EINVAL;
    for (size_t i = 0; i < g->ctx.size(); i++)  // asynchronous launches: the GPUs run concurrently
        if (int rc = gsub(g, (int)i, glrtx_render(g->ctx[i], p))) return rc;
    return GLRTX_OK;
}

int glrtx_group_render_frames(glrtx_group *g, const glrtx_params *p, const float *seeds_xy, int n_frames) {
    if (!g) return GLRTX_EINVAL;
    for (size_t i = 0; i < g->ctx.size(); i++)
        if (int rc = gsub(g, (int)i, glrtx_render_frames(g->ctx[i], p, seeds_xy, n_frames))) return rc;
    return GLRTX_OK;
}

int glrtx_group_sync(glrtx_group *g) {
    if (!g) return GLRTX_EINVAL;
    for (size_t i = 0; i < g->ctx.size(); i++)
        if (int rc = gsub(g, (int)i, glrtx_sync(g->ctx[i]))) return rc;
    return GLRTX_OK;
}

int glrtx_group_read_accum(glrtx_group *g, float *dst, size_t dst_pitch_bytes) {
    if (!g || !dst) return GLRTX_EINVAL;
    if (dst_pitch_bytes < (size_t)g->width * sizeof(float4)) return gfail(g, GLRTX_EINVAL, "glrtx_group_read_accum: dst pitch too small");
    if (int rc = group_gather(g)) return rc;
    glrtx_ctx *r = g->ctx[0];
    GHIP_TRY(g, hipStreamSynchronize(r->stream));
    GHIP_TRY(g, hipMemcpy2D(dst, dst_pitch_bytes, g->full.p, r->pitch_bytes, (size_t)g->width * sizeof(float4), (size_t)g->height, hipMemcpyDeviceToHost));
    return glrtx_group_sync(g);
}

int glrtx_group_resolve_rgba8(glrtx_group *g, uint8_t *dst, size_t dst_pitch_bytes, float gamma, int flip_y) {
    if (!g || !dst) return GLRTX_EINVAL;
    if (!(gamma > 0.f)) return gfail(g, GLRTX_EINVAL, "glrtx_group_resolve_rgba8: gamma must be positive");
    if (dst_pitch_bytes < (size_t)g->width * 4) return gfail(g, GLRTX_EINVAL, "glrtx_group_resolve_rgba8: dst pitch too small");
    if (int rc = group_gather(g)) return rc;
    glrtx_ctx *r = g->ctx[0];
    const size_t bytes = (size_t)g->width * 4 * (size_t)g->height;
    if (g->full8.bytes < bytes) {
        dev_free(g->full8);
        GHIP_TRY(g, hipMalloc(&g->full8.p, bytes));
        g->full8.bytes = bytes;
    }
    const dim3 grid = resolve_grid(g->width, g->height);
    hipLaunchKernelGGL(resolve_kernel<kResolvePer>, grid, dim3(256), 0, r->stream, (const float4 *)g->full.p, (int)(r->pitch_bytes / sizeof(float4)), g->width, g->height,
                       (uchar4 *)g->full8.p, g->width, 1.0f / gamma, flip_y ? 1 : 0);
    GHIP_TRY(g, hipGetLastError());
    GHIP_TRY(g, hipStreamSynchronize(r->stream));
    GHIP_TRY(g, hipMemcpy2D(dst, dst_pitch_bytes, g->full8.p, (size_t)g->width * 4, (size_t)g->width * 4, (size_t)g->height, hipMemcpyDeviceToHost));
    return glrtx_group_sync(g);
}

int glrtx_group_gather_copies(const glrtx_group *g) { return g ? g->gather_copies : 0; }

int glrtx_group_get_stats(const glrtx_group *g, glrtx_stats *out) {
    if (!g || !out || g->ctx.empty()) return GLRTX_EINVAL;
    glrtx_stats s{};
    for (size_t i = 0; i < g->ctx.size(); i++) {
        glrtx_stats t{};
        if (int rc = glrtx_get_stats(g->ctx[i], &t)) return rc;
        if (i == 0) s = t;
        else {
            s.rays += t.rays; s.rays_untraced += t.rays_untraced; s.paths += t.paths; s.owned_rows += t.owned_rows;
            s.fallback_launches += t.fallback_launches; s.fallback_last |= t.fallback_last;
            s.kernel_ms_total = std::max(s.kernel_ms_total, t.kernel_ms_total);  // the GPUs run side by side
            s.accumulate_ms_total = std::max(s.accumulate_ms_total, t.accumulate_ms_total);
            s.kernel_ms_last = std::max(s.kernel_ms_last, t.kernel_ms_last);
        }
    }
    *out = s;
    return GLRTX_OK;
}

}  // extern "C"
