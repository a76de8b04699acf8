/*
 * pt_oracle.c -- CPU restatement of the reference's per-pixel path tracer.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product path (opengl-raytracer_amd/)
 * never does.
 *
 * PARITY STATUS: PINNED.  This restatement is checked bit-for-bit against the
 * reference's own unmodified fragment shader executed on Mesa llvmpipe
 * (oracle/glref.py -> tests/golden/*.npz, tests/test_oracle_golden.py).
 *
 * What it restates (all line numbers are src/shaders/raytrace.frag of the
 * reference unless noted):
 *   rand()            :104-111      main()              :565-614
 *   radiance()        :409-559      intersect(Ray,...)  :276-335
 *   intersectBBox     :259-274      intersect(Ray,Tri)  :226-257
 *   sampleDirect      :337-403      fresnelConductor    :158-178
 *   GGX               :180-184      microfacetGGXBRDF   :186-193
 *   sampleGGXVNDF     :195-214      weightedGGXPDF      :216-219
 * Buffer layouts: scene.h:16-35, trimesh.h:15-25, bvh.h:84-100 (SURVEY.md
 * Appendix B).  Uniform semantics: window.cpp:230-269.
 *
 * Evaluation order.  The reference image is a chaotic function of float32
 * rounding (the RNG is fract(sin(dot)*c)), so every expression below is
 * written in the association order in which llvmpipe (Mesa 23.2.1, LLVM 15)
 * actually evaluates the reference shader, as observed from its compiler's
 * own debug output and confirmed by the golden images:
 *   - no mul+add contraction anywhere except inside sin/cos;
 *   - dot(a,b) over 3 components is ((a.z*b.z + a.y*b.y) + a.x*b.x);
 *   - normalize(v) = v * (1/sqrt(dot(v,v))), 1/x and a/b are IEEE divisions;
 *   - min/max return the other operand when one is NaN (x86 minps/maxps plus a
 *     NaN select; see the FMIN/FMAX macros) -- this decides BVH culling for
 *     rays with an exactly-zero direction component (0*inf slabs);
 *   - several algebraic rewrites (u+v>1 tested as inv*(U+V)>1, sqrt(d)*sqrt(d)
 *     folded to d, (a-b)+b folded to a, constant products pre-multiplied) --
 *     each is marked "[order]" where it occurs;
 *   - sin/cos are the Cephes single-precision routines with fused
 *     multiply-adds.
 * Compile with -ffp-contract=off (oracle/Makefile does).
 */
#include <math.h>
#include <stdint.h>
#include <xmmintrin.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define PT_EPS 1.0e-4f
#define PT_INFTY 1.0e8f
#define PT_PI 3.14159274101257324f /* float(3.14159265358979...) */
#define PT_2PI 6.28318548202514648f /* float(2*PI), folded by the GLSL compiler [order] */

/* GLSL min/max as llvmpipe lowers them (gallivm "NaN returns the other operand"):
 *   two unknown operands: r = minps(a, b) = a<b ? a : b, then b-is-NaN selects a;
 *   one operand a constant c: minps(x, c) = x<c ? x : c (so NaN x gives c).
 * The raw x86 forms are also used directly by the sin/cos clamp. */
#define MINPS(a, b) ((a) < (b) ? (a) : (b))
#define MAXPS(a, b) ((a) > (b) ? (a) : (b))
#define FMIN(a, b) ((b) != (b) ? (a) : MINPS(a, b))
#define FMAX(a, b) ((b) != (b) ? (a) : MAXPS(a, b))
#define FMIN_C(x, c) MINPS(x, c) /* min(x, const) and min(const, x) */
#define FMAX_C(x, c) MAXPS(x, c) /* max(x, const) and max(const, x) */

#define INL static inline __attribute__((always_inline))

#ifdef PT_TRACE
#include <stdio.h>
#define TRACE(...) fprintf(stderr, __VA_ARGS__)
#else
#define TRACE(...) ((void)0)
#endif

typedef struct {
    const float *vert;  /* nV * 5 texels * 3 floats (pos, normal, uv, tangent, binormal) */
    const float *tri;   /* nT * 4 floats (i, j, k, material) */
    const float *mat;   /* nM * 6 texels * 3 floats */
    const float *light; /* nL * 4 floats */
    const float *bvh;   /* nN * 3 texels * 3 floats (min, max, children) */
    int n_vert, n_tri, n_mat, n_light, n_nodes;
} pt_scene;

typedef struct {
    float c2w[16]; /* column-major, as glUniformMatrix4fv(..., GL_FALSE, ...) */
    float s2c[16];
    float aperture, focal;
    float seed[2];
    int n_samples, max_depth;
    int width, height;
} pt_params;

typedef struct { float x, y, z; } v3;

/* ------------------------------------------------------------------ sin/cos
 * Cephes sinf/cosf as evaluated by llvmpipe (gallivm): every a*b+c below that
 * is written with fmaf is a single fused operation; everything else is not.
 */
INL float pt_sincos_core(float xabs, int je, int use_sinpoly) {
    float yf = (float)je;
    float x = __builtin_fmaf(yf, -0.78515625f, xabs);
    x = __builtin_fmaf(yf, -2.4187564849853515625e-4f, x);
    x = __builtin_fmaf(yf, -3.77489497744594108e-8f, x);
    float z = x * x;
    if (use_sinpoly) {
        float s = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
        s = __builtin_fmaf(s, z, -1.6666654611e-1f);
        s = s * z;
        return __builtin_fmaf(s, x, x);
    } else {
        float c = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
        c = __builtin_fmaf(c, z, 4.166664568298827e-2f);
        c = c * z;
        c = c * z;
        c = c - z * 0.5f;
        return c + 1.0f;
    }
}

INL float pt_clamp_unit(float r, float x) {
    union { float f; uint32_t u; } b; b.f = x;
    if ((b.u & 0x7f800000u) == 0x7f800000u) return NAN;
    r = MINPS(r, 1.0f);
    r = MAXPS(r, -1.0f);
    return r;
}

INL float pt_sin(float x) {
    union { float f; uint32_t u; } b; b.f = x;
    float xabs = fabsf(x);
    int j1 = (int)(xabs * 1.27323954473516f) + 1;
    int je = j1 & ~1;
    uint32_t sign = (b.u ^ ((uint32_t)j1 << 29)) & 0x80000000u;
    union { float f; uint32_t u; } r;
    r.f = pt_sincos_core(xabs, je, (je & 2) == 0);
    r.u ^= sign;
    return pt_clamp_unit(r.f, x);
}

INL float pt_cos(float x) {
    float xabs = fabsf(x);
    int je = ((int)(xabs * 1.27323954473516f) + 1) & ~1;
    int j2 = je - 2;
    uint32_t sign = ((uint32_t)(~j2) & 4u) << 29;
    union { float f; uint32_t u; } r;
    r.f = pt_sincos_core(xabs, je, (j2 & 2) == 0);
    r.u ^= sign;
    return pt_clamp_unit(r.f, x);
}

/* ---------------------------------------------------------------- rand() :104-111 */
typedef struct { float x, y, sx, sy; } pt_rng; /* state + seed */

INL float pt_rand(pt_rng *s) {
    const float a = 12.9898f, b = 78.233f, c = 43758.5453f;
    float dy = (s->y - s->sy) * b; /* shared by both updates (old state.y) */
    float t = dy + (s->x - s->sx) * a;
    float p = pt_sin(t) * c;
    s->x = p - floorf(p);
    t = dy + (s->x - s->sx) * a;
    p = pt_sin(t) * c;
    s->y = p - floorf(p);
    return s->x;
}

/* ------------------------------------------------------------ buffer fetches
 * Out-of-range texelFetch yields zeros on Mesa; keep that so a malformed scene
 * cannot crash the checker. */
INL v3 fetch3(const float *buf, int n_texels, int i) {
    v3 r = {0.f, 0.f, 0.f};
    if (i >= 0 && i < n_texels) { r.x = buf[3 * i]; r.y = buf[3 * i + 1]; r.z = buf[3 * i + 2]; }
    return r;
}
INL void fetch4(const float *buf, int n_texels, int i, float o[4]) {
    if (i >= 0 && i < n_texels) memcpy(o, buf + 4 * (size_t)i, 16);
    else o[0] = o[1] = o[2] = o[3] = 0.f;
}

/* dot over 3 components in the order llvmpipe evaluates it [order] */
INL float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return (az * bz + ay * by) + ax * bx;
}
INL float rsq(float x) { return 1.0f / sqrtf(x); }

/* ------------------------------------------ intersect(Ray, Triangle) :226-257
 * Returns t or INFTY; *n is written only on a hit (callers select on t<tHit). */
INL float pt_tri(v3 o, v3 d, v3 v0, v3 v1, v3 v2, v3 n0, v3 n1, v3 n2, int want_normal, v3 *n) {
    float e1x = v1.x - v0.x, e1y = v1.y - v0.y, e1z = v1.z - v0.z;
    float e2x = v2.x - v0.x, e2y = v2.y - v0.y, e2z = v2.z - v0.z;
    float px = d.y * e2z - d.z * e2y;
    float py = d.z * e2x - d.x * e2z;
    float pz = d.x * e2y - d.y * e2x;
    float det = dot3(e1x, e1y, e1z, px, py, pz);
    if (-PT_EPS < det && det < PT_EPS) return PT_INFTY;
    float inv = 1.0f / det;
    float tx = o.x - v0.x, ty = o.y - v0.y, tz = o.z - v0.z;
    float U = dot3(tx, ty, tz, px, py, pz);
    float u = U * inv;
    if (u < 0.0f || 1.0f < u) return PT_INFTY;
    float qx = ty * e1z - tz * e1y;
    float qy = tz * e1x - tx * e1z;
    float qz = tx * e1y - ty * e1x;
    float V = dot3(d.x, d.y, d.z, qx, qy, qz);
    float v = V * inv;
    if (v < 0.0f || 1.0f < inv * (U + V)) return PT_INFTY; /* u+v>1 as inv*(U+V)>1 [order] */
    float t = dot3(e2x, e2y, e2z, qx, qy, qz) * inv;
    if (PT_EPS >= t) return PT_INFTY;
    if (want_normal) {
        float w0 = (1.0f - u) - v;
        float nx = (w0 * n0.x + u * n1.x) + v * n2.x;
        float ny = (w0 * n0.y + u * n1.y) + v * n2.y;
        float nz = (w0 * n0.z + u * n1.z) + v * n2.z;
        float r = rsq(dot3(nx, ny, nz, nx, ny, nz));
        n->x = nx * r; n->y = ny * r; n->z = nz * r;
    }
    return t;
}

/* ------------------------------------------- intersect(Ray, Intersection) :276-335 */
typedef struct { v3 norm; float tHit; int mtrl; int hit; int tri; unsigned visits; } pt_isect;

INL void pt_traverse(const pt_scene *sc, v3 o, v3 d, int want_normal, pt_isect *is) {
    int stack[64];
    int pos = 0;
    stack[0] = 0;
    is->tHit = PT_INFTY; is->norm.x = is->norm.y = is->norm.z = 0.f; is->mtrl = 0; is->hit = 0; is->tri = -1; is->visits = 0;
    const int n_bvh_texels = sc->n_nodes * 3, n_vert_texels = sc->n_vert * 5;
    float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z; /* :260, recomputed per node there */
    while (pos >= 0) {
        int slot = pos;
        int node = stack[slot];
        is->visits++;
        pos -= 1;
        v3 bmin = fetch3(sc->bvh, n_bvh_texels, node * 3 + 0);
        v3 bmax = fetch3(sc->bvh, n_bvh_texels, node * 3 + 1);
        v3 ch = fetch3(sc->bvh, n_bvh_texels, node * 3 + 2);
        if (ch.z < 0.0f) {
            /* fork :295-309, intersectBBox :259-274 */
            float fx = (bmax.x - o.x) * ix, fy = (bmax.y - o.y) * iy, fz = (bmax.z - o.z) * iz;
            float nx = (bmin.x - o.x) * ix, ny = (bmin.y - o.y) * iy, nz = (bmin.z - o.z) * iz;
            float tmaxx = FMAX(fx, nx), tmaxy = FMAX(fy, ny), tmaxz = FMAX(fz, nz);
            float tminx = FMIN(fx, nx), tminy = FMIN(fy, ny), tminz = FMIN(fz, nz);
            float t1 = FMIN(tmaxy, tmaxz); t1 = FMIN(tmaxx, t1);
            float t0 = FMAX(tminy, tminz); t0 = FMAX(tminx, t0);
            /* (t1 >= t0 && t0 <= tHit) evaluated as min(t1, tHit) >= t0 [order] */
            float tc = FMIN(t1, is->tHit);
            if (tc >= t0) {
                if (ch.x >= 0.0f) { stack[slot & 63] = (int)ch.x; pos = slot; }
                if (ch.y >= 0.0f) { pos += 1; stack[pos & 63] = (int)ch.y; }
            }
        } else {
            int index = (int)ch.z;
            float tr[4];
            fetch4(sc->tri, sc->n_tri, index, tr);
            int i0 = (int)tr[0] * 5, i1 = (int)tr[1] * 5, i2 = (int)tr[2] * 5;
            v3 v0 = fetch3(sc->vert, n_vert_texels, i0), v1 = fetch3(sc->vert, n_vert_texels, i1),
               v2 = fetch3(sc->vert, n_vert_texels, i2);
            v3 n0 = {0, 0, 0}, n1 = {0, 0, 0}, n2 = {0, 0, 0}, n = is->norm;
            if (want_normal) {
                n0 = fetch3(sc->vert, n_vert_texels, i0 + 1);
                n1 = fetch3(sc->vert, n_vert_texels, i1 + 1);
                n2 = fetch3(sc->vert, n_vert_texels, i2 + 1);
            }
            float dist = pt_tri(o, d, v0, v1, v2, n0, n1, n2, want_normal, &n);
            if (dist < is->tHit) {
                is->norm = n;
                is->mtrl = (int)tr[3];
                is->hit = 1;
                is->tri = index;
            }
            is->tHit = FMIN(is->tHit, dist);
        }
    }
}

#ifdef PT_ORDERED_EXPERIMENT
/* EXPERIMENT (not part of the oracle proper): a near-child-first traversal with the reference's
 * DFS leaf rank as tie-break, run next to the reference-order traversal to count (a) disagreements
 * in (tri, t) and (b) node visits of both orders.  Build: gcc -DPT_ORDERED_EXPERIMENT ... */
#include <stdio.h>
#include <stdlib.h>
unsigned long long g_exp[16]; /* 0 rays, 1 mismatches, 2 ref visits, 3 ord visits, 4 ref max, 5 ord max */
static int *g_rank = NULL;    /* triangle -> position in the reference's right-first DFS leaf order */
static void exp_build_rank(const pt_scene *sc) {
    g_rank = (int *)malloc(sizeof(int) * (size_t)(sc->n_tri + 1));
    int *st = (int *)malloc(sizeof(int) * (size_t)(sc->n_nodes + 2));
    int sp = 0, r = 0;
    st[sp++] = 0;
    while (sp) {
        int n = st[--sp];
        const float *c = sc->bvh + 9 * (size_t)n + 6;
        if (c[2] < 0.0f) { if (c[0] >= 0) st[sp++] = (int)c[0]; if (c[1] >= 0) st[sp++] = (int)c[1]; }
        else g_rank[(int)c[2]] = r++;
    }
    free(st);
}
static float exp_box(const pt_scene *sc, int node, v3 o, float ix, float iy, float iz, float tHit, int *pass) {
    const float *b = sc->bvh + 9 * (size_t)node;
    float fx = (b[3] - o.x) * ix, fy = (b[4] - o.y) * iy, fz = (b[5] - o.z) * iz;
    float nx = (b[0] - o.x) * ix, ny = (b[1] - o.y) * iy, nz = (b[2] - o.z) * iz;
    float tmaxx = FMAX(fx, nx), tmaxy = FMAX(fy, ny), tmaxz = FMAX(fz, nz);
    float tminx = FMIN(fx, nx), tminy = FMIN(fy, ny), tminz = FMIN(fz, nz);
    float t1 = FMIN(tmaxy, tmaxz); t1 = FMIN(tmaxx, t1);
    float t0 = FMAX(tminy, tminz); t0 = FMAX(tminx, t0);
    *pass = FMIN(t1, tHit) >= t0;
    return t0;
}
static void exp_ordered(const pt_scene *sc, v3 o, v3 d, float *t_out, int *tri_out, unsigned *visits) {
    struct { int node; float t0; } st[128];
    int sp = 0, best = -1;
    float tHit = PT_INFTY, ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
    const int nvt = sc->n_vert * 5;
    int pass; float t0 = exp_box(sc, 0, o, ix, iy, iz, tHit, &pass);
    const float *c0 = sc->bvh + 6;
    if (c0[2] >= 0.0f) { pass = 1; t0 = -PT_INFTY; }
    if (pass) { st[sp].node = 0; st[sp].t0 = t0; sp++; }
    while (sp) {
        sp--;
        int n = st[sp].node;
        if (!(tHit >= st[sp].t0)) continue;
        (*visits)++;
        const float *c = sc->bvh + 9 * (size_t)n + 6;
        if (c[2] < 0.0f) {
            int kid[2] = {(int)c[0], (int)c[1]}, ok[2] = {0, 0}; float kt[2] = {-PT_INFTY, -PT_INFTY};
            for (int k = 0; k < 2; k++) {
                if (!(c[k] >= 0.0f)) continue;
                if (sc->bvh[9 * (size_t)kid[k] + 8] >= 0.0f) ok[k] = 1; /* leaf child: never box-tested */
                else kt[k] = exp_box(sc, kid[k], o, ix, iy, iz, tHit, &ok[k]);
            }
            /* near child on top of the stack */
            int first = (ok[0] && ok[1]) ? (kt[0] <= kt[1] ? 0 : 1) : (ok[0] ? 0 : 1);
            int second = 1 - first;
            if (ok[second]) { st[sp].node = kid[second]; st[sp].t0 = kt[second]; sp++; }
            if (ok[first]) { st[sp].node = kid[first]; st[sp].t0 = kt[first]; sp++; }
        } else {
            int index = (int)c[2]; float tr[4]; fetch4(sc->tri, sc->n_tri, index, tr);
            int i0 = (int)tr[0] * 5, i1 = (int)tr[1] * 5, i2 = (int)tr[2] * 5;
            v3 v0 = fetch3(sc->vert, nvt, i0), v1 = fetch3(sc->vert, nvt, i1), v2 = fetch3(sc->vert, nvt, i2), z = {0, 0, 0}, n3;
            float dist = pt_tri(o, d, v0, v1, v2, z, z, z, 0, &n3);
            if (dist < tHit || (dist == tHit && best >= 0 && dist < PT_INFTY && g_rank[index] < g_rank[best])) best = index;
            tHit = FMIN(tHit, dist);
        }
    }
    *t_out = tHit; *tri_out = best;
}
#endif

#ifdef PT_ORDERED_EXPERIMENT
/* EXPERIMENT 2: step counts of the device traversal (children's boxes in the parent, reference order, t0 recheck at pop)
 * under two record layouts.  g_exp[8] fork steps, [9] leaf steps (current layout: one 64-B record per step);
 * [10] fetches with 128-B "slot" records (a fork + its first-visited child inline), [11] inline visits saved. */
static unsigned char *g_kind = NULL; /* 1 = processed inline from the parent's slot */
static void exp_build_kind(const pt_scene *sc) {
    g_kind = (unsigned char *)calloc((size_t)sc->n_nodes + 1, 1);
    int *st = (int *)malloc(sizeof(int) * (size_t)(sc->n_nodes + 2));
    int sp = 0;
    st[sp++] = 0;
    while (sp) {
        int n = st[--sp];
        const float *c = sc->bvh + 9 * (size_t)n + 6;
        if (c[2] >= 0.0f) continue;
        int x = (int)c[0], y = (int)c[1];
        if (c[1] >= 0.0f) { g_kind[y] = g_kind[n] ? 0 : 1; st[sp++] = y; }
        if (c[0] >= 0.0f) { g_kind[x] = 0; st[sp++] = x; }
    }
    free(st);
}
static void exp_slot_sim(const pt_scene *sc, v3 o, v3 d) {
    struct { int node; float t0; } st[128];
    int sp = 0;
    float tHit = PT_INFTY, ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
    const int nvt = sc->n_vert * 5;
    unsigned long long nf = 0, nl = 0, fetch = 0, inl = 0;
    int pass; float t0 = exp_box(sc, 0, o, ix, iy, iz, tHit, &pass);
    if (sc->bvh[8] >= 0.0f) { pass = 1; t0 = -PT_INFTY; }
    if (pass) { st[sp].node = 0; st[sp].t0 = t0; sp++; }
    while (sp) {
        sp--;
        int n = st[sp].node;
        if (!(tHit >= st[sp].t0)) continue;
        for (;;) {
            const float *c = sc->bvh + 9 * (size_t)n + 6;
            if (g_kind[n]) inl++; else fetch++;
            if (c[2] < 0.0f) {
                nf++;
                int kid[2] = {(int)c[0], (int)c[1]}, ok[2] = {0, 0}; float kt[2] = {-PT_INFTY, -PT_INFTY};
                for (int k = 0; k < 2; k++) {
                    if (!(c[k] >= 0.0f)) continue;
                    if (sc->bvh[9 * (size_t)kid[k] + 8] >= 0.0f) ok[k] = 1;
                    else kt[k] = exp_box(sc, kid[k], o, ix, iy, iz, tHit, &ok[k]);
                }
                if (ok[0]) { st[sp].node = kid[0]; st[sp].t0 = kt[0]; sp++; }
                if (ok[1]) { n = kid[1]; continue; }  /* first-visited child: no stack round trip */
                break;
            } else {
                nl++;
                int index = (int)c[2]; float tr[4]; fetch4(sc->tri, sc->n_tri, index, tr);
                int i0 = (int)tr[0] * 5, i1 = (int)tr[1] * 5, i2 = (int)tr[2] * 5;
                v3 v0 = fetch3(sc->vert, nvt, i0), v1 = fetch3(sc->vert, nvt, i1), v2 = fetch3(sc->vert, nvt, i2), z = {0, 0, 0}, n3;
                float dist = pt_tri(o, d, v0, v1, v2, z, z, z, 0, &n3);
                tHit = FMIN(tHit, dist);
                break;
            }
        }
    }
    _Pragma("omp atomic") g_exp[8] += nf;
    _Pragma("omp atomic") g_exp[9] += nl;
    _Pragma("omp atomic") g_exp[10] += fetch;
    _Pragma("omp atomic") g_exp[11] += inl;
}
#endif

/* fresnelConductor :158-178; association order as compiled [order] */
INL float pt_fresnel1(float c2, float s2, float cosI, float eta, float k) {
    float eta2 = eta * eta, k2 = k * k;
    float temp0 = (eta2 - s2) - k2;
    float a2pb2 = sqrtf(FMAX_C(temp0 * temp0 + (4.0f * k2) * eta2, 0.0f));
    float temp1 = a2pb2 + c2;
    float a = sqrtf(FMAX_C((a2pb2 + temp0) * 0.5f, 0.0f));
    float temp2 = (2.0f * a) * cosI;
    float Rs2 = (temp1 - temp2) / (temp1 + temp2);
    float temp3 = a2pb2 * c2 + s2 * s2;
    float temp4 = temp2 * s2;
    float Rp2 = (Rs2 * (temp3 - temp4)) / (temp3 + temp4);
    return 0.5f * (Rp2 + Rs2);
}

/* 1/GGX-denominator :180-184 as (PI*ax) * ((ay*l2)*l2) [order]; returns GGX */
INL float pt_ggx(float hx, float hy, float hz, float ax, float ay) {
    float sx = hx / ax, sy = hy / ay;
    float l2 = (hz * hz + sy * sy) + sx * sx;
    return 1.0f / ((PT_PI * ax) * ((ay * l2) * l2));
}


#ifdef PT_ORDERED_EXPERIMENT
#define EXP_CHECK(IS, O, D)                                                                         \
    do {                                                                                            \
        float t2_; int tri2_; unsigned v2_ = 0;                                                     \
        exp_ordered(sc, O, D, &t2_, &tri2_, &v2_);                                                  \
        exp_slot_sim(sc, O, D);                                                                     \
        int bad_ = !((t2_ == (IS).tHit || (t2_ != t2_ && (IS).tHit != (IS).tHit)) && tri2_ == (IS).tri); \
        _Pragma("omp atomic") g_exp[0] += 1;                                                        \
        if (bad_) { _Pragma("omp atomic") g_exp[1] += 1; }                                          \
        _Pragma("omp atomic") g_exp[2] += (IS).visits;                                              \
        _Pragma("omp atomic") g_exp[3] += v2_;                                                      \
        if ((IS).visits > g_exp[4]) g_exp[4] = (IS).visits;                                         \
        if (v2_ > g_exp[5]) g_exp[5] = v2_;                                                         \
        if ((IS).visits > 128) { _Pragma("omp atomic") g_exp[6] += 1; }                             \
        if (v2_ > 128) { _Pragma("omp atomic") g_exp[7] += 1; }                                     \
    } while (0)
#else
#define EXP_CHECK(IS, O, D) ((void)0)
#endif

/* radiance() :409-559 with sampleDirect() :337-403 inlined */
/* ------------------------------------------------------------------ extensions -- PARITY UNPINNED
 * Analytic spheres, dielectric materials and Whitted-style termination do not exist in the reference (triangle meshes only,
 * raytrace.frag:226-257; MTRL_DIELECTRIC declared at :32 and never branched on), so nothing below can be checked against it.
 * This is the CPU statement of the device kernel's extension path (csrc/pt_kernel.hip.h: sphere_t, traverse_ext, the EXT
 * branches of shade_core), operation for operation, so that tests can compare the two bit for bit; it is validated against
 * the pinned triangle path through the tessellation limit (tests/test_ext.py).  Inactive unless pt_oracle_set_ext is called:
 * the pinned restatement above and below is then exactly what it was. */
#define PT_EXT_DIELECTRIC 1
#define PT_EXT_WHITTED 2
static struct { const float *spheres; int n_spheres; int flags; int active; } g_ext = {0, 0, 0, 0};
void pt_oracle_set_ext(const float *spheres5, int n_spheres, int flags) {
    g_ext.spheres = spheres5; g_ext.n_spheres = n_spheres; g_ext.flags = flags;
    g_ext.active = (n_spheres > 0 || flags != 0);
}
INL float pt_sphere_t(const float *sp, v3 o, v3 d) {
    float cx = o.x - sp[0], cy = o.y - sp[1], cz = o.z - sp[2];
    float A = dot3(d.x, d.y, d.z, d.x, d.y, d.z);
    float B = dot3(cx, cy, cz, d.x, d.y, d.z);
    float C = dot3(cx, cy, cz, cx, cy, cz) - sp[3] * sp[3];
    float disc = B * B - A * C;
    if (!(disc >= 0.0f)) return PT_INFTY;
    float sq = sqrtf(disc);
    float t0 = (-B - sq) / A, t1 = (-B + sq) / A;
    float t = t0 > PT_EPS ? t0 : t1;
    return t > PT_EPS ? t : PT_INFTY;
}
/* closest hit over the BVH's triangles and the spheres */
INL void pt_traverse_ext(const pt_scene *sc, v3 o, v3 d, int want_normal, pt_isect *is) {
    pt_traverse(sc, o, d, want_normal, is);
    for (int k = 0; k < g_ext.n_spheres; k++) {
        const float *sp = g_ext.spheres + 5 * k;
        float t = pt_sphere_t(sp, o, d);
        if (t < is->tHit) {
            is->tHit = t; is->hit = 1; is->tri = -2 - k; is->mtrl = (int)sp[4];
            if (want_normal) {
                float qx = (o.x + t * d.x) - sp[0], qy = (o.y + t * d.y) - sp[1], qz = (o.z + t * d.z) - sp[2];
                float r = rsq(dot3(qx, qy, qz, qx, qy, qz));
                is->norm.x = qx * r; is->norm.y = qy * r; is->norm.z = qz * r;
            }
        }
    }
}

INL v3 pt_radiance(const pt_scene *sc, const pt_params *pr, pt_rng *rng, v3 o, v3 d, uint64_t *rays) {
    v3 L = {0.f, 0.f, 0.f}, beta = {1.f, 1.f, 1.f};
    const int n_mat_texels = sc->n_mat * 6, n_vert_texels = sc->n_vert * 5;
    const int nL = sc->n_light;
    const float nLf = (float)nL;

    const int ext = g_ext.active;
    int spec = 0; /* extension: the previous bounce was specular */
    for (int depth = 0; depth < pr->max_depth; depth++) {
        pt_isect is;
        if (ext) pt_traverse_ext(sc, o, d, 1, &is); else pt_traverse(sc, o, d, 1, &is);
        (*rays)++;
        EXP_CHECK(is, o, d);
        const v3 n = is.norm;
        TRACE("depth %d hit %d t %.9g mtrl %d n %.9g %.9g %.9g o %.9g %.9g %.9g d %.9g %.9g %.9g\n", depth, is.hit,
              is.tHit, is.mtrl, n.x, n.y, n.z, o.x, o.y, o.z, d.x, d.y, d.z);

        /* :420-422 */
        float tt = is.tHit + PT_EPS;
        v3 x = {o.x + tt * d.x, o.y + tt * d.y, o.z + tt * d.z};
        int m6 = is.mtrl * 6;
        int type = (int)fetch3(sc->mat, n_mat_texels, m6).x;
        v3 e = fetch3(sc->mat, n_mat_texels, m6 + 1);

        /* dot(-d, n); also woLocal.z */
        float woz = (-(d.z * n.z) - (d.y * n.y)) - (d.x * n.x);

        int spec_out = 0;
        if (type == 5 && woz >= PT_EPS && (!ext || is.hit)) {
            /* :424-487 volume branch compiled out (ENABLE_VOLUME 0): ray unchanged */
        } else if (ext && is.hit && type == 4 && (g_ext.flags & PT_EXT_DIELECTRIC)) {
            /* extension: smooth dielectric (param0 = tint, param1.x = index of refraction), see shade_core<true> */
            if (depth == 0 || spec) { L.x = L.x + beta.x * e.x; L.y = L.y + beta.y * e.y; L.z = L.z + beta.z * e.z; }
            v3 tint = fetch3(sc->mat, n_mat_texels, m6 + 2);
            float ior = fetch3(sc->mat, n_mat_texels, m6 + 3).x;
            float rl = rsq(dot3(d.x, d.y, d.z, d.x, d.y, d.z));
            float ux = d.x * rl, uy = d.y * rl, uz = d.z * rl;
            float ci0 = (-(uz * n.z) - (uy * n.y)) - (ux * n.x);
            int entering = ci0 > 0.0f;
            float fnx = entering ? n.x : -n.x, fny = entering ? n.y : -n.y, fnz = entering ? n.z : -n.z;
            float ci = fabsf(ci0);
            float eta = entering ? 1.0f / ior : ior;
            float k = 1.0f - (eta * eta) * (1.0f - ci * ci);
            float F = 1.0f, ct = 0.0f;
            if (k > 0.0f) {
                ct = sqrtf(k);
                float rs = (eta * ci - ct) / (eta * ci + ct);
                float rp = (ci - eta * ct) / (ci + eta * ct);
                F = 0.5f * (rs * rs + rp * rp);
            }
            float pick = pt_rand(rng);
            float hx = o.x + is.tHit * d.x, hy = o.y + is.tHit * d.y, hz = o.z + is.tHit * d.z;
            if (pick < F) {
                float two = 2.0f * ci;
                d.x = ux + two * fnx; d.y = uy + two * fny; d.z = uz + two * fnz;
                o.x = hx + fnx * (2.0f * PT_EPS); o.y = hy + fny * (2.0f * PT_EPS); o.z = hz + fnz * (2.0f * PT_EPS);
            } else {
                float g = eta * ci - ct;
                d.x = eta * ux + g * fnx; d.y = eta * uy + g * fny; d.z = eta * uz + g * fnz;
                o.x = hx - fnx * (2.0f * PT_EPS); o.y = hy - fny * (2.0f * PT_EPS); o.z = hz - fnz * (2.0f * PT_EPS);
            }
            beta.x = beta.x * tint.x; beta.y = beta.y * tint.y; beta.z = beta.z * tint.z;
            spec_out = 1;
        } else {
            /* :490-499 (specularReflect / passedVolume are never true in the reference; the extension sets the former) */
            if ((depth == 0 || (ext && spec)) && is.hit) {
                L.x = L.x + beta.x * e.x; L.y = L.y + beta.y * e.y; L.z = L.z + beta.z * e.z;
            }
            if (!is.hit) break;

            /* :502-506 local frame; the compiler keeps the select as 0/1 multipliers [order] */
            float B = (0.1f < fabsf(n.x)) ? 1.0f : 0.0f, A = 1.0f - B;
            float ux = B * n.z;
            float nuy = A * n.z; /* = -u.y */
            float uz = A * n.y - B * n.x;
            float vx = n.y * uz + nuy * n.z;
            float vy = n.z * ux - n.x * uz;
            float vz = -(nuy * n.x) - (n.y * ux);
            float wox = (-(d.z * uz) + nuy * d.y) - (d.x * ux);
            float woy = (-(d.z * vz) - (d.y * vy)) - (d.x * vx);

            v3 f = {0.f, 0.f, 0.f};
            float pdf = 1.0f;
            v3 wiL = {0.f, 0.f, 1.0f};
            if (type == 2) {
                /* :511-519 */
                float ra = pt_rand(rng);
                float rb = pt_rand(rng);
                float r1 = PT_2PI * ra;
                float r2s = sqrtf(rb);
                wiL.x = pt_cos(r1) * r2s;
                wiL.y = pt_sin(r1) * r2s;
                wiL.z = sqrtf(1.0f - rb);
                v3 alb = fetch3(sc->mat, n_mat_texels, m6 + 2);
                f.x = alb.x / PT_PI; f.y = alb.y / PT_PI; f.z = alb.z / PT_PI;
                pdf = wiL.z / PT_PI;
            } else if (type == 3) {
                /* :520-532 */
                v3 kap = fetch3(sc->mat, n_mat_texels, m6 + 2);
                v3 eta = fetch3(sc->mat, n_mat_texels, m6 + 3);
                v3 alp = fetch3(sc->mat, n_mat_texels, m6 + 4);
                float ax = alp.x, ay = alp.y;
                float u0 = pt_rand(rng);
                float u1 = pt_rand(rng);
                /* sampleGGXVNDF :195-214 */
                float sx = wox * ax, sy = woy * ay;
                float lw = (woz * woz + sy * sy) + sx * sx; /* |woStretch|^2, reused below */
                float rw = rsq(lw);
                float vhx = sx * rw, vhy = sy * rw, vhz = woz * rw;
                float lensq = vhx * vhx + vhy * vhy;
                float q = rsq(lensq);
                float T1x = (0.0f < lensq) ? -(vhy * q) : 1.0f;
                float T1y = (0.0f < lensq) ? vhx * q : 0.0f;
                float rr = sqrtf(u0);
                float phi = PT_2PI * u1;
                float t1 = rr * pt_cos(phi);
                float t2r = rr * pt_sin(phi);
                float s = 0.5f * (1.0f + vhz);
                float c1 = 1.0f - t1 * t1;
                float t2 = (1.0f - s) * sqrtf(c1) + s * t2r;
                float T2y = vhz * T1x;
                float zq = vhz * T1y; /* = -T2.x */
                float T2z = vhx * T1y - vhy * T1x;
                float nhx = t1 * T1x - zq * t2;
                float nhy = t1 * T1y + t2 * T2y;
                float nhz = t2 * T2z;
                float sq2 = sqrtf(FMAX_C(c1 - t2 * t2, 0.0f));
                nhx = nhx + sq2 * vhx; nhy = nhy + sq2 * vhy; nhz = nhz + sq2 * vhz;
                float nex = nhx * ax, ney = nhy * ay, nez = FMAX_C(nhz, 0.0f);
                float rn = rsq((nez * nez + ney * ney) + nex * nex);
                float whx = nex * rn, why = ney * rn, whz = nez * rn;
                /* wiLocal = 2 dot(wh, wo) wh - wo :527 */
                float dwh = (whz * woz + why * woy) + whx * wox;
                float two = 2.0f * dwh;
                float hx2 = two * whx, hy2 = two * why, hz2 = two * whz; /* = wi + wo [order] */
                wiL.x = hx2 - wox; wiL.y = hy2 - woy; wiL.z = hz2 - woz;
                /* fresnelConductor(wiLocal.z, eta, kappa) :528 */
                float c2 = wiL.z * wiL.z, s2 = 1.0f - c2;
                float Fx = pt_fresnel1(c2, s2, wiL.z, eta.x, kap.x);
                float Fy = pt_fresnel1(c2, s2, wiL.z, eta.y, kap.y);
                float Fz = pt_fresnel1(c2, s2, wiL.z, eta.z, kap.z);
                /* microfacetGGXBRDF(wi, wo, alpha) :186-193; normalize(wi+wo) uses 2 dwh wh [order] */
                float rh = rsq((hz2 * hz2 + hy2 * hy2) + hx2 * hx2);
                float D = pt_ggx(hx2 * rh, hy2 * rh, hz2 * rh, ax, ay);
                float wisx = wiL.x * ax, wisy = wiL.y * ay;
                float len_wi = sqrtf((c2 + wisy * wisy) + wisx * wisx);
                float len_wo = sqrtf(lw);
                float den = 2.0f * (fabsf(woz) * len_wi + fabsf(wiL.z) * len_wo);
                float brdf = D / den;
                f.x = Fx * brdf; f.y = Fy * brdf; f.z = Fz * brdf;
                /* weightedGGXPDF(wi, wo, wh, alpha) :216-219 */
                float D2 = pt_ggx(whx, why, whz, ax, ay);
                float g1 = 0.5f / (len_wo + woz);
                float pn = (g1 * D2) * FMAX_C(dwh, 0.0f);
                float dwi = (wiL.z * whz + wiL.y * why) + wiL.x * whx;
                pdf = pn / FMAX_C(dwi, PT_EPS);
            }

            /* isBlack(f) || pdf == 0 compiled as min(|f|, |pdf|) == 0 [order] :534 */
            {
                float lf = sqrtf((f.z * f.z + f.y * f.y) + f.x * f.x);
                float ap = fabsf(pdf);
                float mn = FMIN(lf, ap);
                if (mn == 0.0f) break;
            }

            /* ---- sampleDirect(x, isect) :337-403 ---- */
            v3 contrib = {0.f, 0.f, 0.f};
            {
                float rl = pt_rand(rng);
                int lid = (int)(rl * nLf);
                if (nL - 1 < lid) lid = nL - 1;
                float lt[4];
                fetch4(sc->light, sc->n_light, lid, lt);
                int i0 = (int)lt[0] * 5, i1 = (int)lt[1] * 5, i2 = (int)lt[2] * 5;
                v3 v0 = fetch3(sc->vert, n_vert_texels, i0), v1 = fetch3(sc->vert, n_vert_texels, i1),
                   v2 = fetch3(sc->vert, n_vert_texels, i2);
                v3 n0 = fetch3(sc->vert, n_vert_texels, i0 + 1), n1 = fetch3(sc->vert, n_vert_texels, i1 + 1),
                   n2 = fetch3(sc->vert, n_vert_texels, i2 + 1);
                float ua = pt_rand(rng);
                float ub = pt_rand(rng);
                if (1.0f < ua + ub) { ua = 1.0f - ua; ub = 1.0f - ub; }
                float w0 = (1.0f - ua) - ub;
                v3 p = {(w0 * v0.x + ua * v1.x) + ub * v2.x, (w0 * v0.y + ua * v1.y) + ub * v2.y,
                        (w0 * v0.z + ua * v1.z) + ub * v2.z};
                v3 nl = {(w0 * n0.x + ua * n1.x) + ub * n2.x, (w0 * n0.y + ua * n1.y) + ub * n2.y,
                         (w0 * n0.z + ua * n1.z) + ub * n2.z};
                /* shadow ray :359-363 (spawnRay :121-123) */
                v3 so = {x.x + n.x * PT_EPS, x.y + n.y * PT_EPS, x.z + n.z * PT_EPS};
                float dvx = p.x - x.x, dvy = p.y - x.y, dvz = p.z - x.z;
                float dd = (dvz * dvz + dvy * dvy) + dvx * dvx;
                float rd = rsq(dd);
                v3 dir = {dvx * rd, dvy * rd, dvz * rd};
                pt_isect sh;
                if (ext) pt_traverse_ext(sc, so, dir, 0, &sh); else pt_traverse(sc, so, dir, 0, &sh);
                (*rays)++;
                EXP_CHECK(sh, so, dir);
                float dist = sqrtf(dd);
                TRACE("  nee lid %d ua %.9g ub %.9g hit %d dist %.9g tS %.9g diff %.9g\n", lid, ua, ub, sh.hit, dist,
                      sh.tHit, fabsf(dist - sh.tHit));
                if (sh.hit && fabsf(dist - sh.tHit) < PT_EPS) {
                    v3 fb = {0.f, 0.f, 0.f};
                    if (type == 2) {
                        fb = fetch3(sc->mat, n_mat_texels, m6 + 2); /* no 1/PI :372 */
                    } else if (type == 3) {
                        v3 alp = fetch3(sc->mat, n_mat_texels, m6 + 4);
                        v3 eta = fetch3(sc->mat, n_mat_texels, m6 + 3);
                        v3 kap = fetch3(sc->mat, n_mat_texels, m6 + 2);
                        float ax = alp.x, ay = alp.y;
                        float cosI = FMAX_C((-(dir.z * n.z) - (dir.y * n.y)) - (dir.x * n.x), 0.0f);
                        float c2 = cosI * cosI, s2 = 1.0f - c2;
                        float Fx = pt_fresnel1(c2, s2, cosI, eta.x, kap.x);
                        float Fy = pt_fresnel1(c2, s2, cosI, eta.y, kap.y);
                        float Fz = pt_fresnel1(c2, s2, cosI, eta.z, kap.z);
                        float wlx = (uz * dir.z - nuy * dir.y) + ux * dir.x;
                        float wly = (vz * dir.z + vy * dir.y) + vx * dir.x;
                        float wlz = (dir.z * n.z + dir.y * n.y) + dir.x * n.x;
                        float hx = wlx + wox, hy = wly + woy, hz = wlz + woz;
                        float rh = rsq((hz * hz + hy * hy) + hx * hx);
                        float D = pt_ggx(hx * rh, hy * rh, hz * rh, ax, ay);
                        float wisx = wlx * ax, wisy = wly * ay;
                        float len_wi = sqrtf((wlz * wlz + wisy * wisy) + wisx * wisx);
                        float wosx = wox * ax, wosy = woy * ay;
                        float len_wo = sqrtf((woz * woz + wosy * wosy) + wosx * wosx);
                        float den = 2.0f * (fabsf(woz) * len_wi + fabsf(wlz) * len_wo);
                        float brdf = D / den;
                        fb.x = Fx * brdf; fb.y = Fy * brdf; fb.z = Fz * brdf;
                    }
                    int lm = (int)lt[3];
                    v3 el = fetch3(sc->mat, n_mat_texels, lm * 6 + 1);
                    float dot0 = (dir.z * n.z + dir.y * n.y) + dir.x * n.x;
                    float dot1 = (-(dir.z * nl.z) - (dir.y * nl.y)) - (dir.x * nl.x);
                    TRACE("  dot0 %.9g dot1 %.9g\n", dot0, dot1);
                    if (0.0f < FMIN(dot0, dot1)) {
                        float e1x = v1.x - v0.x, e1y = v1.y - v0.y, e1z = v1.z - v0.z;
                        float e2x = v2.x - v0.x, e2y = v2.y - v0.y, e2z = v2.z - v0.z;
                        float cx = e1y * e2z - e1z * e2y;
                        float cy = e1z * e2x - e1x * e2z;
                        float cz = e1x * e2y - e1y * e2x;
                        float G = (dot0 * dot1) / dd; /* dist*dist folded to dd [order] */
                        float area = 0.5f * sqrtf((cz * cz + cy * cy) + cx * cx);
                        float lpdf = 1.0f / (area * nLf);
                        contrib.x = ((el.x * fb.x) * G) / lpdf;
                        contrib.y = ((el.y * fb.y) * G) / lpdf;
                        contrib.z = ((el.z * fb.z) * G) / lpdf;
                    }
                }
                /* :539 */
                L.x = L.x + beta.x * contrib.x; L.y = L.y + beta.y * contrib.y; L.z = L.z + beta.z * contrib.z;
                /* :542-544; wi is not renormalised */
                float wix = (ux * wiL.x + vx * wiL.y) + n.x * wiL.z;
                float wiy = (-(nuy * wiL.x) + vy * wiL.y) + n.y * wiL.z;
                float wiz = (uz * wiL.x + vz * wiL.y) + n.z * wiL.z;
                o = so;
                d.x = wix; d.y = wiy; d.z = wiz;
                float cw = FMAX_C((n.z * wiz + n.y * wiy) + n.x * wix, 0.0f);
                beta.x = beta.x * ((f.x * cw) / pdf);
                beta.y = beta.y * ((f.y * cw) / pdf);
                beta.z = beta.z * ((f.z * cw) / pdf);
            }
            if (ext && (g_ext.flags & PT_EXT_WHITTED) && type == 2) break; /* Whitted: direct light only at a diffuse surface */
        }
        spec = spec_out;

        /* Russian roulette :549-555 */
        if (2 < depth) {
            float pm = FMAX(beta.y, beta.z);
            pm = FMAX(beta.x, pm);
            float pq = FMIN_C(pm, 0.95f);
            float rr = pt_rand(rng);
            if (pq < rr) break;
            beta.x = beta.x / pq; beta.y = beta.y / pq; beta.z = beta.z / pq;
        }
    }
    v3 r = {FMIN_C(L.x, 100.0f), FMIN_C(L.y, 100.0f), FMIN_C(L.z, 100.0f)}; /* :558 */
    return r;
}

/* One pixel of main() :565-614.  acc = (L.r, L.g, L.b, count), read-modify-write
 * (replaces the ping-pong FBO pair, window.cpp:214-252). */
INL void pt_pixel(const pt_scene *sc, const pt_params *pr, int px, int py, float *acc, uint64_t *rays) {
    const float W = (float)pr->width, H = (float)pr->height;
    const float fcx = (float)px + 0.5f, fcy = (float)py + 0.5f;
    pt_rng rng;
    rng.x = fcx / W; rng.y = fcy / H; rng.sx = pr->seed[0]; rng.sy = pr->seed[1];
    const float *S = pr->s2c, *C = pr->c2w;
    float Lr = acc[0], Lg = acc[1], Lb = acc[2], cnt = acc[3];
    for (int i = 0; i < pr->n_samples; i++) {
        float r0 = pt_rand(&rng);
        float r1 = pt_rand(&rng);
        float nx = ((fcx + r0) / W) * 2.0f + -1.0f;
        float ny = ((fcy + r1) / H) * 2.0f + -1.0f;
        /* s2c * (nx, ny, 0, 1) as (col0*nx + col3) + col1*ny [order] */
        float tx = (S[0] * nx + S[12]) + S[4] * ny;
        float ty = (S[1] * nx + S[13]) + S[5] * ny;
        float tz = (S[2] * nx + S[14]) + S[6] * ny;
        float tw = (S[3] * nx + S[15]) + S[7] * ny;
        float cx = tx / tw, cy = ty / tw, cz = tz / tw;
        float rn = rsq((cz * cz + cy * cy) + cx * cx);
        float dx = cx * rn, dy = cy * rn, dz = cz * rn;
        float ox = 0.0f, oy = 0.0f;
        if (0.0f < pr->aperture) {
            /* thin lens :589-598 */
            float ra = pt_rand(&rng);
            float rb = pt_rand(&rng);
            float r = sqrtf(ra) * pr->aperture;
            float th = PT_2PI * rb;
            ox = r * pt_cos(th);
            oy = r * pt_sin(th);
            float ft = (-pr->focal) / dz;
            float fx = dx * ft - ox, fy = dy * ft - oy, fz = dz * ft;
            float rf = rsq((fz * fz + fy * fy) + fx * fx);
            dx = fx * rf; dy = fy * rf; dz = fz * rf;
        }
        /* c2w * (ox, oy, 0, 1) as (col0*ox + col3) + col1*oy; divide by w */
        float wx = (C[0] * ox + C[12]) + C[4] * oy;
        float wy = (C[1] * ox + C[13]) + C[5] * oy;
        float wz = (C[2] * ox + C[14]) + C[6] * oy;
        float ww = (C[3] * ox + C[15]) + C[7] * oy;
        v3 o = {wx / ww, wy / ww, wz / ww};
        /* c2w * (d, 0) as (col0*dx + col1*dy) + col2*dz; normalize */
        float ex = (C[0] * dx + C[4] * dy) + C[8] * dz;
        float ey = (C[1] * dx + C[5] * dy) + C[9] * dz;
        float ez = (C[2] * dx + C[6] * dy) + C[10] * dz;
        float re = rsq((ez * ez + ey * ey) + ex * ex);
        v3 d = {ex * re, ey * re, ez * re};
        v3 Ls = pt_radiance(sc, pr, &rng, o, d, rays);
        Lr = Lr + Ls.x; Lg = Lg + Ls.y; Lb = Lb + Ls.z;
        cnt = cnt + 1.0f;
    }
    acc[0] = Lr; acc[1] = Lg; acc[2] = Lb; acc[3] = cnt;
}

__attribute__((target_clones("default", "fma")))
static uint64_t pt_render_rows(const pt_scene *sc, const pt_params *pr, float *accum, size_t pitch_floats,
                               int y0, int y1, int threads) {
    uint64_t total = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
#endif
    for (int y = y0; y < y1; y++) {
        uint64_t rays = 0;
        /* llvmpipe's rasteriser threads run with denormals flushed (MXCSR FTZ|DAZ); per OpenMP thread */
        const unsigned csr = _mm_getcsr();
        _mm_setcsr(csr | 0x8040u);
        for (int x = 0; x < pr->width; x++) pt_pixel(sc, pr, x, y, accum + (size_t)y * pitch_floats + 4 * (size_t)x, &rays);
        _mm_setcsr(csr);
        total += rays;
    }
    (void)threads;
    return total;
}

/* ------------------------------------------------------------------ C entry
 * accum: height rows of pitch_bytes, each width float4 (L.rgb, count); row 0 is
 * gl_FragCoord.y = 0.5 (bottom row, GL convention).  Renders rows [y0, y1).
 * Returns the number of rays traced (executions of intersect(), :276). */
uint64_t pt_oracle_render(const pt_scene *sc, const pt_params *pr, float *accum, size_t pitch_bytes,
                          int y0, int y1, int threads) {
#ifdef PT_ORDERED_EXPERIMENT
    exp_build_rank(sc);
    exp_build_kind(sc);
#endif
    return pt_render_rows(sc, pr, accum, pitch_bytes / sizeof(float), y0, y1, threads);
}

/* RNG probe for tests: n successive rand() values for pixel (px,py). */
void pt_oracle_rand_stream(float W, float H, int px, int py, float sx, float sy, int n, float *out) {
    pt_rng r;
    r.x = ((float)px + 0.5f) / W; r.y = ((float)py + 0.5f) / H; r.sx = sx; r.sy = sy;
    for (int i = 0; i < n; i++) out[i] = pt_rand(&r);
}

void pt_oracle_sincos(const float *x, int n, float *s, float *c) {
    for (int i = 0; i < n; i++) { s[i] = pt_sin(x[i]); c[i] = pt_cos(x[i]); }
}

/* ------------------------------------------------------------------ resolve pass
 * screen.frag:15-25 -- L = rgb / count; L = pow(clamp(L, 0, 1), 1 / u_gamma); out = (L, 1) -- rendered into an RGBA8 colour
 * buffer (window.cpp:297-317) and read back with glReadPixels(GL_RGBA, GL_UNSIGNED_BYTE) (window.cpp:383-388).
 * Restated as llvmpipe (Mesa 23.2.1, LLVM 15) evaluates it, from its compiler's debug output for the reference's
 * unmodified shader and pinned by tests/golden/screen_*.npz:
 *   - rgb / count and 1 / u_gamma are IEEE divisions; clamp = min(max(x, 0), 1) with x86 maxps/minps operand order
 *     (a NaN quotient clamps to 0);
 *   - pow(x, y) = exp2(log2(x) * y), 0 for x == 0;
 *     log2: e = exponent, m = mantissa in [1, 2), t = (m - 1) / (m + 1), z = t * t, minimax polynomial in z evaluated with
 *           FUSED multiply-adds in the grouping below, result = fma(t, P, e);
 *     exp2: argument clamped to [-126.99999, 128], split by floor, minimax polynomial in the fraction with fused
 *           multiply-adds, scaled by 2^floor through the exponent bits;
 *   - float -> unorm8: min(v, 1) * 255, cvtps2dq (round to nearest even), saturating packs; alpha = 255.
 * The reference samples the accumulators through GL_LINEAR samplers at gl_FragCoord / u_windowSize: exactly the texel at
 * power-of-two sizes; at other sizes a neighbour leaks in at the 1e-7 level, which this restatement and
 * the device kernel do not reproduce (fixture screen_npot_* records how many bytes that moves).  The filter also adds the
 * neighbours with weight 0: a NaN or infinite texel poisons the pixels around it (0 * inf), and a count of -0 comes out as +0
 * (v / +0 = +inf where the exact read gives -inf): non-finite texels and negative-zero counts are outside this restatement's
 * domain -- the renderer produces neither (radiance is clamped to 100, counts count up from +0). */
INL float rs_log2(float x) {
    uint32_t i; memcpy(&i, &x, 4);
    const float ef = (float)((int)((i & 0x7f800000u) >> 23) - 127);
    const uint32_t mi = (i & 0x007fffffu) | 0x3f800000u;
    float m; memcpy(&m, &mi, 4);
    const float t = (m - 1.0f) / (m + 1.0f);
    const float z = t * t, z2 = z * z;
    const float a = fmaf(z2, 0x1.a07ab2p-2f, 0x1.27a642p-1f);
    const float b = fmaf(z2, 0x1.9d062cp-2f, 0x1.ec6ff2p-1f);
    const float c = fmaf(z2, a, 0x1.715476p+1f);
    const float d = fmaf(b, z, c);
    float r = fmaf(t, d, ef);
    if (!(x < INFINITY)) r = INFINITY;       /* fcmp uge x, inf (true for NaN) */
    if (!(x < 0.0f) && !(x > 0.0f)) r = -INFINITY; /* fcmp ueq x, 0 */
    if (!(x >= 0.0f)) r = NAN;               /* fcmp ult x, 0 */
    return r;
}
INL float rs_exp2(float t) {
    t = (128.0f < t) ? 128.0f : t;                         /* minps(128, t) */
    t = (-0x1.fbfffep+6f > t) ? -0x1.fbfffep+6f : t;     /* maxps(-126.99999, t) */
    const float fl = floorf(t);
    const float f = t - fl;
    const uint32_t ei = (uint32_t)((int)fl + 127) << 23;
    float scale; memcpy(&scale, &ei, 4);
    const float z = f * f;
    const float a = fmaf(z, 0x1.ec320ap-10f, 0x1.c95446p-5f);
    const float b = fmaf(z, 0x1.26900cp-7f, 0x1.ebd5a8p-3f);
    const float c = fmaf(z, a, 0x1.62e4f6p-1f);
    const float d = fmaf(z, b, 1.0f);
    return scale * fmaf(c, f, d);
}
INL unsigned char rs_channel(float v, float count, float inv_gamma) {
    float L = v / count;
    L = (L > 0.0f) ? L : 0.0f;  /* maxps(L, 0): 0 when L is NaN */
    L = (L < 1.0f) ? L : 1.0f;  /* minps(L, 1) */
    float r = (!(L < 0.0f) && !(L > 0.0f)) ? 0.0f : rs_exp2(rs_log2(L) * inv_gamma);
    r = (1.0f < r) ? 1.0f : r;  /* minps(1, r) */
    long q = lrintf(r * 255.0f); /* cvtps2dq under the default rounding mode */
    return (unsigned char)(q < 0 ? 0 : (q > 255 ? 255 : q));
}

__attribute__((target_clones("default", "fma")))
void pt_oracle_resolve(const float *accum, size_t pitch_bytes, int width, int height, float gamma, int flip_y, unsigned char *out,
                       size_t out_pitch_bytes) {
    /* llvmpipe runs this pass, like every fragment shader, with denormals flushed (MXCSR FTZ | DAZ): a quotient v / count or an exponent 1 / u_gamma below
     * FLT_MIN is 0 there (found in round 4 with counts of 3e38 and gammas of 3e38: pow(x, 0) = 1 against a denormal power; no committed fixture is affected) */
    const unsigned csr = _mm_getcsr();
    _mm_setcsr(csr | 0x8040u);
    const float inv_gamma = 1.0f / gamma;
    for (int y = 0; y < height; y++) {
        const float *row = (const float *)((const char *)accum + (size_t)y * pitch_bytes);
        unsigned char *o = out + (size_t)(flip_y ? height - 1 - y : y) * out_pitch_bytes;
        for (int x = 0; x < width; x++) {
            o[4 * x + 0] = rs_channel(row[4 * x + 0], row[4 * x + 3], inv_gamma);
            o[4 * x + 1] = rs_channel(row[4 * x + 1], row[4 * x + 3], inv_gamma);
            o[4 * x + 2] = rs_channel(row[4 * x + 2], row[4 * x + 3], inv_gamma);
            o[4 * x + 3] = 255;
        }
    }
    _mm_setcsr(csr);
}

#ifdef PT_ORDERED_EXPERIMENT
void pt_oracle_exp_stats(unsigned long long *out) { for (int i = 0; i < 16; i++) out[i] = g_exp[i]; }
#endif

int pt_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
