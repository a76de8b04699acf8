/*
 * glref.c -- headless Mesa llvmpipe GL 4.5 runner (TEST INFRASTRUCTURE, this container only).
 *
 * Purpose: execute GLSL programs -- in particular the reference's *unmodified*
 * src/shaders/raytrace.{vert,frag}, whose text is handed in at run time by
 * oracle/glref.py (read from /root/reference, never copied into this repo) --
 * on the CPU through Mesa's swrast_dri.so without an X server, so that golden
 * float images can be generated for tests/golden/ and the C restatement in
 * oracle/pt_oracle.c can be pinned against the real reference.
 *
 * Recipe follows SURVEY.md Appendix D: dlopen swrast_dri.so, take the
 * DRI_Core + DRI_SWRast extensions, provide a DRI_SWRastLoader with no-op
 * image callbacks, create a GL 4.5 core context, resolve entry points through
 * _glapi_get_proc_address.
 *
 * Mirrors (for the render call) what the reference host does per frame:
 * window.cpp:213-295 (uniforms + bindings + glDrawArrays(GL_TRIANGLES,0,6))
 * and window.cpp:366-381 / framebuffer_object.cpp:40-105 (RGB32F + R32F
 * LINEAR/CLAMP_TO_EDGE attachments).  This file contains no reference code.
 *
 * Built by oracle/Makefile into oracle/_ref/libglref.so (git-ignored).
 * Not thread-safe; one context per process.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <GL/glcorearb.h>
#include <GL/internal/dri_interface.h>

#ifndef __DRI_API_OPENGL_CORE
#define __DRI_API_OPENGL_CORE 3
#endif

static char g_err[4096];
static const __DRIcoreExtension *g_core;
static const __DRIswrastExtension *g_swrast;
static __DRIscreen *g_screen;
static __DRIcontext *g_ctx;
static __DRIdrawable *g_draw;
static void *(*g_getproc)(const char *);
static GLuint g_vao;
#define SCRATCH_UNIT 15

#define GLF(ret, name, ...) static ret (APIENTRYP p_##name)(__VA_ARGS__)
GLF(const GLubyte *, glGetString, GLenum);
GLF(GLenum, glGetError, void);
GLF(GLuint, glCreateShader, GLenum);
GLF(void, glShaderSource, GLuint, GLsizei, const GLchar *const *, const GLint *);
GLF(void, glCompileShader, GLuint);
GLF(void, glGetShaderiv, GLuint, GLenum, GLint *);
GLF(void, glGetShaderInfoLog, GLuint, GLsizei, GLsizei *, GLchar *);
GLF(GLuint, glCreateProgram, void);
GLF(void, glAttachShader, GLuint, GLuint);
GLF(void, glLinkProgram, GLuint);
GLF(void, glGetProgramiv, GLuint, GLenum, GLint *);
GLF(void, glGetProgramInfoLog, GLuint, GLsizei, GLsizei *, GLchar *);
GLF(void, glUseProgram, GLuint);
GLF(GLint, glGetUniformLocation, GLuint, const GLchar *);
GLF(void, glUniform1i, GLint, GLint);
GLF(void, glUniform1f, GLint, GLfloat);
GLF(void, glUniform2f, GLint, GLfloat, GLfloat);
GLF(void, glUniform3f, GLint, GLfloat, GLfloat, GLfloat);
GLF(void, glUniformMatrix4fv, GLint, GLsizei, GLboolean, const GLfloat *);
GLF(void, glGenTextures, GLsizei, GLuint *);
GLF(void, glDeleteTextures, GLsizei, const GLuint *);
GLF(void, glBindTexture, GLenum, GLuint);
GLF(void, glActiveTexture, GLenum);
GLF(void, glTexImage2D, GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void *);
GLF(void, glTexParameteri, GLenum, GLenum, GLint);
GLF(void, glGetTexImage, GLenum, GLint, GLenum, GLenum, void *);
GLF(void, glGenBuffers, GLsizei, GLuint *);
GLF(void, glDeleteBuffers, GLsizei, const GLuint *);
GLF(void, glBindBuffer, GLenum, GLuint);
GLF(void, glBufferData, GLenum, GLsizeiptr, const void *, GLenum);
GLF(void, glTexBuffer, GLenum, GLenum, GLuint);
GLF(void, glGenFramebuffers, GLsizei, GLuint *);
GLF(void, glDeleteFramebuffers, GLsizei, const GLuint *);
GLF(void, glBindFramebuffer, GLenum, GLuint);
GLF(void, glFramebufferTexture2D, GLenum, GLenum, GLenum, GLuint, GLint);
GLF(GLenum, glCheckFramebufferStatus, GLenum);
GLF(void, glDrawBuffers, GLsizei, const GLenum *);
GLF(void, glViewport, GLint, GLint, GLsizei, GLsizei);
GLF(void, glClearColor, GLfloat, GLfloat, GLfloat, GLfloat);
GLF(void, glClear, GLbitfield);
GLF(void, glGenVertexArrays, GLsizei, GLuint *);
GLF(void, glBindVertexArray, GLuint);
GLF(void, glDrawArrays, GLenum, GLint, GLsizei);
GLF(void, glFinish, void);
GLF(void, glPixelStorei, GLenum, GLint);
GLF(void, glDisable, GLenum);
GLF(void, glReadPixels, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, void *);
GLF(void, glReadBuffer, GLenum);

static int load_gl(void) {
#define L(name)                                                  \
    do {                                                         \
        *(void **)(&p_##name) = g_getproc(#name);                \
        if (!p_##name) {                                         \
            snprintf(g_err, sizeof g_err, "missing GL entry %s", #name); \
            return -1;                                           \
        }                                                        \
    } while (0)
    L(glGetString); L(glGetError); L(glCreateShader); L(glShaderSource); L(glCompileShader);
    L(glGetShaderiv); L(glGetShaderInfoLog); L(glCreateProgram); L(glAttachShader); L(glLinkProgram);
    L(glGetProgramiv); L(glGetProgramInfoLog); L(glUseProgram); L(glGetUniformLocation);
    L(glUniform1i); L(glUniform1f); L(glUniform2f); L(glUniform3f); L(glUniformMatrix4fv);
    L(glGenTextures); L(glDeleteTextures); L(glBindTexture); L(glActiveTexture); L(glTexImage2D);
    L(glTexParameteri); L(glGetTexImage); L(glGenBuffers); L(glDeleteBuffers); L(glBindBuffer);
    L(glBufferData); L(glTexBuffer); L(glGenFramebuffers); L(glDeleteFramebuffers); L(glBindFramebuffer);
    L(glFramebufferTexture2D); L(glCheckFramebufferStatus); L(glDrawBuffers); L(glViewport);
    L(glClearColor); L(glClear); L(glGenVertexArrays); L(glBindVertexArray); L(glDrawArrays);
    L(glFinish); L(glPixelStorei); L(glDisable); L(glReadPixels); L(glReadBuffer);
#undef L
    return 0;
}

/* ---- DRI_SWRastLoader: the driver never needs a real window ------------- */
static void ld_getDrawableInfo(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p) {
    (void)d; (void)p; *x = 0; *y = 0; *w = 16; *h = 16;
}
static void ld_putImage(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p) {
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p;
}
static void ld_getImage(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p) {
    (void)d; (void)x; (void)y; (void)p; memset(data, 0, (size_t)w * h * 4);
}
static void ld_putImage2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *p) {
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p;
}
static void ld_getImage2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *p) {
    (void)d; (void)x; (void)y; (void)w; (void)p; memset(data, 0, (size_t)stride * h);
}
static const __DRIswrastLoaderExtension g_loader = {
    .base = {__DRI_SWRAST_LOADER, 3},
    .getDrawableInfo = ld_getDrawableInfo,
    .putImage = ld_putImage,
    .getImage = ld_getImage,
    .putImage2 = ld_putImage2,
    .getImage2 = ld_getImage2,
};
static const __DRIextension *g_loader_exts[] = {&g_loader.base, NULL};

const char *glref_last_error(void) { return g_err; }

/* Returns 0 on success. */
int glref_init(void) {
    if (g_ctx) return 0;
    const char *drv = getenv("GLREF_DRIVER");
    if (!drv) drv = "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so";
    void *api = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!api) { snprintf(g_err, sizeof g_err, "dlopen libglapi: %s", dlerror()); return -1; }
    void *h = dlopen(drv, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { snprintf(g_err, sizeof g_err, "dlopen %s: %s", drv, dlerror()); return -1; }
    const __DRIextension **(*getexts)(void) =
        (const __DRIextension **(*)(void))dlsym(h, "__driDriverGetExtensions_swrast");
    if (!getexts) { snprintf(g_err, sizeof g_err, "no __driDriverGetExtensions_swrast"); return -1; }
    g_getproc = (void *(*)(const char *))dlsym(api, "_glapi_get_proc_address");
    if (!g_getproc) { snprintf(g_err, sizeof g_err, "no _glapi_get_proc_address"); return -1; }

    const __DRIextension **exts = getexts();
    for (int i = 0; exts[i]; i++) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) g_core = (const __DRIcoreExtension *)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) g_swrast = (const __DRIswrastExtension *)exts[i];
    }
    if (!g_core || !g_swrast || g_swrast->base.version < 4) {
        snprintf(g_err, sizeof g_err, "DRI_Core/DRI_SWRast(v4) not found");
        return -1;
    }
    const __DRIconfig **configs = NULL;
    g_screen = g_swrast->createNewScreen2(0, g_loader_exts, exts, &configs, NULL);
    if (!g_screen || !configs || !configs[0]) { snprintf(g_err, sizeof g_err, "createNewScreen2 failed"); return -1; }
    uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 5};
    unsigned err = 0;
    g_ctx = g_swrast->createContextAttribs(g_screen, __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &err, NULL);
    if (!g_ctx) { snprintf(g_err, sizeof g_err, "createContextAttribs failed (err %u)", err); return -1; }
    g_draw = g_swrast->createNewDrawable(g_screen, configs[0], NULL);
    if (!g_draw) { snprintf(g_err, sizeof g_err, "createNewDrawable failed"); return -1; }
    if (!g_core->bindContext(g_ctx, g_draw, g_draw)) { snprintf(g_err, sizeof g_err, "bindContext failed"); return -1; }
    if (load_gl()) return -1;
    p_glGenVertexArrays(1, &g_vao);
    p_glBindVertexArray(g_vao);
    p_glPixelStorei(GL_PACK_ALIGNMENT, 1);
    p_glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    p_glDisable(GL_DEPTH_TEST);
    p_glDisable(GL_BLEND);
    return 0;
}

/* which: 0 renderer, 1 version, 2 GLSL version, 3 vendor */
const char *glref_string(int which) {
    static const GLenum e[] = {GL_RENDERER, GL_VERSION, GL_SHADING_LANGUAGE_VERSION, GL_VENDOR};
    if (!g_ctx || which < 0 || which > 3) return "";
    return (const char *)p_glGetString(e[which]);
}

static GLuint compile(GLenum type, const char *src) {
    GLuint s = p_glCreateShader(type);
    p_glShaderSource(s, 1, &src, NULL);
    p_glCompileShader(s);
    GLint ok = 0;
    p_glGetShaderiv(s, GL_COMPILE_STATUS, &ok);
    if (!ok) {
        size_t n = (size_t)snprintf(g_err, sizeof g_err, "%s shader compile failed:\n",
                                    type == GL_VERTEX_SHADER ? "vertex" : "fragment");
        p_glGetShaderInfoLog(s, (GLsizei)(sizeof g_err - n - 1), NULL, g_err + n);
        return 0;
    }
    return s;
}

/* Returns program id (>0) or 0 on failure. */
unsigned glref_program(const char *vs_src, const char *fs_src) {
    GLuint vs = compile(GL_VERTEX_SHADER, vs_src);
    if (!vs) return 0;
    GLuint fs = compile(GL_FRAGMENT_SHADER, fs_src);
    if (!fs) return 0;
    GLuint p = p_glCreateProgram();
    p_glAttachShader(p, vs);
    p_glAttachShader(p, fs);
    p_glLinkProgram(p);
    GLint ok = 0;
    p_glGetProgramiv(p, GL_LINK_STATUS, &ok);
    if (!ok) {
        size_t n = (size_t)snprintf(g_err, sizeof g_err, "link failed:\n");
        p_glGetProgramInfoLog(p, (GLsizei)(sizeof g_err - n - 1), NULL, g_err + n);
        return 0;
    }
    return p;
}

void glref_use(unsigned prog) { p_glUseProgram(prog); }
int glref_uniform1i(unsigned prog, const char *name, int v) {
    GLint l = p_glGetUniformLocation(prog, name); if (l < 0) return -1; p_glUniform1i(l, v); return 0;
}
int glref_uniform1f(unsigned prog, const char *name, float v) {
    GLint l = p_glGetUniformLocation(prog, name); if (l < 0) return -1; p_glUniform1f(l, v); return 0;
}
int glref_uniform2f(unsigned prog, const char *name, float a, float b) {
    GLint l = p_glGetUniformLocation(prog, name); if (l < 0) return -1; p_glUniform2f(l, a, b); return 0;
}
int glref_uniform3f(unsigned prog, const char *name, float a, float b, float c) {
    GLint l = p_glGetUniformLocation(prog, name); if (l < 0) return -1; p_glUniform3f(l, a, b, c); return 0;
}
/* column-major, untransposed: same as shader_program.cpp:151-156 */
int glref_uniform_mat4(unsigned prog, const char *name, const float *m) {
    GLint l = p_glGetUniformLocation(prog, name); if (l < 0) return -1;
    p_glUniformMatrix4fv(l, 1, GL_FALSE, m); return 0;
}

/* Buffer texture; comps = 1 (R32F), 3 (RGB32F) or 4 (RGBA32F). out[0]=texture, out[1]=buffer. */
int glref_tbo(const float *data, size_t nbytes, int comps, unsigned *out) {
    GLenum fmt = comps == 1 ? GL_R32F : comps == 3 ? GL_RGB32F : GL_RGBA32F;
    GLuint buf, tex;
    static const float zero[4] = {0, 0, 0, 0};
    p_glGenBuffers(1, &buf);
    p_glBindBuffer(GL_TEXTURE_BUFFER, buf);
    if (nbytes == 0) { data = zero; nbytes = sizeof(float) * (size_t)comps; }
    p_glBufferData(GL_TEXTURE_BUFFER, (GLsizeiptr)nbytes, data, GL_STATIC_DRAW);
    p_glGenTextures(1, &tex);
    p_glActiveTexture(GL_TEXTURE0 + SCRATCH_UNIT); /* never disturb the units 0..8 the shader samples */
    p_glBindTexture(GL_TEXTURE_BUFFER, tex);
    p_glTexBuffer(GL_TEXTURE_BUFFER, fmt, buf);
    p_glBindTexture(GL_TEXTURE_BUFFER, 0);
    p_glBindBuffer(GL_TEXTURE_BUFFER, 0);
    out[0] = tex; out[1] = buf;
    return p_glGetError() == GL_NO_ERROR ? 0 : -1;
}
void glref_tbo_free(unsigned tex, unsigned buf) { p_glDeleteTextures(1, &tex); p_glDeleteBuffers(1, &buf); }
void glref_bind_tbo(int unit, unsigned tex) {
    p_glActiveTexture(GL_TEXTURE0 + unit); p_glBindTexture(GL_TEXTURE_BUFFER, tex);
}

/* 2D float texture, LINEAR + CLAMP_TO_EDGE like framebuffer_object.cpp:57-60. comps in {1,3,4}. */
unsigned glref_tex2d(int w, int h, int comps, const float *init) {
    GLenum ifmt = comps == 1 ? GL_R32F : comps == 3 ? GL_RGB32F : GL_RGBA32F;
    GLenum fmt = comps == 1 ? GL_RED : comps == 3 ? GL_RGB : GL_RGBA;
    GLuint tex;
    p_glGenTextures(1, &tex);
    p_glActiveTexture(GL_TEXTURE0 + SCRATCH_UNIT);
    p_glBindTexture(GL_TEXTURE_2D, tex);
    p_glTexImage2D(GL_TEXTURE_2D, 0, (GLint)ifmt, w, h, 0, fmt, GL_FLOAT, init);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    p_glBindTexture(GL_TEXTURE_2D, 0);
    return tex;
}
void glref_tex_free(unsigned tex) { p_glDeleteTextures(1, &tex); }
void glref_bind_tex2d(int unit, unsigned tex) {
    p_glActiveTexture(GL_TEXTURE0 + unit); p_glBindTexture(GL_TEXTURE_2D, tex);
}
/* comps selects the read-back format (GL_RED / GL_RGB / GL_RGBA), GL_FLOAT. */
/* RGBA8 (unorm) colour target: what the reference's screen pass renders into -- the window's default framebuffer
 * (window.cpp:297-317) -- and what saveCurrentFrame reads back with glReadPixels(GL_RGBA, GL_UNSIGNED_BYTE) (:383-388). */
unsigned glref_tex2d_rgba8(int w, int h) {
    GLuint tex;
    p_glGenTextures(1, &tex);
    p_glActiveTexture(GL_TEXTURE0 + SCRATCH_UNIT);
    p_glBindTexture(GL_TEXTURE_2D, tex);
    p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA8, w, h, 0, GL_RGBA, GL_UNSIGNED_BYTE, NULL);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    p_glBindTexture(GL_TEXTURE_2D, 0);
    return tex;
}
/* glReadPixels(0, 0, w, h, GL_RGBA, GL_UNSIGNED_BYTE) of colour attachment 0 of `fbo` (row 0 = bottom row). */
int glref_read_pixels_rgba8(unsigned fbo, int w, int h, unsigned char *out) {
    p_glBindFramebuffer(GL_FRAMEBUFFER, fbo);
    p_glReadBuffer(GL_COLOR_ATTACHMENT0);
    p_glPixelStorei(GL_PACK_ALIGNMENT, 1);
    p_glReadPixels(0, 0, w, h, GL_RGBA, GL_UNSIGNED_BYTE, out);
    p_glBindFramebuffer(GL_FRAMEBUFFER, 0);
    GLenum e = p_glGetError();
    if (e != GL_NO_ERROR) { snprintf(g_err, sizeof g_err, "GL error 0x%x after glReadPixels", e); return -1; }
    return 0;
}
void glref_read_tex2d(unsigned tex, int comps, float *out) {
    GLenum fmt = comps == 1 ? GL_RED : comps == 3 ? GL_RGB : GL_RGBA;
    p_glActiveTexture(GL_TEXTURE0 + SCRATCH_UNIT);
    p_glBindTexture(GL_TEXTURE_2D, tex);
    p_glGetTexImage(GL_TEXTURE_2D, 0, fmt, GL_FLOAT, out);
    p_glBindTexture(GL_TEXTURE_2D, 0);
}

/* FBO with n colour attachments (textures made by glref_tex2d). Returns fbo id or 0. */
unsigned glref_fbo(int n, const unsigned *texs) {
    GLuint f;
    GLenum bufs[8];
    if (n > 8) n = 8;
    p_glGenFramebuffers(1, &f);
    p_glBindFramebuffer(GL_FRAMEBUFFER, f);
    for (int i = 0; i < n; i++) {
        p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0 + i, GL_TEXTURE_2D, texs[i], 0);
        bufs[i] = GL_COLOR_ATTACHMENT0 + i;
    }
    p_glDrawBuffers(n, bufs);
    GLenum st = p_glCheckFramebufferStatus(GL_FRAMEBUFFER);
    p_glBindFramebuffer(GL_FRAMEBUFFER, 0);
    if (st != GL_FRAMEBUFFER_COMPLETE) {
        snprintf(g_err, sizeof g_err, "FBO incomplete: 0x%x", st);
        return 0;
    }
    return f;
}
void glref_fbo_free(unsigned f) { p_glDeleteFramebuffers(1, &f); }

/* Clear (if do_clear) and draw the 6-vertex fullscreen pass into fbo at w x h; glFinish. */
int glref_draw(unsigned fbo, int w, int h, int do_clear) {
    p_glBindFramebuffer(GL_FRAMEBUFFER, fbo);
    p_glViewport(0, 0, w, h);
    if (do_clear) {
        p_glClearColor(0.f, 0.f, 0.f, 0.f);
        p_glClear(GL_COLOR_BUFFER_BIT);
    }
    p_glBindVertexArray(g_vao);
    p_glDrawArrays(GL_TRIANGLES, 0, 6);
    p_glFinish();
    p_glBindFramebuffer(GL_FRAMEBUFFER, 0);
    GLenum e = p_glGetError();
    if (e != GL_NO_ERROR) { snprintf(g_err, sizeof g_err, "GL error 0x%x after draw", e); return -1; }
    return 0;
}
