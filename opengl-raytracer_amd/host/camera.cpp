// camera.cpp -- the camera matrices the reference builds with GLM (GLM is not
// vendored in the reference tree; these are the standard right-handed,
// depth -1..1 definitions):
//   viewM = lookAt(origin, target, up)                       scene.cpp:93
//   projM = perspective(radians(fov), W/H, near, far)        scene.cpp:113
//   c2w = inverse(viewM * modelM), s2c = inverse(projM)      window.cpp:230-233
// Column-major float[16]: m[col*4 + row].
#include <cmath>
#include <cstring>

#include "glrt_host.h"

namespace {
inline void cross(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
inline float dot(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline void normalize(float v[3]) {
    float r = 1.0f / std::sqrt(dot(v, v));
    v[0] *= r; v[1] *= r; v[2] *= r;
}
}  // namespace

extern "C" {

void glrt_look_at(const float eye[3], const float center[3], const float up[3], float out[16]) {
    float f[3] = {center[0] - eye[0], center[1] - eye[1], center[2] - eye[2]};
    normalize(f);
    float s[3];
    cross(f, up, s);
    normalize(s);
    float u[3];
    cross(s, f, u);
    float m[16] = {s[0], u[0], -f[0], 0.f,  //
                   s[1], u[1], -f[1], 0.f,  //
                   s[2], u[2], -f[2], 0.f,  //
                   -dot(s, eye), -dot(u, eye), dot(f, eye), 1.f};
    std::memcpy(out, m, sizeof m);
}

void glrt_perspective(float fovy_deg, float aspect, float z_near, float z_far, float out[16]) {
    const float fovy = fovy_deg * 0.01745329251994329576923690768489f;
    const float t = std::tan(fovy / 2.0f);
    std::memset(out, 0, 16 * sizeof(float));
    out[0] = 1.0f / (aspect * t);
    out[5] = 1.0f / t;
    out[10] = -(z_far + z_near) / (z_far - z_near);
    out[11] = -1.0f;
    out[14] = -(2.0f * z_far * z_near) / (z_far - z_near);
}

void glrt_mat4_mul(const float a[16], const float b[16], float out[16]) {
    float r[16];
    for (int c = 0; c < 4; c++)
        for (int row = 0; row < 4; row++) {
            float acc = 0.f;
            for (int k = 0; k < 4; k++) acc += a[k * 4 + row] * b[c * 4 + k];
            r[c * 4 + row] = acc;
        }
    std::memcpy(out, r, sizeof r);
}

// Cofactor expansion in float, like glm::inverse.
int glrt_mat4_inverse(const float m[16], float out[16]) {
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] +
             m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] -
             m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] +
             m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] -
              m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] -
             m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] +
             m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] -
             m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] +
              m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] +
             m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] -
             m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] +
              m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] -
              m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] -
             m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] +
             m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] -
              m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] +
              m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0.f) return GLRT_HOST_EINVAL;
    det = 1.0f / det;
    for (int i = 0; i < 16; i++) out[i] = inv[i] * det;
    return GLRT_HOST_OK;
}

void glrt_frame_seed(uint32_t frame, float out[2]) {
    double a = 0.137 + 0.6180340 * (double)frame;
    double b = 0.731 + 0.3819660 * (double)frame;
    out[0] = (float)(a - std::floor(a));
    out[1] = (float)(b - std::floor(b));
}

}  // extern "C"
