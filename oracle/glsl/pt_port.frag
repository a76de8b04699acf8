#version 410
// pt_port.frag -- this repository's OWN GLSL statement of the path tracer that oracle/pt_oracle.c restates in C.
//
// TEST INFRASTRUCTURE / CPU BASELINE ONLY.  Written from oracle/pt_oracle.c (operation for operation, in that file's
// evaluation order), not from the reference's shader file, which never leaves /root/reference.  Its purpose: the GPU box has
// Mesa's llvmpipe but not the reference checkout, so the "reference timed through llvmpipe on the box's own host cores"
// that BASELINE.json asks for is THIS shader run through oracle/glref (bench.py: cpu_baseline.llvmpipe).  In the build
// container tests/test_glsl_port.py checks it, on llvmpipe, against the golden images the reference's unmodified shader
// produced there (tests/golden/*.npz).
//
// Inputs are the reference's five flat buffers in their wire format (SURVEY.md Appendix B) as texture buffers, and the
// per-frame uniforms of Window::render (window.cpp:230-243).  Output 0 = previous sum + this frame's radiance, output 1 =
// previous count + samples (the two accumulation targets, window.cpp:366-381), previous values fetched exactly (texelFetch).

uniform samplerBuffer vertTex;   // RGB32F, 5 texels per vertex: position, normal, (uv, tangent, binormal unused)
uniform samplerBuffer triTex;    // RGBA32F: i, j, k, material
uniform samplerBuffer matTex;    // RGB32F, 6 texels per material
uniform samplerBuffer lightTex;  // RGBA32F: emitting triangles
uniform samplerBuffer nodeTex;   // RGB32F, 3 texels per node: box min, box max, (left, right, -1) | (-1, -1, triangle)
uniform sampler2D prevSum;
uniform sampler2D prevCount;
uniform mat4 camToWorld;
uniform mat4 screenToCam;
uniform float lensRadius;
uniform float focusDist;
uniform vec2 frameSeed;
uniform vec2 imageSize;
uniform int samplesPerPass;
uniform int depthLimit;
uniform int lightCount;

layout(location = 0) out vec4 outSum;
layout(location = 1) out vec4 outCount;

const float TINY = 1.0e-4;
const float FAR = 1.0e8;
const float PI_F = 3.14159274101257324;
const float TWO_PI_F = 6.28318548202514648;

vec2 rngState;

float nextRandom() {
    float dy = (rngState.y - frameSeed.y) * 78.233;
    float t = dy + (rngState.x - frameSeed.x) * 12.9898;
    rngState.x = fract(sin(t) * 43758.5453);
    t = dy + (rngState.x - frameSeed.x) * 12.9898;
    rngState.y = fract(sin(t) * 43758.5453);
    return rngState.x;
}

float dot3(vec3 a, vec3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }
float rsq(float x) { return 1.0 / sqrt(x); }

// closest hit state
float bestT;
vec3 bestN;
int bestMat;
bool bestHit;

float triangleT(vec3 o, vec3 d, vec3 v0, vec3 v1, vec3 v2, vec3 n0, vec3 n1, vec3 n2, bool wantNormal, inout vec3 nOut) {
    vec3 e1 = v1 - v0;
    vec3 e2 = v2 - v0;
    vec3 p = vec3(d.y * e2.z - d.z * e2.y, d.z * e2.x - d.x * e2.z, d.x * e2.y - d.y * e2.x);
    float det = dot3(e1, p);
    if (-TINY < det && det < TINY) return FAR;
    float inv = 1.0 / det;
    vec3 tv = o - v0;
    float U = dot3(tv, p);
    float u = U * inv;
    if (u < 0.0 || 1.0 < u) return FAR;
    vec3 q = vec3(tv.y * e1.z - tv.z * e1.y, tv.z * e1.x - tv.x * e1.z, tv.x * e1.y - tv.y * e1.x);
    float V = dot3(d, q);
    float v = V * inv;
    if (v < 0.0 || 1.0 < inv * (U + V)) return FAR;
    float t = dot3(e2, q) * inv;
    if (TINY >= t) return FAR;
    if (wantNormal) {
        float w0 = (1.0 - u) - v;
        vec3 n = vec3((w0 * n0.x + u * n1.x) + v * n2.x, (w0 * n0.y + u * n1.y) + v * n2.y, (w0 * n0.z + u * n1.z) + v * n2.z);
        nOut = n * rsq(dot3(n, n));
    }
    return t;
}

void closestHit(vec3 o, vec3 d, bool wantNormal) {
    int stack[64];
    int pos = 0;
    stack[0] = 0;
    bestT = FAR; bestN = vec3(0.0); bestMat = 0; bestHit = false;
    vec3 inv = vec3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    while (pos >= 0) {
        int slot = pos;
        int node = stack[slot];
        pos -= 1;
        vec3 bmin = texelFetch(nodeTex, node * 3).xyz;
        vec3 bmax = texelFetch(nodeTex, node * 3 + 1).xyz;
        vec3 ch = texelFetch(nodeTex, node * 3 + 2).xyz;
        if (ch.z < 0.0) {
            vec3 f = (bmax - o) * inv;
            vec3 n = (bmin - o) * inv;
            vec3 hi = max(f, n);
            vec3 lo = min(f, n);
            float t1 = min(hi.x, min(hi.y, hi.z));
            float t0 = max(lo.x, max(lo.y, lo.z));
            if (min(t1, bestT) >= t0) {
                if (ch.x >= 0.0) { stack[slot & 63] = int(ch.x); pos = slot; }
                if (ch.y >= 0.0) { pos += 1; stack[pos & 63] = int(ch.y); }
            }
        } else {
            int index = int(ch.z);
            vec4 tr = texelFetch(triTex, index);
            int i0 = int(tr.x) * 5, i1 = int(tr.y) * 5, i2 = int(tr.z) * 5;
            vec3 v0 = texelFetch(vertTex, i0).xyz, v1 = texelFetch(vertTex, i1).xyz, v2 = texelFetch(vertTex, i2).xyz;
            vec3 n0 = vec3(0.0), n1 = vec3(0.0), n2 = vec3(0.0);
            if (wantNormal) {
                n0 = texelFetch(vertTex, i0 + 1).xyz; n1 = texelFetch(vertTex, i1 + 1).xyz; n2 = texelFetch(vertTex, i2 + 1).xyz;
            }
            vec3 nn = bestN;
            float t = triangleT(o, d, v0, v1, v2, n0, n1, n2, wantNormal, nn);
            if (t < bestT) { bestN = nn; bestMat = int(tr.w); bestHit = true; }
            bestT = min(bestT, t);
        }
    }
}

float conductorFresnel(float c2, float s2, float cosI, float eta, float k) {
    float eta2 = eta * eta, k2 = k * k;
    float t0 = (eta2 - s2) - k2;
    float a2pb2 = sqrt(max(t0 * t0 + (4.0 * k2) * eta2, 0.0));
    float t1 = a2pb2 + c2;
    float a = sqrt(max((a2pb2 + t0) * 0.5, 0.0));
    float t2 = (2.0 * a) * cosI;
    float Rs2 = (t1 - t2) / (t1 + t2);
    float t3 = a2pb2 * c2 + s2 * s2;
    float t4 = t2 * s2;
    float Rp2 = (Rs2 * (t3 - t4)) / (t3 + t4);
    return 0.5 * (Rp2 + Rs2);
}

float ggxD(vec3 h, float ax, float ay) {
    float sx = h.x / ax, sy = h.y / ay;
    float l2 = (h.z * h.z + sy * sy) + sx * sx;
    return 1.0 / ((PI_F * ax) * ((ay * l2) * l2));
}

vec3 pathRadiance(vec3 o, vec3 d) {
    vec3 L = vec3(0.0), beta = vec3(1.0);
    float nLf = float(lightCount);
    for (int depth = 0; depth < depthLimit; depth++) {
        closestHit(o, d, true);
        vec3 n = bestN;
        bool hit = bestHit;
        float tt = bestT + TINY;
        vec3 x = o + tt * d;
        int m6 = bestMat * 6;
        int type = int(texelFetch(matTex, m6).x);
        vec3 e = texelFetch(matTex, m6 + 1).xyz;
        float woz = (-(d.z * n.z) - (d.y * n.y)) - (d.x * n.x);
        if (type == 5 && woz >= TINY) {
            // participating medium entered from the front: the volume integrator is compiled out, the ray goes on unchanged
        } else {
            if (depth == 0 && hit) L = L + beta * e;
            if (!hit) break;
            float B = (0.1 < abs(n.x)) ? 1.0 : 0.0, A = 1.0 - B;
            float ux = B * n.z;
            float nuy = A * n.z;
            float uz = A * n.y - B * n.x;
            float vx = n.y * uz + nuy * n.z;
            float vy = n.z * ux - n.x * uz;
            float vz = -(nuy * n.x) - (n.y * ux);
            float wox = (-(d.z * uz) + nuy * d.y) - (d.x * ux);
            float woy = (-(d.z * vz) - (d.y * vy)) - (d.x * vx);
            vec3 f = vec3(0.0);
            float pdf = 1.0;
            vec3 wiL = vec3(0.0, 0.0, 1.0);
            if (type == 2) {
                float ra = nextRandom();
                float rb = nextRandom();
                float r1 = TWO_PI_F * ra;
                float r2s = sqrt(rb);
                wiL = vec3(cos(r1) * r2s, sin(r1) * r2s, sqrt(1.0 - rb));
                f = texelFetch(matTex, m6 + 2).xyz / PI_F;
                pdf = wiL.z / PI_F;
            } else if (type == 3) {
                vec3 kap = texelFetch(matTex, m6 + 2).xyz;
                vec3 eta = texelFetch(matTex, m6 + 3).xyz;
                vec3 alp = texelFetch(matTex, m6 + 4).xyz;
                float ax = alp.x, ay = alp.y;
                float u0 = nextRandom();
                float u1 = nextRandom();
                float sx = wox * ax, sy = woy * ay;
                float lw = (woz * woz + sy * sy) + sx * sx;
                float rw = rsq(lw);
                vec3 vh = vec3(sx * rw, sy * rw, woz * rw);
                float lensq = vh.x * vh.x + vh.y * vh.y;
                float q = rsq(lensq);
                float T1x = (0.0 < lensq) ? -(vh.y * q) : 1.0;
                float T1y = (0.0 < lensq) ? vh.x * q : 0.0;
                float rr = sqrt(u0);
                float phi = TWO_PI_F * u1;
                float t1 = rr * cos(phi);
                float t2r = rr * sin(phi);
                float s = 0.5 * (1.0 + vh.z);
                float c1 = 1.0 - t1 * t1;
                float t2 = (1.0 - s) * sqrt(c1) + s * t2r;
                float T2y = vh.z * T1x;
                float zq = vh.z * T1y;
                float T2z = vh.x * T1y - vh.y * T1x;
                vec3 nh = vec3(t1 * T1x - zq * t2, t1 * T1y + t2 * T2y, t2 * T2z);
                float sq2 = sqrt(max(c1 - t2 * t2, 0.0));
                nh = nh + sq2 * vh;
                vec3 ne = vec3(nh.x * ax, nh.y * ay, max(nh.z, 0.0));
                float rn = rsq((ne.z * ne.z + ne.y * ne.y) + ne.x * ne.x);
                vec3 wh = ne * rn;
                float dwh = (wh.z * woz + wh.y * woy) + wh.x * wox;
                float two = 2.0 * dwh;
                vec3 h2 = two * wh;
                wiL = vec3(h2.x - wox, h2.y - woy, h2.z - woz);
                float c2 = wiL.z * wiL.z, s2 = 1.0 - c2;
                vec3 F = vec3(conductorFresnel(c2, s2, wiL.z, eta.x, kap.x), conductorFresnel(c2, s2, wiL.z, eta.y, kap.y),
                              conductorFresnel(c2, s2, wiL.z, eta.z, kap.z));
                float rh = rsq((h2.z * h2.z + h2.y * h2.y) + h2.x * h2.x);
                float D = ggxD(h2 * rh, ax, ay);
                float wisx = wiL.x * ax, wisy = wiL.y * ay;
                float lenWi = sqrt((c2 + wisy * wisy) + wisx * wisx);
                float lenWo = sqrt(lw);
                float den = 2.0 * (abs(woz) * lenWi + abs(wiL.z) * lenWo);
                float brdf = D / den;
                f = F * brdf;
                float D2 = ggxD(wh, ax, ay);
                float g1 = 0.5 / (lenWo + woz);
                float pn = (g1 * D2) * max(dwh, 0.0);
                float dwi = (wiL.z * wh.z + wiL.y * wh.y) + wiL.x * wh.x;
                pdf = pn / max(dwi, TINY);
            }
            {
                float lf = sqrt((f.z * f.z + f.y * f.y) + f.x * f.x);
                if (min(lf, abs(pdf)) == 0.0) break;
            }
            // next-event estimation: one light triangle, one point on it, one shadow ray
            vec3 contrib = vec3(0.0);
            float rl = nextRandom();
            int lid = int(rl * nLf);
            if (lightCount - 1 < lid) lid = lightCount - 1;
            vec4 lt = texelFetch(lightTex, lid);
            int i0 = int(lt.x) * 5, i1 = int(lt.y) * 5, i2 = int(lt.z) * 5;
            vec3 v0 = texelFetch(vertTex, i0).xyz, v1 = texelFetch(vertTex, i1).xyz, v2 = texelFetch(vertTex, i2).xyz;
            vec3 n0 = texelFetch(vertTex, i0 + 1).xyz, n1 = texelFetch(vertTex, i1 + 1).xyz, n2 = texelFetch(vertTex, i2 + 1).xyz;
            float ua = nextRandom();
            float ub = nextRandom();
            if (1.0 < ua + ub) { ua = 1.0 - ua; ub = 1.0 - ub; }
            float w0 = (1.0 - ua) - ub;
            vec3 p = vec3((w0 * v0.x + ua * v1.x) + ub * v2.x, (w0 * v0.y + ua * v1.y) + ub * v2.y, (w0 * v0.z + ua * v1.z) + ub * v2.z);
            vec3 nl = vec3((w0 * n0.x + ua * n1.x) + ub * n2.x, (w0 * n0.y + ua * n1.y) + ub * n2.y, (w0 * n0.z + ua * n1.z) + ub * n2.z);
            vec3 so = x + n * TINY;
            vec3 dv = p - x;
            float dd = (dv.z * dv.z + dv.y * dv.y) + dv.x * dv.x;
            vec3 dir = dv * rsq(dd);
            closestHit(so, dir, false);
            float dist = sqrt(dd);
            if (bestHit && abs(dist - bestT) < TINY) {
                vec3 fb = vec3(0.0);
                if (type == 2) {
                    fb = texelFetch(matTex, m6 + 2).xyz;
                } else if (type == 3) {
                    vec3 alp = texelFetch(matTex, m6 + 4).xyz;
                    vec3 eta = texelFetch(matTex, m6 + 3).xyz;
                    vec3 kap = texelFetch(matTex, m6 + 2).xyz;
                    float ax = alp.x, ay = alp.y;
                    float cosI = max((-(dir.z * n.z) - (dir.y * n.y)) - (dir.x * n.x), 0.0);
                    float c2 = cosI * cosI, s2 = 1.0 - c2;
                    vec3 F = vec3(conductorFresnel(c2, s2, cosI, eta.x, kap.x), conductorFresnel(c2, s2, cosI, eta.y, kap.y),
                                  conductorFresnel(c2, s2, cosI, eta.z, kap.z));
                    float wlx = (uz * dir.z - nuy * dir.y) + ux * dir.x;
                    float wly = (vz * dir.z + vy * dir.y) + vx * dir.x;
                    float wlz = (dir.z * n.z + dir.y * n.y) + dir.x * n.x;
                    vec3 h = vec3(wlx + wox, wly + woy, wlz + woz);
                    float rh = rsq((h.z * h.z + h.y * h.y) + h.x * h.x);
                    float D = ggxD(h * rh, ax, ay);
                    float wisx = wlx * ax, wisy = wly * ay;
                    float lenWi = sqrt((wlz * wlz + wisy * wisy) + wisx * wisx);
                    float wosx = wox * ax, wosy = woy * ay;
                    float lenWo = sqrt((woz * woz + wosy * wosy) + wosx * wosx);
                    float den = 2.0 * (abs(woz) * lenWi + abs(wlz) * lenWo);
                    fb = F * (D / den);
                }
                vec3 el = texelFetch(matTex, int(lt.w) * 6 + 1).xyz;
                float dot0 = (dir.z * n.z + dir.y * n.y) + dir.x * n.x;
                float dot1 = (-(dir.z * nl.z) - (dir.y * nl.y)) - (dir.x * nl.x);
                if (0.0 < min(dot0, dot1)) {
                    vec3 e1 = v1 - v0, e2 = v2 - v0;
                    vec3 c = vec3(e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x);
                    float G = (dot0 * dot1) / dd;
                    float area = 0.5 * sqrt((c.z * c.z + c.y * c.y) + c.x * c.x);
                    float lpdf = 1.0 / (area * nLf);
                    contrib = ((el * fb) * G) / lpdf;
                }
            }
            L = L + beta * contrib;
            vec3 wi = vec3((ux * wiL.x + vx * wiL.y) + n.x * wiL.z, (-(nuy * wiL.x) + vy * wiL.y) + n.y * wiL.z,
                           (uz * wiL.x + vz * wiL.y) + n.z * wiL.z);
            o = so;
            d = wi;
            float cw = max((n.z * wi.z + n.y * wi.y) + n.x * wi.x, 0.0);
            beta = beta * ((f * cw) / pdf);
        }
        if (2 < depth) {
            float pm = max(beta.x, max(beta.y, beta.z));
            float pq = min(pm, 0.95);
            float rr = nextRandom();
            if (pq < rr) break;
            beta = beta / pq;
        }
    }
    return min(L, vec3(100.0));
}

void main() {
    vec2 fc = gl_FragCoord.xy;
    rngState = fc / imageSize;
    ivec2 px = ivec2(fc);
    vec3 sum = texelFetch(prevSum, px, 0).rgb;
    float cnt = texelFetch(prevCount, px, 0).r;
    for (int i = 0; i < samplesPerPass; i++) {
        float r0 = nextRandom();
        float r1 = nextRandom();
        float nx = ((fc.x + r0) / imageSize.x) * 2.0 + -1.0;
        float ny = ((fc.y + r1) / imageSize.y) * 2.0 + -1.0;
        vec4 t = (screenToCam[0] * nx + screenToCam[3]) + screenToCam[1] * ny;
        vec3 c = vec3(t.x / t.w, t.y / t.w, t.z / t.w);
        vec3 dc = c * rsq((c.z * c.z + c.y * c.y) + c.x * c.x);
        float ox = 0.0, oy = 0.0;
        if (0.0 < lensRadius) {
            float ra = nextRandom();
            float rb = nextRandom();
            float r = sqrt(ra) * lensRadius;
            float th = TWO_PI_F * rb;
            ox = r * cos(th);
            oy = r * sin(th);
            float ft = (-focusDist) / dc.z;
            vec3 fp = vec3(dc.x * ft - ox, dc.y * ft - oy, dc.z * ft);
            dc = fp * rsq((fp.z * fp.z + fp.y * fp.y) + fp.x * fp.x);
        }
        vec4 w = (camToWorld[0] * ox + camToWorld[3]) + camToWorld[1] * oy;
        vec3 o = vec3(w.x / w.w, w.y / w.w, w.z / w.w);
        vec3 ev = (camToWorld[0].xyz * dc.x + camToWorld[1].xyz * dc.y) + camToWorld[2].xyz * dc.z;
        vec3 d = ev * rsq((ev.z * ev.z + ev.y * ev.y) + ev.x * ev.x);
        sum = sum + pathRadiance(o, d);
        cnt = cnt + 1.0;
    }
    outSum = vec4(sum, 1.0);
    outCount = vec4(cnt, 0.0, 0.0, 1.0);
}
