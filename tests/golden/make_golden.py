#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the reference's UNMODIFIED fragment shader on Mesa llvmpipe.

Run in the build container only (needs /root/reference and Mesa's swrast_dri.so):

    make -C oracle && make -C opengl-raytracer_amd host && python tests/golden/make_golden.py

Every fixture stores the inputs (the five scene buffers in the reference wire format, the uniform
values) and the outputs read back from the GL framebuffer (RGB32F colour + R32F count, single
pass from cleared accumulators unless 'frames' is given), plus the GL renderer string.  The
reference's shader text is read at run time by oracle/glref.py and is never written here: a fixture
is data only.  The fixture list follows SURVEY.md section 8(c).
"""
from __future__ import annotations

import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "opengl-raytracer_amd" / "python"))

from glrt_amd import scenes  # noqa: E402
from glrt_amd.scenes import SceneBuilder, camera, conductor, diffuse, emitter, icosphere, make_params, media, quad  # noqa: E402
from oracle.glref import GLRef  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent
g = GLRef()


def save(name, scene, params, rows=None, frames=None):
    rgb, cnt = g.render_reference(scene, params, frames=frames)
    if rows is not None:
        rgb, cnt = rgb[rows[0]:rows[1]], cnt[rows[0]:rows[1]]
    np.savez_compressed(
        OUT / f"{name}.npz",
        vert=scene["vert"], tri=scene["tri"], mat=scene["mat"], light=scene["light"], bvh=scene["bvh"],
        c2w=params["c2w"], s2c=params["s2c"],
        scalars=np.array([params["width"], params["height"], params["max_depth"], params["n_samples"]], np.int32),
        fparams=np.array([params["seed"][0], params["seed"][1], params["aperture"], params["focal"]], np.float32),
        rows=np.array(rows if rows is not None else (0, params["height"]), np.int32),
        frames=np.array(frames if frames is not None else np.zeros((0, 2)), np.float32).reshape(-1, 2),
        out_rgb=rgb, out_count=cnt, renderer=np.array(g.info()))
    print(f"{name}: {rgb.shape} mean {rgb.mean():.5f} nonzero {np.count_nonzero(rgb.sum(-1))}")


def with_params(params, **kw):
    p = dict(params)
    p.update(kw)
    return p


# 1. emitter-only view: L = e exactly where the lamp is seen (camera, traversal, accumulate)
b = SceneBuilder()
lamp = b.add_material(emitter((3.0, 2.0, 1.0)))
grey = b.add_material(diffuse((0.6, 0.6, 0.6)))
b.add_mesh(*quad((-1, -1, 0), (2, 0, 0), (0, 2, 0)), lamp)
b.add_mesh(*quad((-4, -4, -1), (8, 0, 0), (0, 8, 0)), grey)
sc = b.build()
c2w, s2c = camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 45.0, 48, 48)
save("emitter_view", sc, make_params(c2w, s2c, 48, 48, 1, 1))

# 2. F6: floor perpendicular to the light normal is lit, wall parallel to the light is (almost) never lit
b = SceneBuilder()
grey = b.add_material(diffuse((0.7, 0.7, 0.7)))
lamp = b.add_material(emitter((20.0, 20.0, 20.0)))
b.add_mesh(*quad((-4, 0, 4), (8, 0, 0), (0, 0, -8)), grey)          # floor (+y)
b.add_mesh(*quad((-4, 4, -4), (8, 0, 0), (0, 0, 8)), grey)          # ceiling-level wall parallel to the lamp (-y)
b.add_mesh(*quad((-4, 0, -4), (8, 0, 0), (0, 8, 0)), grey)          # back wall (+z)
b.add_mesh(*quad((-1, 3.9, -1), (2, 0, 0), (0, 0, 2)), lamp)        # lamp facing down (-y)
sc = b.build()
c2w, s2c = camera((0, 2, 9), (0, 2, 0), (0, 1, 0), 40.0, 64, 64)
save("f6_floor_wall", sc, make_params(c2w, s2c, 64, 64, 3, 2, seed=(0.31, 0.62)))

# 3. conductor quad (GGX VNDF sampling, Fresnel, NEE on a conductor)
b = SceneBuilder()
cu = b.add_material(conductor(scenes.COPPER["eta"], scenes.COPPER["kappa"], 0.15))
gold = b.add_material(conductor((0.143, 0.375, 1.442), (3.983, 2.386, 1.603), 0.4))
lamp = b.add_material(emitter((12.0, 12.0, 12.0)))
grey = b.add_material(diffuse((0.5, 0.5, 0.5)))
b.add_mesh(*quad((-3, 0, 3), (3, 0, 0), (0, 0, -6)), cu)
b.add_mesh(*quad((0, 0, 3), (3, 0, 0), (0, 0, -6)), gold)
b.add_mesh(*quad((-3, 0, -3), (6, 0, 0), (0, 5, 0)), grey)
b.add_mesh(*quad((-1.5, 4.5, -1.5), (3, 0, 0), (0, 0, 3)), lamp)
sc = b.build()
c2w, s2c = camera((0, 2.5, 7), (0, 0.8, 0), (0, 1, 0), 40.0, 64, 64)
save("conductor_quads", sc, make_params(c2w, s2c, 64, 64, 6, 4, seed=(0.77, 0.12)))

# 4. C1 (3 icospheres + ground + lamp; subdiv 1 = 244 triangles) across depths and sample counts
sc, pr = scenes.config_c1(64, 64, subdiv=1)
for depth in (1, 4, 8, 16):
    for spp in (1, 16):
        save(f"c1_d{depth}_spp{spp}", sc, with_params(pr, max_depth=depth, n_samples=spp))

# 5. thin-lens depth of field
save("c1_dof", sc, with_params(pr, max_depth=4, n_samples=4, aperture=0.35, focal=8.5, seed=(0.402, 0.913)))

# 6. same triangles behind a chain BVH (linear scan) and a SAH BVH
sc3, pr3 = scenes.config_c3(72, 40, max_depth=2, n=1500, bvh="chain")
save("tris1500_chain", sc3, with_params(pr3, n_samples=2))
save("tris1500_sah", scenes.rebuild_bvh(sc3, "sah"), with_params(pr3, n_samples=2))

# 7. non-power-of-two size, single pass
sc, pr = scenes.config_c1(50, 38, subdiv=1)
save("c1_npot_50x38", sc, with_params(pr, max_depth=5, n_samples=3, seed=(0.555, 0.25)))

# 8. many small lights with distinct emissions: the first NEE rand() (light choice) is observable per pixel
b = SceneBuilder()
grey = b.add_material(diffuse((0.8, 0.8, 0.8)))
b.add_mesh(*quad((-6, 0, 6), (12, 0, 0), (0, 0, -12)), grey)
rng = np.random.default_rng(7)
for i in range(64):
    m = b.add_material(emitter(tuple(rng.uniform(1.0, 30.0, 3))))
    x, z = -5.25 + 1.5 * (i % 8), -5.25 + 1.5 * (i // 8)
    # upright, facing +z: the floor in front of each lamp is a perpendicular receiver (lit)
    b.add_mesh(*quad((x, 0.4, z), (0.5, 0, 0), (0, 0.5, 0)), m)
sc = b.build()
c2w, s2c = camera((0, 6, 10), (0, 0, 0), (0, 1, 0), 45.0, 64, 48)
save("many_lights", sc, make_params(c2w, s2c, 64, 48, 2, 2, seed=(0.045, 0.871)))

# 9. MTRL_MEDIA seen from the front falls through with the ray unchanged; from behind it is black
b = SceneBuilder()
fog = b.add_material(media())
grey = b.add_material(diffuse((0.7, 0.7, 0.7)))
lamp = b.add_material(emitter((9.0, 9.0, 9.0)))
b.add_mesh(*quad((-1.5, 0.2, 1), (3, 0, 0), (0, 2.5, 0)), fog)        # faces +z (towards the camera)
b.add_mesh(*quad((-4, 0, 4), (8, 0, 0), (0, 0, -8)), grey)
b.add_mesh(*icosphere(1, 0.8, (0, 0.8, -1)), grey)
b.add_mesh(*quad((-2, 0.5, -3), (4, 0, 0), (0, 3, 0)), lamp)            # upright lamp facing +z
sc = b.build()
c2w, s2c = camera((0, 2, 7), (0, 1, 0), (0, 1, 0), 40.0, 48, 48)
save("media_front", sc, make_params(c2w, s2c, 48, 48, 8, 2, seed=(0.29, 0.58)))

# 10. no lights at all (u_nLights = 0, H8): NEE fetches out of range and contributes nothing finite
b = SceneBuilder()
grey = b.add_material(diffuse((0.7, 0.7, 0.7)))
b.add_mesh(*quad((-4, 0, 4), (8, 0, 0), (0, 0, -8)), grey)
b.add_mesh(*icosphere(1, 1.0, (0, 1.0, 0)), grey)
sc = b.build()
c2w, s2c = camera((0, 2, 7), (0, 1, 0), (0, 1, 0), 40.0, 32, 32)
save("no_lights", sc, make_params(c2w, s2c, 32, 32, 3, 1))

# 11. ping-pong accumulation over 3 frames at a power-of-two size (window.cpp:213-252): the previous
#     frame is read through a LINEAR sampler, exact at POT sizes
sc, pr = scenes.config_c1(64, 64, subdiv=1)
save("c1_pingpong3", sc, with_params(pr, max_depth=4, n_samples=1),
     frames=[(0.137, 0.731), (0.755034, 0.112966), (0.373068, 0.494932)])

# 12. a primary ray with an exactly-zero direction component (centre column, jitter exactly 0.5): its
#     0*inf slab products are NaN and pin the min/max NaN rule of the traversal.  Rows 56..64 only.
sc, pr = scenes.config_c1(200, 120, max_depth=16, n_samples=4)
save("c1_zero_dir_rows", sc, pr, rows=(56, 64))

print("renderer:", g.info())

# ---- math fixtures from diagnostic shaders (own GLSL, not reference code): how this GL implementation
#      evaluates sin/cos and the classic fract(sin(dot)) hash recurrence
rng = np.random.default_rng(123)
xs = np.concatenate([rng.uniform(-92.0, 92.0, 128 * 128 - 2049), np.linspace(-8.0, 8.0, 2049)]).astype(np.float32)
fs = """#version 410
uniform samplerBuffer u_in;
out vec4 o;
void main() {
    int i = int(gl_FragCoord.y) * 128 + int(gl_FragCoord.x);
    float x = texelFetch(u_in, i).x;
    o = vec4(sin(x), cos(x), sqrt(abs(x)), inversesqrt(abs(x) + 1.0));
}
"""
o, = g.run_fragment(fs, 128, 128, (4,), [("u_in", xs, 1)])
o = o.reshape(-1, 4)
np.savez_compressed(OUT / "math_sincos.npz", x=xs, sin=o[:, 0], cos=o[:, 1], sqrt=o[:, 2], rsqrt=o[:, 3],
                    renderer=np.array(g.info()))
st = rng.uniform(0.0, 1.0, (64 * 64, 4)).astype(np.float32)
st[::3, 0] = (np.floor(st[::3, 0] * 1920) + 0.5) / 1920.0  # pixel-like states
st[::3, 1] = (np.floor(st[::3, 1] * 1080) + 0.5) / 1080.0
fs = """#version 410
uniform samplerBuffer u_in;
out vec4 o;
vec2 s; vec2 seed;
float next() {
    s.x = fract(sin(dot(s - seed, vec2(12.9898, 78.233))) * 43758.5453);
    s.y = fract(sin(dot(s - seed, vec2(12.9898, 78.233))) * 43758.5453);
    return s.x;
}
void main() {
    int i = int(gl_FragCoord.y) * 64 + int(gl_FragCoord.x);
    vec4 v = texelFetch(u_in, i);
    s = v.xy; seed = v.zw;
    float a = next(); float b = next(); float c = next();
    o = vec4(a, b, c, s.y);
}
"""
o, = g.run_fragment(fs, 64, 64, (4,), [("u_in", st, 4)])
np.savez_compressed(OUT / "math_rand.npz", state_seed=st, out=o.reshape(-1, 4), renderer=np.array(g.info()))
print("math fixtures written")
