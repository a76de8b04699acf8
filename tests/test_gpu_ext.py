"""The HIP extension kernel (analytic spheres, dielectric, Whitted termination; SURVEY.md 8(f) f4, PARITY UNPINNED -- see
tests/test_ext.py) against its CPU statement in oracle/pt_oracle.c, bit for bit, through the C ABI."""
import numpy as np
import pytest

from conftest import assert_bit_equal
from glrt_amd import device, host, scenes
from oracle import pt_oracle

pytestmark = pytest.mark.gpu


def render_ext(d, sc, pr, spheres, flags, seeds=None):
    d.upload_scene(sc)
    d.upload_spheres(spheres)
    d.set_extensions(flags)
    d.set_partition(0, 1, 16)
    d.resize(pr["width"], pr["height"])
    d.reset_stats()
    d.count_rays(True)
    try:
        if seeds is None:
            d.render(pr)
        else:
            d.render_frames(pr, seeds)  # several frames through the frames entry point (one launch per frame on this path)
        d.sync()
        return d.read_accum(), int(d.stats().rays)
    finally:
        d.set_extensions(0)
        d.upload_spheres(None)


CASES = [
    ("spheres", dict(max_depth=4, n_samples=4), False, 0),
    ("spheres_deep", dict(max_depth=16, n_samples=2), False, 0),
    ("glass", dict(max_depth=8, n_samples=4), True, device.EXT_DIELECTRIC),
    ("glass_flag_off", dict(max_depth=8, n_samples=2), True, 0),
    ("whitted", dict(max_depth=8, n_samples=4), False, device.EXT_WHITTED),
    ("glass_whitted", dict(max_depth=12, n_samples=4), True, device.EXT_DIELECTRIC | device.EXT_WHITTED),
]


@pytest.mark.parametrize("name,kw,glass,flags", CASES, ids=[c[0] for c in CASES])
def test_extension_kernel_equals_its_cpu_statement(gpu_device, name, kw, glass, flags):
    sc, pr, sph = scenes.config_spheres(96, 80, glass=glass, **kw)
    ref, ref_rays = pt_oracle.render(sc, pr, spheres=sph, ext_flags=flags)
    acc, rays = render_ext(gpu_device, sc, pr, sph, flags)
    assert rays == ref_rays
    assert_bit_equal(acc, ref, name)


def test_extensions_on_triangle_meshes_and_over_several_frames(gpu_device):
    """EXT_DIELECTRIC applies to triangles too (an icosphere of glass), frames accumulate through glrtx_render_frames, and with
    everything switched off again the context renders the pinned image."""
    sc, pr, none = scenes.config_spheres(80, 64, max_depth=8, n_samples=2, subdiv=2, glass=True)
    seeds = [host.frame_seed(f) for f in range(3)]
    ref = np.zeros((64, 80, 4), np.float32)
    ref_rays = 0
    for sd in seeds:
        _, r = pt_oracle.render(sc, dict(pr, seed=sd), accum=ref, ext_flags=pt_oracle.EXT_DIELECTRIC)
        ref_rays += r
    acc, rays = render_ext(gpu_device, sc, pr, None, device.EXT_DIELECTRIC, seeds)
    assert rays == ref_rays
    assert_bit_equal(acc, ref, "glass icosphere, 3 frames")
    d = gpu_device
    d.clear(); d.reset_stats()
    d.render(pr); d.sync()
    pinned, _ = pt_oracle.render(sc, pr)
    assert_bit_equal(d.read_accum(), pinned, "extensions off again")


def test_sphere_upload_validation(gpu_device):
    d = gpu_device
    sc, pr, sph = scenes.config_spheres(32, 32)
    d.upload_scene(sc)
    bad = sph.copy(); bad[1, 3] = 0.0
    with pytest.raises(device.GlrtxError) as e:
        d.upload_spheres(bad)
    assert e.value.code == device.GLRTX_ESCENE and "radius" in str(e.value)
    bad = sph.copy(); bad[2, 4] = 99
    with pytest.raises(device.GlrtxError) as e:
        d.upload_spheres(bad)
    assert e.value.code == device.GLRTX_ESCENE and "material" in str(e.value)
    with pytest.raises(device.GlrtxError):
        d.set_extensions(64)
    d.upload_spheres(None)


def test_extension_kernel_on_hostile_sphere_lists(gpu_device):
    """Up to 40 overlapping and nested spheres with an occasional radius of 1e-30 / 1e30 or a centre at infinity (what glrtx_upload_spheres accepts; radii that are
    not positive are refused with GLRTX_ESCENE), indices of refraction of 0, 1, 1e-30, 1e30, negative, NaN, inf, a glass icosphere on the triangle path, every
    combination of the two flags, depths 1 .. 8: device == CPU statement, ray counts included.  (tools/gpu_ext_fuzz.py: 600 scenes in round 4, 0 mismatches.)"""
    from glrt_amd.scenes import SceneBuilder, quad, conductor, diffuse, emitter, dielectric, camera, make_params
    rng = np.random.default_rng(9)
    spec = [0.0, -0.0, -1.0, 1e-30, 1e30, np.inf, -np.inf, np.nan, 1e-45, 3e38, 1.0]
    d = gpu_device
    W, H = 48, 36
    accepted = refused = 0
    try:
        for it in range(80):
            b = SceneBuilder()
            ior = float(rng.choice([1.5, 1.0, 0.0, 1e-30, 1e30, -1.5, np.nan, np.inf, 0.7, 2.4]))
            mats = [b.add_material(diffuse((0.7, 0.7, 0.7))), b.add_material(diffuse((0.8, 0.3, 0.3))), b.add_material(dielectric(ior)),
                    b.add_material(conductor(scenes.COPPER["eta"], scenes.COPPER["kappa"], 0.2))]
            lamp = b.add_material(emitter((10.0, 10.0, 10.0)))
            b.add_mesh(*quad((-10, 0, 10), (20, 0, 0), (0, 0, -20)), mats[0])
            b.add_mesh(*quad((-1, 5, -1), (2, 0, 0), (0, 0, 2)), lamp)
            if rng.integers(0, 3) == 0:
                b.add_mesh(*scenes.icosphere(1, 0.8, (0.0, 0.8, 2.0)), mats[2])
            n = int(rng.integers(0, 41))
            sph = np.zeros((n, 5), np.float32)
            for i in range(n):
                sph[i, :3] = rng.uniform(-3, 3, 3) + [0, 1.5, 0]
                sph[i, 3] = rng.uniform(0.05, 1.2)
                sph[i, 4] = mats[int(rng.integers(0, 4))]
                if rng.integers(0, 5) == 0:
                    sph[i, int(rng.integers(0, 4))] = spec[int(rng.integers(0, len(spec)))]
            sc = b.build("sah")
            c2w, s2c = camera((0, 3, 9), (0, 1, 0), (0, 1, 0), 40.0, W, H)
            p = make_params(c2w, s2c, W, H, int(rng.integers(1, 9)), int(rng.integers(1, 3)), seed=(float(rng.uniform()), float(rng.uniform())))
            flags = int(rng.integers(0, 4))
            d.upload_scene(sc)
            try:
                d.upload_spheres(sph if n else None)
            except device.GlrtxError as e:
                assert e.code == device.GLRTX_ESCENE, e
                refused += 1
                continue
            accepted += 1
            ref, ref_rays = pt_oracle.render(sc, p, spheres=sph if n else None, ext_flags=flags)
            d.set_extensions(flags); d.set_partition(0, 1, 16); d.resize(W, H); d.clear(); d.reset_stats(); d.count_rays(True)
            d.render(p); d.sync()
            acc = d.read_accum()
            same = (acc.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(acc) & np.isnan(ref))
            assert same.all(), f"scene {it}: {int((~same).any(-1).sum())} pixels differ (flags {flags}, ior {ior}, {n} spheres)"
            assert d.stats().rays == ref_rays, f"scene {it}"
    finally:
        d.set_extensions(0)
        d.upload_spheres(None)
    assert accepted > 30 and refused > 5, (accepted, refused)
