"""The C++ drop-in facade (glrt::Scene::parse + glrt::Window::mainloop, glrt_main) end to end on the GPU:
JSON + OBJ in, output.png out, compared with the same frames driven through the Python binding."""
import subprocess

import numpy as np
import pytest

from conftest import PKG
from glrt_amd import host, scenes

pytestmark = pytest.mark.gpu


def _c1_builder(subdiv=1):
    b = scenes.SceneBuilder()
    grey = b.add_material(scenes.diffuse((0.7, 0.7, 0.7)))
    red = b.add_material(scenes.diffuse((0.8, 0.3, 0.3)))
    cu = b.add_material(scenes.conductor(scenes.COPPER["eta"], scenes.COPPER["kappa"], 0.2))
    lamp = b.add_material(scenes.emitter((10.0, 10.0, 10.0)))
    b.add_mesh(*scenes.quad((-10, 0, 10), (20, 0, 0), (0, 0, -20)), grey)
    b.add_mesh(*scenes.icosphere(subdiv, 1.0, (-2.2, 1.0, 0.0)), red)
    b.add_mesh(*scenes.icosphere(subdiv, 1.0, (0.0, 1.0, 0.0)), cu)
    b.add_mesh(*scenes.icosphere(subdiv, 1.0, (2.2, 1.0, 0.0)), grey)
    b.add_mesh(*scenes.quad((-1, 5, -1), (2, 0, 0), (0, 0, 2)), lamp)
    return b


@pytest.mark.parametrize("in_flight", [1, 2, 8])
def test_glrt_main_renders_json_scene_like_the_binding(tmp_path, gpu_device, in_flight):
    from PIL import Image
    w, h, depth, frames = 96, 64, 4, 3
    b = _c1_builder()
    js = scenes.export_json_obj(b, tmp_path, w, h, (0, 3, 9), (0, 1, 0), (0, 1, 0), 40.0)
    out = tmp_path / "out.png"
    r = subprocess.run([str(PKG / "lib" / "glrt_main"), "-i", str(js), "-s", "4", "--max-depth", str(depth), "--frames",
                        str(frames), "--frames-in-flight", str(in_flight), "--out", str(out)], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "#triangle: 244" in r.stdout and "Save:" in r.stdout
    img = np.asarray(Image.open(out))
    assert img.shape == (h, w, 4)

    # same frames through the C ABI from Python: same BVH builder, same camera helpers, same seeds
    # (material ids follow shape order in Scene::parse, so rebuild the scene with one material per shape)
    b2 = scenes.SceneBuilder()
    for pos, nrm, mid in zip(b._pos, b._nrm, b._mid):
        b2.add_mesh(pos, nrm, b2.add_material(b.materials[int(mid[0])]))
    scene = b2.build()
    c2w, s2c = scenes.camera((0, 3, 9), (0, 1, 0), (0, 1, 0), 40.0, w, h)
    params = scenes.make_params(c2w, s2c, w, h, depth, 1)
    d = gpu_device
    d.upload_scene(scene)
    d.set_partition(0, 1, 16)
    d.resize(w, h)
    for f in range(frames):
        d.render(dict(params, seed=host.frame_seed(f), focal=0.0))  # absent focalLength parses as 0 (scene.cpp:71-74)
    d.sync()
    ref = d.resolve_rgba8(2.2, True)
    assert np.array_equal(img, ref)


def test_glrt_main_with_gpu_built_lbvh_gives_the_same_image(tmp_path, gpu_device):
    """--bvh lbvh builds the tree on the device inside Scene::parse; tree shape does not change the image
    (exact ties aside: the image is compared with a tolerance of a few differing pixels)."""
    from PIL import Image
    b = _c1_builder()
    js = scenes.export_json_obj(b, tmp_path, 96, 64, (0, 3, 9), (0, 1, 0), (0, 1, 0), 40.0)
    imgs = {}
    for kind in ("sah", "lbvh", "lbvh-cpu", "sah-gpu", "sah-levels-cpu", "reference"):
        out = tmp_path / f"{kind}.png"
        r = subprocess.run([str(PKG / "lib" / "glrt_main"), "-i", str(js), "--max-depth", "3", "--frames", "2", "--bvh", kind,
                            "--out", str(out)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        if kind in ("lbvh", "sah-gpu"):
            assert "built on the GPU" in r.stdout
        if kind == "reference":  # round 6: the reference host's own tree (glrt_bvh_build_reference), never re-ordered
            assert "the reference host's own tree" in r.stdout and "the light side first" not in r.stdout
        imgs[kind] = np.asarray(Image.open(out)).astype(np.int32)
    assert np.array_equal(imgs["lbvh"], imgs["lbvh-cpu"])
    assert np.array_equal(imgs["sah-gpu"], imgs["sah-levels-cpu"])  # round 5: the binned SAH built on the device == its CPU statement
    assert (np.abs(imgs["sah-gpu"] - imgs["sah"]).max(-1) > 0).mean() < 0.01
    assert (np.abs(imgs["lbvh"] - imgs["sah"]).max(-1) > 0).mean() < 0.01
    assert (np.abs(imgs["reference"] - imgs["sah"]).max(-1) > 0).mean() < 0.01


def test_glrt_main_on_several_partitions_writes_the_identical_png(tmp_path):
    """glrt_main --devices 0,0 / 0,0,0 (glrt::Window on a glrtx_group; both shares on this GPU) and --save-every-frame
    (the reference's cadence, window.cpp:164): the PNG is byte-identical to the single-context run's."""
    b = _c1_builder()
    js = scenes.export_json_obj(b, tmp_path, 96, 72, (0, 3, 9), (0, 1, 0), (0, 1, 0), 40.0)
    pngs = {}
    for name, extra in (("one", []), ("two", ["--devices", "0,0"]), ("three", ["--devices", "0,0,0"]), ("every", ["--devices", "0,0", "--save-every-frame"])):
        out = tmp_path / f"{name}.png"
        r = subprocess.run([str(PKG / "lib" / "glrt_main"), "-i", str(js), "--max-depth", "4", "--frames", "5", "--frames-in-flight", "2",
                            "--out", str(out)] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.count("Save:") == (5 if name == "every" else 1)
        pngs[name] = out.read_bytes()
    assert pngs["one"] == pngs["two"] == pngs["three"] == pngs["every"]


def _sphere_scene_json(tmp_path, w, h):
    """BASELINE configs[0] read literally ("3 spheres + 1 ground plane") as a JSON scene for the facade's extension syntax, plus
    the same scene as flat buffers + sphere rows for the C ABI."""
    import json
    b = scenes.SceneBuilder()
    grey = b.add_material(scenes.diffuse((0.7, 0.7, 0.7)))
    lamp = b.add_material(scenes.emitter((10.0, 10.0, 10.0)))
    b.add_mesh(*scenes.quad((-10, 0, 10), (20, 0, 0), (0, 0, -20)), grey)
    b.add_mesh(*scenes.quad((-1, 5, -1), (2, 0, 0), (0, 0, 2)), lamp)
    js = scenes.export_json_obj(b, tmp_path, w, h, (0, 3, 9), (0, 1, 0), (0, 1, 0), 40.0)
    doc = json.loads(js.read_text())
    doc["scene"] += [
        {"type": "sphere", "center": [-2.2, 1.0, 0.0], "radius": 1.0, "material": "diffuse", "reflectance": [0.8, 0.3, 0.3]},
        {"type": "sphere", "center": [0.0, 1.0, 0.0], "radius": 1.0, "material": "dielectric", "ior": 1.5},
        {"type": "sphere", "center": [2.2, 1.0, 0.0], "radius": 1.0, "material": "diffuse", "reflectance": [0.7, 0.7, 0.7]},
    ]
    js.write_text(json.dumps(doc, indent=1))
    red = b.add_material(scenes.diffuse((0.8, 0.3, 0.3)))
    glass = b.add_material(scenes.dielectric(1.5))
    grey2 = b.add_material(scenes.diffuse((0.7, 0.7, 0.7)))
    spheres = np.array([[-2.2, 1.0, 0.0, 1.0, red], [0.0, 1.0, 0.0, 1.0, glass], [2.2, 1.0, 0.0, 1.0, grey2]], np.float32)
    return js, b.build(), spheres


def test_glrt_main_extension_scene_matches_the_c_abi(tmp_path, gpu_device):
    """glrt_main --extensions on a JSON scene with "sphere" shapes and a "dielectric" material (syntax of this build, not of the
    reference) == the same scene through glrtx_upload_spheres / glrtx_set_extensions; without --extensions the dielectric is
    the reference's "Unsupported material" abort, and sphere shapes alone are ignored like any non-obj shape."""
    from PIL import Image
    from glrt_amd import device
    w, h, depth, frames = 96, 64, 6, 3
    js, scene, spheres = _sphere_scene_json(tmp_path, w, h)
    out = tmp_path / "ext.png"
    cmd = [str(PKG / "lib" / "glrt_main"), "-i", str(js), "--max-depth", str(depth), "--frames", str(frames), "--out", str(out)]
    r = subprocess.run(cmd + ["--extensions"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "3 analytic spheres, dielectric" in r.stdout
    img = np.asarray(Image.open(out))
    c2w, s2c = scenes.camera((0, 3, 9), (0, 1, 0), (0, 1, 0), 40.0, w, h)
    params = scenes.make_params(c2w, s2c, w, h, depth, 1)
    d = gpu_device
    d.upload_scene(scene)
    d.upload_spheres(spheres)
    d.set_extensions(device.EXT_DIELECTRIC)
    try:
        d.set_partition(0, 1, 16)
        d.resize(w, h)
        for f in range(frames):
            d.render(dict(params, seed=host.frame_seed(f), focal=0.0))
        d.sync()
        assert np.array_equal(img, d.resolve_rgba8(2.2, True))
    finally:
        d.set_extensions(0)
        d.upload_spheres(None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "Unsupported material: dielectric" in (r.stdout + r.stderr)


def test_glrt_main_requires_input():
    r = subprocess.run([str(PKG / "lib" / "glrt_main")], capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stdout


def test_glrt_main_order_by_hits_gives_the_same_image(tmp_path, gpu_device):
    """--order-by-hits: one calibration frame, every fork's children ordered by the hits it counted (glrtx_hit_histogram, glrt_bvh_order_by_hits), the scene uploaded
    again -- the image is the plain run's (this scene has no exactly tied triangles) and the run says how many forks it exchanged."""
    from PIL import Image
    b = _c1_builder()
    js = scenes.export_json_obj(b, tmp_path, 96, 64, (0, 3, 9), (0, 1, 0), (0, 1, 0), 40.0)
    imgs = {}
    for tag, extra in (("plain", []), ("hits", ["--order-by-hits"])):
        out = tmp_path / f"{tag}.png"
        r = subprocess.run([str(PKG / "lib" / "glrt_main"), "-i", str(js), "--max-depth", "4", "--frames", "3", "--out", str(out)] + extra,
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        assert ("children ordered by the hits of a calibration frame" in r.stdout) == (tag == "hits")
        imgs[tag] = np.asarray(Image.open(out)).astype(np.int32)
    assert (np.abs(imgs["hits"] - imgs["plain"]).max(-1) > 0).mean() < 0.01
