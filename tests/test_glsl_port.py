"""oracle/glsl/pt_port.frag -- this repository's own GLSL statement of the path tracer, the shader bench.py times on llvmpipe on the GPU box
(cpu_baseline.llvmpipe; the reference's file cannot travel there) -- run on llvmpipe HERE and compared, bit for bit, with the images the
reference's unmodified shader produced on the same llvmpipe (tests/golden/*.npz).  Needs Mesa's swrast_dri.so (build container and GPU box
both have it); skipped where it is absent."""
import json
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_bit_equal, golden_names, load_golden

glport = pytest.importorskip("oracle.glport")
pytestmark = pytest.mark.skipif(not glport.available(), reason="Mesa llvmpipe (swrast_dri.so) or oracle/_ref/libglref.so absent")


@pytest.mark.parametrize("name", golden_names())
def test_own_glsl_port_reproduces_the_reference_images_on_llvmpipe(name):
    scene, params, rows, frames, rgb, cnt = load_golden(name)
    if len(frames) > 1 and (params["width"] & (params["width"] - 1) or params["height"] & (params["height"] - 1)):
        pytest.skip("multi-frame fixture at a non-power-of-two size: the reference's LINEAR sampler bleeds there (SURVEY.md F7), the port fetches exactly")
    r, c = glport.render(scene, params, frames=frames or None)
    assert_bit_equal(r[rows[0]:rows[1]], rgb, f"{name} rgb")
    assert_bit_equal(c[rows[0]:rows[1]], cnt, f"{name} count")


def test_llvmpipe_leg_as_bench_runs_it():
    """The child process bench.py starts for cpu_baseline.llvmpipe: JSON line, image equal to the C restatement over the same frames."""
    r = subprocess.run([sys.executable, "-m", "oracle.glport", "--config", "c1", "--frames", "2", "--warmup", "1", "--check"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["available"] and "llvmpipe" in j["renderer"] and j["ms_per_frame"] > 0 and j["cores"] >= 1
    assert j["image_vs_c_restatement"] == "bit-identical" and j["rays"] > 0
    assert np.isfinite(j["ms_per_frame"])
