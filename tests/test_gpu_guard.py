"""The persistent render kernel is bounded: pt_render_wgwf's trip guards end a launch whose queue bookkeeping has slipped
as a failed launch (GLRTX_EDEVICE at the next fold) instead of a kernel that never ends -- and the bookkeeping they watch over, the forwarding
address a parked ray's path leaves behind when it moves to another queue position, has a named regression test against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT, assert_bit_equal
from glrt_amd import host, scenes

pytestmark = pytest.mark.gpu

FAULT_LIB = PKG / "lib" / "libglrtx_fault.so"

_CHILD = r"""
import os, pathlib, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {pkg!r})
from glrt_amd import device, host, scenes
device.lib_path = lambda: pathlib.Path({lib!r})
os.environ["GLRTX_SUSPEND_MAX"] = "64"   # park whenever a wave's share of the queue is used up
os.environ["GLRTX_BLOCK_PATHS"] = "256"
sc, pr = scenes.CONFIGS["headline"](width=96, height=64)
d = device.Device(); d.upload_scene(sc); d.resize(96, 64)
try:
    d.render_frames(pr, [host.frame_seed(i) for i in range(4)])
    d.sync()
    print("RESULT completed")
except device.GlrtxError as e:
    print("RESULT error", e.code, str(e))
    # the context is usable afterwards: the next launch (no parking: nothing to lose) renders and folds without an error
    os.environ["GLRTX_SUSPEND_MAX"] = "0"
    d.clear(); d.render(dict(pr, seed=host.frame_seed(9))); d.sync()
    print("RESULT recovered", int(d.stats().device_error_pending))
"""


def test_trip_guard_turns_a_lost_forwarding_address_into_a_failed_launch():
    """libglrtx_fault.so (-DGLRTX_FAULT_INJECT) loses the forwarding address of every parked ray: the ray is dropped and its path waits for ever.  The product
    build of round 5 would never have left the loop; now the workgroup sees two trips in which no ray is dealt and no path moves, reports through the
    context's guard words and leaves, and glrtx_sync returns GLRTX_EDEVICE naming the guard.  Runs in a child process under a time limit."""
    if not FAULT_LIB.exists():
        pytest.skip("libglrtx_fault.so not built (make -C opengl-raytracer_amd diag)")
    code = _CHILD.format(root=str(ROOT), pkg=str(PKG / "python"), lib=str(FAULT_LIB))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    out = r.stdout
    assert r.returncode == 0, (r.returncode, out[-2000:], r.stderr[-2000:])
    assert "RESULT error -2" in out and "trip guard" in out and "no path moved" in out, out[-2000:]
    assert "RESULT recovered 0" in out, out[-2000:]


@pytest.mark.parametrize("block_paths", [256, 1024])
def test_parked_ray_forwarding_regression(gpu_device, monkeypatch, block_paths):
    """The same launch shape on the PRODUCT library against the oracle: parking forced at every opportunity (GLRTX_SUSPEND_MAX=64), small path queues so that
    the live paths change position every trip, depth 8 and 3 samples per pixel so that paths die, restart in place and finish while a neighbour's ray is
    parked -- every one of those moves relies on the forwarding address wg_shade_phase leaves behind the parked-ray mark (pt_kernel.hip.h).  Bit-exact, ray
    count included, and no guard fires."""
    from oracle import pt_oracle
    monkeypatch.setenv("GLRTX_SUSPEND_MAX", "64")
    monkeypatch.setenv("GLRTX_BLOCK_PATHS", str(block_paths))
    d = gpu_device
    scene, params = scenes.CONFIGS["headline"](width=96, height=64, n_samples=3)
    seeds = [host.frame_seed(i) for i in range(4)]
    ref, ref_rays = None, 0
    for sd in seeds:
        ref, n = pt_oracle.render(scene, dict(params, seed=sd), accum=ref)
        ref_rays += n
    d.upload_scene(scene); d.resize(96, 64); d.clear(); d.reset_stats(); d.count_rays(True)
    d.render_frames(params, seeds); d.sync()
    st = d.stats()
    assert st.rays == ref_rays and st.device_error_pending == 0
    assert_bit_equal(d.read_accum(), ref, f"parked-ray forwarding, block_paths {block_paths}")
    d.count_rays(False)
