"""Soak of the two device tree builders against their CPU statements (glrtx_build_lbvh == glrt_bvh_build_lbvh, glrtx_build_bvh_sah == glrt_bvh_build_sah_levels, bit for bit,
depth included) over random triangle sets the fixed sizes of the suite do not reach: sizes 1 ... 30 000, uniform / clustered / lattice-aligned (many equal centres and
equal costs) / duplicated / sliver / huge-and-tiny mixtures.
    python tools/gpu_builder_soak.py [cases] [first seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
import numpy as np
from glrt_amd import device, host

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def soup(rng):
    kind = rng.integers(0, 6)
    n = int(rng.choice([rng.integers(1, 70), rng.integers(60, 140), rng.integers(100, 3000), rng.integers(3000, 30000)], p=[0.25, 0.2, 0.4, 0.15]))
    ext = float(10.0 ** rng.uniform(-2, 4))
    if kind == 0:  # uniform
        c = rng.uniform(-ext, ext, (n, 1, 3)); e = rng.normal(0, ext * 0.02, (n, 3, 3))
    elif kind == 1:  # clusters of very different density
        k = int(rng.integers(1, 8)); centres = rng.uniform(-ext, ext, (k, 3)); radii = ext * 10.0 ** rng.uniform(-4, -0.5, k)
        which = rng.integers(0, k, n)
        c = (centres[which] + rng.normal(0, 1, (n, 3)) * radii[which, None])[:, None, :]; e = rng.normal(0, 1, (n, 3, 3)) * radii[which, None, None] * 0.1
    elif kind == 2:  # lattice: many equal centres / equal box faces
        g = int(rng.integers(2, 12)); c = (rng.integers(0, g, (n, 1, 3)) * (ext / g)).astype(np.float64)
        e = np.tile(rng.choice([ext / g, ext / (2 * g)]) * np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float64), (n, 1, 1))
        e = e[:, :, rng.permutation(3)]
    elif kind == 3:  # duplicates
        m = max(1, n // int(rng.integers(2, 20))); c0 = rng.uniform(-ext, ext, (m, 1, 3)); e0 = rng.normal(0, ext * 0.05, (m, 3, 3))
        pick = rng.integers(0, m, n); c, e = c0[pick], e0[pick]
    elif kind == 4:  # slivers and needles
        c = rng.uniform(-ext, ext, (n, 1, 3)); e = rng.normal(0, ext * 0.2, (n, 3, 3)) * 10.0 ** rng.uniform(-6, 0, (n, 1, 3))
    else:  # a few huge triangles over many tiny ones
        c = rng.uniform(-ext, ext, (n, 1, 3)); e = rng.normal(0, ext * 0.005, (n, 3, 3))
        big = rng.random(n) < 0.01; e[big] *= 300.0
    pos = (c + e).astype(np.float32)
    vert = np.zeros((n * 3, 5, 3), np.float32); vert[:, 0] = pos.reshape(-1, 3); vert[:, 1] = (0, 1, 0)
    tri = np.concatenate([np.arange(n * 3, dtype=np.float32).reshape(n, 3), np.zeros((n, 1), np.float32)], 1)
    return kind, n, vert.reshape(-1, 3), tri


d = device.Device()
bad, t0, by_kind = 0, time.time(), {}
for i in range(cases):
    rng = np.random.default_rng(seed0 + i)
    kind, n, vert, tri = soup(rng)
    by_kind[int(kind)] = by_kind.get(int(kind), 0) + 1
    for name, dev_fn, cpu_kind in (("lbvh", d.build_lbvh, "lbvh"), ("sah", d.build_bvh_sah, "sahl")):
        nodes, depth = dev_fn(vert, tri)[:2]
        want, want_depth = host.build_bvh(vert, tri, cpu_kind)
        same = nodes.shape == want.shape and np.array_equal(np.asarray(nodes).view(np.uint32), np.asarray(want).view(np.uint32)) and depth == want_depth
        if not same:
            bad += 1
            print(f"seed {seed0 + i} kind {kind} n {n}: {name} differs (depth {depth} / {want_depth})", flush=True)
    if (i + 1) % 50 == 0:
        print(f"{i + 1} cases, {bad} differences, {time.time() - t0:.0f} s", flush=True)
print(f"{cases} cases from seed {seed0} (per kind {dict(sorted(by_kind.items()))}), both builders: {bad} differences")
sys.exit(1 if bad else 0)
