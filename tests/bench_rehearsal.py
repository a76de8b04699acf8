"""CPU rehearsal of bench.py's N > 1 control flow (TEST INFRASTRUCTURE, not collected by pytest).

Runs bench.main() with bench.RENDERER_FACTORY replaced by a renderer that holds one rank's rows in a CPU tensor and
fills them with the oracle (oracle/pt_oracle.c standing in for the device kernel), and with the gloo backend.  Started
as `python tests/bench_rehearsal.py --gpus 2 ...` WITHOUT WORLD_SIZE it exercises bench.py's self-launch: bench.main
re-launches sys.argv[0] -- this file -- as N ranks under torch.distributed.run, each of which lands here again.
What is under test: the self-spawn, the step/launch plan, seeds, the row partition, the root-only gather with the
precomputed index, the max-over-ranks timing and the JSON line.  The final image is written for comparison."""
import os
import pathlib
import sys
import types

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "opengl-raytracer_amd" / "python"))

import bench  # noqa: E402

OUT = os.environ.get("GLRT_REHEARSAL_OUT")


class OracleRenderer:
    def __init__(self, rank, world, local_rank, scene, params, bvh):
        import torch
        from glrt_amd import dist
        from oracle import pt_oracle
        self.o = pt_oracle
        self.scene, self.params = scene, params
        self.rank, self.world = rank, world
        W, H = params["width"], params["height"]
        self.ys = dist.owned_rows(rank, world, bench.STRIPE, H)
        self.full = np.zeros((H, W, 4), np.float32)  # rows of other ranks stay untouched
        self.accum = torch.zeros((dist.max_owned_rows(world, bench.STRIPE, H), W, 4), dtype=torch.float32)
        self.device = torch.device("cpu")
        self.counting = False
        self.st = types.SimpleNamespace(rays=0, rays_untraced=0, launches=0, kernel_launches=0, kernel_ms_total=0.0)
        self.frames = []

    def render_frames(self, f0, n, seed_of):
        import time
        t = time.perf_counter()
        for f in range(f0, f0 + n):
            self.frames.append(f)
            for s0 in range(0, len(self.ys), bench.STRIPE):
                seg = self.ys[s0:s0 + bench.STRIPE]
                _, r = self.o.render(self.scene, dict(self.params, seed=seed_of(f)), accum=self.full, rows=(int(seg[0]), int(seg[-1]) + 1), threads=1)
                if self.counting:
                    self.st.rays += r
        import torch
        self.accum[:len(self.ys)] = torch.from_numpy(self.full[self.ys])
        self.st.launches += n
        self.st.kernel_launches += 1
        self.st.kernel_ms_total += (time.perf_counter() - t) * 1e3

    def render_full_reference(self, f0, n, seed_of, per_launch=16):
        import torch
        if os.environ.get("GLRT_REHEARSAL_FAIL_RANK0") == "2":
            raise RuntimeError("rehearsal: the one-rank reference render failed on purpose")
        W, H = self.params["width"], self.params["height"]
        full = np.zeros((H, W, 4), np.float32)
        for f in range(f0, f0 + n):
            self.o.render(self.scene, dict(self.params, seed=seed_of(f)), accum=full, threads=1)
        return torch.from_numpy(full)

    def time_one_rank(self, f0, n, seed_of, per_launch):
        """The strong-scaling denominator bench.py measures on rank 0 alone while the other ranks wait (GpuRenderer.time_one_rank).  GLRT_REHEARSAL_FAIL_RANK0=1 makes it
        raise: the run must still finish (rank 0 must reach the barrier the others wait in) and say so in scaling_strong.error."""
        import time
        if os.environ.get("GLRT_REHEARSAL_FAIL_RANK0"):
            raise RuntimeError("rehearsal: the one-GPU denominator failed on purpose")
        t = time.perf_counter()
        self.render_full_reference(f0, n, seed_of)
        return time.perf_counter() - t

    def cleared(self):
        self.full[:] = 0.0  # bench.py zeroed `accum` without a reset_stats behind it

    def device_sync(self):
        pass

    def count_rays(self, on):
        self.counting = on

    def sync(self):
        pass

    def stats(self):
        return types.SimpleNamespace(**vars(self.st))  # a snapshot: bench.py takes differences

    def reset_stats(self):
        self.full[:] = 0.0 if not self.accum.any() else self.full  # bench zeroes accum right before: keep both in step
        self.st = types.SimpleNamespace(rays=0, rays_untraced=0, launches=0, kernel_launches=0, kernel_ms_total=0.0)

    def timer_begin(self):
        import time
        self._t = time.perf_counter()

    def timer_end(self):
        import time
        return (time.perf_counter() - self._t) * 1e3

    def close(self):
        if OUT:
            np.save(os.path.join(OUT, f"rows_rank{self.rank}.npy"), self.accum.numpy())
            np.save(os.path.join(OUT, f"frames_rank{self.rank}.npy"), np.asarray(self.frames))


def small_config():
    from glrt_amd import scenes
    return scenes.config_c1(48, 72, max_depth=3, n_samples=1, subdiv=1)


if __name__ == "__main__":
    from glrt_amd import dist, scenes
    scenes.CONFIGS["rehearsal"] = small_config
    bench.RENDERER_FACTORY = OracleRenderer
    if OUT:  # keep the image the root gathered
        orig = dist.RowGather.gather_to_root

        def keep(self, local, dst=0):
            img = orig(self, local, dst)
            if img is not None:
                np.save(os.path.join(OUT, "gathered.npy"), img.numpy())
            return img
        dist.RowGather.gather_to_root = keep
        orig_finish = dist.RowGather.finish

        def keep_finish(self, handle):  # the strong-scaling region gathers asynchronously
            img = orig_finish(self, handle)
            if img is not None:
                np.save(os.path.join(OUT, "gathered.npy"), img.numpy())
            return img
        dist.RowGather.finish = keep_finish
    if os.environ.get("GLRT_REHEARSAL_CORRUPT_GATHER"):  # negative case: a collective that delivers one wrong value must show up in config.gather_check
        clean = dist.RowGather.gather_to_root

        def corrupt(self, local, dst=0):
            img = clean(self, local, dst)
            if img is not None:
                img[3, 5, 1] += 1.0
            return img
        dist.RowGather.gather_to_root = corrupt
    bench.main()
