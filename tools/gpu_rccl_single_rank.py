"""Diagnostic for a 1-GPU box: RCCL with ONE rank -- torch.distributed's nccl backend initialised the way bench.py does it (device_id), a barrier, RowGather's blocking and asynchronous
gather to the root and its all_gather on this rank's rows.  It cannot show what eight ranks do, but it shows that the library loads on the box, that the collective calls bench.py makes are
accepted by the backend with these tensor shapes (views of one receive buffer as the gather list), and that they run on the stream bench.py makes current.   python tools/gpu_rccl_single_rank.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "opengl-raytracer_amd", "python"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch  # noqa: E402
import torch.distributed as td  # noqa: E402

from glrt_amd import dist  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
stream = torch.cuda.Stream(device=0); torch.cuda.set_stream(stream)
td.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
H, W, STRIPE = 1080, 1920, 8
rows = dist.max_owned_rows(1, STRIPE, H)
local = torch.arange(rows * W * 4, dtype=torch.float32, device="cuda").reshape(rows, W, 4)
g = dist.RowGather(1, STRIPE, H, local)
td.barrier()
img = g.gather_to_root(local)
h = g.gather_to_root_async(local.clone()); img2 = g.finish(h)
img3 = g.all_gather(local)
torch.cuda.synchronize()
ok = bool(torch.equal(img, local) and torch.equal(img2, local) and torch.equal(img3, local))
print(f"RCCL single rank: backend {td.get_backend()}, gather / async gather / all_gather of {tuple(local.shape)} float32 {'identical to the input' if ok else 'WRONG'}; torch {torch.__version__}")
td.destroy_process_group()
sys.exit(0 if ok else 1)
