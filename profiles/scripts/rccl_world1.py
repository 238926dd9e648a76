"""The final gather of pooled features on the RCCL backend with a single rank
(the one-GPU box cannot hold two): process-group set-up, the count exchange and
the padded all-gather of mvfusion.distributed.gather_pooled on device tensors."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-fusion_amd"))
from mvfusion.distributed import gather_pooled  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.rand((300000, 26), device=dev)
y = torch.randint(0, 19, (300000,), device=dev)
f, l, c = gather_pooled(x, y)
torch.cuda.synchronize()
assert torch.equal(f, x) and torch.equal(l, y), "gather mismatch"
print("rccl world-1 gather ok", tuple(f.shape), tuple(l.shape), c)
dist.barrier()
dist.destroy_process_group()
