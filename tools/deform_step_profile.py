#!/usr/bin/env python3
"""The decoder-free product deform network (splatfields_amd.deform_field.SplatFields, the sampler owns its planes) at 100 k
points: K training steps (forward + backward of a generic loss), for `rocprofv3 --kernel-trace --stats`; prints the wall time."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from splatfields_amd.deform_field import SplatFields

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
xyz = (torch.rand(n, 3, device=dev) * 2 - 1).requires_grad_(True)
t_emb = torch.full((n, 1), 0.37, device=dev)
net = SplatFields(n_frames=50, composition_rank=10, encoder=None, flow_model="offset").to(dev)
params = list(net.parameters())


def step():
    for p in params:
        p.grad = None
    xyz.grad = None
    out = net(xyz, t_emb)
    sum((v * v).mean() for k, v in out.items() if torch.is_tensor(v) and k != "flow").backward()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
print(json.dumps({"points": n, "steps": steps, "ms_per_step": ms}))
