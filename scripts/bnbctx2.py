"""bench.secondary() on its own, then after a torch CUDA tensor has been made."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
import bench
api = mvolps_amd.api()
mvolps_amd.require_device()
if sys.argv[1] == "t":
    import torch
    torch.cuda.set_device(0)
    t = torch.tensor([1.0], dtype=torch.float64, device="cuda"); print(float(t.item()))
r = bench.secondary(api, with_cpu=False)
print(sys.argv[1], {k: round(v.get("nodes_per_s", 0)) for k, v in r.items()}, flush=True)
