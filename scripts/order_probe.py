import sys, os
sys.path.insert(0, os.getcwd())
mode = sys.argv[1]
if mode == "torch_import_first":
    import torch
from raytracing_amd import _lib
print(mode, "devices seen by the library:", _lib.load().rt_device_count(), flush=True)
import torch
try:
    torch.cuda.init()
    print(mode, "torch sees", torch.cuda.device_count(), torch.zeros(3, device="cuda").sum().item(), flush=True)
except Exception as e:
    print(mode, "torch failed:", repr(e)[:200], flush=True)
maps = open("/proc/self/maps").read()
print(mode, sorted({l.split()[-1] for l in maps.splitlines() if "amdhip64" in l or "hsa-runtime" in l}), flush=True)
