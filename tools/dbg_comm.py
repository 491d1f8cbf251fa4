import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
which = sys.argv[1] if len(sys.argv) > 1 else "torch"
if which == "torch":
    import torch
    print("torch cuda", torch.cuda.is_available(), torch.cuda.device_count())
from slam_toolbox_amd import capi, comm
L = capi.lib()
print("kh_device_count before", L.kh_device_count())
uid = comm.unique_id()
print("uid ok; kh_device_count after rccl", L.kh_device_count())
for line in open("/proc/self/maps"):
    if ("hip" in line or "rccl" in line or "hsa" in line) and "r-xp" in line:
        print(line.split()[-1])
try:
    c = comm.Communicator(0, 0, 1, uid)
    print("comm ok")
except Exception as e:
    print("comm failed", e)
