import sys, math, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
from common import Scenario, make_hip_matcher
from slam_toolbox_amd.scan_matcher import _scan_array
def free(): f,t=torch.cuda.mem_get_info(0); return f/1e9
print("start", free())
n=256
hm = make_hip_matcher("C2", max_batch=n)
print("after create", free())
sc = Scenario(seed=70, n_base=6, start=3)
q, base = sc.hip_scans()
for b in range(n): hm.AddScans(q, base, slot=b)
print("after add scans", free())
corr = ((0.15, 0.15), (0.005, 0.005), math.radians(20.0), math.radians(0.5))
centers=np.asarray([sc.query_pose]*n)
hm.CorrelateScanBatch(None, centers, *corr, True, False, scan_array=(_scan_array([q]*n), n))
print("after correlate batch", free())
hm.close()
print("after close", free())
