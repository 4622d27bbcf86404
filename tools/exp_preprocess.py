"""Experiment: duration of pof_scan_preprocess for different output subsets (run under rocprofv3 --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from planar_optical_flow_amd import ops, synth
B=4096
sb=synth.make_batch(seed=2,B=B,T=2)
tab=ops.phi_table()
o,r,c=sb.det_csr()
det=ops.DetCSR.from_numpy(o,r,c,"cuda")
scans=torch.from_numpy(sb.scans).cuda(); o0=torch.from_numpy(sb.odom0).cuda(); o1=torch.from_numpy(sb.odom1).cuda()
variants=[("flow",),("xy",),("valid_mask",),("flow","exclude_mask"),("target_cls",),("target_cls","target_reg"),("flow","target_cls","target_reg","exclude_mask")]
ws=torch.empty(ops.scan_preprocess_workspace_bytes(B,len(r)),dtype=torch.uint8,device="cuda")
for v in variants:
    out={}
    for it in range(12):
        ops.scan_preprocess(scans,tab,o0,o1,det,want=v,out=out,workspace=ws)
    torch.cuda.synchronize()
print("variants", variants)
