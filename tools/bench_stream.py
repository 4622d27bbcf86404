"""Streaming DR-SPAAM step latency (one scan per call): eager launches vs one hipGraph replay."""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(90, exit=True)       # a stuck step reports where it is instead of hanging the box
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from planar_optical_flow_amd import synth
from planar_optical_flow_amd.streaming import StreamingDetector
from planar_optical_flow_amd.src.depracted.model.dr_spaam import SpatialDROW

torch.manual_seed(3)
model = SpatialDROW(num_scans=5, num_pts=56, alpha=0.5, window_size=11, pedestrian_only=True).cuda().eval()
for B in ([int(v) for v in sys.argv[1:]] or [1, 8]):
    scans = torch.from_numpy(synth.make_batch(seed=9, B=B, T=40).scans).cuda()     # [B, 40, 450]
    res = {}
    for graph in (False, True):
        det = StreamingDetector(model, batch=B, graph=graph)
        outs = []
        for t in range(8):
            cls, reg = det(scans[:, t])
            outs.append((cls.clone(), reg.clone()))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(8, 40):
            det(scans[:, t])
        torch.cuda.synchronize()
        res[graph] = ((time.perf_counter() - t0) / 32 * 1e3, outs)
    same = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(res[False][1], res[True][1]))
    print("streaming step B=%d: eager %.3f ms, hipGraph replay %.3f ms per scan (identical outputs: %s)"
          % (B, res[False][0], res[True][0], same), flush=True)
