"""Capture the batch-1 streaming step of DR-SPAAM in round 1's form (the cutout clears its area word with a 4-byte
hipMemsetAsync node) and DUMP the captured graph -- no replay: the hang of round 1 appeared on the second replay,
so nothing here can provoke it.  Prints what the evidence needs: the memset node, its predecessors / successors,
whether every path into the cutout kernels passes through it, and which other nodes use the same address.

    POF_CUTOUT_CLEAR=memset python tools/diag_graph_dump.py [owned|dropped] > gpurun_out/r3_graph_dump.txt

`dropped`: the 4-byte workspace is a torch.empty made and dropped inside the capture (round 1);
`owned`  : the detector-owned workspace allocated before the capture (round 3).
"""
import os
import re
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from planar_optical_flow_amd import ops, streaming                              # noqa: E402
from planar_optical_flow_amd.src.depracted.model.dr_spaam import SpatialDROW  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "dropped"
assert os.environ.get("POF_CUTOUT_CLEAR") == "memset", "run with POF_CUTOUT_CLEAR=memset"
torch.manual_seed(3)
model = SpatialDROW(num_scans=5, num_pts=56, alpha=0.5, window_size=11, pedestrian_only=True).cuda().eval()
det = streaming.StreamingDetector(model, batch=1, graph=False)
scan = torch.rand(450, device="cuda") * 8 + 1
det(scan)                                   # first scan: eager, creates the template
det(scan)                                   # steady-state step, eager (lazy initialisations)
if mode == "dropped":
    det._cut_ws = None                      # ops.cutout then allocates (and drops) its word inside the capture
torch.cuda.synchronize()
# torch's debug_dump() writes nothing on ROCm: keep the captured graph and call hipGraphDebugDotPrint on its handle
import ctypes
g = torch.cuda.CUDAGraph(keep_graph=True)
with torch.cuda.graph(g):
    cls, reg, tmpl, fused = det._step(False)
    det.template.copy_(tmpl)
path = os.path.join(REPO, "gpurun_out", "r3_stream_graph_%s.dot" % mode)
os.makedirs(os.path.dirname(path), exist_ok=True)
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipGraphDebugDotPrint.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint]
rc = hip.hipGraphDebugDotPrint(ctypes.c_void_p(g.raw_cuda_graph()), path.encode(), 1)     # 1 = verbose
print("hipGraphDebugDotPrint rc", rc)
txt = open(path).read()
nodes = dict(re.findall(r'"?(graph_\d+_node_\d+|\d+)"?\s*\[([^\]]*)\]', txt))
edges = re.findall(r'"?(graph_\d+_node_\d+|\d+)"?\s*->\s*"?(graph_\d+_node_\d+|\d+)"?', txt)
print("mode", mode, "nodes", len(nodes), "edges", len(edges), "dot bytes", len(txt))
mem = [n for n, a in nodes.items() if "MEMSET" in a.upper()]
print("memset nodes:", len(mem))
for n in mem:
    print("  ", n, " ".join(nodes[n].split())[:400])
    print("   predecessors:", [a for a, b in edges if b == n])
    succ = [b for a, b in edges if a == n]
    print("   successors:", [(b, " ".join(nodes.get(b, "").split())[:120]) for b in succ])
cut = [n for n, a in nodes.items() if "cutout" in a]
print("cutout kernel nodes:", [(n, " ".join(nodes[n].split())[:100]) for n in cut])
for n in cut:
    print("   predecessors of", n, ":", [a for a, b in edges if b == n])
print("(no replay was issued)")
