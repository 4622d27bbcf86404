"""Drop-in for the reference's ``src/utils/rotate_iou.py`` host wrapper.

``rotate_iou_gpu_eval`` keeps the reference signature (:363-404) but runs the
HIP kernel of libpof_hip.so instead of a numba.cuda one.  ``rotate_iou_batched``
is the form the evaluation loop should use: one launch for the whole set instead
of one launch per sample (src/model/box_regression_fn.py:76-82).
"""
import numpy as np
import torch

from planar_optical_flow_amd import ops


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0, is_3d=False):
    """boxes [N,5|7], query_boxes [K,5|7] (centers, dims, angle; 3-D rows are
    x,y,z,l,w,h,rot) -> IoU [N,K] float32."""
    boxes = np.asarray(boxes).astype(np.float32)
    query_boxes = np.asarray(query_boxes).astype(np.float32)
    n, k = boxes.shape[0], query_boxes.shape[0]
    if n == 0 or k == 0:
        return np.zeros((n, k), dtype=np.float32)
    dev = torch.device("cuda", device_id)
    out = ops.rotate_iou(torch.from_numpy(boxes).to(dev), torch.from_numpy(query_boxes).to(dev),
                         criterion=criterion, is_3d=is_3d)
    return out.cpu().numpy().astype(boxes.dtype)


def rotate_iou_batched(boxes, query_lists, criterion=-1, device_id=0, is_3d=False):
    """boxes [G,s] (one per group), query_lists: list of G arrays [K_g,s].
    Returns list of G arrays [K_g] with the IoU of box g against its queries."""
    s = 7 if is_3d else 5
    g = len(query_lists)
    kmax = max([len(q) for q in query_lists] + [1])
    q = np.zeros((g, kmax, s), dtype=np.float32)
    kv = np.zeros(g, dtype=np.int32)
    for i, ql in enumerate(query_lists):
        ql = np.asarray(ql, dtype=np.float32).reshape(-1, s)
        q[i, :len(ql)] = ql
        kv[i] = len(ql)
    dev = torch.device("cuda", device_id)
    b = torch.from_numpy(np.asarray(boxes, dtype=np.float32).reshape(g, 1, s)).to(dev)
    out = ops.rotate_iou(b, torch.from_numpy(q).to(dev), criterion=criterion, is_3d=is_3d,
                         k_valid=torch.from_numpy(kv).to(dev)).cpu().numpy()
    return [out[i, 0, :kv[i]] for i in range(g)]
