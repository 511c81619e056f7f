"""`matrix_nms` with the reference's signature (model/matrix_nms.py:102-151), executed by the
HIP kernels (ppy_nms_candidates_f32 + ppy_matrix_nms_f32).  One image: boxes [M,4],
scores [M,C] on a ROCm device -> [K,6] rows (label, score, x0,y0,x1,y1) or [[-1]*6]."""
import torch

from ppyolo_hip import ops


def matrix_nms(bboxes, scores, score_threshold, post_threshold, nms_top_k, keep_top_k, use_gaussian=False,
               gaussian_sigma=2., return_index=False):
    M, C = scores.shape
    dev = bboxes.device
    b = bboxes.detach().float().contiguous().view(1, M, 4)
    s = scores.detach().float().contiguous().view(1, M, C)
    ck = torch.zeros((1, M * C), dtype=torch.int32, device=dev)
    ci = torch.zeros((1, M * C), dtype=torch.int32, device=dev)
    cc = torch.zeros((1,), dtype=torch.int32, device=dev)
    dets = torch.zeros((1, keep_top_k, 6), dtype=torch.float32, device=dev)
    cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    keep = torch.zeros((1, keep_top_k), dtype=torch.int32, device=dev)
    ops.nms_candidates(s, score_threshold, ck, ci, cc)
    ops.matrix_nms(b, C, ck, ci, cc, post_threshold, nms_top_k, keep_top_k, use_gaussian, gaussian_sigma, dets, cnt,
                   keep)
    k = int(cnt.item())
    pred = dets[0, :max(k, 1)].clone()
    return (pred, keep[0, :k].clone()) if return_index else pred
