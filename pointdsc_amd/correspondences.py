"""Correspondence construction on the GPU -- the step in front of ``PointDSC.forward``.

Mirrors what the reference does with numpy on the host inside its datasets / demo:

    datasets/ThreeDMatch.py:283-290   distance = sqrt(2 - 2 * src_desc @ tgt_desc.T + 1e-6); argmin (+ mutual check)
    datasets/ThreeDMatch.py:299-308   gather the keypoints, corr_pos = concat(src, tgt) - mean       (in_dim = 6)
    demo_registration.py:101-108      the same without the mutual check
    evaluation/test_3DLoMatch.py:45-48  the torch form of the 3DLoMatch caller: argmax of the inner products (metric="ip");
                                        equal to the arg-min above only for exactly unit-length descriptors

``build_correspondences`` returns exactly the three tensors the forward consumes (plus the index pairs), batched as
``[1, Nc, .]`` like the reference's data loader hands them over.  The Ns x Nt distance matrix is never materialised:
one fused MFMA GEMM + arg-min kernel (csrc/match.hip).  GPU only; no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import _lib


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (pointdsc_amd has no CPU path)")
    return t.detach().to(torch.float32).contiguous()


def match_descriptors(src_desc: torch.Tensor, tgt_desc: torch.Tensor, want_dist: bool = False, metric: str = "l2"):
    """nn_idx [Ns] int32 (and nn_dist [Ns]) of ``argmin(sqrt(2 - 2 * src_desc @ tgt_desc.T + 1e-6), axis=1)`` (metric "l2"), or of
    ``argmax(src_desc @ tgt_desc.T, dim=-1)`` (metric "ip", evaluation/test_3DLoMatch.py:45-46; nn_dist is then the inner product)."""
    if metric not in ("l2", "ip"):
        raise ValueError(f'metric must be "l2" or "ip", got {metric!r}')
    lib = _lib.load()
    s, t = _chk(src_desc, "src_desc"), _chk(tgt_desc, "tgt_desc")
    ns, d = s.shape
    nt = t.shape[0]
    if t.shape[1] != d:
        raise ValueError("descriptor lengths differ")
    idx = torch.empty(ns, device=s.device, dtype=torch.int32)
    dist = torch.empty(ns, device=s.device, dtype=torch.float32) if want_dist else None
    nb = int(lib.pdsc_match_scratch_bytes(ns, nt))
    scratch = torch.empty(nb, device=s.device, dtype=torch.uint8)
    with torch.cuda.device(s.device):
        fn = lib.pdsc_match_descriptors if metric == "l2" else lib.pdsc_match_descriptors_ip
        _lib.check(fn(_p(s), _p(t), ns, nt, d, _p(idx), _p(dist), _p(scratch), nb, torch.cuda.current_stream().cuda_stream),
                   "pdsc_match_descriptors" + ("" if metric == "l2" else "_ip"))
    return (idx, dist) if want_dist else idx


def build_correspondences(src_desc: torch.Tensor, tgt_desc: torch.Tensor, src_keypts: torch.Tensor,
                          tgt_keypts: torch.Tensor, use_mutual: bool = False, metric: str = "l2") -> Dict[str, torch.Tensor]:
    """descriptors [Ns,D] / [Nt,D] (L2-normalised), keypoints [Ns,3] / [Nt,3]  ->
    {'corr_pos' [1,Nc,6], 'src_keypts' [1,Nc,3], 'tgt_keypts' [1,Nc,3], 'corr' [Nc,2] int32}.
    ``use_mutual`` keeps only mutual nearest neighbours (ThreeDMatch.py:286-288); Nc is then data dependent, which costs
    the one host synchronisation that reads it.  ``metric="ip"`` matches by the largest inner product, as the 3DLoMatch caller
    does (evaluation/test_3DLoMatch.py:45-48)."""
    lib = _lib.load()
    skp, tkp = _chk(src_keypts, "src_keypts"), _chk(tgt_keypts, "tgt_keypts")
    s2t = match_descriptors(src_desc, tgt_desc, metric=metric)
    t2s = match_descriptors(tgt_desc, src_desc, metric=metric) if use_mutual else None
    ns, dev = s2t.shape[0], s2t.device
    corr = torch.empty(ns, 2, device=dev, dtype=torch.int32)
    count = torch.empty(1, device=dev, dtype=torch.int32)
    corr_pos = torch.empty(ns, 6, device=dev, dtype=torch.float32)
    src_sel = torch.empty(ns, 3, device=dev, dtype=torch.float32)
    tgt_sel = torch.empty(ns, 3, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.pdsc_select_correspondences(_p(s2t), _p(t2s), ns, _p(corr), _p(count), st), "pdsc_select_correspondences")
        _lib.check(lib.pdsc_build_corr_pos(_p(skp), _p(tkp), _p(corr), _p(count), _p(corr_pos), _p(src_sel), _p(tgt_sel), st),
                   "pdsc_build_corr_pos")
    nc = int(count.item()) if use_mutual else ns
    return {"corr_pos": corr_pos[None, :nc], "src_keypts": src_sel[None, :nc], "tgt_keypts": tgt_sel[None, :nc],
            "corr": corr[:nc]}
