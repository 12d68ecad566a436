"""Classical baselines of the reference that share the hot path's kernels.

``SM`` mirrors ``baseline_scripts/baseline_3DMatch.py:19-53`` (spectral matching: N x N compatibility matrix + 10 power
iterations + top-10 % selection + weighted Procrustes) with the reference's argument meaning; the arithmetic runs in
``csrc/spectral.hip`` (matrix written once, one HBM-bound mat-vec launch per iteration).  GPU only.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def SM(corr: torch.Tensor, src_keypts: torch.Tensor, tgt_keypts: torch.Tensor, inlier_threshold: float, top_ratio: float = 0.1,
       num_iterations: int = 10, return_eig: bool = False, form: str = "auto"):
    """corr [N,6] (or [bs,N,6]) centred correspondence coordinates, src/tgt_keypts [bs,N,3] ->
    (pred_trans [bs,4,4], pred_labels [bs,N]) like the reference's SM(corr, src_keypts, tgt_keypts, args, top_ratio)
    with ``args.inlier_threshold`` passed explicitly; bs > 1 = independent pairs.  ``form``: "streaming" (matrix in HBM, any N),
    "resident" (opt-in: the matrix stays in the chip's vector registers, N <= 5120; needs the GPU to itself -- cooperative launch,
    refused under graph capture) or "auto" (= "streaming", always); same bits either way."""
    forms = {"auto": 0, "streaming": 1, "resident": 2}
    if form not in forms:
        raise ValueError(f"form must be one of {sorted(forms)}, got {form!r}")
    lib = _lib.load()
    if not corr.is_cuda:
        raise RuntimeError("pointdsc_amd has no CPU path: move the tensors to the GPU")
    src = src_keypts.detach().to(torch.float32).contiguous()
    tgt = tgt_keypts.detach().to(torch.float32).contiguous()
    bs, n = src.shape[0], src.shape[1]
    c = corr.detach().to(torch.float32).reshape(bs, n, 6).contiguous()
    dev = c.device
    num_top = int(n * top_ratio)                               # python double arithmetic, as the reference (:47)
    trans = torch.empty(bs, 4, 4, device=dev, dtype=torch.float32)
    labels = torch.empty(bs, n, device=dev, dtype=torch.float32)
    eig = torch.empty(bs, n, device=dev, dtype=torch.float32)
    nb = int(lib.pdsc_sm_workspace_bytes(bs, n))
    ws = torch.empty(nb, device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        rc = lib.pdsc_sm_baseline_form(C.c_void_p(c.data_ptr()), C.c_void_p(src.data_ptr()), C.c_void_p(tgt.data_ptr()),
                                       float(inlier_threshold), num_top, int(num_iterations), C.c_void_p(trans.data_ptr()),
                                       C.c_void_p(labels.data_ptr()), C.c_void_p(eig.data_ptr()), C.c_void_p(ws.data_ptr()), nb,
                                       bs, n, forms[form], torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "pdsc_sm_baseline")
    return (trans, labels, eig) if return_eig else (trans, labels)


CONFIDENCE_METHODS = {"eig_value": 0, "eig_value_ratio": 1, "xMx": 2}


def cal_confidence(M: torch.Tensor, leading_eig: torch.Tensor, method: str = "eig_value", num_iterations: int = 10) -> torch.Tensor:
    """``PointDSC.cal_confidence`` (reference models/PointDSC.py:366-401): M [bs,N,N] (or [bs,N,ld], ld >= N a multiple of 4,
    e.g. the matrix of ``ops.spatial_compat``), leading_eig [bs,N] -> confidence [bs,1] with the reference's three
    methods; ``num_iterations`` = the module's power-iteration count (used by 'eig_value_ratio')."""
    lib = _lib.load()
    if method not in CONFIDENCE_METHODS:
        raise ValueError(f"method must be one of {sorted(CONFIDENCE_METHODS)}")
    if not M.is_cuda:
        raise RuntimeError("pointdsc_amd has no CPU path: move the tensors to the GPU")
    v = leading_eig.detach().to(torch.float32).contiguous()
    bs, n = v.shape
    m = M.detach().to(torch.float32)
    if m.shape[-1] % 4 != 0:                                 # float4 row loads
        m = torch.nn.functional.pad(m, (0, 4 - m.shape[-1] % 4))
    m = m.contiguous()
    ld = m.shape[-1]
    conf = torch.empty(bs, device=m.device, dtype=torch.float32)
    nb = int(lib.pdsc_cal_confidence_workspace_bytes(bs, n))
    ws = torch.empty(nb, device=m.device, dtype=torch.uint8)
    with torch.cuda.device(m.device):
        rc = lib.pdsc_cal_confidence(C.c_void_p(m.data_ptr()), ld, C.c_void_p(v.data_ptr()), CONFIDENCE_METHODS[method],
                                     int(num_iterations), C.c_void_p(conf.data_ptr()), C.c_void_p(ws.data_ptr()), nb, bs, n,
                                     torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "pdsc_cal_confidence")
    return conf[:, None]
