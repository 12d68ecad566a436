"""CPU oracle for the PointDSC test-time hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement, in plain fp32 torch-CPU tensor algebra, of what the
reference computes in ``PointDSC.forward`` with ``'testing' in data``:

    /root/reference/models/PointDSC.py:128-197   forward (testing mode, and the validation forward without 'testing')
    /root/reference/models/PointDSC.py:9-77      NonLocalBlock / NonLocalNet
    /root/reference/models/PointDSC.py:199-217   pick_seeds
    /root/reference/models/PointDSC.py:234-336   cal_seed_trans
    /root/reference/models/PointDSC.py:338-358   cal_leading_eigenvector (power)
    /root/reference/models/PointDSC.py:403-438   post_refinement
    /root/reference/models/common.py:7-69        rigid_transform_3d, knn
    /root/reference/utils/SE3.py:43-57,73-96     transform, integrate_trans

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
it, and only as the checker.  The product path (``pointdsc_amd``) never imports this module.

Pinning status: the reference ships no tests or golden vectors, so the oracle is
pinned against the reference *implementation itself*: ``oracle/check_against_reference.py`` imports
``/root/reference`` in the build container, runs both on identical seeded inputs and asserts the
agreement recorded in ``tests/golden/PINNING.json``; the same script writes the fixtures under
``tests/golden/`` that travel to the GPU box (where ``/root/reference`` does not exist).

Deliberate, documented differences from the raw reference (both only matter on exact ties, where the
reference itself is backend-defined):
  * ``pick_seeds`` orders equal keys by ascending index (``torch.argsort`` is unstable);
  * ``knn`` orders equal distances by ascending index (``torch.topk`` tie order is unspecified).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

BN_EPS = 1e-5  # nn.BatchNorm1d default, used by every BN in the reference model

# Timing mode (bench.py's cpu_baseline leg ONLY): the N x N distance stages call the reference's own ops
# (torch.norm of the broadcast difference, models/PointDSC.py:151-152,327) instead of the fp64-emulated fma / sqrt
# chain below, which exists to make thresholds bit-reproducible and costs ~1.35x the reference's run time.  Results
# in timing mode agree with the exact mode to the last ulp or two of a distance; parity checks never use it.
_TIMING_MODE = False


def set_timing_mode(on: bool) -> None:
    global _TIMING_MODE
    _TIMING_MODE = bool(on)



# --------------------------------------------------------------------------------------------------
# a-1: pairwise distances and the spatial-consistency matrix (reference models/PointDSC.py:150-153)
# --------------------------------------------------------------------------------------------------
def _fma32(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """fp32 fused multiply-add emulated through fp64 (the 48-bit product is exact in fp64)."""
    return (a.double() * b.double() + c.double()).float()


def _sqrt32(x: torch.Tensor) -> torch.Tensor:
    """Correctly rounded fp32 sqrt (sqrt in fp64 of an fp32 value rounds correctly to fp32).
    torch.sqrt on CPU fp32 tensors is NOT correctly rounded (0.7 % of values are 1 ulp off, measured here),
    while torch.norm's final sqrt, numpy's and the GPU's v_sqrt-based sqrtf sequence are."""
    return torch.sqrt(x.double()).float()


def pairwise_dist(x: torch.Tensor) -> torch.Tensor:
    """``torch.norm(x[:, None] - x[None], dim=-1)`` for x [N,3].

    torch's CPU vector-norm kernel evaluates sqrt(fma(dz,dz,fma(dy,dy,dx*dx))) (measured bit-for-bit
    in this container, see oracle/check_against_reference.py); the same chain is what the HIP
    kernels use (``fmaf``), so distance thresholds are decided on identical bits.
    """
    d = x[:, None, :] - x[None, :, :]
    if _TIMING_MODE:
        return torch.norm(d, dim=-1)
    dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
    return _sqrt32(_fma32(dz, dz, _fma32(dy, dy, dx * dx)))


def spatial_compat(src: torch.Tensor, tgt: torch.Tensor, sigma_spat: torch.Tensor):
    """Returns (src_dist [N,N], compat [N,N]); compat = clamp(1 - (ds-dt)^2 / sigma^2, min=0)."""
    src_dist = pairwise_dist(src)
    diff = src_dist - pairwise_dist(tgt)
    s2 = sigma_spat.reshape(()).float() ** 2          # fp32 square, as `self.sigma_spat ** 2`
    compat = torch.clamp(1.0 - diff ** 2 / s2, min=0)
    return src_dist, compat


# --------------------------------------------------------------------------------------------------
# a-2/a-3: SCNonlocal encoder (reference models/PointDSC.py:27-45, 65-77)
# --------------------------------------------------------------------------------------------------
def _conv(sd: Dict[str, torch.Tensor], name: str, x: torch.Tensor) -> torch.Tensor:
    """Conv1d(kernel_size=1) on a [Cin, N] map: W[Cout,Cin] @ x + b."""
    w = sd[name + ".weight"][:, :, 0]
    return w @ x + sd[name + ".bias"][:, None]


def _bn(sd: Dict[str, torch.Tensor], name: str, x: torch.Tensor) -> torch.Tensor:
    """BatchNorm1d in eval mode: (x - running_mean) / sqrt(running_var + eps) * gamma + beta."""
    mean = sd[name + ".running_mean"][:, None]
    var = sd[name + ".running_var"][:, None]
    return (x - mean) / torch.sqrt(var + BN_EPS) * sd[name + ".weight"][:, None] + sd[name + ".bias"][:, None]


def nonlocal_block(sd, prefix: str, feat: torch.Tensor, compat: torch.Tensor, num_channels: int):
    """One SCNonlocal block on feat [C,N]; returns (res [C,N], message [C,N])."""
    q = _conv(sd, prefix + ".projection_q", feat)
    k = _conv(sd, prefix + ".projection_k", feat)
    v = _conv(sd, prefix + ".projection_v", feat)
    fa = (q.t() @ k) / (num_channels ** 0.5)                 # [N(o), N(i)]
    weight = torch.softmax(compat * fa, dim=-1)
    message = (weight @ v.t()).t().contiguous()              # [C, N(o)]
    m = _conv(sd, prefix + ".fc_message.0", message)
    m = torch.relu(_bn(sd, prefix + ".fc_message.1", m))
    m = _conv(sd, prefix + ".fc_message.3", m)
    m = torch.relu(_bn(sd, prefix + ".fc_message.4", m))
    m = _conv(sd, prefix + ".fc_message.6", m)
    return feat + m, message


def encoder(sd, corr_pos: torch.Tensor, compat: torch.Tensor, num_layers: int, num_channels: int,
            collect: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """NonLocalNet.forward for one pair; corr_pos [N,in_dim] -> features [N,C]."""
    feat = _conv(sd, "encoder.layer0", corr_pos.t())
    for i in range(num_layers):
        p = f"encoder.blocks.PointCN_layer_{i}"
        feat = torch.relu(_bn(sd, p + ".1", _conv(sd, p + ".0", feat)))
        feat, _ = nonlocal_block(sd, f"encoder.blocks.NonLocal_layer_{i}", feat, compat, num_channels)
        if collect is not None:
            collect.append(feat.t().contiguous())
    return feat.t().contiguous()


# --------------------------------------------------------------------------------------------------
# a-4: normalisation + confidence head (reference models/PointDSC.py:156, 107-113, 171)
# --------------------------------------------------------------------------------------------------
def l2_normalize(feat: torch.Tensor) -> torch.Tensor:
    """F.normalize(p=2, dim=-1): x / max(||x||_2, 1e-12)."""
    n = torch.sqrt((feat * feat).sum(-1, keepdim=True))
    return feat / torch.clamp(n, min=1e-12)


def classify(sd, feat: torch.Tensor) -> torch.Tensor:
    """classification MLP 128->32->32->1 on the un-normalised features [N,C] -> logits [N]."""
    x = feat.t()
    x = torch.relu(_conv(sd, "classification.0", x))
    x = torch.relu(_conv(sd, "classification.2", x))
    return _conv(sd, "classification.4", x)[0]


# --------------------------------------------------------------------------------------------------
# a-5: NMS seed selection (reference models/PointDSC.py:199-217)
# --------------------------------------------------------------------------------------------------
def nms_keys(src_dist: torch.Tensor, scores: torch.Tensor, radius: float) -> torch.Tensor:
    """key[i] = score[i] * is_local_max[i];  is_local_max[i] = all_j (s_i >= s_j  or  d_ij >= R)."""
    rel = (scores[:, None] >= scores[None, :]) | (src_dist >= radius)
    return scores * rel.all(dim=-1).float()


def pick_seeds(src_dist, scores, radius: float, max_num: int) -> torch.Tensor:
    """Top ``max_num`` indices by descending key, equal keys in ascending index order."""
    keys = nms_keys(src_dist, scores, radius)
    return torch.sort(keys, descending=True, stable=True).indices[:max_num]


# --------------------------------------------------------------------------------------------------
# a-6: feature-space kNN of the seeds (reference models/common.py:48-69, models/PointDSC.py:250-252)
# --------------------------------------------------------------------------------------------------
def knn_dist_rows(normed: torch.Tensor, seeds: torch.Tensor) -> torch.Tensor:
    """Rows ``seeds`` of the reference's ``2 - 2 * x @ x.T`` (only those rows are consumed)."""
    return 2 - 2 * (normed[seeds] @ normed.t())


def knn_of_seeds(normed: torch.Tensor, seeds: torch.Tensor, k: int) -> torch.Tensor:
    """[S,k] neighbour indices: the k+1 smallest distances, rank 0 dropped (models/common.py:68)."""
    dist = knn_dist_rows(normed, seeds)
    order = torch.sort(dist, dim=-1, descending=False, stable=True).indices
    return order[:, 1:k + 1].contiguous()


# --------------------------------------------------------------------------------------------------
# a-7: per-seed k x k compatibility (reference models/PointDSC.py:257-278)
# --------------------------------------------------------------------------------------------------
def seed_matrices(normed, src, tgt, knn_idx, sigma: torch.Tensor, sigma_spat: torch.Tensor):
    f = normed[knn_idx]                                   # [S,k,C]
    feat_m = torch.clamp(1 - (1 - f @ f.transpose(1, 2)) / sigma.reshape(()) ** 2, min=0)
    s, t = src[knn_idx], tgt[knn_idx]                     # [S,k,3]
    ds = _sqrt32(((s[:, :, None, :] - s[:, None, :, :]) ** 2).sum(-1))
    dt = _sqrt32(((t[:, :, None, :] - t[:, None, :, :]) ** 2).sum(-1))
    spat_m = torch.clamp(1 - (ds - dt) ** 2 / sigma_spat.reshape(()) ** 2, min=0)
    total = feat_m * spat_m
    k = total.shape[1]
    total[:, torch.arange(k), torch.arange(k)] = 0
    return total


# --------------------------------------------------------------------------------------------------
# a-8: power iteration (reference models/PointDSC.py:347-358)
# --------------------------------------------------------------------------------------------------
def allclose_flags(new: torch.Tensor, last: torch.Tensor, rtol=1e-5, atol=1e-8) -> torch.Tensor:
    """Per-matrix form of torch.allclose(new, last): |new-last| <= atol + rtol*|last| everywhere."""
    return ((new - last).abs() <= atol + rtol * last.abs()).all(dim=-1)


def power_iteration(M: torch.Tensor, num_iterations: int):
    """Leading eigenvector of every M[s] (k x k); the early exit is global over all s.

    Returns (vec [S,k], iterations_run).
    """
    v = torch.ones_like(M[:, :, 0])
    last = v
    ran = 0
    for _ in range(num_iterations):
        v = torch.einsum("sij,sj->si", M, v)
        v = v / (torch.sqrt((v * v).sum(-1, keepdim=True)) + 1e-6)
        ran += 1
        if bool(allclose_flags(v, last).all()):
            break
        last = v
    return v, ran


# --------------------------------------------------------------------------------------------------
# a-9: weighted Procrustes (reference models/common.py:7-45, utils/SE3.py:73-96)
# --------------------------------------------------------------------------------------------------
def rigid_transform_3d(A: torch.Tensor, B: torch.Tensor, weights: Optional[torch.Tensor] = None,
                       weight_threshold: float = 0.0) -> torch.Tensor:
    """A,B [bs,n,3], weights [bs,n] -> [bs,4,4] with p_B ~ R p_A + t.  Does not mutate ``weights``."""
    bs = A.shape[0]
    w = torch.ones_like(A[:, :, 0]) if weights is None else weights.clone()
    w[w < weight_threshold] = 0
    wsum = w.sum(dim=1, keepdim=True)[:, :, None] + 1e-6
    cA = (A * w[:, :, None]).sum(dim=1, keepdim=True) / wsum
    cB = (B * w[:, :, None]).sum(dim=1, keepdim=True) / wsum
    Am, Bm = A - cA, B - cB
    H = Am.transpose(1, 2) @ (w[:, :, None] * Bm)          # == Am^T diag(w) Bm
    U, _, Vh = torch.linalg.svd(H)
    V = Vh.transpose(1, 2)
    d = torch.det(V @ U.transpose(1, 2))
    D = torch.eye(3)[None].repeat(bs, 1, 1)
    D[:, 2, 2] = d
    R = V @ D @ U.transpose(1, 2)
    t = cB.transpose(1, 2) - R @ cA.transpose(1, 2)
    T = torch.eye(4)[None].repeat(bs, 1, 1)
    T[:, :3, :3] = R
    T[:, :3, 3:4] = t
    return T


def transform(pts: torch.Tensor, trans: torch.Tensor) -> torch.Tensor:
    """pts [N,3], trans [4,4] -> R pts + t."""
    return (trans[:3, :3] @ pts.t() + trans[:3, 3:4]).t()


# --------------------------------------------------------------------------------------------------
# a-10: hypothesis scoring (reference models/PointDSC.py:325-335)
# --------------------------------------------------------------------------------------------------
def residuals(trans: torch.Tensor, src: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
    """trans [S,4,4] -> L2 [S,N] = || R_s src + t_s - tgt ||."""
    pred = torch.einsum("snm,mk->snk", trans[:, :3, :3], src.t()) + trans[:, :3, 3:4]
    d = pred.permute(0, 2, 1) - tgt[None]
    if _TIMING_MODE:
        return torch.norm(d, dim=-1)
    dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
    return _sqrt32(_fma32(dz, dz, _fma32(dy, dy, dx * dx)))


def score_hypotheses(trans, src, tgt, inlier_threshold: float):
    L2 = residuals(trans, src, tgt)
    counts = (L2 < inlier_threshold).sum(dim=-1)
    fitness = (L2 < inlier_threshold).float().mean(dim=-1)
    best = int(torch.argmax(fitness))
    labels = (L2[best] < inlier_threshold).float()
    return counts, best, labels


# --------------------------------------------------------------------------------------------------
# a-11: post refinement (reference models/PointDSC.py:403-438)
# --------------------------------------------------------------------------------------------------
def refine_threshold(inlier_threshold: float) -> float:
    """reference :415-418 -- exact float equality with 0.10 selects the 3DMatch schedule."""
    return 0.10 if inlier_threshold == 0.10 else 1.2


def post_refinement(trans: torch.Tensor, src, tgt, inlier_threshold: float, max_iters: int = 20):
    """trans [4,4]; returns (refined [4,4], iterations that re-solved)."""
    thr = refine_threshold(inlier_threshold)
    prev = 0
    solved = 0
    for _ in range(max_iters):
        d = transform(src, trans) - tgt
        L2 = _sqrt32(_fma32(d[:, 2], d[:, 2], _fma32(d[:, 1], d[:, 1], d[:, 0] * d[:, 0])))
        inl = L2 < thr
        n = int(inl.sum())
        if abs(n - prev) < 1:
            break
        prev = n
        w = 1 / (1 + (L2 / thr) ** 2)
        trans = rigid_transform_3d(src[None, inl], tgt[None, inl], w[None, inl])[0]
        solved += 1
    return trans, solved


# --------------------------------------------------------------------------------------------------
# whole path
# --------------------------------------------------------------------------------------------------
def forward_testing(sd: Dict[str, torch.Tensor], corr_pos, src_keypts, tgt_keypts, *,
                    num_layers=12, num_channels=128, num_iterations=10, ratio=0.1,
                    inlier_threshold=0.10, k=40, nms_radius=0.10, return_stages=False):
    """Batched entry: corr_pos [bs,N,6], src/tgt [bs,N,3]; bs>1 == independent bs=1 calls
    (SURVEY.md section 8a note 6).  ``sigma``/``sigma_spat`` are read from ``sd`` like the reference
    reads its Parameters (note 3)."""
    sd = {k_: v.detach().float().cpu() for k_, v in sd.items()}
    outs_T, outs_L, stages = [], [], []
    for b in range(corr_pos.shape[0]):
        st = _forward_one(sd, corr_pos[b].float().cpu(), src_keypts[b].float().cpu(), tgt_keypts[b].float().cpu(),
                          num_layers, num_channels, num_iterations, ratio, inlier_threshold, k, nms_radius)
        outs_T.append(st["final_trans"])
        outs_L.append(st["final_labels"])
        stages.append(st)
    res = {"final_trans": torch.stack(outs_T), "final_labels": torch.stack(outs_L), "M": None}
    if return_stages:
        res["stages"] = stages
    return res


def _forward_one(sd, corr_pos, src, tgt, num_layers, num_channels, num_iterations, ratio,
                 inlier_threshold, k, nms_radius):
    N = corr_pos.shape[0]
    st = {}
    src_dist, compat = spatial_compat(src, tgt, sd["sigma_spat"])
    layer_feats: List[torch.Tensor] = []
    feat = encoder(sd, corr_pos, compat, num_layers, num_channels, collect=layer_feats)
    normed = l2_normalize(feat)
    conf = classify(sd, feat)
    num_seeds = int(N * ratio)                                # python double arithmetic, note 8
    keys = nms_keys(src_dist, conf, nms_radius)
    seeds = torch.sort(keys, descending=True, stable=True).indices[:num_seeds]
    kk = min(k, N - 1)
    knn_idx = knn_of_seeds(normed, seeds, kk)
    M = seed_matrices(normed, src, tgt, knn_idx, sd["sigma"], sd["sigma_spat"])
    vec, iters = power_iteration(M, num_iterations)
    w = vec / (vec.sum(-1, keepdim=True) + 1e-6)
    seed_trans = rigid_transform_3d(src[knn_idx], tgt[knn_idx], w)
    counts, best, labels = score_hypotheses(seed_trans, src, tgt, inlier_threshold)
    initial = seed_trans[best]
    final, solved = post_refinement(initial, src, tgt, inlier_threshold)
    st.update(src_dist=src_dist, compat=compat, layer_feats=layer_feats, feat=feat, normed=normed,
              confidence=conf, nms_keys=keys, seeds=seeds, knn_idx=knn_idx, seed_M=M, eigvec=vec,
              power_iters=iters, seed_weights=w, seed_trans=seed_trans, counts=counts, best=best,
              initial_trans=initial, refine_solves=solved, final_trans=final, final_labels=labels)
    return st


# --------------------------------------------------------------------------------------------------
# validation forward: no 'testing' key, module in eval() mode (reference models/PointDSC.py:158-163,:176,:190-191;
# caller libs/trainer.py:158-222).  SURVEY.md section 8 f-1.
# --------------------------------------------------------------------------------------------------
def feature_compat(normed: torch.Tensor, sigma: torch.Tensor) -> torch.Tensor:
    """M = clamp(1 - (1 - F F^T) / sigma^2, 0, 1) with a zero diagonal, for one pair: normed [N,C] -> [N,N]."""
    M = normed @ normed.t()
    M = torch.clamp(1 - (1 - M) / sigma.reshape(()) ** 2, min=0, max=1)
    n = M.shape[0]
    M[torch.arange(n), torch.arange(n)] = 0
    return M


def forward_validation(sd: Dict[str, torch.Tensor], corr_pos, src_keypts, tgt_keypts, *,
                       num_layers=12, num_channels=128, num_iterations=10, ratio=0.1,
                       inlier_threshold=0.10, k=40, nms_radius=0.10, return_stages=False):
    """Batched: corr_pos [bs,N,6].  Differences from the testing forward: M is built, seeds are the top int(N*ratio)
    correspondences by confidence (no NMS; equal logits by ascending index), the power iteration's early exit is
    taken over the seeds of ALL pairs of the batch (one torch.allclose over [bs*S, k]), there is no refinement and
    the returned labels are the logits."""
    sd = {k_: v.detach().float().cpu() for k_, v in sd.items()}
    bs, N = corr_pos.shape[0], corr_pos.shape[1]
    num_seeds = int(N * ratio)
    kk = min(k, N - 1)
    per = []
    for b in range(bs):
        corr, src, tgt = corr_pos[b].float().cpu(), src_keypts[b].float().cpu(), tgt_keypts[b].float().cpu()
        _, compat = spatial_compat(src, tgt, sd["sigma_spat"])
        feat = encoder(sd, corr, compat, num_layers, num_channels)
        normed = l2_normalize(feat)
        conf = classify(sd, feat)
        M = feature_compat(normed, sd["sigma"])
        seeds = torch.sort(conf, descending=True, stable=True).indices[:num_seeds]
        knn_idx = knn_of_seeds(normed, seeds, kk)
        seed_M = seed_matrices(normed, src, tgt, knn_idx, sd["sigma"], sd["sigma_spat"])
        per.append(dict(src=src, tgt=tgt, feat=feat, normed=normed, confidence=conf, M=M, seeds=seeds, knn_idx=knn_idx, seed_M=seed_M))
    vec_all, iters = power_iteration(torch.cat([p["seed_M"] for p in per], dim=0), num_iterations)   # global early exit
    outs_T, outs_L, outs_M = [], [], []
    for b, p in enumerate(per):
        vec = vec_all[b * num_seeds:(b + 1) * num_seeds]
        w = vec / (vec.sum(-1, keepdim=True) + 1e-6)
        seed_trans = rigid_transform_3d(p["src"][p["knn_idx"]], p["tgt"][p["knn_idx"]], w)
        counts, best, _ = score_hypotheses(seed_trans, p["src"], p["tgt"], inlier_threshold)
        p.update(eigvec=vec, seed_trans=seed_trans, counts=counts, best=best, power_iters=iters)
        outs_T.append(seed_trans[best]); outs_L.append(p["confidence"]); outs_M.append(p["M"])
    res = {"final_trans": torch.stack(outs_T), "final_labels": torch.stack(outs_L), "M": torch.stack(outs_M)}
    if return_stages:
        res["stages"] = per
    return res


# --------------------------------------------------------------------------------------------------
# evaluation metric used for the parity report (reference libs/loss.py:44-51)
# --------------------------------------------------------------------------------------------------
def registration_errors(trans: torch.Tensor, gt: torch.Tensor):
    """(RE degrees, TE centimetres) for [4,4] transforms, the reference's recall definition."""
    R, t, gR, gt_t = trans[:3, :3], trans[:3, 3], gt[:3, :3], gt[:3, 3]
    re = torch.acos(torch.clamp((torch.trace(R.t() @ gR) - 1) / 2.0, min=-1, max=1)) * 180 / math.pi
    te = torch.sqrt(((t - gt_t) ** 2).sum()) * 100
    return float(re), float(te)
