"""Stage-level tensor wrappers over the C-ABI (one per entry point of include/pointdsc_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; all arithmetic happens in
libpointdsc_hip.so.  Every wrapper validates device/dtype/contiguity, allocates outputs with torch and
enqueues on ``torch.cuda.current_stream()``.  Names and argument meaning follow the reference functions
they stand in for (``rigid_transform_3d``, ``knn`` ... in /root/reference/models/common.py).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _on_device(fn):
    """Run the wrapper with the first CUDA tensor argument's device current (the library launches on the calling
    thread's current device and on that device's current stream), like PointDSC.forward does."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        for a in list(args) + list(kwargs.values()):
            if torch.is_tensor(a) and a.is_cuda:
                with torch.cuda.device(a.device):
                    return fn(*args, **kwargs)
            if isinstance(a, (tuple, list)) and a and torch.is_tensor(a[0]) and a[0].is_cuda:
                with torch.cuda.device(a[0].device):
                    return fn(*args, **kwargs)
        return fn(*args, **kwargs)
    return wrapped


def _chk(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (pointdsc_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def compat_ld(n: int) -> int:
    return int(_lib.load().pdsc_compat_ld(n))


@_on_device
def spatial_compat(src_keypts: torch.Tensor, tgt_keypts: torch.Tensor, sigma_spat: torch.Tensor,
                   want_dist: bool = False):
    """[bs,N,3] x2 -> compat [bs,N,ld] (view [..., :N] is the reference matrix), optional src_dist."""
    lib = _lib.load()
    src, tgt = _chk(src_keypts, "src_keypts"), _chk(tgt_keypts, "tgt_keypts")
    sig = _chk(sigma_spat.reshape(-1), "sigma_spat")
    bs, n = src.shape[0], src.shape[1]
    ld = compat_ld(n)
    compat = torch.empty(bs, n, ld, device=src.device, dtype=torch.float32)
    dist = torch.empty_like(compat) if want_dist else None
    _lib.check(lib.pdsc_spatial_compat(_p(src), _p(tgt), _p(sig), _p(compat), _p(dist), ld, bs, n, _stream()),
               "pdsc_spatial_compat")
    return (compat, dist) if want_dist else compat


@_on_device
def spatial_compat_u16(src_keypts: torch.Tensor, tgt_keypts: torch.Tensor, sigma_spat: torch.Tensor,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[bs,N,3] x2 -> unorm16 compat [bs,N,ld] (int16 storage of uint16 bits) in the attention kernel's tile order
    (pdsc_spatial_compat_u16); decode with `decode_compat_u16`."""
    lib = _lib.load()
    src, tgt = _chk(src_keypts, "src_keypts"), _chk(tgt_keypts, "tgt_keypts")
    sig = _chk(sigma_spat.reshape(-1), "sigma_spat")
    bs, n = src.shape[0], src.shape[1]
    ld = compat_ld(n)
    if out is None:
        out = torch.empty(bs, n, ld, device=src.device, dtype=torch.int16)
    _lib.check(lib.pdsc_spatial_compat_u16(_p(src), _p(tgt), _p(sig), _p(out), ld, bs, n, _stream()), "pdsc_spatial_compat_u16")
    return out


def decode_compat_u16(c16: torch.Tensor, n: int) -> torch.Tensor:
    """unorm16 tile-order matrix -> fp32 [bs,N,N] in natural column order (tests)."""
    bs, rows, ld = c16.shape
    u = (c16.to(torch.int32) & 0xFFFF).to(torch.float32) / 65535.0
    j = torch.arange(ld, device=c16.device)
    r = j & 31
    pos = (j & ~31) + 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3)
    return u[:, :, pos][:, :, :n]


@_on_device
def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False,
           residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [M,K] @ weight[Nout,K]^T (+bias)(relu)(+residual) -> [M,Nout]."""
    lib = _lib.load()
    x, weight = _chk(x, "x"), _chk(weight, "weight")
    m, k = x.shape
    nout = weight.shape[0]
    y = torch.empty(m, nout, device=x.device, dtype=torch.float32)
    b = _chk(bias, "bias") if bias is not None else None
    r = _chk(residual, "residual") if residual is not None else None
    _lib.check(lib.pdsc_linear(_p(x), k, _p(weight), _p(b), _p(r), nout, _p(y), nout, m, k, nout, int(relu), _stream()),
               "pdsc_linear")
    return y


@_on_device
def classifier_hidden(feat: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """feat [M,128] -> h2 [M,32] = relu(w2 relu(w1 feat + b1) + b2) in one launch (pdsc_classifier_hidden): bit-identical to
    linear(linear(feat, w1, b1, relu=True), w2, b2, relu=True)."""
    lib = _lib.load()
    x = _chk(feat, "feat")
    w1, b1, w2, b2 = _chk(w1, "w1"), _chk(b1, "b1"), _chk(w2, "w2"), _chk(b2, "b2")
    assert x.shape[1] == 128 and w1.shape == (32, 128) and w2.shape == (32, 32) and b1.shape == (32,) and b2.shape == (32,)
    h2 = torch.empty(x.shape[0], 32, device=x.device, dtype=torch.float32)
    _lib.check(lib.pdsc_classifier_hidden(_p(x), _p(w1), _p(b1), _p(w2), _p(b2), _p(h2), x.shape[0], _stream()), "pdsc_classifier_hidden")
    return h2


@_on_device
def layer0(corr_pos: torch.Tensor, w0_padded: torch.Tensor, b0: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    x = _chk(corr_pos, "corr_pos").reshape(-1, corr_pos.shape[-1])
    w, b = _chk(w0_padded, "w0"), _chk(b0, "b0")
    y = torch.empty(x.shape[0], 128, device=x.device, dtype=torch.float32)
    _lib.check(lib.pdsc_layer0(_p(x), x.shape[1], _p(w), _p(b), _p(y), x.shape[0], _stream()), "pdsc_layer0")
    return y


@_on_device
def layer_fused(msg, res, feat_in, tail_w=None, head_w=None, want_feat=False):
    """Fused point-wise chain (pdsc_layer_fused).  tail_w = (w1,b1,w2,b2,w3,b3) folded fc_message of layer i,
    head_w = (wp,bp,wq,bq) folded PointCN / stacked qkv of layer i+1.  Returns (feat or None, featB or None, qkv or None)."""
    lib = _lib.load()
    src = msg if msg is not None else feat_in
    m, dev = src.shape[0], src.device
    tail = [_chk(w, "tail_w") for w in tail_w] if tail_w is not None else [None] * 6
    head = [_chk(w, "head_w") for w in head_w] if head_w is not None else [None] * 4
    feat = torch.empty(m, 128, device=dev, dtype=torch.float32) if (want_feat or head_w is None) else None
    featB = torch.empty(m, 128, device=dev, dtype=torch.float32) if head_w is not None else None
    qkv = torch.empty(m, 384, device=dev, dtype=torch.float32) if head_w is not None else None
    args = [_p(_chk(msg, "msg")) if msg is not None else None, _p(_chk(res, "res")) if res is not None else None,
            _p(_chk(feat_in, "feat_in")) if feat_in is not None else None, _p(feat), _p(featB), _p(qkv)]
    args += [_p(w) for w in tail] + [_p(w) for w in head]
    _lib.check(lib.pdsc_layer_fused(*args, m, _stream()), "pdsc_layer_fused")
    return feat, featB, qkv


@_on_device
def sc_attention(qkv: torch.Tensor, compat: torch.Tensor, bs: int, n: int, nsplit: int = 0) -> torch.Tensor:
    """qkv [bs*N,384] (q pre-scaled by log2(e)/sqrt(128)), compat [bs,N,ld] -> msg [bs*N,128]."""
    lib = _lib.load()
    qkv, compat = _chk(qkv, "qkv"), _chk(compat, "compat")
    ld = compat.shape[-1]
    msg = torch.empty(bs * n, 128, device=qkv.device, dtype=torch.float32)
    nb = int(lib.pdsc_attention_scratch_bytes(bs, n, nsplit))
    scratch = torch.empty(max(nb, 16), device=qkv.device, dtype=torch.uint8)
    _lib.check(lib.pdsc_sc_attention(_p(qkv), _p(compat), ld, _p(msg), _p(scratch), nb, bs, n, nsplit, _stream()),
               "pdsc_sc_attention")
    return msg


@_on_device
def pack_qkv_split(qkv: torch.Tensor, bs: int, n: int):
    """fp32 (q|k|v) rows [bs*N,384] -> (q_split, kv_tiles) byte tensors in the layout of csrc/split_layout.h."""
    lib = _lib.load()
    qkv = _chk(qkv, "qkv")
    qs = torch.empty(int(lib.pdsc_split_q_bytes(bs, n)), device=qkv.device, dtype=torch.uint8)
    kv = torch.zeros(int(lib.pdsc_split_kv_bytes(bs, n)), device=qkv.device, dtype=torch.uint8)
    _lib.check(lib.pdsc_pack_qkv_split(_p(qkv), _p(qs), _p(kv), bs, n, _stream()), "pdsc_pack_qkv_split")
    return qs, kv


@_on_device
def sc_attention_split(q_split: torch.Tensor, kv_tiles: torch.Tensor, compat: torch.Tensor, bs: int, n: int,
                       nsplit: int = 0, merge: bool = True, layout: str = "rows"):
    """Split-precision (fp16 hi/lo, three MFMAs per operand pair) attention on the packed streams -> msg [bs*N,128].
    merge=False (needs a key split > 1): returns (scratch, nsplit) with the un-merged partials for layer_fused_x3;
    layout="pf": those partials in point-fragment order (csrc/split_layout.h) for layer_fused_io."""
    lib = _lib.load()
    c16 = compat.dtype == torch.int16                  # unorm16 matrix of spatial_compat_u16
    compat = _chk(compat, "compat", torch.int16 if c16 else torch.float32)
    qs, kv = _chk(q_split, "q_split", torch.uint8), _chk(kv_tiles, "kv_tiles", torch.uint8)
    if nsplit <= 0:
        nsplit = int(lib.pdsc_attention_split_default_split(bs, n))
    msg = torch.empty(bs * n, 128, device=compat.device, dtype=torch.float32) if merge else None
    nb = int(lib.pdsc_attention_split_scratch_bytes(bs, n, nsplit))
    scratch = torch.empty(max(nb, 16), device=compat.device, dtype=torch.uint8)
    if layout == "pf":       # un-merged partials in point-fragment order (enum pdsc_partial_layout)
        assert not merge, "point-fragment partials are merged by the fused layer kernel"
        _lib.check(lib.pdsc_sc_attention_split_partials(_p(qs), _p(kv), _p(compat), 1 if c16 else 0, compat.shape[-1], _p(scratch), nb,
                                                        bs, n, nsplit, 1, _stream()), "pdsc_sc_attention_split_partials")
        return scratch, nsplit
    fn = lib.pdsc_sc_attention_split_u16 if c16 else lib.pdsc_sc_attention_split
    _lib.check(fn(_p(qs), _p(kv), _p(compat), compat.shape[-1], _p(msg), _p(scratch), nb, bs, n, nsplit, _stream()),
               "pdsc_sc_attention_split")
    return msg if merge else (scratch, nsplit)


LAYER_GEMMS = {"f32": 0, "h3": 1}      # enum pdsc_layer_gemm


@_on_device
def frag_weights_tail(tail_w, gemm: str = "f32") -> torch.Tensor:
    """(fc1 w, b, fc2 w, b, fc3 w, b) fp32 [out][in] -> the fragment-ordered tail stream of pdsc_layer_fused_frag(_fmt);
    gemm = "h3": the chunks as fp16 hi / scaled-lo pairs (enum pdsc_layer_gemm)."""
    lib = _lib.load()
    out = torch.empty(int(lib.pdsc_wfrag_tail_bytes()), dtype=torch.uint8, device=tail_w[0].device)
    _lib.check(lib.pdsc_wfrag_build_tail_fmt(*[_p(_chk(w, "tail_w")) for w in tail_w], _p(out), LAYER_GEMMS[gemm], _stream()),
               "pdsc_wfrag_build_tail_fmt")
    return out


@_on_device
def frag_weights_head(head_w, gemm: str = "f32") -> torch.Tensor:
    """(pcn w, b, qkv w, b): pcn kept fp32 (gemm "f32") or fp16 hi / scaled lo ("h3"), q|k|v -> fp16 hi / lo, biases as
    one more k-step -> the head stream."""
    lib = _lib.load()
    out = torch.empty(int(lib.pdsc_wfrag_head_bytes()), dtype=torch.uint8, device=head_w[0].device)
    _lib.check(lib.pdsc_wfrag_build_head_fmt(*[_p(_chk(w, "head_w")) for w in head_w], _p(out), LAYER_GEMMS[gemm], _stream()),
               "pdsc_wfrag_build_head_fmt")
    return out


@_on_device
def layer_fused_split(msg, res, feat_in, tail_w, head_w, bs: int, n: int, want_qkv: bool = False, partials=None,
                      qkv_split: bool = False, frag: bool = False, gemm: str = "f32", want_feat: bool = True):
    """pdsc_layer_fused_split: like layer_fused, rows = bs pairs of n points, head emits the split streams.
    partials = (scratch, nsplit) from sc_attention_split(..., merge=False) replaces msg.
    qkv_split: run the q|k|v projection in split precision (fp16 hi/lo weights).
    frag: go through pdsc_layer_fused_frag (weights as fragment-ordered streams; implies qkv_split) -- the entry the
    forward uses; gemm = "h3" (frag only): fc1..fc3 / PointCN in the fp16 hi / scaled-lo arithmetic.
    want_feat = False: no feat output (what the forward asks of every layer but the last); head_w = None (frag only):
    tail only (the forward's last layer).
    Returns (feat or None, featB or None, qkv or None, q_split or None, kv_tiles or None)."""
    lib = _lib.load()
    src = res if res is not None else feat_in
    m, dev = src.shape[0], src.device
    assert m == bs * n
    tail = [_chk(w, "tail_w") for w in tail_w] if tail_w is not None else [None] * 6
    has_head = head_w is not None
    assert has_head or frag, "tail-only launches exist on fragment streams only"
    head = [_chk(w, "head_w") for w in head_w] if has_head else [None] * 4
    feat = torch.empty(m, 128, device=dev, dtype=torch.float32) if tail_w is not None and (want_feat or not has_head) else None
    featB = torch.empty(m, 128, device=dev, dtype=torch.float32) if has_head else None
    qkv = torch.empty(m, 384, device=dev, dtype=torch.float32) if want_qkv and has_head else None
    qs = torch.empty(int(lib.pdsc_split_q_bytes(bs, n)), device=dev, dtype=torch.uint8) if has_head else None
    kv = torch.zeros(int(lib.pdsc_split_kv_bytes(bs, n)), device=dev, dtype=torch.uint8) if has_head else None
    part_o = part_ml = None
    nsplit = npad = 0
    if partials is not None:
        scratch, nsplit = partials
        npad = (n + 255) // 256 * 256
        part_o = C.c_void_p(scratch.data_ptr())
        part_ml = C.c_void_p(scratch.data_ptr() + bs * nsplit * npad * 128 * 4)
    args = [_p(_chk(msg, "msg")) if msg is not None else None, part_o, part_ml, nsplit, npad,
            _p(_chk(res, "res")) if res is not None else None,
            _p(_chk(feat_in, "feat_in")) if feat_in is not None else None, _p(feat), _p(featB), _p(qkv), _p(qs), _p(kv)]
    if frag:
        wf_tail = frag_weights_tail(tail, gemm) if tail_w is not None else None
        wf_head = frag_weights_head(head, gemm) if has_head else None
        args += [_p(wf_tail), _p(wf_head)]
        _lib.check(lib.pdsc_layer_fused_frag_fmt(*args, LAYER_GEMMS[gemm], bs, n, _stream()), "pdsc_layer_fused_frag_fmt")
        return feat, featB, qkv, qs, kv
    assert gemm == "f32", "the H3 arithmetic exists on fragment streams only"
    args += [_p(w) for w in tail] + [_p(w) for w in head]
    wqs = split_weight(head[2]) if qkv_split else None
    _lib.check(lib.pdsc_layer_fused_split(*args, _p(wqs), bs, n, _stream()), "pdsc_layer_fused_split")
    return feat, featB, qkv, qs, kv


PF_PARTIALS, PF_RES, PF_FEATB = 1, 2, 4      # enum pdsc_layer_io


def pf_rows(n: int) -> int:
    """rows per pair of a point-fragment buffer: whole tiles of 32."""
    return (n + 31) // 32 * 32


def rows_to_pf(x: torch.Tensor, bs: int, n: int) -> torch.Tensor:
    """[bs*n,128] rows -> point-fragment order [bs * pf_rows(n) * 128] (padding rows = copies of the pair's last row)."""
    t = pf_rows(n) // 32
    x = x.reshape(bs, n, 128)
    x = torch.cat([x, x[:, -1:].expand(bs, t * 32 - n, 128)], dim=1)            # [bs, t*32, 128]
    x = x.reshape(bs, t, 32, 16, 2, 4)                                          # [b, tile, l31, q, h, e]
    return x.permute(0, 1, 3, 4, 2, 5).contiguous().reshape(-1)                 # [b, tile, q, h, l31, e]: lane = 32 h + l31


def pf_to_rows(x: torch.Tensor, bs: int, rows_per_pair: int) -> torch.Tensor:
    """point-fragment buffer with rows_per_pair (multiple of 32) rows per pair -> [bs, rows_per_pair, 128]."""
    t = rows_per_pair // 32
    x = x.reshape(bs, t, 16, 2, 32, 4)
    return x.permute(0, 1, 4, 2, 3, 5).contiguous().reshape(bs, rows_per_pair, 128)


@_on_device
def layer_fused_io(res, feat_in, tail_w, head_w, bs: int, n: int, io_flags: int, partials=None):
    """pdsc_layer_fused_frag_io (H3 GEMMs, the forward's output set, hand-offs in point-fragment order per io_flags):
    partials = (scratch, nsplit) of sc_attention_split(merge=False, layout "pf" if io_flags & PF_PARTIALS);
    res in PF order if io_flags & PF_RES; featB comes back in PF order if io_flags & PF_FEATB.
    tail_w None = head only (feat_in rows), head_w None = tail only.  Returns (feat or None, featB or None, q_split, kv_tiles)."""
    lib = _lib.load()
    dev = (res if res is not None else feat_in).device
    m = bs * n
    has_tail, has_head = tail_w is not None, head_w is not None
    feat = torch.empty(m, 128, device=dev, dtype=torch.float32) if has_tail and not has_head else None
    fb_rows = bs * pf_rows(n) if io_flags & PF_FEATB else m
    featB = torch.empty(fb_rows * 128, device=dev, dtype=torch.float32) if has_head else None
    qs = torch.empty(int(lib.pdsc_split_q_bytes(bs, n)), device=dev, dtype=torch.uint8) if has_head else None
    kv = torch.zeros(int(lib.pdsc_split_kv_bytes(bs, n)), device=dev, dtype=torch.uint8) if has_head else None
    part_o = part_ml = None
    nsplit = npad = 0
    if partials is not None:
        scratch, nsplit = partials
        npad = (n + 255) // 256 * 256
        part_o = C.c_void_p(scratch.data_ptr())
        part_ml = C.c_void_p(scratch.data_ptr() + bs * nsplit * npad * 128 * 4)
    wf_tail = frag_weights_tail([_chk(w, "tail_w") for w in tail_w], "h3") if has_tail else None
    wf_head = frag_weights_head([_chk(w, "head_w") for w in head_w], "h3") if has_head else None
    _lib.check(lib.pdsc_layer_fused_frag_io(None, part_o, part_ml, nsplit, npad, _p(res) if res is not None else None,
                                            _p(_chk(feat_in, "feat_in")) if feat_in is not None else None, _p(feat), _p(featB),
                                            _p(qs), _p(kv), _p(wf_tail), _p(wf_head), LAYER_GEMMS["h3"], io_flags, bs, n, _stream()),
               "pdsc_layer_fused_frag_io")
    return feat, featB, qs, kv


def split_weight(w: torch.Tensor) -> torch.Tensor:
    """fp32 matrix [out,in] -> uint8 tensor holding fp16 hi [out,in] then fp16 lo [out,in] (layout of pdsc_wsplit_build)."""
    w = _chk(w, "weight")
    hi = w.to(torch.float16)
    lo = (w - hi.float()).to(torch.float16)
    return torch.cat([hi.reshape(-1), lo.reshape(-1)]).view(torch.uint8)


@_on_device
def layer_fused_x3(msg, res, feat_in, tail_w, head_w, bs: int, n: int, partials=None, want_qkv: bool = False,
                   want_feat: bool = False):
    """pdsc_layer_fused_x3 (split-precision chain).  tail_w/head_w as in layer_fused (fp32 matrices; split here).
    partials = (scratch uint8 tensor, nsplit) left by sc_attention_split(..., merge=False) replaces msg.
    Returns (feat or None, featB or None, qkv or None, q_split or None, kv_tiles or None)."""
    lib = _lib.load()
    src = res if res is not None else feat_in
    m, dev = src.shape[0], src.device
    assert m == bs * n
    tail = list(tail_w) if tail_w is not None else [None] * 6
    head = list(head_w) if head_w is not None else [None] * 4
    keep = []

    def mat(w):
        if w is None:
            return None
        keep.append(split_weight(w))
        return _p(keep[-1])

    def vec(b):
        return None if b is None else _p(_chk(b, "bias"))

    has_tail, has_head = tail_w is not None, head_w is not None
    feat = torch.empty(m, 128, device=dev, dtype=torch.float32) if (has_tail and (want_feat or not has_head)) else None
    featB = torch.empty(m, 128, device=dev, dtype=torch.float32) if has_head else None
    qkv = torch.empty(m, 384, device=dev, dtype=torch.float32) if (has_head and want_qkv) else None
    qs = torch.empty(int(lib.pdsc_split_q_bytes(bs, n)), device=dev, dtype=torch.uint8) if has_head else None
    kv = torch.zeros(int(lib.pdsc_split_kv_bytes(bs, n)), device=dev, dtype=torch.uint8) if has_head else None
    part_o = part_ml = None
    nsplit = npad = 0
    if partials is not None:
        scratch, nsplit = partials
        npad = (n + 255) // 256 * 256
        part_o = C.c_void_p(scratch.data_ptr())
        part_ml = C.c_void_p(scratch.data_ptr() + bs * nsplit * npad * 128 * 4)
    args = [_p(_chk(msg, "msg")) if msg is not None else None, part_o, part_ml, nsplit, npad,
            _p(_chk(res, "res")) if res is not None else None, _p(_chk(feat_in, "feat_in")) if feat_in is not None else None,
            _p(feat), _p(featB), _p(qkv), _p(qs), _p(kv),
            mat(tail[0]), vec(tail[1]), mat(tail[2]), vec(tail[3]), mat(tail[4]), vec(tail[5]),
            mat(head[0]), vec(head[1]), mat(head[2]), vec(head[3])]
    _lib.check(lib.pdsc_layer_fused_x3(*args, bs, n, _stream()), "pdsc_layer_fused_x3")
    return feat, featB, qkv, qs, kv


STATS_COLUMNS = ("success", "RE_deg", "TE_cm", "num_gt_inliers", "gt_inlier_ratio", "num_true_positives", "precision", "recall", "f1")


@_on_device
def eval_stats(trans, gt_trans, pred_labels, gt_labels, re_thre: float = 15.0, te_thre: float = 30.0) -> torch.Tensor:
    """[bs,9] evaluation rows on the device (columns: STATS_COLUMNS) -- libs/loss.py:44-51,96-100 and the stats row of
    evaluation/test_3DMatch.py:90-98 without the per-pair device->host copies."""
    lib = _lib.load()
    T, G = _chk(trans, "trans"), _chk(gt_trans, "gt_trans")
    p, g = _chk(pred_labels, "pred_labels"), _chk(gt_labels.to(torch.float32), "gt_labels")
    bs, n = p.shape
    stats = torch.empty(bs, 9, device=p.device, dtype=torch.float32)
    _lib.check(lib.pdsc_eval_stats(_p(T), _p(G), _p(p), _p(g), float(re_thre), float(te_thre), _p(stats), bs, n, _stream()),
               "pdsc_eval_stats")
    return stats


@_on_device
def feature_compat(normed: torch.Tensor, sigma: torch.Tensor, bs: int, n: int) -> torch.Tensor:
    """normed [bs*N,128] -> M [bs,N,N] = clamp(1 - (1 - F F^T)/sigma^2, 0, 1), zero diagonal (models/PointDSC.py:158-163)."""
    lib = _lib.load()
    normed, sig = _chk(normed, "normed"), _chk(sigma.reshape(-1), "sigma")
    m = torch.empty(bs, n, n, device=normed.device, dtype=torch.float32)
    _lib.check(lib.pdsc_feature_compat(_p(normed), _p(sig), _p(m), n, bs, n, _stream()), "pdsc_feature_compat")
    return m


@_on_device
def normalize_confidence(feat, h2, w3, b3) -> Tuple[torch.Tensor, torch.Tensor]:
    lib = _lib.load()
    feat, h2, w3, b3 = _chk(feat, "feat"), _chk(h2, "h2"), _chk(w3.reshape(-1), "w3"), _chk(b3.reshape(-1), "b3")
    m = feat.shape[0]
    normed = torch.empty_like(feat)
    conf = torch.empty(m, device=feat.device, dtype=torch.float32)
    _lib.check(lib.pdsc_normalize_confidence(_p(feat), _p(h2), _p(w3), _p(b3), _p(normed), _p(conf), m, _stream()),
               "pdsc_normalize_confidence")
    return normed, conf


@_on_device
def normalize_confidence_pf(feat, h2, w3, b3):
    """feat [bs,N,128], h2 [bs,N,32] -> (normed [bs,N,128], normed_pf [bs,ceil(N/32)*32,128] point-fragment order, conf [bs,N])."""
    lib = _lib.load()
    feat, h2, w3, b3 = _chk(feat, "feat"), _chk(h2, "h2"), _chk(w3.reshape(-1), "w3"), _chk(b3.reshape(-1), "b3")
    bs, n = feat.shape[0], feat.shape[1]
    normed = torch.empty_like(feat)
    normed_pf = torch.empty(bs, (n + 31) // 32 * 32, 128, device=feat.device, dtype=torch.float32)
    conf = torch.empty(bs, n, device=feat.device, dtype=torch.float32)
    _lib.check(lib.pdsc_normalize_confidence_pf(_p(feat), _p(h2), _p(w3), _p(b3), _p(normed), _p(normed_pf), _p(conf), bs, n, _stream()),
               "pdsc_normalize_confidence_pf")
    return normed, normed_pf, conf


@_on_device
def nms_keys(src_keypts, conf, radius: float) -> torch.Tensor:
    lib = _lib.load()
    src, conf = _chk(src_keypts, "src_keypts"), _chk(conf, "conf")
    bs, n = src.shape[0], src.shape[1]
    keys = torch.empty(bs, n, device=src.device, dtype=torch.float32)
    _lib.check(lib.pdsc_nms_keys(_p(src), _p(conf), float(radius), _p(keys), bs, n, _stream()), "pdsc_nms_keys")
    return keys


@_on_device
def nms_keys_grid(src_keypts, conf, radius: float) -> torch.Tensor:
    """pdsc_nms_keys_grid: the same keys from the 3 x 3 neighbouring cells of a 2-D grid (what the forward calls)."""
    lib = _lib.load()
    src, conf = _chk(src_keypts, "src_keypts"), _chk(conf, "conf")
    bs, n = src.shape[0], src.shape[1]
    keys = torch.empty(bs, n, device=src.device, dtype=torch.float32)
    nb = int(lib.pdsc_nms_workspace_bytes(bs, n))
    ws = torch.empty(nb, device=src.device, dtype=torch.uint8)
    _lib.check(lib.pdsc_nms_keys_grid(_p(src), _p(conf), float(radius), _p(keys), _p(ws), nb, bs, n, _stream()), "pdsc_nms_keys_grid")
    return keys


@_on_device
def rank_select(keys, num_seeds: int) -> torch.Tensor:
    lib = _lib.load()
    keys = _chk(keys, "keys")
    bs, n = keys.shape
    seeds = torch.empty(bs, num_seeds, device=keys.device, dtype=torch.int32)
    _lib.check(lib.pdsc_rank_select(_p(keys), _p(seeds), bs, n, num_seeds, _stream()), "pdsc_rank_select")
    return seeds


@_on_device
def pick_seeds(src_keypts, scores, R: float, max_num: int) -> torch.Tensor:
    """reference PointDSC.pick_seeds (dists replaced by the keypoints they were computed from)."""
    return rank_select(nms_keys(src_keypts, scores, R), max_num).long()


@_on_device
def knn_seeds(normed, seeds, k: int, return_dist: bool = False, form: str = "auto", normed_pf=None):
    """normed [bs,N,128], seeds [bs,S] int32 -> knn_idx [bs,S,k] int32.  form: "auto" (the library's choice), "matrix" (Gram rows
    written to HBM, then a selection launch) or "fused" (one launch, no S x N matrix; the returned distances are then undefined).
    normed_pf: the rows in point-fragment order (ops.normalize_confidence_pf), the fused form's fast column operand."""
    forms = {"auto": 0, "matrix": 1, "fused": 2}
    if return_dist:
        #  only the matrix form writes the S x N distances: "auto" may pick the fused form for large batches
        if form == "fused":
            raise ValueError('knn_seeds(return_dist=True) needs form="matrix" (the fused form never writes the S x N distance matrix)')
        form = "matrix"
    lib = _lib.load()
    normed, seeds = _chk(normed, "normed"), _chk(seeds, "seeds", torch.int32)
    bs, n = normed.shape[0], normed.shape[1]
    s = seeds.shape[1]
    ld = compat_ld(n)
    dist = torch.empty(bs, s, ld, device=normed.device, dtype=torch.float32)
    idx = torch.empty(bs, s, k, device=normed.device, dtype=torch.int32)
    _lib.check(lib.pdsc_knn_seeds_form(_p(normed), _p(normed_pf) if normed_pf is not None else None, _p(seeds), _p(dist), _p(idx), bs, n, s, k,
                                       forms[form], _stream()), "pdsc_knn_seeds")
    return (idx, dist[..., :n]) if return_dist else idx


@_on_device
def seed_power_iteration(normed, src, tgt, knn_idx, sigma, sigma_spat, num_iterations: int, want_M: bool = False):
    lib = _lib.load()
    normed, src, tgt = _chk(normed, "normed"), _chk(src, "src"), _chk(tgt, "tgt")
    knn_idx = _chk(knn_idx, "knn_idx", torch.int32)
    sigma, sigma_spat = _chk(sigma.reshape(-1), "sigma"), _chk(sigma_spat.reshape(-1), "sigma_spat")
    bs, n = src.shape[0], src.shape[1]
    s, k = knn_idx.shape[1], knn_idx.shape[2]
    iters = torch.zeros(bs, s, max(num_iterations, 1), 64, device=src.device, dtype=torch.float32)
    mask = torch.empty(bs, device=src.device, dtype=torch.int32)
    M = torch.empty(bs, s, k, k, device=src.device, dtype=torch.float32) if want_M else None
    _lib.check(lib.pdsc_seed_power_iteration(_p(normed), _p(src), _p(tgt), _p(knn_idx), _p(sigma), _p(sigma_spat),
                                             _p(iters), _p(mask), _p(M), bs, n, s, k, num_iterations, _stream()),
               "pdsc_seed_power_iteration")
    return iters, mask, M


@_on_device
def seed_transforms(src, tgt, knn_idx, iters, mask, num_iterations: int):
    lib = _lib.load()
    src, tgt = _chk(src, "src"), _chk(tgt, "tgt")
    knn_idx = _chk(knn_idx, "knn_idx", torch.int32)
    bs, n = src.shape[0], src.shape[1]
    s, k = knn_idx.shape[1], knn_idx.shape[2]
    trans = torch.empty(bs, s, 4, 4, device=src.device, dtype=torch.float32)
    w = torch.empty(bs, s, k, device=src.device, dtype=torch.float32)
    _lib.check(lib.pdsc_seed_transforms(_p(src), _p(tgt), _p(knn_idx), _p(iters), _p(mask), _p(trans), _p(w),
                                        bs, n, s, k, num_iterations, _stream()), "pdsc_seed_transforms")
    return trans, w


@_on_device
def rigid_transform_3d(A: torch.Tensor, B: torch.Tensor, weights: Optional[torch.Tensor] = None,
                       weight_threshold: float = 0) -> torch.Tensor:
    """Drop-in for reference models/common.py:rigid_transform_3d: A,B [bs,n,3] -> [bs,4,4]."""
    lib = _lib.load()
    A, B = _chk(A, "A"), _chk(B, "B")
    bs, n = A.shape[0], A.shape[1]
    w = _chk(weights, "weights") if weights is not None else None
    T = torch.empty(bs, 4, 4, device=A.device, dtype=torch.float32)
    _lib.check(lib.pdsc_rigid_transform_3d(_p(A), _p(B), _p(w), float(weight_threshold), _p(T), bs, n, _stream()),
               "pdsc_rigid_transform_3d")
    return T


@_on_device
def score_hypotheses(seed_trans, src, tgt, inlier_threshold: float):
    lib = _lib.load()
    seed_trans, src, tgt = _chk(seed_trans, "seed_trans"), _chk(src, "src"), _chk(tgt, "tgt")
    bs, n = src.shape[0], src.shape[1]
    s = seed_trans.shape[1]
    counts = torch.empty(bs, s, device=src.device, dtype=torch.int32)
    _lib.check(lib.pdsc_score_hypotheses(_p(seed_trans), _p(src), _p(tgt), float(inlier_threshold), _p(counts),
                                         bs, n, s, _stream()), "pdsc_score_hypotheses")
    best = torch.empty(bs, device=src.device, dtype=torch.int32)
    initial = torch.empty(bs, 4, 4, device=src.device, dtype=torch.float32)
    labels = torch.empty(bs, n, device=src.device, dtype=torch.float32)
    _lib.check(lib.pdsc_select_best(_p(counts), _p(seed_trans), _p(src), _p(tgt), float(inlier_threshold), _p(best),
                                    _p(initial), _p(labels), bs, n, s, _stream()), "pdsc_select_best")
    return counts, best, initial, labels


@_on_device
def post_refinement(initial_trans, src, tgt, threshold: float, max_iters: int = 20):
    lib = _lib.load()
    initial_trans, src, tgt = _chk(initial_trans, "initial_trans"), _chk(src, "src"), _chk(tgt, "tgt")
    bs, n = src.shape[0], src.shape[1]
    final = torch.empty(bs, 4, 4, device=src.device, dtype=torch.float32)
    solves = torch.empty(bs, device=src.device, dtype=torch.int32)
    _lib.check(lib.pdsc_post_refinement(_p(initial_trans), _p(src), _p(tgt), float(threshold), max_iters, _p(final),
                                        _p(solves), bs, n, _stream()), "pdsc_post_refinement")
    return final, solves
