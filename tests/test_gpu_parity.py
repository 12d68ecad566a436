"""GPU parity tests (-m gpu): every stage of the HIP path, called through the C-ABI, against the CPU oracle
on identical seeded inputs, then the whole path against the committed golden fixtures (outputs of the
unmodified reference).  Each stage test feeds ORACLE inputs to ONE stage so failures do not cascade.

Bars: bit-exact for integer/index/mask work (compat bits, NMS keys, seeds, inlier counts, best index, labels);
floating-point stages within the tolerance written next to each assert; R/t within 1e-4 (BASELINE.json).
"""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import pointdsc_oracle as O
from pointdsc_amd import PointDSC, _lib, ops, synthetic, workloads

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]
GOLDEN = ROOT / "tests" / "golden"
DEV = "cuda:0"
KW = dict(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1, inlier_threshold=0.10,
          sigma_d=0.10, k=40, nms_radius=0.10)
ORACLE_KEYS = ("num_layers", "num_channels", "num_iterations", "ratio", "inlier_threshold", "k", "nms_radius")
QSCALE = float(np.log2(np.e) / np.sqrt(128.0))
# A/B record kernels and PDSC_* environment knobs exist in experiments builds only (POINTDSC_HIP_LIB=.../libpointdsc_hip_exp.so)
EXPERIMENTS = bool(_lib.load().pdsc_experiments_enabled())
needs_experiments = pytest.mark.skipif(not EXPERIMENTS, reason="experiments builds only (python -m pointdsc_amd.build --experiments)")
PRECISIONS = ["fp16x3", "fp32", pytest.param("fp16x3_all", marks=needs_experiments)]


def g(t):
    return t.to(DEV).contiguous()


_CASES = {}


def case(n, pair_seed=21, wseed=6, inlier_ratio=0.3, kw=None):
    """Oracle stages for one seeded pair (cached per session)."""
    key = (n, pair_seed, wseed, inlier_ratio, json.dumps(kw or {}, sort_keys=True))
    if key not in _CASES:
        mk = dict(KW, **(kw or {}))
        model = PointDSC(**mk)
        sd = synthetic.make_state_dict(model.state_dict(), seed=wseed)
        model.load_state_dict(sd)
        model = model.eval().to(DEV)
        pair = synthetic.make_pair(n, inlier_ratio=inlier_ratio, seed=pair_seed)
        res = O.forward_testing(sd, pair["corr_pos"], pair["src_keypts"], pair["tgt_keypts"], return_stages=True,
                                **{k: mk[k] for k in ORACLE_KEYS})
        _CASES[key] = dict(model=model, sd=sd, pair=pair, st=res["stages"][0], res=res, kw=mk)
    return _CASES[key]


# ------------------------------------------------------------------------------------------------------
# a-1 compat
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [257, 1000, 2053])
def test_spatial_compat_bit_exact(n):
    c = case(n)
    src, tgt = g(c["pair"]["src_keypts"]), g(c["pair"]["tgt_keypts"])
    compat, dist = ops.spatial_compat(src, tgt, g(c["sd"]["sigma_spat"]), want_dist=True)
    assert compat.shape[-1] == ops.compat_ld(n)
    assert torch.equal(compat[0, :, :n].cpu(), c["st"]["compat"])            # bit-exact vs oracle == reference
    assert torch.equal(dist[0, :, :n].cpu(), c["st"]["src_dist"])
    assert float(compat[0, :, n:].abs().sum()) == 0.0                          # padding columns are zero
    only = ops.spatial_compat(src, tgt, g(c["sd"]["sigma_spat"]))
    assert torch.equal(only, compat)


def test_exact_math_primitives_match_ieee():
    """sqrt_rn / InvariantDivisor of compat.hip vs correctly rounded results on 16M values incl. the ranges the
    kernel sees (squared distances 0 .. 1e4) and far outside them."""
    import ctypes as C
    from pointdsc_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(0)
    n = 1 << 22
    chunks = [rs.random_sample(n) * 30.0, rs.random_sample(n) * 1e4, np.exp(rs.uniform(-60, 40, n)),
              np.concatenate([[0.0, 1.0, 4.0, 2.0, 1e-30, 3.0e38], rs.random_sample(n - 6)])]
    x = torch.from_numpy(np.concatenate(chunks).astype(np.float32))
    for b in (0.1 ** 2, 1.2 ** 2, float(np.float32(0.1) * np.float32(0.1)), 3.0, 7.3e-3):
        b = float(np.float32(b))
        xd = g(x)
        sq, dv = torch.empty_like(xd), torch.empty_like(xd)
        _lib.check(lib.pdsc_selftest_exact_math(C.c_void_p(xd.data_ptr()), b, C.c_void_p(sq.data_ptr()), C.c_void_p(dv.data_ptr()),
                                                x.numel(), torch.cuda.current_stream().cuda_stream), "selftest")
        want_sq = np.sqrt(x.numpy().astype(np.float64)).astype(np.float32)
        assert np.array_equal(sq.cpu().numpy(), want_sq)
        want_dv = (x.numpy() / np.float32(b)).astype(np.float32)             # IEEE fp32 division
        normal = (want_dv > 1e-30) & (want_dv < 1e30)                        # quotients that survive `1 - q` at all
        got = dv.cpu().numpy()
        assert np.array_equal(got[normal], want_dv[normal])
        assert np.all(got[x.numpy() == 0] == 0)


@pytest.mark.parametrize("scale,sigma,n", [(3.0, 0.1, 5000), (60.0, 1.2, 3000)])
def test_spatial_compat_bit_exact_large(scale, sigma, n):
    pair = synthetic.make_pair(n, seed=77, scale=scale, noise=0.01 * scale / 3)
    sig = torch.tensor([sigma])
    compat = ops.spatial_compat(g(pair["src_keypts"]), g(pair["tgt_keypts"]), g(sig))
    _, want = O.spatial_compat(pair["src_keypts"][0], pair["tgt_keypts"][0], sig)
    assert torch.equal(compat[0, :, :n].cpu(), want)
    assert torch.equal(compat[0, :, :n], compat[0, :, :n].transpose(0, 1))   # exactly symmetric


def test_spatial_compat_batched_and_kitti_scale():
    b = synthetic.make_batch(3, 300, seed=50, scale=60.0, noise=0.1)
    sig = torch.tensor([1.2])
    compat = ops.spatial_compat(g(b["src_keypts"]), g(b["tgt_keypts"]), g(sig))
    for i in range(3):
        _, want = O.spatial_compat(b["src_keypts"][i], b["tgt_keypts"][i], sig)
        assert torch.equal(compat[i, :, :300].cpu(), want)


@pytest.mark.parametrize("n,bs,scale,sigma", [(33, 1, 3.0, 0.1), (257, 2, 3.0, 0.1), (1000, 3, 3.0, 0.1), (5000, 2, 3.0, 0.1),
                                                (3000, 1, 60.0, 1.2), (10000, 1, 3.0, 0.1)])
def test_spatial_compat_u16_tracks_the_fp32_matrix(n, bs, scale, sigma):
    """pdsc_spatial_compat_u16 (the matrix the split-precision attention streams), r03 contract: u = round(c * 65535) with c
    evaluated on the hardware's 1-ulp sqrt (packed fp32 math), in the attention kernel's tile order.  Against the bit-exact
    fp32 matrix c32 of pdsc_spatial_compat: |u - round(c32 * 65535)| <= 2 units (bound: 2 (ulp(d_src) + ulp(d_tgt)) |d_src -
    d_tgt| / sigma^2 * 65535 + the rounding), equal to it on the overwhelming majority of the entries;
    diagonal exactly 65535; entries that are clamped to 0 with a margin are exactly 0; symmetric bit for bit; padding zero."""
    batch = synthetic.make_batch(bs, n, seed=40 + n, scale=scale, noise=scale / 300.0)
    src, tgt, sig = g(batch["src_keypts"]), g(batch["tgt_keypts"]), g(torch.tensor([sigma]))
    c32 = ops.spatial_compat(src, tgt, sig)[:, :, :n]
    c16 = ops.spatial_compat_u16(src, tgt, sig)
    assert c16.shape == (bs, n, ops.compat_ld(n))
    u = (c16.to(torch.int32) & 0xFFFF)
    dec = ops.decode_compat_u16(c16, n)
    want = torch.round(c32.double() * 65535.0)
    got = torch.round(dec.double() * 65535.0)
    diff = (got - want).abs()
    assert float(diff.max()) <= 2.0
    nz = want > 0
    if bool(nz.any()):
        assert float((diff[nz] > 0).float().mean()) < 0.05      # expected: a few % of the non-zero entries, by one unit
    assert float((diff > 1).float().mean()) < 1e-4
    assert float((dec - c32).abs().max()) <= 2.5 / 65535
    # unclamped value in fp64: entries below -4e-5 (two units and the fp32 round-off away from the clamp) are exactly zero
    sd = (batch["src_keypts"].double()[:, :, None] - batch["src_keypts"].double()[:, None]).norm(dim=-1)
    td = (batch["tgt_keypts"].double()[:, :, None] - batch["tgt_keypts"].double()[:, None]).norm(dim=-1)
    c64 = (1.0 - (sd - td) ** 2 / float(torch.tensor(sigma, dtype=torch.float32) ** 2)).to(DEV)
    assert int((dec[c64 < -4e-5] != 0).sum()) == 0
    assert bool((torch.diagonal(got, dim1=1, dim2=2) == 65535).all())
    assert bool((got[c32 == 1.0] == 65535).all())
    assert torch.equal(dec, dec.transpose(1, 2))
    # columns >= N of the padded rows (tile order keeps them inside their own 32-group): all zero
    j = torch.arange(ops.compat_ld(n), device=DEV)
    r = j & 31
    pos = (j & ~31) + 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3)
    assert int(u[:, :, pos[n:]].abs().sum()) == 0


# ------------------------------------------------------------------------------------------------------
# a-2 point-wise layers
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,k,nout", [(1, 8, 1), (63, 32, 32), (64, 64, 64), (65, 128, 128), (1000, 128, 384),
                                      (257, 128, 100), (5000, 64, 128), (130, 128, 32)])
@pytest.mark.parametrize("relu,res", [(False, False), (True, False), (False, True)])
def test_linear_matches_fp64(m, k, nout, relu, res):
    gen = torch.Generator().manual_seed(m * 7 + k + nout)
    x = torch.randn(m, k, generator=gen)
    w = torch.randn(nout, k, generator=gen) / k ** 0.5
    b = torch.randn(nout, generator=gen)
    r = torch.randn(m, nout, generator=gen) if res else None
    y = ops.linear(g(x), g(w), g(b), relu=relu, residual=g(r) if res else None).cpu()
    want = x.double() @ w.double().T + b.double()
    if relu:
        want = want.clamp(min=0)
    if res:
        want = want + r.double()
    assert (y.double() - want).abs().max() < 2e-6 * max(1.0, float(want.abs().max()))   # fp32 accumulate, K<=128


def test_layer0_matches_oracle():
    c = case(1000)
    sd = c["sd"]
    w0 = torch.zeros(128, 16)
    w0[:, :6] = sd["encoder.layer0.weight"][:, :, 0]
    y = ops.layer0(g(c["pair"]["corr_pos"]), g(w0), g(sd["encoder.layer0.bias"])).cpu()
    want = (sd["encoder.layer0.weight"][:, :, 0] @ c["pair"]["corr_pos"][0].T + sd["encoder.layer0.bias"][:, None]).T
    assert (y - want).abs().max() < 1e-6


@pytest.mark.parametrize("in_dim", [1, 8, 9, 12, 16])
def test_layer0_wide_inputs(in_dim):
    """encoder.layer0 for the reference's other input widths (datasets/ThreeDMatch.py:299-312 builds 6, 9 and 12 columns)."""
    gen = torch.Generator().manual_seed(in_dim)
    x = torch.randn(777, in_dim, generator=gen)
    w = torch.randn(128, in_dim, generator=gen) / in_dim ** 0.5
    b = torch.randn(128, generator=gen)
    w0 = torch.zeros(128, 16)
    w0[:, :in_dim] = w
    y = ops.layer0(g(x), g(w0), g(b)).cpu()
    want = (x.double() @ w.double().T + b.double())
    assert (y.double() - want).abs().max() < 2e-6


def test_forward_with_twelve_input_columns():
    """in_dim = 12 end to end (the reference's 'in_dim == 12' branch concatenates normals): same stages, wider first conv."""
    kw = dict(KW, in_dim=12, num_layers=3)
    model = PointDSC(**kw)
    sd = synthetic.make_state_dict(model.state_dict(), seed=3)
    model.load_state_dict(sd)
    model = model.eval().to(DEV)
    pair = synthetic.make_pair(600, inlier_ratio=0.4, seed=5)
    gen = torch.Generator().manual_seed(1)
    corr12 = torch.cat([pair["corr_pos"], torch.randn(1, 600, 6, generator=gen) * 0.1], dim=-1)
    res = model({"corr_pos": g(corr12), "src_keypts": g(pair["src_keypts"]), "tgt_keypts": g(pair["tgt_keypts"]), "testing": True})
    ref = O.forward_testing(sd, corr12, pair["src_keypts"], pair["tgt_keypts"],
                            **{k: kw[k] for k in ("num_layers", "num_channels", "num_iterations", "ratio", "inlier_threshold", "k", "nms_radius")})
    assert int((res["final_labels"].cpu() != ref["final_labels"]).sum()) == 0
    assert float((res["final_trans"].cpu() - ref["final_trans"]).abs().max()) < 1e-4


@pytest.mark.parametrize("m", [1, 31, 32, 33, 1000])
def test_layer_fused_matches_fp64_chain(m):
    """pdsc_layer_fused (tail+head, head only, tail only) vs the five GEMMs in fp64."""
    gen = torch.Generator().manual_seed(m)
    rnd = lambda *shape: torch.randn(*shape, generator=gen)  # noqa: E731
    msg, res = rnd(m, 128), rnd(m, 128)
    w1, b1, w2, b2, w3, b3 = rnd(64, 128) / 11, rnd(64), rnd(64, 64) / 8, rnd(64), rnd(128, 64) / 8, rnd(128)
    wp, bp, wq, bq = rnd(128, 128) / 11, rnd(128), rnd(384, 128) / 11, rnd(384)
    d = lambda t: t.double()  # noqa: E731
    feat = d(res) + (torch.relu(torch.relu(d(msg) @ d(w1).T + d(b1)) @ d(w2).T + d(b2)) @ d(w3).T + d(b3))
    featB = torch.relu(feat @ d(wp).T + d(bp))
    qkv = featB @ d(wq).T + d(bq)
    tail_w, head_w = [g(x) for x in (w1, b1, w2, b2, w3, b3)], [g(x) for x in (wp, bp, wq, bq)]
    tol = lambda want: 3e-6 * max(1.0, float(want.abs().max()))  # noqa: E731
    f, fb, q = ops.layer_fused(g(msg), g(res), None, tail_w, head_w, want_feat=True)
    assert (f.cpu().double() - feat).abs().max() < tol(feat)
    assert (fb.cpu().double() - featB).abs().max() < tol(featB)
    assert (q.cpu().double() - qkv).abs().max() < 2 * tol(qkv)
    f2, fb2, q2 = ops.layer_fused(g(msg), g(res), None, tail_w, None)
    assert fb2 is None and q2 is None and torch.equal(f2, f)
    _, fb3, q3 = ops.layer_fused(None, None, f, None, head_w)
    assert torch.equal(fb3, fb) and torch.equal(q3, q)


# ------------------------------------------------------------------------------------------------------
# a-3 attention
# ------------------------------------------------------------------------------------------------------
def _attention_ref(q, k, v, compat):
    s = (q.double() @ k.double().T) / np.sqrt(128.0)
    w = torch.softmax(compat.double() * s, dim=-1)
    return w @ v.double()


@pytest.mark.parametrize("n,bs", [(257, 1), (1000, 2), (2053, 1), (96, 3), (33, 1)])
@pytest.mark.parametrize("nsplit", [1, 0, 3])
def test_sc_attention_matches_fp64_softmax(n, bs, nsplit):
    gen = torch.Generator().manual_seed(n + bs)
    batch = synthetic.make_batch(bs, n, seed=70 + n)
    compat = ops.spatial_compat(g(batch["src_keypts"]), g(batch["tgt_keypts"]), g(torch.tensor([0.1])))
    q, k, v = (torch.randn(bs, n, 128, generator=gen) * s for s in (2.0, 2.0, 1.0))
    qkv = torch.cat([q * QSCALE, k, v], dim=-1).reshape(bs * n, 384)
    msg = ops.sc_attention(g(qkv), compat, bs, n, nsplit=nsplit).cpu().reshape(bs, n, 128)
    for b in range(bs):
        want = _attention_ref(q[b], k[b], v[b], compat[b, :, :n].cpu())
        err = (msg[b].double() - want).abs().max()
        assert err < 5e-6 * max(1.0, float(want.abs().max())), (b, float(err))   # fp32 MFMA + online softmax


def test_sc_attention_online_softmax_rescale_branch():
    """Force the running max to jump late (one key dominates one query in the LAST tile) and early."""
    n = 320
    gen = torch.Generator().manual_seed(5)
    q, k, v = torch.randn(n, 128, generator=gen), torch.randn(n, 128, generator=gen), torch.randn(n, 128, generator=gen)
    k[n - 3] = q[7] * 3.0            # spike in the last tile for query 7
    k[2] = q[200] * 3.0              # spike in the first tile for query 200
    compat = torch.ones(1, n, ops.compat_ld(n))
    qkv = torch.cat([q * QSCALE, k, v], dim=-1)
    for nsplit in (1, 2, 5):
        msg = ops.sc_attention(g(qkv), g(compat), 1, n, nsplit=nsplit).cpu()
        want = _attention_ref(q, k, v, compat[0, :, :n])
        assert (msg.double() - want).abs().max() < 1e-5



# ------------------------------------------------------------------------------------------------------
# a-3 split-precision attention (fp16 hi/lo operands, three MFMAs per operand pair) and its operand streams
# ------------------------------------------------------------------------------------------------------
def _split(x):
    """split_layout.h: hi = f16(x), lo = f16(x - hi), round to nearest even, lo unscaled (r05: fp16 parts; bf16 in r01-r04)."""
    hi = x.float().half()
    lo = (x.float() - hi.float()).half()
    return hi, lo


def _pack_reference(qkv, bs, n):
    """CPU restatement of pointdsc_amd/csrc/split_layout.h: (q_split bytes, kv_tiles bytes)."""
    tiles = (n + 31) // 32
    qkv = qkv.reshape(bs, n, 384)
    qh, ql = _split(qkv[..., :128])
    qs = torch.cat([qh, ql], dim=-1).reshape(bs * n, 256)
    pad = torch.zeros(bs, tiles * 32, 256)
    pad[:, :n] = qkv[..., 128:]
    k = pad[..., :128].reshape(bs, tiles, 32, 128)
    v = pad[..., 128:].reshape(bs, tiles, 32, 128)
    # K image, chunk-major: [16 chunks of 8 channels][32 keys][8]  (8 KiB per plane, no padding)
    kimg = k.reshape(bs, tiles, 32, 16, 8).permute(0, 1, 3, 2, 4).contiguous()
    # V^T image, chunk-major: [4 chunks of 8 keys][128 channels][8];
    # chunk jh = 2j+h, element e = V[16j + 8(e>>2) + 4h + (e&3)][channel]
    vimg = torch.zeros(bs, tiles, 4, 128, 8)
    for jh in range(4):
        for e in range(8):
            kk = 16 * (jh >> 1) + 8 * (e >> 2) + 4 * (jh & 1) + (e & 3)
            vimg[:, :, jh, :, e] = v[:, :, kk, :]
    kh, kl = _split(kimg.reshape(bs, tiles, -1))
    vh, vl = _split(vimg.reshape(bs, tiles, -1))
    kv = torch.cat([kh, kl, vh, vl], dim=-1)
    assert kv.shape[-1] * 2 == 32768
    # images sit 37 KiB apart (SPL_TILE_STRIDE); the 5 KiB in between are never touched (ops allocates the stream zeroed)
    kv = torch.cat([kv, torch.zeros(bs, tiles, (37 * 1024 - 32768) // 2, dtype=kv.dtype)], dim=-1)
    return qs.view(torch.uint8).reshape(-1), kv.contiguous().view(torch.uint8).reshape(-1)


@pytest.mark.parametrize("n,bs", [(33, 1), (96, 3), (257, 2), (1000, 1)])
def test_pack_qkv_split_streams_bit_exact(n, bs):
    gen = torch.Generator().manual_seed(n)
    qkv = torch.randn(bs * n, 384, generator=gen) * torch.logspace(-3, 1, 384)[None]
    qs, kv = ops.pack_qkv_split(g(qkv), bs, n)
    want_qs, want_kv = _pack_reference(qkv, bs, n)
    assert torch.equal(qs.cpu(), want_qs)
    assert torch.equal(kv.cpu(), want_kv)


@pytest.mark.parametrize("n,bs", [(33, 1), (257, 2), (1000, 1)])
def test_layer_fused_split_emits_the_streams_of_its_own_qkv(n, bs):
    """The fused layer kernel's head writes the split streams directly; they must be exactly the packing of the
    fp32 q|k|v it would have written, and every fp32 output must equal the M-row entry point's."""
    gen = torch.Generator().manual_seed(n)
    rnd = lambda *shape: torch.randn(*shape, generator=gen)  # noqa: E731
    m = bs * n
    msg, res = rnd(m, 128), rnd(m, 128)
    tail_w = [g(x) for x in (rnd(64, 128) / 11, rnd(64), rnd(64, 64) / 8, rnd(64), rnd(128, 64) / 8, rnd(128))]
    head_w = [g(x) for x in (rnd(128, 128) / 11, rnd(128), rnd(384, 128) / 11, rnd(384))]
    f, fb, qkv, qs, kv = ops.layer_fused_split(g(msg), g(res), None, tail_w, head_w, bs, n, want_qkv=True)
    f0, fb0, qkv0 = ops.layer_fused(g(msg), g(res), None, tail_w, head_w, want_feat=True)
    assert torch.equal(f, f0) and torch.equal(fb, fb0) and torch.equal(qkv, qkv0)
    want_qs, want_kv = _pack_reference(qkv.cpu(), bs, n)
    assert torch.equal(qs.cpu(), want_qs) and torch.equal(kv.cpu(), want_kv)
    _, _, none_qkv, qs2, kv2 = ops.layer_fused_split(g(msg), g(res), None, tail_w, head_w, bs, n)
    assert none_qkv is None and torch.equal(qs2, qs) and torch.equal(kv2, kv)


@needs_experiments
@pytest.mark.parametrize("n,bs", [(1, 1), (31, 1), (33, 2), (1000, 1)])
def test_layer_fused_x3_matches_fp64_chain(n, bs):
    """Split-precision fused chain (tail+head, head only, tail only) vs the five GEMMs in fp64; its streams are exactly
    the packing of its own fp32 q|k|v."""
    gen = torch.Generator().manual_seed(n)
    rnd = lambda *shape: torch.randn(*shape, generator=gen)  # noqa: E731
    m = bs * n
    msg, res = rnd(m, 128), rnd(m, 128)
    w1, b1, w2, b2, w3, b3 = rnd(64, 128) / 11, rnd(64), rnd(64, 64) / 8, rnd(64), rnd(128, 64) / 8, rnd(128)
    wp, bp, wq, bq = rnd(128, 128) / 11, rnd(128), rnd(384, 128) / 11, rnd(384)
    d = lambda t: t.double()  # noqa: E731
    feat = d(res) + (torch.relu(torch.relu(d(msg) @ d(w1).T + d(b1)) @ d(w2).T + d(b2)) @ d(w3).T + d(b3))
    featB = torch.relu(feat @ d(wp).T + d(bp))
    qkv = featB @ d(wq).T + d(bq)
    tail_w, head_w = [g(x) for x in (w1, b1, w2, b2, w3, b3)], [g(x) for x in (wp, bp, wq, bq)]
    # 2^-16 per product, sqrt(K) accumulation, five chained GEMMs: a few 1e-5 of the largest activation
    tol = lambda want: 4e-5 * max(1.0, float(want.abs().max()))  # noqa: E731
    f, fb, q, qs, kv = ops.layer_fused_x3(g(msg), g(res), None, tail_w, head_w, bs, n, want_qkv=True, want_feat=True)
    assert (f.cpu().double() - feat).abs().max() < tol(feat)
    assert (fb.cpu().double() - featB).abs().max() < tol(featB)
    assert (q.cpu().double() - qkv).abs().max() < 2 * tol(qkv)
    want_qs, want_kv = _pack_reference(q.cpu(), bs, n)
    assert torch.equal(qs.cpu(), want_qs) and torch.equal(kv.cpu(), want_kv)
    f2, fb2, _, _, _ = ops.layer_fused_x3(g(msg), g(res), None, tail_w, None, bs, n)
    assert fb2 is None and torch.equal(f2, f)
    _, fb3, q3, qs3, kv3 = ops.layer_fused_x3(None, None, f, None, head_w, bs, n, want_qkv=True)
    assert torch.equal(fb3, fb) and torch.equal(q3, q) and torch.equal(qs3, qs) and torch.equal(kv3, kv)
    # and against the exact fp32 kernel
    f0, fb0, q0 = ops.layer_fused(g(msg), g(res), None, tail_w, head_w, want_feat=True)
    assert (f - f0).abs().max() < tol(feat) and (fb - fb0).abs().max() < tol(featB) and (q - q0).abs().max() < 2 * tol(qkv)


@pytest.mark.parametrize("n,bs", [(1, 1), (33, 2), (1000, 1)])
def test_layer_fused_split_precision_qkv_projection(n, bs):
    """Default mode: only the q|k|v projection runs in split precision; feat / featB are bit-identical to the exact
    kernel, q|k|v within the split tolerance, streams = packing of the kernel's own fp32 q|k|v."""
    gen = torch.Generator().manual_seed(100 + n)
    rnd = lambda *shape: torch.randn(*shape, generator=gen)  # noqa: E731
    m = bs * n
    msg, res = rnd(m, 128), rnd(m, 128)
    tail_w = [g(x) for x in (rnd(64, 128) / 11, rnd(64), rnd(64, 64) / 8, rnd(64), rnd(128, 64) / 8, rnd(128))]
    head_w = [g(x) for x in (rnd(128, 128) / 11, rnd(128), rnd(384, 128) / 11, rnd(384))]
    f0, fb0, q0, _, _ = ops.layer_fused_split(g(msg), g(res), None, tail_w, head_w, bs, n, want_qkv=True)
    f1, fb1, q1, qs1, kv1 = ops.layer_fused_split(g(msg), g(res), None, tail_w, head_w, bs, n, want_qkv=True, qkv_split=True)
    assert torch.equal(f0, f1) and torch.equal(fb0, fb1)
    want = fb0.cpu().double() @ head_w[2].cpu().double().T + head_w[3].cpu().double()
    assert (q1.cpu().double() - want).abs().max() < 2e-5 * max(1.0, float(want.abs().max()))
    want_qs, want_kv = _pack_reference(q1.cpu(), bs, n)
    assert torch.equal(qs1.cpu(), want_qs) and torch.equal(kv1.cpu(), want_kv)
    _, fb2, q2, qs2, kv2 = ops.layer_fused_split(None, None, f0, None, head_w, bs, n, want_qkv=True, qkv_split=True)
    assert torch.equal(fb2, fb1) and torch.equal(q2, q1) and torch.equal(qs2, qs1) and torch.equal(kv2, kv1)


@pytest.mark.parametrize("n,bs", [(1, 1), (33, 2), (1000, 3), (3000, 3)])
def test_layer_fused_frag_streams_match_natural_weights(n, bs):
    """pdsc_layer_fused_frag (the forward's entry for large problems: wavefront-resident kernel, weights as
    MFMA-fragment-ordered streams, bias as one more k-step) against pdsc_layer_fused_split on natural-layout weights
    (which takes the workgroup-per-tile kernel for small problems: other summation order, so fp32 round-off apart)
    and against the fp64 chain; its streams are exactly the packing of its own q|k|v; head-only reproduces it."""
    gen = torch.Generator().manual_seed(200 + n)
    rnd = lambda *shape: torch.randn(*shape, generator=gen)  # noqa: E731
    m = bs * n
    msg, res = rnd(m, 128), rnd(m, 128)
    tail_w = [g(x) for x in (rnd(64, 128) / 11, rnd(64), rnd(64, 64) / 8, rnd(64), rnd(128, 64) / 8, rnd(128))]
    head_w = [g(x) for x in (rnd(128, 128) / 11, rnd(128), rnd(384, 128) / 11, rnd(384))]
    f0, fb0, q0, qs0, kv0 = ops.layer_fused_split(g(msg), g(res), None, tail_w, head_w, bs, n, want_qkv=True, qkv_split=True)
    f1, fb1, q1, qs1, kv1 = ops.layer_fused_split(g(msg), g(res), None, tail_w, head_w, bs, n, want_qkv=True, frag=True)
    for got, ref in ((f1, f0), (fb1, fb0), (q1, q0)):
        assert (got - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max()))
    want_qs, want_kv = _pack_reference(q1.cpu(), bs, n)
    assert torch.equal(qs1.cpu(), want_qs) and torch.equal(kv1.cpu(), want_kv)
    _, fb2, q2, qs2, kv2 = ops.layer_fused_split(None, None, f1, None, head_w, bs, n, want_qkv=True, frag=True)
    assert torch.equal(fb2, fb1) and torch.equal(q2, q1) and torch.equal(qs2, qs1) and torch.equal(kv2, kv1)
    d = lambda t: t.cpu().double()  # noqa: E731
    feat = d(res) + (torch.relu(torch.relu(d(msg) @ d(tail_w[0]).T + d(tail_w[1])) @ d(tail_w[2]).T + d(tail_w[3])) @ d(tail_w[4]).T
                     + d(tail_w[5]))
    assert (d(f1) - feat).abs().max() < 2e-5 * max(1.0, float(feat.abs().max()))
    featB = torch.relu(feat @ d(head_w[0]).T + d(head_w[1]))
    assert (d(fb1) - featB).abs().max() < 2e-5 * max(1.0, float(featB.abs().max()))
    qkv = featB @ d(head_w[2]).T + d(head_w[3])
    assert (d(q1) - qkv).abs().max() < 4e-5 * max(1.0, float(qkv.abs().max()))


@needs_experiments
@pytest.mark.parametrize("n,bs,nsplit", [(257, 1, 2), (1000, 2, 3), (300, 3, 4)])
def test_layer_fused_x3_merges_attention_partials(n, bs, nsplit):
    """Un-merged key-split partials fed to the layer kernel == merged msg fed to it (the merge arithmetic is the
    combine kernel's)."""
    gen = torch.Generator().manual_seed(n)
    rnd = lambda *shape: torch.randn(*shape, generator=gen)  # noqa: E731
    batch = synthetic.make_batch(bs, n, seed=5 + n)
    compat = ops.spatial_compat(g(batch["src_keypts"]), g(batch["tgt_keypts"]), g(torch.tensor([0.1])))
    qkv = torch.cat([rnd(bs * n, 128) * 0.3 * QSCALE, rnd(bs * n, 128) * 0.3, rnd(bs * n, 128)], dim=-1)
    qs, kv = ops.pack_qkv_split(g(qkv), bs, n)
    msg = ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=nsplit)
    partials = ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=nsplit, merge=False)
    res = rnd(bs * n, 128)
    tail_w = [g(x) for x in (rnd(64, 128) / 11, rnd(64), rnd(64, 64) / 8, rnd(64), rnd(128, 64) / 8, rnd(128))]
    head_w = [g(x) for x in (rnd(128, 128) / 11, rnd(128), rnd(384, 128) / 11, rnd(384))]
    a = ops.layer_fused_x3(msg, g(res), None, tail_w, head_w, bs, n, want_feat=True)
    b = ops.layer_fused_x3(None, g(res), None, tail_w, head_w, bs, n, partials=partials, want_feat=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    a = ops.layer_fused_split(msg, g(res), None, tail_w, head_w, bs, n)            # the exact-fp32 chain merges the same way
    b = ops.layer_fused_split(None, g(res), None, tail_w, head_w, bs, n, partials=partials)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])


@pytest.mark.parametrize("n,bs,nsplit", [(257, 1, 5), (1000, 1, 8), (300, 2, 6), (1000, 1, 7)])
def test_block_layer_kernel_merges_up_to_eight_partials(n, bs, nsplit):
    """Small problems (N = 1000 x 1 pair: the attention plan splits the keys 8 ways) merge the partials inside the
    workgroup-per-tile layer kernel as well: == the same kernel fed the combine kernel's merged msg."""
    gen = torch.Generator().manual_seed(n + nsplit)
    rnd = lambda *shape: torch.randn(*shape, generator=gen)  # noqa: E731
    batch = synthetic.make_batch(bs, n, seed=9 + n)
    compat = ops.spatial_compat(g(batch["src_keypts"]), g(batch["tgt_keypts"]), g(torch.tensor([0.1])))
    qkv = torch.cat([rnd(bs * n, 128) * 0.3 * QSCALE, rnd(bs * n, 128) * 0.3, rnd(bs * n, 128)], dim=-1)
    qs, kv = ops.pack_qkv_split(g(qkv), bs, n)
    msg = ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=nsplit)
    partials = ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=nsplit, merge=False)
    res = rnd(bs * n, 128)
    tail_w = [g(x) for x in (rnd(64, 128) / 11, rnd(64), rnd(64, 64) / 8, rnd(64), rnd(128, 64) / 8, rnd(128))]
    head_w = [g(x) for x in (rnd(128, 128) / 11, rnd(128), rnd(384, 128) / 11, rnd(384))]
    assert _lib.load().pdsc_layer_prefers_block(bs, n) == 1
    a = ops.layer_fused_split(msg, g(res), None, tail_w, head_w, bs, n, want_qkv=True, qkv_split=True)
    b = ops.layer_fused_split(None, g(res), None, tail_w, head_w, bs, n, want_qkv=True, qkv_split=True, partials=partials)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("n,bs,nsplit", [(257, 1, 2), (1000, 2, 3), (300, 3, 4), (5000, 2, 2)])
def test_layer_fused_frag_merges_attention_partials(n, bs, nsplit):
    """pdsc_layer_fused_frag = layer_wave_kernel (the kernel the bench times) fed the UN-MERGED key-split partials
    (part_o / part_ml, nsplit 2..4) == the same kernel fed the merged msg of the combine kernel, and == the fp64 merge."""
    gen = torch.Generator().manual_seed(n + 1)
    rnd = lambda *shape: torch.randn(*shape, generator=gen)  # noqa: E731
    batch = synthetic.make_batch(bs, n, seed=5 + n)
    compat = ops.spatial_compat(g(batch["src_keypts"]), g(batch["tgt_keypts"]), g(torch.tensor([0.1])))
    qkv = torch.cat([rnd(bs * n, 128) * 0.3 * QSCALE, rnd(bs * n, 128) * 0.3, rnd(bs * n, 128)], dim=-1)
    qs, kv = ops.pack_qkv_split(g(qkv), bs, n)
    msg = ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=nsplit)
    partials = ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=nsplit, merge=False)
    res = rnd(bs * n, 128)
    tail_w = [g(x) for x in (rnd(64, 128) / 11, rnd(64), rnd(64, 64) / 8, rnd(64), rnd(128, 64) / 8, rnd(128))]
    head_w = [g(x) for x in (rnd(128, 128) / 11, rnd(128), rnd(384, 128) / 11, rnd(384))]
    a = ops.layer_fused_split(msg, g(res), None, tail_w, head_w, bs, n, want_qkv=True, frag=True)
    b = ops.layer_fused_split(None, g(res), None, tail_w, head_w, bs, n, want_qkv=True, frag=True, partials=partials)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # fp64 merge of the raw partials: msg = sum_s o_s 2^(m_s - m) / sum_s l_s 2^(m_s - m)
    scratch, ns = partials
    npad = (n + 255) // 256 * 256
    flat = scratch.view(torch.float32).cpu().double()
    po = flat[: bs * ns * npad * 128].reshape(bs, ns, npad, 128)[:, :, :n]
    ml = flat[bs * ns * npad * 128: bs * ns * npad * 130].reshape(bs, ns, npad, 2)[:, :, :n]
    m = ml[..., 0].max(dim=1, keepdim=True).values
    wgt = torch.exp2(ml[..., 0] - m)
    want = ((po * wgt[..., None]).sum(1) / (ml[..., 1] * wgt).sum(1)[..., None]).reshape(bs * n, 128)
    assert (msg.cpu().double() - want).abs().max() < 1e-6 * max(1.0, float(want.abs().max()))
    d = lambda t: t.cpu().double()  # noqa: E731
    feat = d(res) + (torch.relu(torch.relu(want @ d(tail_w[0]).T + d(tail_w[1])) @ d(tail_w[2]).T + d(tail_w[3])) @ d(tail_w[4]).T
                     + d(tail_w[5]))
    assert (d(b[0]) - feat).abs().max() < 2e-5 * max(1.0, float(feat.abs().max()))


def _attention_split_model(q, k, v, compat):
    """fp64 evaluation of the split arithmetic: S = qh kh + qh kl + ql kh; O = (P V) with V = vh + vl."""
    (qh, ql), (kh, kl), (vh, vl) = _split(q), _split(k), _split(v)
    d = lambda t: t.double()  # noqa: E731
    s = d(qh) @ d(kh).T + d(qh) @ d(kl).T + d(ql) @ d(kh).T
    w = torch.softmax(compat.double() * s * np.log(2.0), dim=-1)
    return w @ (d(vh) + d(vl))


@pytest.mark.parametrize("n,bs", [(257, 1), (1000, 2), (2053, 1), (96, 3), (33, 1), (5000, 1), (1500, 9)])   # (1500, 9): the 8-wave kernel
@pytest.mark.parametrize("nsplit", [1, 0, 3])
@pytest.mark.parametrize("fmt", ["f32", "u16"])
def test_sc_attention_split_matches_fp64_softmax(n, bs, nsplit, fmt):
    gen = torch.Generator().manual_seed(n + bs)
    batch = synthetic.make_batch(bs, n, seed=70 + n)
    if fmt == "u16":        # the kernel streams the unorm16 matrix; the fp64 model uses exactly the values it decodes
        compat16 = ops.spatial_compat_u16(g(batch["src_keypts"]), g(batch["tgt_keypts"]), g(torch.tensor([0.1])))
        compat = ops.decode_compat_u16(compat16, n)
    else:
        compat16 = None
        compat = ops.spatial_compat(g(batch["src_keypts"]), g(batch["tgt_keypts"]), g(torch.tensor([0.1])))
    # network-like magnitudes (|q.k|/sqrt(C) of order 1) and a harsh case (logits of order 30)
    # (bounds: 5 x the largest measured value, profiles/r05_k_att_err.txt; with the bf16 pairs of rounds 1-4 they were 2e-5 / 5e-4)
    for qk_scale, tol_true in ((0.35, 8e-7), (2.0, 1.5e-5)):
        q, k, v = (torch.randn(bs, n, 128, generator=gen) * s for s in (qk_scale, qk_scale, 1.0))
        qkv = torch.cat([q * QSCALE, k, v], dim=-1).reshape(bs * n, 384)
        qs, kv = ops.pack_qkv_split(g(qkv), bs, n)
        msg = ops.sc_attention_split(qs, kv, compat16 if fmt == "u16" else compat, bs, n, nsplit=nsplit).cpu().reshape(bs, n, 128)
        for b in range(bs):
            cm = compat[b, :, :n].cpu()
            want = _attention_ref(q[b], k[b], v[b], cm)
            scale = max(1.0, float(want.abs().max()))
            err = float((msg[b].double() - want).abs().max())
            assert err < tol_true * scale, (qk_scale, b, err)     # split-precision error, grows with |logit|
            model = _attention_split_model(q[b] * QSCALE, k[b], v[b], cm)
            errm = float((msg[b].double() - model).abs().max())
            assert errm < tol_true * scale, (qk_scale, b, errm)   # against the fp64 evaluation of the same split operands


def test_sc_attention_split_online_softmax_rescale_branch():
    n = 320
    gen = torch.Generator().manual_seed(5)
    q, k, v = torch.randn(n, 128, generator=gen), torch.randn(n, 128, generator=gen), torch.randn(n, 128, generator=gen)
    k[n - 3] = q[7] * 3.0            # spike in the last tile for query 7
    k[2] = q[200] * 3.0              # spike in the first tile for query 200
    compat = torch.ones(1, n, ops.compat_ld(n))
    qkv = torch.cat([q * QSCALE, k, v], dim=-1)
    qs, kv = ops.pack_qkv_split(g(qkv), 1, n)
    for nsplit in (1, 2, 5):
        msg = ops.sc_attention_split(qs, kv, g(compat), 1, n, nsplit=nsplit).cpu()
        want = _attention_ref(q, k, v, compat[0, :, :n])
        assert (msg.double() - want).abs().max() < 2e-4


@pytest.mark.parametrize("nats", [17.0, 17.5])
def test_sc_attention_split_keeps_softmax_weights_below_the_fp16_denormal_floor(nats):
    """The softmax weights enter the P V product as fp16 hi + lo pairs, and fp16's floor is ABSOLUTE (2^-25 rounds to 0, 2^-24
    is the smallest step) where fp32's is not: with the row maximum mapped to p = 1, keys 17 nats below it would be rounded
    to 0 or to twice their weight, and 4095 such keys carry 1e-4 of the row sum.  The kernel keeps the maximum at
    p = 2^7 .. 2^15 instead (ATT_P_BIAS, attention_split.hip), so those keys keep their weight.  What remains in this
    construction is fp32 ACCUMULATION: 16-key partial sums of 5e-5 added to an accumulator holding the dominant key's 128
    (ulp 1.5e-5) -- any fp32 kernel loses that much, whence the 25 % bound on the low keys' share (measured 8.5 %)."""
    n = 4096
    a = float(np.sqrt(nats * np.sqrt(128.0)))
    q = torch.zeros(n, 128); q[:, 0] = a
    k = torch.zeros(n, 128); k[1:, 0] = -a            # key 0: logit 0 = the maximum; every other key `nats` below it
    v = torch.ones(n, 128); v[0] = -1.0
    compat = torch.ones(1, n, ops.compat_ld(n))
    qkv = torch.cat([q * QSCALE, k, v], dim=-1)
    qs, kv = ops.pack_qkv_split(g(qkv), 1, n)
    want = _attention_split_model(q * QSCALE, k, v, compat[0, :, :n])
    tail = float((want[0, 0] + 1.0) / 2.0)                              # the share of the row sum the low keys hold
    assert 0.5 * (n - 1) * np.exp(-nats) < tail < 1.1 * (n - 1) * np.exp(-nats)
    for nsplit in (1, 0, 3):
        msg = ops.sc_attention_split(qs, kv, g(compat), 1, n, nsplit=nsplit).cpu().double()
        err = float((msg - want).abs().max())
        assert err < 0.25 * 2.0 * tail, (nsplit, err, tail)            # (maximum at p = 1: 0.45 * 2 tail at 17.0 nats, 2 tail at 17.5)


@pytest.mark.parametrize("n,bs", [(1, 1), (33, 2), (1000, 3), (3000, 3)])
def test_layer_fused_frag_h3_gemms_match_the_fp32_gemms(n, bs):
    """PDSC_LAYER_GEMM_H3 (fc1..fc3 / PointCN as fp16 hi + scaled-lo operands, three f16 MFMAs per operand pair) against
    the same kernel on fp32 MFMAs and against the fp64 chain: ~2^-21 per product, i.e. fp32 round-off class; streams are
    the packing of its own q|k|v; head-only and the partial-merging input reproduce it bit for bit."""
    gen = torch.Generator().manual_seed(300 + n)
    rnd = lambda *shape: torch.randn(*shape, generator=gen)  # noqa: E731
    m = bs * n
    msg, res = rnd(m, 128), rnd(m, 128)
    tail_w = [g(x) for x in (rnd(64, 128) / 11, rnd(64), rnd(64, 64) / 8, rnd(64), rnd(128, 64) / 8, rnd(128))]
    head_w = [g(x) for x in (rnd(128, 128) / 11, rnd(128), rnd(384, 128) / 11, rnd(384))]
    f0, fb0, q0, _, _ = ops.layer_fused_split(g(msg), g(res), None, tail_w, head_w, bs, n, want_qkv=True, frag=True)
    f1, fb1, q1, qs1, kv1 = ops.layer_fused_split(g(msg), g(res), None, tail_w, head_w, bs, n, want_qkv=True, frag=True, gemm="h3")
    d = lambda t: t.cpu().double()  # noqa: E731
    feat = d(res) + (torch.relu(torch.relu(d(msg) @ d(tail_w[0]).T + d(tail_w[1])) @ d(tail_w[2]).T + d(tail_w[3])) @ d(tail_w[4]).T
                     + d(tail_w[5]))
    featB = torch.relu(feat @ d(head_w[0]).T + d(head_w[1]))
    qkv = featB @ d(head_w[2]).T + d(head_w[3])
    for name, got, ref, want, tol in (("feat", f1, f0, feat, 3e-6), ("featB", fb1, fb0, featB, 3e-6), ("qkv", q1, q0, qkv, 4e-5)):
        scale = max(1.0, float(want.abs().max()))
        e_h3, e_f32 = float((d(got) - want).abs().max()) / scale, float((d(ref) - want).abs().max()) / scale
        print(f"{name}: vs fp64 chain: h3 {e_h3:.2e}  fp32 MFMA {e_f32:.2e}")
        assert e_h3 < tol, (name, e_h3, e_f32)
    want_qs, want_kv = _pack_reference(q1.cpu(), bs, n)
    assert torch.equal(qs1.cpu(), want_qs) and torch.equal(kv1.cpu(), want_kv)
    _, fb2, q2, qs2, kv2 = ops.layer_fused_split(None, None, f1, None, head_w, bs, n, want_qkv=True, frag=True, gemm="h3")
    assert torch.equal(fb2, fb1) and torch.equal(q2, q1) and torch.equal(qs2, qs1) and torch.equal(kv2, kv1)
    only_tail = ops.layer_fused_split(g(msg), g(res), None, tail_w, head_w, bs, n, frag=True, gemm="h3")[0]
    assert torch.equal(only_tail, f1)


@pytest.mark.parametrize("n,bs,nsplit", [(33, 2, 0), (1000, 3, 2), (3001, 3, 0), (5000, 2, 2), (5000, 5, 4)])
def test_layer_h3_pipelined_kernel_is_bit_identical_to_the_generic_one(n, bs, nsplit):
    """layer_h3_kernel (csrc/layer_h3.hip: branch-free chunk loop, epilogues under the next tile's MFMAs -- what the forward
    runs with layer_gemm = "h3") takes exactly the output set the forward asks for; with any other set
    pdsc_layer_fused_frag_fmt falls back to layer_wave_kernel's H3 form.  Same MFMAs in the same order: every stream and row
    must agree bit for bit, for tail + head, head only and tail only, on merged msg and on un-merged partials, with ragged
    last tiles (the pipelined kernel stores unpredicated: rows beyond a pair's end repeat its last row)."""
    gen = torch.Generator().manual_seed(400 + n)
    rnd = lambda *shape: torch.randn(*shape, generator=gen)  # noqa: E731
    m = bs * n
    res = g(rnd(m, 128))
    tail_w = [g(x) for x in (rnd(64, 128) / 11, rnd(64), rnd(64, 64) / 8, rnd(64), rnd(128, 64) / 8, rnd(128))]
    head_w = [g(x) for x in (rnd(128, 128) / 11, rnd(128), rnd(384, 128) / 11, rnd(384))]
    if nsplit:
        batch = synthetic.make_batch(bs, n, seed=8 + n)
        compat = ops.spatial_compat(g(batch["src_keypts"]), g(batch["tgt_keypts"]), g(torch.tensor([0.1])))
        qkv = torch.cat([rnd(m, 128) * 0.3 * QSCALE, rnd(m, 128) * 0.3, rnd(m, 128)], dim=-1)
        qs, kv = ops.pack_qkv_split(g(qkv), bs, n)
        src = dict(msg=None, partials=ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=nsplit, merge=False))
    else:
        src = dict(msg=g(rnd(m, 128)), partials=None)
    kw = dict(frag=True, gemm="h3", partials=src["partials"])
    # sentinel rows around the outputs would catch a store past a pair's end: the streams are exact-size allocations, so
    # compare whole buffers instead (pads included: both kernels zero them)
    f_gen, fb_gen, _, qs_gen, kv_gen = ops.layer_fused_split(src["msg"], res, None, tail_w, head_w, bs, n, want_qkv=True, **kw)
    none_f, fb, none_q, qs1, kv1 = ops.layer_fused_split(src["msg"], res, None, tail_w, head_w, bs, n, want_feat=False, **kw)
    assert none_f is None and none_q is None
    assert torch.equal(fb, fb_gen) and torch.equal(qs1, qs_gen) and torch.equal(kv1, kv_gen)
    f_tail = ops.layer_fused_split(src["msg"], res, None, tail_w, None, bs, n, **kw)[0]                 # tail only
    assert torch.equal(f_tail, f_gen)
    _, fb2, _, qs2, kv2 = ops.layer_fused_split(None, None, f_gen, None, head_w, bs, n, frag=True, gemm="h3")   # head only
    assert torch.equal(fb2, fb_gen) and torch.equal(qs2, qs_gen) and torch.equal(kv2, kv_gen)


@pytest.mark.parametrize("n,bs,nsplit", [(5000, 1, 6), (1000, 2, 8), (2053, 1, 5), (3001, 1, 7)])
def test_h3_layer_kernel_merges_up_to_eight_key_splits(n, bs, nsplit):
    """r03: layer_h3_kernel merges 5..8 key-split partials itself (the plans of 1-3 pairs of N = 5000 / 10000 per GPU: no
    attention_combine launch, no msg round trip).  Same arithmetic as the combine kernel: feeding the point-fragment partials
    equals feeding the merged msg, bit for bit."""
    gen = torch.Generator().manual_seed(800 + n)
    rnd = lambda *shape: torch.randn(*shape, generator=gen)  # noqa: E731
    m = bs * n
    batch = synthetic.make_batch(bs, n, seed=19 + n)
    compat = ops.spatial_compat(g(batch["src_keypts"]), g(batch["tgt_keypts"]), g(torch.tensor([0.1])))
    qkv = torch.cat([rnd(m, 128) * 0.3 * QSCALE, rnd(m, 128) * 0.3, rnd(m, 128)], dim=-1)
    qs, kv = ops.pack_qkv_split(g(qkv), bs, n)
    msg = ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=nsplit)
    pf = ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=nsplit, merge=False, layout="pf")
    res = g(rnd(m, 128))
    res_pf = ops.rows_to_pf(res, bs, n)
    tail_w = [g(x) for x in (rnd(64, 128) / 11, rnd(64), rnd(64, 64) / 8, rnd(64), rnd(128, 64) / 8, rnd(128))]
    head_w = [g(x) for x in (rnd(128, 128) / 11, rnd(128), rnd(384, 128) / 11, rnd(384))]
    _, fb_r, _, qs_r, kv_r = ops.layer_fused_split(msg, res, None, tail_w, head_w, bs, n, frag=True, gemm="h3", want_feat=False)
    _, fb_p, qs_p, kv_p = ops.layer_fused_io(res_pf, None, tail_w, head_w, bs, n, ops.PF_PARTIALS | ops.PF_RES | ops.PF_FEATB, partials=pf)
    assert torch.equal(qs_p, qs_r) and torch.equal(kv_p, kv_r)
    assert torch.equal(ops.pf_to_rows(fb_p, bs, ops.pf_rows(n))[:, :n].reshape(m, 128), fb_r)


@pytest.mark.parametrize("n,nsplit", [(4100, 2), (5000, 3), (8190, 5), (16384, 8)])
def test_four_wavefront_layer_kernel_equals_the_wavefront_per_tile_kernel(n, nsplit):
    """r03: launches of at most 2560 tiles (N = 1000 x 1: 32 tiles, one dependency chain of 42 weight chunks per wavefront) take
    layer_h3_coop_kernel (csrc/layer_coop.hip): four wavefronts share a tile's output tiles and hand the stages' operands over
    in LDS.  Same MFMA order per output tile, so it must agree with layer_h3_kernel bit for bit: a batch of pairs with more than
    2560 tiles (one wavefront per tile) against the same pairs one at a time (at most 2560 tiles each), in every form the
    forward launches -- tail + head with point-fragment hand-offs, tail + head in rows, head only, tail only."""
    tiles = (n + 31) // 32
    bs = 2560 // tiles + 1
    assert bs * tiles > 2560 >= tiles
    gen = torch.Generator().manual_seed(4400 + n)
    rnd = lambda *shape: torch.randn(*shape, generator=gen)  # noqa: E731
    m = bs * n
    batch = synthetic.make_batch(bs, n, seed=29 + n)
    compat = ops.spatial_compat(g(batch["src_keypts"]), g(batch["tgt_keypts"]), g(torch.tensor([0.1])))
    qkv = torch.cat([rnd(m, 128) * 0.3 * QSCALE, rnd(m, 128) * 0.3, rnd(m, 128)], dim=-1)
    qs, kv = ops.pack_qkv_split(g(qkv), bs, n)
    msg = ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=nsplit)
    pf, _ = ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=nsplit, merge=False, layout="pf")
    res = g(rnd(m, 128))
    res_pf = ops.rows_to_pf(res, bs, n)
    tail_w = [g(x) for x in (rnd(64, 128) / 11, rnd(64), rnd(64, 64) / 8, rnd(64), rnd(128, 64) / 8, rnd(128))]
    head_w = [g(x) for x in (rnd(128, 128) / 11, rnd(128), rnd(384, 128) / 11, rnd(384))]
    npad, prow = (n + 255) // 256 * 256, ops.pf_rows(n)
    po = pf.view(torch.float32)[:bs * nsplit * npad * 128].reshape(bs, -1)
    ml = pf.view(torch.float32)[bs * nsplit * npad * 128:bs * nsplit * npad * 130].reshape(bs, -1)
    q_bytes, kv_bytes = qs.numel() // bs, kv.numel() // bs
    flags = ops.PF_PARTIALS | ops.PF_RES | ops.PF_FEATB

    _, fb_io, qs_io, kv_io = ops.layer_fused_io(res_pf, None, tail_w, head_w, bs, n, flags, partials=(pf, nsplit))
    _, fb_r, _, qs_r, kv_r = ops.layer_fused_split(msg, res, None, tail_w, head_w, bs, n, frag=True, gemm="h3", want_feat=False)
    _, fb_h, _, qs_h, kv_h = ops.layer_fused_split(None, None, res, None, head_w, bs, n, frag=True, gemm="h3")
    ft_t, _, _, _, _ = ops.layer_fused_split(msg, res, None, tail_w, None, bs, n, frag=True, gemm="h3")
    for b in range(bs):
        rows = slice(b * n, (b + 1) * n)
        one = torch.cat([po[b], ml[b]]).contiguous().view(torch.uint8)
        _, fb1, qs1, kv1 = ops.layer_fused_io(res_pf.view(bs, -1)[b].contiguous(), None, tail_w, head_w, 1, n, flags, partials=(one, nsplit))
        assert torch.equal(fb1, fb_io.view(bs, -1)[b])
        assert torch.equal(qs1, qs_io.view(bs, -1)[b]) and torch.equal(kv1, kv_io.view(bs, -1)[b])
        _, fb1, _, qs1, kv1 = ops.layer_fused_split(msg[rows].contiguous(), res[rows].contiguous(), None, tail_w, head_w, 1, n, frag=True,
                                                    gemm="h3", want_feat=False)
        assert torch.equal(fb1, fb_r[rows]) and torch.equal(qs1, qs_r.view(bs, -1)[b]) and torch.equal(kv1, kv_r.view(bs, -1)[b])
        _, fb1, _, qs1, kv1 = ops.layer_fused_split(None, None, res[rows].contiguous(), None, head_w, 1, n, frag=True, gemm="h3")
        assert torch.equal(fb1, fb_h[rows]) and torch.equal(qs1, qs_h.view(bs, -1)[b]) and torch.equal(kv1, kv_h.view(bs, -1)[b])
        ft1, _, _, _, _ = ops.layer_fused_split(msg[rows].contiguous(), res[rows].contiguous(), None, tail_w, None, 1, n, frag=True, gemm="h3")
        assert torch.equal(ft1, ft_t[rows])
    assert q_bytes * bs == qs_io.numel() and kv_bytes * bs == kv_io.numel() and prow * 128 * bs == fb_io.numel()


@pytest.mark.parametrize("n,bs,nsplit", [(33, 2, 2), (1000, 3, 3), (3001, 2, 2), (5000, 2, 2), (5000, 5, 4)])
def test_point_fragment_hand_offs_reproduce_the_row_order_chain(n, bs, nsplit):
    """The forward with layer_gemm = "h3" hands the key-split partials (attention -> layer kernel) and featB (layer kernel ->
    next layer kernel's residual) over in point-fragment order (csrc/split_layout.h): each lane stores / loads its own
    accumulator registers, 1 KiB of consecutive memory per instruction, no LDS transposition on either side.  Same arithmetic:
    the partials are the row-order partials permuted, and the layer kernel's streams / rows agree bit for bit with the
    row-order chain, ragged last tiles included (padding rows = copies of the pair's last row)."""
    gen = torch.Generator().manual_seed(500 + n)
    rnd = lambda *shape: torch.randn(*shape, generator=gen)  # noqa: E731
    m = bs * n
    batch = synthetic.make_batch(bs, n, seed=9 + n)
    compat = ops.spatial_compat(g(batch["src_keypts"]), g(batch["tgt_keypts"]), g(torch.tensor([0.1])))
    qkv = torch.cat([rnd(m, 128) * 0.3 * QSCALE, rnd(m, 128) * 0.3, rnd(m, 128)], dim=-1)
    qs, kv = ops.pack_qkv_split(g(qkv), bs, n)
    rows = ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=nsplit, merge=False)
    pf = ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=nsplit, merge=False, layout="pf")
    npad = (n + 255) // 256 * 256
    cut = bs * nsplit * npad * 128
    po_rows = rows[0].view(torch.float32)[:cut].reshape(bs * nsplit, npad, 128)
    po_pf = ops.pf_to_rows(pf[0].view(torch.float32)[:cut], bs * nsplit, npad)
    assert torch.equal(po_pf[:, :n], po_rows[:, :n])
    last_tile_end = (n + 31) // 32 * 32
    assert torch.equal(po_pf[:, n:last_tile_end], po_pf[:, n - 1:n].expand(-1, last_tile_end - n, -1))     # padding = copies of row n-1
    ml_rows = rows[0].view(torch.float32)[cut:cut + bs * nsplit * npad * 2].reshape(bs * nsplit, npad, 2)
    ml_pf = pf[0].view(torch.float32)[cut:cut + bs * nsplit * npad * 2].reshape(bs * nsplit, npad, 2)
    assert torch.equal(ml_pf[:, :n], ml_rows[:, :n])

    res = g(rnd(m, 128))
    res_pf = ops.rows_to_pf(res, bs, n)
    tail_w = [g(x) for x in (rnd(64, 128) / 11, rnd(64), rnd(64, 64) / 8, rnd(64), rnd(128, 64) / 8, rnd(128))]
    head_w = [g(x) for x in (rnd(128, 128) / 11, rnd(128), rnd(384, 128) / 11, rnd(384))]
    _, fb_r, _, qs_r, kv_r = ops.layer_fused_split(None, res, None, tail_w, head_w, bs, n, frag=True, gemm="h3", partials=rows, want_feat=False)
    _, fb_p, qs_p, kv_p = ops.layer_fused_io(res_pf, None, tail_w, head_w, bs, n, ops.PF_PARTIALS | ops.PF_RES | ops.PF_FEATB, partials=pf)
    assert torch.equal(qs_p, qs_r) and torch.equal(kv_p, kv_r)
    fb_back = ops.pf_to_rows(fb_p, bs, ops.pf_rows(n))
    assert torch.equal(fb_back[:, :n].reshape(m, 128), fb_r)
    assert torch.equal(fb_back[:, n:], fb_back[:, n - 1:n].expand(-1, ops.pf_rows(n) - n, -1))
    # mixed: row-order partials + PF residual, row-order featB out
    _, fb_m, qs_m, kv_m = ops.layer_fused_io(res_pf, None, tail_w, head_w, bs, n, ops.PF_RES, partials=rows)
    assert torch.equal(fb_m.reshape(m, 128), fb_r) and torch.equal(qs_m, qs_r) and torch.equal(kv_m, kv_r)
    # tail only (the forward's last layer): feat in row order
    f_r = ops.layer_fused_split(None, res, None, tail_w, None, bs, n, frag=True, gemm="h3", partials=rows)[0]
    f_p = ops.layer_fused_io(res_pf, None, tail_w, None, bs, n, ops.PF_PARTIALS | ops.PF_RES, partials=pf)[0]
    assert torch.equal(f_p, f_r)
    # head only (the forward's first launch): feat_in rows -> featB PF
    _, fb_h, qs_h, kv_h = ops.layer_fused_io(None, f_r, None, head_w, bs, n, ops.PF_FEATB)
    _, fb_hr, _, qs_hr, kv_hr = ops.layer_fused_split(None, None, f_r, None, head_w, bs, n, frag=True, gemm="h3")
    assert torch.equal(ops.pf_to_rows(fb_h, bs, ops.pf_rows(n))[:, :n].reshape(m, 128), fb_hr) and torch.equal(qs_h, qs_hr) and torch.equal(kv_h, kv_hr)


@pytest.mark.parametrize("layout", ["pf", "rows"])
@pytest.mark.parametrize("n,bs,nsplit", [(5000, 5, 2), (4100, 3, 3), (2053, 4, 1), (1000, 2, 4), (288, 3, 2)])
def test_attention_without_wasted_work_writes_the_same_partials(n, bs, nsplit, layout, monkeypatch):
    """(experiments builds only)  Two r04 savings of the split-precision attention launch, against the r01-r03 behaviour kept
    behind A/B knobs: waves whose 32 queries all lie past the pair's last row skip their arithmetic (PDSC_ATT_ALL_WAVES=1: they
    compute), and the key split's last tile runs as peeled tail code without the QK^T of the tile after it (PDSC_ATT_PEEL=0:
    straight-line loop).  Neither touches anything a consumer reads: partials (rows below the pair's last 32-row tile) and (m, l)
    (rows below N) agree bit for bit, in both hand-off orders, with whole idle waves (n = 5000: 3 of 8, n = 2053: 7 of 8, n = 288:
    7 of 8 in the second block), the four-wavefront kernel of small launches and the unsplit path (merged rows)."""
    if not _lib.load().pdsc_experiments_enabled():
        pytest.skip("the r01-r03 behaviour is an A/B record: experiments builds only")
    gen = torch.Generator().manual_seed(700 + n)
    batch = synthetic.make_batch(bs, n, seed=13 + n)
    compat = ops.spatial_compat_u16(g(batch["src_keypts"]), g(batch["tgt_keypts"]), g(torch.tensor([0.1])))
    qkv = torch.cat([torch.randn(bs * n, 128, generator=gen) * 0.3 * QSCALE, torch.randn(bs * n, 128, generator=gen) * 0.3,
                     torch.randn(bs * n, 128, generator=gen)], dim=-1)
    qs, kv = ops.pack_qkv_split(g(qkv), bs, n)
    npad = (n + 255) // 256 * 256
    cut = bs * nsplit * npad * 128

    def run():
        if nsplit == 1:
            return ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=1).clone(), None
        t = ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=nsplit, merge=False, layout=layout)[0].view(torch.float32)
        o = t[:cut].reshape(bs * nsplit, npad, 128)
        o = (ops.pf_to_rows(t[:cut], bs * nsplit, npad) if layout == "pf" else o)[:, : (n + 31) // 32 * 32 if layout == "pf" else n]
        return o.clone(), t[cut:cut + bs * nsplit * npad * 2].reshape(bs * nsplit, npad, 2)[:, :n].clone()

    monkeypatch.setenv("PDSC_ATT_ALL_WAVES", "1")
    monkeypatch.setenv("PDSC_ATT_PEEL", "0")
    want = run()
    for env in ({"PDSC_ATT_ALL_WAVES": "0", "PDSC_ATT_PEEL": "0"}, {"PDSC_ATT_ALL_WAVES": "1", "PDSC_ATT_PEEL": "1"},
                {"PDSC_ATT_ALL_WAVES": "0", "PDSC_ATT_PEEL": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = run()
        assert torch.equal(got[0], want[0]), env
        assert want[1] is None or torch.equal(got[1], want[1]), env


@pytest.mark.parametrize("fmt", ["f32", "u16"])
@pytest.mark.parametrize("n,bs,nsplit", [(5000, 13, 2), (4100, 16, 2), (2053, 32, 2)])
def test_persistent_attention_writes_the_same_partials(n, bs, nsplit, fmt, monkeypatch):
    """(experiments builds only: POINTDSC_HIP_LIB=pointdsc_amd/libpointdsc_hip_exp.so)  PDSC_ATT_PERSIST=1: one workgroup per CU walks its (pair, key split, query block) items, the run-ahead loads of an
    item's last tiles fetching the next item's first tiles.  Same arithmetic per item: the point-fragment partials (O and
    (m, l)) must be the one-item-per-workgroup kernel's bit for bit -- uneven item counts per workgroup, ragged last query
    blocks (other compat row clamp from one item to the next), both compat formats."""
    if not _lib.load().pdsc_experiments_enabled():
        pytest.skip("the persistent form is an A/B record: experiments builds only")
    gen = torch.Generator().manual_seed(600 + n)
    batch = synthetic.make_batch(bs, n, seed=11 + n)
    sig = g(torch.tensor([0.1]))
    compat = (ops.spatial_compat_u16 if fmt == "u16" else ops.spatial_compat)(g(batch["src_keypts"]), g(batch["tgt_keypts"]), sig)
    qkv = torch.cat([torch.randn(bs * n, 128, generator=gen) * 0.3 * QSCALE, torch.randn(bs * n, 128, generator=gen) * 0.3,
                     torch.randn(bs * n, 128, generator=gen)], dim=-1)
    qs, kv = ops.pack_qkv_split(g(qkv), bs, n)
    monkeypatch.setenv("PDSC_ATT_PERSIST", "0")
    want = ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=nsplit, merge=False, layout="pf")[0].clone()
    monkeypatch.setenv("PDSC_ATT_PERSIST", "1")
    got = ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=nsplit, merge=False, layout="pf")[0]
    npad = (n + 255) // 256 * 256
    cut = bs * nsplit * npad * 128
    a_o, b_o = (t.view(torch.float32)[:cut].reshape(bs * nsplit, npad, 128)[:, : (n + 31) // 32 * 32] for t in (want, got))
    assert torch.equal(a_o, b_o)
    a_ml, b_ml = (t.view(torch.float32)[cut:cut + bs * nsplit * npad * 2].reshape(bs * nsplit, npad, 2)[:, :n] for t in (want, got))
    assert torch.equal(a_ml, b_ml)


def test_layer_fused_frag_h3_small_and_large_magnitudes():
    """The fp16 halves of H3 at the edges of their range: activations of order 1e-5 (hi parts are fp16 subnormals: they
    must not be flushed by the conversions or the MFMA) and of order 1e3, zero biases so nothing masks them."""
    gen = torch.Generator().manual_seed(17)
    rnd = lambda *shape: torch.randn(*shape, generator=gen)  # noqa: E731
    n, bs = 700, 2
    m = bs * n
    z = lambda k: torch.zeros(k)  # noqa: E731
    tail_w = [g(x) for x in (rnd(64, 128) / 11, z(64), rnd(64, 64) / 8, z(64), rnd(128, 64) / 8, z(128))]
    head_w = [g(x) for x in (rnd(128, 128) / 11, z(128), rnd(384, 128) / 11, z(384))]
    d = lambda t: t.cpu().double()  # noqa: E731
    for mag in (1e-5, 1.0, 1e3):
        msg, res = rnd(m, 128) * mag, rnd(m, 128) * mag
        f1, fb1, _, _, _ = ops.layer_fused_split(g(msg), g(res), None, tail_w, head_w, bs, n, want_qkv=True, frag=True, gemm="h3")
        feat = d(res) + torch.relu(torch.relu(d(msg) @ d(tail_w[0]).T) @ d(tail_w[2]).T) @ d(tail_w[4]).T
        featB = torch.relu(feat @ d(head_w[0]).T)
        e = max(float((d(f1) - feat).abs().max() / feat.abs().max()), float((d(fb1) - featB).abs().max() / featB.abs().max()))
        print(f"magnitude {mag:g}: relative error {e:.2e}")
        assert e < 3e-6, (mag, e)


@pytest.mark.parametrize("n,bs,nsplit", [(257, 1, 2), (1000, 2, 3), (5000, 2, 2)])
def test_layer_fused_frag_h3_merges_attention_partials(n, bs, nsplit):
    gen = torch.Generator().manual_seed(n + 2)
    rnd = lambda *shape: torch.randn(*shape, generator=gen)  # noqa: E731
    batch = synthetic.make_batch(bs, n, seed=6 + n)
    compat = ops.spatial_compat(g(batch["src_keypts"]), g(batch["tgt_keypts"]), g(torch.tensor([0.1])))
    qkv = torch.cat([rnd(bs * n, 128) * 0.3 * QSCALE, rnd(bs * n, 128) * 0.3, rnd(bs * n, 128)], dim=-1)
    qs, kv = ops.pack_qkv_split(g(qkv), bs, n)
    msg = ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=nsplit)
    partials = ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=nsplit, merge=False)
    res = rnd(bs * n, 128)
    tail_w = [g(x) for x in (rnd(64, 128) / 11, rnd(64), rnd(64, 64) / 8, rnd(64), rnd(128, 64) / 8, rnd(128))]
    head_w = [g(x) for x in (rnd(128, 128) / 11, rnd(128), rnd(384, 128) / 11, rnd(384))]
    a = ops.layer_fused_split(msg, g(res), None, tail_w, head_w, bs, n, want_qkv=True, frag=True, gemm="h3")
    b = ops.layer_fused_split(None, g(res), None, tail_w, head_w, bs, n, want_qkv=True, frag=True, partials=partials, gemm="h3")
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("n,bs", [(1000, 10), (2053, 5)])
def test_layer_gemm_h3_agrees_with_fp32_gemms_through_the_encoder(n, bs):
    """model.layer_gemm = "h3" vs "f32" through the 12 layers on the wavefront-resident layer kernels (batches large enough
    that the size rule does not pick the workgroup-per-tile kernel, which has no H3 form): features within 2e-6, same seeds."""
    assert _lib.load().pdsc_layer_prefers_block(bs, n) == 0
    c = case(n)
    model = c["model"]
    batch = synthetic.make_batch(bs, n, seed=21, inlier_ratio=0.3)
    s_per = int(n * 0.1)
    out = {}
    for gemm in ("f32", "h3"):
        model.layer_gemm = gemm
        res = _forward(model, batch)
        out[gemm] = (model.workspace_view("featA", bs, n)[: bs * n * 128].reshape(bs * n, 128).cpu().clone(),
                     model.workspace_view("seeds", bs, n, torch.int32)[: bs * s_per].cpu().clone(), res)
    model.layer_gemm = LAYER_GEMM_DEFAULT
    scale = max(1.0, float(out["f32"][0].abs().max()))
    err = float((out["f32"][0] - out["h3"][0]).abs().max()) / scale
    print(f"feature difference h3 vs f32 GEMMs: {err:.2e}")
    assert err < 2e-6
    assert not torch.equal(out["f32"][0], out["h3"][0]), "the H3 path did not run"
    for i in range(bs):      # same seed sets (two keys closer than the feature difference may trade the last place)
        a, b = set(out["f32"][1][i * s_per:(i + 1) * s_per].tolist()), set(out["h3"][1][i * s_per:(i + 1) * s_per].tolist())
        assert len(a ^ b) <= 2, (i, sorted(a ^ b))
    # poses: a pair whose hypothesis ranking sits on a near-tie may land on another seed (DESIGN.md "tolerance edge": ~2-5 % of
    # random pairs at N = 1000, in the reference's own fp32-vs-fp64 comparison too): at most one of the batch, all within 1e-4
    dT = (out["f32"][2]["final_trans"] - out["h3"][2]["final_trans"]).abs().amax(dim=(1, 2))
    print("pose difference h3 vs f32 GEMMs per pair:", [f"{float(x):.1e}" for x in dT])
    assert int((dT >= 1e-5).sum()) <= 2, dT.tolist()       # (what the poses must satisfy is the census's business)


@pytest.mark.parametrize("n", [257, 1000])
def test_split_and_fp32_attention_agree_through_the_encoder(n):
    """Split precision vs exact fp32 through the 12 layers: features within 8e-6 (attention split, default) /
    3e-5 (GEMMs split too), identical seed sets and labels, R/t within 1e-5."""
    c = case(n)
    model = c["model"]
    out = {}
    modes = (("fp32", "f32"), ("fp16x3", "u16"), ("fp16x3", "f32")) + ((("fp16x3_all", "u16"),) if EXPERIMENTS else ())
    for prec, fmt in modes:
        model.attention_precision, model.compat_format = prec, fmt
        res = _forward(model, c["pair"])
        out[prec if fmt == "u16" or prec == "fp32" else prec + "_f32compat"] = (
            model.workspace_view("featA", 1, n)[: n * 128].reshape(n, 128).cpu().clone(),
            model.workspace_view("seeds", 1, n, torch.int32)[: int(n * 0.1)].cpu().clone(), res)
    model.attention_precision, model.compat_format = "fp16x3", COMPAT_FORMAT_DEFAULT
    scale = max(1.0, float(out["fp32"][0].abs().max()))
    print("feature error vs fp32:", {k: float((out["fp32"][0] - v[0]).abs().max()) / scale for k, v in out.items()})
    for prec, tol in (("fp16x3", 8e-6), ("fp16x3_f32compat", 8e-6)) + ((("fp16x3_all", 3e-5),) if EXPERIMENTS else ()):
        assert (out["fp32"][0] - out[prec][0]).abs().max() < tol * scale, prec
        # same seed SET (two seeds whose confidence differs by less than the feature tolerance may swap ranks)
        assert set(out["fp32"][1].tolist()) == set(out[prec][1].tolist()), prec
        assert torch.equal(out["fp32"][2]["final_labels"], out[prec][2]["final_labels"])
        assert (out["fp32"][2]["final_trans"] - out[prec][2]["final_trans"]).abs().max() < 1e-5


# ------------------------------------------------------------------------------------------------------
# encoder end to end (a-2 + a-3 chained over 12 layers) and a-4
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("n", [257, 1000])
def test_encoder_and_head_match_oracle(n, precision):
    c = case(n)
    c["model"].attention_precision = precision
    data = {k: g(c["pair"][k]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    data["testing"] = True
    c["model"](data)
    torch.cuda.synchronize()
    feat = c["model"].workspace_view("featA", 1, n)[: n * 128].reshape(n, 128).cpu()
    scale = float(c["st"]["feat"].abs().max())
    assert (feat - c["st"]["feat"]).abs().max() < 3e-5 * max(scale, 1.0)          # 12 layers of fp32 roundoff
    normed = c["model"].workspace_view("normed", 1, n)[: n * 128].reshape(n, 128).cpu()
    assert (normed - c["st"]["normed"]).abs().max() < 2e-5
    conf = c["model"].workspace_view("conf", 1, n)[:n].cpu()
    assert (conf - c["st"]["confidence"]).abs().max() < 3e-5 * max(scale, 1.0)
    c["model"].attention_precision = "fp16x3"


def test_normalize_confidence_stage():
    c = case(1000)
    sd, feat = c["sd"], c["st"]["feat"]
    h1 = torch.relu(feat @ sd["classification.0.weight"][:, :, 0].T + sd["classification.0.bias"])
    h2 = torch.relu(h1 @ sd["classification.2.weight"][:, :, 0].T + sd["classification.2.bias"])
    normed, conf = ops.normalize_confidence(g(feat), g(h2), g(sd["classification.4.weight"]), g(sd["classification.4.bias"]))
    assert (normed.cpu() - c["st"]["normed"]).abs().max() < 1e-6
    assert (conf.cpu() - c["st"]["confidence"]).abs().max() < 1e-5
    zero = torch.zeros(4, 128)                                                     # eps branch of F.normalize
    nz, _ = ops.normalize_confidence(g(zero), g(torch.zeros(4, 32)), g(sd["classification.4.weight"]), g(sd["classification.4.bias"]))
    assert float(nz.abs().sum()) == 0.0


# ------------------------------------------------------------------------------------------------------
# a-5 NMS seeds (fed with the oracle's confidence: result must be identical)
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [257, 1000, 2053])
def test_nms_keys_and_seeds_bit_exact(n):
    c = case(n)
    src, conf = g(c["pair"]["src_keypts"]), g(c["st"]["confidence"][None])
    keys = ops.nms_keys(src, conf, KW["nms_radius"])
    assert torch.equal(keys[0].cpu(), c["st"]["nms_keys"])
    seeds = ops.rank_select(keys, int(n * KW["ratio"]))
    assert torch.equal(seeds[0].cpu().long(), c["st"]["seeds"])


@pytest.mark.parametrize("n,bs,scale,radius", [(257, 1, 3.0, 0.1), (1000, 3, 3.0, 0.1), (5000, 2, 3.0, 0.1), (3000, 2, 60.0, 0.6),
                                               (2000, 1, 3.0, 5.0), (1500, 2, 3.0, 1e-4), (700, 1, 3.0, 0.0), (10000, 1, 3.0, 0.1)])
def test_nms_keys_grid_is_bit_identical_to_the_n2_kernel(n, bs, scale, radius):
    """pdsc_nms_keys_grid (cell grid + 3 x 3 window, what the forward calls) == pdsc_nms_keys (all N^2 pairs), bit for bit:
    dense and sparse radii (one cell / 64 x 64 cells), clustered points, duplicates, radius 0 (fallback), batches."""
    gen = torch.Generator().manual_seed(n + bs)
    batch = synthetic.make_batch(bs, n, seed=60 + n, scale=scale, noise=scale / 300.0)
    src = batch["src_keypts"].clone()
    src[:, : n // 10] = src[:, n // 10: 2 * (n // 10)]              # exact duplicates
    src[:, -(n // 8):] = src[:, -(n // 8):] * 0.01 + 1.0             # a tight cluster (many points per cell)
    conf = torch.randn(bs, n, generator=gen)
    conf[:, ::7] = conf[:, 1::7][:, : conf[:, ::7].shape[1]]         # equal confidences
    a = ops.nms_keys(g(src), g(conf), radius)
    b = ops.nms_keys_grid(g(src), g(conf), radius)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))     # incl. the sign of suppressed zeros
    assert 0 < int((a != g(conf)).sum()) < bs * n or radius <= 1e-4


def test_seed_ties_resolve_by_ascending_index():
    n = 500
    pair = synthetic.make_pair(n, seed=3)
    conf = torch.full((1, n), -0.25)                 # all-negative, all-equal logits: keys tie at -0.25 / -0
    conf[0, ::7] = -0.5
    src_dist, _ = O.spatial_compat(pair["src_keypts"][0], pair["tgt_keypts"][0], torch.tensor([0.1]))
    want = O.pick_seeds(src_dist, conf[0], 0.1, 50)
    got = ops.pick_seeds(g(pair["src_keypts"]), g(conf), 0.1, 50)
    assert torch.equal(got[0].cpu(), want)


# ------------------------------------------------------------------------------------------------------
# a-6 kNN (fed with the oracle's features and seeds)
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [257, 1000, 2053])
def test_knn_of_seeds(n):
    c = case(n)
    normed, seeds = c["st"]["normed"], c["st"]["seeds"]
    k = min(KW["k"], n - 1)
    idx, dist = ops.knn_seeds(g(normed[None]), g(seeds[None].int()), k, return_dist=True)
    idx, dist = idx[0].cpu().long(), dist[0].cpu()
    want_dist = O.knn_dist_rows(normed, seeds)
    assert (dist - want_dist).abs().max() < 2e-6                                   # fp32 Gram rows
    # the selection itself is exact on the GPU's own distances: ascending (dist, index), rank 0 dropped
    order = torch.sort(dist, dim=-1, stable=True).indices[:, 1:k + 1]
    assert torch.equal(idx, order)
    # against the oracle's neighbour sets: only near-ties at the k-th boundary may differ
    want = c["st"]["knn_idx"]
    same = sum(set(a.tolist()) == set(b.tolist()) for a, b in zip(idx, want))
    assert same >= 0.9 * len(seeds)
    for s in range(len(seeds)):
        diff = set(idx[s].tolist()) ^ set(want[s].tolist())
        if diff:
            # ... or at the rank-0 boundary: the dropped "self" is whichever near-duplicate sorts first
            # (reference models/common.py:68 only ASSUMES rank 0 is the point itself)
            vals = torch.sort(want_dist[s]).values
            first, kth = float(vals[0]), float(vals[k])
            assert all(min(abs(float(want_dist[s, j]) - kth), abs(float(want_dist[s, j]) - first)) < 1e-5 for j in diff), \
                [(j, float(want_dist[s, j]), first, kth) for j in diff]


# ------------------------------------------------------------------------------------------------------
# a-7 / a-8 / a-9 seed solver (fed with the oracle's neighbour lists)
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [257, 1000])
def test_seed_matrices_power_iteration_and_transforms(n):
    c = case(n)
    st, sd = c["st"], c["sd"]
    normed, src, tgt = g(st["normed"][None]), g(c["pair"]["src_keypts"]), g(c["pair"]["tgt_keypts"])
    knn = g(st["knn_idx"][None].int())
    iters, mask, M = ops.seed_power_iteration(normed, src, tgt, knn, g(sd["sigma"]), g(sd["sigma_spat"]), 10, want_M=True)
    assert (M[0].cpu() - st["seed_M"]).abs().max() < 2e-5            # (ds-dt)^2/sigma^2 amplifies 1 ulp of distance
    k = st["knn_idx"].shape[1]
    ran = st["power_iters"]
    m = int(mask[0].item()) & 0x3FF
    chosen = (m & -m).bit_length() - 1 if m else 9
    assert chosen == ran - 1 or (ran == 10 and m == 0)
    vec = iters[0, :, chosen, :k].cpu()
    assert (vec - st["eigvec"]).abs().max() < 2e-5
    assert float(iters[0, :, :, k:].abs().sum()) == 0.0
    trans, w = ops.seed_transforms(src, tgt, knn, iters, mask, 10)
    assert (w[0].cpu() - st["seed_weights"]).abs().max() < 2e-6
    d = (trans[0].cpu() - st["seed_trans"]).abs().amax(dim=(1, 2))
    # hypotheses from near-degenerate neighbourhoods are ill-conditioned for ANY solver; the bulk must agree tightly
    assert float((d < 1e-4).float().mean()) > 0.97, float((d < 1e-4).float().mean())
    assert float(d.median()) < 5e-6


_TRAINED_CASES = {}


def trained_case(pair_index):
    """Oracle stages of pair `pair_index` of the trained-like N = 1000 family (tests/golden/trained_3dmatch.npz weights)."""
    if pair_index not in _TRAINED_CASES:
        name = "trained_n1000_b1"
        w = workloads.WORKLOADS[name]
        kw = dict(w["model"])
        sd = workloads.state_dict(name, PointDSC(**kw).state_dict())
        pair = workloads.batch(name, pair_index, 1)
        res = O.forward_testing(sd, pair["corr_pos"], pair["src_keypts"], pair["tgt_keypts"], return_stages=True,
                                **{k: kw[k] for k in ORACLE_KEYS})
        _TRAINED_CASES[pair_index] = dict(sd=sd, pair=pair, st=res["stages"][0], kw=kw)
    return _TRAINED_CASES[pair_index]


@pytest.mark.parametrize("pair_index", [2, 3, 7])            # 20 %, 40 % and 40 % inliers (inlier cycle 5 / 10 / 20 / 40 %)
def test_discrete_stages_are_exact_on_trained_features(pair_index):
    """the seeded-weight stage tests above tolerate 10 % of the seeds with another neighbour set and 3 % of the
    hypotheses off by more than 1e-4, because seeded weights collapse the feature space (top-k gaps of 5e-7).  On the trained-like
    checkpoint there is no such excuse: each stage is fed the ORACLE's own inputs (its normalised features, its seeds, its neighbour
    lists), so only the stage's arithmetic differs, and the discrete results must be EQUAL --
      a-6: the 40-neighbour set of every seed (100 %, no allowance);
      a-7/8/9: every hypothesis within 1e-4 (100 %; median 5e-6), weights within 2e-6;
      a-10: the vote count of every hypothesis and the chosen one, computed from the oracle's hypotheses: equal."""
    c = trained_case(pair_index)
    st, sd, pair = c["st"], c["sd"], c["pair"]
    n = pair["corr_pos"].shape[1]
    k = min(c["kw"]["k"], n - 1)
    normed, seeds = st["normed"], st["seeds"]
    idx = ops.knn_seeds(g(normed[None]), g(seeds[None].int()), k)[0].cpu().long()
    want = st["knn_idx"]
    differ = [s_ for s_ in range(len(seeds)) if set(idx[s_].tolist()) != set(want[s_].tolist())]
    assert differ == [], (pair_index, "seeds whose neighbour set differs", differ[:8])
    src, tgt = g(pair["src_keypts"]), g(pair["tgt_keypts"])
    knn = g(want[None].int())
    iters, mask, _ = ops.seed_power_iteration(g(normed[None]), src, tgt, knn, g(sd["sigma"]), g(sd["sigma_spat"]), 10)
    trans, w_ = ops.seed_transforms(src, tgt, knn, iters, mask, 10)
    assert (w_[0].cpu() - st["seed_weights"]).abs().max() < 2e-6
    d = (trans[0].cpu() - st["seed_trans"]).abs().amax(dim=(1, 2))
    assert float(d.max()) < 1e-4 and float(d.median()) < 5e-6, (pair_index, float(d.max()), float(d.median()))
    counts, best, _initial, labels = ops.score_hypotheses(g(st["seed_trans"][None]), src, tgt, c["kw"]["inlier_threshold"])
    assert torch.equal(counts[0].cpu().long(), st["counts"].long()), (pair_index, int((counts[0].cpu().long() != st["counts"].long()).sum()))
    assert int(best[0]) == int(st["best"]) and torch.equal(labels[0].cpu(), st["final_labels"])


def test_power_iteration_global_early_exit():
    """Rigid pair + identical features: every k x k block is c*(J-I) and converges at the 2nd iterate."""
    n, S, k = 200, 20, 40
    pair = synthetic.make_pair(n, inlier_ratio=1.1, noise=0.0, seed=9)
    feat = torch.ones(1, n, 128) / 128 ** 0.5
    knn = torch.stack([torch.randperm(n, generator=torch.Generator().manual_seed(s))[:k] for s in range(S)])[None].int()
    src, tgt = g(pair["src_keypts"]), g(pair["tgt_keypts"])
    iters, mask, M = ops.seed_power_iteration(g(feat), src, tgt, g(knn), g(torch.tensor([1.0])), g(torch.tensor([0.1])), 10, want_M=True)
    want_M = O.seed_matrices(feat[0], pair["src_keypts"][0], pair["tgt_keypts"][0], knn[0].long(), torch.tensor([1.0]), torch.tensor([0.1]))
    assert (M[0].cpu() - want_M).abs().max() < 1e-3        # compat of an exactly rigid pair is 1 up to rounding of distances
    vec, ran = O.power_iteration(M[0].cpu(), 10)
    m = int(mask[0].item()) & 0x3FF
    assert m != 0 and (m & -m).bit_length() == ran          # first converged iteration == oracle's break point
    assert (iters[0, :, ran - 1, :k].cpu() - vec).abs().max() < 1e-6


@pytest.mark.parametrize("bs,n", [(1, 3), (4, 40), (2, 1000), (3, 5000)])
def test_rigid_transform_3d_matches_lapack_path(bs, n):
    rs = np.random.RandomState(n + bs)
    A = torch.from_numpy(rs.standard_normal((bs, n, 3)).astype(np.float32)) * 2
    R = torch.from_numpy(np.stack([synthetic.random_rotation(rs) for _ in range(bs)]))
    B = A @ R.transpose(1, 2) + torch.from_numpy(rs.standard_normal((bs, 1, 3)).astype(np.float32))
    B = B + torch.from_numpy(rs.standard_normal((bs, n, 3)).astype(np.float32)) * 0.05
    w = torch.from_numpy(rs.random_sample((bs, n)).astype(np.float32))
    for weights, thr in ((None, 0.0), (w, 0.0), (w, 0.5)):
        if n == 3 and thr > 0:
            continue        # thresholding 3 points leaves a rank-deficient problem: R is not defined by the data
        T = ops.rigid_transform_3d(g(A), g(B), g(weights) if weights is not None else None, thr).cpu()
        want = O.rigid_transform_3d(A, B, weights, thr)
        assert (T - want).abs().max() < 2e-5, (bs, n, thr)
        Rg = T[:, :3, :3].double()
        assert (Rg @ Rg.transpose(1, 2) - torch.eye(3, dtype=torch.float64)).abs().max() < 1e-6
        assert (torch.det(Rg) - 1).abs().max() < 1e-6
    w_dev = g(w)
    ops.rigid_transform_3d(g(A), g(B), w_dev, 0.5)
    assert torch.equal(w_dev.cpu(), w)                      # unlike the reference, weights are not zeroed in place


def test_rigid_transform_3d_degenerate_inputs():
    rs = np.random.RandomState(0)
    # coplanar points (rank-2 covariance): R is still unique and must match the LAPACK path
    A = torch.from_numpy(rs.standard_normal((2, 60, 3)).astype(np.float32))
    A[:, :, 2] = 0.0
    R = torch.from_numpy(synthetic.random_rotation(rs))
    B = A @ R.T + torch.tensor([0.1, 0.2, 0.3])
    T = ops.rigid_transform_3d(g(A), g(B)).cpu()
    assert (T - O.rigid_transform_3d(A, B)).abs().max() < 2e-5
    # reflection case: the det correction must produce a proper rotation
    Bm = B.clone()
    Bm[:, :, 0] = -Bm[:, :, 0]
    Tm = ops.rigid_transform_3d(g(A + torch.from_numpy(rs.standard_normal((2, 60, 3)).astype(np.float32)) * 0.3), g(Bm)).cpu()
    assert (torch.det(Tm[:, :3, :3].double()) - 1).abs().max() < 1e-6
    # collinear / empty / all-zero-weight inputs: R not defined by the data, result must still be a finite rotation
    C = torch.zeros(1, 10, 3)
    C[0, :, 0] = torch.arange(10.0)
    for A_, B_, w_ in ((C, C * 2, None), (A[:1], B[:1], torch.zeros(1, 60)), (A[:1, :0], B[:1, :0], None)):
        Td = ops.rigid_transform_3d(g(A_), g(B_), g(w_) if w_ is not None else None).cpu()
        Rd = Td[:, :3, :3].double()
        assert torch.isfinite(Td).all() and (Rd @ Rd.transpose(1, 2) - torch.eye(3, dtype=torch.float64)).abs().max() < 1e-6


# ------------------------------------------------------------------------------------------------------
# a-10 / a-11 (fed with the oracle's hypotheses)
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [257, 1000, 2053])
def test_hypothesis_scoring_bit_exact(n):
    c = case(n)
    st = c["st"]
    src, tgt = g(c["pair"]["src_keypts"]), g(c["pair"]["tgt_keypts"])
    counts, best, initial, labels = ops.score_hypotheses(g(st["seed_trans"][None]), src, tgt, KW["inlier_threshold"])
    assert torch.equal(counts[0].cpu().long(), st["counts"])
    assert int(best[0]) == st["best"]
    assert torch.equal(initial[0].cpu(), st["seed_trans"][st["best"]])
    assert torch.equal(labels[0].cpu(), st["final_labels"])


def test_argmax_takes_first_of_equal_counts():
    n = 300
    pair = synthetic.make_pair(n, seed=4)
    T = torch.eye(4).repeat(1, 6, 1, 1)
    T[0, 4, 0, 3] = 0.5                                # different transform, fewer inliers
    counts, best, _, _ = ops.score_hypotheses(g(T), g(pair["src_keypts"]), g(pair["tgt_keypts"]), 0.1)
    assert int(best[0]) == 0 and int(counts[0, 0]) == int(counts[0, 1])


@pytest.mark.parametrize("n", [257, 1000, 2053])
def test_post_refinement_matches_oracle(n):
    c = case(n)
    st = c["st"]
    src, tgt = g(c["pair"]["src_keypts"]), g(c["pair"]["tgt_keypts"])
    final, solves = ops.post_refinement(g(st["initial_trans"][None]), src, tgt, 0.10, 20)
    assert int(solves[0]) == st["refine_solves"]
    assert (final[0].cpu() - st["final_trans"]).abs().max() < 1e-5
    # a perturbed start needs several re-solves: exercises the loop, must converge to the same pose
    start = st["initial_trans"].clone()
    start[:3, 3] += 0.03
    want, want_solves = O.post_refinement(start, c["pair"]["src_keypts"][0], c["pair"]["tgt_keypts"][0], 0.10)
    got, got_solves = ops.post_refinement(g(start[None]), src, tgt, 0.10, 20)
    assert int(got_solves[0]) == want_solves and (got[0].cpu() - want).abs().max() < 2e-5
    # no inliers at all: first iteration breaks, pose returned unchanged
    far = torch.eye(4)[None].clone()
    far[0, :3, 3] = 100.0
    same, s0 = ops.post_refinement(g(far), src, tgt, 0.10, 20)
    assert int(s0[0]) == 0 and torch.equal(same.cpu(), far)


# ------------------------------------------------------------------------------------------------------
# whole path: vs oracle, vs golden fixtures (= reference outputs), batching, edge sizes, errors
# ------------------------------------------------------------------------------------------------------
def _forward(model, pair_or_batch):
    data = {k: g(pair_or_batch[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    data["testing"] = True
    with torch.no_grad():
        res = model(data)
    torch.cuda.synchronize()
    return res


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("n", [257, 1000, 2053])
def test_forward_matches_oracle(n, precision):
    c = case(n)
    c["model"].attention_precision = precision
    res = _forward(c["model"], c["pair"])
    c["model"].attention_precision = "fp16x3"
    assert res["M"] is None and res["final_trans"].shape == (1, 4, 4) and res["final_labels"].shape == (1, n)
    assert torch.equal(res["final_labels"].cpu(), c["res"]["final_labels"])            # inlier mask bit-exact
    assert (res["final_trans"].cpu() - c["res"]["final_trans"]).abs().max() < 1e-4      # R/t within 1e-4
    re, te = O.registration_errors(res["final_trans"][0].cpu(), c["pair"]["gt_trans"][0])
    assert re < 1.0 and te < 5.0


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", ["n257_s0", "n1000_s1", "n1000_s2_defaultbn", "n2053_s3", "kitti_n1500_s4", "n5000_s5",
                                  "kitti_n5000_s8", "lomatch_n10000_s7"])
def test_forward_matches_reference_golden(name, precision):
    """Outputs of the unmodified reference (oracle/check_against_reference.py) incl. BASELINE.json configs[3] (KITTI,
    N=5000, sigma_d=1.2, threshold 0.6: evaluation/test_KITTI.py:166-170,188) and configs[4] (N=10000) sizes."""
    model, batch, fx = _golden_model_and_pair(name)
    model.attention_precision = precision
    res = _forward(model, batch)
    flips = int((res["final_labels"].cpu() != torch.from_numpy(fx["ref_final_labels"])).sum())
    dT = float((res["final_trans"].cpu() - torch.from_numpy(fx["ref_final_trans"])).abs().max())
    assert flips == 0, f"{flips} label flips vs the reference"
    assert dT < (1e-3 if bool(fx["tie_case"]) else 1e-4), dT


def _golden_model_and_pair(name):
    fx = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    kw = json.loads(str(fx["model_json"]))
    model = PointDSC(**kw)
    shift = float(fx["logit_shift"]) if "logit_shift" in fx.files else synthetic.DEFAULT_LOGIT_SHIFT
    sd = synthetic.make_state_dict(model.state_dict(), seed=int(fx["wseed"]), randomize_bn=bool(fx["randomize_bn"]),
                                   logit_shift=shift)
    assert abs(sum(float(v.double().sum()) for v in sd.values()) - float(fx["weights_checksum"])) < 1e-6
    model.load_state_dict(sd)
    model = model.eval().to(DEV)
    batch = {k: torch.from_numpy(fx[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    return model, batch, fx


@pytest.mark.parametrize("name,bs,pos", [("n5000_s5", 4, 1), ("n5000_s5", 8, 7), ("n5000_s5", 16, 0), ("n5000_s5", 32, 19),
                                         ("kitti_n5000_s8", 2, 1), ("kitti_n5000_s8", 16, 5),
                                         ("lomatch_n10000_s7", 2, 0), ("lomatch_n10000_s7", 8, 6)])
def test_reference_golden_pair_inside_a_batch(name, bs, pos):
    """The golden pair (reference output known) placed at position `pos` of a batch of bs pairs: the per-GPU shares of
    BASELINE.json configs[2..4] on 8/4/2/1 GPUs.  The batch size selects the launch plans (attention key split, layer
    kernel variant, merge of the partials) -- the result of a pair must not depend on its neighbours beyond 1e-4."""
    model, one, fx = _golden_model_and_pair(name)
    n = one["corr_pos"].shape[1]
    kw = json.loads(str(fx["model_json"]))
    scale = 60.0 if kw["sigma_d"] > 1.0 else 3.0
    fill = synthetic.make_batch(bs, n, seed=7000 + bs, inlier_ratio=0.25, scale=scale, noise=scale / 300.0)
    batch = {k: fill[k].clone() for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    for k in batch:
        batch[k][pos] = one[k][0]
    res = _forward(model, batch)
    flips = int((res["final_labels"][pos].cpu() != torch.from_numpy(fx["ref_final_labels"])[0]).sum())
    dT = float((res["final_trans"][pos].cpu() - torch.from_numpy(fx["ref_final_trans"])[0]).abs().max())
    assert flips == 0, f"{flips} label flips vs the reference"
    assert dT < 1e-4, dT


_BENCH_MODELS = {}

# (family, batch) -> {pair: rule} of test_bench_workload_matches_reference_golden: which of the timed pairs sit outside the strict
# fp32 contract and by which NAMED rule of tools/parity_census.py:explain (each with its own bounded check).  Everything not listed
# is held to labels bit-exact and R/t < 1e-4 against the reference's fp32 output.
_GOLDEN_PAIR_RULES = {
    # the reference's k-th / (k+1)-th neighbour distances of the winning seed are 1.2e-7 apart (recorded gap)
    ("multiway_n20000_b1", 1): {0: "knn-tie"},
}


LAYER_GEMM_DEFAULT = PointDSC().layer_gemm
COMPAT_FORMAT_DEFAULT = PointDSC().compat_format


def _bench_model(name):
    if name not in _BENCH_MODELS:
        w = workloads.WORKLOADS[name]
        model = PointDSC(**w["model"])
        sd = workloads.state_dict(name, model.state_dict())
        fx = np.load(GOLDEN / f"bench_{name}.npz", allow_pickle=False)
        assert abs(sum(float(v.double().sum()) for v in sd.values()) - float(fx["weights_checksum"])) < 1e-6
        model.load_state_dict(sd)
        _BENCH_MODELS[name] = (model.eval().to(DEV), fx)
    return _BENCH_MODELS[name]


@pytest.mark.parametrize("name,bs", [("n1000_b1", 1),
                                     ("n5000_b32", 4), ("n5000_b32", 8), ("n5000_b32", 16), ("n5000_b32", 32),
                                     ("kitti_n5000_b16", 2), ("kitti_n5000_b16", 4), ("kitti_n5000_b16", 8), ("kitti_n5000_b16", 16),
                                     ("lomatch_n10000_b8", 1), ("lomatch_n10000_b8", 2), ("lomatch_n10000_b8", 4), ("lomatch_n10000_b8", 8),
                                     # the reference's real evaluation sizes (evaluation/test_KITTI.py:120, multiway/test_multi_ate.py:245)
                                     ("kitti_n12000_b4", 4), ("kitti_n12000_b4", 1), ("multiway_n20000_b1", 1)])
def test_bench_workload_matches_reference_golden(name, bs):
    """THE TIMED PATH: the first bs pairs of a bench workload (bs = the per-GPU share on 8/4/2/1 GPUs; bs = global batch
    is exactly what `bench.py --config name` times on one GPU, same pairs, same weights, same launch plans) against
    the outputs of the unmodified reference on those pairs (oracle/make_bench_goldens.py, reference
    models/PointDSC.py:128-197 looped per pair): inlier masks bit-exact, R/t within 1e-4."""
    model, fx = _bench_model(name)
    w = workloads.WORKLOADS[name]
    n = w["num_corr"]
    batch = workloads.batch(name, 0, bs)
    g_pairs = min(bs, fx["ref_final_trans"].shape[0])
    chk = np.array([float(batch[k][:fx["ref_final_trans"].shape[0]].double().sum()) for k in ("corr_pos", "src_keypts", "tgt_keypts")])
    if bs >= fx["ref_final_trans"].shape[0]:
        assert np.allclose(chk, fx["input_checksum"], rtol=0, atol=1e-6), "synthetic inputs differ from the fixture's"
    res = _forward(model, batch)
    want_lab = torch.from_numpy(np.unpackbits(fx["ref_final_labels_bits"], axis=1)[:, :n].astype(np.float32))
    want_T = torch.from_numpy(fx["ref_final_trans"])
    flips = int((res["final_labels"][:g_pairs].cpu() != want_lab[:g_pairs]).sum())
    dT = (res["final_trans"][:g_pairs].cpu() - want_T[:g_pairs]).abs().amax(dim=(1, 2))
    assert flips == 0, f"{flips} label flips vs the reference"
    has_census = (GOLDEN / f"census_{name}.npz").exists() and (GOLDEN / f"census_internals_{name}.npz").exists()
    # R/t within 1e-4 -- or, where the reference's recorded decisions exist for the family, a NAMED discrete cause from those
    # records, checked below pair by pair like every other pair of the census (multiway_n20000_b1 pair 0: the reference's
    # own k-th / (k+1)-th neighbour distances of the winning seed are 1.2e-7 apart, so which of the two enters the set is
    # decided by the last bit of its fp32 Gram; bf16-split builds landed on the reference's side, the fp16-split build
    # does not).  Without such records the bound is strict.
    assert has_census or bool((dT < 1e-4).all()), dT.tolist()
    # ... and EVERY pair of the batch against the census fixture (reference fp32 and fp64 outputs of the same pairs)
    if not (GOLDEN / f"census_{name}.npz").exists():       # (the large-N workloads have the golden pairs only)
        return
    cfx = _census_fixture(name)
    ok, d32, dbest, f32, which = _census_judge(res["final_trans"], res["final_labels"], cfx, n)
    # a pair outside the fp32 contract needs a recorded discrete cause (test_parity_census; tools/parity_census.py:explain)
    mod = _census_module()
    ixp = GOLDEN / f"census_internals_{name}.npz"
    strict = (d32 < 1e-4) & (f32 == 0)
    used = {}
    if not bool(strict.all()):
        ix = np.load(ixp, allow_pickle=False)
        rxp = GOLDEN / f"census_refine_{name}.npz"
        rx = np.load(rxp, allow_pickle=False) if rxp.exists() else None
        dec = mod.decisions(model, bs, n)
        l32 = np.unpackbits(cfx["ref32_final_labels_bits"][:bs], axis=1)[:, :n]
        for i in np.flatnonzero(~strict.numpy()).tolist():
            flipped = np.flatnonzero((res["final_labels"][i].cpu().numpy() > 0) != (l32[i] > 0))
            okx, why = mod.explain(i, {k: v[i] for k, v in dec.items()}, ix, {k: batch[k][i] for k in ("src_keypts", "tgt_keypts")},
                                   float(w["model"]["inlier_threshold"]), float(w["pair"]["scale"]), flipped if float(d32[i]) < 1e-4 else None,
                                   nms_radius=float(w["model"]["nms_radius"]), rx=rx, T_here=res["final_trans"][i].cpu().numpy(),
                                   T_ref=cfx["ref32_final_trans"][i], label_flips=int(f32[i]))
            # r06: the reference not reproducing itself on a pair passes NOTHING by itself -- outside the fp32
            # contract a pair needs the reference's fp64 output under the same contract (ok[i]) or a NAMED rule with its bounded check
            assert okx or bool(ok[i]), (i, float(d32[i]), int(f32[i]), why)
            used[i] = "fp64 reference" if bool(ok[i]) and not okx else mod.rule_of(why)
    # the golden pairs this test is parametrised on use EXACTLY these rules (no silent widening -- a new pair or a new
    # rule on these pairs fails here and has to be added by name)
    assert used == _GOLDEN_PAIR_RULES.get((name, bs), {}), (name, bs, used)
    # every pair of the batch, golden or not, must register (well-conditioned workload) and be a rigid motion
    T = res["final_trans"].cpu().double()
    assert (T[:, :3, :3] @ T[:, :3, :3].transpose(1, 2) - torch.eye(3, dtype=torch.float64)).abs().max() < 1e-5
    for i in range(bs):
        re, te = O.registration_errors(res["final_trans"][i].cpu(), batch["gt_trans"][i])
        assert re < 1.0 and te < (60.0 if "kitti" in name else 5.0), (i, re, te)


_CENSUS = {}


def _census_fixture(name):
    if name not in _CENSUS:
        _CENSUS[name] = np.load(GOLDEN / f"census_{name}.npz", allow_pickle=False)
    return _CENSUS[name]


def _census_judge(trans, labels, fx, n, first=0):
    """tools/parity_census.py:judge -- the contract of BASELINE.json with no looser tolerance for any pair: labels bit-exact
    and R/t within 1e-4 of the reference's fp32 output, or (pairs on which the reference's own two precisions land on
    different hypotheses) of its fp64 output."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("parity_census", ROOT / "tools" / "parity_census.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.judge(trans, labels, fx, n, first)


def _census_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("parity_census", ROOT / "tools" / "parity_census.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("gemm", ["h3", "f32"])
@pytest.mark.parametrize("name,step", [("n5000_b32", 32), ("n5000_b32", 4), ("kitti_n5000_b16", 16), ("kitti_n5000_b16", 2),
                                       ("lomatch_n10000_b8", 8), ("lomatch_n10000_b8", 1), ("n1000_b1", 1), ("n1000_b1", 16),
                                       # 32 pairs at the reference's real KITTI evaluation size (evaluation/test_KITTI.py:120)
                                       ("kitti_n12000_b4", 4), ("kitti_n12000_b4", 1),
                                       # 16 pairs at the reference's multiway size (multiway/test_multi_ate.py:245)
                                       ("multiway_n20000_b1", 1), ("multiway_n20000_b1", 4)])
def test_parity_census(name, step, gemm):
    """Parity census: 256 seeded pairs per workload family (64 at N = 10 000, 32 at N = 12 000, 16 at N = 20 000), pair i = the bench workload's pair i, run in batches
    of the bench's global batch and of its 8-GPU share, with both layer-GEMM arithmetics.  Every pair must meet BASELINE.json's
    contract against the unmodified reference's fp32 output (labels bit-exact, R/t within 1e-4).  A pair outside it passes ONLY
    with a recorded discrete cause, checked against what the reference itself decided on that pair
    (tests/golden/census_internals_<name>.npz, oracle/make_census_internals.py; rules in tools/parity_census.py:explain):
      * the result equals the reference's fp64 output under the same contract (r06: that the reference does not reproduce itself
        on a pair passes nothing by itself any more), or
      * degenerate solve: the reference's own recorded singular values say its last refinement solve ran on two correspondences;
        the pose must then be a proper rigid motion that sends those correspondences where the reference's pose sends them, or
      * hypothesis tie: another hypothesis chosen, within one vote of the winner in the reference's votes AND in this run's, or
      * refinement: same hypothesis, the inlier-count sequence leaves the reference's by exactly one vote, or
      * kNN tie: the seed's 40-neighbour set differs and the reference recorded that seed's top-k boundary gap at round-off level, or
      * label edge: flipped labels sit within 8 ulps of the threshold under the reference's hypothesis.
    No allowance in number or size: one pair without such a record fails the test.  The records of the excused pairs are printed
    (and committed under profiles/ by tools/parity_census.py)."""
    model, _ = _bench_model(name)
    fx = _census_fixture(name)
    chk = sum(float(workloads.batch(name, 0, 1)[k][0].double().sum()) for k in ("corr_pos", "src_keypts", "tgt_keypts"))
    assert abs(chk - float(fx["input_checksum"][0])) < 1e-6, "synthetic inputs differ from the fixture's"
    mod = _census_module()
    try:
        rep, _m = mod.run_family(name, [step], layer_gemm=gemm, model=model)
    finally:
        model.layer_gemm = LAYER_GEMM_DEFAULT
    r = rep[step]
    assert r["unexcused"] is not None, f"tests/golden/census_internals_{name}.npz is missing"
    print(f"{name} x{step} {gemm}: median dT {r['median_dT']:.1e}; outside the fp32 contract {r['outside_fp32_contract']}; "
          f"reference not self-consistent on {r['reference_not_self_consistent']}")
    for d in r["outside_fp32_contract_detail"]:
        print("   ", json.dumps(d))
    assert r["unexcused"] == [], [d for d in r["outside_fp32_contract_detail"] if d["pair"] in r["unexcused"]]


TRAINED_CENSUS = [("trained_n1000_b1", 1), ("trained_n1000_b1", 16), ("trained_n5000_b32", 32), ("trained_n5000_b32", 4),
                  ("trained_kitti_n5000_b16", 16), ("trained_kitti_n5000_b16", 2), ("trained_lomatch_n10000_b8", 8), ("trained_lomatch_n10000_b8", 1),
                  ("trained_kitti_n12000_b4", 4), ("trained_kitti_n12000_b4", 1), ("trained_multiway_n20000_b1", 1)]


@pytest.mark.parametrize("arith", ["default", "exact_fp32"])
@pytest.mark.parametrize("name,step", TRAINED_CENSUS)
def test_parity_census_trained_like_weights(name, step, arith):
    """r05: the census on TRAINED-LIKE weights -- the unmodified reference trained with its own training
    forward and losses on synthetic pairs (oracle/make_trained_fixture.py) until its logits separate inliers (+-6 instead of the
    +-0.02 of seeded weights), inlier ratios cycling 5 / 10 / 20 / 40 %.  Same rules as test_parity_census with one more: in this regime
    top-k boundary gaps are not at round-off level, so NO pair may need the `knn-tie` rule.  Also checked: the Registration-Recall
    surrogate -- the reference's success / RE / TE columns (libs/loss.py:44-51) and this run's agree on every pair (success equal,
    RE within 0.1 deg, TE within 0.01 cm x scale) -- with the shipped arithmetic and with the exact-fp32 mode."""
    if not (GOLDEN / f"census_{name}.npz").exists():
        pytest.skip(f"tests/golden/census_{name}.npz not generated")
    model, _ = _bench_model(name)
    fx = _census_fixture(name)
    chk = sum(float(workloads.batch(name, 0, 1)[k][0].double().sum()) for k in ("corr_pos", "src_keypts", "tgt_keypts"))
    assert abs(chk - float(fx["input_checksum"][0])) < 1e-6, "synthetic inputs differ from the fixture's"
    mod = _census_module()
    over = dict(attention_precision="fp32", compat_format="f32", layer_gemm="f32") if arith == "exact_fp32" else {}
    try:
        rep, _m = mod.run_family(name, [step], model=model, **over)
    finally:
        model.layer_gemm, model.compat_format, model.attention_precision = LAYER_GEMM_DEFAULT, COMPAT_FORMAT_DEFAULT, "fp16x3"
    r = rep[step]
    print(f"{name} x{step} {arith}: strict pass rate {r['strict_fp32_contract_pass_rate']:.4f}, median dT {r['median_dT']:.1e}, max {r['max_dT_vs_fp32_reference']:.1e}; "
          f"outside the fp32 contract {r['outside_fp32_contract']}; excuses {r['excuses_used']}; registration {json.dumps(r['registration'])}")
    for d in r["outside_fp32_contract_detail"]:
        print("   ", json.dumps(d))
    assert r["unexcused"] == [], [d for d in r["outside_fp32_contract_detail"] if d["pair"] in r["unexcused"]]
    assert "knn-tie" not in r["excuses_used"], r["outside_fp32_contract_detail"]
    reg = r["registration"]
    assert reg["pairs_where_success_differs"] == [] and reg["recall_here"] == reg["recall_reference_fp32"], reg
    scale = float(workloads.WORKLOADS[name]["pair"]["scale"]) / 3.0
    strict_idx = set(range(r["pairs"])) - set(r["outside_fp32_contract"])
    if len(strict_idx) == r["pairs"]:
        # (RE = acos((tr - 1) / 2) at angles of 0.01 .. 0.05 deg: d(angle) = d(trace) / (2 sin(angle)) turns an fp32 ulp of the trace
        #  into several 1e-2 deg in the reference's number and in this one alike -- the column is compared at the 0.1 deg the
        #  reference's evaluation tables print)
        assert reg["max_abs_RE_diff_deg"] < 0.1 and reg["max_abs_TE_diff_cm"] < 0.01 * scale * 3.0, reg


# Bounds of test_trained_checkpoint_stage_decisions_follow_the_reference, per family: (share of seed-list positions that may differ,
# share of seeds whose neighbour set may differ, largest recorded top-k gap among those, share of hypotheses whose vote count may differ
# by one).  MEASURED (profiles/r06_stage_census.txt), then fixed at about twice the measurement; every single difference must ALSO
# carry its named near-tie from the reference's own records (below) -- the shares only keep the near-tie class from growing silently.
_STAGE_BOUNDS = {"trained_n1000_b1": (0.01, 0.01, 5e-3, 0.002), "trained_n5000_b32": (0.03, 0.015, 5e-3, 0.002),
                 "trained_kitti_n5000_b16": (0.06, 0.015, 5e-3, 0.002), "trained_lomatch_n10000_b8": (0.06, 0.015, 5e-3, 0.002)}
# measured (profiles/r06_stage_census.txt), default arithmetic | exact fp32 where run:
#   seed-list positions that differ      n1000 0.22 % | 0.05 %   n5000 0.92 %   KITTI 2.6 % | 1.5 %   N = 10 000 2.4 %
#     (every one between correspondences whose reference logits are closer than this run's own logit deviation on the pair: ratio <= 0.98)
#   neighbour sets that differ           n1000 0.29 % | 0.09 %   n5000 0.46 %   KITTI 0.56 % | 0.53 %   N = 10 000 0.51 %
#   largest recorded top-k gap among them  n1000 1.0e-4 | 1.2e-5   n5000 2.4e-4   KITTI 2.5e-3 | 8.7e-4   N = 10 000 2.6e-5
#   largest logit deviation / logit range  n1000 5.3e-3 | 1.1e-3   n5000 1.7e-3   KITTI 1.4e-2 | 2.7e-3   N = 10 000 1.9e-4
#   votes that differ by one             0 | 0, 1, 6 | 6, 0 of 22 k - 58 k hypotheses; by more: none; chosen hypothesis: never differs.
# i.e. exact fp32 in another summation order than torch's CPU kernels already leaves these classes non-empty (the floor); the split-
# precision default sits at 1.1 - 4 x that floor, and neither ever changes a vote by more than one or the hypothesis chosen.


@pytest.mark.parametrize("name,step,pairs,arith", [("trained_n1000_b1", 1, 64, "default"), ("trained_n1000_b1", 16, 256, "default"),
                                                   ("trained_n5000_b32", 32, 64, "default"), ("trained_kitti_n5000_b16", 16, 64, "default"),
                                                   ("trained_lomatch_n10000_b8", 8, 32, "default"),
                                                   # the same census with EXACT fp32 arithmetic (fp32 MFMA everywhere, fp32 matrix): the
                                                   # shares of near-tie decisions it leaves are the summation-order floor any fp32
                                                   # implementation other than torch's CPU kernels has; the default arithmetic must not
                                                   # be held to less than that (profiles/r06_stage_census.txt prints both)
                                                   ("trained_n1000_b1", 16, 256, "exact_fp32"), ("trained_kitti_n5000_b16", 16, 64, "exact_fp32")])
def test_trained_checkpoint_stage_decisions_follow_the_reference(name, step, pairs, arith):
    """the stage tests on discrete work (a-5 seeds, a-6 neighbour sets, a-10 votes) tolerate a few per cent
    of near-tie differences because SEEDED weights collapse the feature space (top-k boundary gaps of 5e-7).  Here the same decisions
    on the trained-like checkpoints, for every pair of the census, against what the unmodified reference itself decided
    (tests/golden/census_internals_<name>.npz: its logits, seeds, neighbour-set hashes, integer votes, chosen hypothesis, refinement
    inlier counts).  Bit-for-bit equality of EVERY decision is not attainable by any implementation that sums in another order than
    torch's CPU kernels (the logits of this checkpoint are 128-term fp32 dot products of +-30: exact fp32 in another order moves them
    by 1e-3 relative, measured) -- what IS demanded:
      * a-5: the seed list has the reference's seeds; two positions may trade places only if the reference's own recorded logits of
        the two correspondences differ by less than twice the largest |logit difference| between this run and the reference on that pair
        (itself bounded: 3e-2 x the pair's logit range; measured 1.4e-2 at the KITTI scale, exact fp32: 2.7e-3);
      * a-6: a seed's 40-neighbour set (FNV hash of the sorted set) equals the reference's unless the reference recorded that seed's
        top-k boundary gap below 5e-3 (measured: 2.5e-3 at worst, KITTI scale; exact fp32: 8.7e-4);
      * a-10: on seeds whose neighbour set is equal, the vote count equals the reference's or differs by one (a correspondence on
        the inlier threshold of a hypothesis computed in another summation order); the chosen hypothesis and the refinement's
        inlier-count sequence are the reference's whenever every vote is;
    and the SHARE of decisions in each near-tie class stays below the family's measured bound (_STAGE_BOUNDS).  On pairs where the
    reference's own logits leave fewer than S positive keys -- about HALF of these pairs: with 5-10 % inliers fewer than S = N / 10
    correspondences are positive local maxima -- the tail of its seed list is torch.argsort's order of keys tied at zero
    (models/PointDSC.py:211-217, recognised from its recorded logits): there the a-5 rule covers the positive-key prefix, a-6 / a-10 the
    seeds both lists share."""
    if not (GOLDEN / f"census_internals_{name}.npz").exists():
        pytest.skip(f"tests/golden/census_internals_{name}.npz not generated")
    model, _ = _bench_model(name)
    mod = _census_module()
    ix = np.load(GOLDEN / f"census_internals_{name}.npz", allow_pickle=False)
    w = workloads.WORKLOADS[name]
    n, S = w["num_corr"], int(w["num_corr"] * w["model"]["ratio"])
    total = min(pairs, ix["seeds32"].shape[0])
    if arith == "exact_fp32":
        model.attention_precision, model.compat_format, model.layer_gemm = "fp32", "f32", "f32"
    try:
        _stage_census_body(name, step, arith, model, mod, ix, w, n, S, total)
    finally:
        model.layer_gemm, model.compat_format, model.attention_precision = LAYER_GEMM_DEFAULT, COMPAT_FORMAT_DEFAULT, "fp16x3"


def _stage_census_body(name, step, arith, model, mod, ix, w, n, S, total):
    zero_key = []
    st = {"pairs": 0, "seed_positions": 0, "seed_positions_differ": 0, "seed_sets_differ": 0, "knn_sets": 0, "knn_sets_differ": 0,
          "knn_max_gap_of_differing": 0.0, "votes": 0, "votes_differ_by_one": 0, "votes_differ_more": 0, "pairs_all_equal": 0,
          "best_differs": 0, "trace_differs": 0, "max_rel_logit_diff": 0.0}
    for first in range(0, total, step):
        g_ = min(step, total - first)
        batch = workloads.batch(name, first, g_)
        _forward(model, batch)
        dec = mod.decisions(model, g_, n)
        conf_here = model.workspace_view("conf", g_, n)[: g_ * n].reshape(g_, n).cpu().numpy()
        for b in range(g_):
            i = first + b
            conf = ix["conf32"][i]
            src = batch["src_keypts"][b].float()
            d = torch.cdist(src[None], src[None])[0].numpy()
            is_max = ((conf[:, None] >= conf[None, :]) | (d >= np.float32(w["model"]["nms_radius"]))).all(axis=1)
            # fewer than S positive keys: the reference's list is then its positive keys in descending order FOLLOWED by keys tied at zero
            # in torch.argsort's backend-defined order -- only the positive-key prefix is comparable (the seeds both lists share are
            # still held to the a-6 / a-10 rules below)
            delta = float(np.abs(conf_here[b] - conf).max())
            rng = float(conf.max() - conf.min())
            # (keys within 2 delta of zero may change sign between the two runs: they belong to the tail, not to the prefix)
            lim = S if int(((conf * is_max) > 0).sum()) >= S else int(((conf * is_max) > 2 * delta).sum())
            if lim < S:
                zero_key.append(i)
            st["pairs"] += 1
            st["max_rel_logit_diff"] = max(st["max_rel_logit_diff"], delta / rng)
            gs, rs = dec["seeds"][b], ix["seeds32"][i]
            pos = np.flatnonzero(gs[:lim] != rs[:lim])
            st["seed_positions"] += lim
            st["seed_positions_differ"] += len(pos)
            # a-5: every differing position holds a correspondence whose reference logit is within 2 delta of the reference's occupant
            for p_ in pos:
                st["max_swap_over_delta"] = max(st.get("max_swap_over_delta", 0.0), abs(float(conf[gs[p_]]) - float(conf[rs[p_]])) / max(delta, 1e-30))
            gs_c, rs_c = gs[:lim], rs[:lim]
            st["seed_sets_differ"] += len(set(gs_c.tolist()) ^ set(rs_c.tolist())) // 2
            # a-6 / a-10 on the seeds both lists hold, matched by correspondence
            rpos = {int(c): j for j, c in enumerate(rs)}
            hashes_here = mod.set_hash_rows(dec["knn"][b])
            all_equal = len(pos) == 0 and lim == S
            for j, c in enumerate(gs[:lim]):                  # (positive-key seeds: the tail of a short list is argsort's tie order)
                r = rpos.get(int(c))
                if r is None or r >= lim:
                    continue
                st["knn_sets"] += 1
                if int(hashes_here[j]) != int(ix["knn_hash32"][i][r]):
                    gap = float(ix["knn_gap32"][i][r])
                    st["knn_sets_differ"] += 1
                    st["knn_max_gap_of_differing"] = max(st["knn_max_gap_of_differing"], gap)
                    all_equal = False
                    continue
                st["votes"] += 1
                dv = abs(int(dec["counts"][b][j]) - int(ix["counts32"][i][r]))
                if dv == 1:
                    st["votes_differ_by_one"] += 1
                elif dv > 1:
                    st["votes_differ_more"] += 1
                all_equal = all_equal and dv == 0
            if all_equal:
                st["pairs_all_equal"] += 1
                if int(dec["best"][b]) != int(ix["best32"][i]):
                    st["best_differs"] += 1
                if not np.array_equal(dec["trace"][b][:21], ix["refine_counts32"][i]):
                    st["trace_differs"] += 1
    print(f"STAGE-CENSUS {name} x{step} {arith}: {json.dumps(st)}; pairs with fewer than S positive keys {zero_key}")
    bpos, bknn, bgap, bvote = _STAGE_BOUNDS[name]
    assert st["knn_max_gap_of_differing"] < bgap, st
    # a-5: two positions trade places only between correspondences whose REFERENCE logits are closer than twice this run's largest logit
    # deviation on that pair; the deviation itself stays below 2e-3 of the pair's logit range
    assert st.get("max_swap_over_delta", 0.0) <= 2.0 and st["max_rel_logit_diff"] <= 3e-2, st
    assert st["seed_positions_differ"] <= bpos * st["seed_positions"], st
    assert st["knn_sets_differ"] <= bknn * st["knn_sets"], st
    assert st["votes_differ_by_one"] <= bvote * st["votes"] and st["votes_differ_more"] == 0, st
    assert st["best_differs"] == 0, st
    st["pairs_with_fewer_than_S_positive_keys"] = len(zero_key)


@pytest.mark.parametrize("fmt", ["f32", "u16"])
@pytest.mark.parametrize("name,bs", [("n5000_b32", 4), ("kitti_n5000_b16", 3)])
def test_forward_is_bit_identical_with_row_order_and_point_fragment_hand_offs(name, bs, fmt, monkeypatch):
    """Whole forward, layer_gemm = "h3": partials / featB handed over in point-fragment order (default) vs plain rows
    (PDSC_LAYER_PF=0) vs the generic H3 kernel (PDSC_LAYER_H3_VARIANT=0): same arithmetic in the same order, so poses and
    labels agree bit for bit, with either storage format of the spatial-consistency matrix."""
    model, _ = _bench_model(name)
    batch = workloads.batch(name, 0, bs)
    # (the per-launch key split: the row-order hand-off exists in that form only -- the leaf form is point-fragment by construction)
    model.compat_format, model.layer_gemm, model.att_leaves = fmt, "h3", "per_launch"
    out = []
    try:
        envs = ({}, {"PDSC_LAYER_PF": "0"}, {"PDSC_LAYER_H3_VARIANT": "0"}) if _lib.load().pdsc_experiments_enabled() else ({}, {})
        for env in envs:     # (the knobs exist in experiments builds; the product library runs the default twice: repeatability)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            res = _forward(model, batch)
            out.append((res["final_trans"].cpu().clone(), res["final_labels"].cpu().clone()))
            for k in env:
                monkeypatch.delenv(k)
    finally:
        model.compat_format, model.layer_gemm, model.att_leaves = COMPAT_FORMAT_DEFAULT, LAYER_GEMM_DEFAULT, LEAVES_DEFAULT
    for T, L in out[1:]:
        assert torch.equal(T, out[0][0]) and torch.equal(L, out[0][1])
    for i in range(bs):
        re, te = O.registration_errors(out[0][0][i], batch["gt_trans"][i])
        assert re < 1.0 and te < (60.0 if "kitti" in name else 5.0), (i, re, te)


def test_bench_timed_path_uses_the_wave_layer_kernel_and_fused_merge():
    """Guards the claim above: at the headline configuration the forward goes through layer_wave_kernel and merges the
    key-split partials inside it (csrc/api.hip:run_forward), i.e. the golden test at bs=32 covers those kernels."""
    lib = _lib.load()
    assert lib.pdsc_layer_prefers_block(32, 5000) == 0 and lib.pdsc_layer_prefers_block(1, 5000) == 1
    ns = lib.pdsc_attention_split_default_split(32, 5000)
    assert 1 < ns <= 4, ns


@pytest.mark.parametrize("n", [20000, 36864])
def test_large_n_properties_up_to_the_documented_limit(n):
    """multiway/test_multi_ate.py:245 feeds up to 20 000 correspondences; 36 864 is the documented maximum (kNN row in one
    workgroup's LDS).  No oracle at these sizes (the CPU path needs tens of GB): size-independent properties instead -- a
    rigid motion close to the ground truth, labels == the inliers of the pre-refinement hypothesis (recomputed on the host
    from the returned initial_trans), inlier mask ~ the ground-truth mask."""
    model, _ = _bench_model("n5000_b32")
    pair = synthetic.make_pair(n, seed=77, inlier_ratio=0.2)
    res = _forward(model, pair)
    T = res["final_trans"][0].cpu().double()
    assert (T[:3, :3] @ T[:3, :3].T - torch.eye(3, dtype=torch.float64)).abs().max() < 1e-5 and abs(float(torch.det(T[:3, :3])) - 1) < 1e-5
    re, te = O.registration_errors(res["final_trans"][0].cpu(), pair["gt_trans"][0])
    assert re < 1.0 and te < 5.0, (re, te)
    init = model.workspace_view("initial_trans", 1, n)[:16].reshape(1, 4, 4).cpu()
    L2 = O.residuals(init, pair["src_keypts"][0], pair["tgt_keypts"][0])[0]
    lab = res["final_labels"][0].cpu()
    assert torch.equal(lab, (L2 < KW["inlier_threshold"]).float())
    tp = float((lab * pair["gt_labels"][0]).sum())
    assert tp / float(pair["gt_labels"][0].sum()) > 0.95 and tp / float(lab.sum()) > 0.95


def test_n_above_the_limit_is_rejected_loudly():
    model, _ = _bench_model("n5000_b32")
    pair = synthetic.make_pair(36865, seed=1, inlier_ratio=0.2)
    with pytest.raises(RuntimeError):
        _forward(model, pair)


def test_forward_is_bitwise_repeatable():
    """Same inputs, same process: poses, labels, features and every seed hypothesis come out bit for bit the same (no
    order-dependent float atomics anywhere on the path), in both compat formats."""
    model, _ = _bench_model("n5000_b32")
    batch = workloads.batch("n5000_b32", 0, 8)
    try:
        for fmt in ("f32", "u16"):
            model.compat_format = fmt
            runs = []
            for _ in range(3):
                res = _forward(model, batch)
                runs.append((res["final_trans"].clone(), res["final_labels"].clone(),
                             model.workspace_view("featA", 8, 5000)[: 8 * 5000 * 128].clone(),
                             model.workspace_view("seed_trans", 8, 5000)[: 8 * 500 * 16].clone()))
            for r in runs[1:]:
                assert all(torch.equal(x, y) for x, y in zip(runs[0], r)), fmt
    finally:
        model.compat_format = COMPAT_FORMAT_DEFAULT


@pytest.mark.parametrize("bs,n,s,k,dup", [(2, 1000, 100, 40, 0), (32, 5000, 500, 40, 0), (3, 2053, 205, 40, 0), (1, 5000, 500, 40, 0),
                                        (4, 700, 70, 47, 0), (2, 1500, 150, 40, 600), (1, 300, 33, 10, 0), (8, 10000, 1000, 40, 0)])
def test_knn_fused_form_equals_the_matrix_form(bs, n, s, k, dup):
    """r05: pdsc_knn_seeds_form 2 computes the seeds' Gram rows on the fp32 matrix cores and selects the k
    neighbours in the same launch -- the S x N distance matrix (320 MB at 32 pairs of N = 5000) is never written.  Same MFMA order
    -> same distance bits; same (distance, index) composite order -> the neighbour indices equal the two-launch form's exactly,
    including rows with hundreds of EQUAL distances (duplicated feature rows: ties resolve by ascending index in both)."""
    rs = np.random.RandomState(700 + n + bs)
    x = rs.standard_normal((bs, n, 128)).astype(np.float32)
    if dup:
        x[:, dup:2 * dup] = x[:, :dup]                      # every distance to a duplicated row appears twice
        x[:, 2 * dup:2 * dup + 200] = x[:, :1]              # ... and 200 copies of row 0: 200-way ties
    x /= np.linalg.norm(x, axis=-1, keepdims=True)
    normed = g(torch.from_numpy(x))
    seeds = g(torch.from_numpy(np.stack([rs.permutation(n)[:s] for _ in range(bs)]).astype(np.int32)))
    want = ops.knn_seeds(normed, seeds, k, form="matrix")
    got = ops.knn_seeds(normed, seeds, k, form="fused")
    torch.cuda.synchronize()
    assert torch.equal(got, want), (int((got != want).sum()), got.shape)
    # ... and with the column operand in point-fragment order, as the forward feeds it (pdsc_normalize_confidence_pf: the same
    # normalised rows -- here re-normalising unit rows changes nothing beyond the last bit, so the distances are re-derived from ITS rows)
    h2 = g(torch.zeros(bs, n, 32))
    rows, rows_pf, conf = ops.normalize_confidence_pf(normed, h2, g(torch.zeros(32)), g(torch.zeros(1)))
    rows2, conf2 = ops.normalize_confidence(normed.reshape(bs * n, 128), h2.reshape(bs * n, 32), g(torch.zeros(32)), g(torch.zeros(1)))
    assert torch.equal(rows.reshape(bs * n, 128), rows2) and torch.equal(conf.reshape(-1), conf2)
    tiles = (n + 31) // 32
    img = rows_pf.reshape(bs, tiles, 16, 2, 32, 4)                     # [pair][tile][q][h][l31][e] -> row l31, channel 8q + 4h + e
    back = img.permute(0, 1, 4, 2, 3, 5).reshape(bs, tiles * 32, 128)[:, :n]
    assert torch.equal(back, rows)
    want_pf = ops.knn_seeds(rows, seeds, k, form="matrix")
    got_pf = ops.knn_seeds(rows, seeds, k, form="fused", normed_pf=rows_pf)
    assert torch.equal(got_pf, want_pf), int((got_pf != want_pf).sum())
    auto = ops.knn_seeds(normed, seeds, k)
    assert torch.equal(auto, want)
    for _ in range(3):                                       # repeated launches: nothing carried over between them
        assert torch.equal(ops.knn_seeds(normed, seeds, k, form="fused"), want)


def test_knn_fused_form_rejections():
    normed = g(torch.nn.functional.normalize(torch.randn(1, 200, 128), dim=-1))
    seeds = g(torch.arange(20, dtype=torch.int32)[None])
    with pytest.raises(RuntimeError, match="fused form"):
        ops.knn_seeds(normed, seeds, 40, form="fused")                   # N < 256
    normed = g(torch.nn.functional.normalize(torch.randn(1, 600, 128), dim=-1))
    with pytest.raises(RuntimeError, match="fused form"):
        ops.knn_seeds(normed, seeds, 60, form="fused")                   # k + 1 > 48
    assert ops.knn_seeds(normed, seeds, 60).shape == (1, 20, 60)         # the library's choice falls back


LEAVES_DEFAULT = PointDSC().att_leaves


@pytest.mark.parametrize("name,sizes", [("n1000_b1", (32, 16, 8, 4, 2, 1)), ("n5000_b32", (32, 16, 8, 4, 2, 1)),
                                        ("kitti_n5000_b16", (16, 8, 4, 2, 1)), ("lomatch_n10000_b8", (8, 4, 2, 1)),
                                        ("kitti_n12000_b4", (4, 2, 1)), ("multiway_n20000_b1", (2, 1))])
def test_canonical_leaves_make_a_pair_independent_of_its_batch(name, sizes):
    """r05: with att_leaves = "canonical" the attention sums a query's keys over a leaf
    structure that depends on N alone -- the launch plan only decides which workgroup computes a leaf -- so the whole forward
    returns BITWISE the same pose and mask for a pair whether it runs alone or with 1 .. 31 others: 32 pairs on one GPU and 4 pairs on
    each of 8 GPUs are the same numbers (reference semantics: one pair per call, models/PointDSC.py:210,414)."""
    model, _ = _bench_model(name)
    big = sizes[0]
    batch = workloads.batch(name, 0, big)
    try:
        model.att_leaves = "canonical"
        full = _forward(model, batch)
        T, L = full["final_trans"].clone(), full["final_labels"].clone()
        for bs in sizes[1:]:
            for first in range(0, big, bs):
                part = _forward(model, {k: batch[k][first:first + bs] for k in batch})
                assert torch.equal(part["final_trans"].view(torch.int32), T[first:first + bs].view(torch.int32)), (name, bs, first)
                assert torch.equal(part["final_labels"], L[first:first + bs]), (name, bs, first)
    finally:
        model.att_leaves = LEAVES_DEFAULT


@pytest.mark.parametrize("n,bs_list", [(1000, (1, 3, 8)), (2053, (1, 2, 5)), (700, (1, 4))])
def test_canonical_leaves_features_independent_of_the_batch(n, bs_list):
    """The same property one level down: the validation forward's feature-similarity matrix M (N x N, a function of every
    feature channel of every correspondence, models/PointDSC.py:158-163) and the logits, bit for bit across batch sizes."""
    c = case(n)
    model = c["model"]
    big = max(bs_list)
    batch = synthetic.make_batch(big, n, seed=520 + n, inlier_ratio=0.3)
    outs = {}
    try:
        model.att_leaves = "canonical"
        for bs in bs_list:
            data = {k: g(batch[k][:bs]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
            with torch.no_grad():
                outs[bs] = model(data)
            torch.cuda.synchronize()
    finally:
        model.att_leaves = LEAVES_DEFAULT
    ref = outs[big]
    for bs in bs_list:
        assert torch.equal(outs[bs]["M"].view(torch.int32), ref["M"][:bs].view(torch.int32)), (n, bs)
        assert torch.equal(outs[bs]["final_labels"].view(torch.int32), ref["final_labels"][:bs].view(torch.int32)), (n, bs)


@pytest.mark.parametrize("name,bs", [("n1000_b1", 1), ("n1000_b1", 16), ("n5000_b32", 2), ("n5000_b32", 4), ("n5000_b32", 16), ("n5000_b32", 32),
                                     ("kitti_n5000_b16", 16), ("lomatch_n10000_b8", 1), ("lomatch_n10000_b8", 8)])
def test_leaf_form_with_as_many_leaves_as_key_splits_reproduces_the_per_launch_bits(name, bs):
    """The leaf form (a workgroup streams through several leaves, each from a fresh online-softmax state, and leaves one partial per
    leaf) against the key-split form (one partial per workgroup): with att_leaves forced to the per-launch plan's key split the leaves
    ARE that plan's key ranges, whichever divisor of the leaf count the leaf plan gives to a workgroup -- so the two forwards agree bit
    for bit.  Pins the in-loop leaf boundary (raw logits, fresh row maximum, accumulators re-zeroed in place, deferred partial store)
    against the code path that has been parity-clean since r01."""
    lib = _lib.load()
    model, _ = _bench_model(name)
    n = workloads.WORKLOADS[name]["num_corr"]
    ns = int(lib.pdsc_attention_split_default_split(bs, n))
    if not 2 <= ns <= 8:
        pytest.skip(f"per-launch key split {ns} outside the leaf form's 2..8")
    batch = workloads.batch(name, 0, bs)
    try:
        model.att_leaves = "per_launch"
        want = _forward(model, batch)
        model.att_leaves = ns
        for rep in range(2):
            got = _forward(model, batch)
            assert torch.equal(got["final_trans"].view(torch.int32), want["final_trans"].view(torch.int32)), (name, bs, ns, rep)
            assert torch.equal(got["final_labels"], want["final_labels"]), (name, bs, ns, rep)
    finally:
        model.att_leaves = LEAVES_DEFAULT


def test_leaf_form_validation_matrix_reproduces_the_per_launch_bits():
    """Same property on the validation forward's N x N feature-similarity matrix (every feature of every correspondence)."""
    lib = _lib.load()
    c = case(1000)
    model = c["model"]
    batch = synthetic.make_batch(3, 1000, seed=610, inlier_ratio=0.3)
    data = {k: g(batch[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    ns = int(lib.pdsc_attention_split_default_split(3, 1000))
    assert 2 <= ns <= 8
    try:
        outs = {}
        for mode in ("per_launch", ns):
            model.att_leaves = mode
            with torch.no_grad():
                outs[mode] = model(data)
            torch.cuda.synchronize()
        assert torch.equal(outs["per_launch"]["M"].view(torch.int32), outs[ns]["M"].view(torch.int32))
        assert torch.equal(outs["per_launch"]["final_labels"].view(torch.int32), outs[ns]["final_labels"].view(torch.int32))
    finally:
        model.att_leaves = LEAVES_DEFAULT


@pytest.mark.parametrize("name,bs,reps", [("n1000_b1", 1, 300), ("n5000_b32", 4, 40), ("n5000_b32", 32, 12), ("lomatch_n10000_b8", 2, 20)])
def test_leaf_form_attention_is_deterministic_over_repeated_launches(name, bs, reps):
    """Hundreds of launches of the canonical-leaf forward return one result."""
    model, _ = _bench_model(name)
    batch = workloads.batch(name, 0, bs)
    data = {k: g(batch[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    data["testing"] = True
    try:
        model.att_leaves = "canonical"
        with torch.no_grad():
            first = model(data)
            T, L = first["final_trans"].clone(), first["final_labels"].clone()
            for rep in range(reps):
                res = model(data)
                assert torch.equal(res["final_trans"].view(torch.int32), T.view(torch.int32)), rep
                assert torch.equal(res["final_labels"], L), rep
        torch.cuda.synchronize()
    finally:
        model.att_leaves = LEAVES_DEFAULT


def test_batched_forward_equals_per_pair_calls():
    c = case(1000)
    batch = synthetic.make_batch(3, 1000, seed=300, inlier_ratio=0.3)
    res = _forward(c["model"], batch)
    for i in range(3):
        one = _forward(c["model"], {k: batch[k][i:i + 1] for k in batch})
        assert torch.equal(res["final_labels"][i], one["final_labels"][0])
        assert (res["final_trans"][i] - one["final_trans"][0]).abs().max() < 2e-5


@pytest.mark.parametrize("n", [11, 30, 64, 129])
def test_small_and_ragged_sizes(n):
    """N < k+1 (k = N-1, reference :250), S = int(N*0.1) as small as 1, tiles with ragged tails."""
    c = case(n, pair_seed=31, inlier_ratio=0.6)
    res = _forward(c["model"], c["pair"])
    flips = int((res["final_labels"].cpu() != c["res"]["final_labels"]).sum())
    assert flips == 0 and (res["final_trans"].cpu() - c["res"]["final_trans"]).abs().max() < 1e-4


def test_kitti_thresholds_select_the_other_refinement_schedule():
    c = case(1500, pair_seed=4, wseed=4, kw=dict(inlier_threshold=0.6, sigma_d=1.2, nms_radius=0.6))
    pair = synthetic.make_pair(1500, seed=4, inlier_ratio=0.3, scale=60.0, noise=0.1)
    sd = c["sd"]
    want = O.forward_testing(sd, pair["corr_pos"], pair["src_keypts"], pair["tgt_keypts"],
                             **{k: c["kw"][k] for k in ORACLE_KEYS})
    res = _forward(c["model"], pair)
    assert c["model"]._config().refine_threshold == pytest.approx(1.2)
    assert torch.equal(res["final_labels"].cpu(), want["final_labels"])
    assert (res["final_trans"].cpu() - want["final_trans"]).abs().max() < 1e-4


def test_checkpoint_sigma_overrides_constructor():
    model = PointDSC(**dict(KW, sigma_d=0.5))
    sd = synthetic.make_state_dict(model.state_dict(), seed=6)
    sd["sigma_spat"] = torch.tensor([0.1])                # the snapshot's value wins (reference note 3)
    model.load_state_dict(sd)
    model = model.eval().to(DEV)
    c = case(1000)
    res = _forward(model, c["pair"])
    assert torch.equal(res["final_labels"].cpu(), c["res"]["final_labels"])


def test_inputs_are_not_mutated_and_errors_are_loud():
    c = case(257)
    data = {k: g(c["pair"][k]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    keep = {k: v.clone() for k, v in data.items()}
    data["testing"] = True
    c["model"](data)
    torch.cuda.synchronize()
    assert all(torch.equal(data[k], keep[k]) for k in keep)
    with pytest.raises(RuntimeError, match="unsupported problem size"):
        c["model"]({"corr_pos": g(torch.zeros(1, 5, 6)), "src_keypts": g(torch.zeros(1, 5, 3)),
                    "tgt_keypts": g(torch.zeros(1, 5, 3)), "testing": True})
    with pytest.raises(RuntimeError, match="K="):
        ops.linear(g(torch.zeros(4, 12)), g(torch.zeros(4, 12)))


# ------------------------------------------------------------------------------------------------------
# validation forward (no 'testing' key, eval() mode): SURVEY.md section 8 f-1
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,bs", [(257, 1), (1000, 2), (65, 3)])
def test_feature_compat_matches_oracle(n, bs):
    gen = torch.Generator().manual_seed(n)
    f = torch.randn(bs, n, 128, generator=gen)
    f[:, 1] = f[:, 0]                                         # an exact duplicate: similarity 1 -> clamps at 1
    normed = O.l2_normalize(f.reshape(-1, 128)).reshape(bs, n, 128)
    for sigma in (1.0, 0.37):
        got = ops.feature_compat(g(normed.reshape(-1, 128)), g(torch.tensor([sigma])), bs, n).cpu()
        for b in range(bs):
            want = O.feature_compat(normed[b], torch.tensor([sigma]))
            assert (got[b] - want).abs().max() < 2e-6 / sigma ** 2      # fp32 dot-product order, amplified by 1/sigma^2
            assert bool((torch.diagonal(got[b]) == 0).all()) and float(got[b].min()) >= 0 and float(got[b].max()) <= 1
            assert torch.equal(got[b], got[b].T)                        # symmetric bit for bit


@pytest.mark.parametrize("precision", ["fp16x3", "fp32"])
@pytest.mark.parametrize("name", ["val_n257_b1", "val_n1000_b3", "val_n2053_b2"])
def test_validation_forward_matches_reference_golden(name, precision):
    fx = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    kw = json.loads(str(fx["model_json"]))
    model = PointDSC(**kw)
    model.load_state_dict(synthetic.make_state_dict(model.state_dict(), seed=int(fx["wseed"])))
    model = model.eval().to(DEV)
    model.attention_precision = precision
    data = {k: g(torch.from_numpy(fx[k])) for k in ("corr_pos", "src_keypts", "tgt_keypts")}      # no 'testing' key
    with torch.no_grad():
        res = model(data)
    torch.cuda.synchronize()
    bs, n = fx["corr_pos"].shape[:2]
    assert res["M"].shape == (bs, n, n) and res["final_labels"].shape == (bs, n) and res["final_trans"].shape == (bs, 4, 4)
    scale = max(1.0, float(np.abs(fx["ref_logits"]).max()))
    assert (res["final_labels"].cpu() - torch.from_numpy(fx["ref_logits"])).abs().max() < 3e-5 * scale      # logits
    assert (res["final_trans"].cpu() - torch.from_numpy(fx["ref_final_trans"])).abs().max() < 1e-4          # R/t
    rows = torch.from_numpy(fx["M_rows"])
    M = res["M"].cpu()
    assert (M[:, rows] - torch.from_numpy(fx["ref_M_rows"])).abs().max() < 3e-5                             # features enter M directly
    assert bool((torch.diagonal(M, dim1=1, dim2=2) == 0).all()) and float(M.min()) >= 0 and float(M.max()) <= 1
    if "ref_M" in fx.files:
        assert (M - torch.from_numpy(fx["ref_M"])).abs().max() < 3e-5


def test_validation_forward_needs_eval_mode():
    c = case(257)
    data = {k: g(c["pair"][k]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    c["model"].train()
    with pytest.raises(RuntimeError, match="eval"):
        c["model"](data)
    c["model"].eval()
    res = c["model"](data)
    assert res["M"] is not None and not torch.equal(res["final_labels"], res["final_labels"].round())       # logits, not 0/1


# ------------------------------------------------------------------------------------------------------
# correspondence construction in front of the path: SURVEY.md section 8 f-2
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["corr_n300_d32", "corr_n1000_d33_mutual", "corr_n5000_d32_mutual"])
def test_build_correspondences_matches_reference_golden(name):
    from oracle import correspondence_oracle as CO
    from pointdsc_amd import correspondences
    fx = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    src, tgt, skp, tkp = CO.make_descriptors(int(fx["ns"]), int(fx["nt"]), int(fx["d"]), int(fx["seed"]))
    idx, dist = correspondences.match_descriptors(g(torch.from_numpy(src)), g(torch.from_numpy(tgt)), want_dist=True)
    assert np.array_equal(idx.cpu().numpy(), fx["ref_source_idx"])                      # index work: exact
    want = CO.nn_distance_matrix(src, tgt).min(axis=1)
    assert np.abs(dist.cpu().numpy() - want).max() < 2e-6
    res = correspondences.build_correspondences(g(torch.from_numpy(src)), g(torch.from_numpy(tgt)), g(torch.from_numpy(skp)),
                                                g(torch.from_numpy(tkp)), use_mutual=bool(fx["mutual"]))
    assert np.array_equal(res["corr"].cpu().numpy(), fx["ref_corr"])
    assert res["corr_pos"].shape == (1, fx["ref_corr"].shape[0], 6)
    # the kernel's column means are correctly rounded (fp64 sums); numpy's float32 mean of 3000 rows is good to ~4e-6
    assert np.abs(res["corr_pos"][0].cpu().numpy() - fx["ref_corr_pos"]).max() < 1e-5
    assert np.array_equal(res["src_keypts"][0].cpu().numpy(), skp[fx["ref_corr"][:, 0]])
    assert np.array_equal(res["tgt_keypts"][0].cpu().numpy(), tkp[fx["ref_corr"][:, 1]])


@pytest.mark.parametrize("name", ["corr_lomatch_n1000_d32", "corr_lomatch_n5000_d32"])
def test_inner_product_matching_matches_the_3dlomatch_callers_lines(name):
    """metric="ip": argmax of the inner products, the form of evaluation/test_3DLoMatch.py:45-48 (fixture: those four reference
    lines executed on seeded NON-unit descriptors, oracle/check_lomatch_matching_against_reference.py -- on most rows the
    arg-min of sqrt(2 - 2<s,t> + 1e-6) picks another target there).  Index work: exact; corr_pos to the rounding of the mean."""
    from oracle.check_lomatch_matching_against_reference import make_inputs
    from pointdsc_amd import correspondences
    fx = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    src, tgt, skp, tkp = make_inputs(dict(ns=int(fx["ns"]), nt=int(fx["nt"]), d=int(fx["d"]), seed=int(fx["seed"])))
    idx, dot = correspondences.match_descriptors(g(torch.from_numpy(src)), g(torch.from_numpy(tgt)), want_dist=True, metric="ip")
    assert np.array_equal(idx.cpu().numpy(), fx["ref_source_idx"])
    assert np.abs(dot.cpu().numpy() - fx["ref_best_dot"]).max() < 4e-6
    l2 = correspondences.match_descriptors(g(torch.from_numpy(src)), g(torch.from_numpy(tgt))).cpu().numpy()
    assert (l2 != fx["ref_source_idx"]).sum() > 100           # the two callers' forms really differ on this input
    res = correspondences.build_correspondences(g(torch.from_numpy(src)), g(torch.from_numpy(tgt)), g(torch.from_numpy(skp)),
                                                g(torch.from_numpy(tkp)), metric="ip")
    assert np.array_equal(res["corr"].cpu().numpy()[:, 1], fx["ref_source_idx"])
    assert np.abs(res["corr_pos"][0].cpu().numpy() - fx["ref_corr_pos"]).max() < 1e-5
    assert np.array_equal(res["tgt_keypts"][0].cpu().numpy(), tkp[fx["ref_source_idx"]])
    # torch.argmax semantics on special values: first index among equal maxima, NaN counts as the maximum
    s = torch.zeros(3, 8); t = torch.zeros(5, 8)
    s[0, 0] = 1.0; s[1, 1] = -1.0; s[2, 2] = 1.0
    t[1, 0] = 2.0; t[3, 0] = 2.0; t[4, 1] = 1.0                    # row 0: tie between targets 1 and 3 -> 1; row 1: all <= 0; row 2: all zero -> 0
    for with_nan in (False, True):
        if with_nan:
            t[2, 2] = float("nan")                                 # 0 * NaN = NaN: target 2 is NaN for every row -> index 2 everywhere
        want = torch.argmax(torch.einsum("ac,bc->ab", s, t), dim=-1).numpy()
        got = correspondences.match_descriptors(g(s), g(t), metric="ip").cpu().numpy()
        assert np.array_equal(got, want), (with_nan, got, want)


@pytest.mark.parametrize("m", [1, 31, 128, 129, 1000, 5000 * 3 + 17, 160000])
def test_classifier_hidden_equals_two_linears(m):
    """pdsc_classifier_hidden (r04: classification.0 .. classification.3, models/PointDSC.py:107-111, in one launch with the hidden
    layer kept in registers) against the two pdsc_linear launches it replaces in the forward: the same MFMA instruction, the same k
    order, operand roles swapped (a*b = b*a) -- bit-identical, whatever the row count (tiles of 128, persistent workgroups)."""
    gen = torch.Generator().manual_seed(m)
    feat = g(torch.randn(m, 128, generator=gen) * 0.7)
    w1, b1 = g(torch.randn(32, 128, generator=gen) * 0.15), g(torch.randn(32, generator=gen) * 0.1)
    w2, b2 = g(torch.randn(32, 32, generator=gen) * 0.3), g(torch.randn(32, generator=gen) * 0.1)
    want = ops.linear(ops.linear(feat, w1, b1, relu=True), w2, b2, relu=True)
    got = ops.classifier_hidden(feat, w1, b1, w2, b2)
    assert got.shape == (m, 32) and torch.equal(got, want)
    assert float((want > 0).float().mean()) > 0.2           # (the comparison is not about all-zero rows)


def test_match_descriptors_ties_and_ragged_sizes():
    """Equal distances -> the first index (np.argmin); sizes that are not multiples of any tile; D not a multiple of 8."""
    from oracle import correspondence_oracle as CO
    from pointdsc_amd import correspondences
    src, tgt, _, _ = CO.make_descriptors(131, 77, 33, seed=9)
    tgt[40] = tgt[3]                          # exact duplicates: both at the same distance from everybody
    tgt[76] = tgt[3]
    src[5] = tgt[3]
    idx = correspondences.match_descriptors(g(torch.from_numpy(src)), g(torch.from_numpy(tgt))).cpu().numpy()
    want = np.argmin(CO.nn_distance_matrix(src, tgt), axis=1)
    assert np.array_equal(idx, want) and idx[5] == 3
    for ns, nt, d in ((1, 1, 8), (5, 700, 32), (700, 5, 64), (257, 129, 1)):
        s, t, _, _ = CO.make_descriptors(ns, nt, d, seed=ns + nt)
        got = correspondences.match_descriptors(g(torch.from_numpy(s)), g(torch.from_numpy(t))).cpu().numpy()
        dm = CO.nn_distance_matrix(s, t)
        # compare through the distances (d = 1: many exact ties are legitimate, the index must still be the first minimum)
        assert np.array_equal(got, np.argmin(dm, axis=1)) or np.abs(dm[np.arange(ns), got] - dm.min(axis=1)).max() < 1e-6


def test_match_descriptors_near_ties_and_nan_distances_resolve_like_numpy():
    """np.argmin keeps the FIRST index among equal distances, and two different radicands can round to one distance; a NaN distance
    (negative radicand: |dot| > 1, or a NaN input) orders first -- the FIRST NaN.  Candidates one ulp apart, a pair of different
    radicands with the same rounded square root, chains of one-ulp improvements, inner products above 1 and a NaN descriptor, on
    inputs whose inner products are exact in any summation order (one non-zero column per source), so that numpy's distances
    are the kernel's.  (Written for an arg-min on radicands with an exact fallback, profiles/r05_p_match_ablation.txt -- measured, not
    shipped; the shipped per-element scan passes it as well.)"""
    from oracle import correspondence_oracle as CO
    from pointdsc_amd import correspondences
    rs = np.random.RandomState(5)
    ns, nt, d = 200, 900, 16
    src = np.zeros((ns, d), np.float32)
    src[np.arange(ns), np.arange(ns) % 5] = 1.0                      # source i = unit vector of column i % 5
    tgt = np.zeros((nt, d), np.float32)
    tgt[:, :4] = rs.uniform(-0.6, 0.6, size=(nt, 4)).astype(np.float32)
    tgt[:, 4] = rs.uniform(-0.6, 0.2, size=nt).astype(np.float32)
    one = np.float32(0.75)

    def ulps(x, k, towards=2.0):
        x = np.float32(x)
        for _ in range(k):
            x = np.nextafter(x, np.float32(towards))
        return x
    # column 0: a later target one ulp closer than an earlier one (neighbouring distances after rounding)
    tgt[100, 0], tgt[500, 0] = one, ulps(one, 1)
    # column 1: the closer one FIRST, then near misses behind it; column 2: a chain of improvements one ulp apart
    tgt[50, 1], tgt[60, 1], tgt[700, 1] = ulps(one, 2), one, ulps(one, 1)
    for k in range(6):
        tgt[120 * k + 7, 2] = ulps(one, k)
    # column 3: inner products above 1 (negative radicand -> NaN distance): np.argmin returns the first of them
    tgt[300, 3], tgt[200, 3], tgt[850, 3] = 1.5, 1.25, 3.0
    # column 4: two DIFFERENT radicands (1.5000011 first, 1.500001 later) that round to the SAME distance 1.2247453: the first index
    # wins in np.argmin, the smaller radicand would win a scan on radicands alone
    tgt[80, 4], tgt[640, 4] = ulps(0.25, 4, towards=-2.0), np.float32(0.25)
    with np.errstate(invalid="ignore"):
        dm = CO.nn_distance_matrix(src, tgt)
    assert dm[4, 80] == dm[4, 640] and tgt[80, 4] != tgt[640, 4]
    for case_nan in (False, True):
        t = tgt.copy()
        if case_nan:
            t[150, 0] = np.nan                                       # row 150: NaN distance for EVERY source (0 * NaN), before most best matches
        got = correspondences.match_descriptors(g(torch.from_numpy(src)), g(torch.from_numpy(t))).cpu().numpy()
        with np.errstate(invalid="ignore"):
            want = np.argmin(CO.nn_distance_matrix(src, t), axis=1)
        assert np.array_equal(got, want), (case_nan, np.flatnonzero(got != want)[:8], got[:10], want[:10])
        if case_nan:
            assert (want == 150).all()
        else:
            assert want[3] == 200 and want[4] == 80 and want[0] == 500 and want[1] == 50 and want[2] == 607, want[:5]


def test_correspondences_feed_the_forward():
    """descriptors -> build_correspondences -> PointDSC.forward: the registration of a synthetic pair is recovered."""
    from pointdsc_amd import correspondences
    c = case(1000)
    rs = np.random.RandomState(0)
    n = 1500
    pair = synthetic.make_pair(n, seed=77, inlier_ratio=1.0, noise=0.005)          # every source point has a true partner
    desc = rs.randn(n, 32).astype(np.float32)
    desc /= np.linalg.norm(desc, axis=1, keepdims=True)
    sdesc = desc + 0.25 * rs.randn(n, 32).astype(np.float32)                         # noisy copy: ~ some wrong matches
    sdesc /= np.linalg.norm(sdesc, axis=1, keepdims=True)
    perm = rs.permutation(n)                                                         # target cloud in another order
    data = correspondences.build_correspondences(g(torch.from_numpy(sdesc)), g(torch.from_numpy(desc[perm])),
                                                 pair["src_keypts"][0].to(DEV), pair["tgt_keypts"][0][perm].to(DEV))
    assert data["corr_pos"].shape == (1, n, 6)
    data["testing"] = True
    res = c["model"](data)
    re, te = O.registration_errors(res["final_trans"][0].cpu(), pair["gt_trans"][0])
    assert re < 1.0 and te < 5.0


# ------------------------------------------------------------------------------------------------------
# spectral-matching baseline (N x N power iteration): SURVEY.md section 8 f-3
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["sm_n257", "sm_n2000", "sm_n5000", "sm_kitti_n1500"])
def test_sm_baseline_matches_reference_golden(name):
    from oracle import sm_oracle
    from pointdsc_amd import baselines
    fx = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    pair = synthetic.make_pair(int(fx["N"]), seed=int(fx["seed"]), inlier_ratio=float(fx["inlier_ratio"]),
                               scale=float(fx["scale"]), noise=float(fx["noise"]))
    thr = float(fx["thr"])
    trans, labels, eig = baselines.SM(g(pair["corr_pos"]), g(pair["src_keypts"]), g(pair["tgt_keypts"]), thr, return_eig=True)
    n = int(fx["N"])
    assert trans.shape == (1, 4, 4) and labels.shape == (1, n) and int(labels.sum()) == int(n * 0.1)
    _, want_labels, want_eig = sm_oracle.sm_baseline(pair["corr_pos"][0], pair["src_keypts"][0], pair["tgt_keypts"][0], thr)
    eig = eig[0].cpu()
    assert (eig - want_eig).abs().max() < 2e-6 * float(want_eig.abs().max()) + 1e-9         # 10 mat-vecs in another order
    # the cut sits inside the inlier block, where neighbouring eigenvector entries can be closer than fp32 round-off:
    # labels must agree everywhere except within that margin of the cut value
    ref_labels = torch.from_numpy(fx["ref_pred_labels"][0])
    cut = float(torch.sort(want_eig, descending=True).values[int(n * 0.1) - 1])
    decided = (want_eig - cut).abs() > 4e-6 * float(want_eig.abs().max())
    assert torch.equal(labels[0].cpu()[decided], ref_labels[decided])
    flips = int((labels[0].cpu() != ref_labels).sum())
    assert flips <= 2 * int((~decided).sum())
    tol = 1e-4 if flips == 0 else 3e-3                                                      # another inlier subset moves the pose
    assert (trans[0].cpu() - torch.from_numpy(fx["ref_pred_trans"][0])).abs().max() < tol * max(1.0, float(fx["scale"]) / 3.0)


@pytest.mark.parametrize("n,bs,iters", [(5000, 1, 10), (5120, 1, 10), (4999, 2, 10), (2053, 3, 7), (1000, 1, 10), (300, 2, 1), (21, 1, 3)])
def test_sm_baseline_register_resident_form_equals_the_streaming_form(n, bs, iters):
    """The spectral-matching baseline has two forms (pdsc_sm_baseline_form): the N x N matrix in the chip's vector registers (N <= 5120:
    one persistent launch per pair, a grid barrier per power iteration), or written to HBM and streamed.  Same arithmetic in the
    same order: poses, labels and the eigenvector agree bit for bit -- full and ragged last column groups, the largest size that
    fits, several pairs, odd and even iteration counts (the y buffers alternate), a matrix smaller than one workgroup's rows."""
    from pointdsc_amd import baselines
    batch = synthetic.make_batch(bs, n, seed=40 + n, inlier_ratio=0.3)
    c, s_, t_ = g(batch["corr_pos"]), g(batch["src_keypts"]), g(batch["tgt_keypts"])
    a = baselines.SM(c, s_, t_, 0.10, num_iterations=iters, return_eig=True, form="resident")
    b = baselines.SM(c, s_, t_, 0.10, num_iterations=iters, return_eig=True, form="streaming")
    for x, y, what in zip(a, b, ("pred_trans", "pred_labels", "leading_eig")):
        assert torch.equal(x.view(torch.int32), y.view(torch.int32)), what
    for _ in range(20):                                   # repeated launches: the barrier counter starts from zero every time
        again = baselines.SM(c, s_, t_, 0.10, num_iterations=iters, return_eig=True, form="resident")
        assert all(torch.equal(x, y) for x, y in zip(again, a))
    auto = baselines.SM(c, s_, t_, 0.10, num_iterations=iters, return_eig=True)          # whichever form the library picks
    assert all(torch.equal(x, y) for x, y in zip(auto, a))
    # one size past the register file: the resident form refuses, the library's own choice is the streaming form
    if n == 5120:
        big = synthetic.make_batch(1, 5121, seed=41, inlier_ratio=0.3)
        args = (g(big["corr_pos"]), g(big["src_keypts"]), g(big["tgt_keypts"]), 0.10)
        with pytest.raises(RuntimeError, match="register-resident"):
            baselines.SM(*args, form="resident")
        assert all(torch.equal(p, q) for p, q in zip(baselines.SM(*args, return_eig=True), baselines.SM(*args, return_eig=True, form="streaming")))


@pytest.mark.timeout(120)
def test_sm_baseline_register_resident_launches_of_two_streams_do_not_share_the_chip():
    """Each register-resident launch needs nearly every compute unit for its grid barrier; two of them dispatched half each would
    wait for each other forever.  The library chains them through an event whatever stream they are enqueued on: 30 rounds of two
    streams launching at the same time complete, each with the streaming form's bits."""
    from pointdsc_amd import baselines
    n = 5000
    a = synthetic.make_batch(1, n, seed=61, inlier_ratio=0.3)
    b = synthetic.make_batch(1, n, seed=62, inlier_ratio=0.3)
    A = tuple(g(a[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts"))
    B = tuple(g(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts"))
    want_a = baselines.SM(*A, 0.10, return_eig=True, form="streaming")
    want_b = baselines.SM(*B, 0.10, return_eig=True, form="streaming")
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(30):
        with torch.cuda.stream(s1):
            ra = baselines.SM(*A, 0.10, return_eig=True, form="resident")
        with torch.cuda.stream(s2):
            rb = baselines.SM(*B, 0.10, return_eig=True, form="resident")
        outs.append((ra, rb))
    s1.synchronize()
    s2.synchronize()
    for ra, rb in outs:
        assert all(torch.equal(x, y) for x, y in zip(ra, want_a)) and all(torch.equal(x, y) for x, y in zip(rb, want_b))


def test_sm_baseline_register_resident_form_is_opt_in_and_refused_under_capture():
    """r05 (spectral.hip): the library's own choice (`form` unset) is the streaming form at every N -- the resident form
    needs the chip to itself, which only the caller can promise -- and a resident launch inside a stream capture is an error (a
    replayed graph would bypass the per-device ordering of resident launches), raised before anything is enqueued."""
    from pointdsc_amd import baselines
    batch = synthetic.make_batch(1, 5000, seed=77, inlier_ratio=0.3)
    args = tuple(g(batch[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts"))
    want = baselines.SM(*args, 0.10, return_eig=True, form="streaming")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(graph, stream=st):
            with pytest.raises(RuntimeError, match="cannot be captured"):
                baselines.SM(*args, 0.10, return_eig=True, form="resident")
            got = baselines.SM(*args, 0.10, return_eig=True)          # the library's choice captures fine
    graph.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(got, want))
    # and outside a capture the opt-in form still returns the streaming form's bits
    res = baselines.SM(*args, 0.10, return_eig=True, form="resident")
    assert all(torch.equal(x, y) for x, y in zip(res, want))


def test_cal_confidence_matches_reference_golden():
    """pdsc_cal_confidence vs the reference's own PointDSC.cal_confidence (models/PointDSC.py:366-401) on seeded pairs:
    M rebuilt bit-exactly by pdsc_spatial_compat, leading eigenvector = the reference's; three methods; batch of 2."""
    from pointdsc_amd import baselines
    fx = np.load(GOLDEN / "confidence.npz", allow_pickle=False)
    for ci in range(int(fx["num_cases"])):
        n, seed, scale, sigma = int(fx[f"c{ci}_n"]), int(fx[f"c{ci}_seed"]), float(fx[f"c{ci}_scale"]), float(fx[f"c{ci}_sigma"])
        pair = synthetic.make_pair(n, seed=seed, inlier_ratio=0.3, scale=scale, noise=scale / 300.0)
        assert abs(float(pair["src_keypts"].double().sum()) - float(fx[f"c{ci}_src_checksum"])) < 1e-6
        M = ops.spatial_compat(g(pair["src_keypts"]), g(pair["tgt_keypts"]), g(torch.tensor([sigma], dtype=torch.float32)))
        v = g(torch.from_numpy(fx[f"c{ci}_leading_eig"]))
        want = fx[f"c{ci}_conf"]
        for k, method in enumerate(("eig_value", "eig_value_ratio", "xMx")):
            got = baselines.cal_confidence(M, v, method=method, num_iterations=10)
            assert got.shape == (1, 1)
            rel = abs(float(got) - float(want[k])) / abs(float(want[k]))
            assert rel < (1e-4 if method == "eig_value_ratio" else 2e-6), (n, method, float(got), float(want[k]))
        # natural [bs, N, N] layout (N not a multiple of 4 is padded by the wrapper) and a batch of two
        Mn = M[:, :, :n].contiguous()
        two = baselines.cal_confidence(torch.cat([Mn, Mn * 0.5]), torch.cat([v, v]), method="eig_value")
        assert abs(float(two[0]) - float(want[0])) / float(want[0]) < 2e-6 and abs(float(two[1]) - 0.5 * float(want[0])) / float(want[0]) < 2e-6
    with pytest.raises(ValueError):
        baselines.cal_confidence(M, v, method="nope")


def test_sm_baseline_batched_equals_per_pair():
    from pointdsc_amd import baselines
    batch = synthetic.make_batch(3, 700, seed=60, inlier_ratio=0.35)
    T, L = baselines.SM(g(batch["corr_pos"]), g(batch["src_keypts"]), g(batch["tgt_keypts"]), 0.10)
    for i in range(3):
        t1, l1 = baselines.SM(g(batch["corr_pos"][i:i + 1]), g(batch["src_keypts"][i:i + 1]), g(batch["tgt_keypts"][i:i + 1]), 0.10)
        assert torch.equal(L[i], l1[0]) and torch.equal(T[i], t1[0])
        re, te = O.registration_errors(T[i].cpu(), batch["gt_trans"][i])
        assert re < 1.0 and te < 5.0


# ------------------------------------------------------------------------------------------------------
# evaluation row on the device: SURVEY.md section 8 f-4
# ------------------------------------------------------------------------------------------------------
def test_eval_stats_match_reference_losses():
    """24 seeded (pose, labels) cases through the reference's TransformationLoss / ClassificationLoss (fixture written by
    oracle/check_metrics_against_reference.py) vs pdsc_eval_stats."""
    fx = np.load(GOLDEN / "metrics.npz", allow_pickle=False)
    stats = ops.eval_stats(g(torch.from_numpy(fx["trans"])), g(torch.from_numpy(fx["gt_trans"])),
                           g(torch.from_numpy(fx["pred_labels"])), g(torch.from_numpy(fx["gt_labels"]))).cpu().double().numpy()
    ref = fx["ref_stats"]
    assert np.array_equal(stats[:, 0], ref[:, 0])                                  # success flags
    assert np.array_equal(stats[:, 3], ref[:, 3]) and np.array_equal(stats[:, 5], ref[:, 5])      # counts: exact
    assert np.abs(stats[:, 1] - ref[:, 1]).max() < 2e-2                            # RE [deg]: acos is ill-conditioned near 0
    assert np.abs(stats[:, 2] - ref[:, 2]).max() < 1e-3                            # TE [cm]
    assert np.abs(stats[:, [4, 6, 7, 8]] - ref[:, [4, 6, 7, 8]]).max() < 1e-6      # ratios


# ------------------------------------------------------------------------------------------------------
# BASELINE.json size (N=5000, 4 pairs per GPU): size-independent properties
# ------------------------------------------------------------------------------------------------------
def test_full_size_batch_properties():
    n, bs = 5000, 4
    c = case(1000)
    model = c["model"]
    batch = synthetic.make_batch(bs, n, seed=1000, inlier_ratio=0.2)
    res = _forward(model, batch)
    T = res["final_trans"].cpu().double()
    R = T[:, :3, :3]
    assert (R @ R.transpose(1, 2) - torch.eye(3, dtype=torch.float64)).abs().max() < 1e-5
    assert (torch.det(R) - 1).abs().max() < 1e-5 and torch.equal(T[:, 3], torch.tensor([[0, 0, 0, 1.0]] * bs, dtype=torch.float64))
    lab = res["final_labels"].cpu()
    assert set(lab.unique().tolist()) <= {0.0, 1.0}
    init = model.workspace_view("initial_trans", bs, n)[: bs * 16].reshape(bs, 4, 4).cpu()
    for i in range(bs):
        # labels are exactly the inliers of the pre-refinement best hypothesis (reference note 1)
        L2 = O.residuals(init[i:i + 1], batch["src_keypts"][i], batch["tgt_keypts"][i])[0]
        assert torch.equal(lab[i], (L2 < KW["inlier_threshold"]).float())
        re, te = O.registration_errors(res["final_trans"][i].cpu(), batch["gt_trans"][i])
        assert re < 1.0 and te < 5.0
        tp = float((lab[i] * batch["gt_labels"][i]).sum())
        assert tp / float(batch["gt_labels"][i].sum()) > 0.95 and tp / float(lab[i].sum()) > 0.95
    # permutation equivariance: shuffling the correspondences permutes the labels and keeps the pose
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1))
    shuf = {k: batch[k][:1, perm] for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    res2 = _forward(model, shuf)
    assert (res2["final_trans"][0].cpu() - res["final_trans"][0].cpu()).abs().max() < 2e-3
    assert float((res2["final_labels"][0].cpu() != lab[0][perm]).float().mean()) < 0.01


# ------------------------------------------------------------------------------------------------------
# ragged batches (real evaluation pairs differ in N; reference evaluation/test_3DMatch.py:126,
# datasets/dataloader.py:6-31) -- pdsc_forward_testing_ragged
# ------------------------------------------------------------------------------------------------------
def _ragged_pairs(sizes, seed0, **kw):
    return [synthetic.make_pair(n, seed=seed0 + i, **kw) for i, n in enumerate(sizes)]


def _as_lists(pairs):
    return {"corr_pos": [g(p["corr_pos"][0]) for p in pairs], "src_keypts": [g(p["src_keypts"][0]) for p in pairs],
            "tgt_keypts": [g(p["tgt_keypts"][0]) for p in pairs], "testing": True}


@pytest.mark.parametrize("sizes", [(3000, 4100, 5000, 5333), (5000, 4999, 4097, 4096, 4095, 3333, 2600, 5001),
                                   (1000, 777, 640, 999), (257, 300, 1000, 2053, 5000)])
def test_ragged_batch_equals_the_single_pair_calls(sizes):
    """A batch of pairs with different N through pdsc_forward_testing_ragged (lists of per-pair tensors) against the same pairs
    one call each: inlier masks bit-exact, R/t within 1e-4 for every pair and within 2e-5 for all but at most one pair of a
    batch (same stages on the same rows; only the launch plans -- fp32 summation orders -- are the batch's, and ~5 % of random
    pairs sit on a hypothesis near-tie that a summation order moves by a few 1e-5).  The last case is
    too heterogeneous for one launch plan: the module groups it."""
    model, _ = _bench_model("n5000_b32")
    pairs = _ragged_pairs(sizes, 900 + len(sizes), inlier_ratio=0.3)
    with torch.no_grad():
        got = model(_as_lists(pairs))
        torch.cuda.synchronize()
        assert isinstance(got["final_labels"], list) and got["final_trans"].shape == (len(sizes), 4, 4) and got["M"] is None
        loose = 0
        for i, p in enumerate(pairs):
            one = _forward(model, p)
            assert got["final_labels"][i].shape == (sizes[i],)
            flips = int((got["final_labels"][i] != one["final_labels"][0]).sum())
            dT = float((got["final_trans"][i] - one["final_trans"][0]).abs().max())
            re, te = O.registration_errors(got["final_trans"][i].cpu(), p["gt_trans"][0])
            assert flips == 0 and dT < 1e-4, (i, sizes[i], flips, dT)
            loose += dT >= 2e-5
            assert re < 1.0 and te < 5.0, (i, re, te)
        assert loose <= 1, loose


def test_large_ragged_batch_is_bitwise_the_single_pair_calls_with_canonical_leaves():
    """24 pairs of 4100 .. 5333 correspondences in one ragged call: the batch takes the fused kNN (24 x 17 workgroups of 32 seeds) and
    the leaf-form attention with every pair cutting ITS OWN tiles into the leaf count of the longest pair; the single-pair calls take
    the two-launch kNN and their own plan.  With canonical leaves (same leaf count on both sides: all sizes > 1504) every pair's
    pose and mask are BIT-IDENTICAL to its own call -- the reference's semantics (one pair per call) at batch throughput."""
    model, _ = _bench_model("n5000_b32")
    sizes = [5333, 5000, 4999, 4100] + [4100 + 53 * i for i in range(20)]
    pairs = _ragged_pairs(sizes, 1700, inlier_ratio=0.3)
    try:
        model.att_leaves = "canonical"
        with torch.no_grad():
            got = model(_as_lists(pairs))
            torch.cuda.synchronize()
            for i, p in enumerate(pairs):
                one = _forward(model, p)
                assert torch.equal(got["final_trans"][i].view(torch.int32), one["final_trans"][0].view(torch.int32)), (i, sizes[i])
                assert torch.equal(got["final_labels"][i], one["final_labels"][0]), (i, sizes[i])
    finally:
        model.att_leaves = LEAVES_DEFAULT


def test_ragged_batch_across_leaf_count_classes_is_bitwise_the_single_pair_calls():
    """`att_leaves = "canonical"` promises that a pair's bits are a function of its N alone.  In a ragged batch the leaf count
    used to come from the LONGEST pair (a 1500-point pair: 8 leaves on its own, 4 next to a 1600-point pair) and a pair with fewer than
    two tiles per leaf silently took the per-launch key split.  r06: the module only lets pairs of one leaf-count class share a launch
    (PointDSC._ragged_groups) -- sizes on both sides of the 1504 boundary and a 300-point pair, every result bit-identical to the call on
    that pair alone."""
    model, _ = _bench_model("n5000_b32")
    sizes = [1500, 1600, 1504, 1505, 1000, 300, 2053, 640]
    assert {int(_lib.load().pdsc_attention_leaf_count(n)) for n in sizes} >= {4, 8}      # (and 2 for the 300-point pair)
    pairs = _ragged_pairs(sizes, 2100, inlier_ratio=0.3)
    try:
        model.att_leaves = "canonical"
        with torch.no_grad():
            got = model(_as_lists(pairs))
            torch.cuda.synchronize()
            for i, p in enumerate(pairs):
                one = _forward(model, p)
                assert torch.equal(got["final_trans"][i].view(torch.int32), one["final_trans"][0].view(torch.int32)), (i, sizes[i])
                assert torch.equal(got["final_labels"][i], one["final_labels"][0]), (i, sizes[i])
    finally:
        model.att_leaves = LEAVES_DEFAULT


@pytest.mark.parametrize("sizes,gemm,fmt", [((1000, 777, 640, 999), "f32", "f32"), ((3000, 4100, 5000, 5333), "f32", "u16"),
                                            ((2053, 2600), "h3", "f32")])
def test_ragged_batch_other_arithmetic_modes(sizes, gemm, fmt):
    """Ragged batches through the other kernels: fp32 layer GEMMs (small problems: the workgroup-per-tile kernel; larger ones:
    layer_wave_kernel) and the fp32 spatial-consistency matrix."""
    model, _ = _bench_model("n5000_b32")
    pairs = _ragged_pairs(sizes, 400 + len(sizes), inlier_ratio=0.3)
    model.layer_gemm, model.compat_format = gemm, fmt
    try:
        with torch.no_grad():
            got = model(_as_lists(pairs))
            torch.cuda.synchronize()
            loose = 0
            for i, p in enumerate(pairs):
                one = _forward(model, p)
                flips = int((got["final_labels"][i] != one["final_labels"][0]).sum())
                dT = float((got["final_trans"][i] - one["final_trans"][0]).abs().max())
                assert flips == 0 and dT < 1e-4, (i, sizes[i], flips, dT)
                loose += dT >= 2e-5
            assert loose <= 1
    finally:
        model.layer_gemm, model.compat_format = LAYER_GEMM_DEFAULT, COMPAT_FORMAT_DEFAULT


@pytest.mark.parametrize("leaves", ["canonical", "per_launch"])
def test_forward_does_not_read_workspace_rows_it_did_not_write(leaves):
    """ADVICE r04 (attention_split.hip: waves whose 32 queries all lie past a pair's last row return without writing their
    partial rows): every consumer must stay inside the rows the producers wrote.  The whole workspace is filled with NaN bit
    patterns between two identical forwards -- uniform batch with a ragged last tile, and a ragged batch -- and the results must
    not move by a bit (a stale-row read would now ingest NaN instead of last launch's plausible numbers)."""
    model, _ = _bench_model("n5000_b32")
    try:
        model.att_leaves = leaves
        for data in ({k: g(v) for k, v in synthetic.make_batch(3, 4999, seed=812, inlier_ratio=0.3).items() if k in ("corr_pos", "src_keypts", "tgt_keypts")},
                     _as_lists(_ragged_pairs((5000, 4777, 4100), 813, inlier_ratio=0.3))):
            data["testing"] = True
            with torch.no_grad():
                first = model(data)
                T = first["final_trans"].clone()
                L = [x.clone() for x in first["final_labels"]] if isinstance(first["final_labels"], list) else first["final_labels"].clone()
                torch.cuda.synchronize()
                for ws in model._workspaces.values():
                    ws[: ws.numel() // 4 * 4].view(torch.int32).fill_(0x7fc00001)          # quiet-NaN patterns everywhere
                again = model(data)
                torch.cuda.synchronize()
            assert torch.equal(again["final_trans"].view(torch.int32), T.view(torch.int32))
            if isinstance(L, list):
                assert all(torch.equal(a, b) for a, b in zip(again["final_labels"], L))
            else:
                assert torch.equal(again["final_labels"], L)
    finally:
        model.att_leaves = LEAVES_DEFAULT


def test_ragged_batch_padded_tensors_and_count_list():
    """The other calling form: tensors padded to the longest pair + data['num_corr']; padding rows hold garbage on purpose
    (NaN): nothing of a pair's result may depend on them; labels past a pair's count are zero.  Bit-identical to the list form."""
    model, _ = _bench_model("n5000_b32")
    sizes = (2500, 3100, 2048, 3099)
    pairs = _ragged_pairs(sizes, 77, inlier_ratio=0.3)
    n_max = max(sizes)
    data = {"testing": True, "num_corr": torch.tensor(sizes)}
    for k, wdt in (("corr_pos", 6), ("src_keypts", 3), ("tgt_keypts", 3)):
        t = torch.full((len(sizes), n_max, wdt), float("nan"))
        for i, p in enumerate(pairs):
            t[i, : sizes[i]] = p[k][0]
        data[k] = g(t)
    model.invalidate_packed_weights()          # the call below is then the FIRST forward after packing: the H3 range probe runs on this
    import warnings as _w                       # ragged batch and must not see the NaN padding (it would fall back to the fp32 GEMMs)
    with torch.no_grad(), _w.catch_warnings():
        _w.simplefilter("error", RuntimeWarning)
        a = model(data)
        assert model.layer_gemm == "h3" and max(model.last_range_probe.values()) < 3.0e4
        b = model(_as_lists(pairs))
    torch.cuda.synchronize()
    assert a["final_labels"].shape == (len(sizes), n_max) and bool(torch.isfinite(a["final_trans"]).all())
    assert torch.equal(a["final_trans"], b["final_trans"])
    for i, n in enumerate(sizes):
        assert torch.equal(a["final_labels"][i, :n], b["final_labels"][i]) and float(a["final_labels"][i, n:].abs().sum()) == 0.0
    # equal counts take the uniform entry point: bitwise the plain batched call
    same = _ragged_pairs((2048, 2048, 2048), 5, inlier_ratio=0.3)
    batch = {k: torch.cat([p[k] for p in same]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    u = _forward(model, batch)
    with torch.no_grad():
        r = model(dict({k: g(batch[k]) for k in batch}, testing=True, num_corr=[2048, 2048, 2048]))
    assert torch.equal(u["final_trans"], r["final_trans"]) and torch.equal(u["final_labels"], r["final_labels"])


def test_ragged_batch_with_pairs_of_at_most_k_rows():
    """The reference clamps the neighbour count per pair, k = min(k, num_corr - 1) (models/PointDSC.py:250); one launch has one k.
    A batch that mixes ordinary pairs with pairs of no more than k = 40 correspondences therefore runs those as their own calls
    (model._ragged_groups): every pair's result is still that of its own call -- bit for bit for the short ones, which ARE their own
    call -- and the C entry point refuses such a batch instead of reading neighbour lists it never selected."""
    import ctypes as C
    model, _ = _bench_model("n5000_b32")
    sizes = (100, 30, 257, 41, 40)
    pairs = _ragged_pairs(sizes, 31, inlier_ratio=0.5)
    with torch.no_grad():
        got = model(_as_lists(pairs))
        torch.cuda.synchronize()
    assert bool(torch.isfinite(got["final_trans"]).all())
    for i, p in enumerate(pairs):
        one = _forward(model, p)
        assert got["final_labels"][i].shape == (sizes[i],)
        if sizes[i] <= 40:
            assert torch.equal(got["final_trans"][i], one["final_trans"][0]) and torch.equal(got["final_labels"][i], one["final_labels"][0])
        else:
            # (same stages on the same rows, the batch's launch plans: at these tiny sizes -- 4 to 25 seeds -- a summation order can
            #  move a near-tie, so the bar here only separates "same registration" from garbage neighbour lists; the strict
            #  per-pair comparison of ragged launches is test_ragged_batch_equals_the_single_pair_calls, N >= 257)
            assert int((got["final_labels"][i] != one["final_labels"][0]).sum()) <= 3
            assert float((got["final_trans"][i] - one["final_trans"][0]).abs().max()) < 1e-3
    # the padded-tensor form of the same batch
    n_max = max(sizes)
    data = {"testing": True, "num_corr": list(sizes)}
    for k, wdt in (("corr_pos", 6), ("src_keypts", 3), ("tgt_keypts", 3)):
        t = torch.zeros(len(sizes), n_max, wdt)
        for i, p in enumerate(pairs):
            t[i, : sizes[i]] = p[k][0]
        data[k] = g(t)
    with torch.no_grad():
        pad = model(data)
    assert torch.equal(pad["final_trans"], got["final_trans"])
    # C ABI: n_min <= k is refused
    lib = _lib.load()
    cfg = model._config()
    bs, n, S = 2, 100, 10
    z = lambda *shape, dt=torch.float32: torch.zeros(*shape, device=DEV, dtype=dt)
    cnt, sds = torch.tensor([100, 30], device=DEV, dtype=torch.int32), torch.tensor([10, 3], device=DEV, dtype=torch.int32)
    nb = int(lib.pdsc_workspace_bytes(C.byref(cfg), bs, n, S))
    ws = torch.empty(nb, device=DEV, dtype=torch.uint8)
    T, L = z(bs, 4, 4), z(bs, n)
    P = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.pdsc_forward_testing_ragged(C.byref(cfg), P(model.packed_weights()), P(model.split_weights()), P(z(bs, n, 6)), P(z(bs, n, 3)),
                                         P(z(bs, n, 3)), bs, n, S, P(cnt), P(sds), 30, P(T), P(L), P(ws), nb, torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and "clamps k per pair" in _lib.last_error()


def test_ragged_batch_rejections():
    model, _ = _bench_model("n5000_b32")
    pairs = _ragged_pairs((300, 400), 3)
    data = _as_lists(pairs)
    # (r06: the exact-fp32 mode takes ragged batches -- test_exact_fp32_mode_takes_ragged_batches; what is still refused:)
    del data["testing"]
    with pytest.raises(NotImplementedError):
        model(data)
    with pytest.raises(ValueError):
        model({"corr_pos": g(torch.zeros(2, 100, 6)), "src_keypts": g(torch.zeros(2, 100, 3)), "tgt_keypts": g(torch.zeros(2, 100, 3)),
               "num_corr": [100, 101], "testing": True})


def test_forwards_in_flight_reproduce_the_plain_calls():
    """pointdsc_amd.pipeline.InFlight: consecutive batches alternate between two HIP streams / workspaces so that one forward's
    latency-bound tail overlaps the next one's first kernels.  Same launches on the same data: bit-identical results, uniform
    and ragged batches, and the packed weights are built before the streams diverge."""
    from pointdsc_amd.pipeline import InFlight
    model, _ = _bench_model("n5000_b32")
    batches = [workloads.batch("n5000_b32", 3 * i, 3) for i in range(5)]
    plain = [_forward(model, b) for b in batches]
    model.invalidate_packed_weights()                      # the runner must rebuild them before going multi-stream
    runner = InFlight(model, depth=2, tail_streams=True)   # (each forward: encoder on the slot's stream, tail on its high-priority stream)
    assert runner.tail_streams
    plain_streams = InFlight(model, depth=2, tail_streams=False)
    for b, p in list(zip(batches, plain))[:2]:
        data = {k: g(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
        data["testing"] = True
        o = plain_streams(data)
        plain_streams.synchronize()
        assert torch.equal(o["final_trans"], p["final_trans"]) and torch.equal(o["final_labels"], p["final_labels"])
    plain_streams.close()
    outs = []
    for b in batches:
        data = {k: g(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
        data["testing"] = True
        outs.append(runner(data))
    runner.synchronize()
    for o, p in zip(outs, plain):
        assert torch.equal(o["final_trans"], p["final_trans"]) and torch.equal(o["final_labels"], p["final_labels"])
    # ... and with every slot's forward replayed as a captured hipGraph (static per-slot inputs, outputs returned as copies)
    gr = InFlight(model, depth=3, graphs=True)
    outs = []
    for rep in range(4):
        for b in batches:
            data = {k: g(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
            data["testing"] = True
            outs.append(gr(data))
    gr.synchronize()
    assert gr.graphs and gr._captured, "hipGraph capture fell back to the eager path"
    for o, p in zip(outs, plain * 4):
        assert torch.equal(o["final_trans"], p["final_trans"]) and torch.equal(o["final_labels"], p["final_labels"])
    pairs = _ragged_pairs((2100, 2600, 2222), 31, inlier_ratio=0.3)
    with torch.no_grad():
        want = model(_as_lists(pairs))
    got = [runner(_as_lists(pairs)) for _ in range(3)] + [gr(_as_lists(pairs))]         # (ragged: eager path in either runner; the two
    runner.synchronize()                                                                # runners own disjoint workspace slots)
    gr.synchronize()
    for r in got:
        assert torch.equal(r["final_trans"], want["final_trans"])
        assert all(torch.equal(a, b) for a, b in zip(r["final_labels"], want["final_labels"]))


def test_replayed_hipgraph_forwards_reproduce_the_plain_calls():
    """InFlight(graphs=True) on the launch-bound shape it exists for (one pair of N = 1000 per forward, bench.py's n1000_b1): every
    slot's forward is captured once and replayed; 4 forwards in flight, 400 replays over 8 different pairs, each bit-identical to
    the plain call (static per-slot inputs are overwritten between replays, outputs come back as copies)."""
    from pointdsc_amd.pipeline import InFlight
    model, _ = _bench_model("n1000_b1")
    datas, plain = [], []
    for i in range(8):
        b = workloads.batch("n1000_b1", i, 1)
        d = {k: g(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
        d["testing"] = True
        datas.append(d)
        plain.append(_forward(model, b))
    gr = InFlight(model, depth=4, graphs=True)
    outs = [gr(datas[i % 8]) for i in range(400)]
    gr.synchronize()
    assert gr.graphs and len(gr._captured) == 4, "hipGraph capture fell back to the eager path"
    bad = [i for i, o in enumerate(outs) if not (torch.equal(o["final_trans"], plain[i % 8]["final_trans"]) and
                                                 torch.equal(o["final_labels"], plain[i % 8]["final_labels"]))]
    assert not bad, f"replayed forwards differing from the plain call: {bad[:10]} of {len(outs)}"
    gr.close()


def test_zero_copy_hipgraph_forwards_reproduce_the_plain_calls():
    """InFlight(graphs=True, zero_copy=True): each slot's graph is captured on the CALLER's input tensors and the graph's own output
    tensors are handed out (no staging copies, no clones).  Two input buffer sets alternate (two graphs per slot), their contents
    are overwritten in place between uses with other pairs, and every result -- read before its slot runs again -- is bit-identical
    to the plain call on the same pair; a third buffer set of another shape falls back to a fresh capture."""
    from pointdsc_amd.pipeline import InFlight
    model, _ = _bench_model("n1000_b1")
    pairs = [workloads.batch("n1000_b1", i, 1) for i in range(6)]
    plain = [_forward(model, b) for b in pairs]
    names = ("corr_pos", "src_keypts", "tgt_keypts")
    bufs = [{k: g(pairs[0][k]).clone() for k in names} for _ in range(2)]
    for b in bufs:
        b["testing"] = True
    depth = 3
    gr = InFlight(model, depth=depth, graphs=True, zero_copy=True)
    pending, bad = [], []
    for i in range(120):
        buf = bufs[i % 2]
        if len(pending) >= 2:                       # the previous forward on THIS buffer set has finished before it is overwritten,
            j, r = pending.pop(0)                   # and its result is read before its slot runs again (depth = 3 > 2 pending)
            r["ready"].synchronize()
            if not (torch.equal(r["final_trans"], plain[j % 6]["final_trans"]) and torch.equal(r["final_labels"], plain[j % 6]["final_labels"])):
                bad.append(j)
        for k in names:
            buf[k].copy_(g(pairs[i % 6][k]))
        pending.append((i, gr(buf)))
    gr.synchronize()
    for j, r in pending:
        if not (torch.equal(r["final_trans"], plain[j % 6]["final_trans"]) and torch.equal(r["final_labels"], plain[j % 6]["final_labels"])):
            bad.append(j)
    assert gr.graphs and gr.zero_copy and len(gr._captured) == depth and all(len(c) == 2 for c in gr._captured.values())
    assert not bad, f"zero-copy replays differing from the plain call: {bad[:10]}"
    # results ARE the graph's output tensors: the same storage comes back `depth` calls later
    r0 = gr(bufs[0]); [gr(bufs[0]) for _ in range(depth - 1)]; r1 = gr(bufs[0])
    gr.synchronize()
    assert r0["final_trans"].data_ptr() == r1["final_trans"].data_ptr()
    gr.close()


@pytest.mark.parametrize("mode", ["plain streams", "tail streams", "hipGraphs"])
def test_forwards_in_flight_stay_exact_under_load(mode):
    """Regression test of r03's exactness findings: batches of 2 pairs of N = 5000 kept in flight -- the shape on
    which the old scoring stage (hipMemsetAsync + atomics; compiler-paired packed-fp32 inlier test) lost votes on 0.2-35 % of the
    forwards depending on the mode -- 600 forwards per mode, every one bit-identical to the plain call on the same batch."""
    from pointdsc_amd.pipeline import InFlight
    model, _ = _bench_model("n5000_b32")
    datas, plain = [], []
    for i in range(4):
        b = workloads.batch("n5000_b32", 2 * i, 2)
        d = {k: g(b[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
        d["testing"] = True
        datas.append(d)
        plain.append(_forward(model, b))
    runner = InFlight(model, depth=3, graphs=True) if mode == "hipGraphs" else InFlight(model, depth=2, tail_streams=(mode == "tail streams"))
    bad = 0
    for rep in range(150):
        outs = [runner(d) for d in datas]
        runner.synchronize()
        bad += sum(not (torch.equal(o["final_trans"], p["final_trans"]) and torch.equal(o["final_labels"], p["final_labels"]))
                   for o, p in zip(outs, plain))
    runner.close()
    assert bad == 0, f"{bad} of 600 forwards in flight ({mode}) differ from the plain call"


def test_other_entry_points_stay_exact_beside_attention_launches():
    """The r04 reproducer (tools/pk_f32_repro.hip) showed packed fp32 with operand selects returning wrong lanes whenever a wave
    shares a CU with a wave that interleaves MFMAs with vector work -- the r03 attention kernel, or 100 % of the launches beside the
    synthetic "mix" neighbour; the library ships no such instruction any more (test_library_ships_no_packed_fp32_with_operand_
    selects).  This is the behavioural side: the entry points outside the testing forward -- the compat builds (hand-written packed
    math with default selects), the NMS keys, the SM baseline, the validation forward, hypothesis scoring -- run 150 times each
    while that synthetic neighbour (pdsc_selftest_mfma_valu_neighbour) AND attention launches of two pairs of N = 5000 keep the chip
    busy on other streams; every result is bit-identical to the unloaded call.  In experiments builds the r03 scoring kernel
    (compiler-paired packed fp32, PDSC_SCORE_SLP=1) serves as the positive control: beside the same neighbour it must miscount."""
    import ctypes as C
    from pointdsc_amd import baselines
    model, _ = _bench_model("n5000_b32")
    n, bs = 5000, 2
    batch = workloads.batch("n5000_b32", 40, bs)
    src, tgt, corr = g(batch["src_keypts"]), g(batch["tgt_keypts"]), g(batch["corr_pos"])
    sig = torch.tensor([0.1], device=DEV)
    conf = torch.randn(bs, n, generator=torch.Generator().manual_seed(5)).to(DEV)
    seed_trans = batch["gt_trans"].repeat_interleave(64, 0).reshape(bs, 64, 4, 4).clone()
    seed_trans[:, :, :3, 3] += 0.01 * torch.randn(bs, 64, 3, generator=torch.Generator().manual_seed(6))
    seed_trans = g(seed_trans)
    small = workloads.batch("n1000_b1", 7, 1)
    sdata = {k: g(small[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}      # validation forward: no 'testing' key
    calls = {
        "compat_u16": lambda: ops.spatial_compat_u16(src, tgt, sig),
        "compat_f32": lambda: ops.spatial_compat(src, tgt, sig),
        "nms_keys": lambda: ops.nms_keys_grid(src, conf, 0.1),
        "sm_baseline": lambda: torch.cat([t.reshape(-1).float() for t in baselines.SM(corr, src, tgt, 0.10)]),
        "validation_forward": lambda: torch.cat([model(sdata)[k].reshape(-1) for k in ("final_trans", "final_labels", "M")]),
        "score_hypotheses": lambda: ops.score_hypotheses(seed_trans, src, tgt, 0.10)[0].float(),
    }
    with torch.no_grad():
        want = {k: f().clone() for k, f in calls.items()}
        torch.cuda.synchronize()
        # the load: split-precision attention launches on a side stream, enqueued ahead of every call under test
        c16 = ops.spatial_compat_u16(src, tgt, sig)
        qkv = (torch.randn(bs * n, 384, generator=torch.Generator().manual_seed(0)) * 0.3).to(DEV)
        qs, kv = ops.pack_qkv_split(qkv, bs, n)
        side, side2 = torch.cuda.Stream(), torch.cuda.Stream()
        sink = torch.zeros(4, device=DEV)
        lib = _lib.load()

        def load_the_chip():
            with torch.cuda.stream(side):
                for _ in range(3):
                    ops.sc_attention_split(qs, kv, c16, bs, n)
            _lib.check(lib.pdsc_selftest_mfma_valu_neighbour(C.c_void_p(sink.data_ptr()), 512, 5000, side2.cuda_stream), "neighbour")

        bad = {k: 0 for k in calls}
        for rep in range(150):
            load_the_chip()
            for k, f in calls.items():
                bad[k] += not torch.equal(f(), want[k])
            torch.cuda.synchronize()
        assert not any(bad.values()), bad
        if EXPERIMENTS:
            # positive control: the compiler-paired scoring kernel of r03 loses votes beside the same neighbour
            import os
            os.environ["PDSC_SCORE_SLP"] = "1"
            try:
                wrong = 0
                for rep in range(50):
                    load_the_chip()
                    wrong += not torch.equal(calls["score_hypotheses"](), want["score_hypotheses"])
                    torch.cuda.synchronize()
            finally:
                os.environ.pop("PDSC_SCORE_SLP", None)
            assert wrong > 0, "the r03 packed-fp32 scoring kernel did not miscount beside the synthetic neighbour"


def test_h3_falls_back_to_fp32_gemms_outside_the_fp16_range():
    """layer_gemm = "h3" carries the operands of the fc_message / PointCN GEMMs as fp16 hi + lo (|x| < 65504).  A checkpoint whose
    folded weights or activations -- HIDDEN ones included -- leave that range must not produce inf / NaN silently: the module
    checks the packed weights and, before the first forward after packing, every activation kind of every layer
    (pdsc_encoder_range_probe), warns and continues with the fp32 GEMMs."""
    kw = dict(KW, num_layers=2)
    pair = synthetic.make_pair(400, inlier_ratio=0.4, seed=3)
    # only a HIDDEN activation out of range, every folded weight far below the weight check's 3e4: v x 1e4 makes the message ~1e3-1e4,
    # fc_message's first conv x 100 lifts its hidden layer to ~1e6, the second conv x 1e-6 brings the chain back to O(1)
    nl = "encoder.blocks.NonLocal_layer_0."
    hidden_only = {nl + "projection_v.weight": 1.0e4, nl + "projection_v.bias": 1.0e4, nl + "fc_message.0.weight": 1.0e2,
                   nl + "fc_message.3.weight": 1.0e-6}
    # (r05) only the attention's operands out of range: k of one layer ~1e5 while every weight stays below 3e4 and the residual
    # stream O(1) -- the attention must leave its fp16 operand pairs
    k_only = {nl + "projection_k.weight": 6.0e4, nl + "projection_k.bias": 6.0e4}     # (|w| stays below the weight check's 3e4: 0.35 x 6e4)
    for scales, kind in (({"encoder.layer0.weight": 3.0e5}, None), ({"encoder.blocks.PointCN_layer_1.0.weight": 1.0e6}, None),
                         (hidden_only, "fc_message hidden 1"), (k_only, "q|k|v")):
        model = PointDSC(**kw)
        sd = synthetic.make_state_dict(model.state_dict(), seed=2)
        for key, scale in scales.items():
            sd[key] = sd[key] * scale
        model.load_state_dict(sd)
        model = model.eval().to(DEV)
        assert model.layer_gemm == "h3"
        with pytest.warns(RuntimeWarning, match="fp16 range"):
            res = _forward(model, pair)
        assert bool(torch.isfinite(res["final_trans"]).all())
        if kind == "q|k|v":
            assert model.attention_precision == "fp32", model.attention_precision      # (the H3 GEMMs leave with it: any kind out of range does that)
            assert model.last_range_probe[kind] > 3.0e4 and model.last_range_probe["feature"] < 3.0e4, model.last_range_probe
        else:
            assert model.layer_gemm == "f32"
            # weights out of range take the attention with them; a hidden fc_message activation alone does not (the attention
            # follows ITS operands: PointCN output and q|k|v)
            att_out = kind is None or not max(model.last_range_probe["PointCN"], model.last_range_probe["q|k|v"]) < 3.0e4
            assert model.attention_precision == ("fp32" if att_out else "fp16x3"), (kind, model.attention_precision, getattr(model, "last_range_probe", None))
        if kind is not None and kind != "q|k|v":
            # only a hidden activation leaves the range: the final features (all the r03 guard looked at) stay small
            probe = model.last_range_probe
            assert probe[kind] > 65504.0 and probe["feature"] < 3.0e4, probe
        again = _forward(model, pair)
        assert bool(torch.isfinite(again["final_trans"]).all())
    # ... and a well-scaled checkpoint stays on H3, its probe inside the range
    model, _ = _bench_model("n5000_b32")
    model.invalidate_packed_weights()
    _forward(model, workloads.batch("n5000_b32", 0, 1))
    assert model.layer_gemm == "h3" and model.attention_precision == "fp16x3" and max(model.last_range_probe.values()) < 3.0e4, model.last_range_probe


def _range_model(layers=2):
    model = PointDSC(**dict(KW, num_layers=layers))
    model.load_state_dict(synthetic.make_state_dict(model.state_dict(), seed=2))
    return model.eval().to(DEV)


def _scaled(pair, factor):
    """The same correspondences with every coordinate multiplied by `factor`: layer0 is linear in corr_pos, so the activations that
    become fp16 operand pairs scale with it -- 1e5 lifts them past 65504 on a checkpoint that is perfectly in range at 3 m."""
    return {k: (v * factor if k in ("corr_pos", "src_keypts", "tgt_keypts") else v) for k, v in pair.items()}


def test_range_guard_catches_a_later_input_that_leaves_the_fp16_range():
    """VERDICT r05 missing 2 / ADVICE r05 (medium): the first-input range probe said nothing about LATER inputs -- an activation past
    65504 became hi = inf, lo = NaN and reached final_trans silently.  r06: every forward carries a device-side sentinel.
      * plain module call (range_guard = "sync"): a benign first batch, then a batch whose activations leave the range: the call
        warns, switches the module to exact fp32 (kept) and returns the fp32 answer of THAT call -- bit for bit what a module set to
        fp32 from the start returns -- not NaN;
      * the library itself (guard "off" = what a C caller gets): the pair that left the range comes back with a NaN pose and a set
        word in the workspace entry "range_flag"; the in-range pair of the same batch is untouched, bit for bit;
      * inside a pipeline (InFlight, "lazy"): the affected forward returns NaN poses, the module warns and switches once the words
        have been read, the forwards that follow are exact fp32."""
    import warnings as _w
    small = synthetic.make_pair(400, inlier_ratio=0.4, seed=3)
    big = _scaled(synthetic.make_pair(400, inlier_ratio=0.4, seed=4), 1.0e5)
    # -- sync
    model = _range_model()
    with _w.catch_warnings():
        _w.simplefilter("error")                                   # the benign batch must not warn
        first = _forward(model, small)
    assert model.attention_precision == "fp16x3" and model.range_fallbacks == 0 and bool(torch.isfinite(first["final_trans"]).all())
    with pytest.warns(RuntimeWarning, match="fp16 range"):
        got = _forward(model, big)
    assert model.attention_precision == "fp32" and model.layer_gemm == "f32" and model.range_fallbacks == 1
    ref = _range_model()
    ref.attention_precision, ref.layer_gemm = "fp32", "f32"
    want = _forward(ref, big)
    assert bool(torch.isfinite(got["final_trans"]).all())
    assert torch.equal(got["final_trans"], want["final_trans"]) and torch.equal(got["final_labels"], want["final_labels"])
    # -- the library alone: pair 0 in range, pair 1 not
    model = _range_model()
    model.range_guard = "off"
    _forward(model, small)                                         # (probe on a benign input, as a C caller's first call would be)
    two = {k: torch.cat([small[k], big[k]]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    res = _forward(model, two)
    alone = _forward(model, small)
    _forward(model, two)
    flags = model.workspace_view("range_flag", 2, 400, torch.int32)[:2].cpu().tolist()
    assert flags == [0, 1], flags
    assert bool(torch.isnan(res["final_trans"][1]).all()) and bool(torch.isfinite(res["final_trans"][0]).all())
    assert torch.equal(res["final_trans"][0], alone["final_trans"][0]) and torch.equal(res["final_labels"][0], alone["final_labels"][0])
    assert model.attention_precision == "fp16x3"                  # ("off": the module does not look)
    # -- lazy, inside a pipeline
    from pointdsc_amd.pipeline import InFlight
    model = _range_model()
    run = InFlight(model, depth=2)
    d_small = dict({k: g(small[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}, testing=True)
    d_big = dict({k: g(big[k]) for k in ("corr_pos", "src_keypts", "tgt_keypts")}, testing=True)
    r0 = run(d_small)
    r1 = run(d_big)
    with pytest.warns(RuntimeWarning, match="fp16 range"):
        run.synchronize()
        model.check_range()
    assert bool(torch.isfinite(r0["final_trans"]).all()) and bool(torch.isnan(r1["final_trans"]).all())
    assert model.attention_precision == "fp32" and model.range_fallbacks == 1
    r2 = run(d_big)
    run.synchronize()
    assert torch.equal(r2["final_trans"], want["final_trans"])
    run.close()


def test_exact_fp32_mode_takes_ragged_batches():
    """VERDICT r05 missing 3: the exact-fp32 mode -- also what the range guard falls back to -- refused ragged batches, so a module that
    had fallen back lost the evaluation loop with one N per pair (evaluation/test_3DMatch.py:126).  The fp32 attention kernel now takes
    the per-pair counts: pair i of a ragged batch = the call on its own N_i rows -- labels bit for bit, the pose within fp32 round-off
    (1e-5: the exact-fp32 kernel plans its key split per launch, so a pair's summation order follows the batch it is in; the
    batch-invariant leaf form exists for the split-precision attention only)."""
    model = _range_model(layers=3)
    model.attention_precision, model.layer_gemm, model.compat_format = "fp32", "f32", "f32"
    counts = [1000, 733, 412, 999]
    pairs = [synthetic.make_pair(c, inlier_ratio=0.35, seed=50 + i) for i, c in enumerate(counts)]
    data = {k: [g(p[k][0]) for p in pairs] for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    data["testing"] = True
    with torch.no_grad():
        res = model(data)
    torch.cuda.synchronize()
    for i, p in enumerate(pairs):
        one = _forward(model, p)
        assert float((res["final_trans"][i] - one["final_trans"][0]).abs().max()) < 1e-5, (i, (res["final_trans"][i] - one["final_trans"][0]).abs().max())
        assert torch.equal(res["final_labels"][i], one["final_labels"][0]), i
    # padded form with NaN in the padding rows: nothing of a result may depend on them
    n_max = max(counts)
    pad = {k: torch.full((len(counts), n_max, pairs[0][k].shape[-1]), float("nan")) for k in ("corr_pos", "src_keypts", "tgt_keypts")}
    for i, p in enumerate(pairs):
        for k in pad:
            pad[k][i, : counts[i]] = p[k][0]
    d2 = dict({k: g(v) for k, v in pad.items()}, testing=True, num_corr=counts)
    with torch.no_grad():
        res2 = model(d2)
    torch.cuda.synchronize()
    assert torch.equal(res2["final_trans"], res["final_trans"])
    for i in range(len(counts)):
        assert torch.equal(res2["final_labels"][i, : counts[i]], res["final_labels"][i])


@pytest.mark.parametrize("gemm,fmt", [("h3", "u16"), ("f32", "f32")])
@pytest.mark.parametrize("n,bs", [(1000, 1), (1003, 3), (2053, 2), (5000, 4)])
def test_forward_reads_nothing_it_did_not_write(n, bs, gemm, fmt):
    """Header contract: results depend on the arguments only.  The workspace is filled with NaN bit patterns before a forward;
    poses and labels must equal those of a forward over a workspace that holds the previous call's (plausible, finite) values --
    any stage that reads a workspace location before this call wrote it would show up as NaN or as a different result."""
    model, _ = _bench_model("n5000_b32")
    batch = synthetic.make_batch(bs, n, seed=700 + n, inlier_ratio=0.3)
    model.compat_format, model.layer_gemm = fmt, gemm
    try:
        want = _forward(model, batch)
        want = (want["final_trans"].clone(), want["final_labels"].clone())
        for ws in model._workspaces.values():
            ws.view(torch.int32 if ws.numel() % 4 == 0 else torch.uint8).fill_(-1 if ws.numel() % 4 == 0 else 255)      # all-ones = NaN
        got = _forward(model, batch)
    finally:
        model.compat_format, model.layer_gemm = COMPAT_FORMAT_DEFAULT, LAYER_GEMM_DEFAULT
    assert bool(torch.isfinite(got["final_trans"]).all())
    assert torch.equal(got["final_trans"], want[0]) and torch.equal(got["final_labels"], want[1])
