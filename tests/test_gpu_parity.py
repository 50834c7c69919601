"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Run on the MI355X box with `pytest -m gpu`.

Tolerances (north_star): class ids and box coordinates equal after integer rounding, scores within
1e-4 (fp32).  Raw head tensors: |err| <= 2e-4 absolute on the logits (a bound that implies 1e-4 on the
scores; measured ~1e-5 after 63 fp32 layers), and the decoded scores themselves within 1e-4 on the zoo
models.  NMS on identical inputs is bit-exact (integer index work)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import yololite_amd as ya
from yololite_amd import _lib
from yololite_amd.program import build_program, synth_state_dict, zoo_meta, make_meta
from oracle import model as omodel
from oracle import postproc as opost

DEV = "cuda:0"


def _oracle_for(meta, sd):
    m = omodel.build_from_meta(meta).eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    return m


def _hip_for(meta, sd, fuse_dw="auto", fuse_stem=True, fuse_uib=False, **kw):
    m = ya.build_model_from_meta(meta, fuse_dw=fuse_dw, fuse_stem=fuse_stem, fuse_uib=fuse_uib, **kw)
    m.load_state_dict(sd)
    return m.to(DEV)


def _x(B, S, seed=1234):
    """uniform u8 image normalised with the ImageNet mean/std."""
    rng = np.random.RandomState(seed)
    u8 = rng.randint(0, 256, size=(B, S, S, 3)).astype(np.float32) / 255.0
    im = (u8 - np.array([0.485, 0.456, 0.406], np.float32)) / np.array([0.229, 0.224, 0.225], np.float32)
    return torch.from_numpy(np.ascontiguousarray(im.transpose(0, 3, 1, 2)))


def _cmp_levels(outs, ref, atol=2e-4, rtol=2e-6, C=None):
    """Raw head logits: max |err| <= 2e-4 (+ 2e-6 * max|ref|: two ulps of the largest logit) per level.  A logit
    error of 2e-4 moves a sigmoid by at most 5e-5 and a score = sigmoid(obj) * sigmoid(cls) by at most 1e-4 -- the
    north_star bar -- which is ALSO asserted directly on the decoded scores when the row layout is known (C)."""
    for l, (o, r) in enumerate(zip(outs, ref)):
        assert tuple(o.shape) == tuple(r.shape)
        err = (o.cpu() - r).abs().max().item()
        bound = atol + rtol * r.abs().max().item()
        assert err <= bound, f"level {l}: max abs err {err} > {bound}"
        if C is not None and C > 1:
            sc = lambda t: torch.sigmoid(t[..., 4]) * torch.sigmoid(t[..., 5:5 + C]).max(-1).values
            serr = (sc(o.cpu()) - sc(r)).abs().max().item()
            assert serr <= 1e-4, f"level {l}: score error {serr} > 1e-4"


# ------------------------------------------------------------------------------------------ forward
TINY = [
    dict(arch="YOLOLiteMS_CPU", backbone="oracle_tiny", num_classes=3, fpn_channels=16, depth_multiple=0.5, head_depth=1),
    dict(arch="YOLOLiteMS_CPU", backbone="oracle_tiny", num_classes=5, fpn_channels=24, depth_multiple=1.0,
         width_multiple=0.85, head_depth=2, use_p6=True, anchors=2),
    dict(arch="YOLOLiteMS", backbone="oracle_tiny", num_classes=1, fpn_channels=20, depth_multiple=1.0, head_depth=1),
    dict(arch="YOLOLiteMS", backbone="oracle_tiny_tf", num_classes=4, fpn_channels=16, depth_multiple=0.5, head_depth=2,
         use_p2=True),
    # efficientnetv2-style: ConvBnAct with residual, fused MBConv (strided / residual), MBConv + squeeze-excite, SiLU
    dict(arch="YOLOLiteMS", backbone="oracle_tiny_v2", num_classes=6, fpn_channels=24, depth_multiple=0.5, head_depth=1),
    # hgnetv2-style (round 5): StemV2 (2x2 convs on a zero-extended map, max-pool, concat), plain and light HG blocks with
    # concat + two 1x1 aggregation convs, residual second block, depthwise downsample, ReLU + learnable affine everywhere
    dict(arch="YOLOLiteMS_CPU", backbone="oracle_tiny_hg", num_classes=4, fpn_channels=16, depth_multiple=0.5, head_depth=1),
    # convnextv2-style (round 5): 4x4 s4 stem on the NHWC4 copy of the input, LayerNorm2d, depthwise 7x7, LayerNorm, Linear +
    # GELU, GlobalResponseNorm gate, gated Linear + residual, LayerNorm2d + 2x2 s2 downsample
    dict(arch="YOLOLiteMS", backbone="oracle_tiny_cnx", num_classes=3, fpn_channels=16, depth_multiple=0.5, head_depth=1),
]


@pytest.mark.parametrize("idx", range(len(TINY)))
@pytest.mark.parametrize("fuse", [True, False, "auto"])
def test_forward_tiny_models(idx, fuse):
    """every block flavour (cn k3/k1, UIB with dw_start / dw_mid / both / strided / residual, TF-SAME
    padding, ReLU/ReLU6/SiLU, dense and depthwise smooth blocks, P2/P6 levels, A=2) on a tiny net;
    odd sizes exercise partial tiles."""
    meta = make_meta(img_size=96, **TINY[idx])
    sd = synth_state_dict(meta, seed=10 + idx)
    x = _x(3, 96, seed=idx)
    with torch.no_grad():
        ref = _oracle_for(meta, sd)(x)
    m = _hip_for(meta, sd, fuse_dw=fuse, fuse_stem=(fuse != False), fuse_uib=(fuse != True))
    outs = m(x.to(DEV))
    _cmp_levels(outs, ref)
    assert m.get_strides() == _oracle_for(meta, sd).get_strides()
    m.export_concat = True
    cat = m(x.to(DEV))
    assert cat.shape == (3, sum(o[0].numel() // o.shape[-1] for o in outs), outs[0].shape[-1])


def test_forward_reference_fixture_weights(golden_dir):
    """HIP forward on the state_dicts stored by the REFERENCE classes (tests/golden/neck_head.npz)
    against the outputs the reference computed."""
    z = np.load(os.path.join(golden_dir, "neck_head.npz"))
    with open(os.path.join(golden_dir, "neck_head_cases.json")) as f:
        cases = json.load(f)
    for c in cases:
        tag, kw = c["tag"], c["kw"]
        meta = make_meta(arch=c["cls"], backbone="oracle_tiny", num_classes=kw["num_classes"], img_size=64,
                         fpn_channels=kw["fpn_channels"], depth_multiple=kw["depth_multiple"],
                         width_multiple=kw["width_multiple"], head_depth=kw["head_depth"], use_p6=kw["use_p6"],
                         use_p2=kw["use_p2"], anchors=kw["num_anchors_per_level"][0])
        sd = {k[len(tag) + 4:]: z[k] for k in z.files if k.startswith(tag + "/sd/")}
        m = _hip_for(meta, sd)
        outs = m(torch.from_numpy(z[f"{tag}/x"]).to(DEV))
        for j, o in enumerate(outs):
            r = z[f"{tag}/out{j}"]
            err = np.abs(o.cpu().numpy() - r).max()
            assert err <= 2e-4 + 2e-6 * np.abs(r).max(), (tag, j, err)       # the bound of _cmp_levels (implies 1e-4 on a score)


@pytest.mark.parametrize("name,B,S", [("edge_n", 2, 640), ("edge_m", 1, 320), ("yololite_m", 1, 256),
                                      ("edge_n", 1, 416), ("yololite_m", 1, 224), ("edge_m", 2, 352),
                                      ("yololite_m_v2", 2, 256), ("yololite_n_v2", 3, 224), ("yololite_m_v2", 1, 640)])
@pytest.mark.parametrize("uib", [False, True])
def test_forward_zoo_models(name, B, S, uib):
    """BASELINE configs 2-4 backbones/necks/heads at reduced batch; uib=True also runs the inverted-residual
    blocks as single fused launches (expand -> depthwise -> project).  416 / 224 / 352: level grids of 13, 7, 11 pixels
    (not multiples of the 4x4 / 4x2 tiles of the depthwise kernels: partial tiles and the fallback kernels)."""
    meta = zoo_meta(name, 80, S)
    sd = synth_state_dict(meta, seed=0)
    x = _x(B, S)
    with torch.no_grad():
        ref = _oracle_for(meta, sd)(x)
    outs = _hip_for(meta, sd, fuse_uib=uib)(x.to(DEV))
    _cmp_levels(outs, ref, C=80)


@pytest.mark.parametrize("name", sorted(__import__("yololite_amd").program.MODEL_ZOO))
def test_every_zoo_config_runs_on_the_gpu(name):
    """EVERY model yaml of the reference whose backbone is restated (configs/models/*.yaml, configs/v2_models/*.yaml ->
    program.MODEL_ZOO) through the HIP path against the oracle: batch 1 at 320 and at 640, C = 80, raw logits and decoded
    scores.  F = 196 / 256 / 512 necks, head_depth 1 / 2 / 3, depth_multiple 1.5 (three dense 3x3 per level), lite0-lite4,
    the three efficientnetv2 sizes, all four mobilenetv4 edge models."""
    for S, seed in ((320, 3), (640, 4)):
        meta = zoo_meta(name, 80, S)
        sd = synth_state_dict(meta, seed=seed)
        x = _x(1, S, seed=seed)
        with torch.no_grad():
            ref = _oracle_for(meta, sd)(x)
        outs = _hip_for(meta, sd)(x.to(DEV))
        _cmp_levels(outs, ref, C=80)


def test_convnext_and_hgnet_ops_are_deterministic_and_batch_invariant():
    """ABI v5 ops (YL_OP_POOL / COPY / LN / GRN / NHWC4, GELU, ReLU + learnable affine): the GRN gate is a fixed-order two-pass
    sum per image (no float atomics), LayerNorm a per-pixel wave reduction -- the forward is bitwise repeatable and an
    image's rows do not depend on the batch it is part of or on the chunk split."""
    for idx in (len(TINY) - 2, len(TINY) - 1):
        meta = make_meta(img_size=96, **TINY[idx])
        m = _hip_for(meta, synth_state_dict(meta, seed=21 + idx))
        ctx = m._ctx_for(96)
        ops = {l.op for l in m.program.layers}
        assert ({_lib.OP_POOL, _lib.OP_COPY} <= ops) if "hg" in TINY[idx]["backbone"] else ({_lib.OP_LN, _lib.OP_GRN, _lib.OP_NHWC4} <= ops)
        x = _x(9, 96, seed=5).to(DEV)
        a = [t.clone() for t in m(x)]
        for streams in (1, 2, 3):
            ctx.set_option("streams", streams)
            for u, v in zip(a, m(x)):
                assert torch.equal(u, v)
        ctx.set_option("streams", 1)
        for b in (0, 4, 8):
            for u, v in zip(a, m(x[b:b + 1])):
                assert torch.equal(u[b:b + 1], v)


@pytest.mark.parametrize("name,S", [("yololite_m", 256), ("yololite_m", 640), ("yololite_n", 320), ("yololite_xl", 256)])
def test_streamed_tap_depthwise_kernel_is_bitwise_the_two_launches(name, S):
    """yl_conv_dws_kernel (round 5: depthwise k x k -> 1x1 with the 1x1 AND the tap weights streamed through LDS, halo patch in
    LDS) against the two launches it replaces (yl_dw_tile_kernel, then the plain 1x1): same fmaf chain per channel, same k
    order and epilogues -> the raw levels are bitwise equal.  tf_efficientnet_lite0 / lite2 / lite4: 5x5 stride 1 and 2, 3x3,
    8 / 13 / 22 n-tile accumulator sets, residual pre-add, channel tails (720 = 45 k-blocks, 1248 = 78)."""
    meta = zoo_meta(name, 80, S)
    sd = synth_state_dict(meta, seed=2)
    x = _x(3, S, seed=9).to(DEV)
    a = _hip_for(meta, sd, fuse_dws=True)
    b = _hip_for(meta, sd, fuse_dws=False)
    oa, ob = a(x), b(x)
    na = sum(1 for l in a.program.layers if l.dw_k and (l.dw_k ** 2 + 1) * l.cin * 4 > 32 * 1024)
    assert na >= 2 and len(b.program.layers) == len(a.program.layers) + na, (na, len(a.program.layers), len(b.program.layers))
    for u, v in zip(oa, ob):
        assert torch.equal(u, v)


def test_squeeze_excite_gate_is_deterministic_and_batch_invariant():
    """YL_OP_SE: the spatial mean is a fixed-order two-pass sum (no float atomics): the gates, and with them the whole
    forward, are bitwise repeatable, independent of the batch an image is part of and of the chunk split."""
    S = 256
    meta = zoo_meta("yololite_m_v2", 80, S)
    m = _hip_for(meta, synth_state_dict(meta, seed=6))
    ctx = m._ctx_for(S)
    assert any(l.op == _lib.OP_SE for l in m.program.layers) and any(l.scale_slot >= 0 for l in m.program.layers)
    x = _x(9, S, seed=17).to(DEV)
    a = [t.clone() for t in m(x)]
    for streams in (1, 2, 3):
        ctx.set_option("streams", streams)
        for u, v in zip(a, m(x)):
            assert torch.equal(u, v), streams
    ctx.set_option("streams", 2)
    for b in (0, 4, 8):
        for u, v in zip(a, m(x[b:b + 1].contiguous())):
            assert torch.equal(u[b:b + 1], v), b
    # the gates themselves against a float64 reference of timm's SqueezeExcite on the HIP depthwise output
    ctx.set_option("reuse_slots", 0)
    m(x)
    prog = m.program
    k = next(i for i, l in enumerate(prog.layers) if l.op == _lib.OP_SE)
    L = prog.layers[k]
    t = ctx.read_slot(L.in_slot, 9, prog.slots[L.in_slot]).double().cpu()
    g = ctx.read_slot(L.out_slot, 9, prog.slots[L.out_slot]).double().cpu().reshape(9, -1)
    mean = t.mean((1, 2))
    r = mean @ torch.from_numpy(L.w.reshape(L.cout, L.cin)).double().T + torch.from_numpy(L.b).double()
    r = r * torch.sigmoid(r)
    ref = torch.sigmoid(r @ torch.from_numpy(L.w2.reshape(L.cin, L.cout)).double().T + torch.from_numpy(L.b2).double())
    assert float((g - ref).abs().max()) < 2e-6
    ctx.set_option("reuse_slots", 1)
    # the partial sums normally come out of the depthwise launch itself (yl_dw_tile_kernel<.., POOL>); with the tiled
    # depthwise kernel switched off the stand-alone pool pass runs: another partition of the same sum
    ctx.set_option("dev_select", _lib.DEV_DW_TILE_OFF)
    for u, v in zip(a, m(x)):
        assert float((u - v).abs().max()) <= 2e-5 and not torch.equal(u, v)
    ctx.set_option("dev_select", 0)
    for u, v in zip(a, m(x)):
        assert torch.equal(u, v)


def test_forward_decoded_is_forward_plus_decode():
    """SURVEY f2 / VERDICT r03 missing 5: the exported decoded triple (export/export_onnx.py:283-296) as ONE call
    (yl_forward_decoded) -- bitwise the two-call form, all centre / size modes."""
    S, B = 160, 5
    meta = zoo_meta("edge_n", 80, S)
    m = _hip_for(meta, synth_state_dict(meta, seed=2))
    x = _x(B, S, seed=4).to(DEV)
    lv = m(x)
    for cm in ("v8", "simple"):
        for wm in ("softplus", "v8", "exp"):
            exp = m.ctx.decode(lv, cm, wm)
            got = m.forward_decoded(x, cm, wm)
            for k in ("box", "obj", "cls"):
                assert got[k].shape == exp[k].shape and torch.equal(got[k], exp[k]), (cm, wm, k)
    assert got["box"].shape == (B, m.ctx.N, 4) and got["cls"].shape == (B, m.ctx.N, 80)


def test_option_defaults_and_arena_growth():
    """ADVICE r03 lows: get_option reports the LIBRARY's values (defaults 1 for fuse_decode / fuse_head / batch_levels /
    reuse_slots, clamped writes); a single-chunk call after a two-chunk run grows arena 0 only."""
    S = 128
    meta = zoo_meta("edge_n", 80, S)
    m = _hip_for(meta, synth_state_dict(meta, seed=2))
    ctx = m._ctx_for(S)
    assert [ctx.get_option(k) for k in ("fuse_decode", "fuse_head", "batch_levels", "reuse_slots", "streams", "graph",
                                        "winograd", "dev_select")] == [1, 1, 1, 1, 2, 0, 1, 0]
    ctx.set_option("streams", 9)
    assert ctx.get_option("streams") == 4
    ctx.set_option("streams", 2)
    with pytest.raises(_lib.YoloLiteHipError):
        ctx.get_option("no_such_option")
    x = _x(16, S, seed=2).to(DEV)
    a = [t.clone() for t in m(x)]
    two = ctx.activation_bytes()                        # two arenas of 8 images
    ctx.forward(x, timed=True)                          # one chunk of 16: arena 0 grows to 16, arena 1 keeps 8
    one = ctx.activation_bytes()
    assert two < one <= two * 3 // 2 + 4096, (two, one)
    for u, v in zip(a, m(x)):
        assert torch.equal(u, v)
    assert ctx.activation_bytes() == one                # capacity only grows, nothing re-allocated on the way back


@pytest.mark.parametrize("mode,bound", [("mfma_bf16", 3e-2), ("mfma_f16", 4e-3), ("store_f16", 8e-3)])
@pytest.mark.parametrize("name,B,S", [("edge_n", 2, 320), ("yololite_m", 1, 256)])
def test_reduced_precision_mfma_modes(name, B, S, mode, bound):
    """optional reduced-precision modes (operands rounded to bf16 / fp16 in registers, fp32 accumulate,
    fp32 tensors).  "mfma_f16" is the counterpart of the reference's fp16 autocast in evaluate_model
    (scripts/helpers/evaluate.py:399,415).  Not the parity path: the bound is 3e-2 (bf16: 8 mantissa bits; ~60 layers) /
    4e-3 (fp16: 11 bits) of the level's largest logit, and the detections of the fp32 run are reproduced to within a few
    percent.  The two modes are exclusive; switching back restores the fp32 bits."""
    meta = zoo_meta(name, 80, S)
    sd = synth_state_dict(meta, seed={"edge_n": 2, "yololite_m": 10}[name], head_noise=2.0)
    x = _x(B, S)
    with torch.no_grad():
        ref = _oracle_for(meta, sd)(x)
    m = _hip_for(meta, sd)
    ctx = m._ctx_for(S)
    f32 = [o.clone() for o in m(x.to(DEV))]
    _cmp_levels(f32, ref)
    ctx.set_option(mode, 1)
    # ("store_f16", round 6: fp16 operands AND fp16 activation tensors in HBM -- every layer's output is rounded once more
    # on its way to memory, as under torch's autocast; bound 8e-3, measured 3.7e-3 ... 5.9e-3 at full size)
    other = "mfma_f16" if mode == "mfma_bf16" else "mfma_bf16"
    assert ctx.get_option(mode) == 1 and ctx.get_option(other) == 0
    if mode == "store_f16":
        assert ctx.get_option("mfma_f16") == 0
        with pytest.raises(_lib.YoloLiteHipError):
            ctx.read_slot(m.program.layers[3].out_slot, B, tuple(m.program.slots[m.program.layers[3].out_slot]))   # fp16 slot: no fp32 hand-out
    b16 = m(x.to(DEV))
    differs = False
    worst = 0.0
    for o, q, r in zip(b16, f32, ref):
        err = (o.cpu() - r).abs().max().item()
        worst = max(worst, err / r.abs().max().item())
        assert err <= bound * r.abs().max().item() + 1e-3, (name, err, r.abs().max().item())
        differs |= not torch.equal(o, q)
    print(f"[{mode} {name}] max logit error / level max: {worst:.2e}")
    assert differs                                              # the mode really switched kernels
    d16, c16 = ctx.predict(x.to(DEV), _lib.POST_MAIN, 0.1, 0.5, per_class_cap=300, max_out=300)
    c16 = c16.cpu().numpy().copy()
    ctx.set_option(mode, 0)
    d32, c32 = ctx.predict(x.to(DEV), _lib.POST_MAIN, 0.1, 0.5, per_class_cap=300, max_out=300)
    c32 = c32.cpu().numpy()
    assert c32.sum() > 50
    assert np.all(np.abs(c16 - c32) <= 0.1 * np.maximum(c32, 10))
    for o, q in zip(m(x.to(DEV)), f32):                         # and back: bitwise the fp32 path again
        assert torch.equal(o, q)


@pytest.mark.parametrize("name,seg,S,B", [("edge_m", True, 320, 4), ("yololite_m_v2", False, 256, 3), ("edge_n", False, 640, 8)])
def test_fp16_storage_mode_on_the_other_configurations(name, seg, S, B):
    """"store_f16" on the shapes the reduced-precision test above does not cover: a seg model (the prototypes and the detection
    levels stay fp32 tensors: out_f32 layers), an efficientnetv2 model (squeeze-excite gates stay fp32 vectors; the pooled
    depthwise launch feeds them), edge_n at 640 under hipGraph replay and two chunk streams.  Logits within 8e-3 of the fp32
    run's level maximum, detection counts within 10 %, bitwise repeatable, an image's rows independent of the batch around it,
    the arena about half the fp32 one, and the fp32 bits back when the option is cleared.  The hgnetv2 / convnextv2 element-wise
    ops are fp32-storage only: the option is refused at the first forward of such a model."""
    import bench                                                 # the benchmark's workload: a calibrated head that detects
    wl = bench.build_workload(name, S, B, seed=1, seg=seg, dev=DEV)
    m, ctx, x = wl["model"], wl["ctx"], wl["x"]
    def lv():
        o = m(x)
        return [t.clone() for t in (o[0] if seg else o)]
    f32 = lv()
    d32, c32 = ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=1024)
    c32 = c32.cpu().numpy().copy()
    ctx.set_option("store_f16", 1)
    ctx.set_option("graph", 1); ctx.set_option("streams", 2)
    h = lv()
    for o, r in zip(h, f32):
        assert float((o - r).abs().max()) <= 8e-3 * float(r.abs().max()) + 1e-3
        assert not torch.equal(o, r)
    for o, r in zip(lv(), h):
        assert torch.equal(o, r)                                 # repeatable
    one = m(x[1:2]); one = one[0] if seg else one
    for o, r in zip(one, h):
        assert torch.equal(o, r[1:2])                            # batch-invariant
    d16, c16 = ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=1024)
    c16 = c16.cpu().numpy()
    assert c32.sum() > 20 and np.all(np.abs(c16 - c32) <= 0.1 * np.maximum(c32, 10)), (c16, c32)
    ctx.set_option("store_f16", 0); ctx.set_option("graph", 0); ctx.set_option("streams", 1)
    for o, r in zip(lv(), f32):
        assert torch.equal(o, r)


def test_fp16_storage_mode_refuses_the_fp32_only_ops():
    meta = make_meta(img_size=96, **TINY[len(TINY) - 1])         # the convnextv2 test vehicle: LayerNorm / GRN / GELU passes
    m = _hip_for(meta, synth_state_dict(meta, seed=4))
    ctx = m._ctx_for(96)
    ctx.set_option("store_f16", 1)
    with pytest.raises(_lib.YoloLiteHipError):
        m(_x(2, 96, seed=1).to(DEV))
    ctx.set_option("store_f16", 0)
    m(_x(2, 96, seed=1).to(DEV))


@pytest.mark.parametrize("name,S,B", [("yololite_m", 224, 3), ("yololite_n", 352, 2), ("yololite_xl", 160, 2), ("yololite_m", 640, 5)])
def test_fused_efficientnet_lite_entry_equals_the_two_launches(name, S, B):
    """yl_stemdw_kernel (round 6): conv_stem (TF-SAME, 3 -> 32) -> blocks.0.0 (depthwise 3x3 + 1x1, 32 -> 16 / 24) of the
    tf_efficientnet_lite backbones as ONE launch against the same model built with fuse_stem=False (plain stem kernel, then the
    depthwise -> 1x1 kernel).  The stem GEMM sums its 27 products in another order (bias in the K pad slot), so the comparison
    is a tolerance, not bitwise: every level within 2e-5 of its largest logit (and both within the oracle's bar, as the zoo sweep
    checks).  224 / 352 / 160: grids of 14 / 22 / 10 tiles with every border case; lite4's 24-channel 1x1 (two n-tiles);
    bitwise repeatable and batch-invariant."""
    meta = zoo_meta(name, 80, S)
    sd = synth_state_dict(meta, seed=7)
    x = _x(B, S, seed=8)
    fused = _hip_for(meta, sd)
    from yololite_amd.model import YOLOLiteHIP
    plain = YOLOLiteHIP(meta, fuse_stem=False)
    plain.load_state_dict(sd); plain.to(DEV)
    assert fused.program.layers[0].op == _lib.OP_STEMBLOCK and fused.program.layers[0].dw_k == 3
    assert plain.program.layers[0].op == _lib.OP_STEM and len(plain.program.layers) == len(fused.program.layers) + 1
    a = [t.clone() for t in fused(x.to(DEV))]
    b = plain(x.to(DEV))
    for u, v in zip(a, b):
        assert float((u - v).abs().max()) <= 2e-5 * float(v.abs().max()) + 1e-6
    for u, v in zip(fused(x.to(DEV)), a):
        assert torch.equal(u, v)
    for u, v in zip(fused(x[1:2].to(DEV)), a):
        assert torch.equal(u, v[1:2])


def test_lanes_and_chunk_graphs_are_bitwise_the_plain_path():
    """side-stream lane for the coarse-level neck/head layers + one hipGraph per batch chunk: same bits as the
    single-stream eager path (scheduling must not change results)."""
    meta = zoo_meta("edge_n", 80, 320)
    sd = synth_state_dict(meta, seed=2, head_noise=2.0)
    m = _hip_for(meta, sd)
    ctx = m._ctx_for(320)
    x = _x(16, 320).to(DEV)
    ctx.set_option("graph", 0); ctx.set_option("streams", 1); ctx.set_option("lanes", 0)
    d0, c0 = ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=300)
    d0, c0 = d0.clone(), c0.clone()
    assert int(c0.sum()) > 100                      # the comparison below must not be about empty results
    ctx.set_option("batch_levels", 0)
    d0b, c0b = ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=300)
    assert torch.equal(c0b, c0)
    for graph in (0, 1):
        for streams in (1, 2, 3):
            for lanes, batch in ((0, 0), (1, 0), (0, 1)):
                ctx.set_option("graph", graph); ctx.set_option("streams", streams); ctx.set_option("lanes", lanes)
                ctx.set_option("batch_levels", batch)
                for _ in range(2):
                    d, c = ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=300)
                    assert torch.equal(c, c0), (graph, streams, lanes, batch)
                    for b in range(16):
                        k = int(c0[b])
                        assert torch.equal(d[b, :k], d0[b, :k]), (graph, streams, lanes, batch, b)
    ctx.set_option("lanes", 0); ctx.set_option("batch_levels", 1)


@pytest.mark.parametrize("name,B,S", [("edge_n", 3, 320), ("edge_n", 2, 640), ("edge_n", 3, 384), ("edge_n", 5, 128),
                                      ("edge_m", 2, 320), ("yololite_m", 1, 256), ("yololite_m_v2", 2, 256)])
def test_convc_kernels_are_bitwise_the_kernels_they_replace(name, B, S):
    """Alternative kernels of yl_convc.hip sum every output's k blocks in the same order as the kernels they replace
    -> identical bits.  "tile_m" 6: wave-autonomous 1x1 / depthwise kernels and the streamed dense 3x3 kernel OFF; 7:
    producer / consumer depthwise -> 1x1 kernel (opt-in) ON, with "dev_select" bit 3 on every layer shape it supports.
    Exception: yl_conv_kxk_kernel (yololite_m's dense 3x3) walks K channel-block-major instead of tap-major (cache
    locality), a different fp32 summation order of the same 2952 products: compared at rounding-noise tolerance.
    "tile_m" 6 also turns off yl_conv_s2c_kernel (round 3: blocks.1.0 3x3 s2 + chained 1x1 from an LDS-staged patch; taken
    where the output width is a multiple of 8: 640, 384, 128 here -- image borders included)."""
    meta = zoo_meta(name, 80, S)
    sd = synth_state_dict(meta, seed=4)
    m = _hip_for(meta, sd)
    ctx = m._ctx_for(S)
    ctx.set_option("winograd", 0)                       # kernel-equivalence test: the direct convolution on both sides
    x = _x(B, S, seed=11).to(DEV)
    a = [t.clone() for t in m(x)]
    ctx.set_option("split_k", 1)                        # latency option (round 4): split-K depthwise -> 1x1 on the <= 20x20 grids
    split = [t.clone() for t in m(x)]
    ctx.set_option("split_k", 0)
    for u, v in zip(split, a):                          # another summation order of the same products: rounding noise only
        assert torch.allclose(u, v, atol=2e-5, rtol=1e-5), float((u - v).abs().max())
    if name == "edge_n" and S in (640, 384):            # 20x20 / 12x12 grids in the last backbone stage: the option bites
        assert any(not torch.equal(u, v) for u, v in zip(split, a))
    for hint in (6, 7):
        ctx.set_option("tile_m", hint)
        ctx.set_option("dev_select", _lib.DEV_DWC_ALL)   # this context only: the opt-in kernel on every shape it supports
        b = m(x)
        ctx.set_option("tile_m", 0)
        ctx.set_option("dev_select", 0)
        for u, v in zip(a, b):
            if name.startswith("yololite_m") and hint == 6:
                assert torch.allclose(u, v, atol=2e-5, rtol=1e-5), (hint, float((u - v).abs().max()))
            else:
                assert torch.equal(u, v), hint


@pytest.mark.parametrize("name,B,S", [("yololite_m", 2, 256), ("yololite_m", 1, 224)])
def test_winograd_option_matches_direct_convolution(name, B, S):
    """Option "winograd": the dense 3x3 stride-1 FPN convs (>= 64 channels) as Winograd F(2x2,3x3).  Not bit-identical
    (the transforms round differently); the raw head logits must stay within the oracle bound of the direct path
    (_cmp_levels: 2e-4 abs / decoded scores 1e-4) and within 1e-4 of the direct HIP result.  224: odd level grids
    (7x7: a partial last Winograd tile row / column).  Mode 1 (every eligible layer, the library default since round 4:
    measured score error vs the oracle equal to the direct path's, profiles/r04_winograd_margin*.json) and mode 2
    (the finest level's layers only)."""
    meta = zoo_meta(name, 80, S)
    sd = synth_state_dict(meta, seed=5)
    m = _hip_for(meta, sd)
    ctx = m._ctx_for(S)
    x = _x(B, S, seed=3)
    ctx.set_option("winograd", 0)
    direct = [t.clone() for t in m(x.to(DEV))]
    with torch.no_grad():
        ref = _oracle_for(meta, sd)(x)
    for mode in (1, 2):                     # 1 (the default): all six FPN convs; 2: the two on the finest level only
        ctx.set_option("winograd", mode)
        wino = [t.clone() for t in m(x.to(DEV))]
        ctx.set_option("winograd", 0)
        again = m(x.to(DEV))
        ndiff = 0
        for u, v, w in zip(direct, wino, again):
            assert torch.equal(u, w)                                   # the switch is clean
            ndiff += int(not torch.equal(u, v))                        # ... and really selects another kernel
            assert float((u - v).abs().max()) <= 1e-4, float((u - v).abs().max())
        assert ndiff == (3 if mode == 1 else 1), (mode, ndiff)         # selective: only the finest level's head sees it
        _cmp_levels(wino, ref, C=80)
    ctx.set_option("winograd", 1)


@pytest.mark.parametrize("name,B,S,seg", [("yololite_m", 3, 256, 0), ("yololite_m", 1, 224, 0), ("yololite_m", 2, 640, 0),
                                          ("edge_m", 2, 320, 1), ("yololite_m_v2", 2, 256, 0)])
def test_winograd_position_split_kernel_is_bitwise_the_first_form(name, B, S, seg):
    """yl_conv_wino2_kernel (round 5: the 16 transform positions dealt to the waves, input window through LDS once per item)
    evaluates the same transform expressions, k order and epilogue as yl_conv_wino_kernel -> identical bits, for every item
    shape ("dev_select" bits 12-13) against the first form ("dev_select" bit 11).  224: odd level grids (partial m-tiles);
    640: the shapes the benchmark runs (4 m-tiles x 3 n-tiles at 80x80); edge_m + seg: the prototype branch's conv on a
    nearest-upsampled input (in_shift 1) with SiLU."""
    meta = zoo_meta(name, 80, S, **(dict(seg=True) if seg else {}))
    sd = synth_state_dict(meta, seed=6)
    m = _hip_for(meta, sd)
    ctx = m._ctx_for(S)
    x = _x(B, S, seed=13).to(DEV)
    ctx.set_option("winograd", 1)
    ctx.set_option("dev_select", _lib.DEV_WINO_V1)
    def run():
        out = m(x)
        return ([t.clone() for t in out[0]] + [out[1].clone()]) if seg else [t.clone() for t in out]   # (+ mask prototypes)
    first = run()
    ctx.set_option("winograd", 0)
    direct = run()
    ctx.set_option("winograd", 1)
    assert any(not torch.equal(u, v) for u, v in zip(first, direct))          # Winograd layers exist in this model
    for shape in (0, 1, 2, 3):
        ctx.set_option("dev_select", shape << _lib.DEV_WINO_SHAPE_SHIFT)
        got = run()
        for u, v in zip(first, got):
            assert torch.equal(u, v), (shape, float((u - v).abs().max()))
    ctx.set_option("dev_select", 0)


def test_small_channel_3x3_window_kernel_is_bitwise_the_direct_kernel():
    """yl_conv_k3w_kernel (round 6: dense 3x3, 16 / 32 -> <= 16 channels on grids of >= 160 x 160 pixels -- efficientnetv2's first
    fused-MBConv blocks at 320 x 320 -- one wave per 4 x 4-pixel tile, 6 x 6 window of every 16-channel block in wave-private LDS, all
    weight fragments in registers) keeps the k order (tap-major, k-blocks inside), the pre-add rule and the epilogues of
    yl_conv_mfma_kernel -> identical bits to the direct path ("winograd" 0) with the kernel switched off ("dev_select" bit 17).  It runs
    only where the direct kernel would (under the default options those layers stay with Winograd: same results as before), selected
    by shape and not by batch (B = 1 and B = 3 take it alike)."""
    meta = zoo_meta("yololite_m_v2", 80, 640)
    sd = synth_state_dict(meta, seed=9)
    m = _hip_for(meta, sd)
    ctx = m._ctx_for(640)
    x = _x(3, 640, seed=21)
    xd = x.to(DEV)
    def run(wino, dev, xin=None):
        ctx.set_option("winograd", wino)
        ctx.set_option("dev_select", dev)
        return [t.clone() for t in m(xd if xin is None else xin)]
    direct_off, direct_on = run(0, _lib.DEV_K3W_OFF), run(0, 0)
    for u, v in zip(direct_off, direct_on):
        assert torch.equal(u, v)
    one = run(0, 0, xd[1:2])                                                 # batch invariance with the kernel on
    for u, v in zip(direct_on, one):
        assert torch.equal(u[1:2], v)
    wino_off, wino_on = run(1, _lib.DEV_K3W_OFF), run(1, 0)
    for u, v in zip(wino_off, wino_on):
        assert torch.equal(u, v)                                             # the default options do not see the kernel
    assert any(not torch.equal(u, v) for u, v in zip(direct_on, wino_on))
    ctx.set_option("dev_select", 0)
    with torch.no_grad():
        ref = _oracle_for(meta, sd)(x)
    _cmp_levels(direct_on, ref, C=80)
    ctx.set_option("winograd", 1)


@pytest.mark.parametrize("name,B,S", [("edge_m", 2, 320), ("edge_m", 3, 224), ("yololite_m", 2, 256), ("edge_l", 1, 320),
                                      ("edge_m", 12, 640)])
def test_window_in_lds_depthwise_kernel_is_bitwise_the_tap_load_kernel(name, B, S):
    """yl_conv_dwl_kernel (round 5: depthwise 3x3 -> wide 1x1 with the 10 x 10-pixel input windows of two 8 x 8-pixel output
    windows in LDS, 1x1 weights triple-buffered, one barrier per k-block) runs the fmaf chain, k order and epilogues of
    yl_conv_dwk_kernel (nine taps per lane from L1/L2) -> identical bits.  "dev_select" bit 15: on every grid (224: 28 x 28 /
    14 x 14 / 7 x 7 levels = partial windows; few items), bit 14: off.  edge_m B = 12 at 640: the default selection takes it
    on the 80 x 80 level (>= 4 windows per CU)."""
    meta = zoo_meta(name, 80, S)
    sd = synth_state_dict(meta, seed=8)
    m = _hip_for(meta, sd)
    ctx = m._ctx_for(S)
    x = _x(B, S, seed=17).to(DEV)
    ctx.set_option("winograd", 0)
    ctx.set_option("dev_select", _lib.DEV_DWL_OFF)
    old = [t.clone() for t in m(x)]
    ctx.set_option("dev_select", _lib.DEV_DWL_ALL if B < 12 else 0)
    new = m(x)
    ctx.set_option("dev_select", 0)
    ctx.set_option("winograd", 1)
    for u, v in zip(old, new):
        assert torch.equal(u, v), float((u - v).abs().max())


def test_tiled_depthwise_kernel_is_bitwise_the_per_output_kernel():
    """yl_dw_tile_kernel (register-tiled stand-alone depthwise, yololite_m's backbone) accumulates every output's taps
    in the (dy, dx) order of yl_dw_kernel -> identical bits.  The switch is a per-context developer option
    ("dev_select" bit 0; it used to be a process-wide environment variable read inside the launcher)."""
    n = 0
    for S, B in ((256, 2), (224, 1)):
        meta = zoo_meta("yololite_m", 80, S)
        m = _hip_for(meta, synth_state_dict(meta, seed=4))
        ctx = m._ctx_for(S)
        x = _x(B, S, seed=11).to(DEV)
        a = [t.clone() for t in m(x)]
        ctx.set_option("dev_select", _lib.DEV_DW_TILE_OFF)
        assert ctx.get_option("dev_select") == _lib.DEV_DW_TILE_OFF
        b = [t.clone() for t in m(x)]
        ctx.set_option("dev_select", 0)
        for u, v in zip(a, b):
            assert torch.isfinite(u).all() and torch.equal(u, v)
            n += 1
    assert n >= 6


def test_forward_batch_invariance_and_determinism_full_size():
    """BASELINE config 2 (edge_n 640x640 B=64): bitwise repeatable, and image i of the batch equals the
    same image run alone (size-independent property; the oracle is too slow at this size)."""
    meta = zoo_meta("edge_n", 80, 640)
    sd = synth_state_dict(meta, seed=0)
    m = _hip_for(meta, sd)
    x = _x(64, 640).to(DEV)
    a = m(x)
    b = m(x)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    for i in (0, 37, 63):
        one = m(x[i:i + 1])
        for u, v in zip(a, one):
            assert torch.equal(u[i:i + 1], v)


# ------------------------------------------------------------------------------------------ decode
def test_decode_matches_oracle(golden_dir):
    z = np.load(os.path.join(golden_dir, "decode.npz"))
    lv = [torch.from_numpy(z[f"rand/level{j}"]) for j in range(3)]
    for cm in ("v8", "simple"):
        for wm in ("softplus", "v8", "exp"):
            d = ya.decode_preds_anchorfree([t.to(DEV) for t in lv], 128, cm, wm)
            np.testing.assert_allclose(d["box"].cpu().numpy(), z[f"rand/{cm}_{wm}/box"], rtol=2e-6, atol=2e-5)
            np.testing.assert_array_equal(d["obj"].cpu().numpy(), z[f"rand/{cm}_{wm}/obj"])
            np.testing.assert_array_equal(d["cls"].cpu().numpy(), z[f"rand/{cm}_{wm}/cls"])
    lv2 = [torch.from_numpy(z[f"a2/level{j}"]).to(DEV) for j in range(2)]
    d = ya.decode_preds_anchorfree(lv2, 96)
    np.testing.assert_allclose(d["box"].cpu().numpy(), z["a2/box"], rtol=2e-6, atol=2e-5)
    zero = [torch.zeros(1, 1, s, s, 8, device=DEV) for s in (80, 40, 20)]
    d = ya.decode_preds_anchorfree(zero, 640)
    np.testing.assert_allclose(d["box"][0, 0].cpu().numpy(), [1.2274113, 1.2274113, 6.7725887, 6.7725887], atol=1e-6)


# ------------------------------------------------------------------------------------------ NMS
@pytest.mark.parametrize("n", [1, 63, 64, 65, 500, 3000, 9000, 20000])
@pytest.mark.parametrize("impl", ["torchvision", "fallback"])
def test_nms_bit_exact(n, impl):
    """identical boxes/scores in -> identical kept indices out (integer/index work: bit-exact),
    including score ties, duplicate boxes, zero-area boxes, n beyond the LDS key capacity."""
    rng = np.random.RandomState(n)
    ctr = rng.rand(n, 2).astype(np.float32) * 300
    wh = rng.rand(n, 2).astype(np.float32) * 60 + 1
    bx = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
    sc = rng.rand(n).astype(np.float32)
    if n >= 64:
        sc[5] = sc[17] = sc[40]                 # score ties -> index order decides
        bx[9] = bx[3]                           # duplicates (IoU == 1)
        bx[11, 2:] = bx[11, :2]                 # zero-area box (NaN IoU with itself-like boxes)
    for thr, cap in ((0.5, 300), (0.3, 10 ** 6), (0.9, 50)):
        exp = opost.nms(bx, sc, thr, cap, impl)
        got = ya.nms(torch.from_numpy(bx).to(DEV), torch.from_numpy(sc).to(DEV), thr, cap, impl).cpu().numpy()
        np.testing.assert_array_equal(got, exp)


# ------------------------------------------------------------------------------------------ pipelines
def _match(got_b, got_s, got_c, exp_b, exp_s, exp_c):
    assert got_c.tolist() == exp_c.tolist()
    np.testing.assert_allclose(got_s, exp_s, rtol=0, atol=1e-5)
    np.testing.assert_array_equal(np.rint(got_b), np.rint(exp_b))
    np.testing.assert_allclose(got_b, exp_b, rtol=0, atol=1e-3)


def _pipe_cases(golden_dir):
    with open(os.path.join(golden_dir, "pipelines_cases.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("idx", range(6))
def test_pipelines_on_golden_levels(golden_dir, idx):
    """main / eval / fallback pipelines on the level tensors of the golden fixtures.  eval and fallback
    expectations are the REFERENCE's own outputs; main is the oracle's."""
    z = np.load(os.path.join(golden_dir, "pipelines.npz"))
    c = _pipe_cases(golden_dir)[idx]
    lv = [torch.from_numpy(z[f"{c['tag']}/level{j}"]) for j in range(3)]
    dl = [t.to(DEV) for t in lv]
    exp = opost.pipeline_main(lv, c["img"], c["conf"], c["iou"], 300)
    got = ya.infer_main_postprocess(dl, c["img"], c["conf"], c["iou"], 300)
    for b in range(c["B"]):
        _match(got["boxes"][b], got["scores"][b], got["classes"][b], exp["boxes"][b], exp["scores"][b], exp["classes"][b])
    dets = ya._decode_batch_to_coco_dets(dl, c["img"], conf_th=c["conf"], iou_th=c["iou"], add_one=True)
    for b, d in enumerate(dets):
        rb, rs, rc = (z[f"{c['tag']}/eval/{b}/{k}"] for k in ("bbox", "score", "cat"))
        assert [x["category_id"] for x in d] == rc.tolist()
        np.testing.assert_allclose([x["score"] for x in d], rs, atol=1e-5)
        np.testing.assert_allclose(np.asarray([x["bbox"] for x in d]).reshape(-1, 4), rb, atol=1e-3)
    fb = ya.decode_anchorfree_like_train(dl, c["img"], conf_th=c["conf"], iou_th=c["iou"], topk=300, nms_impl="greedy")
    for b in range(c["B"]):
        rb, rs, rc = (z[f"{c['tag']}/fallback/{b}/{k}"] for k in ("boxes", "scores", "classes"))
        assert fb["boxes"][b].shape == rb.shape
        if c["tag"] == "c3" and b == 0:        # deliberate score ties: order among equal scores unspecified
            assert sorted(fb["classes"][b].cpu().tolist()) == sorted(rc.tolist())
            np.testing.assert_allclose(np.sort(fb["scores"][b].cpu().numpy()), np.sort(rs), atol=1e-5)
        else:
            _match(fb["boxes"][b].cpu().numpy(), fb["scores"][b].cpu().numpy(), fb["classes"][b].cpu().numpy(), rb, rs, rc)


@pytest.mark.parametrize("mode", ["main", "eval", "fallback_topk", "fallback_tv"])
def test_pipelines_full_size_vs_oracle(mode):
    """N = 8400 candidates, C = 80, raw head ~ N(0,2) (SURVEY 8d stress input): hundreds of survivors at
    conf 0.4, thousands at 0.001."""
    g = torch.Generator().manual_seed(99)
    lv = [torch.randn(2, 1, s, s, 85, generator=g) * 2.0 for s in (80, 40, 20)]
    dl = [t.to(DEV) for t in lv]
    if mode == "main":
        exp = opost.pipeline_main(lv, 640, 0.4, 0.5, 300)
        got = ya.infer_main_postprocess(dl, 640, 0.4, 0.5, 300)
    elif mode == "eval":
        _, raw = opost.pipeline_eval(lv, 640, 0.001, 0.65)
        exp = {"boxes": [r[0] for r in raw], "scores": [r[1] for r in raw], "classes": [r[2] for r in raw]}
        ctx = ya.postprocess.context_for(dl, 640)
        dets, counts = ctx.postprocess(dl, _lib.POST_EVAL, 0.001, 0.65, per_class_cap=0, max_out=ctx.N)
        rows = [dets[b, :int(counts[b])].cpu().numpy() for b in range(2)]
        got = {"boxes": [r[:, :4] for r in rows], "scores": [r[:, 4] for r in rows],
               "classes": [r[:, 5].astype(np.int64) for r in rows]}
    else:
        impl = "greedy" if mode == "fallback_topk" else "torchvision"       # nms() without / with torchvision
        exp = opost.pipeline_fallback(lv, 640, 0.3, 0.6, topk=100, nms_impl="fallback" if impl == "greedy" else impl)
        fb = ya.decode_anchorfree_like_train(dl, 640, 0.3, 0.6, topk=100, nms_impl=impl)
        got = {k: [t.cpu().numpy() for t in fb[k]] for k in fb}
    for b in range(2):
        assert len(exp["scores"][b]) > 50
        _match(got["boxes"][b], got["scores"][b], got["classes"][b], exp["boxes"][b], exp["scores"][b], exp["classes"][b])


def test_infer_main_flow_against_reference_json(golden_dir):
    """tools/infer.py main() end to end (reference JSON in tests/golden/infer_main.npz): checkpoint file ->
    load_model_names_imgsize_from_ckpt -> preprocess -> HIP forward + decode + NMS + back-map."""
    import tempfile
    z = np.load(os.path.join(golden_dir, "infer_main.npz"))
    with open(os.path.join(golden_dir, "infer_main_meta.json")) as f:
        meta = json.load(f)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    with tempfile.TemporaryDirectory() as td:
        ck = os.path.join(td, "tiny.pt")
        torch.save({"state_dict": sd, "meta": meta}, ck)
        model, names, img_size = ya.load_model_names_imgsize_from_ckpt(ck, torch.device(DEV))
        with pytest.raises(RuntimeError):
            torch.save({"weights": sd}, ck)
            ya.load_model_names_imgsize_from_ckpt(ck, torch.device(DEV))
    assert names == ["a", "b", "c"] and img_size == 96
    for name in ("sq", "wide"):
        x, bm = ya.preprocess_batch(model._ctx_for(img_size), [z[f"img_{name}"]])     # the product's one pre-processing path
        outs = model(x)
        got = ya.infer_main_postprocess(outs, img_size, 0.4, 0.5, backmap=[tuple(bm[0])])
        assert got["classes"][0].tolist() == z[f"{name}/class_id"].tolist()
        np.testing.assert_allclose(got["scores"][0], z[f"{name}/score"], atol=1e-4)
        np.testing.assert_array_equal(np.rint(got["boxes"][0]), np.rint(z[f"{name}/bbox_xyxy"]))


def test_pip_api_predict_speed_split(tmp_path, golden_dir):
    """YoloLite(path).predict(): dict surface of the pip package (README.md:20-42) with the pre / infer / post split
    measured by HIP events (yl_last_timing); the event-split run and the overlapped run return the same detections."""
    from yololite_amd.api import YoloLite
    z = np.load(os.path.join(golden_dir, "infer_main.npz"))
    with open(os.path.join(golden_dir, "infer_main_meta.json")) as f:
        meta = json.load(f)
    ck = str(tmp_path / "tiny.pt")
    torch.save({"state_dict": {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}, "meta": meta}, ck)
    yl = YoloLite(ck, device=DEV)
    imgs = [z["img_sq"], z["img_wide"]] * 4
    r1 = yl.predict(imgs)
    r2 = yl.predict(imgs, profile=False)
    for a, b, name in zip(r1, r2, ["sq", "wide"] * 4):
        assert a["classes"].tolist() == z[f"{name}/class_id"].tolist() == b["classes"].tolist()
        np.testing.assert_array_equal(a["boxes"], b["boxes"])
        assert a["masks"] is None
        sp = a["speed"]
        assert set(sp) == {"pre_ms", "infer_ms", "post_ms", "total_ms"} and all(v > 0 for v in sp.values())
        assert abs(sp["total_ms"] - (sp["pre_ms"] + sp["infer_ms"] + sp["post_ms"])) < 1e-9
        assert set(b["speed"]) == {"pre_ms", "infer_post_ms", "total_ms"}


def test_predict_fused_equals_forward_plus_postprocess_and_graph():
    meta = zoo_meta("edge_n", 80, 320)
    sd = synth_state_dict(meta, seed=1, head_noise=2.0)
    m = _hip_for(meta, sd)
    x = _x(4, 320).to(DEV)
    outs = m(x)
    ctx = m._ctx_for(320)
    d1, c1 = ctx.postprocess(outs, _lib.POST_MAIN, 0.02, 0.5, 300)
    d2, c2 = ctx.predict(x, _lib.POST_MAIN, 0.02, 0.5, 300)
    assert torch.equal(c1, c2) and int(c1.min()) > 0
    for b in range(4):
        assert torch.equal(d1[b, :int(c1[b])], d2[b, :int(c2[b])])
    ctx.set_option("graph", 1)                      # hipGraph replay of the same launch list
    for _ in range(2):
        d3, c3 = ctx.predict(x, _lib.POST_MAIN, 0.02, 0.5, 300)
        assert torch.equal(c1, c3)
        for b in range(4):
            assert torch.equal(d1[b, :int(c1[b])], d3[b, :int(c3[b])])
    ctx.set_option("graph", 0)


@pytest.mark.parametrize("lanes,streams", [(2, 1), (3, 1), (2, 2)])
def test_serving_pipeline_and_cloned_contexts_are_bitwise_the_plain_calls(lanes, streams):
    """serving.ServingPipeline (K batches in flight on K contexts, yl_clone: shared weights) hands back, for every batch,
    exactly the rows a plain ctx.predict of that batch produces -- 7 batches of different images through 2 / 3 lanes, the
    hand-back order is the submission order, and destroying the ORIGINAL context first leaves the clones usable (the last
    owner frees the weights)."""
    from yololite_amd.serving import ServingPipeline
    S, B = 320, 6
    meta = zoo_meta("edge_n", 80, S)
    sd = synth_state_dict(meta, seed=1, head_noise=2.0)
    m = _hip_for(meta, sd)
    xs = [_x(B, S, seed=40 + i).to(DEV) for i in range(7)]
    plain = m._ctx_for(S)
    want = []
    for x in xs:
        d, c = plain.predict(x, _lib.POST_MAIN, 0.02, 0.5, 300)
        want.append((d.cpu(), c.cpu()))
    assert min(int(c.min()) for _, c in want) > 0
    pipe = ServingPipeline(plain, lanes=lanes, streams_per_lane=streams, graph=True)
    assert len({int(c.handle.value) for c in pipe.ctxs}) == lanes
    got = []
    for x in xs:
        r = pipe.submit(x, _lib.POST_MAIN, 0.02, 0.5, 300)
        if r is not None:
            got.append((r[0].cpu(), r[1].cpu()))
    assert len(got) == len(xs) - lanes
    got += [(d.cpu(), c.cpu()) for d, c in pipe.flush()]
    assert len(got) == len(xs) and pipe.flush() == []
    for (d0, c0), (d1, c1) in zip(want, got):
        assert torch.equal(c0, c1)
        for b in range(B):
            assert torch.equal(d0[b, :int(c0[b])], d1[b, :int(c1[b])])
    # ownership: drop the model (and with it the original context); a clone still predicts the same rows
    clone = pipe.ctxs[1]
    del pipe, plain
    m._ctxs.clear(); m.ctx = None
    del m
    import gc
    gc.collect()
    d, c = clone.predict(xs[0], _lib.POST_MAIN, 0.02, 0.5, 300)
    assert torch.equal(c.cpu(), want[0][1]) and torch.equal(d.cpu()[0, :int(c[0])], want[0][0][0, :int(c[0])])


@pytest.mark.parametrize("name,seg", [("edge_n", False), ("edge_m", True), ("yololite_m", False)])
def test_liveness_slot_reuse_is_bitwise_and_smaller(name, seg):
    """activation tensors placed by liveness in one arena per batch chunk (default) vs one buffer per tensor:
    same bits (levels, detections, prototypes, masks), a fraction of the memory."""
    from yololite_amd.program import MODEL_ZOO
    S, B = 256, 16
    meta = make_meta(num_classes=80, img_size=S, seg=seg, **MODEL_ZOO[name])
    sd = synth_state_dict(meta, seed={"edge_n": 2, "edge_m": 9, "yololite_m": 10}[name], head_noise=2.0)
    m = _hip_for(meta, sd)
    ctx = m._ctx_for(S)
    x = _x(B, S, seed=2).to(DEV)
    res, mem = {}, {}
    for reuse in (1, 0):
        ctx.set_option("reuse_slots", reuse)
        for streams in (1, 2):
            ctx.set_option("streams", streams)
            out = m(x)
            lv = [t.clone() for t in (out[0] if seg else out)]
            pr = out[1].clone() if seg else None
            d, c, idx = ctx.predict(x, _lib.POST_MAIN, 0.01, 0.5, per_class_cap=300, max_out=512, want_idx=True)
            mk = ctx.masks(c, idx, 512).clone() if seg else None
            res[(reuse, streams)] = (lv, pr, d.clone(), c.clone(), mk)
            mem[(reuse, streams)] = ctx.activation_bytes()
    ref = res[(0, 1)]
    assert int(ref[3].sum()) > 0
    for k, v in res.items():
        for a, b in zip(v[0], ref[0]):
            assert torch.equal(a, b), k
        assert torch.equal(v[3], ref[3]), k
        for b in range(B):
            n = min(int(ref[3][b]), 512)
            assert torch.equal(v[2][b, :n], ref[2][b, :n]), k
        if seg:
            assert torch.equal(v[1], ref[1]) and torch.equal(v[4], ref[4]), k
    # (streams = 2 after streams = 1: the chunk arenas keep the capacity of the 16-image chunk -- buffers only grow)
    assert mem[(1, 1)] * 3 < mem[(0, 1)] and mem[(1, 2)] * 2 < mem[(0, 2)], mem
    ctx.set_option("reuse_slots", 1); ctx.set_option("streams", 2)


def test_cached_graph_survives_post_workspace_growth():
    """forward(B=16) sizes the activations only; predict(B=4) captures a graph with the post workspaces of a
    4-image batch baked in; predict(B=16) reallocates those workspaces WITHOUT growing the activations; the next
    predict(B=4) must not replay the stale graph (it did: use-after-free of the freed workspaces)."""
    meta = zoo_meta("edge_n", 80, 320)
    sd = synth_state_dict(meta, seed=1, head_noise=2.0)
    m = _hip_for(meta, sd)
    ctx = m._ctx_for(320)
    x = _x(16, 320).to(DEV)
    ctx.set_option("graph", 0)
    ref, cref = ctx.predict(x[:4], _lib.POST_MAIN, 0.02, 0.5, 300)
    ref, cref = ref.clone(), cref.clone()
    m2 = _hip_for(meta, sd)
    c2 = m2._ctx_for(320)
    c2.set_option("graph", 1)
    c2.forward(x)
    d, c = c2.predict(x[:4], _lib.POST_MAIN, 0.02, 0.5, 300)
    assert torch.equal(c, cref)
    junk = c2.predict(x, _lib.POST_MAIN, 0.02, 0.5, 300)
    filler = [torch.full((1 << 20,), float("nan"), device=DEV) for _ in range(8)]     # reuse the freed blocks
    for _ in range(2):
        d, c = c2.predict(x[:4], _lib.POST_MAIN, 0.02, 0.5, 300)
        assert torch.equal(c, cref) and int(c.min()) > 0
        for b in range(4):
            assert torch.equal(d[b, :int(c[b])], ref[b, :int(cref[b])])
    del junk, filler


@pytest.mark.parametrize("idx", range(len(TINY)))
def test_fused_decode_epilogue_equals_decode_kernel(idx):
    """yl_predict decodes inside the head-output conv (no raw level tensor); yl_postprocess runs the decode
    kernel on the levels yl_forward wrote.  Same arithmetic on the same logits: bitwise equal detections for
    every post mode, C == 1, A == 2, P2/P6 levels and the centre/size variants."""
    meta = make_meta(img_size=96, **TINY[idx])
    sd = synth_state_dict(meta, seed=20 + idx, head_noise=2.0)
    m = _hip_for(meta, sd)
    ctx = m._ctx_for(96)
    x = _x(5, 96, seed=idx).to(DEV)
    outs = [o.clone() for o in m(x)]
    cases = [(_lib.POST_MAIN, 0.05, 0.5, 300, 0, "v8", "softplus"), (_lib.POST_FALLBACK, 0.05, 0.45, 300, 50, "v8", "softplus"),
             (_lib.POST_EVAL, 0.001, 0.65, 0, 0, "v8", "softplus"), (_lib.POST_MAIN, 0.05, 0.5, 300, 0, "simple", "v8"),
             (_lib.POST_MAIN, 0.05, 0.5, 300, 0, "v8", "exp")]
    total = 0
    for mode, conf, iou, cap, topk, cm, wm in cases:
        d1, c1 = ctx.postprocess(outs, mode, conf, iou, cap, topk, center_mode=cm, wh_mode=wm)
        for fuse in (1, 0):
            ctx.set_option("fuse_decode", fuse)
            d2, c2 = ctx.predict(x, mode, conf, iou, cap, topk, center_mode=cm, wh_mode=wm)
            assert torch.equal(c1, c2), (mode, cm, wm, fuse)
            for b in range(5):
                k = min(int(c1[b]), d1.shape[1])
                assert torch.equal(d1[b, :k], d2[b, :k]), (mode, cm, wm, fuse, b)
        ctx.set_option("fuse_decode", 1)
        total += int(c1.sum())
    assert total > 0


def test_cli_infer_and_evaluate(tmp_path, golden_dir):
    """tools/infer.py and tools/evaluate.py (reference CLI surface) end to end on the tiny golden checkpoint."""
    import subprocess, sys
    from PIL import Image
    z = np.load(os.path.join(golden_dir, "infer_main.npz"))
    with open(os.path.join(golden_dir, "infer_main_meta.json")) as f:
        meta = json.load(f)
    ck = str(tmp_path / "tiny.pt")
    torch.save({"state_dict": {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}, "meta": meta}, ck)
    (tmp_path / "imgs").mkdir()
    for name in ("sq", "wide"):
        Image.fromarray(z[f"img_{name}"][..., ::-1]).save(str(tmp_path / "imgs" / f"{name}.png"))
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "infer.py"), "--weights", ck, "--img_dir",
                        str(tmp_path / "imgs"), "--save_txt"], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    for name in ("sq", "wide"):
        with open(tmp_path / "runs" / "infer" / "1" / "json" / f"{name}.json") as f:
            dets = json.load(f)["detections"]
        assert [d["class_id"] for d in dets] == z[f"{name}/class_id"].tolist()
        np.testing.assert_array_equal(np.rint([d["bbox_xyxy"] for d in dets]), np.rint(z[f"{name}/bbox_xyxy"]))
        assert (tmp_path / "runs" / "infer" / "1" / "labels" / f"{name}.txt").exists()
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "evaluate.py"), "--weights", ck, "--test_folder",
                        str(tmp_path / "imgs"), "--batch_size", "2"], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.splitlines()[0])["detections"] > 0
    # detections.json == the oracle flow of the evaluate path (preprocess_albumentations -> forward ->
    # pipeline_eval at conf 0.001 / iou 0.65; helpers.py:87-153): identical lists on the HIP forward of the oracle's tensor
    from oracle import preproc as opre
    with open(tmp_path / "runs" / "evaluate" / "1" / "detections.json") as f:
        dj = json.load(f)
    mdl, _, S0 = ya.load_model_names_imgsize_from_ckpt(ck, torch.device(DEV))
    for k, name in enumerate(("sq", "wide")):                       # sorted file order
        xx, _ = opre.preprocess_albumentations(z[f"img_{name}"], S0)
        lv = [t.cpu() for t in mdl(torch.from_numpy(xx[None]).to(DEV))]
        exp, _ = opost.pipeline_eval(lv, S0, 0.001, 0.65)
        got = [d for d in dj if d["image_id"] == k]
        assert len(got) > 0 and all(d["file_name"] == f"{name}.png" for d in got)
        assert [d["category_id"] for d in got] == [d["category_id"] for d in exp[0]], name
        np.testing.assert_allclose([d["score"] for d in got], [d["score"] for d in exp[0]], rtol=0, atol=1e-6)
        np.testing.assert_allclose(np.sort(np.asarray([d["bbox"] for d in got]), 0), np.sort(np.asarray([d["bbox"] for d in exp[0]]), 0),
                                   rtol=0, atol=1e-3)              # as sets: ulp-level score ties may swap two rows
    # dataset layout with YOLO labels: images/ + labels/ -> device P/R/F1 curves + confusion matrix stats
    ds = tmp_path / "ds"
    (ds / "images").mkdir(parents=True); (ds / "labels").mkdir()
    for name in ("sq", "wide"):
        Image.fromarray(z[f"img_{name}"][..., ::-1]).save(str(ds / "images" / f"{name}.png"))
        (ds / "labels" / f"{name}.txt").write_text("0 0.5 0.5 0.4 0.4\n1 0.25 0.3 0.2 0.2\n")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "evaluate.py"), "--weights", ck, "--test_folder",
                        str(ds), "--batch_size", "2"], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.splitlines()[0])
    assert 0.0 <= out["best_f1"] <= 1.0 and 0.0 <= out["best_conf"] <= 1.0
    run = tmp_path / "runs" / "evaluate" / "2"
    assert (run / "curves.json").exists()
    assert "Total FP:" in (run / "confusion_matrices" / "confusion_matrix_stats.txt").read_text()


def test_class_argmax_ties_match_torch_first_max():
    """score/class selection on logits built to tie in the sigmoid domain: saturation (> 17 -> 1.0),
    near-saturation neighbours, underflow (< -104 -> 0.0), subnormal range, exact duplicates."""
    C = 12
    rows = []
    def row(cls, obj=3.0):
        return [0.1, -0.2, 0.3, 0.2, obj] + list(cls)
    rows.append(row([18.0, 25.0, 30.0] + [-5.0] * 9))                       # all 1.0f: first wins (index 0)
    rows.append(row([-3.0, 14.2, 14.21, 14.2] + [0.0] * 8))                 # near saturation, may round equal
    rows.append(row([-200.0] * 5 + [-150.0] + [-300.0] * 6))                # all exactly 0.0: index 0
    rows.append(row([-103.9, -103.2, -103.5] + [-120.0] * 9))               # subnormal spacing
    rows.append(row([1.25, 2.5, 2.5, 0.0, 2.5] + [-1.0] * 7))               # exact duplicates: first of them
    rows.append(row([9.99, 10.0, 10.000001, 9.999999] + [3.0] * 8))
    rng = np.random.RandomState(0)
    for _ in range(58):
        base = rng.randn(C).astype(np.float32) * 6
        rows.append(row(base))
    lv = torch.tensor(rows, dtype=torch.float32).reshape(1, 1, 8, 8, 5 + C)
    exp = opost.pipeline_main([lv], 64, conf=-1.0, iou=1.0, per_class_cap=300)
    got = ya.infer_main_postprocess([lv.to(DEV)], 64, conf=-1.0, iou=1.0, per_class_cap=300)
    assert got["classes"][0].tolist() == exp["classes"][0].tolist()
    np.testing.assert_allclose(got["scores"][0], exp["scores"][0], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("S", [96, 640])
def test_preprocess_bit_exact_vs_oracle(S):
    """GPU letterbox + normalise (yl_preprocess) against the CPU restatement of the reference's
    pre-processing with OpenCV-style fixed-point bilinear (oracle/preproc.py): integer resize bit-exact,
    fp32 normalisation evaluated op by op -> identical floats.  Up/down-scaling, odd sizes, portrait /
    landscape / square, already-at-size images in one batch."""
    from oracle import preproc as opre
    rng = np.random.RandomState(7)
    shapes = [(S, S), (S // 2, S), (S, S // 3 + 1), (37, 53), (S * 2 + 3, S + 11), (1080 // 4, 1920 // 4), (S - 1, S - 1)]
    imgs = [rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8) for h, w in shapes]
    ctx = ya.postprocess.context_for([torch.empty((1, 1, s, s, 6)) for s in (S // 8, S // 16, S // 32)], S)
    x, bm = ya.preprocess_batch(ctx, imgs)
    assert x.shape == (len(imgs), 3, S, S)
    for i, im in enumerate(imgs):
        ref, (padx, pady, scale, w0, h0) = opre.preprocess(im, S)
        np.testing.assert_array_equal(x[i].cpu().numpy(), ref)
        assert tuple(bm[i]) == (padx, pady, scale, w0, h0)
    x2, bm2 = ya.preprocess_batch(ctx, imgs[:2], letterbox=False)
    for i in range(2):
        r = opre.resize_linear_u8(imgs[i], S, S)
        ref = ((r[..., ::-1].astype(np.float32) / 255.0 - opre.MEAN) / opre.STD).transpose(2, 0, 1)
        np.testing.assert_array_equal(x2[i].cpu().numpy(), ref)
    # the evaluate path's pipeline (LongestMaxSize + PadIfNeeded + A.Normalize, or A.Resize with --no_letterbox)
    for lb in (True, False):
        x3, bm3 = ya.preprocess_batch(ctx, imgs, letterbox=lb, norm="albumentations")
        for i, im in enumerate(imgs):
            ref, geo = opre.preprocess_albumentations(im, S, resize=not lb)
            np.testing.assert_array_equal(x3[i].cpu().numpy(), ref)
            assert tuple(bm3[i]) == tuple(float(v) for v in geo)
    # the two arithmetics really differ ((u8/255 - mean)/std vs (u8 - 255 mean) * (1 / (255 std))): same geometry
    # (image 0, --no_letterbox), values within float rounding of each other but not all bit-equal
    a3, a2 = x3[0].cpu().numpy(), x2[0].cpu().numpy()
    assert not np.array_equal(a3, a2) and float(np.abs(a3 - a2).max()) < 1e-5
    ya.preprocess_batch(ctx, imgs[:1])                                  # back to the infer arithmetic


@pytest.mark.parametrize("name,B,S", [("edge_n", 2, 320), ("edge_m", 2, 320)])
def test_seg_model_forward_and_masks(name, B, S):
    """BASELINE config 4 family (edge_m + instance-seg head; build-defined branch, parity unpinned -- the
    reference has no mask code): levels incl. mask coefficients and prototypes vs the oracle's own
    restatement, then masks for the HIP detections vs the oracle's mask assembly (mask IoU >= 0.999)."""
    from yololite_amd.program import MODEL_ZOO
    meta = make_meta(num_classes=80, img_size=S, seg=True, **MODEL_ZOO[name])
    sd = synth_state_dict(meta, seed=3, head_noise=2.0)
    for k, v in sd.items():                      # boxes a few strides wide, else the masks are empty
        if k.endswith(".out.box.bias"):
            v[2::4] += 3.0
            v[3::4] += 3.0
    x = _x(B, S, seed=5)
    orc = _oracle_for(meta, sd)
    with torch.no_grad():
        ref_lv, ref_pr = orc(x)
    m = _hip_for(meta, sd)
    lv, pr = m(x.to(DEV))
    _cmp_levels(lv, ref_lv)
    assert pr.shape == ref_pr.shape == (B, 32, S // 4, S // 4)
    assert (pr.cpu() - ref_pr).abs().max().item() <= 1e-4 + 1e-4 * ref_pr.abs().max().item()
    ctx = m._ctx_for(S)
    dets, counts, idx = ctx.predict(x.to(DEV), _lib.POST_MAIN, 0.02, 0.5, per_class_cap=300, want_idx=True)
    cn = counts.cpu().numpy()
    assert cn.min() > 3
    lv_cpu = [t.cpu() for t in lv]
    dec = opost.decode_levels([t[..., :85] for t in lv_cpu], S)
    keep = [idx[b, :cn[b]].cpu().numpy() for b in range(B)]
    boxes = [dec["box"][b][torch.as_tensor(keep[b], dtype=torch.long)].numpy() for b in range(B)]
    total = 0
    for thr in (0.5, 0.02):            # synthetic coefficients give sparse masks at 0.5; 0.02 exercises dense ones
        masks = ctx.masks(counts, idx, dets.shape[1], thr=thr).cpu().numpy()
        exp = opost.masks_for(lv_cpu, pr.cpu(), 80, S, keep, boxes, thr=thr)
        for b in range(B):
            got = masks[b, :cn[b]].astype(bool)
            ref = exp[b].astype(bool)
            inter, union = (got & ref).sum(), (got | ref).sum()
            assert union == 0 or inter / union >= 0.999, (thr, b, inter, union)
            total += int(union)
    assert total > 100
    # image-resolution masks through a letterbox: original images of different sizes / aspect ratios
    hw = [(S // 2 + 7, S), (S + 40, S * 2 // 3)][:B]
    bm = []
    for (h0, w0) in hw:
        sc = min(S / h0, S / w0)
        nh, nw = int(round(h0 * sc)), int(round(w0 * sc))
        bm.append(((S - nw) // 2, (S - nh) // 2, sc, w0, h0))
    bmt = torch.tensor(bm, dtype=torch.float32)
    d3, c3, i3 = ctx.predict(x.to(DEV), _lib.POST_MAIN, 0.02, 0.5, per_class_cap=300, want_idx=True, backmap=bmt)
    got = ctx.masks_image(d3, c3, i3, backmap=bmt, thr=0.3)
    keep3 = [i3[b, :int(c3[b])].cpu().numpy() for b in range(B)]
    box3 = [d3[b, :int(c3[b]), :4].cpu().numpy() for b in range(B)]
    exp3 = opost.masks_image_for(lv_cpu, pr.cpu(), 80, S, keep3, box3, hw, backmap=bm, thr=0.3)
    tot = 0
    for b in range(B):
        g, e = got[b].cpu().numpy().astype(bool), exp3[b].astype(bool)
        assert g.shape == e.shape == (int(c3[b]), hw[b][0], hw[b][1])
        inter, union = (g & e).sum(), (g | e).sum()
        assert union == 0 or inter / union >= 0.999, (b, inter, union)
        tot += int(union)
    assert tot > 100
    # detections themselves: same as the detector-only pipeline run on the detection part of the rows
    exp_det = opost.pipeline_main([t[..., :85] for t in lv_cpu], S, 0.02, 0.5, 300)
    for b in range(B):
        d = dets[b, :cn[b]].cpu().numpy()
        _match(d[:, :4], d[:, 4], d[:, 5].astype(np.int64), exp_det["boxes"][b], exp_det["scores"][b], exp_det["classes"][b])


def test_batch_size_changes_keep_buffers_graphs_and_mask_inputs():
    """activation capacity only grows.  Alternating batch sizes (a tail batch, B = 4 / 16) replay
    their own cached hipGraphs with unchanged results and unchanged memory; yl_masks_image refuses a batch larger than
    the last forward's (its level buffers / prototypes would be another batch's), and works again after that batch."""
    from yololite_amd.program import MODEL_ZOO
    S = 256
    meta = make_meta(num_classes=80, img_size=S, seg=True, **MODEL_ZOO["edge_n"])
    sd = synth_state_dict(meta, seed=3, head_noise=2.0)
    m = _hip_for(meta, sd)
    ctx = m._ctx_for(S)
    xs = {16: _x(16, S, seed=21).to(DEV), 4: _x(4, S, seed=22).to(DEV), 7: _x(7, S, seed=23).to(DEV)}
    ctx.set_option("graph", 1)
    ref, mem = {}, None
    for rnd in range(3):
        for B in (16, 4, 7, 16, 4):
            d, c, i = ctx.predict(xs[B], _lib.POST_MAIN, 0.05, 0.5, per_class_cap=300, want_idx=True)
            got = (d.clone(), c.clone())
            if B not in ref:
                ref[B] = got
            assert torch.equal(got[1], ref[B][1]), (rnd, B)
            for b in range(B):
                n = int(got[1][b])
                assert torch.equal(got[0][b, :n], ref[B][0][b, :n]), (rnd, B, b)
            if mem is None:
                mem = ctx.activation_bytes()                       # allocated for the largest batch first
            assert ctx.activation_bytes() == mem, (rnd, B)         # a smaller batch re-plans inside the allocation
            mk = ctx.masks_image(d, c, i, packed=True)             # the batch that just ran: fine
            assert len(mk) == B
    d16, c16, i16 = ctx.predict(xs[16], _lib.POST_MAIN, 0.05, 0.5, per_class_cap=300, want_idx=True)
    m16 = [t.clone() for t in ctx.masks_image(d16, c16, i16, packed=True)]
    ctx.predict(xs[4], _lib.POST_MAIN, 0.05, 0.5, per_class_cap=300)
    with pytest.raises(_lib.YoloLiteHipError):                     # level buffers now hold the B = 4 batch
        ctx.masks_image(d16, c16, i16, packed=True)
    d16b, c16b, i16b = ctx.predict(xs[16], _lib.POST_MAIN, 0.05, 0.5, per_class_cap=300, want_idx=True)
    for a, b in zip(m16, ctx.masks_image(d16b, c16b, i16b, packed=True)):
        assert torch.equal(a, b)
    ctx.set_option("graph", 0)


def test_split_head_output_of_a_seg_model_is_bitwise_the_combined_launch():
    """Round 4: under yl_predict a seg model's head-output conv (5+C+NM = 117 columns: no float4 epilogue, raw rows needed for
    the mask coefficients -> the generic kernel's scalar epilogue at 40 TFLOP/s) runs as TWO launches of the same 1x1 conv --
    rows [0, 85) with the decode in the epilogue and no raw rows (the detector's fast path), rows [85, 117) as a plain 1x1
    storing into the level rows' coefficient columns.  Same k order per output: detections AND masks are bit-identical to
    the combined launch ("fuse_head" 0).  Round 6: on levels with enough pixels the two parts are ONE launch again
    (yl_conv_pws_kernel's decode form with the coefficient columns from a second weight image: the rows are read once) --
    edge_m at 640 x 640, B = 8 takes it on the 80 x 80 level and the two-launch form on the others."""
    from yololite_amd.program import MODEL_ZOO
    for name, S, B in (("edge_m", 320, 2), ("edge_n", 320, 2), ("edge_m", 640, 8)):
        meta = make_meta(num_classes=80, img_size=S, seg=True, **MODEL_ZOO[name])
        sd = synth_state_dict(meta, seed=3, head_noise=2.0)
        for k, v in sd.items():
            if k.endswith(".out.box.bias"):
                v[2::4] += 3.0
                v[3::4] += 3.0
        m = _hip_for(meta, sd)
        ctx = m._ctx_for(S)
        x = _x(B, S, seed=5).to(DEV)
        res = {}
        for fh in (0, 1):
            ctx.set_option("fuse_head", fh)
            d, c, i = ctx.predict(x, _lib.POST_MAIN, 0.02, 0.5, per_class_cap=300, max_out=128, want_idx=True)
            mk = ctx.masks_image(d, c, i, packed=True)
            res[fh] = (d.clone(), c.clone(), [t.clone() for t in mk])
        ctx.set_option("fuse_head", 1)
        assert int(res[0][1].min()) > 0 and torch.equal(res[0][1], res[1][1])
        for b in range(B):
            n = min(int(res[0][1][b]), 128)
            assert torch.equal(res[0][0][b, :n], res[1][0][b, :n]), (name, b)
            assert torch.equal(res[0][2][b], res[1][2][b]), (name, b)
        assert sum(int(t.ne(0).sum()) for t in res[1][2]) > 100


def test_masks_image_every_batch_size_17_to_48():
    """the launcher asked for 28 bytes too little dynamic LDS (hand-counted level tables), so
    the per-image item prefix pre[B-6..B] lay past the request and -- for the B whose request ended on an allocation
    granule -- read as 0: no mask byte written.  One forward at B = 48; yl_masks_image on the first B images for
    EVERY B in 17..48, packed and uint8, into a POISONED fixed-capacity arena must give exactly the masks of the same
    images in the B = 16 / B = 48 calls (a launch that writes nothing leaves the poison)."""
    from yololite_amd.program import MODEL_ZOO
    S, BM, MO = 128, 48, 64
    meta = make_meta(num_classes=80, img_size=S, seg=True, **MODEL_ZOO["edge_n"])
    sd = synth_state_dict(meta, seed=3, head_noise=2.0)
    for k, v in sd.items():
        if k.endswith(".out.box.bias"):
            v[2::4] += 3.0
            v[3::4] += 3.0
    m = _hip_for(meta, sd)
    ctx = m._ctx_for(S)
    x = _x(BM, S, seed=31).to(DEV)
    d, c, i = ctx.predict(x, _lib.POST_MAIN, 0.05, 0.5, per_class_cap=300, max_out=MO, want_idx=True)
    cn = c.cpu().numpy().clip(0, MO)
    assert cn.min() > 0
    for packed in (True, False):
        arena = torch.empty((BM * MO * S * (S // 8 if packed else S),), device=DEV, dtype=torch.uint8)

        def run(B):
            arena.fill_(0xA5)
            v = ctx.masks_image(d[:B].contiguous(), c[:B].contiguous(), i[:B].contiguous(), packed=packed, arena=arena)
            return [v[b, :cn[b]].clone() for b in range(B)]
        ref16, ref48 = run(16), run(BM)
        assert sum(int(t.ne(0).sum()) for t in ref16) > 100
        for b in range(16):
            assert torch.equal(ref16[b], ref48[b]), (packed, b)
        # the written entries hold mask data, not poison: a uint8 mask is 0/1; a packed 128-px row is 4 words whose
        # poison value would be 0xA5A5A5A5 everywhere
        for t in ref48:
            assert int(t.max()) <= 1 if not packed else not bool((t == torch.tensor(0xA5A5A5A5 - (1 << 32), dtype=torch.int32, device=DEV)).all())
        for B in range(17, BM):
            got = run(B)
            for b in range(B):
                assert torch.equal(got[b], ref48[b]), (packed, B, b)


def test_pip_api_predict_on_a_seg_checkpoint(tmp_path):
    """YoloLite(path).predict() on a (build-defined) seg checkpoint -- `masks` is a list of
    [N_i, h0, w0] arrays at the ORIGINAL image sizes (README.md:38-42), equal to the oracle's masks_image_for on the
    oracle's levels / prototypes through the same letterbox (mask IoU >= 0.999); the shared context is left in its
    default mode (time_split restored)."""
    from yololite_amd.api import YoloLite
    from yololite_amd.program import MODEL_ZOO
    from oracle import preproc as opre
    S = 320
    meta = make_meta(num_classes=80, img_size=S, seg=True, **MODEL_ZOO["edge_n"])
    meta["names"] = [f"c{i}" for i in range(80)]
    sd = synth_state_dict(meta, seed=3, head_noise=2.0)
    for k, v in sd.items():                      # boxes a few strides wide, else the masks are empty
        if k.endswith(".out.box.bias"):
            v[2::4] += 3.0
            v[3::4] += 3.0
    ck = str(tmp_path / "seg.pt")
    torch.save({"state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}, "meta": meta}, ck)
    rng = np.random.RandomState(2)
    imgs = [rng.randint(0, 256, size=hw + (3,)).astype(np.uint8) for hw in ((240, 320), (400, 250), (320, 320))]
    yl = YoloLite(ck, device=DEV)
    res = yl.predict(imgs, conf=0.05)
    ctx = yl.model._ctx_for(S)
    assert ctx.get_option("time_split", 0) == 0
    orc = _oracle_for(meta, sd)
    tot = 0
    for im, r in zip(imgs, res):
        h0, w0 = im.shape[:2]
        n = len(r["scores"])
        assert n > 3 and r["masks"].shape == (n, h0, w0) and r["masks"].dtype == np.uint8
        xx, (padx, pady, scale, _, _) = opre.preprocess(im, S)
        with torch.no_grad():
            lv, pr = orc(torch.from_numpy(xx[None]))
        # the oracle's own detections of this image must be the API's (same candidates -> same coefficients)
        exp = opost.pipeline_main([t[..., :85] for t in lv], S, 0.05, 0.5, 300)
        if exp["classes"][0].tolist() != r["classes"].tolist():
            continue                                               # a threshold / NMS decision inside the fp32 drift
        dec = opost.decode_levels([t[..., :85] for t in lv], S)
        sc, _ = opost.score_candidates(dec["obj"][0].squeeze(-1), dec["cls"][0])
        # candidate index of every detection: match decoded + back-mapped boxes (unique with overwhelming probability)
        bm_boxes = opost.backmap(dec["box"][0].numpy().copy(), padx, pady, scale, w0, h0)
        keep = []
        for bb, ss in zip(r["boxes"], r["scores"]):
            d = np.abs(bm_boxes - bb[None]).max(1) + np.abs(sc.numpy() - ss)
            keep.append(int(d.argmin()))
        em = opost.masks_image_for(lv, pr, 80, S, [np.asarray(keep)], [r["boxes"]], [(h0, w0)],
                                   backmap=[(padx, pady, scale, w0, h0)])[0].astype(bool)
        g = r["masks"].astype(bool)
        inter, union = (g & em).sum(), (g | em).sum()
        assert union == 0 or inter / union >= 0.999, (inter, union)
        tot += int(union)
    assert tot > 100, tot


@pytest.mark.parametrize("S,B", [(640, 2), (320, 3), (384, 2)])
def test_ir_fusion_is_bitwise_the_two_launch_form(S, B):
    """yl_ir_kernel (round 3): yololite_m's early EfficientNet-Lite inverted-residual blocks (stride 1 and 2, TF-SAME
    pads, 3x3 and 5x5, with and without residual) as ONE launch each -- same k orders, tap order and epilogues as
    conv_pw followed by depthwise + conv_pwl, so the raw levels must be bit-identical to the unfused program.  640: all
    seven blocks fuse; 320 / 384: a subset (tile divisibility), the rest runs the two-launch form inside the same model."""
    meta = zoo_meta("yololite_m", 80, S)
    sd = synth_state_dict(meta, seed=6)
    x = _x(B, S, seed=31).to(DEV)
    mf = ya.build_model_from_meta(meta, fuse_ir=True); mf.load_state_dict(sd); mf.to(DEV)
    mu = ya.build_model_from_meta(meta, fuse_ir=False); mu.load_state_dict(sd); mu.to(DEV)
    nf = sum(1 for l in mf.program.layers if l.name.endswith(".ir"))
    assert nf >= (7 if S == 640 else 2) and not any(l.name.endswith(".ir") for l in mu.program.layers)
    assert len(mf.program.layers) == len(mu.program.layers) - nf
    for a, b in zip(mf(x), mu(x)):
        assert torch.equal(a, b)
    # and in the bf16-MFMA mode the fused program stays within that mode's bound of the fp32 result
    ctx = mf._ctx_for(S)
    ref = [t.clone() for t in mf(x)]
    ctx.set_option("mfma_bf16", 1)
    for a, r in zip(mf(x), ref):
        assert float((a - r).abs().max()) <= 3e-2 * max(float(r.abs().max()), 1.0)
    ctx.set_option("mfma_bf16", 0)
    for a, r in zip(mf(x), ref):
        assert torch.equal(a, r)


@pytest.mark.parametrize("S,B", [(640, 3), (384, 2), (128, 5), (96, 7)])
def test_window_in_lds_head_kernel_is_bitwise_the_tap_load_kernel(S, B):
    """yl_conv_dpw_kernel (round 6: the head launch with the depthwise input windows in wave-private LDS rings, copied by
    LDS-DMA) against yl_conv_dpp_kernel ("dev_select" bit 16: nine fragment-shaped tap loads per block): same tap order, fma
    chain, k order and decode epilogue -> the same BITS in every detection row, main and eval post modes.  128 / 96: every
    tile touches the image border (the zero-buffer sources of the window copies); B = 5 / 7: tile counts that are not a
    multiple of the workgroup's waves (the trailing waves run the loop on zero-source windows)."""
    meta = zoo_meta("edge_n", 80, S)
    sd = synth_state_dict(meta, seed=1, head_noise=2.0)
    m = _hip_for(meta, sd)
    ctx = m._ctx_for(S)
    x = _x(B, S, seed=61).to(DEV)
    res = {}
    try:
        for dv in (_lib.DEV_DPW_OFF, 0):
            ctx.set_option("dev_select", dv)
            for mode, conf, iou, cap in ((_lib.POST_MAIN, 0.02, 0.5, 300), (_lib.POST_EVAL, 0.001, 0.65, 0)):
                mo = 1024 if mode == _lib.POST_MAIN else ctx.N
                d, c = ctx.predict(x, mode, conf, iou, per_class_cap=cap, max_out=mo)
                res[(dv, mode)] = (d.cpu().numpy().copy(), c.cpu().numpy().copy())
    finally:
        ctx.set_option("dev_select", 0)
    for mode in (_lib.POST_MAIN, _lib.POST_EVAL):
        d0, c0 = res[(_lib.DEV_DPW_OFF, mode)]
        d1, c1 = res[(0, mode)]
        assert np.array_equal(c0, c1) and int(c0.sum()) > 0
        for i in range(B):
            assert np.array_equal(d0[i, :c0[i]].view(np.uint32), d1[i, :c1[i]].view(np.uint32)), (mode, i)


def test_stream_overlap_probe_and_pipeline_lane_streams():
    """yl_streams_overlap (round 6): ROCm hands a stream its hardware queue at creation, round-robin over every stream the
    process has created, and two streams whose queues alias serialise their kernel chains (46 k -> 39 k images/s).  The probe
    must (a) answer for any pair, (b) say 0 for a stream against ITSELF (one in-order queue: the two chains cannot overlap),
    (c) be what ServingPipeline picks its lane streams by: after a varying number of throw-away streams the lanes still
    overlap each other and the submitting stream."""
    import ctypes as C
    from yololite_amd.serving import ServingPipeline
    lib = _lib.load()
    r = C.c_int32(-1)
    s0 = torch.cuda.Stream(device=DEV)
    assert lib.yl_streams_overlap(0, C.c_void_p(s0.cuda_stream), C.c_void_p(s0.cuda_stream), C.byref(r)) == _lib.YL_OK
    assert r.value == 0
    assert lib.yl_streams_overlap(0, None, C.c_void_p(s0.cuda_stream), C.byref(r)) == _lib.YL_OK and r.value in (0, 1)
    assert lib.yl_streams_overlap(0, None, None, None) != _lib.YL_OK
    meta = zoo_meta("edge_n", 80, 128)
    m = _hip_for(meta, synth_state_dict(meta, seed=3))
    ctx = m._ctx_for(128)
    hip = C.CDLL("libamdhip64.so")
    junk = []
    for extra in (0, 1, 2, 3):
        for _ in range(extra):                       # shift the round-robin position of the next streams
            h = C.c_void_p()
            assert hip.hipStreamCreateWithFlags(C.byref(h), 1) == 0
            junk.append(h)
        pipe = ServingPipeline(ctx, lanes=2, streams_per_lane=1, graph=True)
        a, b = pipe.streams
        cur = torch.cuda.current_stream(torch.device(DEV))
        for u, v in ((a, b), (a, cur), (b, cur)):
            assert lib.yl_streams_overlap(0, C.c_void_p(u.cuda_stream), C.c_void_p(v.cuda_stream), C.byref(r)) == _lib.YL_OK
            assert r.value == 1, (extra, u, v)
        del pipe
    for h in junk:
        hip.hipStreamDestroy(h)


@pytest.mark.parametrize("S,B,taken", [(640, 3, True), (384, 2, True), (128, 5, True), (352, 2, False)])
def test_fused_head_launch_is_bitwise_and_never_writes_the_trunk_tensor(S, B, taken):
    """yl_conv_dpp_kernel (round 3): under yl_predict the head branches of edge_n -- depthwise 3x3 -> 1x1 trunk -> 1x1 head
    output -> decode -- run as ONE launch for all levels (option "fuse_head", default on).  Same k orders and arithmetic
    as yl_conv_dwc/dwt_kernel + yl_conv_pwt_kernel<DEC>: detections of both post modes are the same BITS with the option
    off; and the launch is really taken: the trunk tensors keep the previous call's contents.  352: a level grid is not a
    multiple of the 4x4 wave tile -- the two-launch form runs, same result."""
    meta = zoo_meta("edge_n", 80, S)
    sd = synth_state_dict(meta, seed=12)
    m = _hip_for(meta, sd)
    ctx = m._ctx_for(S)
    ctx.set_option("reuse_slots", 0)                      # every tensor keeps its own memory: slots can be read back
    trunks = [l for l in m.program.layers if ".trunk." in l.name and l.dw_k == 3]
    assert len(trunks) == 3
    # the same option covers yl_conv_dpq_kernel: blocks.2.5 (depthwise 3x3 -> 1x1 48->192 -> 1x1 192->48 + residual, a
    # MobileNetV4 UIB block with a start depthwise only) as one launch; its expanded tensor is never written either
    exp = [l for l in m.program.layers if l.name.endswith("blocks.2.5.pw_exp.conv")]
    assert len(exp) == 1 and exp[0].dw_k == 3 and exp[0].cout == 192
    xa, xb = _x(B, S, seed=51).to(DEV), _x(B, S, seed=52).to(DEV)

    def run(x, mode, conf, iou, cap):
        mo = 1024 if mode == _lib.POST_MAIN else ctx.N
        d, c = ctx.predict(x, mode, conf, iou, per_class_cap=cap, max_out=mo)
        return d.cpu().numpy().copy(), c.cpu().numpy().copy()

    def slots():
        return [ctx.read_slot(l.out_slot, B, (S // st, S // st, l.cout)).clone()
                for l, st in zip(trunks + exp, (8, 16, 32, 16))]

    try:
        ctx.set_option("fuse_head", 0)
        ref = {k: run(xb, *k) for k in ((_lib.POST_MAIN, 0.25, 0.5, 300), (_lib.POST_EVAL, 0.001, 0.65, 0))}
        run(xa, _lib.POST_MAIN, 0.25, 0.5, 300)
        ta = slots()                                      # trunk outputs of xa, written by the two-launch form
        ctx.set_option("fuse_head", 1)
        for k, (d0, c0) in ref.items():
            d1, c1 = run(xb, *k)
            assert np.array_equal(c0, c1) and (k[0] != _lib.POST_EVAL or int(c0.sum()) > 0)
            for i in range(B):
                assert np.array_equal(d0[i, :c0[i]].view(np.uint32), d1[i, :c1[i]].view(np.uint32))
        same = [torch.equal(a, b) for a, b in zip(ta, slots())]
        assert all(same) if taken else not any(same)      # taken: untouched by the fused launches on xb
        lv1 = [t.clone() for t in m(xb)]                  # raw levels (yl_forward): the pair launch, no head fusion
        ctx.set_option("fuse_head", 0)
        for a, b in zip(m(xb), lv1):
            assert torch.equal(a, b)
        run(xb, _lib.POST_MAIN, 0.25, 0.5, 300)
        assert not all(torch.equal(a, b) for a, b in zip(ta, slots()))
    finally:
        ctx.set_option("fuse_head", 1); ctx.set_option("reuse_slots", 1)


@pytest.mark.parametrize("name,S,B", [("edge_n", 640, 2), ("edge_n", 384, 3), ("edge_n", 320, 3)])
def test_uib_and_lateral_fusion_through_the_ir_kernel_is_bitwise(name, S, B):
    """Round 3: MobileNetV4 UIB blocks without a start depthwise and the FPN pairs lateral{k} (1x1 + bias + upsample-add)
    -> smooth{k} (depthwise block) run through yl_ir_kernel where it is instantiated; blocks.1.1 (1x1) is chained in the
    epilogue of blocks.1.0 (3x3 s2).  Same arithmetic order as the
    stand-alone launches: the raw levels are bit-identical to the program built with both fusions off."""
    meta = zoo_meta(name, 80, S)
    sd = synth_state_dict(meta, seed=8)
    x = _x(B, S, seed=41).to(DEV)
    mf = _hip_for(meta, sd)
    mu = _hip_for(meta, sd, fuse_uir=False, fuse_lat=False, fuse_chain=False)
    names_f = [l.name for l in mf.program.layers]
    # 320: the 20x20 / 40x40 grids have no 8x8 / 8x16 workgroup tiling -- those blocks keep the two-launch form (the 4x20
    # tiling was measured slower and is not instantiated); only the chained 1x1 remains
    assert S == 320 or (any("+smooth" in n for n in names_f) and any(n.endswith(".uib") for n in names_f))
    assert any(l.c3 > 0 and l.op == 1 for l in mf.program.layers)           # blocks.1.0 with blocks.1.1 chained in its epilogue
    assert not any("+smooth" in l.name or l.name.endswith(".uib") or (l.c3 > 0 and l.op == 1) for l in mu.program.layers)
    assert len(mu.program.layers) > len(mf.program.layers)
    for a, b in zip(mf(x), mu(x)):
        assert torch.equal(a, b)
