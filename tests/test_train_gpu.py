"""Training slice: gradients of the reference's segmentation loss for every parameter downstream of the LLM's
last hidden state, against the reference's own autograd (tests/golden/train_slice.npz, made by oracle/gen_golden.py::case_train_slice
from train_joint_v2.py's calc_instance_loss / BCELoss / DiceLoss over the reference modules)."""
import numpy as np
import pytest
import torch

from tests import util as U
from tests.test_model_gpu import _ullsam_tiny

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _inputs(g):
    rng = np.random.default_rng(int(g["img_seed"]))
    hid = rng.standard_normal((1, 1024, 256), dtype=np.float32)
    img = rng.standard_normal((1, 256, 64, 64), dtype=np.float32)
    assert np.array_equal(hid.reshape(-1)[::1009], g["hid_sample"])
    yy, xx = np.mgrid[0:1024, 0:1024].astype(np.float32)
    gt = np.stack([((xx - 300) ** 2 + (yy - 340) ** 2 < 150 ** 2), ((xx - 700) ** 2 + (yy - 610) ** 2 < 220 ** 2)]).astype(np.float32)[:, None]
    t = lambda a: torch.from_numpy(a).to(DEV)
    return t(hid), t(img), (t(g["pts"]), t(g["lbl"])), t(gt)


def test_segmentation_loss_and_gradients_equal_the_reference_autograd():
    """fp32: loss / bce / dice to 1e-5 relative; every parameter gradient the reference produces (125 tensors: mlp2, llm_scale_factor,
    llm_bias, point / not-a-point embeddings, the whole two-way transformer, both transposed convolutions, the four hypernetwork MLPs)
    within 1e-3 of the tensor's largest entry, and its L2 norm within 1e-3; parameters the reference leaves without a gradient (IoU head,
    mask-input convolutions, box-corner embeddings) have none here either."""
    from ullsam_amd.training import segmentation_loss
    g = U.gold("train_slice")
    m = _ullsam_tiny(torch.float32)
    for p in m.parameters():
        p.requires_grad_(False)
    trainable = {n: p for n, p in m.named_parameters() if n.startswith(("mlp2.", "prompt_encoder.", "mask_decoder."))}
    for p in trainable.values():
        p.requires_grad_(True)
    hid, img, pts, gt = _inputs(g)
    loss, bce, dice = segmentation_loss(m, hid, img, pts, gt)
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"]), (loss.item(), float(g["loss"]))
    assert abs(bce.item() - float(g["bce"])) < 1e-5 * float(g["bce"]) and abs(dice.item() - float(g["dice"])) < 1e-5 * float(g["dice"])
    loss.backward()
    names = [str(n) for n in g["names"]]
    worst = (0.0, "")
    for n in names:
        ref = g["g:" + n].astype(np.float64)
        p = trainable[n]
        assert p.grad is not None, n
        full = p.grad.float().cpu().numpy().reshape(-1).astype(np.float64)
        stride = max(1, full.size // 2048)
        got = full[::stride]
        scale = np.abs(ref).max()
        diff = np.abs(got - ref).max()
        # (+ 1e-7 absolute: gradients that are zero in exact arithmetic -- every k_proj.bias, by softmax's shift invariance -- are ~1e-9 of
        # rounding noise on both sides, against typical entries of 1e-3 ... 1)
        assert diff < 1e-3 * scale + 1e-7, (n, diff, scale)
        if scale > 1e-6:
            worst = max(worst, (diff / scale, n))
        nref = float(g["n:" + n])
        assert abs(np.sqrt((full ** 2).sum()) - nref) < 1e-3 * nref + 1e-6, (n, np.sqrt((full ** 2).sum()), nref)
    for n, p in trainable.items():
        if n not in names:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{n}: the reference leaves this parameter without a gradient"
    print("worst relative gradient error", worst)


def test_bf16_model_gets_the_fp32_step_on_its_rounded_weights():
    """The trainer's default keeps the model in bf16 (train_joint_v2.py:1599,1676).  The step widens bf16 parameters and computes in fp32, so a
    bf16 model's gradients are those of the fp32 model holding the same (bf16-rounded) weights, rounded to bf16: compared tensor by tensor on
    the decoder-side slice."""
    from ullsam_amd import training
    from ullsam_amd.training import segmentation_loss
    g = U.gold("train_slice")
    hid, img, pts, gt = _inputs(g)
    grads = []
    old_switch, training.BF16_LINEAR = training.BF16_LINEAR, False      # this test is about the fp32-arithmetic route (the bf16 GEMM route: next test)
    for widen in (False, True):
        m = _ullsam_tiny(torch.bfloat16)
        if widen:
            m = m.float()                                  # the same rounded weights, held in fp32
        for n, p in m.named_parameters():
            p.requires_grad_(n.startswith(("mlp2.", "prompt_encoder.", "mask_decoder.")))
        loss, _, _ = segmentation_loss(m, hid, img, pts, gt)
        loss.backward()
        grads.append((float(loss.detach()), {n: p.grad for n, p in m.named_parameters() if p.grad is not None}))
    training.BF16_LINEAR = old_switch
    (l16, g16), (l32, g32) = grads
    assert abs(l16 - l32) < 1e-5 * abs(l32)               # (sums by atomics: the two runs differ in the order of additions)
    assert set(g16) == set(g32) and len(g16) >= 125
    for n in g16:
        assert g16[n].dtype == torch.bfloat16
        a, b = g16[n].float(), g32[n]
        assert float((a - b).abs().max()) <= 2.0 ** -8 * float(b.abs().max()) + 1e-7, n   # one bf16 rounding of each entry


@pytest.mark.parametrize("fixture", ["train_llm_slice", "train_llm_slice_pad"])
def test_gradients_through_the_frozen_llm_reach_mlp1(fixture):
    """(train_llm_slice_pad: the same slice with the prompt LEFT-PADDED by 37 positions and attention_mask = 0 there: the training attention's
    key_mask branches -- padded keys, fully masked padded query rows, causal + padding summing to -inf -- forward and backward.)
    Second slice: vision features -> pixel_shuffle -> mlp1 -> image-token splice -> InternLM2 (2 layers, frozen: RMSNorm, wqkv, RoPE,
    causal grouped attention with the padding mask, wo, SwiGLU) -> hidden states of the image tokens -> the segmentation branch.  Against
    tests/golden/train_llm_slice.npz (the reference's autograd through its own LLM): the hidden states, the loss, the gradients of mlp1
    (reachable only through the LLM's backward), mlp2 and the sampled decoder tensors within 1e-3 of each tensor's largest entry, and
    d loss / d vision features; the LLM's own parameters receive no gradient."""
    from ullsam_amd import ops
    from ullsam_amd.training import llm_image_hidden, segmentation_loss
    g = U.gold(fixture)
    m = _ullsam_tiny(torch.float32)
    for n, p in m.named_parameters():
        p.requires_grad_(not n.startswith(("language_model.", "vision_model.")))
    rng = np.random.default_rng(int(g["seed"]))
    feat = rng.standard_normal((1, 256, 64, 64), dtype=np.float32)
    img = rng.standard_normal((1, 256, 64, 64), dtype=np.float32)
    t = lambda a: torch.from_numpy(a).to(DEV)
    rows = ops.transpose(t(feat).reshape(1, 256, 4096), 1, 256, 4096).requires_grad_(True)      # NHWC rows of the vision features
    ids = t(g["ids"]).long()
    yy, xx = np.mgrid[0:1024, 0:1024].astype(np.float32)
    gt = np.stack([((xx - 300) ** 2 + (yy - 340) ** 2 < 150 ** 2), ((xx - 700) ** 2 + (yy - 610) ** 2 < 220 ** 2)]).astype(np.float32)[:, None]
    amask = t(g["attention_mask"]).long()
    assert (fixture == "train_llm_slice_pad") == bool((amask == 0).any())
    hidden = llm_image_hidden(m, rows, ids, amask)
    hs = hidden.detach().float().cpu().numpy().reshape(-1)[::97]
    assert np.abs(hs - g["hidden_sample"]).max() < 2e-4 * max(1.0, np.abs(g["hidden_sample"]).max()), np.abs(hs - g["hidden_sample"]).max()
    loss, bce, dice = segmentation_loss(m, hidden, t(img), (t(g["pts"]), t(g["lbl"])), t(gt))
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"]), (loss.item(), float(g["loss"]))
    loss.backward()
    params = dict(m.named_parameters())
    worst = (0.0, "")
    for n in [str(x) for x in g["names"]] + ["vit_features"]:
        ref = g["g:" + n].astype(np.float64)
        if n == "vit_features":   # the fixture's gradient is NCHW, ours NHWC rows
            full = rows.grad.reshape(1, 4096, 256).permute(0, 2, 1).contiguous().cpu().numpy().reshape(-1).astype(np.float64)
            got = full[::257]
        else:
            full = params[n].grad.float().cpu().numpy().reshape(-1).astype(np.float64)
            got = full[::max(1, full.size // 2048)]
        scale, diff = np.abs(ref).max(), np.abs(got - ref).max()
        assert diff < 1e-3 * scale + 1e-7, (n, diff, scale)
        worst = max(worst, (diff / scale, n))
        nref = float(g["n:" + n])
        assert abs(np.sqrt((full ** 2).sum()) - nref) < 1e-3 * nref + 1e-6, (n, np.sqrt((full ** 2).sum()), nref)
    assert all(p.grad is None for n, p in params.items() if n.startswith("language_model."))
    print("worst relative gradient error", worst)


def test_gradients_reach_every_vision_model_parameter():
    """Third slice: the image goes through the vision model with gradients (the trainer's second ViT call, train_joint_v2.py:1014-1021):
    patch embedding, pos_embed, a 14x14-windowed block with padded windows and a global block over 64 x 64 tokens (decomposed relative-position
    terms and their tables included), the neck's 1x1 / 3x3 convolutions and LayerNorm2ds.  Against tests/golden/train_vit_slice.npz: the
    image embedding, the loss and all 37 vision-model gradients within 1e-3 of each tensor's largest entry."""
    from ullsam_amd.training import segmentation_loss, vision_feature_rows
    g = U.gold("train_vit_slice")
    m = _ullsam_tiny(torch.float32)
    for n, p in m.named_parameters():
        p.requires_grad_(n.startswith("vision_model."))
    rng = np.random.default_rng(int(g["seed"]))
    hid = rng.standard_normal((1, 1024, 256), dtype=np.float32)
    assert np.array_equal(hid.reshape(-1)[::1009], g["hid_sample"])
    from oracle import ullsam_oracle as O  # noqa: F401
    x = U.rand_image((1, 3, 1024, 1024), seed=13)
    t = lambda a: torch.from_numpy(a).to(DEV)
    yy, xx = np.mgrid[0:1024, 0:1024].astype(np.float32)
    gt = np.stack([((xx - 300) ** 2 + (yy - 340) ** 2 < 150 ** 2), ((xx - 700) ** 2 + (yy - 610) ** 2 < 220 ** 2)]).astype(np.float32)[:, None]
    rows = vision_feature_rows(m.vision_model, t(x))
    emb = rows.detach().reshape(1, 4096, 256).permute(0, 2, 1).contiguous().cpu().numpy().reshape(-1)[::997]     # NCHW order of the fixture
    assert np.abs(emb - g["emb_sample"]).max() < 2e-4 * max(1.0, np.abs(g["emb_sample"]).max()), np.abs(emb - g["emb_sample"]).max()
    loss, bce, dice = segmentation_loss(m, t(hid), None, (t(g["pts"]), t(g["lbl"])), t(gt), image_rows=rows)
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"]), (loss.item(), float(g["loss"]))
    loss.backward()
    params = dict(m.named_parameters())
    worst = (0.0, "")
    for n in [str(v) for v in g["names"]]:
        ref = g["g:" + n].astype(np.float64)
        full = params[n].grad.float().cpu().numpy().reshape(-1).astype(np.float64)
        got = full[::max(1, full.size // 2048)]
        scale, diff = np.abs(ref).max(), np.abs(got - ref).max()
        assert diff < 1e-3 * scale + 1e-7, (n, diff, scale)
        worst = max(worst, (diff / scale, n))
        nref = float(g["n:" + n])
        assert abs(np.sqrt((full ** 2).sum()) - nref) < 1e-3 * nref + 1e-6, (n, np.sqrt((full ** 2).sum()), nref)
    print("worst relative gradient error", worst)


@pytest.mark.parametrize("mfma_linear", [True, False])
def test_whole_train_step_gradients_of_every_trainable_module(mfma_linear):
    """One whole step of the reference's trainer (train_joint_v2.py:990-1100 on the tiny composite; fixture train_step.npz: model(...) with
    output_hidden_states, the second vision_model call, prompt encoder, mask decoder, upsample, calc_instance_loss; LLM frozen) against
    ullsam_amd.training.train_step_loss: the loss and the gradient of every parameter the reference's step produces one for -- vision model,
    mlp1, mlp2, prompt encoder, mask decoder -- within 1e-3 of the tensor's largest entry; none for the LLM."""
    import time
    from ullsam_amd import training
    from ullsam_amd.training import train_step_loss
    training.MFMA_LINEAR = mfma_linear      # nn.Linear on the fp32 MFMA GEMM where shapes allow / on the plain kernel everywhere
    g = U.gold("train_step")
    m = _ullsam_tiny(torch.float32)
    for n, p in m.named_parameters():
        p.requires_grad_(not n.startswith("language_model."))
    t = lambda a: torch.from_numpy(a).to(DEV)
    x = t(U.rand_image((1, 3, 1024, 1024), seed=int(g["seed"])))
    ids = t(g["ids"]).long()
    yy, xx = np.mgrid[0:1024, 0:1024].astype(np.float32)
    gt = np.stack([((xx - 300) ** 2 + (yy - 340) ** 2 < 150 ** 2), ((xx - 700) ** 2 + (yy - 610) ** 2 < 220 ** 2)]).astype(np.float32)[:, None]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss, bce, dice = train_step_loss(m, x, ids, torch.ones_like(ids), (t(g["pts"]), t(g["lbl"])), t(gt))
    assert abs(loss.item() - float(g["loss"])) < 2e-5 * float(g["loss"]), (loss.item(), float(g["loss"]))
    loss.backward()
    torch.cuda.synchronize(); print(f"step (forward + backward, mfma_linear={mfma_linear}): {time.perf_counter() - t0:.3f} s")
    training.MFMA_LINEAR = True
    params = dict(m.named_parameters())
    names = [str(v) for v in g["names"]]
    assert {n.split(".")[0] for n in names} == {"vision_model", "mlp1", "mlp2", "prompt_encoder", "mask_decoder"}
    worst = (0.0, "")
    for n in names:
        ref = g["g:" + n].astype(np.float64)
        assert params[n].grad is not None, n
        full = params[n].grad.float().cpu().numpy().reshape(-1).astype(np.float64)
        got = full[::max(1, full.size // 512)]
        scale, diff = np.abs(ref).max(), np.abs(got - ref).max()
        assert diff < 1e-3 * scale + 1e-7, (n, diff, scale)
        if scale > 1e-6:
            worst = max(worst, (diff / scale, n))
        nref = float(g["n:" + n])
        assert abs(np.sqrt((full ** 2).sum()) - nref) < 1e-3 * nref + 1e-6, (n, np.sqrt((full ** 2).sum()), nref)
    assert all(p.grad is None for n, p in params.items() if n.startswith("language_model."))
    print(len(names), "gradients; worst relative error", worst)


def test_segmentation_loss_gradient_by_central_differences_on_another_prompt_shape():
    """A prompt shape no fixture has (ONE instance, three clicks of which one carries the padding label -1): the analytic gradient of a few
    scalar parameters -- llm_bias, llm_scale_factor, an entry of not_a_point_embed (reached through the -1 click AND the padding point), of a
    mask token, of a hypernetwork weight and of the first transposed convolution -- against central differences of the loss itself."""
    from ullsam_amd.training import segmentation_loss
    g = U.gold("train_slice")
    m = _ullsam_tiny(torch.float32)
    for n, p in m.named_parameters():
        p.requires_grad_(n.startswith(("prompt_encoder.", "mask_decoder.")))
    hid, img, _, gt = _inputs(g)
    pts = torch.tensor([[[400.0, 300.0], [900.0, 128.0], [10.0, 10.0]]], device=DEV)
    lbl = torch.tensor([[1, 0, -1]], device=DEV, dtype=torch.int32)
    gt = gt[:1]
    f = lambda: segmentation_loss(m, hid, img, (pts, lbl), gt)[0]
    loss = f()
    loss.backward()
    pe, md = m.prompt_encoder, m.mask_decoder
    probes = [(pe.llm_bias, 0), (pe.llm_scale_factor, 0), (pe.not_a_point_embed.weight, 17), (md.mask_tokens.weight, 5),
              (md.output_hypernetworks_mlps[0].layers[2].weight, 100), (md.output_upscaling[0].weight, 4321)]
    for p, i in probes:
        if p.numel() > 4096:   # large tensors: the entry with the largest gradient (a tiny one drowns in the fp32 rounding of the loss)
            i = int(p.grad.reshape(-1).abs().argmax())
        ana = float(p.grad.reshape(-1)[i])
        flat = p.data.reshape(-1)
        old = float(flat[i])
        eps = 2e-3
        with torch.no_grad():
            flat[i] = old + eps; lp = float(f().detach())
            flat[i] = old - eps; lm_ = float(f().detach())
            flat[i] = old
        num = (lp - lm_) / (2 * eps)
        assert abs(num - ana) < 5e-2 * abs(ana) + 4e-4, (tuple(p.shape), i, ana, num)   # fp32 loss: ~2e-7 of rounding over a 4e-3 step = 1e-4 on the quotient


@pytest.mark.parametrize("M,N,K,batch,a_t,b_t", [(196, 196, 80, 7, False, True), (196, 80, 196, 7, False, False), (300, 131, 1081, 3, True, False),
                                                     (1081, 128, 1081, 2, True, True), (64, 48, 16, 1, False, False), (257, 129, 33, 2, True, True),
                                                     (14, 80, 5600, 14, True, False), (5, 32, 65536, 2, False, True), (100, 64, 4100, 3, False, False)])
def test_train_matmul_on_the_matrix_pipe_equals_the_scalar_kernel(M, N, K, batch, a_t, b_t):
    """ullsam_train_matmul's MFMA form (v_mfma_f32_32x32x2_f32, exact fp32) against its one-output-per-thread form and float64, over the operand
    layouts the train step uses (row-major and transposed views, batch strides, ragged M / N / K tails, accumulate; the last three shapes
    take the split-k route: few output tiles under a long sum, partials added in order)."""
    from ullsam_amd import _lib
    from ullsam_amd.training import _mm
    rng = np.random.default_rng(M * 131 + N * 7 + K)
    A = torch.from_numpy(rng.standard_normal((batch, K, M) if a_t else (batch, M, K), dtype=np.float32)).to(DEV)
    B = torch.from_numpy(rng.standard_normal((batch, N, K) if b_t else (batch, K, N), dtype=np.float32)).to(DEV)
    C0 = torch.from_numpy(rng.standard_normal((batch, M, N), dtype=np.float32)).to(DEV)
    sa = (M * K, 1, M) if a_t else (M * K, K, 1)
    sb = (N * K, 1, K) if b_t else (N * K, N, 1)
    ref = (A.double().transpose(1, 2) if a_t else A.double()) @ (B.double().transpose(1, 2) if b_t else B.double())
    outs = []
    lib = _lib.load()
    for on in (1, 0):
        old = lib.ullsam_train_set_matmul_mfma(on)
        try:
            C = C0.clone()
            _mm(A, B, C, M, N, K, sa, sb, (M * N, N, 1), batch=batch, accumulate=False)
            Cacc = C0.clone()
            _mm(A, B, Cacc, M, N, K, sa, sb, (M * N, N, 1), batch=batch, accumulate=True)
            torch.cuda.synchronize()
        finally:
            lib.ullsam_train_set_matmul_mfma(old)
        outs.append(C)
        scale = float(ref.abs().max())
        assert float((C.double() - ref).abs().max()) < 2e-6 * scale * max(1.0, K ** 0.5 / 8), (on, M, N, K)
        assert float((Cacc.double() - ref - C0.double()).abs().max()) < 2e-6 * scale * max(1.0, K ** 0.5 / 8) + 1e-6
    assert float((outs[0] - outs[1]).abs().max()) < 1e-5 * float(ref.abs().max())
    # the bf16-rounding form (autocast's matmul): equal to the fp64 product of the bf16-rounded operands up to fp32 accumulation
    if M >= 64 and N >= 48:
        Cb = torch.empty_like(C0)
        _mm(A, B, Cb, M, N, K, sa, sb, (M * N, N, 1), batch=batch, bf16=True)
        Ar, Br = A.bfloat16().double(), B.bfloat16().double()
        refb = (Ar.transpose(1, 2) if a_t else Ar) @ (Br.transpose(1, 2) if b_t else Br)
        assert float((Cb.double() - refb).abs().max()) < 2e-6 * float(refb.abs().max()) * max(1.0, K ** 0.5 / 8)


@pytest.mark.parametrize("B,H,Sq,Sk,kw,causal,masked", [(2, 3, 196, 196, 14, -1, False), (1, 2, 70, 1081, 0, 0, True), (1, 2, 33, 4096, 64, -1, False),
                                                          (2, 2, 50, 300, 0, 5, True), (1, 1, 9, 2000, 0, -1, False)])
def test_attention_row_pass_register_form_equals_the_three_pass_form(B, H, Sq, Sk, kw, causal, masked):
    """ullsam_train_attn_rows with the row held in registers (Sk <= 4096: one wave per row up to 256 keys, four waves above) against its
    three-pass form: forward (softmax of S + decomposed bias + causal / padding masks, incl. a fully padded batch entry) and backward (dS and the
    bias-gradient rows), to fp32 rounding of the row sums."""
    from ullsam_amd import _lib, ops
    lib = _lib.load()
    rng = np.random.default_rng(Sq * 7 + Sk)
    t = lambda a: torch.from_numpy(a).to(DEV)
    S0 = t(rng.standard_normal((B * H, Sq, Sk), dtype=np.float32) * 3)
    dP0 = t(rng.standard_normal((B * H, Sq, Sk), dtype=np.float32))
    bh = bw = None
    if kw:
        bh, bw = t(rng.standard_normal((B * H, Sq, Sk // kw), dtype=np.float32)), t(rng.standard_normal((B * H, Sq, kw), dtype=np.float32))
    km = None
    if masked:
        km_np = np.ones((B, Sk), np.int32)
        km_np[0, :Sk // 3] = 0
        km = t(km_np)
    outs = []
    for on in (1, 0):
        old = lib.ullsam_train_set_rows_reg(on)
        try:
            P = S0.clone()
            _lib.call("ullsam_train_attn_rows", P.data_ptr(), None, ops._p(bh), ops._p(bw), None, None, ops._p(km), B, H, Sq, Sk, max(kw, 1), causal, 0,
                      torch.cuda.current_stream().cuda_stream)
            dS = dP0.clone()
            dbh = torch.full_like(bh, 7.0) if kw else None
            dbw = torch.full_like(bw, 7.0) if kw else None
            _lib.call("ullsam_train_attn_rows", P.data_ptr(), dS.data_ptr(), ops._p(bh), ops._p(bw), ops._p(dbh), ops._p(dbw), None, B, H, Sq, Sk, max(kw, 1),
                      causal, 1, torch.cuda.current_stream().cuda_stream)
            # and the fused use (have_p 0 with dP): S -> P and dP -> dS in one call
            P2, dS2 = S0.clone(), dP0.clone()
            _lib.call("ullsam_train_attn_rows", P2.data_ptr(), dS2.data_ptr(), ops._p(bh), ops._p(bw), ops._p(dbh.clone() if kw else None), ops._p(dbw.clone() if kw else None),
                      ops._p(km), B, H, Sq, Sk, max(kw, 1), causal, 0, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        finally:
            lib.ullsam_train_set_rows_reg(old)
        assert float((P2 - P).abs().max()) < 1e-6 and float((dS2 - dS).abs().max()) < 1e-5
        outs.append((P, dS, dbh, dbw))
    (P1, d1, h1, w1), (P0, d0, h0, w0) = outs
    assert abs(float(P1.sum()) - B * H * Sq) < 1e-3 * B * H * Sq
    assert float((P1 - P0).abs().max()) < 2e-6
    assert float((d1 - d0).abs().max()) < 2e-5 * max(1.0, float(d0.abs().max()))
    if kw:
        assert float((h1 - h0).abs().max()) < 2e-5 * max(1.0, float(h0.abs().max())) and float((w1 - w0).abs().max()) < 2e-5 * max(1.0, float(w0.abs().max()))


def test_train_step_module_under_ddp_world_1():
    """TrainStep wrapped in DistributedDataParallel (RCCL, world size 1: what one box offers): the hooks fire, the gradients equal the plain step's."""
    import os
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    from ullsam_amd.training import TrainStep, train_step_loss
    g = U.gold("train_step")
    t = lambda a: torch.from_numpy(a).to(DEV)
    x = t(U.rand_image((1, 3, 1024, 1024), seed=int(g["seed"])))
    ids = t(g["ids"]).long()
    yy, xx = np.mgrid[0:1024, 0:1024].astype(np.float32)
    gt = t(np.stack([((xx - 300) ** 2 + (yy - 340) ** 2 < 150 ** 2), ((xx - 700) ** 2 + (yy - 610) ** 2 < 220 ** 2)]).astype(np.float32)[:, None])
    m = _ullsam_tiny(torch.float32)
    for n, p in m.named_parameters():
        p.requires_grad_(not n.startswith("language_model."))
    loss, _, _ = train_step_loss(m, x, ids, torch.ones_like(ids), (t(g["pts"]), t(g["lbl"])), gt)
    loss.backward()
    ref = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    for p in m.parameters():
        p.grad = None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29581")
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        step = DistributedDataParallel(TrainStep(m), device_ids=[0], find_unused_parameters=True)
        loss2, _, _ = step(x, ids, torch.ones_like(ids), t(g["pts"]), t(g["lbl"]), gt)
        loss2.backward()
        torch.cuda.synchronize()
    finally:
        if own:
            dist.destroy_process_group()
    assert abs(float(loss2.detach()) - float(loss.detach())) < 1e-5 * abs(float(loss.detach()))
    for n, r in ref.items():
        got = dict(m.named_parameters())[n].grad
        assert got is not None and float((got - r).abs().max()) <= 1e-4 * float(r.abs().max()) + 1e-7, n


def test_bf16_model_runs_its_large_linears_on_bf16_gemms():
    """training.BF16_LINEAR (default): the large linears of a bf16 model (>= 256 rows, dimensions % 64: the ViT's qkv / proj / lin1 / lin2, patch
    embedding, neck, mlp1 / mlp2) run forward, dX and dW on the bf16 MFMA GEMM, and the score / probability products of the ViT's and the LLM's
    attention round their operands to bf16 -- what the reference's trainer computes (bf16 model under autocast(bf16), train_joint_v2.py:1665,1676).  Against the fp32-arithmetic route on the same bf16 weights: the loss within 2e-3, every
    gradient within bf16-activation noise of its tensor's scale (and the bf16 route really is taken: LinearBf16Fn nodes in the graph)."""
    from ullsam_amd import training
    g = U.gold("train_step")
    t = lambda a: torch.from_numpy(a).to(DEV)
    x = t(U.rand_image((1, 3, 1024, 1024), seed=int(g["seed"])))
    ids = t(g["ids"]).long()
    yy, xx = np.mgrid[0:1024, 0:1024].astype(np.float32)
    gt = t(np.stack([((xx - 300) ** 2 + (yy - 340) ** 2 < 150 ** 2), ((xx - 700) ** 2 + (yy - 610) ** 2 < 220 ** 2)]).astype(np.float32)[:, None])
    res = []
    old_switch = training.BF16_LINEAR
    try:
        for on in (True, False):
            training.BF16_LINEAR = on
            m = _ullsam_tiny(torch.bfloat16)
            for n, p in m.named_parameters():
                p.requires_grad_(not n.startswith("language_model."))
            loss, _, _ = training.train_step_loss(m, x, ids, torch.ones_like(ids), (t(g["pts"]), t(g["lbl"])), gt)
            if on:
                seen, stack, hit = set(), [loss.grad_fn], 0
                while stack:
                    f = stack.pop()
                    if f is None or f in seen:
                        continue
                    seen.add(f)
                    hit += "LinearBf16Fn" in type(f).__name__
                    stack.extend(nf for nf, _ in f.next_functions)
                assert hit >= 8, hit
            loss.backward()
            torch.cuda.synchronize()
            res.append((float(loss.detach()), {n: p.grad.float() for n, p in m.named_parameters() if p.grad is not None}))
    finally:
        training.BF16_LINEAR = old_switch
    (lb, gb), (lf, gf) = res
    assert abs(lb - lf) < 2e-3 * abs(lf), (lb, lf)
    assert set(gb) == set(gf)
    worst, worst_cos = (0.0, ""), (1.0, "")
    for n in gf:
        scale = float(gf[n].abs().max())
        if scale < 1e-6:
            continue
        e = float((gb[n] - gf[n]).abs().max()) / scale
        cos = float((gb[n] * gf[n]).sum() / (gb[n].norm() * gf[n].norm()))
        worst = max(worst, (e, n))
        worst_cos = min(worst_cos, (cos, n))
        # linears alone: worst 0.011; with the attention products in bf16 the relative-position tables, whose gradients are sums of dS = P (dP - sum P dP)
        # over thousands of (query, key) pairs with cancellation, move by up to 0.06 of their largest entry while keeping their direction
        # ... and a ReLU unit of the decoder's MLP whose pre-activation sits at zero may switch (one row of a lin1 gradient appears / disappears: 0.11 of the
        # tensor's largest entry at cosine 0.99995): the bound on single entries is loose, the bound on the tensor's direction is not
        assert e < 0.25 and cos > 0.995, (n, e, cos)
    print("bf16-GEMM route vs fp32-arithmetic route on the same bf16 weights: loss", lb, lf, "worst relative gradient difference", worst, "worst cosine", worst_cos)


def test_bf16_model_runs_the_frozen_llm_on_bf16_gemms():
    """A bf16 model's frozen LLM linears (wqkv, wo, w1, w3, w2) run on the bf16 MFMA GEMM, forward and dX -- the trainer's autocast semantics:
    activations rounded to bf16 at each GEMM, fp32 accumulation.  Against the same rounded weights held in fp32 (the fp32-arithmetic step): the
    hidden states and the gradients that pass through the LLM (mlp1, d loss / d vision features) agree to bf16-activation noise."""
    from ullsam_amd import ops
    from ullsam_amd.training import llm_image_hidden, segmentation_loss
    g = U.gold("train_llm_slice")
    rng = np.random.default_rng(int(g["seed"]))
    feat = rng.standard_normal((1, 256, 64, 64), dtype=np.float32)
    img = rng.standard_normal((1, 256, 64, 64), dtype=np.float32)
    t = lambda a: torch.from_numpy(a).to(DEV)
    ids = t(g["ids"]).long()
    yy, xx = np.mgrid[0:1024, 0:1024].astype(np.float32)
    gt = t(np.stack([((xx - 300) ** 2 + (yy - 340) ** 2 < 150 ** 2), ((xx - 700) ** 2 + (yy - 610) ** 2 < 220 ** 2)]).astype(np.float32)[:, None])
    res = []
    for widen in (False, True):
        m = _ullsam_tiny(torch.bfloat16)
        if widen:
            m = m.float()
        for n, p in m.named_parameters():
            p.requires_grad_(n.startswith("mlp1."))
        rows = ops.transpose(t(feat).reshape(1, 256, 4096), 1, 256, 4096).requires_grad_(True)
        hidden = llm_image_hidden(m, rows, ids, torch.ones_like(ids))
        loss, _, _ = segmentation_loss(m, hidden, t(img), (t(g["pts"]), t(g["lbl"])), gt)
        loss.backward()
        res.append((hidden.detach().float(), float(loss.detach()), rows.grad.float(), {n: p.grad.float() for n, p in m.named_parameters() if p.grad is not None}))
    (h16, l16, r16, g16), (h32, l32, r32, g32) = res
    rel = lambda a, b: float((a - b).abs().max()) / float(b.abs().max())
    assert float((h16 - h32).abs().max()) > 0, "the bf16 model must not have taken the fp32 GEMM path"
    assert rel(h16, h32) < 3e-2 and abs(l16 - l32) < 2e-3 * abs(l32), (rel(h16, h32), l16, l32)
    assert rel(r16, r32) < 6e-2, rel(r16, r32)
    for n in g32:
        assert rel(g16[n], g32[n]) < 6e-2, (n, rel(g16[n], g32[n]))


def _trainer_losses(pred, gt, smooth=1e-7):
    """The trainer's OWN loss code (torch; train_joint_v2.py:605-665, 774-812: BCEWithLogits and Dice per instance, batch means), as it runs
    unchanged on top of the modules' outputs."""
    bce = torch.nn.functional.binary_cross_entropy_with_logits(pred.flatten(2), gt.flatten(2), reduction="none").mean(-1)
    p = pred.sigmoid().flatten(2)
    t = gt.flatten(2)
    dice = 1 - (2 * (p * t).sum(-1) + smooth) / (p.sum(-1) + t.sum(-1) + smooth)
    return (bce + dice).mean(), bce.mean(), dice.mean()


def test_reference_trainer_step_runs_on_the_modules_own_forwards():
    """train_joint_v2.py:988-1100 line for line, WITHOUT the training.train_step_loss entry point: model.train(); outputs = model(pixel_values,
    input_ids, attention_mask, labels, output_hidden_states=True); image_embeddings = model.vision_model(pixel_values); model.prompt_encoder(
    points, boxes=None, masks=None, llm_hidden_states=outputs.hidden_states.repeat(bs, 1, 1, 1)); model.mask_decoder(...); the trainer's own
    F.interpolate and losses; loss = 0 * outputs.loss + seg_loss; backward.  In train() mode with gradients enabled the modules' forwards
    dispatch to the autograd graph over the HIP kernels (eval() / no_grad keeps the inference kernels).  Checked against the reference's own
    autograd (fixture train_step.npz): the loss and all 168 parameter gradients; and eval() mode still returns the inference path's tensors."""
    import torch.nn.functional as F
    g = U.gold("train_step")
    m = _ullsam_tiny(torch.float32)
    for n, p in m.named_parameters():
        p.requires_grad_(not n.startswith("language_model."))
    t = lambda a: torch.from_numpy(a).to(DEV)
    pixel_values = t(U.rand_image((1, 3, 1024, 1024), seed=int(g["seed"])))
    input_ids = t(g["ids"]).long()
    attention_mask = torch.ones_like(input_ids)
    labels = input_ids.clone()
    points, point_labels = t(g["pts"]), t(g["lbl"])
    yy, xx = np.mgrid[0:1024, 0:1024].astype(np.float32)
    masks = t(np.stack([((xx - 300) ** 2 + (yy - 340) ** 2 < 150 ** 2), ((xx - 700) ** 2 + (yy - 610) ** 2 < 220 ** 2)]).astype(np.float32)[:, None])
    # eval(): the inference kernels, no graph
    m.eval()
    out_eval = m(pixel_values=pixel_values, input_ids=input_ids, attention_mask=attention_mask, return_dict=True, use_cache=False, output_hidden_states=True)
    assert not out_eval.hidden_states.requires_grad and not m.vision_model(pixel_values).requires_grad
    # ---- the trainer's step (train_joint_v2.py:944, 988-1100)
    m.train()
    outputs = m(pixel_values=pixel_values, input_ids=input_ids, attention_mask=attention_mask, image_flags=None, labels=labels, return_dict=True,
                use_cache=False, img_context_token_id=92546, output_hidden_states=True)
    loss = outputs.loss
    assert loss is not None and torch.isfinite(loss) and loss.requires_grad      # differentiable, as the reference's (its other branch back-propagates it); here it enters as 0 * loss
    last_hidden_state = outputs.hidden_states
    assert last_hidden_state.shape == (1, 256, 64, 64) and last_hidden_state.requires_grad
    assert float((last_hidden_state.detach() - out_eval.hidden_states.float()).abs().max()) < 1e-3      # same values as the inference path
    image_embeddings = m.vision_model(pixel_values)
    image_pe = m.prompt_encoder.get_dense_pe().to(DEV)
    bs = points.shape[0]
    if last_hidden_state.shape[0] != bs:
        last_hidden_state = last_hidden_state.repeat(bs, 1, 1, 1)
    sparse_embeddings, dense_embeddings = m.prompt_encoder(points=(points, point_labels), boxes=None, masks=None, llm_hidden_states=last_hidden_state)
    low_res_masks, iou_predictions = m.mask_decoder(image_embeddings=image_embeddings, image_pe=image_pe, sparse_prompt_embeddings=sparse_embeddings,
                                                    dense_prompt_embeddings=dense_embeddings, multimask_output=False)
    assert low_res_masks.shape == (bs, 1, 256, 256) and iou_predictions.shape == (bs, 1)
    pred_masks = F.interpolate(low_res_masks, (m.vision_model.img_size, m.vision_model.img_size), mode="bilinear", align_corners=False)
    seg_loss, bce, dice = _trainer_losses(pred_masks, masks)
    loss = 0 * loss + seg_loss
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["loss"])) < 2e-5 * float(g["loss"]), (loss.item(), float(g["loss"]))
    params = dict(m.named_parameters())
    names = [str(v) for v in g["names"]]
    worst = (0.0, "")
    for n in names:
        ref = g["g:" + n].astype(np.float64)
        assert params[n].grad is not None, n
        full = params[n].grad.float().cpu().numpy().reshape(-1).astype(np.float64)
        got = full[::max(1, full.size // 512)]
        scale, diff = np.abs(ref).max(), np.abs(got - ref).max()
        assert diff < 1e-3 * scale + 1e-7, (n, diff, scale)
        if scale > 1e-6:
            worst = max(worst, (diff / scale, n))
    assert all(p.grad is None for n, p in params.items() if n.startswith("language_model."))
    print(len(names), "gradients through the modules' own forwards; worst relative error", worst)
    # validation as the trainer runs it (train_joint_v2.py:917: torch.no_grad()) stays on the inference kernels even in train() mode
    with torch.no_grad():
        assert not m.vision_model(pixel_values).requires_grad


def test_reference_trainer_step_under_ddp_as_the_trainer_wraps_it():
    """train_joint_v2.py:1755-1761 wraps the MODEL: DistributedDataParallel(model, device_ids, output_device, find_unused_parameters=True) +
    _set_static_graph(), then calls model(...) through DDP and model.module.vision_model / prompt_encoder / mask_decoder beside it
    (:1014-1050).  Same statements here under RCCL at world size 1 (what one box offers): loss and gradients equal the fixture's, over two
    iterations (the static graph is recorded in the first)."""
    import os
    import torch.distributed as dist
    import torch.nn.functional as F
    from torch.nn.parallel import DistributedDataParallel
    g = U.gold("train_step")
    m = _ullsam_tiny(torch.float32)
    for n, p in m.named_parameters():
        p.requires_grad_(not n.startswith("language_model."))
    t = lambda a: torch.from_numpy(a).to(DEV)
    pixel_values = t(U.rand_image((1, 3, 1024, 1024), seed=int(g["seed"])))
    input_ids = t(g["ids"]).long()
    points, point_labels = t(g["pts"]), t(g["lbl"])
    yy, xx = np.mgrid[0:1024, 0:1024].astype(np.float32)
    masks = t(np.stack([((xx - 300) ** 2 + (yy - 340) ** 2 < 150 ** 2), ((xx - 700) ** 2 + (yy - 610) ** 2 < 220 ** 2)]).astype(np.float32)[:, None])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29583")
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        model = DistributedDataParallel(m, device_ids=[0], output_device=0, find_unused_parameters=True)
        model._set_static_graph()
        model.train()
        for it in range(2):
            for p in m.parameters():
                p.grad = None
            outputs = model(pixel_values=pixel_values, input_ids=input_ids, attention_mask=torch.ones_like(input_ids), image_flags=None, labels=input_ids.clone(),
                            return_dict=True, use_cache=False, img_context_token_id=92546, output_hidden_states=True)
            last_hidden_state = outputs.hidden_states
            image_embeddings = model.module.vision_model(pixel_values)
            image_pe = model.module.prompt_encoder.get_dense_pe().to(DEV)
            bs = points.shape[0]
            if last_hidden_state.shape[0] != bs:
                last_hidden_state = last_hidden_state.repeat(bs, 1, 1, 1)
            sparse_embeddings, dense_embeddings = model.module.prompt_encoder(points=(points, point_labels), boxes=None, masks=None, llm_hidden_states=last_hidden_state)
            low_res_masks, _ = model.module.mask_decoder(image_embeddings=image_embeddings, image_pe=image_pe, sparse_prompt_embeddings=sparse_embeddings,
                                                         dense_prompt_embeddings=dense_embeddings, multimask_output=False)
            pred_masks = F.interpolate(low_res_masks, (1024, 1024), mode="bilinear", align_corners=False)
            seg_loss, _, _ = _trainer_losses(pred_masks, masks)
            loss = 0 * outputs.loss + seg_loss
            loss.backward()
            torch.cuda.synchronize()
            assert abs(loss.item() - float(g["loss"])) < 2e-5 * float(g["loss"]), (it, loss.item(), float(g["loss"]))
            params = dict(m.named_parameters())
            for n in [str(v) for v in g["names"]]:
                ref = g["g:" + n].astype(np.float64)
                assert params[n].grad is not None, (it, n)
                full = params[n].grad.float().cpu().numpy().reshape(-1).astype(np.float64)
                got = full[::max(1, full.size // 512)]
                assert np.abs(got - ref).max() < 1e-3 * np.abs(ref).max() + 1e-7, (it, n)
    finally:
        if own:
            dist.destroy_process_group()


def _ullsam_real_dims(dtype):
    """The composite at the bench configuration's HEAD DIMENSIONS: SAM ViT-B width (768 = 12 heads x 64; one windowed + one global block on the
    64 x 64 grid) and one 7B-shaped InternLM2 layer (hidden 4096, 32 heads / 8 KV heads x 128, intermediate 14336), filled like the fixtures."""
    from ullsam_amd.build_sam import _build_sam
    from ullsam_amd.modeling.configuration_internvl_chat import InternVLChatConfig
    from ullsam_amd.modeling.modeling_internvl_sam import InternVLSAMModel
    from tests.test_model_gpu import _fill_model_from_rule
    c = U.LLM_7B_L1
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with torch.device(DEV):
            sam = _build_sam(768, 2, 12, [1])
            cfg = InternVLChatConfig(vision_config={"architectures": ["SAM-ViT-B-16"]},
                                     llm_config=dict(architectures=["InternLM2ForCausalLM"], vocab_size=c["vocab"], hidden_size=c["hidden"],
                                                     intermediate_size=c["inter"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
                                                     num_key_value_heads=c["kv_heads"], bias=False, max_position_embeddings=32768,
                                                     rope_theta=c["rope_theta"], rms_norm_eps=c["eps"]),
                                     downsample_ratio=0.5, template="internlm2-chat", ps_version="v2", force_image_size=1024)
            m = InternVLSAMModel(cfg, vision_model=sam.image_encoder, prompt_encoder=sam.prompt_encoder, mask_decoder=sam.mask_decoder)
    finally:
        torch.set_default_dtype(old)
    return _fill_model_from_rule(m.to(DEV).to(dtype), 0)


def test_whole_train_step_gradients_at_real_head_dimensions():
    """The reference trainer's step (train_joint_v2.py:990-1100; fixture train_step_real.npz = the reference's own autograd) on a composite with the
    bench configuration's head dimensions -- ViT width 768 (12 heads x 64), a windowed and a GLOBAL block on the 64 x 64 grid (the matrix-form
    attention: 4096 x 4096 scores per head), one 7B-shaped InternLM2 layer (32 heads / 8 KV heads x 128, 14336 intermediate) with the MFMA linears
    on their real shapes -- through the modules' own forwards: the loss and every parameter gradient within 1e-3 of the tensor's largest entry."""
    import torch.nn.functional as F
    from ullsam_amd import training
    assert training.MFMA_LINEAR
    g = U.gold("train_step_real")
    m = _ullsam_real_dims(torch.float32)
    for n, p in m.named_parameters():
        p.requires_grad_(not n.startswith("language_model."))
    t = lambda a: torch.from_numpy(a).to(DEV)
    pixel_values = t(U.rand_image((1, 3, 1024, 1024), seed=int(g["seed"])))
    input_ids = t(g["ids"]).long()
    points, point_labels = t(g["pts"]), t(g["lbl"])
    yy, xx = np.mgrid[0:1024, 0:1024].astype(np.float32)
    masks = t(np.stack([((xx - 300) ** 2 + (yy - 340) ** 2 < 150 ** 2), ((xx - 700) ** 2 + (yy - 610) ** 2 < 220 ** 2)]).astype(np.float32)[:, None])
    m.train()
    torch.cuda.synchronize(); t0 = __import__("time").perf_counter()
    outputs = m(pixel_values=pixel_values, input_ids=input_ids, attention_mask=torch.ones_like(input_ids), return_dict=True, use_cache=False,
                output_hidden_states=True)
    image_embeddings = m.vision_model(pixel_values)
    bs = points.shape[0]
    sparse, dense = m.prompt_encoder(points=(points, point_labels), boxes=None, masks=None, llm_hidden_states=outputs.hidden_states.repeat(bs, 1, 1, 1))
    low, _ = m.mask_decoder(image_embeddings=image_embeddings, image_pe=m.prompt_encoder.get_dense_pe(), sparse_prompt_embeddings=sparse,
                            dense_prompt_embeddings=dense, multimask_output=False)
    pred = F.interpolate(low, (1024, 1024), mode="bilinear", align_corners=False)
    loss, bce, dice = _trainer_losses(pred, masks)
    loss.backward()
    torch.cuda.synchronize(); print(f"step at real head dimensions (forward + backward, fp32): {__import__('time').perf_counter() - t0:.3f} s, "
                                    f"peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB")
    assert abs(loss.item() - float(g["loss"])) < 5e-5 * float(g["loss"]), (loss.item(), float(g["loss"]))
    params = dict(m.named_parameters())
    names = [str(v) for v in g["names"]]
    assert {n.split(".")[0] for n in names} == {"vision_model", "mlp1", "mlp2", "prompt_encoder", "mask_decoder"}
    worst = (0.0, "")
    for n in names:
        ref = g["g:" + n].astype(np.float64)
        assert params[n].grad is not None, n
        full = params[n].grad.float().cpu().numpy().reshape(-1).astype(np.float64)
        got = full[::max(1, full.size // 512)]
        scale, diff = np.abs(ref).max(), np.abs(got - ref).max()
        assert diff < 1e-3 * scale + 1e-7, (n, diff, scale)
        if scale > 1e-6:
            worst = max(worst, (diff / scale, n))
        nref = float(g["n:" + n])
        assert abs(np.sqrt((full ** 2).sum()) - nref) < 1e-3 * nref + 1e-6, (n, np.sqrt((full ** 2).sum()), nref)
    assert all(p.grad is None for n, p in params.items() if n.startswith("language_model."))
    print(len(names), "gradients at real head dimensions; worst relative error", worst)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_two_runs_of_a_train_step_are_bit_equal(dtype):
    """No sum of the step depends on scheduling: column sums, LayerNorm parameter gradients, the loss sums and the scalar gradients go through
    per-block partials added in a fixed order, the bilinear upsample's and the table gathers' adjoints are gathers, split-k products reduce in
    order, and every attention runs in the matrix form (MATRIX_ATTN_FROM = 0).  Two runs of the whole step at the real head geometry give the
    same bits: the loss and every gradient."""
    from ullsam_amd import training
    assert training.MATRIX_ATTN_FROM == 0
    g = U.gold("train_step_real")
    t = lambda a: torch.from_numpy(a).to(DEV)
    m = _ullsam_real_dims(dtype)
    for n, p in m.named_parameters():
        p.requires_grad_(not n.startswith("language_model."))
    x = t(U.rand_image((1, 3, 1024, 1024), seed=int(g["seed"])))
    ids = t(g["ids"]).long()
    yy, xx = np.mgrid[0:1024, 0:1024].astype(np.float32)
    gt = t(np.stack([((xx - 300) ** 2 + (yy - 340) ** 2 < 150 ** 2), ((xx - 700) ** 2 + (yy - 610) ** 2 < 220 ** 2)]).astype(np.float32)[:, None])
    runs = []
    for it in range(2):
        for p in m.parameters():
            p.grad = None
        loss, _, _ = training.train_step_loss(m, x, ids, torch.ones_like(ids), (t(g["pts"]), t(g["lbl"])), gt)
        loss.backward()
        torch.cuda.synchronize()
        runs.append((loss.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    (l0, g0), (l1, g1) = runs
    assert torch.equal(l0, l1)
    assert set(g0) == set(g1) and len(g0) >= 60
    diff = [n for n in g0 if not torch.equal(g0[n], g1[n])]
    assert not diff, diff[:8]


def test_small_attention_kernel_route_equals_the_matrix_route():
    """MATRIX_ATTN_FROM > 0 sends attentions with Sq * Sk below it to the one-workgroup-per-query kernel (backward by atomics: faster for the
    decoder's 7-token self-attention, not bit-reproducible).  Same gradients as the matrix form to fp32 rounding."""
    from ullsam_amd import training
    g = U.gold("train_slice")
    hid, img, pts, gt = _inputs(g)
    res = []
    old = training.MATRIX_ATTN_FROM
    try:
        for thr in (0, 1 << 12):
            training.MATRIX_ATTN_FROM = thr
            m = _ullsam_tiny(torch.float32)
            for n, p in m.named_parameters():
                p.requires_grad_(n.startswith(("mlp2.", "prompt_encoder.", "mask_decoder.")))
            loss, _, _ = training.segmentation_loss(m, hid, img, pts, gt)
            loss.backward()
            res.append((float(loss.detach()), {n: p.grad for n, p in m.named_parameters() if p.grad is not None}))
    finally:
        training.MATRIX_ATTN_FROM = old
    (l0, g0), (l1, g1) = res
    assert abs(l0 - l1) < 1e-6 * abs(l0) and set(g0) == set(g1)
    for n in g0:
        assert float((g0[n] - g1[n]).abs().max()) <= 2e-5 * float(g0[n].abs().max()) + 1e-8, n


def test_box_prompts_gradients_equal_the_reference_autograd():
    """The trainer forwards `boxes=boxes` to the prompt encoder (train_joint_v2.py:975,1038,1057; prompt_encoder.py:96-103: two corner embeddings with
    point_embeddings[2] / [3], no pad point).  Fixture train_slice_box.npz = the REFERENCE's autograd of the decoder-side slice with a box per instance next to
    its clicks; here through the modules' own forwards in train() mode: loss to 1e-5 relative, every gradient within 1e-3 of the tensor's largest entry --
    the corner embeddings' among them."""
    import torch.nn.functional as F
    from ullsam_amd import training
    g = U.gold("train_slice_box")
    m = _ullsam_tiny(torch.float32)
    for n, p in m.named_parameters():
        p.requires_grad_(n.startswith(("mlp2.", "prompt_encoder.", "mask_decoder.")))
    m.train()
    hid, img, pts, gt = _inputs(g)
    boxes = torch.from_numpy(g["boxes"]).to(DEV)
    last = training._rows_to_nchw(training.dense_feature_rows(m, hid), 1, 64, 64).repeat(pts[0].shape[0], 1, 1, 1)       # text_aware_dense_feature, then train_joint_v2.py:1052-1054
    sp, de = m.prompt_encoder(points=pts, boxes=boxes, masks=None, llm_hidden_states=last)
    assert sp.shape == (2, 2 + 2, 256) and sp.requires_grad                                                          # two clicks + two corners, no pad point
    low, iou = m.mask_decoder(image_embeddings=img, image_pe=m.prompt_encoder.get_dense_pe(), sparse_prompt_embeddings=sp, dense_prompt_embeddings=de,
                              multimask_output=False)
    assert err_np(low.detach().cpu().numpy().reshape(-1)[::61], g["low_sample"]) < 1e-3 * max(1.0, float(np.abs(g["low_sample"]).max()))
    pred = F.interpolate(low, (1024, 1024), mode="bilinear", align_corners=False)
    loss, bce, dice = _trainer_losses(pred, gt)
    assert abs(loss.item() - float(g["loss"])) < 2e-5 * float(g["loss"]), (loss.item(), float(g["loss"]))
    loss.backward()
    params = dict(m.named_parameters())
    worst = (0.0, "")
    for n in (str(v) for v in g["names"]):
        ref = g["g:" + n].astype(np.float64)
        if params[n].grad is None:
            assert float(np.abs(ref).max()) == 0.0, n            # (the reference reports a zero gradient for not_a_point_embed: no pad point with boxes)
            continue
        full = params[n].grad.float().cpu().numpy().reshape(-1).astype(np.float64)
        got = full[::max(1, full.size // 2048)]
        scale, diff = np.abs(ref).max(), np.abs(got - ref).max()
        assert diff < 1e-3 * scale + 1e-7, (n, diff, scale)
        if scale > 1e-6:
            worst = max(worst, (diff / scale, n))
    for k in (2, 3):
        assert float(params[f"prompt_encoder.point_embeddings.{k}.weight"].grad.abs().max()) > 0
    print("box prompts: worst relative gradient error", worst)


def test_mask_prompt_gradients_equal_the_reference_autograd():
    """The last prompt kind: masks (prompt_encoder.py:54-62 mask_downscaling, :105-108 _embed_masks; the dense embedding IS the downscaled mask, :187-188).  Fixture
    train_mask_prompt.npz = the REFERENCE's PromptEncoder with a point and a mask per prompt, the scalar sum(dense * R) / numel for a seeded R, its autograd gradients with respect
    to every parameter of mask_downscaling and to the masks.  Here through PromptEncoder.forward in train() mode (training.mask_downscaling_forward: the two stride-2
    convolutions as Linears over 2x2 pixel blocks, LayerNorm2d and GELU on NHWC rows, the 1x1 convolution): forward to 1e-5, every gradient within 1e-3 of its tensor's largest entry."""
    g = U.gold("train_mask_prompt")
    m = _ullsam_tiny(torch.float32)
    pe = m.prompt_encoder
    for n, p in m.named_parameters():
        p.requires_grad_(n.startswith("prompt_encoder."))
    m.train()
    rng = np.random.default_rng(21)                                     # oracle/gen_golden.py::mask_prompt_inputs restated
    masks_np = (rng.standard_normal((2, 1, 256, 256), dtype=np.float32) * 2.0 + 0.25).astype(np.float32)
    R = rng.standard_normal((2, 256, 64, 64), dtype=np.float32)
    pts = torch.tensor([[[300.0, 340.0]], [[700.0, 610.0]]], device=DEV)
    lbl = torch.tensor([[1], [0]], dtype=torch.int32, device=DEV)
    masks = torch.from_numpy(masks_np).to(DEV).requires_grad_(True)
    sparse, dense = pe(points=(pts, lbl), boxes=None, masks=masks, llm_hidden_states=None)
    assert dense.shape == (2, 256, 64, 64) and dense.requires_grad
    assert err_np(dense.detach().cpu().numpy().reshape(-1)[::997], g["dense_sample"]) < 1e-5 * max(1.0, float(np.abs(g["dense_sample"]).max()))
    assert err_np(sparse.detach().cpu().numpy(), g["sparse"]) < 1e-5
    loss = (dense * torch.from_numpy(R).to(DEV)).sum() / dense.numel()
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * abs(float(g["loss"])) + 1e-9, (loss.item(), float(g["loss"]))
    loss.backward()
    params = dict(m.named_parameters())
    worst = (0.0, "")
    for n in (str(v) for v in g["names"]):
        ref = g["g:" + n].astype(np.float64)
        got = params[n].grad.float().cpu().numpy().reshape(-1).astype(np.float64)
        scale, diff = np.abs(ref).max(), np.abs(got - ref).max()
        assert diff < 1e-3 * scale + 1e-10, (n, diff, scale)
        worst = max(worst, (diff / scale, n))
    gm = masks.grad.cpu().numpy().reshape(-1)[::61].astype(np.float64)
    assert np.abs(gm - g["g_masks_sample"]).max() < 1e-3 * float(g["g_masks_absmax"]), np.abs(gm - g["g_masks_sample"]).max()
    print("mask prompts: worst relative gradient error", worst)
    # the inference path (eval mode, ops.mask_downscale) gives the same dense embedding
    m.eval()
    with torch.no_grad():
        _, dense_inf = pe(points=(pts, lbl), boxes=None, masks=masks.detach(), llm_hidden_states=None)
    assert float((dense_inf.float() - dense.detach()).abs().max()) < 1e-4 * max(1.0, float(dense.detach().abs().max()))


def err_np(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_lm_loss_of_the_composite_is_differentiable_into_mlp1(dtype):
    """the reference's trainer has a branch that back-propagates `outputs.loss` itself (train_joint_v2.py:1000: masks is None or
    use_llm_hidden_states False), into mlp1 through the frozen LLM.  `.loss` of the composite's forward in train() mode is the differentiable HIP
    cross entropy over the head's logits (training.LMLossFn): its value equals torch's cross entropy on the inference path's logits, its gradient with
    respect to the hidden states equals torch autograd's of the same head + loss, loss.backward() alone fills mlp1's gradients, and an incoming gradient
    of exactly zero (the segmentation branch's 0 * loss) costs no GEMM and leaves zeros."""
    import torch.nn.functional as F
    from ullsam_amd import training
    g = U.gold("train_step")
    m = _ullsam_tiny(dtype)
    for n, p in m.named_parameters():
        p.requires_grad_(not n.startswith("language_model."))
    t = lambda a: torch.from_numpy(a).to(DEV)
    pixel_values = t(U.rand_image((1, 3, 1024, 1024), seed=int(g["seed"]))).to(dtype)
    input_ids = t(g["ids"]).long()
    labels = input_ids.clone()
    labels[:, :30] = -100                                                                # ignored positions, as the trainer's prompt masking produces
    m.train()
    out = m(pixel_values=pixel_values, input_ids=input_ids, attention_mask=torch.ones_like(input_ids), labels=labels, return_dict=True, use_cache=False)
    assert out.loss.requires_grad
    # value: torch's cross entropy on the inference path's logits (fp32 accumulate either way)
    ref_val = F.cross_entropy(out.logits[..., :-1, :].reshape(-1, out.logits.shape[-1]).float(), labels[..., 1:].reshape(-1), ignore_index=-100)
    assert abs(out.loss.item() - ref_val.item()) < (1e-4 if dtype == torch.float32 else 2e-2) * abs(ref_val.item()), (out.loss.item(), ref_val.item())
    out.loss.backward()
    g1 = {n: p.grad.float().clone() for n, p in m.named_parameters() if n.startswith("mlp1.") and p.grad is not None}
    assert len(g1) >= 4 and all(torch.isfinite(v).all() and float(v.abs().max()) > 0 for v in g1.values()), list(g1)
    # gradient with respect to the hidden states against torch autograd of the same head + loss
    lm = m.language_model
    S, D = input_ids.shape[1], lm.output.weight.shape[1]
    gen = torch.Generator(device=DEV); gen.manual_seed(3)
    h = (torch.randn(1, S, D, device=DEV, generator=gen) * 0.7).requires_grad_(True)
    loss_h = training.lm_loss(lm, h, labels)
    (3.0 * loss_h).backward()
    h2 = h.detach().clone().requires_grad_(True)
    W = lm.output.weight.detach().float()
    x2 = h2[:, :-1].reshape(-1, D)
    if dtype == torch.bfloat16:
        x2 = x2.bfloat16().float()                                                       # the head GEMM rounds its activations to bf16 at the door
    ref = 3.0 * F.cross_entropy(x2 @ W.T, labels[:, 1:].reshape(-1), ignore_index=-100)
    ref.backward()
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    assert abs(loss_h.item() * 3.0 - ref.item()) < tol * abs(ref.item())
    scale = float(h2.grad.abs().max())
    assert float((h.grad - h2.grad).abs().max()) < (1e-3 if dtype == torch.float32 else 3e-2) * scale, (float((h.grad - h2.grad).abs().max()), scale)
    assert float(h.grad[:, -1].abs().max()) == 0.0                                       # the last position's logits are never used
    # 0 * loss: exact zeros, no GEMMs
    h3 = h.detach().clone().requires_grad_(True)
    (0 * training.lm_loss(lm, h3, labels)).backward()
    assert float(h3.grad.abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("rows,D", [(1081, 4096), (37, 96), (64, 2048), (9, 4), (300, 1284)])
def test_rmsnorm_backward_against_torch_autograd(rows, D):
    """training.RMSNormFn (InternLM2RMSNorm, modeling_internlm2.py:75-89): dx and dw against torch's autograd on the same formula in float64 -- the row-in-registers kernel
    (one workgroup per row, float4 fetches) at the 7B width, at widths that leave most of its threads idle and at one that is not a multiple of the workgroup's stride."""
    from ullsam_amd import training as T
    g = torch.Generator(device=DEV); g.manual_seed(rows + D)
    x = torch.randn(rows, D, device=DEV, generator=g, requires_grad=True)
    w = (1.0 + 0.1 * torch.randn(D, device=DEV, generator=g)).requires_grad_(True)
    go = torch.randn(rows, D, device=DEV, generator=g)
    y = T.RMSNormFn.apply(x, w, 1e-6)
    y.backward(go)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    yd = xd * torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + 1e-6) * wd
    yd.backward(go.double())
    assert float((y.detach().double() - yd.detach()).abs().max()) < 1e-5 * max(1.0, float(yd.detach().abs().max()))
    assert float((x.grad.double() - xd.grad).abs().max()) < 2e-5 * max(1.0, float(xd.grad.abs().max()))
    assert float((w.grad.double() - wd.grad).abs().max()) < 2e-5 * max(1.0, float(wd.grad.abs().max()))
