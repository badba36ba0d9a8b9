"""CPU-only checks: the C-ABI library loads and exports what include/ullsam_hip.h declares, the host mirror of the
reference interface (state_dict layouts, configs, packing), no CPU fallback, and the N>1 path under gloo (world_size 2)."""
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch

from oracle import ullsam_oracle as O
from tests import util as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ullsam_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "ullsam_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ullsam_[a-z0-9_]+)\s*\(", src)))


def test_c_abi_exports_every_declared_symbol(lib):
    from ullsam_amd import _lib
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ullsam_hip.h but not exported"
    bound = set(_lib.SIGNATURES) | set(_lib.PLAIN)
    assert bound == set(syms), (bound ^ set(syms))
    from ullsam_amd import _lib
    hdr_v = int(re.search(r'#define ULLSAM_ABI_VERSION (\d+)', open(os.path.join(ROOT, "include", "ullsam_hip.h")).read()).group(1))
    assert lib.ullsam_abi_version() == _lib.ABI_VERSION == hdr_v
    assert lib.ullsam_last_error_string() is not None
    # the measurement knobs are host-side state: settable without a GPU, and an unknown key is an error with a message, not a crash
    assert lib.ullsam_set_gemm_tuning(1, 7) == 0 and lib.ullsam_set_gemm_tuning(0, 4) == 0
    assert lib.ullsam_set_gemm_tuning(99, 0) != 0 and b"unknown key" in lib.ullsam_last_error_string()


def test_c_abi_argument_counts_match_header():
    from ullsam_amd import _lib
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ullsam_hip.h")).read(), flags=re.S)
    for name, args in _lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\((.*?)\)\s*;" % name, src, flags=re.S)
        assert m, name
        n = len([a for a in m.group(1).split(",") if a.strip()])
        assert n == len(args), (name, n, len(args))


def test_no_cpu_fallback_and_no_oracle_in_product():
    from ullsam_amd import _lib, ops
    with pytest.raises((_lib.UllsamError, TypeError, ValueError)):
        ops.gemm(torch.zeros(4, 64), torch.zeros(8, 64))  # CPU tensors: must raise, never compute
    for dp, _, files in os.walk(os.path.join(ROOT, "ullsam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f"{f} must not use the oracle"
                assert "/root/reference" not in txt


def test_state_dict_layouts_match_reference():
    from ullsam_amd.build_sam import sam_model_registry
    sam = sam_model_registry["vit_b"]()
    exp = {}
    exp.update(O.vit_shapes(prefix="image_encoder."))
    exp.update(O.prompt_encoder_shapes(prefix="prompt_encoder."))
    exp.update(O.mask_decoder_shapes(prefix="mask_decoder."))
    got = {k: tuple(v.shape) for k, v in sam.state_dict().items()}
    assert got == {k: tuple(v) for k, v in exp.items()}
    assert set(sam_model_registry) == {"default", "vit_h", "vit_l", "vit_b"}
    assert sam.mask_threshold == 0.0 and sam.image_format == "RGB" and sam.image_encoder.img_size == 1024
    assert not sam.training


def test_composite_state_dict_and_attributes():
    from tests.test_model_gpu import _ullsam_tiny  # noqa: F401  (constructor only; no GPU touched)
    from ullsam_amd.build_sam import _build_sam
    from ullsam_amd.modeling.configuration_internvl_chat import InternVLChatConfig
    from ullsam_amd.modeling.modeling_internvl_sam import InternVLSAMModel
    c = U.LLM_TINY
    sam = _build_sam(128, 2, 2, [1])
    cfg = InternVLChatConfig(llm_config=dict(architectures=["InternLM2ForCausalLM"], vocab_size=c["vocab"], hidden_size=c["hidden"],
                                             intermediate_size=c["inter"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
                                             num_key_value_heads=c["kv_heads"], bias=False), ps_version="v2", template="internlm2-chat")
    m = InternVLSAMModel(cfg, vision_model=sam.image_encoder, prompt_encoder=sam.prompt_encoder, mask_decoder=sam.mask_decoder)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == {k: tuple(v.shape) for k, v in U.ullsam_tiny_params().items()}
    assert m.img_context_token_id == 92546 and m.num_image_token == 1024.0 and isinstance(m.num_image_token, float)
    assert m.template == "internlm2-chat" and m.system_message
    for a in ("vision_model", "prompt_encoder", "mask_decoder", "language_model", "mlp1", "mlp2"):
        assert hasattr(m, a)


def test_config_defaults_follow_reference():
    from ullsam_amd.modeling.configuration_internlm2 import InternLM2Config
    from ullsam_amd.modeling.configuration_internvl_chat import InternVLChatConfig
    c = InternLM2Config()
    assert (c.vocab_size, c.hidden_size, c.intermediate_size, c.num_hidden_layers, c.num_attention_heads) == (103168, 4096, 11008, 32, 32)
    assert c.num_key_value_heads == 32 and c.rms_norm_eps == 1e-6 and c.bias is True and c.rope_theta == 10000
    with pytest.raises(ValueError):
        InternLM2Config(rope_scaling={"type": "bogus", "factor": 2.0})
    v = InternVLChatConfig()
    assert v.downsample_ratio == 0.5 and v.ps_version == "v1" and v.select_layer == -1
    with pytest.raises(ValueError):
        InternVLChatConfig(llm_config={"architectures": ["Nope"]})


def test_weight_packing():
    from ullsam_amd.packing import pack_conv3x3, pack_convT_k2s2, pack_w13
    rng = np.random.default_rng(0)
    w1, w3 = rng.standard_normal((128, 16), dtype=np.float32), rng.standard_normal((128, 16), dtype=np.float32)
    p = pack_w13(torch.from_numpy(w1), torch.from_numpy(w3)).numpy()
    for t in range(2):
        assert (p[t * 128:t * 128 + 64] == w1[t * 64:(t + 1) * 64]).all() and (p[t * 128 + 64:(t + 1) * 128] == w3[t * 64:(t + 1) * 64]).all()
    # conv3x3 pack == the oracle's im2col column order
    x = rng.standard_normal((1, 5, 6, 8), dtype=np.float32)
    w = rng.standard_normal((4, 8, 3, 3), dtype=np.float32)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    cols = np.concatenate([xp[:, ky:ky + 5, kx:kx + 6] for ky in range(3) for kx in range(3)], -1)
    y = cols @ pack_conv3x3(torch.from_numpy(w)).numpy().T
    assert np.abs(y - O.conv3x3_nhwc(x, w)).max() < 1e-5
    # convT pack: GEMM columns (ky, kx, co) reproduce ConvTranspose2d(k=2, s=2)
    xi = rng.standard_normal((1, 8, 3, 3), dtype=np.float32)
    wt, bt = rng.standard_normal((8, 4, 2, 2), dtype=np.float32), rng.standard_normal(4, dtype=np.float32)
    W, Bv = pack_convT_k2s2(torch.from_numpy(wt), torch.from_numpy(bt))
    g = xi.transpose(0, 2, 3, 1).reshape(9, 8) @ W.numpy().T + Bv.numpy()  # [pix, (ky,kx,co)]
    ref = O._conv_transpose_k2s2(xi, wt, bt)
    for pix in range(9):
        y0, x0 = divmod(pix, 3)
        for ky in range(2):
            for kx in range(2):
                assert np.abs(g[pix, (ky * 2 + kx) * 4:(ky * 2 + kx + 1) * 4] - ref[0, :, 2 * y0 + ky, 2 * x0 + kx]).max() < 1e-5


def test_shard_range_partitions():
    from ullsam_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 32, 33):
        for ws in (1, 2, 3, 8):
            spans = [shard_range(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _fake_step(images: torch.Tensor):
    """Stand-in for the per-image path (images are independent units): a deterministic per-image function with the real
    output shapes' structure (fp32 logits, u8 masks, int64 token ids)."""
    low = torch.sin(images * 3.0 + images.flip(-1)).reshape(-1, 1, 8, 8)          # per-image, also for a shard of zero images
    mk = (low > 0.1).to(torch.uint8).repeat(1, 1, 2, 2)
    tok = (images.reshape(images.shape[0], 64)[:, :5] * 1000).long()
    return low, mk, tok


def _gloo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from ullsam_amd import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_total = 5  # ragged: rank 0 gets 3 images, rank 1 gets 2
        a, b = parallel.shard_range(n_total, rank, world)
        low = torch.stack([torch.full((1, 4, 4), float(i)) for i in range(a, b)])
        mk = (low > 1.5).to(torch.uint8)
        tok = torch.arange(a, b).reshape(-1, 1).repeat(1, 3)
        glow, gmk, gtok = parallel.gather_mask_results(low, mk, tok)
        ok = glow.shape[0] == n_total and all(float(glow[i, 0, 0, 0]) == i for i in range(n_total))
        ok = ok and gmk.sum().item() == 3 * 16 and gtok[:, 0].tolist() == list(range(n_total))
        ok = ok and glow.dtype == torch.float32 and gmk.dtype == torch.uint8 and gtok.dtype == torch.int64
        eq = parallel.all_gather_rows(torch.full((2, 2), float(rank)), counts=[2, 2])  # equal shards: single collective
        ok = ok and eq[:, 0].tolist() == [0.0, 0.0, 1.0, 1.0]
        # DP equivalence (distributed level): sharded run + gather == the unsharded run, bit for bit,
        # for an even and a ragged split and for fewer images than ranks (n_img = 1: rank 1 holds ZERO rows and must still take
        # part in the collective instead of raising before it), through the blocking and the overlapped (async) exchange
        for n_img in (4, 7, 1):
            imgs = torch.from_numpy(np.random.default_rng(n_img).random((n_img, 8, 8), dtype=np.float32))
            full = _fake_step(imgs)
            a, b = parallel.shard_range(n_img, rank, world)
            mine = _fake_step(imgs[a:b])
            counts = [parallel.shard_range(n_img, r, world)[1] - parallel.shard_range(n_img, r, world)[0] for r in range(world)]
            got = parallel.gather_mask_results(*mine, counts=counts)
            pend = parallel.gather_mask_results_async(mine[0], mine[1], None, counts=counts)
            got_async = pend.wait()
            ok = ok and all(torch.equal(g, f) for g, f in zip(got, full))
            ok = ok and torch.equal(got_async[0], full[0]) and torch.equal(got_async[1], full[1]) and got_async[2] is None
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_gather_is_rank_ordered():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def test_packed_gather_without_process_group_is_identity():
    from ullsam_amd import parallel
    low = torch.randn(3, 1, 4, 4)
    mk = (low > 0).to(torch.uint8)
    a, b, c = parallel.gather_mask_results(low, mk, None)
    assert torch.equal(a, low) and torch.equal(b, mk) and c is None


def _bfs_regions(work):
    """8-connected components by flood fill (the tests' own labeller), labels in raster order of each region's first pixel."""
    H, W = work.shape
    lab = np.zeros((H, W), np.int32)
    n = 0
    for y in range(H):
        for x in range(W):
            if work[y, x] and lab[y, x] == 0:
                n += 1
                stack = [(y, x)]
                lab[y, x] = n
                while stack:
                    cy, cx = stack.pop()
                    for dy in (-1, 0, 1):
                        for dx in (-1, 0, 1):
                            yy, xx = cy + dy, cx + dx
                            if 0 <= yy < H and 0 <= xx < W and work[yy, xx] and lab[yy, xx] == 0:
                                lab[yy, xx] = n
                                stack.append((yy, xx))
    return lab, n


def _remove_small_regions_oracle(mask, area_thresh, mode):
    """utils/amg.py:267-291 restated over the flood-fill labeller."""
    correct_holes = mode == "holes"
    regions, n = _bfs_regions(correct_holes ^ mask)
    sizes = np.array([(regions == i).sum() for i in range(1, n + 1)])
    small = [i + 1 for i, s_ in enumerate(sizes) if s_ < area_thresh]
    if not small:
        return mask, False
    fill = [0] + small
    if not correct_holes:
        fill = [i for i in range(n + 1) if i not in fill]
        if not fill:
            fill = [int(np.argmax(sizes)) + 1]
    return np.isin(regions, fill), True


def test_remove_small_regions_and_coco_rle():
    from ullsam_amd.utils import amg as A
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:48, 0:64]
    blob = ((yy - 20) ** 2 + (xx - 30) ** 2) < 150
    cases = [rng.random((48, 64)) > 0.55, blob | (rng.random((48, 64)) > 0.97), blob & ~(rng.random((48, 64)) > 0.9),
             np.zeros((8, 8), bool), np.ones((8, 8), bool), np.eye(9, dtype=bool)]   # eye: diagonal pixels are ONE 8-connected region
    for m in cases:
        for mode in ("islands", "holes"):
            for thresh in (1, 6, 40, 10 ** 6):
                got, ch = A.remove_small_regions(m, thresh, mode)
                ref, ch_ref = _remove_small_regions_oracle(m, thresh, mode)
                assert ch == ch_ref and np.array_equal(got, ref), (mode, thresh)
    # COCO compressed RLE: round trip through the decoder, negative differences and multi-group values included
    for shape in ((37, 53), (5, 4)):
        m = rng.random(shape) > 0.5
        rle = A.mask_to_rle_numpy(m) if hasattr(A, "mask_to_rle_numpy") else None
        flat = m.T.reshape(-1)
        change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
        counts = np.diff(np.concatenate([[0], change, [flat.size]])).tolist()
        if flat[0]:
            counts = [0] + counts
        enc = A.coco_encode_rle({"size": list(shape), "counts": counts})
        assert isinstance(enc["counts"], str) and all(48 <= ord(c) < 48 + 64 for c in enc["counts"])
        assert A.coco_decode_rle(enc)["counts"] == counts
        assert np.array_equal(A.rle_to_mask({"size": list(shape), "counts": A.coco_decode_rle(enc)["counts"]}), m)
    enc = A.coco_encode_rle({"size": [2048, 2048], "counts": [0, 5, 100000, 3, 2000000, 7, 2094289]})
    assert enc["counts"][:2] == "05" and A.coco_decode_rle(enc)["counts"] == [0, 5, 100000, 3, 2000000, 7, 2094289]


def test_checkpoint_loading_conventions(tmp_path):
    """Reference-layout checkpoints load without key surgery: {"model": sd} torch files, DDP 'module.' prefixes, and
    InternLM2 safetensors re-prefixed with language_model. (train_joint_v2.py:1466-1555)."""
    from safetensors.torch import save_file
    from ullsam_amd import checkpoint
    from ullsam_amd.build_sam import _build_sam
    from ullsam_amd.modeling.configuration_internvl_chat import InternVLChatConfig
    from ullsam_amd.modeling.modeling_internvl_sam import InternVLSAMModel
    c = U.LLM_TINY
    llm_cfg = dict(architectures=["InternLM2ForCausalLM"], vocab_size=128, hidden_size=c["hidden"], intermediate_size=c["inter"],
                   num_hidden_layers=1, num_attention_heads=c["heads"], num_key_value_heads=c["kv_heads"], bias=False)

    def make():
        sam = _build_sam(128, 1, 2, [0])
        return InternVLSAMModel(InternVLChatConfig(llm_config=dict(llm_cfg), ps_version="v2"), vision_model=sam.image_encoder,
                                prompt_encoder=sam.prompt_encoder, mask_decoder=sam.mask_decoder)

    src, dst = make(), make()
    with torch.no_grad():
        for p in src.parameters():
            p.normal_()
    sd = src.state_dict()
    torch.save({"model": {"module." + k: v for k, v in sd.items()}, "epoch": 3}, tmp_path / "final.pt")
    missing, unexpected = checkpoint.load_ullsam_checkpoint(dst, str(tmp_path / "final.pt"))
    assert not missing and not unexpected
    assert all(torch.equal(a, b) for a, b in zip(src.state_dict().values(), dst.state_dict().values()))
    # a checkpoint exactly as the reference's save_checkpoint writes it (train_joint_v2.py:1254-1263): optimizer + scheduler
    # state and the argparse.Namespace of the run (with a pathlib path inside) next to the weights
    import argparse, pathlib
    opt = torch.optim.AdamW([p for p in src.parameters()][:2], lr=1e-4)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s_: 1.0)
    torch.save({"model": sd, "optimizer": opt.state_dict(), "scheduler": sched.state_dict(), "epoch": 23, "step": 100,
                "args": argparse.Namespace(save_dir=pathlib.Path("out"), training_mode="all", lr=1e-4)}, tmp_path / "ref_shaped.pt")
    dst3 = make()
    missing, unexpected = checkpoint.load_ullsam_checkpoint(dst3, str(tmp_path / "ref_shaped.pt"))
    assert not missing and not unexpected
    assert all(torch.equal(a, b) for a, b in zip(src.state_dict().values(), dst3.state_dict().values()))
    # safetensors with bare InternLM2 keys + foreign InternViT keys that must be ignored + one tensor of the wrong shape that must
    # be skipped rather than raise (the reference's shape filter, train_joint_v2.py:1534-1545)
    dst2 = make()
    llm = {k[len("language_model."):]: v.contiguous() for k, v in sd.items() if k.startswith("language_model.")}
    llm["vision_model.embeddings.foo"] = torch.zeros(2)
    llm["model.norm.weight"] = torch.zeros(7)
    save_file(llm, str(tmp_path / "model.safetensors"))
    missing, unexpected = checkpoint.load_llm_safetensors(dst2, str(tmp_path))
    assert not unexpected and [k for k in missing if k.startswith("language_model.")] == ["language_model.model.norm.weight"]
    assert torch.equal(dst2.language_model.output.weight, src.language_model.output.weight)


def test_chat_builds_the_reference_prompt():
    """InternVLSAMModel.chat (modeling_internvl_sam.py:272-335): the query string handed to the tokenizer equals what the reference
    builds with its "internlm2-chat" conversation template (tests/golden/chat_prompt.json, produced by the reference's own code);
    eos = the template separator; the response is cut at the separator; history is appended."""
    import json
    import types
    from ullsam_amd.modeling.modeling_internvl_sam import InternVLSAMModel
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "chat_prompt.json")))
    for c in cases:
        seen = {}

        class Tok:
            def convert_tokens_to_ids(self, t):
                return {"<IMG_CONTEXT>": 92546, "<|im_end|>": 92542}[t]

            def __call__(self, query, return_tensors="pt"):
                seen["query"] = query
                return {"input_ids": torch.zeros((1, 3), dtype=torch.long), "attention_mask": torch.ones((1, 3), dtype=torch.long)}

            def batch_decode(self, out, skip_special_tokens=True):
                return ["  the answer<|im_end|> trailing"]

        def fake_generate(**kw):
            seen["gen"] = kw
            return torch.zeros((1, 2), dtype=torch.long)

        me = types.SimpleNamespace(system_message=c["system"], num_image_token=c["num_image_token"], device="cpu", generate=fake_generate,
                                   template="internlm2-chat")
        n_img = sum(c["num_patches_list"])
        pv = torch.zeros((n_img, 3, 8, 8)) if n_img else None
        hist = [tuple(h) for h in c["history"]] if c["history"] else None
        resp, new_hist = InternVLSAMModel.chat(me, Tok(), pv, c["question"], {"max_new_tokens": 4, "do_sample": False}, history=hist,
                                               return_history=True, num_patches_list=c["num_patches_list"] or None)
        assert seen["query"] == c["query"]
        assert seen["gen"]["eos_token_id"] == 92542 and me.img_context_token_id == 92546
        assert resp == "the answer" and new_hist[-1][1] == "the answer" and len(new_hist) == len(c["history"] or []) + 1


def test_bench_host_helpers(tmp_path, monkeypatch):
    """bench.py's host-side pieces that need no GPU: the synthetic prompt layout (one contiguous run of 1024 <IMG_CONTEXT> ids
    bracketed by <img> / </img>, S = 1081 for the default text lengths), and the traffic record being refused when it was
    measured on other kernel sources."""
    import json
    import bench
    ids = bench.make_input_ids(20, 34, seed=1, batch=3)
    assert ids.shape == (3, 1081) and ids.dtype == np.int64
    for row in ids:
        pos = np.flatnonzero(row == 92546)
        assert len(pos) == 1024 and pos[-1] - pos[0] == 1023 and row[pos[0] - 1] == 92544 and row[pos[-1] + 1] == 92545 and row[0] == 1
    assert np.array_equal(ids, O.make_input_ids(20, 34, seed=1, batch=3))       # same layout as the oracle's generator
    from ullsam_amd import build as B
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    assert bench.traffic_record()[0] is None                                      # no record
    json.dump({"csrc_digest": "0" * 64, "hbm_bytes_per_launch": 1.0}, open(tmp_path / "profiles" / "r03_pmc_bench_traffic.json", "w"))
    val, why = bench.traffic_record()
    assert val is None and "stale" in why
    json.dump({"csrc_digest": B._digest(), "hbm_bytes_per_launch": 123.4}, open(tmp_path / "profiles" / "r03_pmc_bench_traffic.json", "w"))
    assert bench.traffic_record()[0] == 123
    # the digest names the kernel sources and flags, not the directory the tree is checked out in (the GPU box runs from a scratch path)
    d0 = B._digest()
    monkeypatch.setattr(B, "FLAGS", B.FLAGS[:-1] + ["/somewhere/else/include"])
    assert B._digest() == d0


def test_library_binds_the_hip_runtime_torch_ships():
    """`_lib.load()` before `import torch` (what `__graft_entry__.build()` followed by `smoke()` does in one process) must still leave ONE
    HIP runtime in the process -- torch's: with /opt/rocm's libamdhip64 bound first, the first kernel launch on a GPU box fails with
    "no ROCm-capable device is detected"."""
    import subprocess, sys
    code = ("from ullsam_amd import _lib; _lib.load(); import sys; assert 'torch' in sys.modules; "
            "hip = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}); print(hip); "
            "assert len(hip) == 1 and '/torch/lib/' in hip[0], hip")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout + r.stderr


def test_bench_gpus2_control_path_runs_end_to_end_in_stub_mode():
    """`python bench.py --gpus 2 --stub`: the parent (WORLD_SIZE unset) spawns `torch.distributed.run` with two ranks through the same
    `_spawn_ranks` route the real bench takes and never touches a GPU; the ranks assert the world size, run warm-up + timed steps between
    barriers with the one-in-flight packed gather, take the max over ranks, and rank 0 prints ONE JSON line.  (What cannot be run without
    GPUs -- RCCL itself -- is the only difference from the real N > 1 launch.)"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--steps", "3", "--warmup", "1", "--batch", "3"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["stub"] is True
    assert len(rec["config"]["ranks"]) == 2 and rec["config"]["ranks"][1].startswith("rank 1")
    assert rec["config"]["global_batch"] == 6 and rec["gathered_rows"] == 6
    # a launcher that sets WORLD_SIZE to something else than --gpus is refused before any work
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=ROOT, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr)


def test_bench_init_rule_is_the_fixture_filler_rule():
    """bench.py fills its model by ullsam_amd.utils.synthetic.fixture_param (rule table: param_init_rule); the reference-generated fixtures
    (incl. the full-depth one) were filled by oracle.fill_param.  Same (mean, std) for every parameter of the composite model, so the bench's
    `mask_iou_vs_fp32` is measured on the weight statistics the full-depth parity fixture pins."""
    from ullsam_amd.utils.synthetic import microscopy_batch, param_init_rule
    P = {}
    P.update(O.vit_shapes(embed_dim=128, depth=2, num_heads=2, global_attn_indexes=(1,), prefix="vision_model."))
    P.update(O.prompt_encoder_shapes(prefix="prompt_encoder."))
    P.update(O.mask_decoder_shapes(prefix="mask_decoder."))
    P.update(O.internlm2_shapes(256, 2, 2, 1, 512, 1000, prefix="language_model."))
    P.update(O.projector_shapes(256))
    from ullsam_amd.utils.synthetic import fixture_param
    import bench
    for k, shp in P.items():
        (m0, s0), (m1, s1) = O.fill_rule(k, shp), param_init_rule(k, shp)
        assert abs(m0 - m1) < 1e-6 and abs(s0 - s1) < 1e-6 * max(1.0, s1), (k, shp, (m0, s0), (m1, s1))
        assert np.array_equal(O.fill_param(k, shp, 0), fixture_param(k, shp, 0)), k      # bench.py's weights ARE the fixtures' weights, bit for bit
    # bench.py's default tiles on rank 0 are the tiles of the full-depth fixture (its masks are scored against the reference's there)
    g = U.gold("full_depth")
    assert [int(v) for v in g["tile_seeds"]] == list(bench.FIXTURE_TILE_SEEDS) == bench.tile_seeds(0, 4)
    assert bench.tile_seeds(1, 4) != bench.tile_seeds(0, 4) and len(set(bench.tile_seeds(0, 8) + bench.tile_seeds(1, 8))) == 16
    assert sum(0.3 <= float(g[f"mask_fill_{i}"]) <= 0.7 for i in range(4)) >= 2
    # the synthetic tile: deterministic, two intensity populations, the click lies inside a cell
    a, pa = microscopy_batch([3])
    b, pb = microscopy_batch([3])
    assert np.array_equal(a, b) and np.array_equal(pa, pb) and a.shape == (1, 3, 1024, 1024) and a.dtype == np.float32
    assert 0.2 < float((a[0, 0] > 0.45).mean()) < 0.7
    x, y = int(pa[0, 0, 0]), int(pa[0, 0, 1])
    assert a[0, 0, y, x] > 0.45


def test_hot_kernels_compile_without_spills():
    """The kernels the bench step, the decode step and (round 5) the mask decoder / automatic mask generator spend their time in must compile with no spilled registers and no scratch: a spilled
    pointer behind an LDS-DMA request waits for the DMA, and spills inside a counted-wait loop serialise it.  Read from
    the code objects of the SHIPPED library (tools/kernel_resources.py), so a compiler or source change that starts spilling fails here."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as KR
    ks = KR.kernels()
    assert len(ks) > 100, len(ks)
    hot = {n: r for n, r in ks.items() if any(re.search(h, n) for h in KR.HOT_BF16)}
    assert len(hot) >= 12, sorted(hot)
    for must in ("gemm_ring8_kernel", "causal128_attn_kernel", "win14_attn_kernel", "win14r_attn_kernel", "gemm_skinny", "norm_block_kernel", "i2t_block_kernel", "kv_proj_kernel", "up1_ln_gelu_kernel",
                 "up2_hyper_kernel", "tok2img_partial_mfma_kernel", "dec_tok_mlp_kernel", "dec_tok_attn_kernel", "dec_heads_kernel", "amg_postprocess_kernel", "rle_emit_kernel"):
        assert any(must in n for n in hot), must
    bad = {n: (r.get("vgpr_spill_count", 0), r.get("private_segment_fixed_size", 0)) for n, r in hot.items()
           if r.get("vgpr_spill_count", 0) or r.get("sgpr_spill_count", 0) or r.get("private_segment_fixed_size", 0)}
    assert not bad, bad
    pers = {n: r for n, r in ks.items() if any(re.search(h, n) for h in KR.HOT_PERSISTENT)}      # round 6: the default GEMM of multi-round launches
    assert len(pers) >= 6, sorted(pers)
    bad = {n: (r.get("vgpr_spill_count", 0), r.get("private_segment_fixed_size", 0)) for n, r in pers.items() if r.get("vgpr_spill_count", 0) or r.get("private_segment_fixed_size", 0)}
    assert not bad, bad
    assert all(r.get("vgpr_count", 0) <= 512 for r in ks.values())


def test_transposed_weight_cache_follows_the_weight():
    """training._TransposeCache (the W^T copies of frozen bf16 linears): keyed on the parameter object, so a freed-and-rebuilt model at the
    same addresses cannot meet a stale copy; in-place updates of the parameter are seen through its version counter; bounded in bytes."""
    import gc
    from ullsam_amd import training as T
    c = T._TransposeCache(max_bytes=4 * 8 * 4 + 16 * 16 * 4)
    w = torch.nn.Parameter(torch.randn(4, 8))
    t = c.get(w)
    assert t.shape == (8, 4) and torch.equal(t, w.detach().t()) and c.get(w) is t
    with torch.no_grad():
        w.mul_(2.0)                                   # an optimizer step / load_state_dict: version bump
    t2 = c.get(w)
    assert t2 is not t and torch.equal(t2, w.detach().t())
    addr = w.data_ptr()
    del w, t, t2
    gc.collect()
    assert len(c) == 0 and c.bytes >= 0               # the entry died with its weight
    w_new = torch.nn.Parameter(torch.randn(4, 8))     # may or may not reuse `addr`: either way nothing stale can be returned
    assert torch.equal(c.get(w_new), w_new.detach().t()), addr
    big = torch.nn.Parameter(torch.randn(16, 16))
    c.get(big)
    assert c.bytes <= c.max_bytes
    big2 = torch.nn.Parameter(torch.randn(16, 16))
    c.get(big2)                                       # over budget: the cache is dropped, then refilled
    assert c.bytes <= c.max_bytes and len(c) == 1
    T.invalidate_transposed_weights()
    assert len(T._WT_CACHE) == 0
    # the module's own cache decides its budget at first use: min(16 GiB, 1 / 8 of the device's memory) for a weight on the GPU, NOTHING for a CPU tensor (every get() a fresh transpose)
    T._WT_CACHE.max_bytes = -1
    w0 = torch.nn.Parameter(torch.randn(4, 8))
    T._WT_CACHE.get(w0)
    assert T._WT_CACHE.max_bytes == 0
    assert torch.equal(T._WT_CACHE.get(w0), w0.detach().t()) and len(T._WT_CACHE) == 0 and T._WT_CACHE.bytes == 0
    T.set_transpose_cache_bytes(1 << 20)
    try:
        assert T._WT_CACHE.get(w0) is T._WT_CACHE.get(w0) and len(T._WT_CACHE) == 1
    finally:
        T.set_transpose_cache_bytes(0)
    assert len(T._WT_CACHE) == 0
    T._WT_CACHE.max_bytes = -1                        # (back to "not decided yet" for whatever runs after this test)


def _gloo_uneven_worker(rank, world, port, q):
    """bench.py's make_step / timed_steps under gloo with UNEQUAL per-rank step times: rank 1 sleeps in every step."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import time
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch, calls = 2, [0]
        base = bench.stub_compute(batch, rank)

        def compute():
            calls[0] += 1
            if rank == 1:
                time.sleep(0.05)
            low, mk = base()
            return low + float(calls[0]), mk          # every step's payload is different: the LAST gather must carry the last step's values

        step = bench.make_step(compute, batch, world)
        t0 = time.perf_counter()
        dt, gathered = bench.timed_steps(step, warmup=2, steps=6, world=world, device="cpu")
        wall = time.perf_counter() - t0
        ok = gathered is not None and gathered[0].shape[0] == batch * world and calls[0] == 8
        # rows of rank r carry rank r's stub image + the step counter of the LAST step (8): nothing dropped, nothing from an earlier step
        for r in range(world):
            want = bench.stub_compute(batch, r)()[0] + 8.0
            ok = ok and torch.equal(gathered[0][r * batch:(r + 1) * batch], want)
        ok = ok and step.drain() is None              # nothing left in flight
        q.put((rank, bool(ok), dt, wall))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_overlapped_gather_with_unequal_step_times():
    """One rank is 50 ms per step slower than the other: the one-in-flight gather handle must neither deadlock nor drop the last exchange,
    and the reported time is the MAX over ranks (both ranks report the slow rank's time)."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_uneven_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert [r[:2] for r in res] == [(0, True), (1, True)], res
    dts = [r[2] for r in res]
    assert abs(dts[0] - dts[1]) < 1e-9 and dts[0] >= 6 * 0.05      # the all-reduced MAX: >= the slow rank's six sleeps, identical on both ranks


def _gloo_amg_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from ullsam_amd import parallel
    from ullsam_amd.automatic_mask_generator import SamAutomaticMaskGenerator
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        class StubGenerator(SamAutomaticMaskGenerator):          # the sharding / gather logic of generate_batch without a GPU model
            def __init__(self):
                self.calls = []

            def generate(self, image):
                self.calls.append(int(image))
                n = int(image) % 4                                # ragged: 0 .. 3 records per tile, RLE lists of different lengths
                return [{"tile": int(image), "k": k, "segmentation": {"size": [8, 8], "counts": list(range(int(image) + k + 1))}} for k in range(n)]

        ok = True
        for n_tiles in (5, 2, 1, 0):                              # ragged split, even split, fewer tiles than ranks, nothing at all
            gen = StubGenerator()
            res = gen.generate_batch(list(range(10, 10 + n_tiles)))
            a, b = parallel.shard_range(n_tiles, rank, world)
            ok = ok and gen.calls == list(range(10 + a, 10 + b))  # this rank generated exactly its own share
            ok = ok and len(res) == n_tiles
            for i, recs in enumerate(res):
                t = 10 + i
                ok = ok and [r["tile"] for r in recs] == [t] * (t % 4) and all(r["segmentation"]["counts"] == list(range(t + r["k"] + 1)) for r in recs)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_mask_generator_shards_tiles_and_gathers_ragged_records():
    """SamAutomaticMaskGenerator.generate_batch under a world-2 gloo group: every rank generates only its share of the tiles, and all ranks end
    up with every tile's (ragged) record list in tile order -- including more ranks than tiles and no tiles at all."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_amg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)], res


def test_sampling_distribution_equals_the_logits_warpers_of_transformers():
    """app.py:469-477 samples captions with temperature 0.7 / top_k 50 / top_p 0.9 through transformers' generate, which is not runnable here
    (SURVEY 8c).  The distribution `_sampling_probs` builds is compared with a numpy restatement of the three logits warpers generate applies, in
    its order -- TemperatureLogitsWarper (scores / T), TopKLogitsWarper (scores < k-th largest -> -inf), TopPLogitsWarper (ascending sort,
    cumulative probability <= 1 - top_p removed, the last token always kept) -- on random logits incl. ties and a peaked row."""
    from ullsam_amd.modeling.modeling_internlm2 import _sampling_probs
    rng = np.random.default_rng(11)
    V = 500
    rows = [rng.standard_normal(V) * 3, rng.standard_normal(V) * 0.1, np.concatenate([[30.0], rng.standard_normal(V - 1)]),
            np.round(rng.standard_normal(V) * 2) / 2]                         # (the last row has many exact ties)
    logits = np.stack(rows).astype(np.float32)

    def hf(scores, T, k, p):
        s = scores.astype(np.float64) / T
        if k:
            kth = np.sort(s)[-min(k, s.size)]
            s = np.where(s < kth, -np.inf, s)
        if p and p < 1.0:
            order = np.argsort(s, kind="stable")                               # ascending
            e = np.exp(s[order] - s[order][-1])
            cum = np.cumsum(e / e.sum())
            remove = cum <= (1.0 - p)
            remove[-1] = False                                                 # min_tokens_to_keep = 1
            s[order[remove]] = -np.inf
        e = np.exp(s - s.max())
        return e / e.sum()

    for T, k, p in [(0.7, 50, 0.9), (1.0, 0, 0.9), (0.7, 50, None), (1.3, 5, 0.5), (1.0, 0, None)]:
        got = _sampling_probs(torch.from_numpy(logits), T, k, p).numpy().astype(np.float64)
        for r in range(logits.shape[0]):
            want = hf(logits[r], T, k, p)
            # compared as multisets of probabilities: WHICH members of a group of tied logits fall outside the nucleus depends on the sort's tie order (in
            # transformers as here); a token exactly on the nucleus boundary may go either way (one token of difference allowed)
            a, b = np.sort(got[r])[::-1], np.sort(want)[::-1]
            na, nb = int((a > 0).sum()), int((b > 0).sum())
            assert abs(na - nb) <= 1, (T, k, p, r, na, nb)
            n = min(na, nb)
            assert np.abs(a[:n] / a[:n].sum() - b[:n] / b[:n].sum()).max() < 1e-5, (T, k, p, r)
            assert abs(got[r].sum() - 1.0) < 1e-5
