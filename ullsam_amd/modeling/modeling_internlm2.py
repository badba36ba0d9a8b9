"""InternLM2 decoder stack on HIP kernels.

API/state_dict mirror of modeling/modeling_internlm2.py: InternLM2ForCausalLM(config) with `.model` (InternLM2Model:
tok_embeddings, layers[i].{attention.{wqkv,wo}, feed_forward.{w1,w3,w2}, attention_norm, ffn_norm}, norm) and `.output`;
forward(...) keyword surface of :1022-1034 / :854-865; prepare_inputs_for_generation (:1112-1149) semantics inside generate().

Per layer (residual stream fp32, GEMM operands in the model dtype):
  RMSNorm -> GEMM wqkv -> de-interleave + RoPE + KV-cache append -> causal GQA flash attention (prefill) or
  strided single-query attention (decode) -> GEMM wo (+residual) -> RMSNorm -> GEMM [w1|w3] with fused SwiGLU ->
  GEMM w2 (+residual).
Not implemented (unused by uLLSAM): output_attentions, InternLM2ForSequenceClassification,
chat/stream_chat.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import nn

from .. import ops
from ..packing import pack_w13
from .common import Linear, Packed
from .configuration_internlm2 import InternLM2Config
from .outputs import BaseModelOutputWithPast, CausalLMOutputWithPast


class InternLM2RMSNorm(Packed):
    """modeling_internlm2.py:129-143."""

    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def w(self):
        return self.f32("w", self.weight)

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        return ops.norm(hidden_states.contiguous(), self.w(), None, self.variance_epsilon, hidden_states.dtype, rms=True)


class InternLM2MLP(Packed):
    def __init__(self, config):
        super().__init__()
        self.w1 = Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.w3 = Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.w2 = Linear(config.intermediate_size, config.hidden_size, bias=False)
        if config.hidden_act != "silu":
            raise NotImplementedError("the fused SwiGLU epilogue implements hidden_act='silu'")

    def w13(self, dt):
        return self.pk("w13", (self.w1.weight, self.w3.weight), lambda: pack_w13(self.w1.weight.detach().to(dt), self.w3.weight.detach().to(dt)))


class InternLM2Attention(Packed):
    def __init__(self, config: InternLM2Config):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self.num_key_value_heads = config.num_key_value_heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        if self.head_dim * self.num_heads != self.hidden_size:
            raise ValueError(f"hidden_size must be divisible by num_heads (got `hidden_size`: {self.hidden_size}"
                             f" and `num_heads`: {self.num_heads}).")
        self.wqkv = Linear(self.hidden_size, (self.num_heads + 2 * self.num_key_value_heads) * self.head_dim, bias=config.bias)
        self.wo = Linear(self.num_heads * self.head_dim, self.hidden_size, bias=config.bias)


class InternLM2DecoderLayer(Packed):
    def __init__(self, config: InternLM2Config):
        super().__init__()
        self.attention = InternLM2Attention(config)
        self.feed_forward = InternLM2MLP(config)
        self.attention_norm = InternLM2RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.ffn_norm = InternLM2RMSNorm(config.hidden_size, eps=config.rms_norm_eps)


class _Embedding(Packed):
    def __init__(self, n, d, padding_idx=None):
        super().__init__()
        self.padding_idx = padding_idx
        self.weight = nn.Parameter(torch.empty(n, d))

    def forward(self, ids: torch.Tensor) -> torch.Tensor:
        """fp32 rows (the residual stream is fp32); cast by the caller if another dtype is wanted."""
        shp = ids.shape
        out = ops.embed_tokens(self.weight.detach(), ids.reshape(1, -1).contiguous(), None, None)
        return out.reshape(*shp, self.weight.shape[1])


class KVCache:
    """Pre-allocated per-layer K/V [B, KVH, cap, hd] in the model dtype (288 GB HBM: size for the whole generation up front).
    `to_tuple()` gives the reference's tuple-of-(k, v) views [B, KVH, len, hd] (modeling_internlm2.py:383-388)."""

    def __init__(self, layers, B, kvh, cap, hd, dtype, device):
        # only rows below `len` are ever read (every kernel takes the live length), so no zero fill
        self.k = [torch.empty((B, kvh, cap, hd), dtype=dtype, device=device) for _ in range(layers)]
        self.v = [torch.empty((B, kvh, cap, hd), dtype=dtype, device=device) for _ in range(layers)]
        self.len, self.cap = 0, cap

    def to_tuple(self):
        return tuple((k[:, :, :self.len], v[:, :, :self.len]) for k, v in zip(self.k, self.v))

    def grow(self, cap: int):
        """The reference's cache is a torch.cat per step and never fills up (modeling_internlm2.py:383-388): when the pre-allocated
        rows run out, move the live rows into a larger allocation instead of failing."""
        if cap <= self.cap:
            return
        for buf in (self.k, self.v):
            for i, old in enumerate(buf):
                new = torch.empty((old.shape[0], old.shape[1], cap, old.shape[3]), dtype=old.dtype, device=old.device)
                new[:, :, :self.len].copy_(old[:, :, :self.len])
                buf[i] = new
        self.cap = cap


class InternLM2Model(Packed):
    def __init__(self, config: InternLM2Config):
        super().__init__()
        self.config = config
        self.padding_idx = config.pad_token_id
        self.vocab_size = config.vocab_size
        self.tok_embeddings = _Embedding(config.vocab_size, config.hidden_size, self.padding_idx)
        self.layers = nn.ModuleList([InternLM2DecoderLayer(config) for _ in range(config.num_hidden_layers)])
        self.norm = InternLM2RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self._rope = None
        self.collect_all_hidden_states = False
        self.fuse_decode = True   # decode steps use the fused norm / RoPE launches (tests switch it off to compare with the separate kernels)
        self.stage_probe = None   # diagnostic tap: stage_probe(layer_index, x) with the fp32 residual stream [B*S, D] after every layer

    def get_input_embeddings(self):
        return self.tok_embeddings

    @property
    def compute_dtype(self):
        return self.layers[0].attention.wqkv.weight.dtype

    # -- RoPE tables (InternLM2RotaryEmbedding :147-180, Linear :184-200, DynamicNTK :204-229), fp32 ---------------------
    def rope_tables(self, seq_len: int, device):
        """cos / sin [rows >= seq_len, head_dim] fp32.  Cached like the reference's rotary modules: rebuilt only when a longer
        sequence arrives (:172-175); with dynamic-NTK scaling the base is rescaled from the sequence length of the call that
        rebuilds the cache, and only when that exceeds max_position_embeddings (:216-221)."""
        cfg = self.config
        hd = cfg.hidden_size // cfg.num_attention_heads
        rs = cfg.rope_scaling
        dynamic = rs is not None and rs["type"] == "dynamic"
        if self._rope is not None and self._rope[0][0] >= seq_len and self._rope[0][1] == str(device):
            return self._rope[1], self._rope[2]
        base = float(cfg.rope_theta)
        scale_t = 1.0
        rows = max(seq_len, 2048)  # rows do not depend on the table length: build a few more than asked for
        if rs is not None and rs["type"] == "linear":
            scale_t = float(rs["factor"])
        elif dynamic and seq_len > cfg.max_position_embeddings:
            f = float(rs["factor"])
            base = base * ((f * seq_len / cfg.max_position_embeddings) - (f - 1)) ** (hd / (hd - 2))
            rows = seq_len
        elif dynamic:
            rows = min(rows, cfg.max_position_embeddings)  # a longer sequence later must trigger the rescale, as in the reference
        inv_freq = 1.0 / (base ** (torch.arange(0, hd, 2).float() / hd))
        t = torch.arange(rows, dtype=inv_freq.dtype) / scale_t
        freqs = torch.einsum("i,j->ij", t, inv_freq)
        emb = torch.cat((freqs, freqs), dim=-1)
        cos, sin = emb.cos().to(device).contiguous(), emb.sin().to(device).contiguous()
        self._rope = ((rows, str(device)), cos, sin)
        return cos, sin

    def new_cache(self, B: int, cap: int, device) -> KVCache:
        c = self.config
        return KVCache(c.num_hidden_layers, B, c.num_key_value_heads, cap, c.hidden_size // c.num_attention_heads, self.compute_dtype, device)

    # -- the layer stack -------------------------------------------------------------------------------------------
    def run_layers(self, x: torch.Tensor, B: int, S: int, pos: torch.Tensor, key_mask: Optional[torch.Tensor],
                   cache: Optional[KVCache], collect: Optional[list] = None) -> torch.Tensor:
        """x fp32 [B*S, D] (updated in place) -> post-final-norm hidden [B*S, D] in the model dtype."""
        cfg = self.config
        dt = self.compute_dtype
        H, KVH = cfg.num_attention_heads, cfg.num_key_value_heads
        hd, G = cfg.hidden_size // H, H // KVH
        past = cache.len if cache is not None else 0
        Sk = past + S
        cos, sin = self.rope_tables(Sk, x.device)
        tmp_k = tmp_v = None
        if cache is None:
            tmp_k = torch.empty((B, KVH, S, hd), dtype=dt, device=x.device)
            tmp_v = torch.empty_like(tmp_k)
        # a decode step of <= 4 sequences (bf16, head_dim 128, hidden <= 4096): norms and RoPE ride in the GEMMs' prologue / epilogue
        fused = S == 1 and cache is not None and hd == 128 and ops.decode_fusable(B, cfg.hidden_size, dt) and self.fuse_decode
        for li, layer in enumerate(self.layers):
            if collect is not None:
                collect.append(x.reshape(B, S, -1).to(dt))
            at, ff = layer.attention, layer.feed_forward
            kc, vc = (cache.k[li], cache.v[li]) if cache is not None else (tmp_k, tmp_v)
            if fused:   # decode step: RMSNorm -> wqkv -> head split + RoPE + KV append is ONE launch (the norm runs while the weight stream starts)
                q = ops.decode_qkv_rope(x, layer.attention_norm.w(), layer.attention_norm.variance_epsilon, at.wqkv.w(dt), at.wqkv.b(), kc, vc,
                                        pos, cos, sin, B, KVH, G, past)
            else:
                xn = ops.norm(x, layer.attention_norm.w(), None, layer.attention_norm.variance_epsilon, dt, rms=True)
                if hd == 128 and S > 8:   # prefill: head split + RoPE + KV append in the wqkv GEMM's epilogue (no qkv round trip)
                    q = ops.gemm_qkv_rope(xn, at.wqkv.w(dt), at.wqkv.b(), kc, vc, pos, cos, sin, B, S, KVH, G, past)
                else:
                    qkv = ops.gemm(xn, at.wqkv.w(dt), at.wqkv.b())
                    q = ops.rope_split(qkv, kc, vc, pos, cos, sin, B, S, KVH, G, hd, past)
            if S > 1:
                a = ops.causal_attention(q, kc, vc, key_mask, B, H, KVH, hd, S, Sk, past)
            elif dt == torch.bfloat16 and hd == 128 and G <= 8:  # decode step: streaming split-K kernel over the cache
                a = ops.decode_attention(q, kc, vc, key_mask, B, H, KVH, hd, Sk)
            else:  # decode step: one query against the cache (no causal term: _prepare_decoder_attention_mask :834)
                cap = kc.shape[2]
                a = ops.naive_attention(q, kc, vc, B, H, KVH, hd, 1, Sk, (H * hd, H * hd, hd), (KVH * cap * hd, hd, cap * hd),
                                        (KVH * cap * hd, hd, cap * hd), (H * hd, H * hd, hd), hd ** -0.5, key_mask=key_mask)
            ops.gemm(a, at.wo.w(dt), at.wo.b(), residual=x, out_f32=True, out=x)
            if fused:
                hmid = ops.gemm_rmsnorm(x, layer.ffn_norm.w(), layer.ffn_norm.variance_epsilon, ff.w13(dt), act=ops.ACT_SWIGLU)
            else:
                xn = ops.norm(x, layer.ffn_norm.w(), None, layer.ffn_norm.variance_epsilon, dt, rms=True)
                hmid = ops.gemm(xn, ff.w13(dt), act=ops.ACT_SWIGLU)
            ops.gemm(hmid, ff.w2.w(dt), None, residual=x, out_f32=True, out=x)
            if self.stage_probe is not None:
                self.stage_probe(li, x)
        if cache is not None:
            cache.len = Sk
        return ops.norm(x, self.norm.w(), None, self.norm.variance_epsilon, dt, rms=True)

    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None):
        cfg = self.config
        if output_attentions:
            raise NotImplementedError("attention probabilities are never materialised by the fused attention kernels")
        use_cache = use_cache if use_cache is not None else cfg.use_cache
        if input_ids is not None and inputs_embeds is not None:
            raise ValueError("You cannot specify both input_ids and inputs_embeds at the same time")
        if input_ids is None and inputs_embeds is None:
            raise ValueError("You have to specify either input_ids or inputs_embeds")
        if inputs_embeds is None:
            B, S = input_ids.shape[:2]
            x = ops.embed_tokens(self.tok_embeddings.weight.detach(), input_ids.contiguous(), None, None)
            dev = input_ids.device
        else:
            B, S = inputs_embeds.shape[:2]
            dev = inputs_embeds.device
            x = inputs_embeds.reshape(B * S, -1).float().contiguous().clone()
        cache = past_key_values if isinstance(past_key_values, KVCache) else None
        if past_key_values is not None and cache is None:
            raise TypeError("past_key_values must be the KVCache returned by a previous call (use_cache=True)")
        past = cache.len if cache is not None else 0
        if cache is None and use_cache:
            cache = self.new_cache(B, max(2 * S, S + 256), dev)
        if cache is not None and past + S > cache.cap:
            cache.grow(max(2 * cache.cap, past + S))
        if position_ids is None:
            position_ids = torch.arange(past, past + S, dtype=torch.long, device=dev).unsqueeze(0).expand(B, S)  # :893-898
        pos = position_ids.to(torch.int32).expand(B, S).contiguous()
        key_mask = None
        if attention_mask is not None:
            key_mask = attention_mask.to(torch.int32).contiguous()
            if key_mask.shape != (B, past + S):
                raise ValueError(f"Attention mask should be of size {(B, past + S)}, but is {tuple(key_mask.shape)}")
        collect = [] if (output_hidden_states and self.collect_all_hidden_states) else None
        h = self.run_layers(x, B, S, pos, key_mask, cache, collect).reshape(B, S, -1)
        all_h = None
        if output_hidden_states:
            # inputs of every layer + post-final-norm (:930-974).  Only the last entry is used by uLLSAM; the per-layer
            # inputs are materialised only when `collect_all_hidden_states` is set (they cost a copy per layer).
            all_h = tuple(collect) + (h,) if collect is not None else (None,) * len(self.layers) + (h,)
        return BaseModelOutputWithPast(last_hidden_state=h, past_key_values=cache if use_cache else None, hidden_states=all_h)


class InternLM2ForCausalLM(Packed):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.model = InternLM2Model(config)
        self.vocab_size = config.vocab_size
        self.output = Linear(config.hidden_size, config.vocab_size, bias=False)

    def get_input_embeddings(self):
        return self.model.tok_embeddings

    def get_output_embeddings(self):
        return self.output

    def lm_head(self, hidden: torch.Tensor) -> torch.Tensor:
        """logits = output(hidden).float()  (:1081-1082); hidden [..., D] in the model dtype -> fp32 [..., V]."""
        dt = self.model.compute_dtype
        h2 = hidden.reshape(-1, hidden.shape[-1]).contiguous()
        return ops.gemm(ops.cast(h2, dt), self.output.w(dt), None, out_f32=True).reshape(*hidden.shape[:-1], self.vocab_size)

    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None, lazy_logits: bool = False):
        out = self.model(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids, past_key_values=past_key_values,
                         inputs_embeds=inputs_embeds, use_cache=use_cache, output_attentions=output_attentions,
                         output_hidden_states=output_hidden_states)
        hidden = out.last_hidden_state
        loss = None
        logits_fn = lambda: self.lm_head(hidden)
        logits = None
        if labels is not None:
            logits = logits_fn()
            loss = torch.nn.functional.cross_entropy(logits[..., :-1, :].reshape(-1, self.vocab_size), labels[..., 1:].reshape(-1).to(logits.device))
        elif not lazy_logits:
            logits = logits_fn()
        return CausalLMOutputWithPast(loss=loss, logits=logits, logits_fn=logits_fn, past_key_values=out.past_key_values,
                                      hidden_states=out.hidden_states)

    # -- generation -------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, input_ids=None, inputs_embeds=None, attention_mask=None, generation_config=None, max_new_tokens=None,
                 do_sample=False, eos_token_id=None, pad_token_id=None, temperature=1.0, top_k=None, top_p=None,
                 use_cache=True, output_hidden_states=None, num_beams=1, **kwargs) -> torch.LongTensor:
        """Token loop with the reference's prepare_inputs_for_generation semantics (:1112-1149): the first step consumes
        inputs_embeds (or input_ids), later steps the last id with position_ids = cumsum(mask)-1.  Returns only the new tokens
        when inputs_embeds is given, prompt + new tokens when input_ids is given (HF GenerationMixin behaviour).
        Greedy by default (bit-exact parity path); do_sample applies temperature / top-k / top-p on the fp32 logits."""
        if num_beams != 1:
            raise NotImplementedError("beam search is not part of the uLLSAM hot path")
        if generation_config is not None:
            g = generation_config if isinstance(generation_config, dict) else generation_config.__dict__
            max_new_tokens = max_new_tokens or g.get("max_new_tokens")
            eos_token_id = eos_token_id if eos_token_id is not None else g.get("eos_token_id")
            do_sample = do_sample or bool(g.get("do_sample", False))
        max_new_tokens = int(max_new_tokens or 20)
        eos = eos_token_id if eos_token_id is not None else self.config.eos_token_id
        eos_set = set(eos) if isinstance(eos, (list, tuple)) else {int(eos)}
        pad = pad_token_id if pad_token_id is not None else (self.config.pad_token_id or 0)
        first = inputs_embeds if inputs_embeds is not None else input_ids
        B, S = first.shape[:2]
        dev = first.device
        mask = torch.ones((B, S), dtype=torch.long, device=dev) if attention_mask is None else attention_mask.long()
        pos = mask.cumsum(-1) - 1
        pos = pos.masked_fill(mask == 0, 1)
        cache = self.model.new_cache(B, S + max_new_tokens + 1, dev)
        out = self.model(input_ids=None if inputs_embeds is not None else input_ids, inputs_embeds=inputs_embeds, attention_mask=mask,
                         position_ids=pos, past_key_values=cache, use_cache=True)
        done = torch.zeros(B, dtype=torch.bool, device=dev)
        track_eos = eos_set != {-1}                        # eos_token_id=-1: run to max_new_tokens, no per-step stop bookkeeping
        eos_t = torch.tensor(sorted(eos_set), dtype=torch.long, device=dev)
        pad_t = torch.tensor(pad, dtype=torch.long, device=dev)
        new: List[torch.Tensor] = []
        h_last = out.last_hidden_state[:, -1]
        # The stop test ("every sequence has emitted eos") needs the token values on the host.  It is evaluated on a copy made
        # `eos_check_every` steps earlier (pinned buffer + event, no stream sync): the host keeps enqueueing steps while the GPU works,
        # and at most that many surplus steps run after the last eos; their tokens are `pad` and are trimmed below, so the returned
        # ids are exactly those of a loop that tests every step.
        # With do_sample the test is made every step (blocking): a surplus step would draw from the global torch RNG and leave its state
        # different from a loop that stops at once, so later sampled calls would not reproduce against the reference loop.
        every = 0 if do_sample else max(1, int(kwargs.get("eos_check_every", 8)))
        flags = torch.empty((max_new_tokens,), dtype=torch.bool, pin_memory=True) if track_eos else None  # one pinned buffer per call
        pending = []  # (step index, event)
        stop_at = None
        mask_full = torch.ones((B, S + max_new_tokens), dtype=torch.int32, device=dev)   # int32: what the kernels take (no per-step conversion)
        mask_full[:, :S] = mask
        pos_next = mask.sum(-1, keepdim=True).to(torch.int32)  # = cumsum(mask)[:, -1]: position id of the next token (cumsum - 1 of the extended mask)
        for step in range(max_new_tokens):
            logits = self.lm_head(h_last)  # fp32 [B, V], last position only
            if do_sample:
                tok = _sample(logits, temperature, top_k, top_p)
            else:
                tok = ops.argmax(logits.contiguous())
            if track_eos:
                tok = torch.where(done, pad_t, tok)
                done = done | torch.isin(tok, eos_t)
            new.append(tok)
            if track_eos:
                flags[step:step + 1].copy_(done.all().reshape(1), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                pending.append((step, ev))
                while pending and (pending[0][1].query() or len(pending) > every):
                    st, ev0 = pending.pop(0)
                    ev0.synchronize()
                    if bool(flags[st]):
                        stop_at = st
                        break
            if stop_at is not None or step == max_new_tokens - 1:
                break
            cur = S + step + 1
            out = self.model(input_ids=tok.reshape(B, 1), attention_mask=mask_full[:, :cur], position_ids=pos_next, past_key_values=cache,
                             use_cache=True)
            pos_next = pos_next + 1
            h_last = out.last_hidden_state[:, -1]
        if stop_at is None:
            for st, ev0 in pending:  # the loop ran out: the earliest step at which everything was done, if any
                ev0.synchronize()
                if bool(flags[st]):
                    stop_at = st
                    break
        if stop_at is not None:
            new = new[:stop_at + 1]
        gen = torch.stack(new, 1)
        return gen if inputs_embeds is not None else torch.cat([input_ids, gen], 1)


def _sampling_probs(logits: torch.Tensor, temperature, top_k, top_p) -> torch.Tensor:
    """The distribution the caption path samples from (app.py:469-477: T = 0.7, top_p = 0.9, top_k = 50 through transformers' generate): temperature,
    then top-k (everything below the k-th logit removed), then nucleus (the smallest prefix of the descending order whose mass BEFORE a token
    is <= top_p is kept, at least one token) -- the order and the rules of transformers' TemperatureLogitsWarper / TopKLogitsWarper /
    TopPLogitsWarper (tests/test_host_cpu.py restates those in numpy)."""
    x = logits / max(float(temperature or 1.0), 1e-5)
    if top_k:
        kth = torch.topk(x, min(int(top_k), x.shape[-1]), dim=-1).values[..., -1:]
        x = x.masked_fill(x < kth, float("-inf"))
    if top_p and top_p < 1.0:
        sx, si = torch.sort(x, descending=True, dim=-1)
        cp = torch.softmax(sx, -1).cumsum(-1)
        rm = cp - torch.softmax(sx, -1) > top_p
        sx = sx.masked_fill(rm, float("-inf"))
        x = torch.full_like(x, float("-inf")).scatter(-1, si, sx)
    return torch.softmax(x, -1)


def _sample(logits: torch.Tensor, temperature, top_k, top_p) -> torch.Tensor:
    """Host-side sampling policy on the kernel-produced logits."""
    return torch.multinomial(_sampling_probs(logits, temperature, top_k, top_p), 1).squeeze(-1)
