"""InternVLSAMModel (uLLSAM composite) on HIP kernels.

API mirror of modeling/modeling_internvl_sam.py:36-442: constructor (config, vision_model, prompt_encoder, mask_decoder,
language_model, use_flash_attn), attributes (.vision_model .prompt_encoder .mask_decoder .language_model .mlp1 .mlp2
.img_context_token_id .num_image_token .template .system_message), forward(...) keyword surface (:106-120) returning an object
with .loss .logits .past_key_values .hidden_states .attentions .image_embeddings, generate(...) (:394-442), extract_feature,
pixel_shuffle, text_aware_dense_feature; same state_dict keys.

Differences, all documented in DESIGN.md:
  * B > 1 works and means "the reference at B = 1, per sample" (the reference raises at B > 1).
  * logits are computed lazily (the reference materialises [B,S,92553] fp32 although the mask path never reads them).
  * activations between stages stay token-major (NHWC); NCHW tensors are produced only at the API boundary.
"""
from __future__ import annotations

import math
import warnings
from typing import List, Optional

import torch
from torch import nn

from .. import ops
from .common import LayerNorm, Linear, Packed
from .configuration_internvl_chat import InternVLChatConfig
from .modeling_internlm2 import InternLM2ForCausalLM
from .outputs import CausalLMOutputWithPast

INTERNLM2_CHAT_SYSTEM = ("You are an AI assistant whose name is InternLM (书生·浦语).")  # conversation.py 'internlm2-chat' template


class InternVLSAMModel(Packed):
    config_class = InternVLChatConfig
    main_input_name = "pixel_values"
    base_model_prefix = "language_model"

    def __init__(self, config: InternVLChatConfig, vision_model=None, prompt_encoder=None, mask_decoder=None, language_model=None,
                 use_flash_attn=True):
        super().__init__()
        self.config = config
        self.patch_size = 16
        self.select_layer = config.select_layer
        self.template = config.template
        self.num_image_token = 64 * 64 * (config.downsample_ratio ** 2)  # float, as in the reference (:54)
        self.downsample_ratio = config.downsample_ratio
        self.ps_version = config.ps_version
        if config.downsample_ratio != 0.5 or config.ps_version == "v1":
            raise NotImplementedError("the HIP shuffle kernels implement downsample_ratio=0.5, ps_version='v2' "
                                      "(what uLLSAM constructs, train_joint_v2.py:1424-1431)")
        if vision_model is not None:
            self.vision_model = vision_model
        if prompt_encoder is not None:
            self.prompt_encoder = prompt_encoder
        if mask_decoder is not None:
            self.mask_decoder = mask_decoder
        if language_model is not None:
            self.language_model = language_model
        elif config.llm_config.architectures[0] == "InternLM2ForCausalLM":
            self.language_model = InternLM2ForCausalLM(config.llm_config)
        else:
            raise NotImplementedError(f"{config.llm_config.architectures[0]} is not implemented.")
        sam_hidden_size = 256
        llm_hidden_size = config.llm_config.hidden_size
        c4 = sam_hidden_size * int(1 / self.downsample_ratio) ** 2
        self.mlp1 = nn.Sequential(LayerNorm(c4), Linear(c4, llm_hidden_size), nn.GELU(), Linear(llm_hidden_size, llm_hidden_size))
        self.mlp2 = nn.Sequential(LayerNorm(llm_hidden_size), Linear(llm_hidden_size, c4), nn.GELU(), Linear(c4, c4))
        self.img_context_token_id = 92546
        self.system_message = INTERNLM2_CHAT_SYSTEM

    @property
    def device(self):
        return self.mlp1[1].weight.device

    @property
    def dtype(self):
        return self.mlp1[1].weight.dtype

    # -- projector stages on token-major tensors ----------------------------------------------------------------------
    def _mlp1_tokens(self, img_tok: torch.Tensor, B: int) -> torch.Tensor:
        """pixel_shuffle(v2) + mlp1 (:226-251, :88-93): image tokens fp32 [B,4096,256] -> vit_embeds fp32 [B*1024, Dl]."""
        dt = self.dtype
        ln, l1, l3 = self.mlp1[0], self.mlp1[1], self.mlp1[3]
        f = ops.pixel_shuffle_ln(img_tok, *ln.wb(), B, 64, 64, 256, ln.eps, dt)
        h = ops.gemm(f, l1.w(dt), l1.b(), act=ops.ACT_GELU)
        return ops.gemm(h, l3.w(dt), l3.b(), out_f32=True)

    def _mlp2_tokens(self, rows: torch.Tensor, B: int) -> torch.Tensor:
        """mlp2 + inverse pixel shuffle (:253-270): hidden rows [B*1024, Dl] (model dtype) -> dense feature tokens fp32 [B,4096,256]."""
        dt = self.dtype
        ln, l1, l3 = self.mlp2[0], self.mlp2[1], self.mlp2[3]
        f = ops.norm(rows, *ln.wb(), ln.eps, dt)
        f = ops.gemm(f, l1.w(dt), l1.b(), act=ops.ACT_GELU)
        f = ops.gemm(f, l3.w(dt), l3.b(), out_f32=True)
        return ops.pixel_unshuffle(f, B, 64, 64, 256)

    # -- reference API ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def extract_feature(self, pixel_values):
        B = pixel_values.shape[0]
        img_tok = self.vision_model.forward_tokens(pixel_values)
        vit = self._mlp1_tokens(img_tok, B).reshape(B, 1024, -1)
        feats = ops.transpose(img_tok, B, 4096, 256).reshape(B, 256, 64, 64).to(self.dtype)
        return vit.to(self.dtype), feats

    @torch.no_grad()
    def text_aware_dense_feature(self, features):
        B = features.shape[0]
        tok = self._mlp2_tokens(ops.cast(features.reshape(B * features.shape[1], -1).contiguous(), self.dtype), B)
        return ops.transpose(tok, B, 4096, 256).reshape(B, 256, 64, 64)

    def pixel_shuffle(self, x, scale_factor=0.5):
        """Layout-only helper kept for API parity (:226-240); the hot path uses the fused gather+LN kernel instead."""
        n, h, w, c = x.size()
        x = x.reshape(n, h, int(w * scale_factor), int(c / scale_factor)).permute(0, 2, 1, 3).contiguous()
        x = x.reshape(n, int(w * scale_factor), int(h * scale_factor), int(c / (scale_factor * scale_factor)))
        return x.permute(0, 2, 1, 3).contiguous()

    def forward(self, pixel_values, input_ids=None, attention_mask=None, position_ids=None, image_flags=None, past_key_values=None,
                labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None,
                img_context_token_id=None):
        if return_dict is False:
            raise NotImplementedError("tuple outputs are not provided; use return_dict=True (what app.py / train_joint_v2.py pass)")
        from .. import training
        if training.wants_autograd(self) and past_key_values is None and position_ids is None:
            # train() mode with gradients on (train_joint_v2.py:988-998): mlp1 -> frozen LLM -> mlp2 as an autograd graph over HIP kernels
            return training.composite_forward(self, pixel_values, input_ids, attention_mask, labels, output_hidden_states)
        with torch.no_grad():
            return self._forward_inference(pixel_values, input_ids, attention_mask, position_ids, image_flags, past_key_values, labels, use_cache,
                                           output_attentions, output_hidden_states, return_dict, img_context_token_id)

    def _forward_inference(self, pixel_values, input_ids=None, attention_mask=None, position_ids=None, image_flags=None, past_key_values=None,
                           labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None,
                           img_context_token_id=None):
        B, S = input_ids.shape
        dev = input_ids.device
        lm = self.language_model
        ids = input_ids.contiguous()
        # image-token scan first, so its tiny D2H copy is long finished when it is checked at the end (no pipeline bubble)
        rank, rng = ops.scan_image_tokens(ids, self.img_context_token_id)  # id is hard-coded, the kwarg is ignored (:102,136)
        # (under HIP-graph capture no host copy / event wait is allowed: the span check below then belongs to the un-captured warm-up
        # call the capturing code has to make on the same ids anyway)
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            rng_host = torch.empty((B, 2), dtype=torch.int32, pin_memory=True)
            rng_host.copy_(rng, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        img_tok = self.vision_model.forward_tokens(pixel_values)                    # [B, 4096, 256] fp32
        vit_embeds = self._mlp1_tokens(img_tok, B)                                    # [B*1024, Dl] fp32
        x = ops.embed_tokens(lm.model.tok_embeddings.weight.detach(), ids, rank, vit_embeds)
        out = lm(inputs_embeds=x.reshape(B, S, -1), attention_mask=attention_mask, position_ids=position_ids,
                 past_key_values=past_key_values, use_cache=use_cache, output_attentions=output_attentions,
                 output_hidden_states=output_hidden_states, labels=labels, lazy_logits=True)
        g = 64
        image_embeddings = ops.transpose(img_tok, B, g * g, 256).reshape(B, 256, g, g).to(self.dtype)
        hidden_states = out.hidden_states
        dense_tok = None
        if out.hidden_states is not None:
            n_img = int(self.num_image_token)
            rows = ops.gather_rows(out.hidden_states[-1].reshape(B * S, -1), rng, B, S, n_img)   # hidden[:, start:end] (:198-200)
            dense_tok = self._mlp2_tokens(rows, B)
            hidden_states = ops.transpose(dense_tok, B, g * g, 256).reshape(B, 256, g, g)
            if not capturing:
                ev.synchronize()
                r = rng_host.numpy()
                if (r[:, 1] <= r[:, 0]).any():
                    raise ValueError("Can not find vision token!")  # (:202-203)
                if ((r[:, 1] - r[:, 0]) != n_img).any():
                    raise RuntimeError(f"image-token span {(r[:, 1] - r[:, 0]).tolist()} != {n_img}: text_aware_dense_feature needs a "
                                       "32x32 token grid (the reference's reshape fails the same way)")
        ret = CausalLMOutputWithPast(loss=out.loss, logits=out._logits, logits_fn=out._logits_fn, past_key_values=out.past_key_values,
                                     hidden_states=hidden_states, attentions=None)
        ret.image_embeddings = image_embeddings
        # token-major copies for the HIP prompt-encoder / mask-decoder fast path (skips three NCHW<->NHWC transposes)
        ret.image_tokens = img_tok
        ret.dense_feature_tokens = dense_tok
        return ret

    @torch.no_grad()
    def generate(self, pixel_values=None, input_ids=None, attention_mask=None, visual_features=None, generation_config=None,
                 output_hidden_states=None, **generate_kwargs) -> torch.LongTensor:
        assert self.img_context_token_id is not None
        lm = self.language_model
        if pixel_values is not None:
            B, S = input_ids.shape
            if visual_features is not None:
                vit = visual_features.float().reshape(-1, visual_features.shape[-1]).contiguous()
            else:
                vit = self._mlp1_tokens(self.vision_model.forward_tokens(pixel_values), pixel_values.shape[0])
            ids = input_ids.contiguous()
            rank, _ = ops.scan_image_tokens(ids, self.img_context_token_id)
            emb = ops.embed_tokens(lm.model.tok_embeddings.weight.detach(), ids, rank, vit).reshape(B, S, -1)
        else:
            emb = lm.get_input_embeddings()(input_ids)
        return lm.generate(inputs_embeds=emb, attention_mask=attention_mask, generation_config=generation_config,
                           output_hidden_states=output_hidden_states, use_cache=True, **generate_kwargs)

    def chat(self, tokenizer, pixel_values, question, generation_config, history=None, return_history=False, num_patches_list=None,
             IMG_START_TOKEN="<img>", IMG_END_TOKEN="</img>", IMG_CONTEXT_TOKEN="<IMG_CONTEXT>", verbose=False):
        """chat (:272-335) with the internlm2-chat template (conversation.py: '<|im_start|>role\\n...<|im_end|>' turns)."""
        if history is None and pixel_values is not None and "<image>" not in question:
            question = question + "\n<image>"
        if num_patches_list is None:
            num_patches_list = [pixel_values.shape[0]] if pixel_values is not None else []
        assert pixel_values is None or len(pixel_values) == sum(num_patches_list)
        if getattr(self, "template", "internlm2-chat") != "internlm2-chat":
            raise NotImplementedError(f"conversation template {self.template!r}: only 'internlm2-chat' (the one uLLSAM configures) is built in")
        self.img_context_token_id = tokenizer.convert_tokens_to_ids(IMG_CONTEXT_TOKEN)
        sep = "<|im_end|>"
        eos_token_id = tokenizer.convert_tokens_to_ids(sep)
        history = [] if history is None else history
        query = f"<|im_start|>system\n{self.system_message}{sep}"
        for (q, a) in history:
            query += f"<|im_start|>user\n{q}{sep}<|im_start|>assistant\n{a}{sep}"
        query += f"<|im_start|>user\n{question}{sep}<|im_start|>assistant\n"
        for n in num_patches_list:
            query = query.replace("<image>", IMG_START_TOKEN + IMG_CONTEXT_TOKEN * int(self.num_image_token) * n + IMG_END_TOKEN, 1)
        mi = tokenizer(query, return_tensors="pt")
        generation_config = dict(generation_config)
        generation_config["eos_token_id"] = eos_token_id
        generation_config.pop("output_hidden_states", None)
        out = self.generate(pixel_values=pixel_values, input_ids=mi["input_ids"].to(self.device),
                            attention_mask=mi["attention_mask"].to(self.device), **generation_config)
        response = tokenizer.batch_decode(out, skip_special_tokens=True)[0].split(sep)[0].strip()
        history.append((question, response))
        return (response, history) if return_history else response
