"""InternLM2Config with the reference's field names and defaults (modeling/configuration_internlm2.py:77-133).
A plain Python class: the HIP path needs the numbers, not HF's PretrainedConfig machinery."""
from __future__ import annotations

import copy


class InternLM2Config:
    model_type = "internlm2"

    def __init__(self, vocab_size=103168, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                 num_attention_heads=32, num_key_value_heads=None, hidden_act="silu", max_position_embeddings=2048,
                 initializer_range=0.02, rms_norm_eps=1e-6, use_cache=True, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                 tie_word_embeddings=False, bias=True, rope_theta=10000, rope_scaling=None, attn_implementation="eager", **kwargs):
        self.vocab_size = vocab_size
        self.max_position_embeddings = max_position_embeddings
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.bias = bias
        self.num_key_value_heads = num_attention_heads if num_key_value_heads is None else num_key_value_heads
        self.hidden_act = hidden_act
        self.initializer_range = initializer_range
        self.rms_norm_eps = rms_norm_eps
        self.use_cache = use_cache
        self.rope_theta = rope_theta
        self.rope_scaling = rope_scaling
        self._rope_scaling_validation()
        self.attn_implementation = attn_implementation or "eager"
        self.pad_token_id, self.bos_token_id, self.eos_token_id = pad_token_id, bos_token_id, eos_token_id
        self.tie_word_embeddings = tie_word_embeddings
        self.architectures = kwargs.pop("architectures", ["InternLM2ForCausalLM"])
        self.output_attentions = kwargs.pop("output_attentions", False)
        self.output_hidden_states = kwargs.pop("output_hidden_states", False)
        self.use_return_dict = kwargs.pop("return_dict", True)
        for k, v in kwargs.items():
            setattr(self, k, v)

    def _rope_scaling_validation(self):
        """configuration_internlm2.py:134-150."""
        if self.rope_scaling is None:
            return
        if not isinstance(self.rope_scaling, dict) or len(self.rope_scaling) != 2:
            raise ValueError("`rope_scaling` must be a dictionary with with two fields, `type` and `factor`, "
                             f"got {self.rope_scaling}")
        t, f = self.rope_scaling.get("type", None), self.rope_scaling.get("factor", None)
        if t is None or t not in ["linear", "dynamic"]:
            raise ValueError(f"`rope_scaling`'s type field must be one of ['linear', 'dynamic'], got {t}")
        if f is None or not isinstance(f, float) or f < 1.0:
            raise ValueError(f"`rope_scaling`'s factor field must be a float >= 1, got {f}")

    def to_dict(self):
        return copy.deepcopy(self.__dict__)
