"""Minimal stand-ins for transformers.modeling_outputs (attribute + index access, lazily computed logits)."""
from __future__ import annotations


class CausalLMOutputWithPast:
    _FIELDS = ("loss", "logits", "past_key_values", "hidden_states", "attentions")

    def __init__(self, loss=None, logits=None, past_key_values=None, hidden_states=None, attentions=None, logits_fn=None):
        self.loss = loss
        self._logits = logits
        self._logits_fn = logits_fn
        self.past_key_values = past_key_values
        self.hidden_states = hidden_states
        self.attentions = attentions

    @property
    def logits(self):
        """The reference materialises [B,S,vocab] fp32 logits on every forward (modeling_internlm2.py:1081-1082; 400 MB
        per image at S=1081) although the mask path never reads them (app.py:596-606): here they are computed on first use."""
        if self._logits is None and self._logits_fn is not None:
            self._logits = self._logits_fn()
            self._logits_fn = None
        return self._logits

    @logits.setter
    def logits(self, v):
        self._logits = v

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        vals = [getattr(self, f) for f in self._FIELDS if getattr(self, f) is not None]
        return vals[k]

    def __setitem__(self, k, v):
        setattr(self, k, v)


class BaseModelOutputWithPast:
    def __init__(self, last_hidden_state=None, past_key_values=None, hidden_states=None, attentions=None):
        self.last_hidden_state = last_hidden_state
        self.past_key_values = past_key_values
        self.hidden_states = hidden_states
        self.attentions = attentions

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        vals = [v for v in (self.last_hidden_state, self.past_key_values, self.hidden_states, self.attentions) if v is not None]
        return vals[k]
