"""InternVLChatConfig with the reference's fields (modeling/configuration_internvl_chat.py:20-100)."""
from __future__ import annotations

import copy

from .configuration_internlm2 import InternLM2Config


class InternVLChatConfig:
    model_type = "internvl_chat"
    is_composition = True

    def __init__(self, vision_config=None, llm_config=None, use_backbone_lora=0, use_llm_lora=0, select_layer=-1,
                 force_image_size=None, downsample_ratio=0.5, template=None, dynamic_image_size=False, use_thumbnail=False,
                 ps_version="v1", min_dynamic_patch=1, max_dynamic_patch=6, img_context_token_id=None, **kwargs):
        if vision_config is None:
            vision_config = {"architectures": ["InternVisionModel"]}
        if llm_config is None:
            llm_config = {"architectures": ["InternLM2ForCausalLM"]}
        self.vision_config = dict(vision_config)  # vestigial in uLLSAM (SAM's ViT is passed in as a module)
        arch = llm_config.get("architectures")[0]
        if arch in ("InternLM2ForCausalLM", "InternVLChatModel"):
            self.llm_config = InternLM2Config(**llm_config)
        else:
            raise ValueError("Unsupported architecture: {}".format(arch))
        self.use_backbone_lora, self.use_llm_lora = use_backbone_lora, use_llm_lora
        self.select_layer = select_layer
        self.force_image_size = force_image_size
        self.downsample_ratio = downsample_ratio
        self.template = template
        self.dynamic_image_size, self.use_thumbnail = dynamic_image_size, use_thumbnail
        self.ps_version = ps_version
        self.min_dynamic_patch, self.max_dynamic_patch = min_dynamic_patch, max_dynamic_patch
        self.img_context_token_id = img_context_token_id
        self.use_return_dict = kwargs.pop("return_dict", True)
        for k, v in kwargs.items():
            setattr(self, k, v)

    def to_dict(self):
        out = copy.deepcopy({k: v for k, v in self.__dict__.items() if k != "llm_config"})
        out["llm_config"] = self.llm_config.to_dict()
        out["model_type"] = self.model_type
        return out
