from .sam import Sam
from .image_encoder import ImageEncoderViT
from .mask_decoder import MaskDecoder
from .prompt_encoder import PromptEncoder
from .transformer import TwoWayTransformer

__all__ = ["Sam", "ImageEncoderViT", "MaskDecoder", "PromptEncoder", "TwoWayTransformer", "InternVLSAMModel"]


def __getattr__(name):  # lazy: the LLM side pulls in more code
    if name == "InternVLSAMModel":
        from .modeling_internvl_sam import InternVLSAMModel
        return InternVLSAMModel
    raise AttributeError(name)
