"""Sam wrapper (API mirror of modeling/sam.py:18-174) over the HIP-backed sub-modules."""
from __future__ import annotations

from typing import Any, Dict, List, Tuple

import torch
from torch import nn

from .. import ops
from .image_encoder import ImageEncoderViT
from .mask_decoder import MaskDecoder
from .prompt_encoder import PromptEncoder


class Sam(nn.Module):
    mask_threshold: float = 0.0
    image_format: str = "RGB"

    def __init__(self, image_encoder: ImageEncoderViT, prompt_encoder: PromptEncoder, mask_decoder: MaskDecoder,
                 pixel_mean: List[float] = [123.675, 116.28, 103.53], pixel_std: List[float] = [58.395, 57.12, 57.375]) -> None:
        super().__init__()
        self.image_encoder = image_encoder
        self.prompt_encoder = prompt_encoder
        self.mask_decoder = mask_decoder
        self.register_buffer("pixel_mean", torch.Tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.Tensor(pixel_std).view(-1, 1, 1), False)

    @property
    def device(self) -> Any:
        return self.pixel_mean.device

    @torch.no_grad()
    def forward(self, batched_input: List[Dict[str, Any]], multimask_output: bool) -> List[Dict[str, torch.Tensor]]:
        """sam.py:53-131.  Images may differ in size: each is normalised + zero padded inside the patch gather."""
        enc = self.image_encoder
        S = enc.img_size
        g = S // enc.patch_size
        mean = self.pixel_mean.reshape(-1).float().contiguous()
        std = self.pixel_std.reshape(-1).float().contiguous()
        shapes = {tuple(r["image"].shape) for r in batched_input}
        if len(shapes) == 1:  # one batched encoder pass
            toks = enc.forward_tokens(torch.stack([r["image"] for r in batched_input], 0).to(self.device), mean, std)
        else:
            toks = torch.cat([enc.forward_tokens(r["image"][None].to(self.device), mean, std) for r in batched_input], 0)
        pe_tok = self.prompt_encoder.dense_pe_tokens()
        sl = slice(1, None) if multimask_output else slice(0, 1)

        def prompts(rec):
            points = (rec["point_coords"], rec["point_labels"]) if "point_coords" in rec else None
            return points, rec.get("boxes", None), rec.get("mask_inputs", None)

        def signature(rec):   # records with equal signatures can share one decoder pass
            pts, bx, mk = prompts(rec)
            return (tuple(rec["image"].shape), tuple(int(v) for v in rec["original_size"]),
                    None if pts is None else (tuple(pts[0].shape), tuple(pts[1].shape)),
                    None if bx is None else tuple(bx.shape), None if mk is None else tuple(mk.shape))

        outputs = []
        if len(batched_input) > 1 and len({signature(r) for r in batched_input}) == 1:
            # The reference loops over the images (sam.py:96-129); every image's prompts are independent of the other images', so the loop is one
            # decoder pass over all (image, prompt) pairs: the prompt tensors are stacked, every prompt gets its image's tokens
            # (tests/test_model_gpu.py::test_sam_forward_batched_equals_per_image: same bits as the loop).
            cat = lambda xs: None if xs[0] is None else torch.cat([x.to(self.device) for x in xs], 0)
            pr = [prompts(r) for r in batched_input]
            points = None if pr[0][0] is None else (cat([p[0][0] for p in pr]), cat([p[0][1] for p in pr]))
            boxes, masks_in = cat([p[1] for p in pr]), cat([p[2] for p in pr])
            bs = self.prompt_encoder._get_batch_size(*pr[0])
            P = bs * len(batched_input)
            sparse = self.prompt_encoder.sparse_tokens(points, boxes)
            if sparse.shape[1] == 0:
                sparse = sparse.expand(P, 0, sparse.shape[-1])
            dense = self.prompt_encoder.dense_tokens(P, masks_in, None)
            img = toks if bs == 1 else toks.repeat_interleave(bs, 0)
            low, iou = self.mask_decoder.predict_masks_tokens(img, pe_tok, sparse, dense, (g, g))
            low, iou = low[:, sl].contiguous(), iou[:, sl]
            rec0 = batched_input[0]
            masks = self.postprocess_masks(low, input_size=rec0["image"].shape[-2:], original_size=rec0["original_size"], _threshold=self.mask_threshold)
            for i in range(len(batched_input)):
                outputs.append({"masks": masks[i * bs:(i + 1) * bs], "iou_predictions": iou[i * bs:(i + 1) * bs], "low_res_logits": low[i * bs:(i + 1) * bs]})
            return outputs
        for i, rec in enumerate(batched_input):
            points, boxes, masks_in = prompts(rec)
            bs = self.prompt_encoder._get_batch_size(points, boxes, masks_in)
            sparse = self.prompt_encoder.sparse_tokens(points, boxes)
            if sparse.shape[1] == 0:
                sparse = sparse.expand(bs, 0, sparse.shape[-1])
            dense = self.prompt_encoder.dense_tokens(bs, None if masks_in is None else masks_in.to(self.device), None)
            low, iou = self.mask_decoder.predict_masks_tokens(toks[i:i + 1], pe_tok, sparse, dense, (g, g))
            low, iou = low[:, sl].contiguous(), iou[:, sl]
            masks = self.postprocess_masks(low, input_size=rec["image"].shape[-2:], original_size=rec["original_size"],
                                           _threshold=self.mask_threshold)
            outputs.append({"masks": masks, "iou_predictions": iou, "low_res_logits": low})
        return outputs

    def postprocess_masks(self, masks: torch.Tensor, input_size: Tuple[int, ...], original_size: Tuple[int, ...],
                          _threshold=None) -> torch.Tensor:
        """sam.py:133-162: bilinear to img_size, crop the padding, bilinear to original_size.
        With `_threshold` set, returns the boolean masks (`> threshold`) directly."""
        S = self.image_encoder.img_size
        up, _ = ops.resize_bilinear(masks.float().contiguous(), (S, S))
        oh, ow = int(original_size[0]), int(original_size[1])
        out, m = ops.resize_bilinear(up, (oh, ow), valid_hw=(int(input_size[0]), int(input_size[1])),
                                     want_float=_threshold is None, threshold=_threshold)
        return out if _threshold is None else m.bool()

    def preprocess(self, x: torch.Tensor) -> torch.Tensor:
        """sam.py:164-174 (API only; forward() fuses this into the patch gather)."""
        x = (x - self.pixel_mean) / self.pixel_std
        S = self.image_encoder.img_size
        h, w = x.shape[-2:]
        return torch.nn.functional.pad(x, (0, S - w, 0, S - h))
