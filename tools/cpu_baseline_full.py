"""Validate bench.py's depth-extrapolated cpu_baseline against a FULL-depth run of the numpy oracle on the same host:
the models the reference ships (SURVEY config 3': SAM ViT-B + InternLM2-1.8B-shaped LLM + decoder), one image, S = 1081.
Writes profiles/rNN_cpu_baseline_full_depth.json.   usage: python tools/cpu_baseline_full.py [round]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from oracle import ullsam_oracle as O

rnd = sys.argv[1] if len(sys.argv) > 1 else "02"
v, c = bench.VIT["b"], bench.LLM["2b"]
cfg = dict(hidden=c["hidden_size"], layers=c["num_hidden_layers"], heads=c["num_attention_heads"], kv_heads=c["num_key_value_heads"],
           inter=c["intermediate_size"], vocab=92553, rope_theta=1e6, eps=1e-5)
P = {}
P.update(O.fill_state(O.vit_shapes(embed_dim=v["dim"], depth=v["depth"], num_heads=v["heads"], global_attn_indexes=tuple(v["glob"]), prefix="vision_model."), 0))
P.update(O.fill_state(O.prompt_encoder_shapes(prefix="prompt_encoder."), 0))
P.update(O.fill_state(O.mask_decoder_shapes(prefix="mask_decoder."), 0))
P.update(O.fill_state(O.internlm2_shapes(cfg["hidden"], cfg["layers"], cfg["heads"], cfg["kv_heads"], cfg["inter"], cfg["vocab"], prefix="language_model."), 0))
P.update(O.fill_state(O.projector_shapes(cfg["hidden"]), 0))
x = np.random.default_rng(1).random((1, 3, 1024, 1024), dtype=np.float32)
ids = bench.make_input_ids(20, 34, seed=1)
pts, lbl = np.array([[[500.0, 500.0]]], np.float32), np.array([[1]], np.int32)
vcfg = dict(depth=v["depth"], num_heads=v["heads"], global_attn_indexes=tuple(v["glob"]), window_size=14)
ts = []
for rep in range(3):
    t = time.perf_counter()
    O.ullsam_mask_path(P, x, ids, pts, lbl, vcfg, cfg)
    ts.append(time.perf_counter() - t)
full = float(np.median(ts[1:]))
ext = bench.cpu_baseline("b", "2b", 1081)
out = {"workload": "uLLSAM mask path, SAM ViT-B + InternLM2-1.8B-shaped (24 layers) + decoder, 1 image, S = 1081, numpy fp32 oracle",
       "host_cores": os.cpu_count(), "full_depth_seconds_runs": [round(t, 2) for t in ts], "full_depth_seconds": round(full, 2),
       "full_depth_images_per_s": round(1.0 / full, 5), "extrapolated_images_per_s": ext["value"], "extrapolated_stages_s": ext["stages_s"],
       "extrapolated_over_full": round(ext["value"] * full, 3),
       "note": "the full-depth run also computes the lm_head logits of the last position only (as the oracle's mask path does); first run is the warm-up"}
json.dump(out, open(os.path.join(ROOT, "profiles", f"r{rnd}_cpu_baseline_full_depth.json"), "w"), indent=1)
print(json.dumps(out))
