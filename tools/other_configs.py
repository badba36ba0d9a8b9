"""The BASELINE.json configurations besides the default bench line, in one go -> profiles/rNN_other_configs.json
(raw JSON lines of bench.py / tools/amg_bench.py, stamped with the git revision passed in; the GPU box has no .git).
usage: python tools/other_configs.py --round 02 --head <git sha>"""
import argparse, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS = [
    ("configs[1] ViT-H + mask decoder, batch 8", ["bench.py", "--llm", "none", "--batch", "8", "--no-cpu-baseline"], {}),
    ("configs[1] with fp8 (e4m3) ViT linears (bench.py --vit-fp8)", ["bench.py", "--llm", "none", "--batch", "8", "--no-cpu-baseline", "--vit-fp8"], {}),
    ("decode (bench.py --mode decode)", ["bench.py", "--mode", "decode"], {}),
    ("config 3' ViT-B + 2B-shaped, batch 4", ["bench.py", "--vit", "b", "--llm", "2b", "--no-cpu-baseline", "--no-iou"], {}),
    ("configs[4] AMG 64x64 points on a 2048^2 tile (bf16)", ["tools/amg_bench.py"], {}),
    ("configs[4] AMG 64x64 points on a 2048^2 tile (fp8 ViT linears)", ["tools/amg_bench.py"], {"ULLSAM_FP8": "1"}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="03")
    ap.add_argument("--head", default="unknown")
    a = ap.parse_args()
    out = {"git_head": a.head, "note": "one call of tools/other_configs.py, default steps/warmup; raw lines of bench.py / tools/amg_bench.py"}
    for name, cmd, env in RUNS:
        r = subprocess.run([sys.executable, *cmd], cwd=ROOT, capture_output=True, text=True, env=dict(os.environ, **env))
        line = next((l for l in reversed(r.stdout.splitlines()) if l.startswith("{")), None)
        out[name] = json.loads(line) if line else {"error": (r.stderr or r.stdout)[-400:]}
        print(name, "->", (line or "FAILED")[:160], flush=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", f"r{a.round}_other_configs.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
