"""Collect the rocprof evidence bench.py's `roofline` refers to, in one go, on the GPU box:

    python tools/collect_evidence.py --round 02 --head <git sha> [--mode mask|decode]

  1. rocprofv3 --kernel-trace            -- python3 bench.py --steps 4 --warmup 2   -> profiles/rNN_kernel_summary_HEAD.txt
  2. rocprofv3 --kernel-trace --pmc FETCH_SIZE  -- python3 bench.py --steps 2 --warmup 1   } -> profiles/rNN_pmc_bench_traffic.json
  3. rocprofv3 --kernel-trace --pmc WRITE_SIZE  -- python3 bench.py --steps 2 --warmup 1   }
Counter passes are separate (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md "rocprofv3 PMC slots") and never
combined with a trace domain other than the kernel trace; the profiled program follows `--` directly.  Every output is stamped
with the git revision passed in (the GPU box has no .git) and with the digest of the kernel sources; bench.py reports
`traffic: null` when that digest differs from the sources it runs on.  This script itself never touches the GPU.
"""
import argparse
import csv
import glob
import json
import os
import re
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
INIT_PAT = re.compile(r"distribution_elementwise|FillFunctor|fill_kernel|copyBuffer|fillBuffer|direct_copy|random_|normal_")


def short(n):
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
    if n.startswith("_Z"):
        base = re.match(r"_Z(\d+)", n)
        k = int(base.group(1))
        name = n[2 + len(base.group(1)):][:k]
        rest = n[2 + len(base.group(1)) + k:]
        ints = re.findall(r"Li(\d+)E", rest)
        return f"{name}[{'bf16' if 'DF16b' in rest else 'f32'}{'<' + ','.join(ints) + '>' if ints else ''}]"
    return n.split("(")[0][:70]


def run(cmd, log):
    env = dict(os.environ, TMPDIR="/tmp")
    with open(log, "w") as f:
        rc = subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, env=env, cwd=ROOT).returncode
    print(f"[evidence] rc={rc}: {' '.join(cmd[:6])} ... -> {log}", flush=True)
    return rc


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def kernel_summary(path, steps, out, stamp):
    rows = list(csv.DictReader(open(path)))
    tot, gemm = defaultdict(lambda: [0, 0.0]), defaultdict(lambda: [0, 0.0])
    init_us = 0.0
    # model construction ends where the first kernel of this library starts (the torch kernels before it -- random init -- are not part of a
    # step, whatever their names); and the FIRST step packs weights lazily (32 torch.cat launches that build w13, 5 ms): with more than one
    # step in the trace it is left out too, so the per-step figure is that of the steady state bench.py times
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    own = [r for r in rows if "at::native" not in r["Kernel_Name"] and not INIT_PAT.search(r["Kernel_Name"])]
    t_first = int(own[0]["Start_Timestamp"]) if own else 0
    first_note = ""
    if own and steps > 1:
        starts = [int(r["Start_Timestamp"]) for r in own if r["Kernel_Name"] == own[0]["Kernel_Name"]]
        if len(starts) == steps:   # the step's first kernel occurs once per step: cut at the second occurrence
            t_first = starts[1]
            steps -= 1
            first_note = "; first step (lazy weight packing) excluded"
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if INIT_PAT.search(r["Kernel_Name"]) or int(r["Start_Timestamp"]) < t_first:
            init_us += d  # model construction / random init, not part of a step
            continue
        n = short(r["Kernel_Name"])
        tot[n][0] += 1
        tot[n][1] += d
        if "gemm" in n:
            g = (n, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
            gemm[g][0] += 1
            gemm[g][1] += d
    allus = sum(v[1] for v in tot.values())
    with open(out, "w") as f:
        f.write(f"# {stamp}\n# {len(rows)} dispatches; {allus / 1e3:.1f} ms GPU time in step kernels = {allus / 1e3 / steps:.2f} ms per step over {steps} steps "
                f"(model-init kernels excluded: {init_us / 1e3:.1f} ms{first_note})\n")
        f.write(f"{'kernel':58s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'share':>6s}\n")
        for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:36]:
            f.write(f"{n[:58]:58s} {c:6d} {us / 1e3:9.2f} {us / c:9.1f} {100 * us / allus:5.1f}%\n")
        f.write("\n# GEMM launches grouped by workgroup count\n")
        for (n, g), (c, us) in sorted(gemm.items(), key=lambda kv: -kv[1][1])[:20]:
            f.write(f"{n[:40]:40s} wgs={g:6d} calls={c:5d} total_ms={us / 1e3:8.2f} avg_us={us / c:8.1f}\n")
    print(open(out).read())


def pmc_total(d, counter, pat):
    cc = find(d, "*counter_collection.csv")
    tot, ids = 0.0, set()
    for r in csv.DictReader(open(cc)):
        if r["Counter_Name"] == counter and pat(r["Kernel_Name"]):
            tot += float(r["Counter_Value"])
            ids.add(r["Dispatch_Id"])
    return tot, len(ids)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="02")
    ap.add_argument("--head", default="unknown")
    ap.add_argument("--mode", default="mask")
    ap.add_argument("--skip-pmc", action="store_true")
    a = ap.parse_args()
    from ullsam_amd import build as B
    digest = B._digest()
    stamp = f"git {a.head}, kernel-source digest {digest[:16]}"
    scratch = os.path.join(ROOT, "gpurun_out", f"ev_r{a.round}_{a.mode}")
    os.makedirs(scratch, exist_ok=True)
    prof = os.path.join(ROOT, "profiles")
    py = "python3"
    if a.mode == "decode":
        bench = [py, "tools/decode_bench.py"]
        steps = 1
    else:
        bench = [py, "bench.py", "--no-cpu-baseline", "--no-iou", "--no-graph"]
    rp = "/opt/rocm/bin/rocprofv3"
    kt = os.path.join(scratch, "kt")
    args = (["--steps", "4", "--warmup", "2"] if a.mode == "mask" else [])
    run([rp, "--kernel-trace", "--output-format", "csv", "-d", kt, "--", *bench, *args], os.path.join(scratch, "kt.log"))
    ktcsv = find(kt, "*kernel_trace.csv")
    name = "kernel_summary_HEAD" if a.mode == "mask" else "decode_summary"
    kernel_summary(ktcsv, 6 if a.mode == "mask" else 1, os.path.join(prof, f"r{a.round}_{name}.txt"),
                   stamp + (" -- rocprofv3 --kernel-trace -- python3 bench.py --steps 4 --warmup 2 (6 steps traced: 2 warm-up + 4 timed; the 5 after the first counted)" if a.mode == "mask"
                            else " -- rocprofv3 --kernel-trace -- python3 tools/decode_bench.py"))
    if a.skip_pmc or a.mode != "mask":
        return
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(scratch, counter.lower())
        run([rp, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--", *bench, "--steps", "2", "--warmup", "1"],
            os.path.join(scratch, counter.lower() + ".log"))
        is_gemm = lambda n: "gemm" in n and "reduce" not in n
        res[counter] = pmc_total(d, counter, is_gemm)
    (fk, n1), (wk, n2) = res["FETCH_SIZE"], res["WRITE_SIZE"]
    out = {"git_head": a.head, "csrc_digest": digest, "gemm_launches_profiled": n1,
           "fetch_bytes_per_launch": 2 * fk * 1024 / max(n1, 1), "write_bytes_per_launch": wk * 1024 / max(n2, 1),
           "hbm_bytes_per_launch": 2 * fk * 1024 / max(n1, 1) + wk * 1024 / max(n2, 1),
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 2 --warmup 1`; FETCH_SIZE (KiB) x2 "
                     "(gfx950: the counter tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); every GEMM main-kernel dispatch; "
                     "Infinity-Cache hits are included in FETCH_SIZE, so this is L2-miss traffic"}
    json.dump(out, open(os.path.join(prof, f"r{a.round}_pmc_bench_traffic.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
