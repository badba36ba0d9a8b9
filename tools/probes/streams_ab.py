"""Same-process A/B of the bench step on 1 vs N HIP streams; interleaved ABBA rounds.  Round 4 (MI355X, configs[2]): 1 stream 77.6 ms per step,
2 streams 91.0, 4 streams 105.3 -- not used by bench.py.
usage: python tools/probes/streams_ab.py [rounds] [1,2,4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

def multi_stream_compute(model, inputs, dtype, streams: int):
    """The batch cut into `streams` contiguous shards, each running bench.mask_path_compute on its own HIP stream, forked from and joined to the
    caller's stream (the first call runs on one stream: lazily packed weights are produced before several streams read them)."""
    x32, pts, lbl, ids = inputs
    B = x32.shape[0]
    from ullsam_amd.parallel import shard_range
    spans = [shard_range(B, r, streams) for r in range(streams)]
    subs = [bench.mask_path_compute(model, tuple(None if t is None else t[a:b].contiguous() for t in inputs), dtype) for a, b in spans]
    side = [torch.cuda.Stream() for _ in range(streams)]
    whole = bench.mask_path_compute(model, inputs, dtype)
    state = {"first": True}

    def compute():
        if state["first"]:      # lazily packed weights / cached tables are produced on ONE stream before several streams read them
            state["first"] = False
            return whole()
        cur = torch.cuda.current_stream()
        outs = []
        for st, fn in zip(side, subs):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(fn())
        for st, o in zip(side, outs):
            cur.wait_stream(st)
            for t in o:
                t.record_stream(cur)     # allocated on the side stream, consumed on the caller's
        return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])

    return compute



rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ns = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "2"])]
batch = int(os.environ.get("BATCH", "4"))
dev = "cuda"
model = bench.build_model("h", "7b", torch.bfloat16, dev)
inputs = bench.make_inputs(batch, 1081, dev, True)
steps = {n: (bench.mask_path_compute(model, inputs, torch.bfloat16) if n == 1 else multi_stream_compute(model, inputs, torch.bfloat16, n)) for n in ns}
outs = {}
with torch.no_grad():
    for n in ns:
        for _ in range(3):
            low, mk = steps[n]()
        outs[n] = (low.float().clone(), mk.clone())
    torch.cuda.synchronize()
    times = {n: [] for n in ns}
    for r in range(rounds):
        for n in (ns if r % 2 == 0 else ns[::-1]):
            steps[n]()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                steps[n]()
            e1.record()
            torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / 4)
for n in ns:
    t = sorted(times[n])
    d = (outs[n][0] - outs[ns[0]][0]).abs().max().item()
    fl = (outs[n][1] != outs[ns[0]][1]).float().mean().item()
    print(f"streams {n}: median {t[len(t) // 2]:.3f} ms/step (min {t[0]:.3f}, max {t[-1]:.3f}) = {batch * 1e3 / t[len(t) // 2]:.2f} images/s; "
          f"logits vs streams {ns[0]}: max abs diff {d:.2e}, mask pixels flipped {fl:.2e}")
