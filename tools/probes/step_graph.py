"""Same-process A/B: the bench step launched kernel by kernel on the stream vs the same step captured once in a HIP graph and replayed.
usage: python tools/probes/step_graph.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = "cuda"
model = bench.build_model("h", "7b", torch.bfloat16, dev)
inputs = bench.make_inputs(4, 1081, dev, True)
step = bench.mask_path_compute(model, inputs, torch.bfloat16)


def timed(fn, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    for _ in range(3):
        ref = step()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    g.replay()
    torch.cuda.synchronize()
    print("captured; logits equal:", torch.equal(out[0], ref[0]), " masks equal:", torch.equal(out[1], ref[1]), flush=True)
    te, tg = [], []
    for r in range(rounds):
        for which in ((0, 1) if r % 2 == 0 else (1, 0)):
            if which == 0:
                te.append(timed(step))
            else:
                tg.append(timed(g.replay))
    te.sort(); tg.sort()
    print(f"stream launches: median {te[len(te) // 2]:.3f} ms/step   graph replay: median {tg[len(tg) // 2]:.3f} ms/step")
