"""Do consecutive bench steps on alternating HIP streams (two steps in flight) finish sooner than on one stream?  Every GEMM's partial last
round of tiles and every kernel's ramp-up / drain leaves CUs idle that the other stream's kernels can take."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = "cuda"
model = bench.build_model("h", "7b", torch.bfloat16, dev)
inputs = bench.make_inputs(4, 1081, dev, True)
step = bench.mask_path_compute(model, inputs, torch.bfloat16)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(n, nstreams):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if nstreams == 1:
        for _ in range(n): step()
    else:
        for s in streams: s.wait_stream(torch.cuda.current_stream())
        for k in range(n):
            with torch.cuda.stream(streams[k % nstreams]):
                step()
        for s in streams: torch.cuda.current_stream().wait_stream(s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    for _ in range(3): step()
    run(4, 2)
    for r in range(4):
        a = run(8, 1); b = run(8, 2)
        print(f"one stream {a:.2f} ms/step   two streams {b:.2f} ms/step")
    # same outputs either way
    torch.cuda.synchronize()
    low1, mk1 = step(); torch.cuda.synchronize()
    with torch.cuda.stream(streams[0]): low2, mk2 = step()
    torch.cuda.synchronize()
    print("outputs equal:", torch.equal(low1, low2), torch.equal(mk1, mk2))
