"""Decode (forward-only) rate of the C1 decoder on the HIP path: frames/s for single-frame decodes under torch.no_grad(),
eager and as a captured hipGraph (N4 row: the figure the reference logs as "FPS" in evaluate())."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
args, model = bench.build(sys.argv[1] if len(sys.argv) > 1 else "c1")
model = model.to(dev).eval()
N = 132
idx = torch.tensor([[37 / N]], dtype=torch.float64, device=dev).reshape(1)
with torch.no_grad():
    for _ in range(5):
        out = model(idx, norm_idx=idx)[0]
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(100):
        out = model(idx, norm_idx=idx)[0]
    torch.cuda.synchronize()
    print(f"eager decode : {100 / (time.time() - t0):8.1f} frames/s")
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    from boosting_nerv_amd import _lib as L
    with torch.cuda.stream(s):
        L.ctx()                                  # the capture stream's library context (and its scratch) must exist before the capture
        with torch.cuda.graph(g, stream=s):
            out = model(idx, norm_idx=idx)[0]
    g.replay(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(300):
        g.replay()
    torch.cuda.synchronize()
    print(f"graph decode : {300 / (time.time() - t0):8.1f} frames/s   ({(time.time() - t0) / 300 * 1e3:.3f} ms/frame)")
    # engine.DecodeGraph: the same captured forward with the decoder's weight fragments prepared once (context plan, ABI 4)
    from boosting_nerv_amd.engine import DecodeGraph
    embed = model(idx, norm_idx=idx)[1][0]
    dg = DecodeGraph(model, idx, embed, idx)
    ref = model(idx, embed, norm_idx=idx)[0]
    assert torch.equal(dg(idx, embed, idx)[0], ref)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(300):
        dg.graph.replay()
    torch.cuda.synchronize()
    print(f"DecodeGraph  : {300 / (time.time() - t0):8.1f} frames/s   ({(time.time() - t0) / 300 * 1e3:.3f} ms/frame), {dg.wplan_entries} planned weight tensors")
