"""greedy inference (Transformer.recognize, transformer.py:45-69) throughput at B=32, eval mode, eager launches"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sbl_for_multilingual_lip_reading_amd import detfill
dev = torch.device("cuda", 0)
m = bench.build_model(dev, False).eval()
x = torch.from_numpy(detfill.synthetic_batch(32, 29, 88, 88, 7)[0]).to(dev)
with torch.no_grad():
    for _ in range(2): ys = m.recognize(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): ys = m.recognize(x)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print("recognize: %.1f ms per batch of 32 clips = %.0f clips/s (16 sequential decoder steps, own argmax fed back on the device)" % (dt * 1e3, 32 / dt))
# the same under one hipGraph (inference has a single schedule: every step feeds its own argmax back)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.no_grad(), torch.cuda.stream(s):
    m.recognize(x)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.no_grad(), torch.cuda.graph(g, stream=s):
    ys_g = m.recognize(x)
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): g.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
same = bool((ys_g[0] == ys[0]).all()) and bool((ys_g[1] == ys[1]).all())
print("recognize under hipGraph replay: %.1f ms per batch = %.0f clips/s (same tokens as eager: %s)" % (dt * 1e3, 32 / dt, same))
