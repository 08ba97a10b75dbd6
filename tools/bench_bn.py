import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbl_for_multilingual_lip_reading_amd import ops
dev = "cuda:0"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, H, C in [("layer1", 22, 64), ("layer2", 11, 128), ("layer3", 6, 256), ("layer4", 3, 512)]:
    rows = 928 * H * H
    dy = torch.randn(rows, C, device=dev); y = torch.randn(rows, C, device=dev); x = torch.randn(rows, C, device=dev)
    mean = torch.zeros(C, device=dev); inv = torch.ones(C, device=dev); gamma = torch.ones(C, device=dev)
    sums = torch.zeros(2 * C, device=dev, dtype=torch.float64); dx = torch.empty_like(x); dres = torch.empty_like(x)
    dg = torch.empty(C, device=dev); db = torch.empty(C, device=dev)
    t1 = timeit(lambda: ops.call("sbl_bn_bwd_reduce", dy.data_ptr(), y.data_ptr(), x.data_ptr(), mean.data_ptr(), inv.data_ptr(), sums.data_ptr(), rows, C, 1, ops._workspace().data_ptr(), ops.WS_BYTES, ops._s()))
    t2 = timeit(lambda: ops.call("sbl_bn_bwd_apply", dy.data_ptr(), y.data_ptr(), x.data_ptr(), mean.data_ptr(), inv.data_ptr(), gamma.data_ptr(), sums.data_ptr(), dx.data_ptr(), dres.data_ptr(), dg.data_ptr(), db.data_ptr(), rows, C, 1, 0, ops._s()))
    mb = rows * C * 4 / 1e6
    print("%s %6.1f MB/tensor  reduce %6.1f us (%.2f TB/s)  apply %6.1f us (%.2f TB/s)" % (name, mb, t1, 3 * mb / t1, t2, 5 * mb / t2), flush=True)
