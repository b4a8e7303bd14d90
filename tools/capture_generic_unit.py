import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "st-gcn-altformer_amd")); sys.path.insert(0, ROOT)
import torch
from stgcn_amd import unit_agcn, Unit2D, set_math_mode
dev = torch.device("cuda:0")
torch.manual_seed(0)
A = torch.rand(3, 22, 22) * (torch.rand(3, 22, 22) < 0.15)
for cin, cout, stride in ((64, 64, 1), (64, 128, 2)):
    gcn = unit_agcn(cin, cout, A.clone()).to(dev).train()
    tcn = Unit2D(cout, cout, kernel_size=9, stride=stride).to(dev).train()
    set_math_mode(tcn, "bf16x3")
    with torch.no_grad():
        gcn.bn.weight.fill_(1.0)
    params = list(gcn.parameters()) + list(tcn.parameters())
    x = torch.randn(64, cin, 90, 22, device=dev).requires_grad_(True)
    def step():
        y = tcn(gcn(x)); y.backward(torch.ones_like(y))
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 20 * 1e3
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    for p in params: p.grad = None
    x.grad = None
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize(); rep = (time.perf_counter() - t0) / 20 * 1e3
        print(f"unit({cin},{cout},stride {stride}): eager {eager:.3f} ms  graph replay {rep:.3f} ms")
    except Exception as e:
        print(f"unit({cin},{cout},stride {stride}): eager {eager:.3f} ms  CAPTURE FAILED: {str(e)[:300]}")
        break
