import os, sys
sys.path.insert(0, "st-gcn-altformer_amd"); sys.path.insert(0, ".")
import torch, bench, stgcn_amd
dev = torch.device("cuda:0")
x = bench.synthetic_clips(256, 180, 22, 0).to(dev)
gcn, tcn = bench.build_stem(22, "SHRE", sys.argv[1] if len(sys.argv) > 1 else "f16mx")
gcn, tcn = gcn.to(dev).eval(), tcn.to(dev).eval()
stgcn_amd.enable_stem_fusion(gcn, tcn)
buf = torch.zeros(8 * 8 * 8, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3): tcn(gcn(x))
    torch.cuda.synchronize()
    os.environ["STGCN_DBG_PTR"] = hex(buf.data_ptr())
    tcn(gcn(x)); torch.cuda.synchronize()
t = buf.cpu().view(8, 8, 8).double()[:, :4, :]
tiles = 16
names = ["chunk0", "main loop total", "pair-end wait+barrier", "epilogue", "pairs 1,2,5,6,7,8 (sum of 6)", "pair 3 (MX, no production)", "pair 4 (tap-8, 96 slots)", "pair 0 (MX + 3 producer blocks)"]
for i, nm in enumerate(names):
    v = t[:, :, i].mean() / tiles
    per = v / 4 if i in (4, 5, 6, 7, 2) else v
    print(f"{nm:36s} {v:10.0f} cycles/tile   {per:9.0f} per period" + (f"  ({per/6:7.0f} per pair)" if i == 4 else ""))
