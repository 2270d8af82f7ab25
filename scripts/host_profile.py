import cProfile, pstats, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench
dev = torch.device("cuda", 0)
bb, nk = bench.build_models(50, dev)
x = torch.randn(2, 3, 800, 1344, device=dev)
with torch.no_grad():
    outs = nk(bb(x))
cots = [torch.randn_like(o).contiguous(memory_format=torch.channels_last) for o in outs]
params = list(bb.parameters()) + list(nk.parameters())
def step():
    for p in params: p.grad = None
    outs = nk(bb(x)); torch.autograd.backward(outs, cots)
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
