"""cProfile of the eager step's host side (autograd multithreading off so that backward runs on the profiled thread)."""
import cProfile, pstats, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
bb, nk = bench.build_models(50, dev)
x = torch.randn(2, 3, 800, 1344, device=dev)
with torch.no_grad():
    outs = nk(bb(x))
cots = [torch.randn_like(o).contiguous(memory_format=torch.channels_last) for o in outs]
del outs
params = list(bb.parameters()) + list(nk.parameters())
def step():
    for p in params: p.grad = None
    outs = nk(bb(x)); torch.autograd.backward(outs, cots)
with torch.autograd.set_multithreading_enabled(False):
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): step()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print("enqueue per step: %.2f ms" % ((t1 - t0) / 5 * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5): step()
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(30)
