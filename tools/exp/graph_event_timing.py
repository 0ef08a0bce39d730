"""Can HIP events recorded INSIDE a captured graph be used for timing after replay?  (feasibility probe)"""
import torch
x = torch.zeros(64 * 1024 * 1024, device='cuda')
s = torch.cuda.Stream()
for ext in (False, True):
    try:
        kw = dict(enable_timing=True)
        if ext:
            kw['external'] = True
        e0, e1 = torch.cuda.Event(**kw), torch.cuda.Event(**kw)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            e0.record()
            x.add_(1.0)
            x.mul_(0.5)
            e1.record()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        print("external=%s elapsed %.3f ms" % (ext, e0.elapsed_time(e1)))
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record(); x.add_(1.0); x.mul_(0.5); a1.record(); torch.cuda.synchronize()
        print("   eager elapsed %.3f ms" % a0.elapsed_time(a1))
    except Exception as exc:
        print("external=%s failed: %r" % (ext, exc))
