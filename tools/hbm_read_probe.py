import torch, time
x = torch.empty(441_000_000 // 4, dtype=torch.float32, device='cuda').normal_()
y = torch.empty(882_000_000 // 4, dtype=torch.float32, device='cuda').normal_()
for t, name in ((x, '441MB'), (y, '882MB')):
    for op, oname in ((lambda a: a.sum(), 'sum'), (lambda a: a.max(), 'max')):
        for _ in range(5): op(t)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): op(t)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(name, oname, '%.4f ms' % ms, '%.2f TB/s' % (t.numel() * 4 / ms / 1e9))
