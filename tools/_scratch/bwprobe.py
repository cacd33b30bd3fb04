import torch, time
for mb in (6.3, 12.5, 25, 50, 100):
    n = int(mb * 1e6 / 4)
    a = torch.randn(n, device='cuda'); b = torch.empty_like(a)
    for _ in range(20): b.copy_(a)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(200): b.copy_(a)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 200 * 1e3
    print("copy %.1f MB read + %.1f MB write: %.2f us  -> %.2f TB/s (r+w)" % (mb, mb, us, 2 * mb / us * 1e-6 * 1e6 / 1e6))
# 4 reads + 3 writes like adam (theta,m,v,g -> theta,m,v), 1.57M elements
n = 1573512
th, m, v, g = [torch.randn(n, device='cuda') for _ in range(4)]
def step():
    torch._foreach_add_([m], [g], alpha=0.1)
for _ in range(10): step()
torch.cuda.synchronize()
