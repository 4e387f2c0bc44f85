import torch
dev="cuda:0"
out = torch.zeros((1024*196, 768), dtype=torch.bfloat16, device=dev)
src = torch.zeros_like(out)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    ts=[]
    for _ in range(5):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/n*1e3)
    return sorted(ts)[2]
mb = out.numel()*2/1e6
a=t(lambda: out.zero_()); print(f"fill {mb:.0f} MB: {a:.1f} us = {mb/a:.2f} TB/s")
b=t(lambda: out.copy_(src)); print(f"copy {mb:.0f} MB (read + write {2*mb:.0f}): {b:.1f} us = {2*mb/b:.2f} TB/s")
u8 = torch.zeros(int(mb*1e6/2), dtype=torch.uint8, device=dev)
c=t(lambda: out.view(-1).copy_(u8)); print(f"u8 -> bf16 convert ({mb/2:.0f} MB read + {mb:.0f} MB write): {c:.1f} us = {1.5*mb/c:.2f} TB/s")
