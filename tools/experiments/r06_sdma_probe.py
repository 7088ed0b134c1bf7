import torch, os
n = 48 << 20
a = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
p = torch.empty(n, dtype=torch.uint8, pin_memory=True)
s = torch.cuda.Stream()
torch.cuda.synchronize()
print("=== copy begins", flush=True)
with torch.cuda.stream(s):
    p.copy_(a, non_blocking=True)
s.synchronize()
print("=== copy ends", flush=True)
