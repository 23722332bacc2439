import torch
dev = "cuda:0"
n = 1 << 20
def timed(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
d = torch.randn(n, 64, device=dev); a = torch.randn(n, 64, device=dev); x = torch.randn(n, 14, device=dev)
d3 = torch.randn(n, 3, device=dev); w3 = torch.randn(3, 64, device=dev); w2 = torch.randn(64, 64, device=dev)
print("dW = d^T a (64x1M)@(1M x64):", timed(lambda: d.t() @ a))
for c in (64, 256, 1024):
    print(f"  split-K bmm chunks={c}:", timed(lambda: torch.bmm(d.view(c, n // c, 64).transpose(1, 2), a.view(c, n // c, 64)).sum(0)))
print("dW1 = d^T x (64x1M)@(1Mx14):", timed(lambda: d.t() @ x))
print(f"  split-K bmm 256:", timed(lambda: torch.bmm(d.view(256, n // 256, 64).transpose(1, 2), x.view(256, n // 256, 14)).sum(0)))
print("dX = d3 @ w3 (1Mx3)@(3x64):", timed(lambda: d3 @ w3))
print("  broadcast:", timed(lambda: (d3.unsqueeze(2) * w3.unsqueeze(0)).sum(1)))
print("  addcmul:", timed(lambda: torch.addcmul(torch.addcmul(d3[:, 0:1] * w3[0], d3[:, 1:2], w3[1]), d3[:, 2:3], w3[2])))
print("dX = d @ w2 (1Mx64)@(64x64):", timed(lambda: d @ w2))
print("dW3 = d3^T a (3x1M)@(1Mx64):", timed(lambda: d3.t() @ a))
print("  bmm256:", timed(lambda: torch.bmm(d3.view(256, n // 256, 3).transpose(1, 2), a.view(256, n // 256, 64)).sum(0)))
print("bias grad d.sum(0):", timed(lambda: d.sum(0)))
print("  via ones @: ", timed(lambda: torch.ones(1, n, device=dev) @ d))
print("fwd x@W1^T (1Mx14)@(14x64):", timed(lambda: x @ torch.randn(14, 64, device=dev)))
print("fwd a@W3^T (1Mx64)@(64x3):", timed(lambda: a @ w3.t()))
