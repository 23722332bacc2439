import ctypes as C, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "membench.so"))
dev = torch.device("cuda:0")
def timeit(fn, iters=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
vp = lambda t: C.c_void_p(t.data_ptr())
for n in (1 << 20, 1 << 22):
  for ld, tiled in ((n, 0), (n + 1024, 0), (n, 1)):
      st = torch.rand((14, ld), device=dev); st2 = torch.empty_like(st); ep = torch.rand((2, ld), device=dev)
      acts = [torch.rand((n, 3), device=dev) for _ in range(4)]
      obs = torch.empty((n, 14), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
      s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
      k = [0]
      for blocks in (1280, 2048):
          def f():
              k[0] += 1
              lib.mb_pattern(vp(st), vp(st), vp(ep), vp(acts[k[0] % 4]), vp(obs), vp(rew), vp(done), C.c_int64(n), C.c_int64(ld), blocks, s, tiled)
          us = timeit(f)
          print(f"n={n} ld={ld} tiled={tiled} pattern(in-place) blocks={blocks}: {us:7.2f} us  -> {189*n/us*1e-3:7.1f} GB/s")
  nin4 = n * 72 // 16; nout4 = n * 117 // 16
  src = torch.rand(nin4 * 4, device=dev); dst = torch.empty(nout4 * 4, device=dev)
  for blocks in (1024, 2048, 8192):
      us = timeit(lambda: lib.mb_copy(vp(src), vp(dst), C.c_int64(nin4), C.c_int64(nout4), blocks, s))
      print(f"n={n} float4 read72+write117 blocks={blocks}: {us:7.2f} us -> {189*n/us*1e-3:7.1f} GB/s")
  n4 = n * 189 // 32
  a = torch.rand(n4 * 4, device=dev); b = torch.empty_like(a)
  for blocks in (2048, 8192):
      us = timeit(lambda: lib.mb_copy2(vp(a), vp(b), C.c_int64(n4), blocks, s))
      print(f"n={n} float4 copy (same total bytes) blocks={blocks}: {us:7.2f} us -> {189*n/us*1e-3:7.1f} GB/s")
