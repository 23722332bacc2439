import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "valubench.so"))
out = torch.empty(256 * 4096, device="cuda:0")
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
names = ["v_fma_f32 x8 indep", "v_pk_fma_f32 x8 indep", "v_pk_mul_f32 x8", "v_mul_f32 x8", "v_fma_f32 dep chain", "v_rcp_f32 x8", "v_mov_b32 x8"]
iters = 20000
for wps in (1, 2, 4, 8):          # waves per SIMD: blocks of 4 waves -> blocks per CU = wps
    blocks = 256 * wps
    for mode, nm in enumerate(names):
        lib.vb(mode, C.c_void_p(out.data_ptr()), blocks, 100, s); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.vb(mode, C.c_void_p(out.data_ptr()), blocks, iters, s); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        instr_per_simd = wps * iters * 8
        ns_per_instr = ms * 1e6 / instr_per_simd
        print(f"waves/SIMD={wps} {nm:24s}: {ms:7.3f} ms  {ns_per_instr:6.3f} ns per wave-instruction per SIMD (= {ns_per_instr*2.4:5.2f} cyc @2.4GHz)")
