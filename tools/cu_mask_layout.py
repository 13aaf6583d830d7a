"""Which compute units a hipExtStreamCreateWithCUMask bit names on this 8-XCD part: a stream copy and an SGEMM (torch) on
streams with different mask patterns.  Finding (MI355X, ROCm 7.2): bit i = XCD i % 8, shader engine (i / 8) % 4, CU i / 32 of
that engine; an XCD that gets no bit at all is NOT restricted.  DESIGN section 10."""
import ctypes as C, time, torch
torch.cuda.init()
hip = C.CDLL("libamdhip64.so")
def mk(fn, words=8):
    s = C.c_void_p()
    m = (C.c_uint32 * words)()
    n = 0
    for i in range(32*words):
        if fn(i): m[i // 32] |= (1 << (i % 32)); n += 1
    r = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(words), m)
    return r, s, n
big = torch.randn(8192, 8192, device="cuda")
a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
pats = {
 "all": lambda i: True,
 "low64": lambda i: i < 64,
 "low128": lambda i: i < 128,
 "mod2": lambda i: i % 2 == 0,
 "mod4": lambda i: i % 4 == 0,
 "mod8": lambda i: i % 8 == 0,
 "pairs_mod4": lambda i: (i // 2) % 4 == 0,
 "quads_mod4": lambda i: (i // 4) % 4 == 0,
 "oct_mod4": lambda i: (i // 8) % 4 == 0,
 "hex_mod2": lambda i: (i // 16) % 2 == 0,
 "first8_of32": lambda i: i % 32 < 8,
 "first16_of32": lambda i: i % 32 < 16,
}
for name, fn in pats.items():
    r, s, n = mk(fn)
    es = torch.cuda.ExternalStream(s.value)
    with torch.cuda.stream(es):
        y = big * 2; c = a @ b
        es.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): y = big * 1.0001
        es.synchronize()
        tm = (time.perf_counter() - t0) / 10
        t0 = time.perf_counter()
        for _ in range(5): c = a @ b
        es.synchronize()
        tc = (time.perf_counter() - t0) / 5
    print("%-14s bits %3d rc %d  stream-copy %.3f ms (%.2f TB/s)   sgemm 4096 %.3f ms (%.1f TF)" % (name, n, r, tm*1e3, 2*big.numel()*4/tm/1e12, tc*1e3, 2*4096**3/tc/1e12), flush=True)
