import sys, time, ctypes, numpy as np, torch
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
hip = ctypes.CDLL('libamdhip64.so')
def masked_stream(reserve, ncu=256):
    words = (ncu + 31) // 32
    m = (ctypes.c_uint32 * words)()
    if isinstance(reserve, int):
        off = set(range(reserve))
    else:
        off = set(reserve)
    for i in range(ncu):
        if i not in off: m[i // 32] |= (1 << (i % 32))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), words, m)
    assert rc == 0, rc
    return st
lib = _lib.load()
n = 8192
for reserve in (0, [32 * x for x in range(8)], [32 * x + y for x in range(8) for y in range(2)], [8 * x for x in range(32)][:8], list(range(16))):
    sA = masked_stream(reserve)
    ctxA = _lib.Context(0, stream=sA.value)
    stB = torch.cuda.Stream(priority=-1)
    ctxB = _lib.Context(0, stream=stB.cuda_stream)
    A = torch.randn(n, 512, dtype=torch.float64, device='cuda'); C = torch.randn(n, n, dtype=torch.float64, device='cuda')
    D = (torch.eye(128, dtype=torch.float64, device='cuda') * 128 + 0.1).contiguous()
    invd = torch.empty(8 * 256, dtype=torch.float64, device='cuda'); info = torch.zeros(1, dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    # gemm alone
    t0 = time.perf_counter()
    for _ in range(3):
        _lib.check(lib.gpt_dev_gemm_nt(ctxA.handle, n, n, 512, -1.0, A.data_ptr(), 512, A.data_ptr(), 512, 1.0, C.data_ptr(), n, 1))
    ctxA.synchronize(); t_g = (time.perf_counter() - t0) / 3
    # potf2 alone
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stB):
        e0.record(stB)
        for _ in range(5):
            D.copy_(torch.eye(128, dtype=torch.float64, device='cuda') * 128 + 0.1)
            _lib.check(lib.gpt_dev_potrf_panel(ctxB.handle, 128, 128, D.data_ptr(), 128, invd.data_ptr(), info.data_ptr(), 0))
        e1.record(stB); e1.synchronize()
    t_p = e0.elapsed_time(e1) / 5
    # concurrent: launch gemm on A, then potf2 x5 on B
    _lib.check(lib.gpt_dev_gemm_nt(ctxA.handle, n, n, 512, -1.0, A.data_ptr(), 512, A.data_ptr(), 512, 1.0, C.data_ptr(), n, 1))
    _lib.check(lib.gpt_dev_gemm_nt(ctxA.handle, n, n, 512, -1.0, A.data_ptr(), 512, A.data_ptr(), 512, 1.0, C.data_ptr(), n, 1))
    time.sleep(0.0002)
    with torch.cuda.stream(stB):
        e0.record(stB)
        for _ in range(5):
            _lib.check(lib.gpt_dev_potrf_panel(ctxB.handle, 128, 128, D.data_ptr(), 128, invd.data_ptr(), info.data_ptr(), 0))
        e1.record(stB); e1.synchronize()
    t_c = e0.elapsed_time(e1) / 5
    ctxA.synchronize()
    print("reserve %s: gemm alone %.3f ms, potf2 alone %.1f us, potf2 during gemm %.1f us" % (str(reserve)[:40], t_g * 1e3, t_p * 1e3, t_c * 1e3))
