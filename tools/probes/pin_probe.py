import time, torch
dev = torch.device("cuda", 0)
src = torch.randn(256, 512)
n = src.numel() * 4
def t(f, k=50):
    f(); torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(k): f()
    b = time.perf_counter()
    torch.cuda.synchronize()
    return (b - a) / k * 1e3
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
pv = pin.view(torch.float32).view(256, 512)
print("copy into torch pin_memory buffer: %.3f ms" % t(lambda: pv.copy_(src)))
reg = torch.empty(n, dtype=torch.uint8)
rc = torch.cuda.cudart().cudaHostRegister(reg.data_ptr(), n, 0)
rv = reg.view(torch.float32).view(256, 512)
print("hostRegister rc", rc, "is_pinned", reg.is_pinned())
print("copy into registered buffer: %.3f ms" % t(lambda: rv.copy_(src)))
s = torch.cuda.Stream()
def h2d(x):
    with torch.cuda.stream(s):
        return x.to(dev, non_blocking=True)
print("H2D enqueue from pin_memory: %.3f ms" % t(lambda: h2d(pv)))
print("H2D enqueue from registered: %.3f ms" % t(lambda: h2d(rv)))
print("H2D from pageable (blocking inside): %.3f ms" % t(lambda: h2d(src)))
print("fresh pin_memory() + copy per call: %.3f ms" % t(lambda: src.pin_memory()))
import numpy as np
a = src.numpy(); pn = pv.numpy(); rn = rv.numpy()
print("numpy copyto pin: %.3f ms, registered: %.3f ms" % (t(lambda: np.copyto(pn, a)), t(lambda: np.copyto(rn, a))))
print("threads", torch.get_num_threads())
