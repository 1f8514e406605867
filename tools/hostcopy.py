"""Diagnostic: host <-> device copy paths a NumPy-boundary step uses (CPU writes into / reads from pinned memory, H2D from pinned / pageable / NumPy-owned memory,
D2H into pinned) timed in isolation on the GPU box (not a benchmark)."""
import sys, time, os, numpy as np, torch
sys.path.insert(0, '/root/repo')
def t(f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
dev = torch.device("cuda", 0)
pin = torch.empty((4096, 10, 25), dtype=torch.float32, pin_memory=True)
pag = torch.empty((4096, 10, 25), dtype=torch.float32)
src32 = np.random.rand(4096, 10, 25).astype(np.float32); src64 = src32.astype(np.float64)
d = torch.empty((4096, 10, 25), dtype=torch.float32, device=dev)
print("CPU write 4MB f32->pinned  %.3f ms" % t(lambda: pin.copy_(torch.as_tensor(src32))))
print("CPU write 4MB f32->pageable %.3f ms" % t(lambda: pag.copy_(torch.as_tensor(src32))))
print("CPU convert f64->pinned f32 %.3f ms" % t(lambda: pin.copy_(torch.as_tensor(src64))))
print("CPU convert f64->pageable f32 %.3f ms" % t(lambda: pag.copy_(torch.as_tensor(src64))))
print("numpy astype f64->f32 %.3f ms" % t(lambda: src64.astype(np.float32)))
print("H2D from pinned 4MB %.3f ms" % t(lambda: d.copy_(pin, non_blocking=True)))
print("H2D from pageable torch 4MB %.3f ms" % t(lambda: d.copy_(pag, non_blocking=True)))
print("H2D from numpy-owned 4MB %.3f ms" % t(lambda: d.copy_(torch.as_tensor(src32), non_blocking=True)))
big = torch.empty((4096, 10, 20, 8), dtype=torch.float32, device=dev).normal_()
hp = torch.empty(big.shape, dtype=torch.float32, pin_memory=True)
print("D2H 26MB to pinned %.3f ms" % t(lambda: hp.copy_(big, non_blocking=True)))
arr = hp.numpy(); dst = np.empty_like(arr)
print("CPU read 26MB from pinned (np.copyto) %.3f ms" % t(lambda: np.copyto(dst, arr)))
arr2 = np.random.rand(*arr.shape).astype(np.float32)
print("CPU read 26MB from pageable (np.copyto) %.3f ms" % t(lambda: np.copyto(dst, arr2)))
# interplay: H2D from pinned then D2H
def both():
    d.copy_(pin, non_blocking=True); hp.copy_(big, non_blocking=True); torch.cuda.current_stream().synchronize()
print("H2D(pinned 4MB) + D2H(26MB) + sync %.3f ms" % t(both))
def both2():
    d.copy_(torch.as_tensor(src32), non_blocking=True); hp.copy_(big, non_blocking=True); torch.cuda.current_stream().synchronize()
print("H2D(numpy 4MB) + D2H(26MB) + sync %.3f ms" % t(both2))
print("threads", torch.get_num_threads())
