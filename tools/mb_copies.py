"""Developer: host <-> device copy paths of the S2 sessions, timed alone (5 MB input chunk, small stats).  python tools/mb_copies.py"""
import time
import numpy as np
import torch
n = 6 * 3 * 48 * 1400
a = np.random.default_rng(0).random(n).astype(np.float32)
pin = torch.empty(n, dtype=torch.float32, pin_memory=True)
dev = torch.empty(n, dtype=torch.float32, device="cuda")
def t(fn, reps=50):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print("MB", n * 4 / 1e6)
print("numpy -> pinned (torch copy_)      %.3f ms" % t(lambda: pin.copy_(torch.from_numpy(a))))
print("numpy -> pinned (np.copyto)        %.3f ms" % t(lambda: np.copyto(pin.numpy(), a)))
print("numpy -> numpy  (np.copyto)        %.3f ms" % t(lambda: np.copyto(np.empty_like(a), a)))
print("pinned -> device (non_blocking)    %.3f ms" % t(lambda: dev.copy_(pin, non_blocking=True)))
print("pageable -> device                 %.3f ms" % t(lambda: dev.copy_(torch.from_numpy(a), non_blocking=True)))
print("pageable -> new device (.to)       %.3f ms" % t(lambda: torch.from_numpy(a).to("cuda", non_blocking=True)))
s = torch.empty(2000, dtype=torch.float32, device="cuda"); sp = torch.empty(2000, dtype=torch.float32, pin_memory=True)
print("small D2H .cpu()                   %.3f ms" % t(lambda: s.cpu()))
print("small D2H pinned + sync            %.3f ms" % t(lambda: (sp.copy_(s, non_blocking=True), torch.cuda.current_stream().synchronize())))
big = torch.empty(6 * 175 * 18710, dtype=torch.float32, device="cuda"); bp = torch.empty(big.numel(), dtype=torch.float32, pin_memory=True)
print("softmax D2H pinned + sync (%.0f MB) %.3f ms" % (big.numel() * 4 / 1e6, t(lambda: (bp.copy_(big, non_blocking=True), torch.cuda.current_stream().synchronize()), 10)))
print("pinned -> numpy copy               %.3f ms" % t(lambda: bp.numpy().copy(), 10))
