"""HBM stream microbenchmark (the measured counterpart of the 8 TB/s vendor figure):
device-to-device copy, read-only reduction and write-only fill of a 4 GiB float64 buffer through PyTorch's
kernels, timed with HIP events.  Prints GB/s (read + write bytes counted for the copy)."""
import torch

n = 1 << 29            # 4 GiB of float64
x = torch.ones(n, dtype=torch.float64, device="cuda:0")
y = torch.empty_like(x)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


t = timed(lambda: y.copy_(x))
print(f"copy  (read 4 GiB + write 4 GiB): {t*1e3:8.3f} ms  {2 * 8 * n / t / 1e9:8.1f} GB/s")
t = timed(lambda: x.sum())
print(f"read  (sum of 4 GiB)            : {t*1e3:8.3f} ms  {8 * n / t / 1e9:8.1f} GB/s")
t = timed(lambda: y.fill_(2.0))
print(f"write (fill of 4 GiB)           : {t*1e3:8.3f} ms  {8 * n / t / 1e9:8.1f} GB/s")
