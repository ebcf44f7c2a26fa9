"""Does the write rate of a plain fill depend on the size of the launch or on the alignment of its base?"""
import torch
x = torch.empty(12_000_000_000, dtype=torch.float32, device="cuda")
def timed(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("base address %% 2 MiB = %d" % (x.data_ptr() % (2 << 20)))
for n in (1_000_000_000, 1_250_000_000, 1_500_000_000, 1_750_000_000, 2_000_000_000, 1 << 30, 3 << 29, 1 << 31 - 1):
    for a in (0, 256, 1024, 1_500_000_000, 1 << 20):
        w = x[a:a + n]
        ms = timed(lambda: w.fill_(2.0))
        print("n = %10d floats at float offset %10d (byte offset %% 4096 = %4d): %.3f ms  %.2f TB/s" % (n, a, (a * 4) % 4096, ms, n * 4 / ms / 1e9))
