"""How fast can anything write 48 GB here?  torch fill_ / zero_ (hipMemset) / a device-to-device copy, for
comparison with the store-only micro-benchmarks and k_tx."""
import torch
n = 48_000_000_000
x = torch.empty(n // 4, dtype=torch.float32, device="cuda")
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = timed(lambda: x.fill_(1.5)); print("fill_  48 GB: %.2f ms  %.2f TB/s written" % (ms, n / ms / 1e9))
ms = timed(lambda: x.zero_()); print("zero_  48 GB: %.2f ms  %.2f TB/s written" % (ms, n / ms / 1e9))
y = torch.empty(n // 8, dtype=torch.float32, device="cuda")
z = x[: n // 8]
ms = timed(lambda: z.copy_(y)); print("copy_  24 GB -> 24 GB: %.2f ms  %.2f TB/s moved (read+write)" % (ms, n / ms / 1e9))
for gb in (1, 4, 16):
    w = x[: gb * 250_000_000]
    ms = timed(lambda: w.fill_(2.0), 10); print("fill_ %2d GB: %.3f ms  %.2f TB/s" % (gb, ms, gb * 1e9 / ms / 1e9))
print("-- 16 GB windows at different offsets of the 48 GB buffer")
for off_gb in (0, 16, 32):
    w = x[off_gb * 250_000_000:(off_gb + 16) * 250_000_000]
    ms = timed(lambda: w.fill_(2.0), 10); print("fill_ 16 GB at +%2d GB: %.3f ms  %.2f TB/s" % (off_gb, ms, 16e9 / ms / 1e9))
print("-- growing footprint")
for gb in (8, 16, 24, 32, 40, 48):
    w = x[: gb * 250_000_000]
    ms = timed(lambda: w.fill_(2.0), 5); print("fill_ %2d GB: %.3f ms  %.2f TB/s" % (gb, ms, gb * 1e9 / ms / 1e9))
print("-- 48 GB as three launches of 16 GB")
def three():
    for off_gb in (0, 16, 32):
        x[off_gb * 250_000_000:(off_gb + 16) * 250_000_000].fill_(2.0)
ms = timed(three, 5); print("3 x 16 GB: %.3f ms  %.2f TB/s" % (ms, 48e9 / ms / 1e9))
