"""HBM calibration on the box that runs the bench: device-to-device copy, read-only reduction
and fill rates through torch (rocm), to put the kernels' GB/s in context."""
import torch, time
dev = 'cuda:0'
n = 512 * 1024 * 1024 // 8          # 512 MB of float64
x = torch.randn(n, dtype=torch.float64, device=dev)
y = torch.empty_like(x)
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3
b = n * 8
dt = t(lambda: y.copy_(x)); print('copy   %.0f us  %.2f TB/s (read+write bytes)' % (dt * 1e6, 2 * b / dt / 1e12))
dt = t(lambda: x.sum());    print('reduce %.0f us  %.2f TB/s (read)' % (dt * 1e6, b / dt / 1e12))
dt = t(lambda: y.fill_(1.0)); print('fill   %.0f us  %.2f TB/s (write)' % (dt * 1e6, b / dt / 1e12))
dt = t(lambda: torch.add(x, 1.0, out=y)); print('add    %.0f us  %.2f TB/s (read+write)' % (dt * 1e6, 2 * b / dt / 1e12))
