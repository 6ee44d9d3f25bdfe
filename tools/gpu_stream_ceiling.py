"""What the HBM delivers to the simplest kernels of this library on vectors that fit no cache
(512 MB each): a dot product (read only: lsb_hip_dot_f64), y += a x (read 2, write 1:
lsb_hip_axpy_f64), torch's own copy and fill for comparison.  usage: gpu_stream_ceiling.py [n]"""
import ctypes, sys
sys.path.insert(0, ".")
import torch
import lsbench_amd as la

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
lib = la._lib.load()
assert la.hip_cdna4_init() == 0
a = torch.rand(n, dtype=torch.float64, device="cuda")
b = torch.rand(n, dtype=torch.float64, device="cuda")
out = torch.zeros(4, dtype=torch.float64, device="cuda")
work = torch.zeros(3 * lib.lsb_hip_partials_capacity() + 16, dtype=torch.float64, device="cuda")
alpha = torch.full((1,), 1e-9, dtype=torch.float64, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())


def timed(name, nbytes, fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name:44s} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s  ({nbytes / 1e6:.0f} MB)", flush=True)


timed("dot (lsb_hip_dot_f64): read a, b", 16 * n, lambda: lib.lsb_hip_dot_f64(n, P(a), P(b), P(out), P(work), st))
timed("nrm2 (lsb_hip_nrm2_f64): read a", 8 * n, lambda: lib.lsb_hip_nrm2_f64(n, P(a), P(out), P(work), st))
timed("axpy (lsb_hip_axpy_f64): read x, y, write y", 24 * n, lambda: lib.lsb_hip_axpy_f64(n, P(alpha), P(a), P(b), st))
timed("torch b.copy_(a): read 1, write 1", 16 * n, lambda: b.copy_(a))
timed("torch a.fill_(0.5): write 1", 8 * n, lambda: a.fill_(0.5))
