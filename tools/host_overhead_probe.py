import sys, time
sys.path.insert(0, "/root/repo/video-to-action-release_amd")
import torch
from v2a_hip import ops
x = torch.randn(1024, device="cuda:0"); y = torch.randn(1024, device="cuda:0"); out = torch.empty_like(x)
def t(f, n=20000):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    dt = (time.perf_counter() - t0) / n; torch.cuda.synchronize(); return dt * 1e6
print("torch.cuda.current_stream().cuda_stream", t(lambda: torch.cuda.current_stream().cuda_stream))
print("_cuda_getCurrentRawStream", t(lambda: torch._C._cuda_getCurrentRawStream(0)))
print("empty_like", t(lambda: torch.empty_like(x)))
print("data_ptr x3", t(lambda: (x.data_ptr(), y.data_ptr(), out.data_ptr())))
print("axpy (out given)", t(lambda: ops.axpy(x, y, 1.0, out=out), 5000))
print("axpy (alloc)", t(lambda: ops.axpy(x, y), 5000))
print("torch add", t(lambda: torch.add(x, y, out=out), 5000))
