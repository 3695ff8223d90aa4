"""Does a forked side stream overlap with the main stream (eager and under hipGraph replay)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
from v2a_hip import ops
dev = "cuda:0"
x = torch.randn(64, 1, 16, 256, device=dev)
w = torch.randn(256, 5 * 256, device=dev) * 0.02
side = torch.cuda.Stream()
N = 40
ops.conv2d(x, w, None, 256, 1, 5, (1, 1), (0, 2))
with torch.cuda.stream(side), ops.ws_lane(1):
    ops.conv2d(x, w, None, 256, 1, 5, (1, 1), (0, 2))
torch.cuda.synchronize()


def serial():
    for _ in range(2 * N):
        ops.conv2d(x, w, None, 256, 1, 5, (1, 1), (0, 2))


def forked():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side), ops.ws_lane(1):
        for _ in range(N):
            ops.conv2d(x, w, None, 256, 1, 5, (1, 1), (0, 2))
    for _ in range(N):
        ops.conv2d(x, w, None, 256, 1, 5, (1, 1), (0, 2))
    main.wait_stream(side)


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


print("eager  serial %.3f ms  forked %.3f ms" % (timeit(serial), timeit(forked)))
for name, fn in (("serial", serial), ("forked", forked)):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    print("graph ", name, "%.3f ms" % timeit(g.replay))
