"""How does a hipGraph replay schedule two independent branches?  Two dependent chains of N small kernels (few workgroups each, so both
fit the chip side by side).  Variants: (a) one chain alone, (b) one graph, side branch captured first then the main branch (what
PolicyEngine._enc_parallel does), (c) one graph, the two branches captured interleaved kernel by kernel, (d) two linear graphs
replayed on two streams, (e) eager on two streams.  Prints the wall time per replay: ~1x chain = concurrent, ~2x = serialised."""
import sys
import time
import torch

dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
w = torch.randn(D, D, device=dev) / D ** 0.5
xs = [torch.randn(D, D, device=dev) for _ in range(2)]
ys = [torch.empty(D, D, device=dev) for _ in range(2)]
side = torch.cuda.Stream()


def chain_step(i):
    torch.mm(xs[i], w, out=ys[i])
    torch.mm(ys[i], w, out=xs[i])


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def capture(body):
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        body()
    return g


for _ in range(3):
    chain_step(0); chain_step(1)
torch.cuda.synchronize()

g_a = capture(lambda: [chain_step(0) for _ in range(N // 2)])


def body_b():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for _ in range(N // 2):
            chain_step(1)
    for _ in range(N // 2):
        chain_step(0)
    main.wait_stream(side)


def body_c():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    for _ in range(N // 2):
        with torch.cuda.stream(side):
            chain_step(1)
        chain_step(0)
    main.wait_stream(side)


g_b = capture(body_b)
g_c = capture(body_c)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s0):
    g_d0 = capture(lambda: [chain_step(0) for _ in range(N // 2)])
with torch.cuda.stream(s1):
    g_d1 = capture(lambda: [chain_step(1) for _ in range(N // 2)])


def run_d():
    cur = torch.cuda.current_stream()
    s0.wait_stream(cur); s1.wait_stream(cur)
    with torch.cuda.stream(s0):
        g_d0.replay()
    with torch.cuda.stream(s1):
        g_d1.replay()
    cur.wait_stream(s0); cur.wait_stream(s1)


def run_e():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    for _ in range(N // 2):
        with torch.cuda.stream(s1):
            chain_step(1)
        chain_step(0)
    cur.wait_stream(s1)


print(f"N = {N} kernels per chain ({D}x{D} fp32 mm)")
print(f"(a) one chain, one graph            {timed(g_a.replay):8.3f} ms")
print(f"(b) 2 branches, side captured first {timed(g_b.replay):8.3f} ms")
print(f"(c) 2 branches, interleaved capture {timed(g_c.replay):8.3f} ms")
print(f"(d) two linear graphs, two streams  {timed(run_d):8.3f} ms")
print(f"(e) eager, two streams              {timed(run_e):8.3f} ms")

# one marked replay of each variant for a rocprofv3 kernel trace (tools/probes/graph_branch_trace.py splits at the marker fills)
marker = torch.zeros(12345, device=dev)
for tag, fn in (("a", g_a.replay), ("b", g_b.replay), ("c", g_c.replay), ("d", run_d), ("e", run_e)):
    torch.cuda.synchronize()
    marker.fill_(1.0)
    torch.cuda.synchronize()
    fn()
    torch.cuda.synchronize()
