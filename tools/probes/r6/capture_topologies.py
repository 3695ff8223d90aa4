"""Which fork/join topologies of a torch.cuda.graph capture survive hipStreamEndCapture on this ROCm build?  Each topology runs in its own
process (a crash is a segfault inside capture_end).  Usage (GPU box): python tools/probes/r6/capture_topologies.py"""
import subprocess
import sys

BODY = r'''
import sys, torch
T = sys.argv[1]
dev = "cuda:0"
x = [torch.zeros(1 << 16, device=dev) for _ in range(8)]
A, B, SA, SB, W = (torch.cuda.Stream(device=dev) for _ in range(5))
def k(i, n=3):
    for _ in range(n):
        x[i].add_(1.0)
def body(M):
    k(0)
    if T == "T1":
        A.wait_stream(M)
        with torch.cuda.stream(A): k(1)
        M.wait_stream(A)
    elif T in ("T2", "T4", "T2W"):
        if T == "T2W":
            W.wait_stream(M)
            with torch.cuda.stream(W): k(5, 10)
        A.wait_stream(M)
        with torch.cuda.stream(A):
            for s in range(4):
                k(1)
                SA.wait_stream(A)
                with torch.cuda.stream(SA): k(2)
            k(1)
            A.wait_stream(SA)
            k(1)
        if T == "T4":
            for s in range(4):
                k(0)
                SB.wait_stream(M)
                with torch.cuda.stream(SB): k(3)
            k(0)
            M.wait_stream(SB)
        else:
            k(0, 12)
        M.wait_stream(A)
        if T == "T2W":
            M.wait_stream(W)
    elif T in ("T3", "T3one"):
        for s in range(1 if T == "T3one" else 4):
            k(0)
            SB.wait_stream(M)
            with torch.cuda.stream(SB): k(3)
        k(0)
        M.wait_stream(SB)
    elif T in ("T5", "T6", "T7"):
        A.wait_stream(M); B.wait_stream(M)
        with torch.cuda.stream(A):
            for s in range(4):
                k(1)
                SA.wait_stream(A)
                with torch.cuda.stream(SA): k(2)
            k(1)
            if T != "T6": A.wait_stream(SA)
            k(1)
        with torch.cuda.stream(B):
            for s in range(4):
                k(4)
                if T != "T7":
                    SB.wait_stream(B)
                    with torch.cuda.stream(SB): k(3)
            k(4)
            if T == "T5": B.wait_stream(SB)
            k(4)
        M.wait_stream(A); M.wait_stream(B)
        if T == "T6":
            M.wait_stream(SA); M.wait_stream(SB)
    elif T in ("T8", "T9", "T10"):
        # T8: the trainer's backward as it is + a branch forked from the origin chain; T9: both chains on side streams, their branches joined
        # into the origin only; T10: T9 with the first chain on the origin stream
        W.wait_stream(M)
        with torch.cuda.stream(W): k(5, 10)
        A.wait_stream(M)
        with torch.cuda.stream(A):
            for s in range(4):
                k(1)
                if T != "T8":
                    SA.wait_stream(A)
                    with torch.cuda.stream(SA): k(2)
            k(1, 3)
        if T == "T9":
            B.wait_stream(M)
        with torch.cuda.stream(B if T == "T9" else M):
            for s in range(4):
                k(4)
                SB.wait_stream(B if T == "T9" else M)
                with torch.cuda.stream(SB): k(3)
            k(4, 3)
        if T == "T8":
            M.wait_stream(SB)
            k(0)
        M.wait_stream(A)
        if T == "T9":
            M.wait_stream(B)
        if T != "T8":
            M.wait_stream(SA); M.wait_stream(SB)
        M.wait_stream(W)
    k(0)
for _ in range(2):
    body(torch.cuda.current_stream())
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body(torch.cuda.current_stream())
g.replay(); torch.cuda.synchronize()
print("PASS", T, [float(t[0]) for t in x[:6]])
'''
for T in ("T1", "T2", "T3", "T6", "T8", "T9", "T10"):
    r = subprocess.run([sys.executable, "-c", BODY, T], capture_output=True, text=True, timeout=300)
    out = [ln for ln in r.stdout.splitlines() if ln.startswith("PASS")]
    print(T, "rc", r.returncode, out[-1] if out else (r.stderr.strip().splitlines() or ["?"])[-1][:150], flush=True)
