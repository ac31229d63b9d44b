"""Timing of the training GEMM (csrc/train.hip) on the three shapes of a ResnetFC layer: forward (x W^T), dgrad (dy W), wgrad (dy^T x).
usage: python tools/time_gemm.py [rows]   (rows = columns of the batch: rays x samples x views, default 327680)"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from diner_amd import train
M = int(sys.argv[1]) if len(sys.argv) > 1 else 327680
N = K = 512
x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; dy = torch.randn(M, N, device="cuda")
y = torch.empty(M, N, device="cuda"); dx = torch.empty(M, K, device="cuda"); dW = torch.zeros(N, K, device="cuda")
cases = {
    "forward  y = relu(x) W^T + b": lambda f: train.gemm(x, W, y, M, N, K, K, K, N, train.TB | train.RELU_A | f),
    "dgrad    dx = dy W        ": lambda f: train.gemm(dy, W, dx, M, K, N, N, K, K, f),
    "wgrad    dW = dy^T relu(x)": lambda f: train.gemm(dy, x, dW, N, K, M, N, K, K, train.TA | train.ATOMIC | train.RELU_B | f,
                                                        k_split=max(1, min(32, M // 1024))),
}
y2 = torch.empty(M, N, device="cuda")
for label, fn in (("forward  y = relu(x) W^T     [lin512   ]", lambda: train.linear512(x, W, y2, relu_in=True)),
                  ("dgrad    dx = dy W            [lin512   ]", lambda: train.linear512(dy, W, y2, transpose=True))):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(f"{label} {M}x{N}x{K}: {dt*1e3:8.3f} ms = {2.0*M*N*K/dt/1e12:7.1f} TFLOP/s (fp32-equivalent; includes packing W)")
dW2 = torch.zeros(N, K, device="cuda"); db2 = torch.zeros(N, device="cuda")
sc = torch.empty(train.lib.diner_wgrad512_scratch_bytes(), dtype=torch.uint8, device="cuda")
for how, ws in (("atomics", None), ("partials", sc)):
    fnw = lambda: train.wgrad512(dy, x, dW2, db2, relu_in=True, scratch=ws)
    fnw(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): fnw()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 20
    print(f"wgrad    dW = dy^T relu(x), db [wgrad512 {how:8s}] {M}x{N}x{K}: {dt*1e3:8.3f} ms = {2.0*M*N*K/dt/1e12:7.1f} TFLOP/s (fp32-equivalent)")
for name, fn in cases.items():
    for label, flag in (("bf16x6", 0), ("fp32 MFMA", train.EXACT)):
        fn(flag); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5): fn(flag)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 5
        print(f"{name} [{label:9s}] {M}x{N}x{K}: {dt*1e3:8.3f} ms = {2.0*M*N*K/dt/1e12:7.1f} TFLOP/s (fp32-equivalent)")
