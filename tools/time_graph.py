"""Does capturing the step loop in a HIP graph pay?  K forward+adjoint steps through the step-level ABI, eager launches vs
one torch.cuda.CUDAGraph replay (the library enqueues everything on torch's current stream).
usage: python tools/time_graph.py [mesh=128] [reps=20]"""
import sys, time
import torch
sys.path.insert(0, ".")
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
K = 10


def timeit(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / K * 1e3


for fwd_only in (True, False):
    r = bench.Runner(n, K, dev)
    r.forward_only = fwd_only
    r.run(K); r.run(K)
    torch.cuda.synchronize()
    eager = timeit(lambda: r.run(K))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        r2 = bench.Runner(n, K, dev)          # a plan bound to the side stream (capture needs a non-default stream)
        r2.forward_only = fwd_only
        r2.run(K); r2.run(K)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            r2.run(K)
    torch.cuda.synchronize()
    graph = timeit(g.replay)
    print(f"mesh {n}^3 {'forward only' if fwd_only else 'forward+adjoint'}: eager {eager:.4f} ms/step, graph {graph:.4f} ms/step", flush=True)
