"""Encode passes of 65536 vectors on one / two / three model handles, each on its own stream: do consecutive passes overlap usefully?
usage: probe_encode_lanes.py [config=c4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
sys.argv = [sys.argv[0]]
import torch
import bench
dev = torch.device("cuda", 0)
models = [bench.load_model(bench.CONFIGS[cfg]["fixture"])[0] for _ in range(3)]
P = bench.mixture_centers(bench.CONFIGS[cfg]["gen"], dev)
n = 65536
chunk = max(1, min(n, 12500))
xs = [torch.cat([bench.gen_chunk(P, c + 7 * l, chunk, dev) for c in range(-(-n // chunk))])[:n].contiguous() for l in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
def run(K, lanes):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(K):
        l = i % lanes
        with torch.cuda.stream(streams[l]):
            models[l].predict_batch_dev(xs[l])
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / K
ref = models[0].predict_batch_dev(xs[0]); torch.cuda.synchronize()
for lanes in (1, 2, 3):
    run(2 * lanes, lanes)
    dt = min(run(12, lanes) for _ in range(3))
    print("%s encode, %d pass(es) of %d vectors in flight: %.3f ms per pass  %.1f M vectors/s" % (cfg, lanes, n, dt * 1e3, n / dt / 1e6))
with torch.cuda.stream(streams[0]):
    again = models[0].predict_batch_dev(xs[0])
torch.cuda.synchronize()
print("same codes:", bool(torch.equal(ref[0], again[0]) and torch.equal(ref[1], again[1])))
