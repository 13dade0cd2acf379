import torch, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import bench as B
dev = torch.device("cuda", 0)
model, z = B.load_model("c3full")
P = B.mixture_centers("relu_mixture", dev)
for n in (62500, 62500, 65536, 131072, 262144):
    x = B.gen_chunk(P, 0, n, dev)
    ts = []
    for i in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); co, fi = model.predict_batch_dev(x); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(n, ["%.2f" % t for t in ts], "%.2f M vec/s" % (n / min(ts) / 1e3))
