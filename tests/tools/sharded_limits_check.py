"""World-1 process group on one GPU: ShardedSearcher.search_batch_dev at limits on every merge route (one-wave kernel <= 512,
stable device sorts above, limit=None) equals the plain searcher.  Usage: python tests/tools/sharded_limits_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from conftest import load_golden
from test_lopq_hip_parity import hip_model
from columbiaimagesearch_amd.distributed import ShardedSearcher
from columbiaimagesearch_amd.lopq import LOPQSearcherHIP

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group(os.environ.get("BACKEND", "nccl"), rank=0, world_size=1)
z, X, Q = load_golden("c2")
m = hip_model(z)
single = LOPQSearcherHIP(m); single.add_codes_array(z["coarse"], z["fine"])
sh = ShardedSearcher(m); sh.add_codes_array(z["coarse"], z["fine"])
q = torch.as_tensor(Q[:32]).cuda().contiguous()
for quota, limit in [(3000, 100), (3000, 512), (5000, 513), (20000, 4000), (2500, None)]:
    a = single.search_batch_dev(q, quota=quota, limit=limit)
    b = sh.search_batch_dev(q, quota=quota, limit=limit)
    torch.cuda.synchronize()
    assert torch.equal(a["ids"], b["ids"]) and torch.equal(a["n_found"], b["n_found"]) and torch.equal(a["visited"], b["visited"])
    da, db = a["dists"], b["dists"]
    assert torch.equal(torch.isnan(da), torch.isnan(db)) and torch.equal(da[~torch.isnan(da)], db[~torch.isnan(db)])
    print("quota %d limit %s ok" % (quota, limit))
dist.destroy_process_group()
