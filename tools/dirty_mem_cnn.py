"""Does a DeepSentibank forward depend on what the workspace memory held before?  Fills most of HBM with NaN (or a constant), frees it,
then compares featurize(one image) with featurize_batch(two images)[0] and with a second run (bit for bit)."""
import io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from columbiaimagesearch_amd.featurizer.synthetic import sentibank_weights
from columbiaimagesearch_amd.featurizer import SentiBankNet
fill = float(sys.argv[1]) if len(sys.argv) > 1 else float("nan")
junk = [torch.full((8 << 30,), fill, dtype=torch.float32, device="cuda") for _ in range(7)]  # 224 GB
torch.cuda.synchronize()
del junk
torch.cuda.empty_cache()
net = SentiBankNet(sentibank_weights(0))
rs = np.random.RandomState(0)
x = torch.from_numpy((rs.randn(2, 3, 227, 227) * 50).astype(np.float32)).cuda()
a = net.forward_dev(x[:1].contiguous()).cpu().numpy()
b = net.forward_dev(x).cpu().numpy()
c = net.forward_dev(x[:1].contiguous()).cpu().numpy()
print("fill", fill, "single == batch[0]:", np.array_equal(a[0], b[0]), "single == single again:", np.array_equal(a, c),
      "finite:", np.isfinite(a).all(), np.isfinite(b).all(), "max diff", np.abs(a[0] - b[0]).max())
