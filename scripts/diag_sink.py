import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from css_amd import ops
from css_amd.networks import resnet
from css_amd.networks.ddp_model import Model_mix
from css_amd.train_step import MixTrainer
hits = {"sink": 0, "miss": 0}
orig = ops._grad_sink
def spy(p, shp):
    r = orig(p, shp)
    hits["sink" if r is not None else "miss"] += 1
    return r
ops._grad_sink = spy
dev = torch.device("cuda:0")
cfg = {"Dataset": {"crop_size": (65, 65), "scale_size": (1.0, 1.0), "mix_mode": "none"}}
m = Model_mix(resnet.resnet101_tv(), num_classes=21, config=cfg, temp=0.5).to(dev).train().set_compute_dtype(torch.bfloat16)
tr = MixTrainer(m, 21, num_queries=32, num_negatives=64)
l, u = torch.randn(2, 3, 65, 65, device=dev), torch.randn(2, 3, 65, 65, device=dev)
lab = torch.randint(0, 21, (2, 65, 65), device=dev)
tr.step(l, lab, u)
torch.cuda.synchronize()
print(hits, "grad norm", float(tr.flat_g.norm()))
