"""Diagnostic: every parameter gradient of the HIP network vs the CPU oracle (same weights, same loss)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import css_oracle as O
from test_network_gpu import build
tag, backbone = sys.argv[1], sys.argv[2]
g = dict(np.load(f"tests/golden/{tag}.npz"))
K, seed, gain = int(g["K"]), int(g["seed"]), float(g["residual_gain"])
sd = O.init_state(backbone, K, 256, seed, gain)
names = O.param_names(backbone, K, 256)
for n in names: sd[n].requires_grad_(True)
x = torch.from_numpy(g["x"])
p, r = O.deeplab_forward(sd, x, backbone, True, K, 256)
((p * torch.from_numpy(g["wp"])).sum() + (r * torch.from_numpy(g["wr"])).sum()).backward()
net = build(backbone, K, seed, gain); net.train()
dev = torch.device("cuda:0")
pp, rr = net(x.to(dev))
((pp * torch.from_numpy(g["wp"]).to(dev)).sum() + (rr * torch.from_numpy(g["wr"]).to(dev)).sum()).backward()
named = dict(net.named_parameters())
rows = []
for n in names:
    a, b = named[n].grad.cpu().double(), sd[n].grad.double()
    rows.append(((a - b).abs().max().item() / (b.abs().max().item() + 1e-30), n, ((a - b).norm() / b.norm()).item()))
print("max-norm: max", max(rows)[:2], "median", sorted(rows)[len(rows)//2][0])
l2 = sorted((r[2], r[1]) for r in rows)
print("rel-L2: max", l2[-1], "median", l2[len(l2)//2][0], "p90", l2[int(len(l2)*0.9)][0])
tot_a = torch.cat([named[n].grad.cpu().double().flatten() for n in names]); tot_b = torch.cat([sd[n].grad.double().flatten() for n in names])
print("all-params rel-L2", ((tot_a - tot_b).norm() / tot_b.norm()).item(), "cosine", torch.nn.functional.cosine_similarity(tot_a, tot_b, dim=0).item())
