import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch
import test_determinism_gpu as T
for dt in (torch.float32, torch.bfloat16):
    a = T._run(dt, 10)
    print(dt, "total loss per step:", [round(float(x), 3) for x in a["hist"][:, 3]], "sup:", [round(float(x), 3) for x in a["hist"][:, 0]], flush=True)
