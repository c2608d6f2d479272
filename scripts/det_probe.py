"""Loss trajectory of the ten-step run of tests/test_determinism_gpu.py in fp32 and bf16 (round 6: the numbers behind its restored learning assertion)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import test_determinism_gpu as T  # noqa: E402

for dt in (torch.float32, torch.bfloat16):
    a = T._run(dt, 10)
    print(dt, "total loss per step:", [round(float(x), 3) for x in a["hist"][:, 3]], "supervised:", [round(float(x), 3) for x in a["hist"][:, 0]], flush=True)
