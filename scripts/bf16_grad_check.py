"""Diagnostic (GPU box): how far are the bf16 gradients of a 4-image 65x65 forward/backward from the fp32 ones, with the round-2 kernel
selection (CSS_NO_WS_CONV=1 CSS_BN_NO_MASK=1) and with the current one - the all-parameter probe of tests/test_dist_gpu.py is dominated
by the stem / layer1 gradients at the far end of a chaotic bf16 backward chain.  Usage: python scripts/bf16_grad_check.py"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_dist_gpu as T  # noqa: E402
import torch  # noqa: E402


def run(world, bf16, extra_env):
    out = tempfile.mktemp(suffix=".json")
    code = T.WORKER % ROOT
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", **extra_env)
        if bf16:
            env["CSS_TEST_BF16"] = "1"
        procs.append(subprocess.Popen([sys.executable, "-c", code, out], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    return json.load(open(out))


cos = lambda a, b: float(torch.nn.functional.cosine_similarity(torch.tensor(a), torch.tensor(b), dim=0))
ref = run(1, False, {})
old = {"CSS_NO_WS_CONV": "1", "CSS_BN_NO_MASK": "1"}
res = {}
for name, env in (("new", {}), ("old", old)):
    for world in (1, 2):
        res[(name, world)] = run(world, True, env)
        r = res[(name, world)]
        print(f"{name} kernels, world {world}: bf16 vs fp32 gradient cosine all {cos(r['grad'], ref['grad']):.4f} tail {cos(r['grad_tail'], ref['grad_tail']):.4f} "
              f"loss {r['loss']:.5f} (fp32 {ref['loss']:.5f})", flush=True)
for name in ("new", "old"):
    a, b = res[(name, 1)], res[(name, 2)]
    print(f"{name} kernels: world1 vs world2 cosine all {cos(a['grad'], b['grad']):.4f} tail {cos(a['grad_tail'], b['grad_tail']):.4f}")
print(f"world 1: new vs old kernels cosine all {cos(res[('new', 1)]['grad'], res[('old', 1)]['grad']):.4f} tail "
      f"{cos(res[('new', 1)]['grad_tail'], res[('old', 1)]['grad_tail']):.4f}")
