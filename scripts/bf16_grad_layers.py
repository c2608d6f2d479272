"""Diagnostic (GPU box): per-parameter cosine of the bf16 gradients of tests/test_dist_gpu.py's experiment (4 images, 65x65) between one
rank and two ranks, and against fp32 - where along the backward chain do the runs part?  Usage: python scripts/bf16_grad_layers.py"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_dist_gpu as T  # noqa: E402
import torch  # noqa: E402

WORKER = T.WORKER.replace("probe = grads[", "torch.save({n: p.grad.detach().cpu() for n, p in net.named_parameters()} if world == 1 else None, sys.argv[1] + '.pt') if world == 1 else None\nprobe = grads[")
# world 2: per-parameter gradients after the all-reduce
WORKER = WORKER.replace("grads = torch.cat([p.grad.flatten() for p in net.parameters()])\nif world > 1:\n    dist.all_reduce(grads)\n    grads /= world",
                        "grads = torch.cat([p.grad.flatten() for p in net.parameters()])\nif world > 1:\n    dist.all_reduce(grads)\n    grads /= world\n    o = 0\n    d = {}\n    for n, p in net.named_parameters():\n        d[n] = grads[o:o + p.numel()].reshape(p.shape).cpu(); o += p.numel()\n    if rank == 0: torch.save(d, sys.argv[1] + '.pt')")


def run(world, bf16):
    out = tempfile.mktemp(suffix=".json")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29572")
        if bf16:
            env["CSS_TEST_BF16"] = "1"
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % ROOT, out], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    return torch.load(out + ".pt")


f32, b1, b2 = run(1, False), run(1, True), run(2, True)
f2 = run(2, False)
cos = lambda a, b: float(torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0))
print(f"{'parameter':58s} {'|g| fp32':>10s} {'b1~f32':>8s} {'b2~f32':>8s} {'b1~b2':>8s} {'f32w2~f32':>9s} {'|b2|/|b1|':>9s}")
names = list(f32)
worst = sorted(names, key=lambda n: -abs(float((b2[n].norm() + 1e-30) / (b1[n].norm() + 1e-30)) - 1.0))[:12]
print("largest |b2|/|b1| deviations:")
for n in worst:
    print(f"{n[:58]:58s} {float(f32[n].norm()):10.3e} {cos(b1[n], f32[n]):8.4f} {cos(b2[n], f32[n]):8.4f} {cos(b1[n], b2[n]):8.4f} {cos(f2[n], f32[n]):9.5f} "
          f"{float(b2[n].norm() / b1[n].norm()):9.4f}  |b1|/|f32| {float(b1[n].norm() / f32[n].norm()):.3f} |b2|/|f32| {float(b2[n].norm() / f32[n].norm()):.3f}")
tot = lambda d: float(torch.cat([d[n].flatten().double() for n in names]).norm())
print("whole-gradient norms: f32", tot(f32), "f32 world2", tot(f2), "bf16 world1", tot(b1), "bf16 world2", tot(b2))
print("sampled rows:")
for n in names[:12] + names[len(names) // 3: len(names) // 3 + 4] + names[-8:]:
    print(f"{n[:58]:58s} {float(f32[n].norm()):10.3e} {cos(b1[n], f32[n]):8.4f} {cos(b2[n], f32[n]):8.4f} {cos(b1[n], b2[n]):8.4f} {cos(f2[n], f32[n]):9.5f} "
          f"{float(b2[n].norm() / b1[n].norm()):9.4f}")
