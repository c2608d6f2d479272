#!/usr/bin/env python3
"""sha256 over the product sources the round's profiles and bench lines measure (css_amd/**/*.py, the HIP / C++ sources and Makefile of
css_amd/csrc, bench.py), file names included, in sorted order.  scripts/collect_profiles.sh writes it beside the profiles it collects
(profiles/rNN_source_sha256.txt); tests/test_host_cpu.py::test_round_profiles_were_collected_from_these_sources recomputes it, so a product
change after the last profile collection turns the CPU suite red until the evidence is collected again.  Needs no git (the GPU box
has none)."""
import hashlib
import os
import sys

SUFFIXES = (".py", ".hip", ".h", ".cpp", ".c")


def product_files(root):
    out = [os.path.join(root, "bench.py")]
    for d, dirs, files in os.walk(os.path.join(root, "css_amd")):
        dirs[:] = sorted(x for x in dirs if x not in ("__pycache__", "build"))
        for f in sorted(files):
            if f.endswith(SUFFIXES) or f == "Makefile":
                out.append(os.path.join(d, f))
    return sorted(out)


def source_hash(root):
    h = hashlib.sha256()
    for p in product_files(root):
        h.update(os.path.relpath(p, root).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


if __name__ == "__main__":
    root = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    print(source_hash(root))
