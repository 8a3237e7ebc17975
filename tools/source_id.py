"""Identity of the kernel sources a measurement belongs to (bench.py, tools/pmc_traffic.py): sha256 over csrc/*.{hip,h} and
include/endo_hip.h in name order, plus the git commit when the tree has one (the GPU boxes get a snapshot without .git)."""
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha256():
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "endoscopydepthestimation-pytorch_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    files.append(os.path.join(ROOT, "include", "endo_hip.h"))
    for path in files:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def git_head():
    try:
        out = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10)
        return out.stdout.strip() if out.returncode == 0 and out.stdout.strip() else None
    except (OSError, subprocess.SubprocessError):
        return None


def source_id():
    return {"csrc_sha256": csrc_sha256(), "git_head": git_head()}


if __name__ == "__main__":
    import json
    print(json.dumps(source_id()))
