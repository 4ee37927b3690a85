"""Stamp a profile-derived file with what it was measured on: `python profiles/stamp.py <file> [...]` writes <file minus extension>.meta.json
with the git hash, a hash of the kernel sources (bench.kernel_sources_sha256) and bench.py's own hash.  bench.py refuses PMC figures whose
stamp does not match the kernel sources it runs on."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    try:
        git = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("GLF_GIT_HASH", "unknown")
    except OSError:
        git = os.environ.get("GLF_GIT_HASH", "unknown")
    meta = {"git_hash": git, "kernel_sources_sha256": bench.kernel_sources_sha256(),
            "bench_sha256": hashlib.sha256(open(os.path.join(ROOT, "bench.py"), "rb").read()).hexdigest()}
    for f in sys.argv[1:]:
        with open(os.path.splitext(f)[0] + ".meta.json", "w") as fh:
            json.dump(dict(meta, file=os.path.basename(f)), fh, indent=1)


if __name__ == "__main__":
    main()
