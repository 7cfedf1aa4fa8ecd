"""Hash of the kernel sources (popsift_amd/csrc/*.hip, *.h): tools/summarize_profiles.py writes it into the counter summary
it makes, bench.py compares it with the tree it runs from and marks quoted counters stale when they differ."""
import glob
import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def kernel_source_hash():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")) + glob.glob(os.path.join(HERE, "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_hash())
