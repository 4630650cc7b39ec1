"""Identity of the kernel sources a profile was taken with: sha256 over csrc/*.{hip,h,sh} + include/crowdnav.h (sorted by name),
first 16 hex digits.  tools/summarize_prof.py stamps profiles/rNN/{counters,traffic}.json with it and bench.py only quotes those
files when the stamp equals the tree's (a profile of another kernel says nothing about this one)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd", "csrc")


def csrc_hash():
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".sh")))
    files.append(os.path.join(ROOT, "include", "crowdnav.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_hash())
