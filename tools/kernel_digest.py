"""Digest of the kernel sources (csrc/*.hip, *.h, Makefile): what a measured per-kernel figure (HBM traffic, per-kernel time) belongs to.
`profiles/traffic_current.json` is stamped with it when the PMC passes are summarised (tools/summarize_profiles.py); bench.py prints
`roofline.traffic` only while the stamp still matches the tree it runs from -- a changed kernel with an old JSON yields null + the reason."""
import glob
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "real-time-multi-object-detection---tracking-system_amd", "csrc")


def csrc_digest(csrc: str = CSRC) -> str:
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "Makefile")))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def load_traffic(path: str, workload_key: str, digest: str):
    """-> (bytes_per_step | None, source-or-reason)"""
    if not os.path.exists(path):
        return None, "no profiles/traffic_current.json"
    tj = json.load(open(path))
    if tj.get("workload_key") != workload_key:
        return None, f"measured for workload {tj.get('workload_key')}, this run is {workload_key}"
    if tj.get("csrc_sha256") != digest:
        return None, ("stale: profiles/traffic_current.json was measured on kernel sources %s..., this tree is %s... -- re-run tools/collect_profiles.sh"
                      % (str(tj.get("csrc_sha256"))[:12], digest[:12]))
    return tj["hbm_bytes_per_step"], tj["source"]


if __name__ == "__main__":
    print(csrc_digest())
