"""bench.py's `roofline.traffic` is measured in separate rocprofv3 PMC passes and kept in profiles/traffic_current.json; the line may only carry it
while it still describes the kernels that ran: the JSON is stamped with a digest of csrc/, a changed kernel source yields null."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_digest import CSRC, csrc_digest, load_traffic      # noqa: E402


def test_traffic_is_null_when_the_kernels_changed(tmp_path):
    csrc = tmp_path / "csrc"
    shutil.copytree(CSRC, csrc, ignore=shutil.ignore_patterns("build", "*.o", ".pytest_cache"))
    d0 = csrc_digest(str(csrc))
    assert d0 == csrc_digest(str(csrc)) and len(d0) == 64
    tj = tmp_path / "traffic.json"
    json.dump({"workload_key": "s-640-8x4", "csrc_sha256": d0, "hbm_bytes_per_step": 123, "source": "pmc"}, open(tj, "w"))
    assert load_traffic(str(tj), "s-640-8x4", d0) == (123, "pmc")
    v, why = load_traffic(str(tj), "s-640-8x2", d0)
    assert v is None and "workload" in why
    with open(csrc / "conv.hip", "a") as f:                      # any change to a kernel source
        f.write("\n// touched\n")
    d1 = csrc_digest(str(csrc))
    assert d1 != d0
    v, why = load_traffic(str(tj), "s-640-8x4", d1)
    assert v is None and why.startswith("stale")
    v, why = load_traffic(str(tmp_path / "missing.json"), "s-640-8x4", d1)
    assert v is None


def test_committed_stamp_is_well_formed():
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_current.json")))
    assert "workload_key" in tj and "hbm_bytes_per_step" in tj
    v, why = load_traffic(os.path.join(ROOT, "profiles", "traffic_current.json"), tj["workload_key"], csrc_digest())
    assert (v is None and why.startswith("stale")) or v == tj["hbm_bytes_per_step"]
