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


def test_library_carries_the_digest_of_the_sources_it_was_built_from(pkg):
    """the stamp must describe the LIBRARY that runs, not the working tree -- the Makefile compiles tools/kernel_digest.py's value into
    build_info.o, which is rebuilt whenever any kernel source changes; after `make` the two agree, and bench.py reads the library's."""
    info = dict(kv.split("=", 1) for kv in pkg._ffi.lib().rtmodt_build_info().decode().split())
    assert info["diag"] == "0"
    assert info["csrc_sha256"] == csrc_digest(), "librtmodt_hip.so is stale: run make in csrc/ (or __graft_entry__.build())"


def test_workload_label_follows_the_arguments():
    """bench.py's config.workload named "config 4 shard" whatever --model / --size / --streams were."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert "config 4" in b.workload_label("s", 640, 8)
    assert "config 5" in b.workload_label("m", 1280, 1)
    assert "config 2/3" in b.workload_label("s", 640, 1)
    for other in (("s", 640, 16), ("m", 640, 8), ("n", 640, 8), ("s", 1280, 1)):
        assert b.workload_label(*other) == "not a BASELINE configuration"
    assert "120 ms" in b.live_stream_cost(4) and "no batching" in b.live_stream_cost(1)
