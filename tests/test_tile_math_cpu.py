"""The integer arithmetic of the conv kernels that can be wrong without any kernel crashing -- division by launch constants
(every epilogue), the XCD-aware tile order, the persistent tile kernel's ownership of pixel tiles, the fused Bottleneck's LDS swizzle, nms_kernel's bit selection and bitonic schedule -- lives in
``csrc/tile_math.h`` as plain C++; this test compiles it with g++ and checks it exhaustively on the host (no GPU)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "real-time-multi-object-detection---tracking-system_amd", "csrc")


def test_tile_math_exhaustive(tmp_path):
    exe = str(tmp_path / "tile_math_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", CSRC, os.path.join(ROOT, "tests", "native", "tile_math_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    sys.stdout.write(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok ") == 8
