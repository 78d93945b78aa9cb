import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order for `-m gpu -x`: a tolerance test must never stand in front of the integer-exact,
# reference-pinned ones.  Rank 0: tracker / zone fixtures (bit-exact, pinned to the reference itself); 1: the integer parts
# of the detector (letterbox, NMS, decode switch); 2: one kernel family per test against the oracle, layer by layer
# (teacher-forced), the benchmarked shape first; 3: engine plumbing (batching, stages, chains, caches, pipeline); 4: the
# free-running fp16-vs-fp32 drift tests.  Rank 2 runs in the order listed, the other ranks in file order.
_GPU_RANK = [
    (0, ("test_gpu_tracker.py", "test_gpu_zones.py")),
    (1, ("test_letterbox_bit_exact", "test_nms_", "test_decode_fast_path", "test_letterbox_fused_into_stem", "test_front_end_fused")),
    (2, ("test_benchmarked_shape_parity", "test_persistent_tile_kernel", "test_persistent_tap_reuse_kernel", "test_forward_layers", "test_tap_reuse_conv_tiles", "test_fused_bottleneck_kernel",
         "test_eight_wave_tiles", "test_weight_stationary_1x1_tiles", "test_conv_with_fused_1x1_tail", "test_bottleneck_with_c2f_cv2_tail", "test_persistent_c2f32",
         "test_head_final_equals", "test_epilogue_variants", "test_neck_concat_read", "test_stem_and_layer1", "test_layer1_pixel_pair", "test_config5", "test_yolov8l_960", "test_converted_checkpoint")),
    (4, ("test_end_to_end_vs_fp32_oracle",)),
]


def _gpu_rank(item):
    for rank, pats in _GPU_RANK:
        for k, p in enumerate(pats):
            if p in item.nodeid:
                return (rank, k if rank == 2 else 0)
    return (3, 0)


def pytest_collection_modifyitems(config, items):
    gpu = [i for i, it in enumerate(items) if it.get_closest_marker("gpu") is not None]
    if not gpu:
        return
    ordered = sorted((items[i] for i in gpu), key=_gpu_rank)         # stable: file order inside a rank
    for slot, it in zip(gpu, ordered):
        items[slot] = it


@pytest.fixture(scope="session")
def pkg():
    import rtmodt_amd  # noqa: F401
    return sys.modules["rtmodt_amd"]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def unpack_frames(z):
    """Inverse of oracle/gen_golden_tracker.py:pack_inputs."""
    off = z["in_offsets"]
    return [(z["in_xyxy"][off[i]:off[i + 1]], z["in_conf"][off[i]:off[i + 1]], z["in_cls"][off[i]:off[i + 1]])
            for i in range(len(off) - 1)]
