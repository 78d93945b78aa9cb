#!/usr/bin/env python3
"""bench.py -- frames/s of the detect + track hot path on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic video: for each of the
rank's S streams (default 8 = BASELINE config 4's per-GPU shard of 64 streams over 8 GPUs)
one 640x640x3 uint8 frame, already resident in HBM, goes through letterbox -> YOLOv8s
fp16 forward -> DFL decode -> per-class NMS -> ByteTrack update; the detections are copied
to the host every step (what Detector._parse does).  Streams are independent, so ranks
never exchange data (weak scaling, no collective on the data path); RCCL is used only for
the start/stop barrier and the max-over-ranks time.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import tempfile
import time

import numpy as np


class HwmonSampler:
    """Clock and package power of one GPU from its hwmon files (`freq1_input`, `power1_input`), read by a thread every 50 ms
    while the LONG window runs (never during the K timed steps that give `value`).  Evidence for where the staged bench sits
    against the chip's power limit (`power1_cap`); every field is None when the files are not there or not readable."""

    def __init__(self, device: int):
        self.dir = None
        self.samples = []
        self._stop = None
        try:
            import ctypes, glob
            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) != 0:
                return
            bdf = buf.value.decode().lower()
            cand = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
            if cand and os.path.exists(os.path.join(cand[0], "freq1_input")):
                self.dir = cand[0]
        except Exception:
            self.dir = None

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return int(f.read().strip())
        except Exception:
            return None

    def start(self):
        if self.dir is None:
            return
        import threading
        self._stop = threading.Event()

        def loop():
            while not self._stop.wait(0.05):
                self.samples.append((self._read("freq1_input"), self._read("power1_input")))
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self):
        if self._stop is None:
            return None
        self._stop.set()
        self._thread.join(timeout=1.0)
        f = [a for a, _ in self.samples if a]
        w = [b for _, b in self.samples if b]
        if not f or not w:
            return None
        cap = self._read("power1_cap")
        return {"sclk_mhz_mean": round(sum(f) / len(f) / 1e6, 0), "sclk_mhz_min": round(min(f) / 1e6, 0), "sclk_mhz_max": round(max(f) / 1e6, 0),
                "package_power_w_mean": round(sum(w) / len(w) / 1e6, 0), "package_power_w_max": round(max(w) / 1e6, 0),
                "power_cap_w": round(cap / 1e6, 0) if cap else None, "samples": len(f),
                "source": "hwmon freq1_input / power1_input every 50 ms during the long window (rank 0's GPU)"}

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense fp16/bf16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--streams", type=int, default=8, help="video streams batched per GPU")
    ap.add_argument("--frames-per-stream", type=int, default=4,
                    help="consecutive frames of every stream per launch (frame batching; the tracker still sees them one at a time, in order)")
    ap.add_argument("--model", default="s")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--ring", type=int, default=16, help="pre-generated frames per stream kept in HBM")
    ap.add_argument("--max-det", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--stages", type=int, default=0, help="stages of the staged engine (0: 3 for HBM-resident frames, 2 for host frames; 1: plain engine)")
    ap.add_argument("--depth", type=int, default=0, help="batches in flight after a submit (0: 2, or stages + 1 in the staged mode)")
    ap.add_argument("--no-compare", action="store_true", help="skip the extra one-frame-per-stream run (profiler passes: one workload per process)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL, default) or gloo (rehearsal on a 1-GPU box)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--host-frames", action="store_true", help="the MAIN measurement feeds frames from (page-locked) host memory (profiling runs; the default run reports this as the `host_frames` object beside the HBM-resident `value`)")
    ap.add_argument("--pageable", action="store_true", help="with --host-frames: ordinary pageable NumPy frames instead of the pinned ring")
    ap.add_argument("--host-mode", default="copy", choices=["copy", "mapped"],
                    help="host frames reach the device by DMA into a staging area (copy) or are read by the stem kernel straight from the mapped page-locked ring (mapped)")
    ap.add_argument("--frames-kind", default="noise", choices=["noise", "structured"],
                    help="synthetic frames: uniform-random bytes (default, the hardest case for the chip's power limit) or smooth blobs on a gradient")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the extra PCIe-inclusive measurement")
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed output self-check")
    ap.add_argument("--no-tracker-stress", action="store_true", help="skip the 200- / 500-box association sequences (BASELINE configs 3 / 5)")
    ap.add_argument("--prewarm", type=float, default=1.0, help="seconds of untimed full-pipeline running before the timed region (besides --warmup steps)")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="run ONLY the cpu_baseline leg (no GPU is touched) and print its object; tests/perf/cpu_baseline_node.py starts several of these side by side for a whole-box figure")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the cpu_baseline leg (0: min(os.cpu_count(), 16))")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendez-vous only: every rank joins the process group, rank 0 prints {n_gpus, ranks}, nobody touches a GPU (CPU test of the launcher path)")
    ap.add_argument("--long", type=float, default=1.0, help="seconds of the additional long steady-state window reported beside the K timed steps (0: off)")
    return ap.parse_args()


def cpu_baseline(pkg, weights, args, budget_s=20.0):
    """The reference's CPU path restated (kind "port"): torch fp32 CPU forward on all host
    cores (what detector.py runs when CUDA is absent: half is forced off, detector.py:76) +
    oracle decode/NMS + the C tracker restatement, timed on a bounded sample."""
    import torch
    from oracle import yolo_oracle as Y
    from oracle import tracker_oracle_c as TC
    cores = args.cpu_threads if getattr(args, "cpu_threads", 0) > 0 else min(os.cpu_count() or 1, 16)          # a 1-GPU box's CPU share; more threads only oversubscribe these small convs
    torch.set_num_threads(cores)
    frames = pkg.synth.frames(4, args.size, args.size, seed=1234)
    trk = TC.TrackerOracleC()
    times = []
    t_start = time.perf_counter()
    i = 0
    with torch.no_grad():
        while True:
            f = frames[i % len(frames)]
            t0 = time.perf_counter()
            x = Y.preprocess(f, args.size, args.size)
            xt = torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))[None]
            heads = pkg.weights.torch_forward(xt, weights, args.model)
            maps = [h[0].permute(1, 2, 0).numpy() for h in heads]
            pred = Y.decode(maps)
            dets, _ = Y.non_max_suppression(pred, 0.35, 0.45, None, False, args.max_det)
            xyxy = Y.scale_boxes(dets[:, :4], args.size, args.size, args.size, args.size) if len(dets) else np.empty((0, 4), np.float32)
            trk.update(xyxy, dets[:, 4], dets[:, 5].astype(np.int32))
            times.append(time.perf_counter() - t0)
            i += 1
            if i >= 3 and (time.perf_counter() - t_start > budget_s or i >= 64):
                break
    t = np.asarray(times[1:])                                  # first frame warms torch's thread pool
    return {"value": round(float(1.0 / t.mean()), 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "host_threads_available": os.cpu_count(),
            "cores_note": "threads actually used = min(os.cpu_count(), 16): a 1-GPU share of the box; these small convs do not scale past that "
                          "(more torch threads only oversubscribe them), so the box's remaining hardware threads are left idle on purpose",
            "sample": f"{len(t)} frames 640x640, YOLOv8{args.model} fp32 torch-CPU forward + NumPy decode/NMS + C tracker, "
                      f"p50 {float(np.median(t)) * 1e3:.1f} ms/frame",
            "p50_ms": round(float(np.median(t)) * 1e3, 2)}


def workload_label(model: str, size: int, streams: int) -> str:
    """Which BASELINE.json configuration (if any) the per-GPU workload is the shard of -- derived from the arguments, never assumed
    (a --model m --size 1280 --streams 1 run used to label itself "config 4 shard")."""
    if (model, size, streams) == ("s", 640, 8):
        return "BASELINE config 4 shard: 64 streams over 8 GPUs = 8 per GPU"
    if (model, size, streams) == ("m", 1280, 1):
        return "BASELINE config 5 shard: 8 streams over 8 GPUs = 1 per GPU"
    if (model, size, streams) == ("s", 640, 1):
        return "BASELINE config 2/3 shape: one stream on one GPU"
    return "not a BASELINE configuration"


def live_stream_cost(F: int, fps_camera: float = 25.0) -> str:
    """What batching F consecutive frames of a stream into one launch set costs a LIVE camera (the headline assumes frames that are already there)."""
    if F <= 1:
        return "no batching over time: every frame is submitted as it arrives"
    return (f"a live {fps_camera:g} fps camera would wait {(F - 1) * 1e3 / fps_camera:.0f} ms to fill a launch set (F - 1 frame periods) -- "
            f"offline / recorded streams pay nothing; value_no_temporal_batching is the F = 1 figure")


def tracker_stress(pkg, dev):
    """BASELINE configs 3 and 5's association load as part of the driver-run record: the 200-box @ 640 (120 frames) and
    500-box @ 1280 (60 frames) sequences of SURVEY 8d through the single-launch tracker, frame by frame, every frame's state digest compared
    with the fixture the REFERENCE's own tracker.py generated (tests/golden/tracker_g5_seq200.npz, tracker_g6_seq500.npz; the digest
    function is the oracle's -- the checker, as in verify_outputs).  Time = wall clock around the synchronous C-ABI call (H2D of the frame's
    detections + one launch + device sync); pairs = live tracks x detections offered to the IoU sweep."""
    import hashlib
    from importlib import import_module
    from oracle import tracker_oracle as T
    core_cls = import_module(pkg.__name__ + ".tracking.tracker")._ByteTrackCore
    out = {}
    for key, fname in (("config3_200_boxes_640", "tracker_g5_seq200.npz"), ("config5_500_boxes_1280", "tracker_g6_seq500.npz")):
        z = np.load(os.path.join(ROOT, "tests", "golden", fname))
        n, canvas, frames, seed = int(z["seq_n"]), int(z["seq_canvas"]), int(z["seq_frames"]), int(z["seq_seed"])
        xy, cf, cl = pkg.synth.box_sequence(n, canvas, frames, seed)
        sha = hashlib.sha256(xy.tobytes() + cf.tobytes() + cl.tobytes()).digest()
        if not np.array_equal(np.frombuffer(sha, dtype=np.uint8), z["in_sha"]):
            raise SystemExit(f"bench tracker_stress: the generated {key} sequence is not the fixture's")
        core = core_cls(device=dev, n_streams=1, max_dets=max(512, n), max_tracks=2048)
        ts, pairs = [], 0
        for rep in range(3):
            core.reset()
            for f in range(frames):
                live = len(core.snapshot(0)["ids"]) if rep == 0 else 0
                t0 = time.perf_counter()
                core.update(xy[f], cf, cl)
                ts.append(time.perf_counter() - t0)
                if rep == 0:
                    pairs += live * n
                    if not np.array_equal(T.state_digest(core.snapshot(0)), z["digest"][f]):
                        raise SystemExit(f"bench tracker_stress FAILED: {key} frame {f}: tracker state differs from the reference-generated fixture")
        p50 = float(np.median(ts))
        out[key] = {"boxes_per_frame": n, "canvas": canvas, "frames": frames, "update_us_p50": round(p50 * 1e6, 1), "update_us_p95": round(float(np.percentile(ts, 95)) * 1e6, 1),
                    "iou_pairs_per_frame_mean": int(pairs / frames), "iou_pairs_per_s": round(pairs / frames / p50, 0),
                    "live_tracks_end": len(core.snapshot(0)["ids"]), "bit_exact_vs_reference_fixture": True, "fixture": "tests/golden/" + fname}
        core.close()
    out["what"] = ("one stream, one launch per frame, greedy assignment (the reference's executable branch); wall clock around the synchronous C-ABI update "
                   "(H2D of the detections + launch + sync); every frame's state digest == the fixture generated by the reference's tracker.py")
    return out


def verify_outputs(pkg, det, frame_ptrs, size, max_det, weights, scale):
    """One more batch through the benchmarked detector (same engine, tiles, stages and streams), after the timed region:
      (1) for every image the fetched detections must equal oracle NMS + scale_boxes applied to the engine's own pre-NMS
          tensor, bit for bit;
      (2) the convs that were just timed: every stored layer of the first and the last image of the batch against the fp32
          oracle fed the engine's own fp16 inputs (teacher forcing), tolerance 2e-3 * max|ref| + 2e-3 as in tests/ (also where
          the tuner keeps an fp16 intermediate in LDS: fused Bottlenecks, conv -> 1x1 pairs, the fused front end) -- a wrong tile in
          conv_mfma64_pt, conv3x3_rows_grp, bottleneck_fused or a tail kernel cannot hide behind (1);
      (3) decode on the engine's own head logits, rtol = atol = 2e-4.
    The oracle is the checker; any mismatch aborts the bench."""
    from oracle import yolo_oracle as Y
    det.enqueue(frame_ptrs, height=size, width=size)
    got = det.fetch()
    n_img = n_box = 0
    for i in range(len(got)):
        _, _, pred = det.debug_fetch(i, want_input=False, want_heads=False)
        dets, _ = Y.non_max_suppression(pred, det.confidence, det.iou, det.classes, det.agnostic_nms, max_det)
        ref = Y.scale_boxes(dets[:, :4], size, size, size, size) if len(dets) else np.empty((0, 4), np.float32)
        d = got[i]
        if len(d) != len(dets) or not np.array_equal(d.xyxy.view(np.int32), ref.view(np.int32)) or \
                not np.array_equal(d.confidence.view(np.int32), dets[:, 4].astype(np.float32).view(np.int32)) or \
                d.class_id.tolist() != dets[:, 5].astype(np.int32).tolist():
            raise SystemExit(f"bench self-check FAILED on image {i}: engine detections differ from oracle NMS on the engine's own pre-NMS tensor")
        n_img += 1
        n_box += len(d)
    # (2) + (3)
    names = [c.name for c in pkg.weights.spec(scale)]
    fused_behind = {"2.cv1", "4.cv1", "6.cv1", "8.cv1", "2.cv2", "15.cv2"}      # consumers of an LDS-resident fp16 intermediate when the tuner fused the pair
    worst, worst_fused, checked = ("", 0.0), ("", 0.0), 0
    images = sorted({0, len(got) - 1})
    for img in images:
        inp, heads, pred = det.debug_fetch(img)
        gpu = {}
        for n in names:
            try:
                gpu[n] = det.debug_layer(n, img).astype(np.float32)
            except pkg._ffi.RtmodtError as e:                  # fused away: lives in LDS only (the consumer is checked instead)
                if e.code != pkg._ffi.E_UNSUPPORTED:
                    raise
        taps = {}
        Y.forward(inp.astype(np.float32), weights, scale, taps=taps, force=gpu)
        for n, g in gpu.items():
            tol = 2e-3 * float(np.abs(taps[n]).max()) + 2e-3      # ONE class since round 5 (the 4e-3 allowance behind fused pairs measured 0.23 of 2e-3)
            err = float(np.abs(taps[n] - g).max())
            if not np.isfinite(err) or err > tol:
                raise SystemExit(f"bench self-check FAILED: image {img} layer {n}: max err {err:.4g} > tol {tol:.4g} vs the fp32 oracle (teacher-forced)")
            if err / tol > worst[1]:
                worst = (n, err / tol)
            if (n in fused_behind or (".m." in n and n.endswith(".cv2"))) and err / tol > worst_fused[1]:
                worst_fused = (n, err / tol)
            checked += 1
        A, maps, off = det.model.n_anchors, [], 0
        for s_ in (size // 8, size // 16, size // 32):
            maps.append(heads[off:off + s_ * s_ * 144].reshape(s_, s_, 144).astype(np.float32)); off += s_ * s_ * 144
        if not np.allclose(pred, Y.decode(maps), rtol=2e-4, atol=2e-4):
            raise SystemExit(f"bench self-check FAILED: image {img}: decode differs from the oracle on the engine's own head logits")
    launches = [n for n, _, _ in det.profile(1)]
    fam = {"conv3x3_pp": sum("pp:" in n and "ppt:" not in n for n in launches) + sum(n.count("pp:") - 1 for n in launches if n.count("pp:") > 1 and "ppt:" not in n),
           "conv_tile_pp": sum("ppt:" in n for n in launches), "conv_mfma64_pt": sum("tile pt:" in n or ", pt:" in n or "[pt:" in n for n in launches),
           "conv3x3_rows": sum("rows" in n for n in launches), "bottleneck_fused": sum("fused bottleneck" in n for n in launches),
           "conv_mfma_tail": sum("tail:" in n for n in launches), "8-wave tiles": sum("/8w" in n for n in launches), "head_final": sum("head_final" in n for n in launches)}
    return {"ok": True, "images": n_img, "boxes": n_box,
            "what": "fetched detections == oracle non_max_suppression + scale_boxes on the engine's pre-NMS tensor (bit-exact), every image of one batch",
            "layers_ok": True, "layers_checked": checked, "layer_images": images,
            "layers_what": "every stored conv output of these images vs the fp32 oracle fed the engine's own inputs, |err| <= 2e-3 * max|ref| + 2e-3 (one class: also behind "
                           "an LDS-resident fp16 intermediate); decode on the engine's head logits rtol = atol = 2e-4",
            "worst_layer": {"name": worst[0], "err_over_tol": round(worst[1], 3)},
            "worst_layer_behind_a_fused_pair": {"name": worst_fused[0], "err_over_tol": round(worst_fused[1], 3)}, "launches_by_kernel_family": fam}


def launch_ranks(args) -> int:
    """`bench.py --gpus N` started WITHOUT a launcher (WORLD_SIZE unset): start the N ranks ourselves, as a CHILD
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>`,
    before this process has imported the package or touched a GPU (a process that has initialised the GPU must never
    exec or be replaced; a plain child is safe).  The child's rank 0 prints the JSON line; it is passed through once,
    after checking that it really reports N GPUs.  Returns the exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in child.stdout:
        if out.startswith("{") and '"n_gpus"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)                          # anything else a rank printed: keep stdout to the one JSON line
    rc = child.wait()
    if rc != 0:
        print(f"bench.py: the {args.gpus}-rank child run failed with exit code {rc}", file=sys.stderr)
        return rc
    if line is None:
        print("bench.py: the ranks finished without printing a result line", file=sys.stderr)
        return 1
    got = json.loads(line).get("n_gpus")
    if got != args.gpus:
        print(f"bench.py: asked for --gpus {args.gpus} but the run reports n_gpus = {got}; refusing to print it", file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    import rtmodt_amd  # noqa: F401
    pkg = sys.modules["rtmodt_amd"]
    if args.cpu_baseline_only:                           # the reference's CPU path restated, alone (no GPU, no process group)
        weights = pkg.weights.synthetic(args.model, input_size=args.size)
        print(json.dumps(cpu_baseline(pkg, weights, args)), flush=True)
        return
    sync = pkg.streams.NodeSync(backend=args.backend)    # RCCL; no process group when WORLD_SIZE == 1
    rank, local_rank, world = sync.rank, sync.local_rank, sync.world
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would not describe the run")
    if args.launch_check:
        ranks = sync.sum_stats([1])[0]
        # the per-rank plumbing of a real run, without a GPU: the single-writer weight file (temporary name + rename, read by everybody
        # after the barrier), one tune-cache file per rank, the per-rank frames/s gather
        wcheck = os.path.join(tempfile.gettempdir(), f"rtmodt_launch_check_{os.environ.get('MASTER_PORT', '0')}.bin")
        if rank == 0:
            tmp = wcheck + f".{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(b"RTMODTW1" + bytes(range(256)) * 64)
            os.replace(tmp, wcheck)
        sync.barrier()
        with open(wcheck, "rb") as f:
            whole = len(f.read()) == 8 + 256 * 64
        tune = os.path.join(tempfile.gettempdir(), f"rtmodt_bench_tune_{os.getpid()}.txt")
        per_rank = sync.gather_floats(float(rank))
        seen_whole = sync.sum_stats([1 if whole else 0])[0]
        sync.barrier()
        if rank == 0:
            os.unlink(wcheck)
            print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_joined": ranks, "backend": args.backend, "per_rank_frames_s": per_rank,
                              "ranks_that_read_the_whole_weight_file": seen_whole, "tune_cache_of_rank_0": tune}), flush=True)
        sync.close()
        return
    if not args.one_device:
        ndev = pkg._ffi.device_count()
        if local_rank >= ndev:
            raise SystemExit(f"rank {rank}: --gpus {args.gpus} needs GPU {local_rank} but only {ndev} are visible (rehearsal on one card: --one-device --backend gloo)")

    from importlib import import_module
    core_cls = import_module(pkg.__name__ + ".tracking.tracker")._ByteTrackCore
    dev = 0 if args.one_device else local_rank
    S, R, size = args.streams, args.ring, args.size

    # ---- synthetic weights (seeded; same file on every rank) ----
    wpath = os.path.join(tempfile.gettempdir(), f"rtmodt_bench_yolov8{args.model}_{size}.rtw")
    weights = None
    if rank == 0:                                       # one writer (temporary name + atomic rename); the others read after the barrier
        weights = pkg.weights.synthetic(args.model, input_size=size)
        tmp = wpath + f".{os.getpid()}"
        pkg.weights.save(tmp, weights, args.model)
        os.replace(tmp, wpath)
    sync.barrier()
    # the tuner's decisions are shared by every detector this process builds for one shape (main run, host-frame leg,
    # self-check): one file per rank
    os.environ.setdefault("RTMODT_TUNE_CACHE", os.path.join(tempfile.gettempdir(), f"rtmodt_bench_tune_{os.getpid()}.txt"))

    # ---- frames resident in HBM: this rank owns the global streams {id : id % world == rank}, seed 1234 + id ----
    per = size * size * 3
    ring = pkg._ffi.DeviceBuffer(S * R * per, dev)
    my_streams = pkg.streams.shard(S * world, world, rank)
    gen = pkg.synth.frames if args.frames_kind == "noise" else pkg.synth.structured_frames
    for s, gid in enumerate(my_streams):
        ring.upload(gen(R, size, size, seed=1234 + gid), offset=s * R * per)

    F = max(1, args.frames_per_stream)

    ptrs = [[ring.ptr + (s * R + r) * per for s in range(S)] for r in range(R)]

    # the same frames in a page-locked host ring: slot (r, s) = frame r of stream s, slots of one r -- and of
    # consecutive r -- back to back, so the S x F frames of a step are ONE contiguous block (one DMA)
    host_ring = host_ptrs = pinned = None

    def make_host_ring():
        nonlocal host_ring, host_ptrs, pinned            # `pinned` owns the page-locked memory: it must outlive every view
        if host_ring is not None:
            return
        if args.pageable:
            host_ring = [[pkg.synth.frames(R, size, size, seed=1234 + gid)[r].copy() for gid in my_streams] for r in range(R)]
            return
        pinned = pkg.pipeline.PinnedFrameRing(R * S, size, size, device=dev)
        for s, gid in enumerate(my_streams):
            fr = pkg.synth.frames(R, size, size, seed=1234 + gid)
            for r in range(R):
                pinned.write(r * S + s, fr[r])
        host_ring = [[pinned.frame(r * S + s) for s in range(S)] for r in range(R)]
        host_ptrs = [[int(host_ring[r][s].ctypes.data) for s in range(S)] for r in range(R)]

    def sync_all(det):
        det.synchronize()                           # hipDeviceSynchronize through the C ABI
        sync.device_synchronize()                   # + torch.cuda.synchronize() when torch.distributed is up

    def measure(F, steps, warmup, collective=True, prewarm_s=None, long_s=0.0, host=False):
        """K timed steps; a step = one launch set over S streams x F consecutive frames (image f * S + s).

        Protocol: W warm-up steps, then an untimed time-based pre-warm of the full pipeline (>= 1 s:
        clocks and caches as in a long-running service, whatever W is), barrier + device sync, then
          * cold:   K batches submitted into the EMPTY pipeline and retired, device sync at the end (fill + drain inside);
          * steady: the pipeline is refilled (S + 1 batches in flight), the clock starts right after a fetch() and stops
                    right after the K-th next fetch(): exactly K batches retired, S + 1 in flight throughout -- `value`.
        Both brackets are closed by a barrier + device sync of every rank."""
        if host:
            make_host_ring()
        mapped = host and args.host_mode == "mapped" and not args.pageable
        # engine: the staged one.  Frames in HBM: three stages (main + two more stage streams + post-processing = the four hardware
        # queues of the process).  Frames from the host: two stages, so that the copy stream keeps a queue of its own and the
        # upload of batch t + 1 runs beside every stage of batch t (measured: 98 % of the HBM-resident rate, against 90 % with
        # three stages and the upload in front of stage 1 on the main stream; profiles/r02/README.md)
        stages = args.stages if args.stages > 0 else (2 if host else 3)
        det = pkg.Detector(wpath, input_size=(size, size), max_det=args.max_det, device=f"cuda:{dev}", batch=S * F,
                           use_graph=not args.no_graph, warmup=False, chains=1 - stages if stages > 1 else 1)
        trk = core_cls(device=dev, n_streams=S, max_dets=max(128, args.max_det), max_tracks=2048)

        def submit(t):
            if mapped:                                  # the stem reads the page-locked frames over PCIe itself
                det.enqueue([pt for f in range(F) for pt in host_ptrs[(t * F + f) % R]], height=size, width=size)
            elif host:                                  # H2D of S x F x 1.23 MB inside the step
                det.enqueue([fr for f in range(F) for fr in host_ring[(t * F + f) % R]])
            else:
                det.enqueue([pt for f in range(F) for pt in ptrs[(t * F + f) % R]], height=size, width=size)
            # ONE tracker launch per step: stream s's workgroup walks over its F frames (slots f * S + s) in order -- tracker.py:58-141
            # still sees every stream one frame at a time (tests: == F per-frame launches == the oracle)
            trk.update_from_detector(det, 0, S, frames_per_stream=F)

        def step(t):
            """Steady state of the pipeline: submit batch t, then collect the oldest batch in flight (whose
            copy to the host overlapped the GPU's work on the newer ones)."""
            submit(t)
            return det.fetch()

        # a generation-2 Python GC pass walks every object torch's import created (~40 ms): keep the
        # collector out of the loop (the per-step garbage is a few hundred short-lived objects)
        gc.collect()
        gc.freeze()
        gc.disable()
        n_st = getattr(det.model, "stages", 1)
        depth = args.depth if args.depth > 0 else (n_st + 1 if n_st > 1 else 2)     # batches in flight after a submit
        pc = time.perf_counter
        t = 0

        def fill():
            nonlocal t
            for _ in range(depth - 1):
                submit(t); t += 1

        def drain():
            for _ in range(depth - 1):
                det.fetch()

        fill()
        for _ in range(warmup):
            step(t); t += 1
        t_end = pc() + (args.prewarm if prewarm_s is None else prewarm_s)
        while pc() < t_end:
            step(t); t += 1
        drain()
        sync_all(det)
        if collective:
            sync.barrier()
            sync_all(det)
        # ---- cold: K batches through an empty pipeline, sync-bracketed ----
        t0 = pc()
        fill()
        for _ in range(max(steps - (depth - 1), 0)):
            step(t); t += 1
        drain()
        sync_all(det)
        cold = pc() - t0
        if collective:
            sync.barrier()
        # ---- steady: exactly K batches retired with the pipeline full on both sides of the clock ----
        fill()
        for _ in range(depth):
            step(t); t += 1
        fwd_ms = tot_ms = 0.0
        n_det = 0
        stamps = [pc()]
        for _ in range(steps):
            out = step(t); t += 1                       # submits one batch, retires one batch: K batches per K steps
            stamps.append(pc())
            a, b = det.last_timing()                    # HIP events on the detector's streams, already complete
            tot_ms += a
            fwd_ms += b
            n_det += sum(len(d) for d in out)
        elapsed = stamps[-1] - stamps[0]
        # ---- the same steady state over a longer window (>= long_s seconds), for the record ----
        long_run = None
        if long_s > 0:
            hw = HwmonSampler(dev) if rank == 0 else None
            if hw:
                hw.start()
            k = 0
            tl0 = pc()
            while pc() - tl0 < long_s or k * S * F < 1000:
                step(t); t += 1; k += 1
            long_run = {"steps": k, "seconds": pc() - tl0, "power_clock": hw.stop() if hw else None, "sampler_on": False}
            if rank == 0:
                # the in-kernel clock is read in a SEPARATE short window behind the long one: the sampler puts a ~20 us single-wave kernel
                # behind every batch's NMS, which would delay that batch's retire -- the long window above runs without it, on every rank alike
                det.clock_sampling(True)                 # one wave behind every batch's NMS reads s_memtime against s_memrealtime
                kc = 0
                tc0 = pc()
                while pc() - tc0 < min(0.35, long_s) or kc < 64:
                    step(t); t += 1; kc += 1
                drain_now = [det.fetch() for _ in range(depth - 1)]      # the sampler's launches sit behind these batches' NMS
                del drain_now
                ghz_mean, ghz_min, ghz_max, n_s = det.clock_read()
                det.clock_sampling(False)
                long_run["in_kernel_clock"] = {"ghz_mean": round(ghz_mean, 4), "ghz_min": round(ghz_min, 4), "ghz_max": round(ghz_max, 4), "samples": n_s,
                                               "window": f"{kc} steps right behind the long window, sampler on"}
                fill()
        drain()                                          # the batches still in flight (outside the timed region)
        sync_all(det)
        if collective:
            sync.barrier()
            sync_all(det)
        gc.enable()
        per = np.diff(np.asarray(stamps)) * 1e3
        return {"elapsed": elapsed, "cold": cold, "fwd_ms": fwd_ms, "tot_ms": tot_ms, "n_det": n_det, "det": det, "trk": trk,
                "step_ms": per, "long": long_run, "stages": n_st, "depth": depth}

    if args.steps < 4:
        raise SystemExit("--steps must be at least 4 (the staged engine keeps 4 batches in flight)")
    m = measure(F, args.steps, args.warmup, long_s=args.long, host=args.host_frames)
    det, trk, elapsed, fwd_ms, tot_ms, n_det = m["det"], m["trk"], m["elapsed"], m["fwd_ms"], m["tot_ms"], m["n_det"]
    flops_step = det.model.conv_flops_per_frame * S * F
    per_rank_fps = sync.gather_floats(S * F * args.steps / elapsed)      # every rank's own frames/s (stragglers show in a SCALE run)
    elapsed = sync.max_time(elapsed)                # MAX over ranks
    cold = sync.max_time(m["cold"])
    n_tracks = sum(len(trk.snapshot(s)["ids"]) for s in range(S))
    # optional stats reduce: ~24 bytes over RCCL, once per run
    n_det, n_tracks_node = sync.sum_stats([n_det, n_tracks])

    if rank != 0:
        sync.barrier()                               # wait for rank 0's extra (untimed) measurements, then leave together
        sync.close()
        return

    frames_total = world * S * F * args.steps
    fps = frames_total / elapsed
    fwd_ms_step = fwd_ms / args.steps
    chains = getattr(det.model, "chains", 1)
    # the stages of consecutive batches overlap: the event span of one batch (first launch -> decoded) then exceeds the
    # step period, which is the forward pass's share of the device in a steady state
    stages = getattr(det.model, "stages", 1)
    fwd_ms_step = min(fwd_ms_step, elapsed / args.steps * 1e3) if chains > 1 or stages > 1 else fwd_ms_step
    achieved = flops_step / (fwd_ms_step * 1e-3) / 1e12
    src = (("pageable" if args.pageable else "page-locked") + f" host memory ({args.host_mode}; PCIe-inclusive)") if args.host_frames else "HBM-resident ring"
    res = {
        "metric": "frames/sec whole-node, YOLOv8s 640x640 fp16 detect + ByteTrack",
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"YOLOv8{args.model} {size}x{size} fp16, {S} synthetic stream(s) per GPU ({workload_label(args.model, size, S)}), "
                               f"{F} consecutive frame(s) of every stream per launch set ({live_stream_cost(F)}), "
                               f"detect (letterbox+forward+decode+NMS, max_det {args.max_det}) + ByteTrack update (frame by frame, in order), "
                               f"frames from: {src}, detections copied to host every step",
                   "baseline_config": workload_label(args.model, size, S),
                   "streams_per_gpu": S, "frames_per_stream_per_step": F, "frames_per_step": S * F * world,
                   "weights": "synthetic seed 0, LSUV-calibrated on noise frames",
                   "parallelism": f"streams sharded {world} ways, no data-path collective"},
        "timing": {"protocol": f"{args.warmup} warm-up steps + {args.prewarm:g} s untimed pre-warm, barrier + device sync, then exactly {args.steps} batches "
                               f"retired with {m['depth']} in flight on both sides of the clock (host clock stamped after fetch()); max over ranks",
                   "cold_ms_per_step": round(cold / args.steps * 1e3, 4),
                   "cold_note": "the same K batches through an EMPTY pipeline, device-sync bracketed (pipeline fill and drain inside the bracket)",
                   "step_ms_p50": round(float(np.median(m["step_ms"])), 4), "step_ms_max": round(float(m["step_ms"].max()), 4),
                   "step_ms_min": round(float(m["step_ms"].min()), 4)},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / MFMA_F16_DENSE_PEAK_TFLOPS, 4), "traffic": None,
                     "kernel": "forward pass = conv_mfma<*> launches (+ stem, SPPF pool, decode); " +
                               (f"{chains} sub-batch chains of {S * F // chains} frames on their own streams, one hipGraph each" if chains > 1 else
                                f"{stages} stages of the net on {stages} streams, consecutive batches overlapped, one hipGraph per stage" if stages > 1 else "one hipGraph"),
                     "chains": chains, "stages": stages,
                     "flops_per_step": int(flops_step), "forward_ms_per_step": round(fwd_ms_step, 4),
                     "batch_latency_ms": round(tot_ms / args.steps, 4),
                     "batch_latency_note": "HIP events, first launch of a batch -> its NMS done (pipeline latency of one batch, not a per-step time)"},
        "per_rank_frames_s": [round(v, 1) for v in per_rank_fps],
        "detections_per_frame": round(n_det / frames_total, 2), "live_tracks_node": n_tracks_node,
        "frames_source": src,
    }
    if m["long"]:
        lr = m["long"]
        res["timing"]["long_window"] = {"value": round(world * S * F * lr["steps"] / lr["seconds"], 1), "unit": "frames/s (rank 0's own clock)",
                                        "steps": lr["steps"], "seconds": round(lr["seconds"], 3), "power_clock": lr.get("power_clock")}
        ikc = lr.get("in_kernel_clock")
        if ikc and ikc["samples"] > 0 and ikc["ghz_mean"] > 0:
            # informational: `peak` / `frac` stay priced at the data sheet's 2.4 GHz.  The shader clock the kernels actually ran at, read INSIDE
            # the device (s_memtime ticks per 100-MHz s_memrealtime tick, one wave behind every batch's NMS during the long window): the chip
            # lowers it under this load, and hwmon / DPM readings overstate it (MI355X_MICROARCH.md, DVFS give-back)
            held = 2500.0 * ikc["ghz_mean"] / 2.4
            res["roofline"]["in_kernel_clock"] = dict(ikc, peak_at_clock=round(held, 1), frac_at_clock=round(res["roofline"]["achieved"] / held, 4),
                                                      source="s_memtime / s_memrealtime x 100 MHz, ~20 us samples on the post-processing stream during the long window")
        pcw = lr.get("power_clock")
        if pcw and isinstance(res.get("roofline"), dict) and res["roofline"].get("achieved"):
            # informational: `peak` / `frac` stay priced at 2.4 GHz; this is the same peak at the clock the chip held under its power limit
            held = 2500.0 * pcw["sclk_mhz_mean"] / 2400.0
            res["roofline"]["at_held_clock"] = {"sclk_mhz": pcw["sclk_mhz_mean"], "peak": round(held, 1), "frac": round(res["roofline"]["achieved"] / held, 4),
                                                "package_power_w": pcw["package_power_w_mean"], "power_cap_w": pcw["power_cap_w"]}

    if not args.host_frames:
        # HBM bytes per step come from separate rocprofv3 PMC passes (tools/collect_profiles.sh), so this line can only carry them while they still
        # describe the kernels it ran: the JSON is stamped with a digest of csrc/, and a tree whose kernels changed since prints null and says why
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from kernel_digest import load_traffic
        # ... of the LIBRARY that ran (compiled in by the Makefile: rtmodt_build_info), not of the working tree -- a stale or diagnostic .so never inherits the figure
        binfo = dict(kv.split("=", 1) for kv in pkg._ffi.lib().rtmodt_build_info().decode().split())
        if binfo.get("diag") != "0":
            traffic, why = None, "this library is a diagnostic build (make DIAG=1): measured figures are not attached to it"
        else:
            traffic, why = load_traffic(os.path.join(ROOT, "profiles", "traffic_current.json"), f"{args.model}-{size}-{S}x{F}", binfo.get("csrc_sha256", "unknown"))
        res["roofline"]["library_build"] = binfo
        res["roofline"]["traffic"] = traffic
        res["roofline"]["traffic_unit"] = "bytes per step (FETCH_SIZE x2 + WRITE_SIZE over the forward-pass launches)"
        res["roofline"]["traffic_source"] = why

    # ---- untimed self-check on one more batch of the SAME detector: NMS survivors, boxes, scores and classes of every image
    # must equal the oracle's non_max_suppression + scale_boxes on the engine's own pre-NMS tensor (the oracle is the checker) ----
    if not args.no_verify:
        res["verified"] = verify_outputs(pkg, det, [pt for f in range(F) for pt in ptrs[f % R]], size, args.max_det,
                                         weights if weights is not None else pkg.weights.load(wpath)[0], args.model)

    # ---- per-kernel view (eager, HIP events around every launch) ----
    prof = det.profile(3)
    conv_ms = sum(ms for name, ms, fl in prof if fl > 0)
    res["roofline"]["conv_launches_per_step"] = sum(1 for _, _, fl in prof if fl > 0)
    res["roofline"]["conv_kernels_ms_eager"] = round(conv_ms, 4)
    res["roofline"]["conv_kernels_frames_eager"] = S * F // chains      # the profiler times one chain's launches alone on the device
    res["roofline"]["conv_kernels_tflops_eager"] = round(sum(fl for _, _, fl in prof) / (conv_ms * 1e-3) / 1e12, 2)
    top = sorted(prof, key=lambda r: -r[1])[:6]
    res["roofline"]["slowest_launches"] = [{"op": n, "ms": round(ms, 4), "tflops": round(fl / (ms * 1e-3) / 1e12, 1) if ms > 0 else 0} for n, ms, fl in top]
    # every launch of the forward pass alone on the device (eager, HIP events): [op, microseconds, TFLOP/s]
    res["roofline"]["launches_eager"] = [[n, round(ms * 1e3, 1), round(fl / (ms * 1e-3) / 1e12, 1) if ms > 0 and fl > 0 else 0] for n, ms, fl in prof]
    det.close()
    trk.close()

    # ---- the same workload with the frames arriving from page-locked HOST memory every step (PCIe-inclusive; the reference's
    # detect() takes a host ndarray, detector.py:98-111).  Reported beside `value`, never as `value` ----
    if world == 1 and not args.host_frames and not args.no_host_leg:
        mh = measure(F, args.steps, args.warmup, collective=False, host=True)
        res["host_frames"] = {"value": round(S * F * args.steps / mh["elapsed"], 1), "unit": "frames/s",
                              "ms_per_step": round(mh["elapsed"] / args.steps * 1e3, 4),
                              "ratio_to_hbm_resident": round(S * F * args.steps / mh["elapsed"] / fps, 4),
                              "mode": args.host_mode, "stages": mh["stages"],
                              "source": f"page-locked ring of {R} frames per stream, {S * F} x {per} B per step over PCIe"}
        mh["det"].close()
        mh["trk"].close()

    # ---- the same workload with fewer frames of every stream per launch set (2: the configuration of rounds 1-2's records;
    # 1: no batching over time), for comparison ----
    if F > 1 and world == 1 and not args.host_frames and not args.no_compare:
        for Fc in sorted({1, 2} - {F}, reverse=True):
            m1 = measure(Fc, args.steps, args.warmup, collective=False, prewarm_s=0.3)
            f1 = m1["fwd_ms"] / args.steps
            if getattr(m1["det"].model, "chains", 1) > 1 or getattr(m1["det"].model, "stages", 1) > 1:
                f1 = min(f1, m1["elapsed"] / args.steps * 1e3)      # overlapped batches: the event span is capped by the step period (as above)
            res[{1: "one_frame_per_stream_per_step", 2: "two_frames_per_stream_per_step"}[Fc]] = {
                "value": round(S * Fc * args.steps / m1["elapsed"], 1), "unit": "frames/s",
                "ms_per_step": round(m1["elapsed"] / args.steps * 1e3, 4), "forward_ms_per_step": round(f1, 4),
                "achieved_tflops": round(m1["det"].model.conv_flops_per_frame * S * Fc / (f1 * 1e-3) / 1e12, 2)}
            if Fc == 1:        # first-class: the configuration exactly as BASELINE config 4 words it (one frame of every stream per step)
                res["value_no_temporal_batching"] = res["one_frame_per_stream_per_step"]["value"]
                res["frac_no_temporal_batching"] = round(res["one_frame_per_stream_per_step"]["achieved_tflops"] / MFMA_F16_DENSE_PEAK_TFLOPS, 4)
            m1["det"].close()
            m1["trk"].close()
    elif F == 1:
        res["value_no_temporal_batching"] = res["value"]
        res["frac_no_temporal_batching"] = res["roofline"]["frac"]

    # ---- BASELINE configs 3 / 5: the 200- and 500-box association sequences, bit-checked against the reference-generated fixtures ----
    if world == 1 and not args.no_tracker_stress:
        res["tracker_stress"] = tracker_stress(pkg, dev)

    # ---- single-stream latency (BASELINE config 1/2 shape: batch 1, sync per frame) ----
    if not args.no_latency:
        det1 = pkg.Detector(wpath, input_size=(size, size), max_det=args.max_det, device=f"cuda:{dev}", batch=1, warmup=False)
        trk1 = core_cls(device=dev, n_streams=1, max_dets=max(128, args.max_det), max_tracks=2048)
        lat = []
        gc.disable()
        for t in range(50 + 300):
            t1 = time.perf_counter()
            det1.enqueue([ring.ptr + (t % R) * per], height=size, width=size)
            trk1.update_from_detector(det1)
            det1.fetch()
            lat.append(time.perf_counter() - t1)
        gc.enable()
        lat = np.asarray(lat[50:]) * 1e3                         # 50 warm-up frames discarded (config/default.yaml:88)
        res["latency_single_stream_ms"] = {"p50": round(float(np.percentile(lat, 50)), 4), "mean": round(float(lat.mean()), 4),
                                           "p95": round(float(np.percentile(lat, 95)), 4), "p99": round(float(np.percentile(lat, 99)), 4),
                                           "fps": round(float(1e3 / lat.mean()), 1), "frames": int(len(lat))}
        det1.close()
        trk1.close()

    if not args.no_cpu_baseline and world == 1:
        if weights is None:
            weights = pkg.weights.load(wpath)[0]
        res["cpu_baseline"] = cpu_baseline(pkg, weights, args)
    print(json.dumps(res), flush=True)
    sync.barrier()
    sync.close()


if __name__ == "__main__":
    main()
