#!/usr/bin/env python3
"""(Lives under tests/ because it times the CPU restatement of the reference's path; only tests, smoke() and the bench's CPU baseline may touch oracle/.)
The reference's CPU path at the level of the WHOLE BOX: bench.py's `cpu_baseline` leg -- torch fp32 CPU forward + NumPy
decode / NMS + the C tracker on 16 threads -- started as N independent processes side by side, each pinned to its own 16 hardware threads, N = os.cpu_count() // 16
by default (the forward pass does not scale past ~16 threads inside one process: separate streams in separate processes is how a CPU deployment would use the box).
Prints one JSON object: aggregate frames/s, the per-process figures, threads used.

    python tests/perf/cpu_baseline_node.py [--procs N] [--threads 16]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    ncpu = os.cpu_count() or 1
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(ncpu))
    procs = args.procs if args.procs > 0 else max(1, len(allowed) // args.threads)
    children = []
    for i in range(procs):
        cpus = allowed[i * args.threads:(i + 1) * args.threads] or allowed
        env = dict(os.environ, OMP_NUM_THREADS=str(args.threads), MKL_NUM_THREADS=str(args.threads))

        def pin(cpus=cpus):
            try:
                os.sched_setaffinity(0, cpus)
            except (AttributeError, OSError):
                pass
        children.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline-only", "--cpu-threads", str(args.threads)],
                                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env, preexec_fn=pin))
    per = []
    for c in children:
        out, _ = c.communicate(timeout=900)
        lines = [l for l in out.splitlines() if l.startswith("{")]
        if c.returncode != 0 or not lines:
            raise SystemExit(f"a cpu_baseline process failed (exit code {c.returncode})")
        per.append(json.loads(lines[-1]))
    print(json.dumps({"value": round(sum(p["value"] for p in per), 2), "unit": "frames/s", "processes": procs, "threads_per_process": args.threads,
                      "hardware_threads_used": procs * args.threads, "hardware_threads_available": len(allowed), "kind": "port",
                      "per_process_frames_s": [p["value"] for p in per], "per_process_p50_ms": [p["p50_ms"] for p in per],
                      "what": "N simultaneous copies of bench.py's cpu_baseline leg (YOLOv8s fp32 torch-CPU forward + NumPy decode/NMS + C tracker), one 16-thread process per 16 hardware threads"}))


if __name__ == "__main__":
    main()
