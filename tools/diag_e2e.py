"""Free-running fp16 engine vs fp32 oracle over many frames, decision by decision (tests/nms_audit.py).

One engine configuration per process (the tile hooks are read from the environment at create):
    python3 tools/diag_e2e.py --tag autotune --autotune 1 --seeds 1234:1250 --out gpurun_out/diag
prints, per frame, the audit line, the survivor difference and whether every difference is covered by near-tie
decisions; writes <out>/<tag>.json (+ the tuned launch list).  Any "hard" row is a kernel defect in that configuration.
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--autotune", type=int, default=1)
    ap.add_argument("--seeds", default="1234:1244")
    ap.add_argument("--hw", default="640x640")
    ap.add_argument("--out", default="gpurun_out/diag")
    ap.add_argument("--dump", type=int, default=0, help="also save the engine's and the oracle's pre-NMS tensors")
    a = ap.parse_args()
    import rtmodt_amd  # noqa: F401
    pkg = sys.modules["rtmodt_amd"]
    from oracle import yolo_oracle as Y
    import nms_audit as NA

    os.makedirs(a.out, exist_ok=True)
    h, w = (int(v) for v in a.hw.split("x"))
    lo, hi = (int(v) for v in a.seeds.split(":"))
    classes = [0, 1, 2, 3, 5, 7, 17, 18]
    wts = pkg.weights.synthetic("s", input_size=640)
    print("weights digest", pkg.weights.digest(wts), flush=True)
    path = os.path.join(tempfile.mkdtemp(prefix="rtmodt_diag_"), "s640.rtw")
    pkg.weights.save(path, wts, "s")
    wts, _, _, _ = pkg.weights.load(path)
    det = pkg.Detector(path, input_size=(640, 640), classes=classes, max_det=300, warmup=False, autotune=bool(a.autotune))
    launches = [n for n, _, _ in det.profile(1)]
    rows = []
    for seed in range(lo, hi):
        frame = np.ascontiguousarray(pkg.synth.frames(1, h, w, seed=seed)[0])
        d = det.detect(frame)
        _, _, pred = det.debug_fetch(0, want_input=False, want_heads=False)
        (rx, rc, rk), im = Y.detect(frame, wts, "s", (640, 640), 0.35, 0.45, classes, 300, return_intermediate=True)
        dets_e, anch_e = Y.non_max_suppression(pred, 0.35, 0.45, classes, False, 300)
        assert np.array_equal(d.xyxy.view(np.int32), Y.scale_boxes(dets_e[:, :4], 640, 640, h, w).view(np.int32)), "engine NMS != oracle NMS on the engine's tensor"
        res = NA.audit(pred, im["pred"], 0.35, 0.45, classes, score_tol=0.01, iou_tol=0.02)
        only_e, only_o = NA.survivors_diff(anch_e, im["anchors"])
        print(f"[{a.tag}] seed {seed}: oracle {len(rc)} / engine {len(d)} detections; only engine {only_e}; only oracle {only_o}")
        print(NA.describe(res), flush=True)
        rows.append(dict(seed=seed, n_oracle=int(len(rc)), n_engine=int(len(d)), only_engine=only_e, only_oracle=only_o,
                         near=[(k, list(an), g) for k, an, g in res["near"]], hard=[(k, list(an), g) for k, an, g in res["hard"]],
                         score_drift=res["score_drift"], iou_drift=res["iou_drift"], n_candidates=res["n_candidates"]))
        if a.dump:
            np.savez_compressed(os.path.join(a.out, f"{a.tag}_{seed}.npz"), pred_e=pred, pred_o=im["pred"], anch_e=anch_e, anch_o=im["anchors"])
    det.close()
    json.dump(dict(tag=a.tag, autotune=a.autotune, env={k: v for k, v in os.environ.items() if k.startswith("RTMODT_")}, launches=launches, frames=rows),
              open(os.path.join(a.out, f"{a.tag}.json"), "w"), indent=1)
    n_hard = sum(len(r["hard"]) for r in rows)
    n_diff = sum(len(r["only_engine"]) + len(r["only_oracle"]) for r in rows)
    print(f"[{a.tag}] {len(rows)} frames: {n_diff} survivor differences, {sum(len(r['near']) for r in rows)} near-tie flips, {n_hard} HARD mismatches")


if __name__ == "__main__":
    main()
