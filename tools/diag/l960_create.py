"""Diagnosis script of round 5: create a YOLOv8l / x detector at a few sizes (no l / x case existed before: a dangling tensor reference in the
Detect head builder gave them garbage shapes).  gpurun -- 'python tools/diag/l960_create.py'"""
import os, sys, tempfile
sys.path.insert(0, '.')
import rtmodt_amd
pkg = sys.modules['rtmodt_amd']
for scale in ('l', 'x'):
  path = os.path.join(tempfile.gettempdir(), scale + '160.rtw')
  pkg.weights.save(path, pkg.weights.synthetic(scale, input_size=160), scale)
  for size, at in ((320, False), (960, True)):
      try:
          det = pkg.Detector(path, input_size=(size, size), max_det=300, warmup=False, autotune=at)
          import numpy as np
          d = det.detect(pkg.synth.frames(1, size, size, seed=3)[0])
          print(scale, size, at, 'ok', len(det.profile(1)), 'launches,', len(d), 'detections')
          det.close()
      except Exception as e:
          print(scale, size, at, 'FAILED', e)
