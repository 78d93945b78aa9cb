import sys, os, tempfile, faulthandler
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import rtmodt_amd
pkg = sys.modules["rtmodt_amd"]
wpath = os.path.join(tempfile.gettempdir(), "dbg_n.rtw")
pkg.weights.save(wpath, pkg.weights.synthetic("s", input_size=320, calibrate=None), "s")
print("creating", flush=True)
det = pkg.Detector(wpath, input_size=(320,320), batch=4, chains=2, warmup=False, autotune=False)
print("created", flush=True)
out = det.detect_batch(list(pkg.synth.frames(4,320,320)))
print("ok", [len(o) for o in out])
