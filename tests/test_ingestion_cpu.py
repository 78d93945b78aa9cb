"""Frame ingest front-end: the reader's contract -- latest frame only, ids, reconnect back-off, give-up --
against scripted capture back-ends; the raw BGR24 back-end against a file.  No GPU."""
import os
import sys
import threading
import time

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmodt_amd  # noqa: E402,F401

pkg = sys.modules["rtmodt_amd"]
ing = pkg.ingestion


class HostRing:
    """Stand-in for pipeline.PinnedFrameRing (which needs the GPU runtime to page-lock memory)."""
    def __init__(self, slots, h, w):
        self.mem = np.zeros((slots, h, w, 3), np.uint8)
        self.slots = slots

    def frame(self, i):
        return self.mem[i % self.slots]


def wait_for(cond, timeout=3.0):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < timeout:
        if cond():
            return True
        time.sleep(0.002)
    return False


def test_reader_signature_matches_reference():
    import inspect
    p = inspect.signature(ing.FrameReader.__init__).parameters
    for name, default in (("buffer_size", 1), ("target_fps", 30), ("reconnect_delay", 3.0), ("max_reconnects", 10), ("resolution", None)):
        assert p[name].default == default
    assert list(p)[:3] == ["self", "source", "backend"]
    assert ing.RTSPReader is ing.FrameReader
    for m in ("start", "read", "stop", "is_alive", "__enter__", "__exit__"):
        assert hasattr(ing.FrameReader, m)
    with pytest.raises(ValueError):
        ing.FrameReader("x", backend="no-such-backend")


def test_latest_frame_only_and_ids():
    frames = pkg.synth.frames(5, 48, 64, seed=3)
    ring = HostRing(3, 48, 64)
    with ing.FrameReader("ring", backend="synthetic", resolution=(64, 48), frames=frames, fps=400.0, ring=ring) as r:
        assert r.is_alive
        assert wait_for(lambda: r.read()[0])
        ok, f1, id1 = r.read()
        assert ok and f1.shape == (48, 64, 3) and np.array_equal(f1, frames[(id1 - 1) % 5])
        assert wait_for(lambda: r.read()[2] >= id1 + 4)
        ok, f2, id2 = r.read(copy=False)
        assert np.array_equal(f2, frames[(id2 - 1) % 5])
        assert any(f2.base is ring.mem or f2 is ring.frame(k) or np.shares_memory(f2, ring.mem) for k in range(3))   # the slot itself, not a copy
        d0 = r.dropped
        time.sleep(0.05)                           # nobody reads for 20 frame periods
        assert wait_for(lambda: r.dropped >= d0 + 5)   # those frames were overwritten, never queued
        ok, f3, id3 = r.read()
        assert id3 >= id2 + 5 and np.array_equal(f3, frames[(id3 - 1) % 5])
    assert not r.is_alive
    ok, f4, _ = r.read()
    assert ok                                     # the last frame stays readable after stop, like the reference's


def test_leased_slots_are_never_rewritten():
    """`read(copy=False)` hands out a page-locked ring slot that the detector may still be uploading (or reading in
    place) while a free-running source produces newer frames.  The slot is a lease: the capture thread skips it until
    `release(frame_id)`; a ring too short for `buffer_size` is refused; running out of leasable slots degrades to a copy (counted in
    `lease_misses`; the reference's `read` never raises) instead of tearing a frame or failing; `stop()` clears the leases."""
    frames = pkg.synth.frames(7, 48, 64, seed=11)
    with pytest.raises(ValueError):
        ing.FrameReader("ring", backend="synthetic", resolution=(64, 48), frames=frames, ring=HostRing(2, 48, 64))
    with pytest.raises(ValueError):
        ing.FrameReader("ring", backend="synthetic", resolution=(64, 48), frames=frames, ring=HostRing(4, 48, 64), buffer_size=3)
    ring = HostRing(5, 48, 64)                     # 3 leasable slots: a detector with three batches in flight
    with ing.FrameReader("ring", backend="synthetic", resolution=(64, 48), frames=frames, fps=2000.0, ring=ring, buffer_size=3) as r:
        held = []
        for _ in range(3):
            last = held[-1][0] if held else 0
            assert wait_for(lambda: r.read()[2] > last)
            ok, f, fid = r.read(copy=False)
            assert ok
            held.append((fid, f, f.copy()))
        assert r.leased == 3
        assert wait_for(lambda: r.read()[2] >= held[-1][0] + 40)      # 40 newer frames went through the two remaining slots
        for fid, view, snap in held:
            assert np.array_equal(view, snap) and np.array_equal(view, frames[(fid - 1) % 7]), fid      # untouched while on lease
        newest = r.read()[2]
        assert wait_for(lambda: r.read()[2] > newest)
        while True:                                   # a fourth lease (of a frame in an unleased slot) is served by a COPY
            ok, f, fid = r.read(copy=False)
            if fid not in [h[0] for h in held]:
                break
        assert ok and r.lease_misses >= 1 and r.leased == 3
        assert not any(np.shares_memory(f, ring.slot(i)) for i in range(ring.slots)) if hasattr(ring, "slot") else True
        r.release(fid)                                # nothing to give back for a copy: a no-op
        assert r.leased == 3
        r.release(held[0][0])
        r.release(held[0][0])                         # idempotent
        assert r.leased == 2
        ok, f, fid = r.read(copy=False)               # ... and now it is granted
        assert ok and r.leased == 3
        for fid_, _, _ in held[1:]:
            r.release(fid_)
        r.release(fid)
        assert r.leased == 0
        r.read(copy=False)
        assert r.leased == 1
    assert r.leased == 0                              # stop() (the context manager's exit) clears what was still on lease


def test_holders_of_one_frame_id_are_counted():
    """(1) two read(copy=False) of the SAME (still latest) frame id are two holders of one lease -- the first release() must not free the slot
    under the second; (2) a copy-served read (no slot could be leased) is a holder of its frame id too, so its release() can never drop the real lease a
    later read of that id obtained."""
    frames = pkg.synth.frames(5, 48, 64, seed=3)
    ring = HostRing(4, 48, 64)                     # 2 leasable slots
    src = ing.FrameReader("ring", backend="synthetic", resolution=(64, 48), frames=frames, fps=1e-3, ring=ring, buffer_size=2)      # ~one frame, then nothing for minutes
    with src as r:
        assert wait_for(lambda: r.read()[0])
        ok, a, fid = r.read(copy=False)
        ok2, b, fid2 = r.read(copy=False)
        assert ok and ok2 and fid == fid2 and np.shares_memory(a, b) and r.leased == 1
        r.release(fid)
        assert r.leased == 1                          # the second holder still has it
        r.release(fid)
        assert r.leased == 0
        r.release(fid)                                # nobody left: a no-op
        assert r.leased == 0
        # (2) fill the leasable slots with OTHER frame ids by hand, so that this id can only be served by a copy
        with r._lock:
            other = [s for s in range(ring.slots) if s != r._latest_slot][:2]
            for s in other:
                r._leases[s] = 1
        ok, c, fid3 = r.read(copy=False)
        assert ok and fid3 == fid and r.lease_misses == 1 and not np.shares_memory(c, a)      # a pageable copy
        with r._lock:
            r._leases.pop(other[0])                   # a slot comes back ...
        ok, d, fid4 = r.read(copy=False)              # ... and the same id now gets a real lease
        assert ok and fid4 == fid and np.shares_memory(d, a) and r.leased == 2
        r.release(fid)                                # the copy's holder is done: the real lease must survive
        assert r.leased == 2 and r._latest_slot in r._leases
        r.release(fid)
        assert r._latest_slot not in r._leases


def test_read_before_first_frame():
    class Never:
        opened = True
        def __init__(self, *a, **k): pass
        def grab(self):
            time.sleep(0.01)
            return True
        def retrieve(self, out=None):
            return False, None
        def release(self):
            self.opened = False
    ing.register_backend("never", Never)
    with ing.FrameReader("s", backend="never") as r:
        time.sleep(0.05)
        ok, f, fid = r.read()
        assert (ok, f, fid) == (False, None, 0)


def test_reconnect_backoff_and_give_up():
    script = {"opens": 0, "fail_opens": 2, "grabs_before_drop": 3}

    class Flaky:
        def __init__(self, source, resolution=None, **_):
            script["opens"] += 1
            if 1 < script["opens"] <= 1 + script["fail_opens"]:
                raise ConnectionError(f"Cannot open stream: {source}")
            self.opened = True
            self.n = 0
        def grab(self):
            self.n += 1
            if script["opens"] == 1 and self.n > script["grabs_before_drop"]:
                return False                       # the stream drops after three frames
            time.sleep(0.001)
            return True
        def retrieve(self, out=None):
            return True, np.full((4, 4, 3), script["opens"], np.uint8)
        def release(self):
            self.opened = False
    ing.register_backend("flaky", Flaky)
    t0 = time.perf_counter()
    r = ing.FrameReader("cam", backend="flaky", reconnect_delay=0.02, max_reconnects=5).start()
    assert wait_for(lambda: script["opens"] >= 4 and r.read()[0] and int(r.read()[1][0, 0, 0]) == 4)
    waited = time.perf_counter() - t0
    assert waited >= 0.02 * (1 + 2 + 3) * 0.9      # back-off grows with the failures in a row: delay x 1, x 2, x 3
    assert r.reconnects == 3 and r.is_alive
    r.stop()
    # a source that never comes back: max_reconnects failures in a row end the thread
    script.update(opens=0, fail_opens=10 ** 6, grabs_before_drop=1)
    r = ing.FrameReader("cam", backend="flaky", reconnect_delay=0.005, max_reconnects=3).start()
    assert wait_for(lambda: not r.is_alive, timeout=5.0)
    assert r.reconnects == 3
    r.stop()
    with pytest.raises(ConnectionError):
        script.update(opens=1, fail_opens=10 ** 6)
        ing.FrameReader("cam", backend="flaky").start()


def test_raw_bgr24_backend(tmp_path):
    frames = pkg.synth.frames(6, 36, 52, seed=9)
    path = tmp_path / "clip.bgr"
    path.write_bytes(frames.tobytes() + b"\x00" * 100)          # a trailing partial frame is the end of the stream
    cap = ing.RawVideoCapture(str(path), resolution=(52, 36))
    got = []
    while cap.grab():
        ok, f = cap.retrieve()
        got.append(f)
    assert len(got) == 6 and all(np.array_equal(a, b) for a, b in zip(got, frames))
    cap.release()
    with pytest.raises(ValueError):
        ing.RawVideoCapture(str(path))
    with pytest.raises(ConnectionError):
        ing.FrameReader(str(tmp_path / "missing.bgr"), backend="raw", resolution=(52, 36)).start()
    # through the reader: the file ends, the reader reconnects (re-opens the file) and keeps serving frames
    r = ing.FrameReader(str(path), backend="raw", resolution=(52, 36), reconnect_delay=0.01, max_reconnects=3).start()
    assert wait_for(lambda: r.reconnects >= 1 and r.read()[0])
    ok, f, fid = r.read()
    assert ok and fid >= 6 and any(np.array_equal(f, fr) for fr in frames)
    r.stop()
