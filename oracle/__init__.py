"""CPU oracle for the detect+track hot path -- TEST INFRASTRUCTURE ONLY.

Nothing in the shipped package imports this directory.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
use it, and there only as the checker / the timed CPU baseline -- never as the
thing that produces the product's results.

Pinning status (see DESIGN.md "Oracle"):

* tracker  -- PINNED: ``tracker_oracle.py`` is checked bit-for-bit against the
  reference's own ``src/tracking/tracker.py`` (greedy branch) through the
  fixtures under ``tests/golden/`` written by ``oracle/gen_golden_tracker.py``.
* detector -- PARITY UNPINNED: the arithmetic lives in un-vendored, un-pinned
  ``ultralytics>=8.1.0`` / ``torchvision>=0.16`` / ``opencv`` (absent here), the
  reference holds no test or golden vector for it, so ``yolo_oracle.py`` is a
  restatement of the published algorithm, self-checked by known-answer
  parameter / FLOP counts and by independent torch-CPU arithmetic.
* lapjv assignment branch -- PARITY UNPINNED (``lap`` absent here).
"""
