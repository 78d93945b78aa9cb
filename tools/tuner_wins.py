"""Which tile the autotuner picked for every launch, from a RTMODT_TUNE_LOG=1 log (stderr of bench.py / any detector build).
    python3 tools/tuner_wins.py gpurun_out/r04/bench3.err > profiles/r04/tuner_wins.txt
A log holds one block per detector that was built (bench.py: 32, 16 and 8 frames per launch); a launch's candidates are the
consecutive "[tune] <name> <tile> <us>" lines with the same name."""
import collections
import re
import sys

pat = re.compile(r"^\[tune\] (\S+(?: \([^)]*\))?)\s+(\S+)\s+([0-9.]+) us\s+\((\d+) KiB LDS\)")
wins = collections.Counter()
cands = collections.Counter()
rows = []
cur, best = None, None


def flush():
    if cur is not None:
        rows.append((cur, best[1], best[0], second, None))
        wins[best[1]] += 1


second = None
for line in open(sys.argv[1]):
    m = pat.match(line)
    if not m:
        continue
    name, tile, us = m.group(1), m.group(2), float(m.group(3))
    if name != cur:
        flush()
        cur, best, second = name, (us, tile), None
    else:
        if us < best[0]:
            best, second = (us, tile), best
        elif second is None or us < second[0]:
            second = (us, tile)
    cands[tile] += 1
flush()
print(f"# {len(rows)} tuned launches in {sys.argv[1]}")
print("# wins by tile (candidate count = launches where it was legal):")
for t, n in sorted(cands.items(), key=lambda kv: -wins[kv[0]]):
    print(f"#   {t:20s} {wins[t]:4d} wins of {n:4d}")
print("# launch, winner, us, runner-up, us")
for name, tile, us, t2, us2 in rows:
    print(f"{name:34s} {tile:20s} {us:8.2f}   {(t2[1] if t2 else '-'):20s} {(t2[0] if t2 else 0):8.2f}")
