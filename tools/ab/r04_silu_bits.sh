#!/bin/bash
# r04: is silu4() (packed multiplies / add) bit-identical to silu()?  pp_probe built twice -- as shipped, and with -DRTMODT_SILU_SCALAR (silu4 calls silu() per
# element) -- prints an FNV hash of every tile's output on every benchmarked shape (3x3, 1x1, 3x3 s2, with and without shortcut); the two listings must be equal.
O=gpurun_out/r04/silu_bits; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/pp_probe_packed tools/probes/pp_probe.hip 2> /dev/null || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRTMODT_SILU_SCALAR -o /tmp/pp_probe_scalar tools/probes/pp_probe.hip 2> /dev/null || exit 1
timeout -k 10 300 /tmp/pp_probe_packed "" 2 | grep -o "^   [a-z0-9:x/-]* \|bits [0-9a-f]*\|^[0-9A-Za-z].*GFLOP)" > $O/packed.txt || exit 1
timeout -k 10 300 /tmp/pp_probe_scalar "" 2 | grep -o "^   [a-z0-9:x/-]* \|bits [0-9a-f]*\|^[0-9A-Za-z].*GFLOP)" > $O/scalar.txt || exit 1
if cmp -s $O/packed.txt $O/scalar.txt; then echo "IDENTICAL: $(grep -c bits $O/packed.txt) outputs hashed"; else echo "DIFFERENT"; diff $O/packed.txt $O/scalar.txt | head; fi | tee $O/result.txt
