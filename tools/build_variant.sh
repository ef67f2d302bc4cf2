#!/bin/bash
# Diagnostic builds of the library with extra flags (e.g. -DCFZ_STAMPS): tools/build_variant.sh tools/_libcfz_stamps.so -DCFZ_STAMPS
# Same two translation units and optimisation level (-O3) as __graft_entry__.build.
out=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/conflict_rez_amd/csrc
T=$(mktemp -d)
F="--offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-value"
/opt/rocm/bin/hipcc $F -O3 "$@" -c -o $T/e.o $C/cfz_engine.hip &
/opt/rocm/bin/hipcc $F -O3 "$@" -c -o $T/p.o $C/cfz_planning.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $T/e.o $T/p.o && rm -rf $T
