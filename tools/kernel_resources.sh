#!/bin/bash
# Registers / LDS / scratch of every kernel of the library (no GPU needed): compiles csrc/gsrast_capi.hip with -save-temps into a
# temporary directory and prints the resource lines of the kernels whose name matches $1 (default: all).
# usage: [EXTRA="-DFOO=1"] tools/kernel_resources.sh [pattern]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d /tmp/gsres.XXXXXX)
cd "$D"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics \
    -fno-slp-vectorize $EXTRA -save-temps -c "$ROOT/saro-gs_amd/csrc/gsrast_capi.hip" -o gs.o 2>/dev/null
S=$(ls *gfx950*.s | head -1)
python3 - "$S" "${1:-.}" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    if not pat.search(name):
        continue
    g = lambda k: (re.search(r"\.amdhsa_" + k + r" (\S+)", body) or [None, "?"])[1]
    print(f"{name[:90]:90s} vgpr {g('next_free_vgpr'):>4s} sgpr {g('next_free_sgpr'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s}")
PY
echo "(assembly: $D/$S)"
