#!/bin/bash
# Build libblaze_hip.so from a git revision's sources into ab/<name>.so (A/B partner for tools/ab_perf.py).
# usage: tools/build_ref_lib.sh <rev> <name>
set -e
rev=${1:-HEAD}; name=${2:-base}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$rev" blaze_amd/csrc include | tar -x -C "$tmp"
mkdir -p "$root/ab"
(cd "$tmp" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value -shared -fPIC -pthread \
    -o "$root/ab/$name.so" blaze_amd/csrc/blaze_hip.hip -ldl -lz)
rm -rf "$tmp"
echo "built ab/$name.so from $rev"
