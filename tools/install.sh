#!/bin/bash
# Installs the drop-in into a prefix with the reference's layout (CMakeLists.txt install rules of the reference:
# include/super4pcs/**, lib/, lib/cmake/Super4PCSConfig*.cmake), so that find_package(Super4PCS) works from
# -DCMAKE_PREFIX_PATH=<prefix> exactly as after the reference's `make install`.   Usage: tools/install.sh <prefix>
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PREFIX="${1:?usage: tools/install.sh <prefix>}"
[ -f "$ROOT/super4pcs_amd/lib/libsuper4pcs_amd.so" ] || { echo "build first: python __graft_entry__.py" >&2; exit 1; }
mkdir -p "$PREFIX/include" "$PREFIX/lib/cmake"
cp -r "$ROOT/include/." "$PREFIX/include/"
cp "$ROOT/super4pcs_amd/lib/libsuper4pcs_amd.so" "$PREFIX/lib/"
cp "$ROOT/cmake/Super4PCSConfig.cmake" "$ROOT/cmake/Super4PCSConfigVersion.cmake" "$PREFIX/lib/cmake/"
echo "installed to $PREFIX"
