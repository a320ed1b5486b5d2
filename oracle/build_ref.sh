#!/bin/bash
# Builds the reference's OWN native module (deformation: dgrad <-> mesh, C++/Eigen/pybind11) straight from the
# sources where they lie under /root/reference, with g++ on its three .cpp files -- not through its CMake build.
# Output goes ONLY to oracle/_ref/ (git-ignored).  Used by tests/oracle for the "next row" dgrad -> mesh solve.
# TEST INFRASTRUCTURE ONLY; build container only (needs /root/reference).
set -euo pipefail
REF=${SDFA_REFERENCE_ROOT:-/root/reference}
SRC=$REF/deformation/cpp
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
mkdir -p "$OUT"
EXT=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
if [ -f "$OUT/deformation$EXT" ] && [ "$OUT/deformation$EXT" -nt "$SRC/src/pybind.cpp" ]; then exit 0; fi
g++ -O2 -std=c++14 -shared -fPIC -w \
    -I"$SRC/ext/eigen3" -I"$SRC/ext/pybind11/include" -I"$SRC/ext/spdlog/include" -I"$SRC/src" -I"$PYINC" \
    "$SRC/src/pybind.cpp" "$SRC/src/log.cpp" "$SRC/src/rotation/utils_rotation.cpp" \
    -o "$OUT/deformation$EXT"
echo "built $OUT/deformation$EXT"
