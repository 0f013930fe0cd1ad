#!/usr/bin/env bash
# Builds libqpn_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libqpn_hip.so"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
      -Wall -Wno-unused-parameter \
      -o "$out" \
      "$here/qpn_capi.hip" "$here/qpn_avi_solve.hip" "$here/qpn_avi_reg.hip" "$here/qpn_avi_big.hip" "$here/qpn_avi_schur.hip" "$here/qpn_avi_schur_big.hip" "$here/qpn_avi_schur_mid.hip" "$here/qpn_kkt.hip" "$here/qpn_verify.hip" \
      "$@"
echo "built $out"
