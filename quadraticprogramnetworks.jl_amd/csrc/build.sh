#!/usr/bin/env bash
# Builds libqpn_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
# One object per translation unit (csrc/_obj/, rebuilt only when the source or a header is newer), compiled in
# parallel, then one link.  Extra arguments go to every compile (e.g. -DQPN_STAMPS, -Rpass-analysis=kernel-resource-usage);
# QPN_OUT=<path> names another output library, QPN_OBJ=<dir> another object directory (diagnostic builds).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${QPN_OUT:-$here/../libqpn_hip.so}"
obj="${QPN_OBJ:-$here/_obj}"
mkdir -p "$obj"
units=(qpn_capi qpn_avi_solve qpn_avi_reg qpn_avi_big qpn_avi_schur qpn_avi_schur_big qpn_avi_schur_big2 qpn_avi_schur_wg qpn_avi_schur_wg2 qpn_avi_schur48 qpn_kkt qpn_pieces qpn_verify)
flags=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-parameter "$@")
# a change of flags rebuilds everything
sig="$(printf '%s ' "${flags[@]}" | sha1sum | cut -c1-16)"
if [[ ! -f "$obj/.flags" || "$(cat "$obj/.flags")" != "$sig" ]]; then rm -f "$obj"/*.o; echo "$sig" > "$obj/.flags"; fi
newest_hdr=0
for h in "$here"/*.h "$here"/../../include/*.h; do t=$(stat -c %Y "$h"); (( t > newest_hdr )) && newest_hdr=$t; done
todo=()
for u in "${units[@]}"; do
    src="$here/$u.hip"; o="$obj/$u.o"
    if [[ ! -f "$o" ]] || (( $(stat -c %Y "$src") > $(stat -c %Y "$o") )) || (( newest_hdr > $(stat -c %Y "$o") )); then todo+=("$u"); fi
done
if (( ${#todo[@]} )); then
    jobs="${QPN_JOBS:-6}"
    printf '%s\n' "${todo[@]}" | xargs -P "$jobs" -I{} hipcc "${flags[@]}" -c "$here/{}.hip" -o "$obj/{}.o"
fi
objs=(); for u in "${units[@]}"; do objs+=("$obj/$u.o"); done
hipcc --offload-arch=gfx950 -fPIC -shared -o "$out" "${objs[@]}"
echo "built $out"
