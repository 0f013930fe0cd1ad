#!/usr/bin/env bash
# Device ISA of one kernel source -> /tmp/qpn_isa/<name>.s plus the per-kernel resource lines.
set -euo pipefail
src="$1"; name="$(basename "${src%.hip}")"; mkdir -p /tmp/qpn_isa
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only "${@:2}" -o "/tmp/qpn_isa/$name.s" "$src" 2>/tmp/qpn_isa/$name.err || { grep -m5 -A3 error /tmp/qpn_isa/$name.err; exit 1; }
grep -E "^\s+\.(name|vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size):" "/tmp/qpn_isa/$name.s"
