#!/bin/bash
# usage: tools/isa.sh <file stem in autoforce_amd/csrc> [kernel-name filter]: gfx950 ISA + per-kernel instruction mix
cd "$(dirname "$0")/../autoforce_amd/csrc"
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics --offload-device-only -S $1.hip -o /tmp/isa/$1.s 2>&1 | grep -E "error|scratch" | head
python3 ../../tools/isa_stats.py /tmp/isa/$1.s "$2"
