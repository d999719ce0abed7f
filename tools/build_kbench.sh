#!/bin/bash
# kbench against the sharded kernel objects of the last __graft_entry__.build() (edmp_amd/csrc/_obj/shard*.o): about a
# minute instead of the quarter of an hour a single translation unit takes.  -DEDMP_STAMPS builds need the single unit:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DEDMP_STAMPS tools/kbench.hip -o tools/kbench_stamps
set -e
cd "$(dirname "$0")/.."
ls edmp_amd/csrc/_obj/shard0.o > /dev/null || python -c "import __graft_entry__ as g; g.build(force=True)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DEDMP_SHARDED -c tools/kbench.hip -o edmp_amd/csrc/_obj/kbench.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 edmp_amd/csrc/_obj/kbench.o edmp_amd/csrc/_obj/shard*.o -o tools/kbench
