// libedmp_hip.hip — single translation unit of libedmp_hip.so (the three parts share the context structs).
// Build (__graft_entry__.build): this file with -DEDMP_SHARDED -c, kernel_shard.hip with -DEDMP_SHARD=0..11 -c, all in
// parallel, then one link.  Without -DEDMP_SHARDED it is still a complete single-unit build (tools/kbench.hip, -DEDMP_STAMPS
// experiments): hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared libedmp_hip.hip -o ../libedmp_hip.so  (~15 min)
#include "unet.hip"
#include "guide.hip"
#include "sampler.hip"
