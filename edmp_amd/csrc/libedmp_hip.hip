// libedmp_hip.hip — single translation unit of libedmp_hip.so (the three parts share the context structs).
// Build (__graft_entry__.build): this file with -DEDMP_SHARDED -c, success.hip -c and kernel_shard.hip with
// -DEDMP_SHARD=0..EDMP_KERNEL_SHARDS-1 -c (kernel_instances.h), all in parallel, then one link.  Without -DEDMP_SHARDED it is still a
// complete single-unit build of everything but success.hip (-DEDMP_STAMPS phase-timing experiments, scripts/phase_stamps.py):
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared libedmp_hip.hip success.hip -o ../libedmp_hip.so  (~15 min)
#include "unet.hip"
#include "guide.hip"
#include "sampler.hip"
#include "rccl_hook.hip"
