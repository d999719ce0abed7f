// libedmp_hip.hip — single translation unit of libedmp_hip.so (the three parts share the context structs).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared libedmp_hip.hip -o ../libedmp_hip.so
#include "unet.hip"
#include "guide.hip"
#include "sampler.hip"
