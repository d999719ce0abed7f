// chain_unit.hip — translation unit of the persistent layer-chain kernel (chain.hip): compiled next to the core unit and the kernel
// shards (__graft_entry__.build), linked into libedmp_hip.so.
#define EDMP_CHAIN_DEFINE 1
#include "common.h"
#include "params.h"
#include "chain.hip"
