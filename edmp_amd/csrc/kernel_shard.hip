// kernel_shard.hip — one shard of the position-tile / whole-level kernel instances (kernel_instances.h), compiled with
// -DEDMP_SHARD=<n> next to the core translation unit (libedmp_hip.hip with -DEDMP_SHARDED) and linked into libedmp_hip.so.
#include "common.h"
#include "params.h"
#include "wide.hip"
#include "bf3.hip"
#include "level.hip"
#include "kernel_instances.h"

#ifndef EDMP_SHARD
#error "compile with -DEDMP_SHARD=<0..EDMP_KERNEL_SHARDS-1>"
#endif

namespace edmp {
// instantiate the entries whose shard number is EDMP_SHARD
#define EDMP_X(sh, K, MS, CG, GS, L, R) EDMP_IF_SHARD(sh, template int launch_wide_t<K, MS, CG, GS, L, R>(const RcbP&, hipStream_t);)
#define EDMP_W(sh, K, MS, CG, GS, L, R) EDMP_IF_SHARD(sh, template int launch_bf3_t<K, MS, CG, GS, L, R>(const RcbP&, hipStream_t);)
#define EDMP_Y(sh, M, C, L, SB, CIN) EDMP_IF_SHARD(sh, template int launch_level_t<M, C, L, SB, CIN>(const LevelP&, hipStream_t);)
#define EDMP_Z(sh, MA, CA, LA, CINA, MB, CB, LB, CINB, SB) EDMP_IF_SHARD(sh, template int launch_level2_t<MA, CA, LA, CINA, MB, CB, LB, CINB, SB>(const LevelP&, const LevelP&, hipStream_t);)
#define EDMP_IF_SHARD(sh, ...) EDMP_IF_SHARD_I(sh, __VA_ARGS__)
#define EDMP_IF_SHARD_I(sh, ...) EDMP_SHARD_##sh(__VA_ARGS__)
#define EDMP_SHARD_0(...)
#define EDMP_SHARD_1(...)
#define EDMP_SHARD_2(...)
#define EDMP_SHARD_3(...)
#define EDMP_SHARD_4(...)
#define EDMP_SHARD_5(...)
#define EDMP_SHARD_6(...)
#define EDMP_SHARD_7(...)
#define EDMP_SHARD_8(...)
#define EDMP_SHARD_9(...)
#define EDMP_SHARD_10(...)
#define EDMP_SHARD_11(...)
#define EDMP_SHARD_12(...)
#define EDMP_SHARD_13(...)
#define EDMP_SHARD_14(...)
#define EDMP_SHARD_15(...)
#define EDMP_SHARD_16(...)
#define EDMP_SHARD_17(...)
#define EDMP_SHARD_18(...)
#define EDMP_SHARD_19(...)
#define EDMP_SHARD_20(...)
#if EDMP_SHARD == 0
#undef EDMP_SHARD_0
#define EDMP_SHARD_0(...) __VA_ARGS__
#elif EDMP_SHARD == 1
#undef EDMP_SHARD_1
#define EDMP_SHARD_1(...) __VA_ARGS__
#elif EDMP_SHARD == 2
#undef EDMP_SHARD_2
#define EDMP_SHARD_2(...) __VA_ARGS__
#elif EDMP_SHARD == 3
#undef EDMP_SHARD_3
#define EDMP_SHARD_3(...) __VA_ARGS__
#elif EDMP_SHARD == 4
#undef EDMP_SHARD_4
#define EDMP_SHARD_4(...) __VA_ARGS__
#elif EDMP_SHARD == 5
#undef EDMP_SHARD_5
#define EDMP_SHARD_5(...) __VA_ARGS__
#elif EDMP_SHARD == 6
#undef EDMP_SHARD_6
#define EDMP_SHARD_6(...) __VA_ARGS__
#elif EDMP_SHARD == 7
#undef EDMP_SHARD_7
#define EDMP_SHARD_7(...) __VA_ARGS__
#elif EDMP_SHARD == 8
#undef EDMP_SHARD_8
#define EDMP_SHARD_8(...) __VA_ARGS__
#elif EDMP_SHARD == 9
#undef EDMP_SHARD_9
#define EDMP_SHARD_9(...) __VA_ARGS__
#elif EDMP_SHARD == 10
#undef EDMP_SHARD_10
#define EDMP_SHARD_10(...) __VA_ARGS__
#elif EDMP_SHARD == 11
#undef EDMP_SHARD_11
#define EDMP_SHARD_11(...) __VA_ARGS__
#elif EDMP_SHARD == 12
#undef EDMP_SHARD_12
#define EDMP_SHARD_12(...) __VA_ARGS__
#elif EDMP_SHARD == 13
#undef EDMP_SHARD_13
#define EDMP_SHARD_13(...) __VA_ARGS__
#elif EDMP_SHARD == 14
#undef EDMP_SHARD_14
#define EDMP_SHARD_14(...) __VA_ARGS__
#elif EDMP_SHARD == 15
#undef EDMP_SHARD_15
#define EDMP_SHARD_15(...) __VA_ARGS__
#elif EDMP_SHARD == 16
#undef EDMP_SHARD_16
#define EDMP_SHARD_16(...) __VA_ARGS__
#elif EDMP_SHARD == 17
#undef EDMP_SHARD_17
#define EDMP_SHARD_17(...) __VA_ARGS__
#elif EDMP_SHARD == 18
#undef EDMP_SHARD_18
#define EDMP_SHARD_18(...) __VA_ARGS__
#elif EDMP_SHARD == 19
#undef EDMP_SHARD_19
#define EDMP_SHARD_19(...) __VA_ARGS__
#elif EDMP_SHARD == 20
#undef EDMP_SHARD_20
#define EDMP_SHARD_20(...) __VA_ARGS__
#else
#error "EDMP_SHARD out of range"
#endif
EDMP_WIDE_INSTANCES(EDMP_X)
EDMP_BF3_INSTANCES(EDMP_W)
EDMP_LEVEL_INSTANCES(EDMP_Y)
EDMP_LEVEL2_INSTANCES(EDMP_Z)
}  // namespace edmp
