// Host-only sanitizer builds of the C-ABI library (tools/asan_host.sh): the few device-side helpers whose VALUE the host
// logic depends on.  Every other symbol the device translation units (*.hip) define -- launchers, the extern "C" entry
// points of the regular solver -- is stubbed AUTOMATICALLY by the script (a function that returns VRT_ENODEVICE), from the
// undefined symbols of the host objects and the declarations of include/voronoirt.h: nothing here goes stale when a
// launcher is added or its signature changes.  GPU sanitizers are not available on this pool.
#include "vrt_internal.h"

namespace vrt {
bool patch_shape_exists(int K, int Q, int NT)
{
    return (K == 1 || K == 2) && (Q == 1 || Q == 2) && (NT == 256 || NT == 512 || NT == 1024);
}
int64_t steps_max_layer(bool f32) { return f32 ? 18432 : 12288; }
int chain_ctrl_words() { return 8 * 32 + 4; }
bool patch_chain_possible(const vrt_plan *, int, bool) { return false; }
bool patch_chain_dataflag(const vrt_plan *, int, bool) { return false; }
int patch_chain_check(vrt_plan *) { return VRT_OK; }
}  // namespace vrt
