// Device-side launchers stubbed out for the HOST-ONLY AddressSanitizer / UBSan build of the C-ABI
// library (tools/asan_host.sh): the host logic under test (grid preparation, schedules, thread
// assignment, argument checking) never reaches them on a device = -1 handle.  GPU sanitizers are
// not available on this pool, so only the CPU side is instrumented.
#include "vrt_internal.h"

namespace vrt {
static int nodev() { return fail(VRT_ENODEVICE, "host-only sanitizer build: no device code"); }
int launch_delaunay_lines(vrt_grid *) { return nodev(); }
int launch_upwind_table(vrt_plan *, int) { return nodev(); }
int launch_boundary(vrt_plan *, const SweepArgs &, const void *, const void *, hipStream_t) { return nodev(); }
int launch_sweep_levels(vrt_plan *, const SweepArgs &, hipStream_t, int64_t *) { return nodev(); }
int launch_reduce_J(vrt_plan *, const SweepArgs &, const double *, void *, int64_t, hipStream_t) { return nodev(); }
int launch_copy_I_out(vrt_plan *, const SweepArgs &, void *, int64_t, hipStream_t) { return nodev(); }
int launch_lambda_update(int64_t, int64_t, int64_t, const double *, const double *, const double *, const double *,
                         double *, unsigned long long *, hipStream_t) { return nodev(); }
int launch_permute_table(vrt_plan *, int, const uint32_t *) { return nodev(); }
int launch_sorted_tables(vrt_plan *, int) { return nodev(); }
int launch_gpos(vrt_plan *, int) { return nodev(); }
int execute_tiles(vrt_plan *, int64_t, int64_t, const void *, const void *, int, const void *, const void *,
                  const double *, void *, void *, hipStream_t, bool) { return nodev(); }
int alpha_to_native(vrt_plan *, int64_t, int64_t, const double *, double *, hipStream_t) { return nodev(); }
int64_t steps_max_layer(bool f32) { return f32 ? 18432 : 12288; }
int launch_line_opacity(vrt_plan *, int64_t, const double *, double, double, const double *, const double *, const double *,
                        const double *, const double *, double *, hipStream_t) { return nodev(); }
int launch_rates_populations(vrt_grid *, int64_t, int64_t, const int64_t *, const double *, const double *, double, double,
                             const double *, const double *, double, const double *, const double *, double, double, double,
                             const double *, const double *, double *, double *, hipStream_t) { return nodev(); }
}  // namespace vrt

using namespace vrt;
extern "C" {
int vrt_regular_create(int64_t, int64_t, int64_t, const double *, const double *, const double *, int, vrt_regular **) { return nodev(); }
void vrt_regular_destroy(vrt_regular *) {}
int vrt_regular_execute_dev(vrt_regular *, int64_t, const double *, const int *, const double *, int64_t, const double *,
                            int64_t, int64_t, const double *, int, double *, void *) { return nodev(); }
int vrt_regular_last_solve_ms(const vrt_regular *, double *) { return nodev(); }
int vrt_short_characteristics(int64_t, int64_t, int64_t, const double *, const double *, const double *, int64_t,
                              const double *, const int *, const double *, int64_t, const double *, int64_t,
                              const double *, int, int, double *) { return nodev(); }
}
