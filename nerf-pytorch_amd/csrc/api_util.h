// Shared by the extern "C" translation units (api.hip, render_abi.hip): error reporting and the buffer-tag registry.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stddef.h>
#include "../../include/nerf_hip.h"

namespace nerf_api {

inline thread_local char g_err[384] = "";

inline int fail_arg(const char* fn, const char* what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", fn, what);
    return NERF_E_BADARG;
}
inline int done(const char* fn, hipError_t e) {
    if (e == hipSuccess) return 0;
    snprintf(g_err, sizeof(g_err), "%s: HIP error %d (%s)", fn, (int)e, hipGetErrorString(e));
    return (int)e;
}

// ---- buffer tags.  The save buffer of a forward (`act`) and the delta buffer of a dgrad exist in five / three layouts
// (nerf_common.h); which one a buffer holds is decided by the entry point that WROTE it, and the entry point that reads it
// must agree.  The library remembers, per buffer address, what its own entry points last wrote there (host memory only:
// a small table, nothing on the device, no pointer is ever dereferenced) so that a mismatched pairing is refused with
// NERF_E_BADARG instead of computing garbage, and so that nerf_field_wgrad_phase(datapath = -1) can pick the datapath
// itself.  Buffers the library has not seen (copied, produced elsewhere) are not checked.
enum ActLayoutKind { ACT_ROWS_F32 = 0, ACT_TILE32_F32 = 1, ACT_TILE32_BF16 = 2, ACT_TILE16_F32 = 3, ACT_TILE16_BF16 = 4, ACT_TILE16_F16 = 5,
                     ACT_TILE16_F16X2 = 6 /* hi + lo words (two-word saves) */ };
enum DeltaKind { DELTA_ROWS_F32 = 0, DELTA_TILE32_F32 = 1, DELTA_TILE32_BF16 = 2, DELTA_TILE32_F16 = 3, DELTA_TILE32_F16X2 = 4 };
struct BufTag { int is_delta, kind, n_rays, n_samples; unsigned long seq; };

void tag_record(const void* buf, int is_delta, int kind, int n_rays, int n_samples);
bool tag_lookup(const void* buf, BufTag* out);
// datapath of nerf_field_wgrad_phase for an (act layout, delta kind) pair, or -1 if the pair does not exist
int datapath_of(int act_kind, int delta_kind);

}  // namespace nerf_api

#define REQUIRE(cond, what) do { if (!(cond)) return nerf_api::fail_arg(__func__, what); } while (0)
