// uvo_experimental.h -- entry points of libuvo_hip.so that are NOT part of the drop-in C ABI (include/uvo_hip.h): measurement hooks the
// reference has no counterpart for.  Exported so that tests and bench.py can reach them through ctypes; a caller of the library
// should not.
#pragma once
#include "../../include/uvo_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Two-pair launch sets (pairs = 2; 1 restores the default): uvo_stereo_submit holds every first pair of two back until the next one
 * arrives and queues the device work of both -- lanes i and i + 1 of the pipeline -- as ONE launch per kernel (upright SURF with four
 * octaves; any other configuration, the synchronous step and the init pairs go alone).  A pair still waiting for its partner is queued
 * by the uvo_stereo_collect that asks for it, so every submit / collect order works; results are those of single launches, pair for
 * pair (tests/test_gpu_batch.py).  Not while pairs are in flight.  Built in round 4 and measured 22-38 % SLOWER than single launches
 * (DESIGN.md section 4): kept for that measurement only, off by default. */
uvo_status uvo_stereo_set_batch(uvo_ctx* c, int pairs);
#ifdef __cplusplus
}
#endif
