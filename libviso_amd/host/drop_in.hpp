// drop_in.hpp — C entry point of the literal per-call loop (viso::sequence_odometry_per_call, viso.hpp) for callers that
// hold the frames as plain arrays (bench.py's `drop_in_per_call` leg, tests/test_gpu_drop_in.py).  The loop itself is
// C++: the patched reference's sequence_odometry (src/viso.cpp:1205-1327) calling the plain C-ABI once per reference
// function per frame.
#pragma once
#include <cstdint>

#include "../../include/viso_hip.h"

extern "C" {
// kp [nf][2][cap][2] float, desc [nf][2][cap][dlen] float, n [nf][2] int32: the layout of viso_batch_upload.
// F[9], prm: what sequence_odometry derives from P1 / P2 (src/viso.cpp:1176-1187).  first_frame: RANSAC stream key of frame 0.
// Outputs (any may be NULL): rec8 [nf][8] doubles = tr[6], ok, n_inl per frame (frame 0: zeros);
// matches [3][nf][cap][3] int32 + match_n [3][nf] (which = 0 stereo, 1 temporal left, 2 temporal right, as
// viso_batch_get_matches); n_circle [nf]; call_us [VISO_PLAIN_N][2] = calls, wall microseconds inside the C++ wrappers;
// loop_s[3] = wall seconds of the loop, of the copyTo carry-over inside it, and 0.
// Returns the number of frames processed, or a negative VISO_ERR_* (text in viso_host_last_error()).
int viso_host_drop_in_run(const float* kp, const float* desc, const int32_t* n, int nf, int cap, int dlen,
                          const double F[9], const viso_param* prm, uint64_t ransac_seed, uint64_t first_frame,
                          double* rec8, int32_t* matches, int32_t* match_n, int32_t* n_circle, double* call_us,
                          double* loop_s);
}
