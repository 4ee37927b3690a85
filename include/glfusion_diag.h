/* Diagnostic entry points of lib/libglfusion_diag.so -- measurement aids of bench.py, NOT part of the product library
 * (libglfusion_hip.so, include/glfusion.h) and not needed to run the path.  Same conventions: C ABI, enqueue-only on a
 * hipStream_t, 0 on success, negative status otherwise (-5 null argument, -1 bad shape, -4 launch failure). */
#ifndef GLFUSION_DIAG_H
#define GLFUSION_DIAG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* Measurement aid (bench.py's roofline.power_limited_peak_measured; not on the path): `blocks` workgroups of 8 wavefronts
 * each issue 4 * iters back-to-back v_mfma_f32_32x32x16_f16 on register operands derived from `seed` (0 = all ones, else
 * pseudo-random halves in [-0.5, 0.5)) with no memory traffic, then store one float per thread to out[blocks * 512].
 * FLOPs of a launch = blocks * 8 * iters * 4 * 32768.  With random operands the matrix cores of an MI355X are power-limited
 * well below the 2.5 PFLOP/s a constant-operand run reaches; timing this launch on the box a bench runs on says how far. */
int glf_probe_mfma_f16(float* out, int blocks, int iters, uint32_t seed, void* stream);
#ifdef __cplusplus
}
#endif
#endif
