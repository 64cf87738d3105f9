/* mmx_hip_lab.h - the two extra entry points of libmmx_hip_lab.so, the MEASUREMENT build of the sources of libmmx_hip.so
 * (`make -C minimax-speech_amd/csrc lab`, -DMMX_LAB=1).  The lab library exports everything include/mmx_hip.h declares, with the
 * same signatures, plus the two switches below; its kernels carry shader-clock stamps (s_memtime) at their phase boundaries
 * and read the stamp buffer's address from a __device__ global.  The product library has neither: tools/decode_lab.py and
 * tools/tail_lab.py select the lab build with MMX_LIB=minimax-speech_amd/lib/libmmx_hip_lab.so.  Not part of the drop-in boundary. */
#ifndef MMX_HIP_LAB_H
#define MMX_HIP_LAB_H
#ifdef __cplusplus
extern "C" {
#endif

/* buf != NULL: every later mmx_skinny2 launch on the current device writes uint64 [workgroups][8 waves][8] stamps (entry,
 * loads issued, loads landed, MFMAs done, barrier, ticket, epilogue done); NULL switches it off.  Call between launches. */
int mmx_lab_skinny_stamps(void* buf);

/* buf != NULL: every later mmx_est_tail launch on the current device writes uint64 [workgroups][waves][64] stamps at its
 * stage boundaries; NULL switches it off.  Call between launches. */
int mmx_lab_tail_stamps(void* buf);

#ifdef __cplusplus
}
#endif
#endif
