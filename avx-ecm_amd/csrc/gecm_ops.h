/* gecm_ops.h — the operator codes of the test-level L0 kernel, shared by the device layer (gecm_dev.h) and the kernels
 * (gecm_kernels.hip).  Kept apart from gecm_dev.h so that a change of the host <-> device-layer interface does not
 * recompile thirty kernel objects. */
#ifndef GECM_OPS_H
#define GECM_OPS_H
enum { GECM_L0_MUL = 0, GECM_L0_SQR = 1, GECM_L0_ADD = 2, GECM_L0_SUB = 3, GECM_L0_ADDSUB = 4 };
#endif
