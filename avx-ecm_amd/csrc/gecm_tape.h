/* gecm_tape.h — stage-1 op tape shared by the host tape compiler and the device interpreter.
 *
 * One byte per event of the reference's stage-1 control flow (ecm_stage1 ecm.c:1806-1854 driving
 * prac ecm.c:565-884).  The chain depends only on B1, never on N or sigma, so it is built once
 * on the host and replayed by every lane of every wave on every GPU.
 *
 *   GECM_OP_PRAC_BEGIN  B = C = A; A = 2A           (ecm.c:603-613; also used for the 2-power
 *                                                    doublings of ecm.c:1815-1822, where B, C are
 *                                                    don't-care)
 *   GECM_OP_PRAC_END    A = A + B, difference C     (ecm.c:868-873)
 *   GECM_OP_STEP | [GECM_OP_SWAP] | rule            one iteration of the while(d != e) loop
 *                                                    (ecm.c:615-866): optional swap of A,B
 *                                                    (ecm.c:617-630) then rule 3, 4, 5 or 9.
 */
#ifndef GECM_TAPE_H
#define GECM_TAPE_H

#define GECM_OP_NOP         0u
#define GECM_OP_PRAC_BEGIN  1u
#define GECM_OP_PRAC_END    2u
#define GECM_OP_STEP        8u
#define GECM_OP_SWAP        4u
#define GECM_OP_RULE_MASK   3u
#define GECM_OP_RULE3       0u
#define GECM_OP_RULE4       1u
#define GECM_OP_RULE5       2u
#define GECM_OP_RULE9       3u

#endif
