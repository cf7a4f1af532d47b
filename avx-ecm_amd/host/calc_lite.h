/* calc_lite.h — see calc_lite.c */
#ifndef CALC_LITE_H
#define CALC_LITE_H
#include "mpl.h"
#ifdef __cplusplus
extern "C" {
#endif
/* evaluate expression s into out; 0 on success */
int calc_lite(mpl_t *out, const char *s);
#ifdef __cplusplus
}
#endif
#endif
