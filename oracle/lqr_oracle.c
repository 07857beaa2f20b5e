/*
 * lqr_oracle.c -- TEST INFRASTRUCTURE ONLY (see lqr_oracle.h).
 * Instantiates the restatement for float and double.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "lqr_oracle.h"

#define REAL float
#define SFX(name) name##_f32
#include "lqr_oracle_impl.inc"
#undef REAL
#undef SFX

#define REAL double
#define SFX(name) name##_f64
#include "lqr_oracle_impl.inc"
#undef REAL
#undef SFX

int lqr_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
