/* lcgs_oracle_bwd.c -- CPU oracle for the build-defined backward pass.  TEST INFRASTRUCTURE ONLY.
 * (filled in below; see lcgs_oracle.h) */
#include "lcgs_oracle.h"
