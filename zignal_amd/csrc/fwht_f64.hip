// double-precision instantiation of the FWHT pass kernels (the reference's arithmetic type)
#include "zmls_internal.h"
#define ZT double
#define ZLAUNCH launch_pass_f64
#define ZPIPELINE launch_pipeline_f64
#include "fwht_kernels.inc"
