// single-precision instantiation of the FWHT pass kernels (optional fp32 mode)
#include "zmls_internal.h"
#define ZT float
#define ZLAUNCH launch_pass_f32
#define ZPIPELINE launch_pipeline_f32
#include "fwht_kernels.inc"
