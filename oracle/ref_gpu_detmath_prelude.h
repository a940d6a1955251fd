// ref_gpu_detmath_prelude.h -- force-included in front of the reference's RENDER KERNEL translation unit (src/gpu_render.cu, translated by hipify-perl) when
// oracle/Makefile builds _ref/ref_gpu_detmath.  TEST INFRASTRUCTURE ONLY.
//
// It changes which function three names resolve to, and nothing else: the reference's kernel calls cosf / sinf (src/gpu_render.cu:104-106, 157-158) and powf
// (:211, :1019-1021) from whatever math library its compiler links -- CUDA's libdevice on the author's machine, ROCm's ocml in _ref/ref_gpu, and no library that
// also exists on a CPU.  Here the three names are mapped onto include/dsrt_detmath.h, the deterministic versions (IEEE + - * / sqrt and explicit fma only) that the
// CPU oracle and the product's default mode use.  The runtime and <cmath> are included FIRST, so that their own declarations are seen under their own names and the
// macros only touch the reference's call sites.  Not a line of the reference is edited; every other operation of the kernel is compiled exactly as in _ref/ref_gpu.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <math.h>
#include "include/dsrt_detmath.h"
#define cosf dsrt_cosf
#define sinf dsrt_sinf
#define powf dsrt_powf
