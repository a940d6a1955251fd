// ref_gpu_prelude.h -- force-included (-include) in front of the reference's HOST translation units when oracle/Makefile builds _ref/ref_gpu.
// TEST INFRASTRUCTURE ONLY.  It supplies nothing: it includes the HIP runtime header first and then renames ONE identifier for the rest of the
// translation unit.  Why: the reference has a class called `texture` in the global namespace (inc/texture.h:14), and so have the HIP runtime
// headers (hip/texture_types.h: the legacy texture-reference template, as CUDA's headers had before CUDA 12); a translation unit that sees both
// does not compile.  With the macro the reference's class is called dsrt_ref_texture_class in every such unit, consistently; no line of the
// reference is edited and the render kernel's unit (src/gpu_render.cu, which includes neither) is compiled without this file.
#pragma once
#include <hip/hip_runtime.h>
#define texture dsrt_ref_texture_class
