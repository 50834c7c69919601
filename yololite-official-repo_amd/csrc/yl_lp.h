// Symbol suffix of the reduced-precision builds of the conv translation units (csrc/build.py compiles yl_conv.hip,
// yl_convc.hip and yl_stemblock.hip three times: fp32, -DYL_BF16=1 (bf16 operands) and -DYL_BF16=1 -DYL_F16=1 (fp16 operands)).
#pragma once
#if defined(YL_F16S) && YL_F16S
#define YL_LP_NAME(n) n##_f16s      /* fourth compilation: fp16 operands AND fp16 activation tensors in HBM */
#elif defined(YL_F16) && YL_F16
#define YL_LP_NAME(n) n##_f16
#else
#define YL_LP_NAME(n) n##_bf16
#endif
