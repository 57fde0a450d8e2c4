// activations.h -- the reference's activate() (src/additionally.h:66-165) for the kernels' epilogues.
//
// Restated with C's promotion rules as the reference's scalar build evaluates them: the argument is a float, every
// literal with a decimal point is a double, `x*(x>0)` is float x int, exp() is the double function.  LINEAR and LEAKY
// are the hot cases and stay inline at the call sites (leaky = (float)(.1 * (double)x)); everything else comes here
// through one wave-uniform switch.  The -quantized convolution applies LEAKY only (y / 10) and leaves every other
// activation undone, exactly like forward_convolutional_layer_q (src/yolov2_forward_network_quantized.c:623-627).
// exp() on the device and glibc's exp() are both accurate to well under 1 ulp of double, so after the rounding to
// float results agree except on rare double-rounding ties (tests allow 1 float ulp on the transcendental ones).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/yolo2_hip.h"

namespace yl {

__device__ __forceinline__ float yl_activate(float x, int a)
{
    switch (a) {
    case YL_LINEAR: return x;                                                              // linear_activate   :84
    case YL_LEAKY: return (x > 0.f) ? x : (float)(.1 * (double)x);                         // leaky_activate    :91
    case YL_LOGISTIC: return (float)(1. / (1. + exp(-(double)x)));                         // logistic_activate :85
    case YL_LOGGY: return (float)(2. / (1. + exp(-(double)x)) - 1.);                       // loggy_activate    :86
    case YL_RELU: return __fmul_rn(x, (x > 0.f) ? 1.f : 0.f);                              // x*(x>0)           :87
    case YL_ELU:                                                                           // (x >= 0)*x + (x < 0)*(exp(x) - 1)   :88
        return (float)((double)__fmul_rn((x >= 0.f) ? 1.f : 0.f, x) + ((x < 0.f) ? 1. : 0.) * (exp((double)x) - 1.));
    case YL_RELIE: return (x > 0.f) ? x : (float)(.01 * (double)x);                        // relie_activate    :89
    case YL_RAMP: return (float)((double)__fmul_rn(x, (x > 0.f) ? 1.f : 0.f) + .1 * (double)x);      // x*(x>0) + .1*x    :90
    case YL_TANH: {                                                                        // (exp(2*x) - 1) / (exp(2*x) + 1), 2*x in float   :92
        const double e = exp((double)__fmul_rn(2.f, x));
        return (float)((e - 1.) / (e + 1.));
    }
    case YL_PLSE:                                                                          // plse_activate     :93-98
        if (x < -4.f) return (float)(.01 * (double)__fadd_rn(x, 4.f));
        if (x > 4.f) return (float)(.01 * (double)__fsub_rn(x, 4.f) + 1.);
        return (float)(.125 * (double)x + .5);
    case YL_STAIR: {                                                                       // stair_activate    :72-77
        const int n = (int)floor((double)x);
        if (n % 2 == 0) return (float)floor((double)x / 2.);
        return (float)((double)__fsub_rn(x, (float)n) + floor((double)x / 2.));
    }
    case YL_HARDTAN: return (x < -1.f) ? -1.f : ((x > 1.f) ? 1.f : x);                     // hardtan_activate  :78-83
    case YL_LHTAN:                                                                         // lhtan_activate    :100-105
        if (x < 0.f) return (float)(.001 * (double)x);
        if (x > 1.f) return (float)(.001 * (double)__fsub_rn(x, 1.f) + 1.);
        return x;
    default: return 0.f;                                                                   // activate(): `return 0` behind the switch
    }
}

// epilogue helper: the two hot cases inline, the rest through the switch
__device__ __forceinline__ float yl_act_epilogue(float v, int act)
{
    if (act == YL_LEAKY) return (v > 0.f) ? v : (float)(.1 * (double)v);
    if (act != YL_LINEAR) return yl_activate(v, act);
    return v;
}

}  // namespace yl
