// What csrc/dp.hip needs to know about a handle (qnet.hip owns the type).  Not part of the C ABI: hidden visibility.
#pragma once
#include "../../include/idqn_hip.h"

struct IdqnDpView {
    float* grad;    // gradient arena: the first n_small floats are the small-leaf region + the caller's 64 reserved floats
    float* losses;  // the K per-head losses of the last step (wherever the caller keeps them)
    long n_small;
    int K, F, J;
};
__attribute__((visibility("hidden"))) int idqn_internal_dp_view(idqn_handle_t h, IdqnDpView* v);
