// bn_fit.hpp -- CPT fitting from pattern counts (reference bayesian/sampler.hpp:81-163).
#pragma once

#include <cstdint>

namespace bnmi {

constexpr int kFitLdsEntries = 4096;

struct FitArgs {
    int32_t n;
    const int32_t* k;
    const int32_t* in_ptr;
    const int32_t* in_idx;
    const int64_t* cpt_off;
    int64_t n_patterns;
    const uint8_t* patterns;            // [node][pattern]
    const unsigned long long* weights;  // [pattern] occurrence counts
    unsigned long long* counts;         // [cpt entries], zeroed by the host
    int64_t n_rows;
    const int32_t* row_node;            // [rows] node of each CPT row
    const int64_t* row_off;             // [rows] offset of each CPT row
    double* cpt_out;                    // [cpt entries], reference row order
};

int launch_fit(const FitArgs& a, void* stream);

}  // namespace bnmi
