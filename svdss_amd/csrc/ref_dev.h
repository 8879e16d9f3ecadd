// ref_dev.h -- what other parts of the library need of a svdss_ref_t (csrc/place.hip): the chromosomes in HBM.
#pragma once
#include <cstdint>

#include "../../include/svdss_hip.h"

struct SvdssRefView {
  int device = -1;
  const uint8_t* d_seq = nullptr;   // the chromosomes back to back, upper-case ASCII
  const int64_t* d_off = nullptr;   // n_chrom + 1 offsets
  int32_t n_chrom = 0;
};
SvdssRefView svdss_ref_view(const svdss_ref_t* ref);
