// rld0.h -- ropebwt3's `.fmd` (rld0, magic "RLD\3") export / import; see rld0.cpp.
#pragma once
#include <stdint.h>

#include <vector>

struct svdss_index;

bool rld0_is_fmd(const char* path);
// BWT symbols 0..5 -> file; SVDSS_OK / SVDSS_EINVAL / SVDSS_EIO
int rld0_write(const char* path, const uint8_t* bwt, int64_t n);
// the header's occurrence count of every symbol (cheap: 72 bytes read)
int rld0_header_counts(const char* path, uint64_t mcnt_out[6]);
// file -> BWT symbols
int rld0_read(const char* path, std::vector<uint8_t>& bwt);
// rank blocks, acc and '$' rows of a BWT into *ix (text and suffix array stay empty)
int svdss_blocks_from_bwt(const uint8_t* bwt, int64_t n, int threads, svdss_index* ix);
// the strings of the collection (LF walks from the sentinel rows), in sentinel order
int rld0_strings_of_bwt(const uint8_t* bwt, int64_t n, int threads, std::vector<std::vector<uint8_t>>& out);
// indices of one string of every reverse-complement pair; SVDSS_EIO if the collection is not closed under it
int rld0_pick_strands(std::vector<std::vector<uint8_t>>& strings, std::vector<int64_t>& picked);
