// call_host.h -- options of the `call` and `smooth` sub-commands (config.hpp:68-103 defaults).
#pragma once
#include <string>

struct CallOptions {
  std::string reference, bam, sfs;
  int threads = 4;
  unsigned min_cluster_weight = 2;   // (unsigned as in the reference, config.hpp:92-96: a negative value on the command
  unsigned min_sv_length = 25;       //  line is a huge one)
  unsigned min_mapq = 20;
  bool useht = true;
  float min_ratio = 0.97f;
  float accp = 0.98f;          // smooth only
  bool verbose = false;         // stage timings on stderr
  int gpus = 1;                 // --gpus N: POA / realignment batches shard by sub-cluster index
  std::string poa;             // --poa <FILE>: consensus alignments as SAM (caller.cpp:65-75)
  std::string clusters;        // --clusters <FILE>: the filled clusters (clusterer.cpp:613-626)
  bool clipped = false;        // --clipped: imprecise SVs from soft-clipped alignments (clipper.cpp; EXPERIMENTAL)
};

int main_call(const CallOptions& o);
int main_smooth(const CallOptions& o);
