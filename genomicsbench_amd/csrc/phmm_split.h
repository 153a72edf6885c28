// phmm_split.h - a hook between the two halves of phmm_launch (phmm_kernels.hip), for the host entry (capi_phmm.hip).
// Kept out of gbx_internal.h: the counter tables under profiles/ are stamped with a hash of each kernel file and gbx_internal.h
// (genomicsbench_amd/srchash.py), and this touches phmm only.
#pragma once
#include <functional>
#include "gbx_internal.h"

namespace gbx {

// The next phmm_launch of the calling thread calls `between` once, after it has queued the passes that group the pairs and lay the
// haplotype streams out - they read the pair lists, the length tables and the haplotypes, NOT the reads' bases and qualities - and
// before it queues the kernels that do: the host entry makes the stream wait for the upload of those there, so that the grouping
// runs under it.  A non-zero return aborts the launch with that status.  One launch only: the hook is taken when the launch starts.
void phmm_set_between(const std::function<int()> *between);

}  // namespace gbx
