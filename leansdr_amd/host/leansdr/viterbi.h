// leansdr_amd/host/leansdr/viterbi.h — the trellis / viterbi_dec / bitpath templates of the reference's viterbi.h are one
// HIP kernel here (k_viterbi in leansdr_amd/csrc/viterbi.hip: a wavefront per tile, a lane per trellis state) behind
// viterbi_sync of leansdr/dvb.h; there is no host-side decoder.
#ifndef LEANSDR_AMD_VITERBI_H
#define LEANSDR_AMD_VITERBI_H
#include "leansdr/dvb.h"
#endif
