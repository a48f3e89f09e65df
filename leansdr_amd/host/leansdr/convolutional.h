// leansdr_amd/host/leansdr/convolutional.h — deconvol_poly2 / dvb_deconvol_sync (--hs) and the convolutional encoder of
// the reference's convolutional.h are HIP kernels here (leansdr_amd/csrc/hs.hip, tx.hip) behind dvb_deconvol_sync_hard and
// dvb_convol of leansdr/dvb.h.
#ifndef LEANSDR_AMD_CONVOLUTIONAL_H
#define LEANSDR_AMD_CONVOLUTIONAL_H
#include "leansdr/dvb.h"
#endif
