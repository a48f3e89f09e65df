// leansdr_amd/host/leansdr/iess.h — IESS-308 scrambling belongs to the HDLC branch of leandvb, which is outside the
// MI355X hot path.  The header exists because graph builders written for the reference include it.
#ifndef LEANSDR_AMD_IESS_H
#define LEANSDR_AMD_IESS_H
#include "leansdr/hdlc.h"
#endif
