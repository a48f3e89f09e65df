// leansdr_amd/host/leansdr/rs.h — the Reed-Solomon (204,188) arithmetic of the reference's rs.h (gf2x_p, rs_engine) runs
// on the GPU here: k_rs_decode / k_rs_encode (leansdr_amd/csrc/fec.hip, tx.hip) behind rs_decoder<u8,0> / rs_encoder of
// leansdr/dvb.h.  Host code that wants the field tables (exp/log over 0x11d, the degree-16 generator) gets them from the
// C ABI; nothing on the host computes with them.
#ifndef LEANSDR_AMD_RS_H
#define LEANSDR_AMD_RS_H
#include "leansdr/dvb.h"

namespace leansdr {
struct rs_tables {            // GF(2^8), x^8+x^4+x^3+x^2+1, alpha = 2 (rs.h:47-112)
  u8 exp[512], log[256], G[17];
  rs_tables() { lsdr_rs_tables(exp, log, G); }
};
}  // namespace leansdr
#endif
