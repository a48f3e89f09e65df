// leansdr_amd/host/leansdr/hdlc.h — HDLC framing (leandvb --hdlc) is OUTSIDE the MI355X hot path (the
// reference's hdlc.h).  A graph builder written for the reference still names the two blocks, so they exist with the
// reference's constructor signatures and public members — and refuse to be built into a graph.
#ifndef LEANSDR_AMD_HDLC_H
#define LEANSDR_AMD_HDLC_H
#include "leansdr/framework.h"

namespace leansdr {

struct etr192_descrambler : runnable {
  etr192_descrambler(scheduler *sch, pipebuf<u8> &, pipebuf<u8> &) : runnable(sch, "etr192_dec") {
    fail("--hdlc: HDLC framing is not part of the MI355X hot path (use the reference's leandvb)");
  }
};

struct hdlc_sync : runnable {
  int resync_period;
  bool header16;
  hdlc_sync(scheduler *sch, pipebuf<u8> &, pipebuf<u8> &, int /*minframesize*/, int /*maxframesize*/,
            pipebuf<int> * = NULL, pipebuf<int> * = NULL, pipebuf<int> * = NULL, pipebuf<int> * = NULL)
      : runnable(sch, "hdlc_sync"), resync_period(32), header16(false) {
    fail("--hdlc: HDLC framing is not part of the MI355X hot path (use the reference's leandvb)");
  }
};

}  // namespace leansdr
#endif
