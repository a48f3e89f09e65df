// leansdr_amd/host/leansdr/gui.h — the X11 scopes (cscope, spectrumscope, slowmultiscope …) are outside the MI355X hot
// path; leandvb only uses them under -DGUI (leandvb.cc:258-268 and the other `#ifdef GUI` sections).  Built without GUI, a
// graph builder needs nothing from this header but its existence; window_placement lives in framework.h.
#ifndef LEANSDR_AMD_GUI_H
#define LEANSDR_AMD_GUI_H
#include "leansdr/framework.h"
#ifdef GUI
#error "leansdr_amd has no GUI blocks: build without -DGUI"
#endif
#endif
