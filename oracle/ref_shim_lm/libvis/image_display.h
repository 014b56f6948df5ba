// Stand-in for <libvis/image_display.h> -- TEST INFRASTRUCTURE ONLY (oracle/_ref build): no window
#ifndef CBA_REF_SHIM_LM_IMAGE_DISPLAY_
#define CBA_REF_SHIM_LM_IMAGE_DISPLAY_
#include "libvis/image.h"
#endif
