// snmf_tu_multi.hip -- the one-process multi-device entry (snmf_multi_*), see snmf_multi.h (snmf_internal.h).
#include "snmf_internal.h"

// ---- multi-GPU entry behind the C ABI (one process, several devices) --------------------------------------------
#include "snmf_multi.h"

