// filter_e1.hip — the k_filter instantiations for max_err = 1 (see filter_tu.hip.h)
#define SCT_FILTER_E 1
#include "filter_tu.hip.h"
