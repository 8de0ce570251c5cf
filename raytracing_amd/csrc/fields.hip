// placeholder translation unit: the field kernels are added in fields.hip proper
#include "rt_common.h"
