#include "rt_common.h"
extern "C" int rt_fields(rt_scene *, rt_candidates *, const double *, const double *, int64_t, const int32_t *,
                         const int32_t *, const int32_t *, const double *, int64_t, const double *, int32_t,
                         const rt_antenna_params *, double *, double *, double *, double *, int32_t *, double *, int) {
    RT_FAIL("rt_fields: not built yet");
}
