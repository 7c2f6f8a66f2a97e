// Guided-image-filter family and steerable-filter weights (placeholder until the kernels land).
#include "common.h"

extern "C" int mdisp_guided_filter(mdisp_ctx*, const mdisp_image*, const mdisp_image*, int, double, int,
                                   const mdisp_image*, int, double, mdisp_image*, mdisp_image*)
{ mdisp_set_error("guided filter kernels are not built yet"); return MDISP_E_UNSUPPORTED; }

extern "C" int mdisp_steer_filter(mdisp_ctx*, const mdisp_image*, mdisp_image*)
{ mdisp_set_error("steer filter kernel is not built yet"); return MDISP_E_UNSUPPORTED; }

extern "C" int mdisp_asw_gifs(mdisp_ctx*, const mdisp_image*, const mdisp_image*, const mdisp_param*, double, int, int,
                              double, double, int, const float*, mdisp_image*)
{ mdisp_set_error("asw_gifs is not built yet"); return MDISP_E_UNSUPPORTED; }
