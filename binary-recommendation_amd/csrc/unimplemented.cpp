// Entry points declared in include/binrec.h whose kernels are not written yet.
// They fail loudly (BR_ERR_UNSUPPORTED + message); there is no fallback.
#include "common.h"
#define BR_TODO(name, ...) extern "C" int name(__VA_ARGS__) { br::set_error(#name ": not implemented in this build"); return BR_ERR_UNSUPPORTED; }

BR_TODO(brInBatchSoftmaxLse, const float*, const float*, const void*, const void*, int, int64_t, int64_t, int, int64_t, float*, double*, brStream)
BR_TODO(brInBatchSoftmaxGrad, const float*, const float*, const void*, const void*, int, int64_t, int64_t, int, int64_t, const float*, float*, float*, brStream)
BR_TODO(brTopKRows, const float*, int64_t, int64_t, int, float*, int32_t*, brStream)
// --- temporary until mlp.hip lands ---
BR_TODO(brDenseForward, const float*, int64_t, const float*, const float*, float*, int64_t, int64_t, int, int, int, const float*, const float*, float, uint64_t, uint32_t, uint32_t, int64_t, double*, brStream)
BR_TODO(brBnFinalize, const double*, double, const float*, const float*, float, float, float*, float*, float*, float*, float*, float*, int, brStream)
BR_TODO(brBnInference, const float*, const float*, const float*, const float*, float, float*, float*, int, brStream)
BR_TODO(brDenseBackwardSlabs, int64_t, int, int)
BR_TODO(brDenseBackward, const float*, int64_t, const float*, int64_t, const float*, int64_t, const float*, int64_t, int, int, int, const float*, const float*, const float*, const double*, double, float, uint32_t, const float*, const float*, const float*, const float*, float, uint32_t, uint64_t, uint32_t, int64_t, float*, int64_t, float*, int, double*, brStream)
BR_TODO(brReduceSlabs, const float*, int, int64_t, float*, brStream)
BR_TODO(brHeadSlabs, int64_t)
BR_TODO(brNeumfHead, const float*, int64_t, const float*, const float*, const float*, const float*, int64_t, int, int, int, float, float*, float*, double*, float*, int64_t, float*, float*, int, brStream)
BR_TODO(brBceLogits, const float*, const float*, int64_t, float, float*, float*, double*, brStream)
