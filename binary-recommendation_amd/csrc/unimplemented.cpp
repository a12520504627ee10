// Entry points declared in include/binrec.h whose kernels are not written yet.
// They fail loudly (BR_ERR_UNSUPPORTED + message); there is no fallback.
#include "common.h"
#define BR_TODO(name, ...) extern "C" int name(__VA_ARGS__) { br::set_error(#name ": not implemented in this build"); return BR_ERR_UNSUPPORTED; }

BR_TODO(brInBatchSoftmaxLse, const float*, const float*, const void*, const void*, int, int64_t, int64_t, int, int64_t, float*, double*, brStream)
BR_TODO(brInBatchSoftmaxGrad, const float*, const float*, const void*, const void*, int, int64_t, int64_t, int, int64_t, const float*, float*, float*, brStream)
BR_TODO(brTopKRows, const float*, int64_t, int64_t, int, float*, int32_t*, brStream)
