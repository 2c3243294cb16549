// Second translation unit of gemm.hip: the experimental kernel variants (see the note above
// comat_gemm_launch_variant in gemm.hip).  Nothing else lives here.
#define COMAT_GEMM_EXP_TU
#include "gemm.hip"
