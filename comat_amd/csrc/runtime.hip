// runtime.hip — error reporting, ABI version and the tuning-option table of libcomat_hip.
#include "common.h"
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";

void comat_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int comat_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        comat_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return COMAT_ELAUNCH;
    }
    return COMAT_OK;
}

extern "C" int comat_abi_version(void) { return COMAT_ABI_VERSION; }
#ifndef COMAT_SRC_HASH
#define COMAT_SRC_HASH "unknown"
#endif
extern "C" const char* comat_build_id(void) { return COMAT_SRC_HASH; }
extern "C" const char* comat_last_error(void) { return g_err; }

static thread_local int g_last_gemm_kernel = -1;
void comat_note_gemm_kernel(int id) { g_last_gemm_kernel = id; }
extern "C" int comat_last_gemm_kernel(void) { return g_last_gemm_kernel; }

// ---- tuning options -------------------------------------------------------------------------------------------
// Kernel-selection switches (A/B runs, microbenchmarks, parity tests of every variant).  Each option takes its value
// from the environment variable COMAT_<NAME> the first time it is read (a launch has a budget of a few microseconds:
// no getenv per call) and can be overridden at run time with comat_set_option().  Results never depend on them
// beyond floating-point summation order (split counts, tile shapes).
namespace {
struct Opt {
    const char* name;
    const char* env;
    int dflt, value;
    bool have;
};
Opt g_opts[COMAT_N_OPTIONS] = {
    {"flash_trim", "COMAT_FLASH_TRIM", 1, 0, false},      // skip all-zero head-dim MFMA steps of the fused attention
    {"flash_tr", "COMAT_FLASH_TR", 1, 0, false},          // transposed LDS images for its k-major operand tiles: 0 never,
                                                          // 1 for head dims <= 64 (where it pays), 2 always
    {"gemm2", "COMAT_GEMM2", 1, 0, false},                // LDS-DMA pipelined GEMM / conv kernel (gemm2.hip)
    {"g2_cfg", "COMAT_G2_CFG", 0, 0, false},              // force its block tile: 1 128x128, 2 128x64, 3 256x128, 4 64x128
    {"g2_splits", "COMAT_G2_SPLITS", 0, 0, false},        // force its split-K count
    {"force_splits", "COMAT_FORCE_SPLITS", 0, 0, false},  // force the split-K count of the general 64x64 kernel
    {"norm_fused", "COMAT_NORM_FUSED", 5, 0, false},      // GroupNorm: ONE launch wherever a (sample, group) fits a workgroup's
                                                          // registers and that pays (HW <= 256); for the rest 3 = three launches
                                                          // (statistics, finalize, apply), 4 = two (the last-arriving statistics block
                                                          // finalises), 5 (default, round 6) = two (every apply block sums the
                                                          // partials in its prologue, one burst of loads; tensors up to 128^2 pixels).
                                                          // 0 = always three, 1 / 2 = always the two-launch forms.
                                                          // Measured: profiles/r06_af_gn_ticket_ab.txt, r06_ag_gn_fin_burst.txt; the
                                                          // cooperative one-launch form of round 5: r05_a_mb_gn_coop.txt
    {"gemm2_tt", "COMAT_GEMM2_TT", 1, 0, false},          // k-major x k-major GEMMs (weight gradients) on the pipelined
                                                          // kernel with hardware transpose reads
    {"flash_kt", "COMAT_FLASH_KT", 4, 0, false},          // fused attention (bf16), two 32-row tiles per iteration: 1 nowhere,
                                                          // 2 in the forward, 3 + dQ, 4 (default) + dK/dV for head dims <= 64,
                                                          // 5 + dK/dV up to head dim 96 (profiles/r03_n_mb_flash_kt_vgpr.txt)
    {"flash_merge", "COMAT_FLASH_MERGE", 1, 0, false},    // fused attention backward: dQ and dK/dV blocks in ONE launch (after a
                                                          // D = rowsum(dO . O) launch): 0 never, 1 (default) when both grids
                                                          // together hold <= 768 blocks, 2 always.  Same bits either way;
                                                          // 256^2 d=160: 70.6 -> 48.9 us, 2 x 8 x 4096^2 d=40 (1024 blocks):
                                                          // 395 -> 470 us (profiles/r03_m_mb_flash_merge.txt)
    {"g2_order", "COMAT_G2_ORDER", 2, 0, false},          // pipelined GEMM / conv: order of the output tiles inside an XCD's
                                                          // chunk: 0 row-block-major (rounds 1-3), 1 column-block-major,
                                                          // 2 (default) per problem by the L2-miss model
                                                          // (tools/xcd_traffic_model.py).  Same tiles, same arithmetic:
                                                          // bit-identical.  Round 4, call 1: the step's gemm2_kernel launches
                                                          // read 24.3 MB each instead of 34.6 (FETCH_SIZE), C2 125.3 -> 122.7 ms;
                                                          // per problem 0 - 8 % faster, never slower (profiles/r04_a_g2_diag.txt)
    {"flash_xcd", "COMAT_FLASH_XCD", 1, 0, false},        // fused attention: 1 = workgroups renumbered so that the blocks of one
                                                          // (batch, head) run on ONE XCD (its K / V - Q / dO in dK/dV - are
                                                          // fetched into one L2 instead of eight).  Bit-identical.  Round 4, call 1:
                                                          // reads per launch 29.1 -> 9.7 MB (forward), 58.9 -> 34.1 (dQ), 62.6 ->
                                                          // 26.1 (dK/dV); kernel times unchanged within 2 % (VALU-bound)
    {"gemm3", "COMAT_GEMM3", 1, 0, false},                // lean k-parallel-wave GEMM (gemm3.hip): 0 never, 1 (default) where its
                                                          // rule wants a problem (<= 32 rows, short contraction: the BLIP text
                                                          // decoder), 2 every eligible problem (tests, microbenchmarks)
    {"g3_cfg", "COMAT_G3_CFG", 0, 0, false},              // force its tile shape: 1 32x32 / 8 waves, 2 64x32 / 8, 3 32x64 / 8,
                                                          // 4 64x64 / 4, 5 64x64 / 8, 6 32x32 / 4, 7 64x32 / 4, 8 32x32 / 16, 9 64x32 / 16
    {"flash_qs", "COMAT_FLASH_QS", 256, 0, false},        // fused attention backward with few key blocks (cross-attention): the query
                                                          // loop of dK / dV is cut into ranges until the grid holds about this many
                                                          // blocks (each range writes an fp32 partial that flash_kv_reduce sums).
                                                          // 256 since round 6 (was 512): profiles/r06_ah_mb_flash_qs.txt
};
}  // namespace

int comat_option(int id) {
    Opt& o = g_opts[id];
    if (!o.have) {
        const char* e = getenv(o.env);
        o.value = e ? atoi(e) : o.dflt;
        o.have = true;
    }
    return o.value;
}

extern "C" int comat_set_option(const char* name, int32_t value) {
    COMAT_REQUIRE(name != nullptr, "comat_set_option: null name");
    for (int i = 0; i < COMAT_N_OPTIONS; ++i)
        if (strcmp(name, g_opts[i].name) == 0) {
            g_opts[i].value = value;
            g_opts[i].have = true;
            return COMAT_OK;
        }
    comat_set_error("comat_set_option: unknown option '%s'", name);
    return COMAT_EINVAL;
}
