// The optional environment switches of DESIGN.md 7b, read once per process.  Kept in a translation unit of its own so
// that both libmtmc_mpn.so and the kernel laboratory (lab/, libmtmc_lab.so) link the same reader.
#include <stdlib.h>

#include "kernels.h"

namespace mtmc {

static Knobs read_knobs() {
  Knobs k;
  auto on = [](const char* name) { return getenv(name) != nullptr; };
  auto num = [](const char* name, long long dflt) { const char* v = getenv(name); return v ? atoll(v) : dflt; };
  k.pass_c_walk = on("MTMC_PASS_C_WALK");
  k.pass_c_small_min = num("MTMC_PASS_C_SMALL_MIN", 32768);
  k.staged_xr = (int)num("MTMC_STAGED_XR", 0);
  k.presplit_rows = (int)num("MTMC_PRESPLIT_ROWS", 0);
  k.pass_c_general = on("MTMC_PASS_C_GENERAL");
  k.pass_c_span = (int)num("MTMC_PASS_C_SPAN", 0);
  // 104 registers -> 4 blocks per CU are resident; twice that, so that CUs that finish early get more (config 4: 72.6 / 73.0 us
  // with 1024 / 2048 blocks, config 5: 656 / 626 us)
  k.pass_c_blocks = (int)num("MTMC_PASS_C_BLOCKS", 256 * 8);
  if (k.pass_c_blocks < 1) k.pass_c_blocks = 256 * 8;
  k.gemm_fp32 = on("MTMC_GEMM_FP32");
  k.gemm_no_f16 = on("MTMC_GEMM_NO_F16");
  k.gemm_no_presplit = on("MTMC_GEMM_NO_PRESPLIT");
  k.gemm_no_staged = on("MTMC_GEMM_NO_STAGED");
  k.l0_pipeline = (int)num("MTMC_L0_PIPELINE", 1);
  k.no_col_blocks = on("MTMC_NO_COL_BLOCKS");
  k.col_blocks = (int)num("MTMC_COL_BLOCKS", 0);
  k.gemm_no_few = on("MTMC_GEMM_NO_FEW");
  k.few_rows_max = (int)num("MTMC_FEW_ROWS_MAX", 1536);    // measured crossover: tools/few_crossover.py, profiles/r05_few_crossover.txt
  k.few_wave_rb = (int)num("MTMC_FEW_WAVE_RB", 0);
  k.few_l0_map = (int)num("MTMC_FEW_L0_MAP", 0);
  return k;
}

const Knobs& knobs() {
  static const Knobs k = read_knobs();
  return k;
}

}  // namespace mtmc
