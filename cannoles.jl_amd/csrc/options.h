// options.h — every plan / execution switch of the library, internal.
//
// The PUBLIC structure (include/cannoles_hip.h: cnl_options) carries the dozen switches a caller or a test may reasonably want
// — plan kind, batch threshold, verbosity, which kernel families and executions are allowed — plus a `tuning` string of
// "key=value" pairs for everything else below (measurement tools, ablations, the randomised option fuzz).  Inside the library
// there is only this structure; nothing here is read from the environment.
//
// Measured-slower experiments stay reachable through `tuning` only (device_ladder_fused = 1, split_batch = 2); switches that had
// become no-ops (staged_large_fronts) or whose two settings measured equal (band_wide_pieces) were removed in round 6.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>

namespace cnl {

// name, default.  (plan_kind: CNL_PLAN_AUTO = 0)
#define CNL_TUNING_INT_FIELDS(X)                                                                                              \
  X(plan_kind, 0)            /* CNL_PLAN_*                                                                                  */ \
  X(order_mode, -1)          /* -1 auto (cost model over all candidates), 0 canonical, 1 nested dissection, 2 minimum degree */ \
  X(nd_leaf, 0)              /* nested-dissection leaf size, 0 = sweep                                                      */ \
  X(relax, -1)               /* relaxed-amalgamation budget (explicit zeros per merged column), -1 = default                */ \
  X(task_cap, 0)             /* fronts per bottom task of a latency plan, 0 = default                                       */ \
  X(multipliers_early, 1)    /* candidates with every multiplier right behind the last variable it touches are considered   */ \
  X(condense, 1)             /* static condensation of the -I block                                                         */ \
  X(direct_records, 1)       /* the register-front kernel condenses on the fly (no separate condense pass)                  */ \
  X(register_front, 1)       /* the register-front kernel may serve the plan (fronts of order <= 64)                        */ \
  X(dense_backend, 1)        /* dense residual blocks go to the dense backend                                               */ \
  X(general_dense, 1)        /* small batches of irregular plans with fronts > 64 as ONE dense matrix; 2: wherever possible */ \
  X(staged, 1)               /* latency plans run their first attempt stage by stage                                        */ \
  X(dataflow, 1)             /* smallest batches run all tasks in one launch per phase, waiting on device counters          */ \
  X(dataflow_waves, 1024)    /* at most this many wavefronts run in dataflow fashion                                        */ \
  X(dataflow_spin_limit, 1 << 22) /* polls before a dataflow wait gives up (counted: cnl_dataflow_timeouts; the call is redone) */ \
  X(waves_per_block, 0)      /* register-front kernel: wavefronts per workgroup, 0 = default (1)                            */ \
  X(v1_tpp, -1) X(v1_ppb, -1) X(v1_lds, -1) /* general kernel: threads per problem, problems per workgroup, work area in LDS */ \
  X(v1_solve, 0)             /* cnl_solve always runs on the general kernel                                                 */ \
  X(lds_pad, 1)              /* per-problem LDS areas 32 banks apart                                                        */ \
  X(ubig, 17)                /* update matrices of order above this live in global scratch                                  */ \
  X(wait_thr, 2)             /* ... and those that wait for more than this many fronts                                      */ \
  X(dense_graph, 1)          /* the dense backend replays its launch sequence as a hipGraph                                 */ \
  X(dense_syrk_wgs, 0)       /* workgroups of the J'WJ kernel, 0 = default                                                  */ \
  X(verbose, 0)              /* log plan decisions on stderr                                                                */ \
  X(multi_share_plan, 1)     /* cnl_multi_create analyses the pattern once for all shards of equal plan kind                */ \
  X(row_products, 1)         /* condensation products of small fronts organised per residual row (plan.h, RF_ROWS)          */ \
  X(split_batch, 1)          /* batches between one and two wavefronts per SIMD: 1 two halves on the chain, 2 chain || single stream (slower), 0 single stream */ \
  X(lean_kernel, 1)          /* plans of fast-class row-form fronts run the kernels' instantiation without the cold paths   */ \
  X(rows_in_backward, 1)     /* the lean kernel recovers the residual components in its backward sweep (no post-pass)       */ \
  X(dense_panel_blocks, 1)   /* dense backend, dn_panel2: 0 never, 1 while the step is latency-bound, 2 always              */ \
  X(host_ladder, 1)          /* host-pointer newton_system drives the rho ladder from the host where no device ladder runs  */ \
  X(device_ladder, 1)        /* staged handles climb the rho ladder inside one fused launch                                 */ \
  X(device_ladder_fused, 0)  /* that launch also makes the first attempt (measured slower: 0.140 against 0.118 ms)          */ \
  X(band_form, 1)            /* fast fronts with band-structured pivot rows skip the structurally zero row updates          */ \
  X(split_tail, 1)           /* the remainder of a batch above a machine-filling one runs on a handle of its own            */ \
  X(band_kernel, 1)          /* band-structured throughput handles run on the band kernels (band.h); 2: chain in one part   */ \
  X(band_problems_per_group, 0) /* band kernels: problems per workgroup (8, 16, 32); 0 = by batch                           */ \
  X(band_movers, 0)          /* EXPERIMENT builds only (-DCNL_EXPERIMENT=1 -DBAND_MW): band kernel with loader wavefronts            */ \
  X(analysis_threads, 0)     /* host threads of the symbolic analysis (candidate orders); 0 = by the hardware, at most 16      */ \
  X(f1_tiles, 1)             /* row f1 streams column tiles through LDS where the pattern allows; 0: gather kernel          */ \
  X(batch_layout, 0)         /* CNL_LAYOUT_*: layout of `vals` at the device-pointer entry points (band handles)            */ \
  X(band_rhs_interleaved, 0) /* measurement: the band kernels also take `rhs` interleaved (batch_layout = 1 handles)        */

struct Tuning {
#define X(name, dflt) int32_t name = dflt;
  CNL_TUNING_INT_FIELDS(X)
#undef X
  int64_t staged_max_batch = 4096;   // CNL_PLAN_AUTO: largest batch planned for latency
  char force_order[32] = {0};        // name of an ordering candidate to force ("" = none)
};

// sets one field by name; false: no such key
inline bool tuning_set(Tuning& t, const std::string& key, long long value) {
#define X(name, dflt) if (key == #name) { t.name = (int32_t)value; return true; }
  CNL_TUNING_INT_FIELDS(X)
#undef X
  if (key == "staged_max_batch") { t.staged_max_batch = value; return true; }
  return false;
}

// "key=value[,key=value...]" (separators: comma, semicolon, blank); returns "" or what is wrong
inline std::string tuning_parse(Tuning& t, const char* text) {
  std::string s(text ? text : "");
  size_t i = 0;
  while (i < s.size()) {
    while (i < s.size() && (s[i] == ',' || s[i] == ';' || s[i] == ' ')) i++;
    if (i >= s.size()) break;
    size_t j = i;
    while (j < s.size() && s[j] != ',' && s[j] != ';' && s[j] != ' ') j++;
    const std::string item = s.substr(i, j - i);
    i = j;
    const size_t eq = item.find('=');
    if (eq == std::string::npos || eq == 0 || eq + 1 >= item.size()) return "cnl_options.tuning: expected key=value, got '" + item + "'";
    const std::string key = item.substr(0, eq), val = item.substr(eq + 1);
    if (key == "force_order") {
      std::memset(t.force_order, 0, sizeof(t.force_order));
      std::strncpy(t.force_order, val.c_str(), sizeof(t.force_order) - 1);
      continue;
    }
    char* end = nullptr;
    const long long v = std::strtoll(val.c_str(), &end, 0);
    if (!end || *end != 0) return "cnl_options.tuning: value of '" + key + "' is not an integer";
    if (!tuning_set(t, key, v)) return "cnl_options.tuning: unknown key '" + key + "'";
  }
  return "";
}

}  // namespace cnl
