// C entry point over csrc/ba_front_plan.h for tests/test_front_plan.py (host only, no HIP): the plan in its flat form.
#include "../../sfm_danpipeline_amd/csrc/ba_front_plan.h"
#include <cstring>

extern "C" int fplan_build_flat(int nc, const unsigned long long* adj, int wpr, int leaf_cols, int* header /*8*/, int* ints, int ints_cap,
                                int* up_order, int order_cap) {
  fplan::Plan P = fplan::build_plan(nc, adj, wpr, leaf_cols);
  header[0] = P.ok ? 1 : 0;
  if (!P.ok) return 0;
  fplan::Flat fl = fplan::flatten(P);
  header[1] = fl.n_fronts;
  header[2] = fl.levels;
  header[3] = fl.max_T;
  header[4] = (int)fl.ints.size();
  header[5] = (int)fl.n_doubles;
  header[6] = P.chain_blocks;
  header[7] = P.chain_tiles;
  if ((int)fl.ints.size() > ints_cap || fl.n_fronts > order_cap) return -1;
  memcpy(ints, fl.ints.data(), fl.ints.size() * sizeof(int));
  memcpy(up_order, fl.up_order.data(), fl.up_order.size() * sizeof(int));
  return 0;
}
